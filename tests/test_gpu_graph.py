"""mmx_solve inside a captured HIP graph (torch.cuda.CUDAGraph -> hipStreamBeginCapture on torch's current stream).

The solve makes no host round trip on its stream -- the one-launch route is a single kernel, MMX_PRECISION_AUTO compacts the
marked elements and sizes its second and third pass on the device (include/mmx.h, MMX_PRECISION_AUTO), the wide route is a
fixed sequence of launches per iteration -- so after one warm-up call (scratch buffers, LDS limits) the whole call can be
captured once and replayed: the small-batch wide solve (some eighty short launches) is where that pays, and a caller that
embeds the solve in its own graph needs it to be legal.  Replays must reproduce the eager call bit for bit, also on new
parameter values written into the captured buffer.
"""
import numpy as np
import pytest

from momentum_amd import capi, humanoid72_landmark_joints, make_humanoid72
from momentum_amd._abi import MMX_PRECISION_AUTO, MMX_PRECISION_F64, MMX_PRECISION_MIXED, MMX_SOLVE_MIXED, MMX_STEP_LM_SCHEDULE, GnOptions
from tests.helpers import make_problem

pytestmark = pytest.mark.gpu
UNIT = 0.01


def _problem(torch, rig, cons, B):
    pb = capi.Problem(capi.RigHandle(rig, 0), B, cons.pos_parent, cons.ori_parent)
    t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(pb.device)
    pb.set_constraints(t(cons.pos_offset, (B, cons.Kp, 3)), t(cons.pos_target, (B, cons.Kp, 3)), t(cons.pos_weight, (B, cons.Kp)),
                       t(cons.ori_offset, (B, cons.Ko, 4)), t(cons.ori_target, (B, cons.Ko, 4)), t(cons.ori_weight, (B, cons.Ko)))  # fmt: skip
    return pb


CASES = {
    # name: (route, options)
    "one_launch_f32": ("fused", dict(regularization=0.05)),
    "one_launch_lm_schedule": ("fused", dict(regularization=0.05, step_rule=MMX_STEP_LM_SCHEDULE)),
    "mixed": ("fused", dict(regularization=0.01, do_line_search=2, precision=MMX_PRECISION_MIXED)),
    # lambda = 1e-3: the single-precision pass marks (nearly) every element -> second pass (mixed) and third (double) run inside the graph
    "auto_with_marked_elements": ("fused", dict(regularization=1e-3, precision=MMX_PRECISION_AUTO)),
    "double_instantiation": ("fused", dict(regularization=0.05, precision=MMX_PRECISION_F64)),
    # every element marked (a bound nothing passes), the mixed pass converges on all of them: second pass only
    "auto_all_marked_mixed_only": ("fused", dict(regularization=0.05, precision=MMX_PRECISION_AUTO, precision_bound=1e-12)),
    "auto_nothing_marked": ("fused", dict(regularization=0.05, precision=MMX_PRECISION_AUTO)),
    "wide_route": ("wide", dict(regularization=0.05)),
    "wide_route_line_search": ("wide", dict(regularization=0.05, do_line_search=2)),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_solve_replays_from_a_captured_graph(torch_cuda, case):
    torch = torch_cuda
    route, kw = CASES[case]
    B = 128
    rig = make_humanoid72(seed=12345, variant="p128", unit=UNIT)
    lm = humanoid72_landmark_joints(rig)
    cons, th0, _ = make_problem(rig, lm, lm, B, seed=4242, perturb=0.3)
    _, th1, _ = make_problem(rig, lm, lm, B, seed=4243, perturb=0.25)
    pb = _problem(torch, rig, cons, B)
    pb.set_route(route)
    opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, **kw)
    dev = pb.device
    outs = lambda: dict(error=torch.empty((B,), dtype=torch.float64, device=dev), iterations=torch.empty((B,), dtype=torch.int32, device=dev),
                        status=torch.empty((B,), dtype=torch.int32, device=dev), error_history=torch.empty((B, 10), dtype=torch.float64, device=dev))  # fmt: skip

    def eager(th):
        o = outs()
        t = torch.from_numpy(th.copy()).to(dev)
        pb.solve(t, opt, outputs=o)
        torch.cuda.synchronize()
        return {k: v.cpu().numpy() for k, v in o.items()}

    ref0, ref1 = eager(th0), eager(th1)  # (also the warm-up: scratch buffers and LDS limits are set before the capture)
    if case != "double_instantiation":  # (the double kernel is no route of the single-precision solve)
        assert pb.last_route() == route
    if case == "auto_with_marked_elements":
        assert int((ref0["status"] & MMX_SOLVE_MIXED != 0).sum()) > B // 2
    theta = torch.from_numpy(th0.copy()).to(dev)
    theta_in = theta.clone()
    go = outs()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            theta.copy_(theta_in)
            pb.solve(theta, opt, outputs=go)
    torch.cuda.current_stream().wait_stream(side)
    for th, ref in ((th0, ref0), (th1, ref1), (th0, ref0)):
        theta_in.copy_(torch.from_numpy(th.copy()).to(dev))
        for v in go.values():
            if v is not theta:
                v.zero_()
        g.replay()
        torch.cuda.synchronize()
        for k in ("theta", "error", "iterations", "status", "error_history"):
            assert np.array_equal(go[k].cpu().numpy(), ref[k], equal_nan=(k in ("error", "error_history", "theta"))), (case, k)
