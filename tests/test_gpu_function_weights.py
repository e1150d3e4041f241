"""Per-element error-function weights: errorFunctionWeights[iBatch][weightsMap[iErr]] of the batched driver
(pymomentum/tensor_ik/tensor_ik.cpp:100-101,137-138; buildMomentumErrorFunctions gives every error function of element
iBatch setWeight(that entry), tensor_ik_utility.cpp:162-177) -- mmx_constraint_data::function_weights [B][C], columns
position, orientation, limits, model parameters, joint block 0, ... -- against the oracle building element b's error
functions with element b's weights, on every route; a zero entry switches the block off for that element like
weightsMap[iErr] < 0 does; and the fold the INTEGRATION.md adapter may use instead for the joint-constraint blocks
(per-constraint weight x per-element function weight) gives the same rows bit for bit."""
import numpy as np
import pytest

from momentum_amd import _abi, capi, humanoid72_landmark_joints, make_humanoid72
from momentum_amd._abi import GnOptions, ParameterLimit
from tests.helpers import make_problem
from tests.test_oracle_joint_blocks import make_block

pytestmark = pytest.mark.gpu
UNIT = 0.01


def _setup(torch, orc, B, seed, with_blocks, host=False):
    from tests.test_gpu_joint_blocks import _device_block

    rig = make_humanoid72(unit=UNIT)
    lm = humanoid72_landmark_joints(rig)
    rng = np.random.default_rng(seed)
    cons, th0, _ = make_problem(rig, lm, lm, B, seed=seed, perturb=0.3, weights="random")
    blocks = [make_block(_abi.MMX_JC_PLANE, rng.choice(rig.num_joints, size=5), rng, weight=1.0, batch=B)] if with_blocks else []
    limits = [ParameterLimit.minmax(int(p), -0.2, 0.2, 1.0) for p in range(6, 14)]
    P = rig.num_params
    mp_t = rng.uniform(-0.2, 0.2, size=(B, P)).astype(np.float32)
    mp_w = rng.uniform(0.0, 1.0, size=(B, P)).astype(np.float32)
    C = 4 + len(blocks)
    fw = rng.uniform(0.3, 3.0, size=(B, C)).astype(np.float32)
    fw[1, 1] = 0.0  # element 1: orientation block off (weightsMap < 0 -> weight 0, tensor_ik_utility.cpp:176)
    fw[2, 2] = 0.0  # element 2: limit block off
    if with_blocks:
        fw[3, 4] = 0.0  # element 3: plane block off
    full = orc.Constraints(cons.pos_parent, cons.pos_offset, cons.pos_target, cons.pos_weight, cons.ori_parent, cons.ori_offset, cons.ori_target,
                           cons.ori_weight, limits=limits, limit_function_weight=2.0, model_target=mp_t, model_weights=mp_w, model_function_weight=0.5,
                           joint_blocks=blocks, function_weights=fw)  # fmt: skip
    pb = capi.Problem(capi.RigHandle(rig, 0), B, cons.pos_parent, cons.ori_parent)
    if host:
        t = lambda a, shp: np.ascontiguousarray(a, np.float32).reshape(shp)
        gb = blocks
    else:
        t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(pb.device)
        gb = [_device_block(torch, k, pb.device) for k in blocks]
    pb.set_constraints(t(cons.pos_offset, (B, cons.Kp, 3)), t(cons.pos_target, (B, cons.Kp, 3)), t(cons.pos_weight, (B, cons.Kp)),
                       t(cons.ori_offset, (B, cons.Ko, 4)), t(cons.ori_target, (B, cons.Ko, 4)), t(cons.ori_weight, (B, cons.Ko)),
                       limits=limits, limit_function_weight=2.0, model_target=t(mp_t, (B, P)), model_weights=t(mp_w, (B, P)),
                       model_function_weight=0.5, joint_blocks=gb, function_weights=t(fw, (B, C)))  # fmt: skip
    return rig, pb, full, th0, fw, cons


@pytest.mark.parametrize("host", [False, True])
@pytest.mark.parametrize("with_blocks", [False, True])
def test_jacobian_rows_carry_the_elements_weights(torch_cuda, orc, with_blocks, host):
    torch = torch_cuda
    B = 6
    rig, pb, full, th0, fw, cons = _setup(torch, orc, B, 11, with_blocks, host)
    theta = np.random.default_rng(5).uniform(-0.3, 0.3, size=(B, rig.num_params)).astype(np.float32)
    jac, res, err = pb.eval_jacobian(torch.from_numpy(theta).to(pb.device))
    jac, res, err = jac.cpu().numpy(), res.cpu().numpy(), err.cpu().numpy()
    for b in range(B):
        J, r, e = orc.eval_jacobian(rig, full.instance(b), theta[b].astype(np.float64), dtype="f64")
        scale = max(1.0, np.abs(J).max())
        assert np.abs(jac[b].T - J).max() <= 3e-5 * scale, (b, np.abs(jac[b].T - J).max())
        assert np.abs(res[b] - r).max() <= 3e-5 * max(1.0, np.abs(r).max())
        assert abs(err[b] - e) <= 1e-5 * max(1.0, abs(e))
    # switched-off blocks leave exactly zero rows for their element only
    Kp, Ko = cons.Kp, cons.Ko
    assert np.all(jac[1].T[3 * Kp : 3 * Kp + 9 * Ko] == 0) and np.any(jac[0].T[3 * Kp : 3 * Kp + 9 * Ko] != 0)
    nb = 5 if with_blocks else 0
    lim0 = 3 * Kp + 9 * Ko + nb
    assert np.all(res[2][lim0 : lim0 + 8] == 0)
    if with_blocks:
        assert np.all(jac[3].T[3 * Kp + 9 * Ko : lim0] == 0) and np.any(jac[0].T[3 * Kp + 9 * Ko : lim0] != 0)


@pytest.mark.parametrize("route", ["fused", "wide", "explicit_jacobian"])
@pytest.mark.parametrize("with_blocks", [False, True])
def test_solve_with_per_element_weights_matches_the_oracle(torch_cuda, orc, with_blocks, route):
    from tests.test_gpu_parity import _sensitivity

    torch = torch_cuda
    B = 6
    rig, pb, full, th0, fw, cons = _setup(torch, orc, B, 23, with_blocks)
    pb.set_route(route)
    for opt in (
        GnOptions.make(min_iterations=8, max_iterations=8, regularization=0.05),
        GnOptions.make(min_iterations=8, max_iterations=8, regularization=0.05, do_line_search=2),
    ):
        out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
        assert pb.last_route() == route
        ref = orc.solve_batch(rig, full, th0, opt, dtype="f64")
        th = out["theta"].cpu().numpy()
        rel = np.linalg.norm(th - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
        tol = np.maximum(1e-5, 3.0 * _sensitivity(orc, rig, full, th0, opt, ref))
        assert np.all(rel <= tol), (route, rel, tol)
        assert np.all(out["status"].cpu().numpy() & 3 == 0)
        h = out["error_history"].cpu().numpy()
        assert np.abs(h - ref["error_history"]).max() <= 1e-4 * max(1.0, np.abs(ref["error_history"]).max())
    # the weights matter: without them the answer differs
    plain = orc.Constraints(cons.pos_parent, cons.pos_offset, cons.pos_target, cons.pos_weight, cons.ori_parent, cons.ori_offset, cons.ori_target,
                            cons.ori_weight, limits=full.limits, limit_function_weight=2.0, model_target=full.model_target, model_weights=full.model_weights,
                            model_function_weight=0.5, joint_blocks=full.joint_blocks)  # fmt: skip
    ref0 = orc.solve_batch(rig, plain, th0, opt, dtype="f64")
    assert np.abs(ref0["theta"] - ref["theta"]).max() > 1e-3


def test_adapter_fold_for_the_joint_constraint_blocks_is_exact(torch_cuda, orc):
    """weight_ of a joint error function multiplies the constraint weight (joint_error_function-inl.h:197-213:
    wgt = constraint.weight * loss'(...) * this->weight_); folding element b's function weight into its per-constraint
    weights and leaving the function weight at 1 is the same product: bit-identical rows."""
    torch = torch_cuda
    B = 5
    rig = make_humanoid72(unit=UNIT)
    lm = humanoid72_landmark_joints(rig)
    cons, th0, _ = make_problem(rig, lm, lm, B, seed=3, perturb=0.3, weights="random")
    fw = np.random.default_rng(1).uniform(0.3, 3.0, size=(B, 2)).astype(np.float32)
    t = lambda pb, a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(pb.device)

    def run(pw, ow, fwt):
        pb = capi.Problem(capi.RigHandle(rig, 0), B, cons.pos_parent, cons.ori_parent)
        pb.set_constraints(t(pb, cons.pos_offset, (B, cons.Kp, 3)), t(pb, cons.pos_target, (B, cons.Kp, 3)), t(pb, pw, (B, cons.Kp)),
                           t(pb, cons.ori_offset, (B, cons.Ko, 4)), t(pb, cons.ori_target, (B, cons.Ko, 4)), t(pb, ow, (B, cons.Ko)),
                           function_weights=None if fwt is None else t(pb, fwt, (B, 2)))  # fmt: skip
        theta = torch.from_numpy(np.random.default_rng(9).uniform(-0.3, 0.3, size=(B, rig.num_params)).astype(np.float32)).to(pb.device)
        j, r, e = pb.eval_jacobian(theta)
        return j.cpu().numpy(), r.cpu().numpy(), e.cpu().numpy()

    a = run(cons.pos_weight, cons.ori_weight, fw)
    b = run(cons.pos_weight * fw[:, :1], cons.ori_weight * fw[:, 1:2], None)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


@pytest.mark.parametrize("line_search,step_rule", [(0, 0), (2, 0), (0, 1)])
def test_lazy_and_by_value_argument_forms_of_the_solve_agree(torch_cuda, orc, line_search, step_rule):
    """The four-workgroup instantiations of the one-launch solve exist twice (mmx_fused.hip, kArgLazy): descriptors read lazily
    from the problem's device struct (shared rig, shared weights: the production form), or passed by value (per-element rigs /
    weights: the selections write into the copies).  Unit function weights send the same problem down the by-value form; the
    two must give the same answer -- same source, same arithmetic: bit for bit -- for the plain, the line-search (generic) and
    the LM instantiation."""
    torch = torch_cuda
    from momentum_amd._abi import MMX_STEP_LM_SCHEDULE

    B = 64
    rig = make_humanoid72(unit=UNIT)
    lm = humanoid72_landmark_joints(rig)
    cons, th0, _ = make_problem(rig, lm, lm, B, seed=11, perturb=0.3, weights="random")
    opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=0.05, do_line_search=line_search,
                         step_rule=MMX_STEP_LM_SCHEDULE if step_rule else 0)  # fmt: skip
    t = lambda pb, a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(pb.device)

    def run(fwt):
        pb = capi.Problem(capi.RigHandle(rig, 0), B, cons.pos_parent, cons.ori_parent)
        pb.set_route("fused")
        pb.set_constraints(t(pb, cons.pos_offset, (B, cons.Kp, 3)), t(pb, cons.pos_target, (B, cons.Kp, 3)), t(pb, cons.pos_weight, (B, cons.Kp)),
                           t(pb, cons.ori_offset, (B, cons.Ko, 4)), t(pb, cons.ori_target, (B, cons.Ko, 4)), t(pb, cons.ori_weight, (B, cons.Ko)),
                           function_weights=None if fwt is None else t(pb, fwt, (B, 2)))  # fmt: skip
        out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
        torch.cuda.synchronize()
        return out["theta"].cpu().numpy(), out["error_history"].cpu().numpy(), out["status"].cpu().numpy()

    lazy = run(None)
    by_value = run(np.ones((B, 2), np.float32))
    assert np.all(lazy[2] & 3 == 0) and np.array_equal(lazy[2], by_value[2])
    assert np.array_equal(lazy[0], by_value[0]), float(np.abs(lazy[0] - by_value[0]).max())
    assert np.array_equal(lazy[1], by_value[1])
    if not line_search and not step_rule:  # ... and it is the right answer (no discrete decisions in the plain rule)
        ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64")
        rel = np.linalg.norm(lazy[0] - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
        assert rel.max() <= 1e-5, rel.max()
