"""More than 512 solved parameters (the reference's kMaxModelParams is 2048, momentum/math/types.h:426-429): the
explicit-Jacobian route takes systems of 513 ... 2048 -- dense J, normal equations on the VALU, unmasked left-looking factor
in HBM, refinement through J (mmx_kernels.hpp kMaxSolved) -- held to the oracle's double solve like every other route;
beyond that the library refuses (MMX_ERR_UNSUPPORTED), it does not fall back."""
import numpy as np
import pytest

from momentum_amd import capi, make_rig300
from momentum_amd._abi import GnOptions
from momentum_amd.rigs import RX, RZ, _build_rig
from tests.helpers import make_problem

pytestmark = pytest.mark.gpu


def _many_parameter_rig(extra_dofs, unit=0.01):
    """make_rig300 (BASELINE configs[4]) with `extra_dofs` rotation parameters on every joint beyond the 72-joint body
    instead of one on 172 of them: 128 + 228 * extra_dofs parameters."""
    base = make_rig300(unit=unit)  # (metres: lambda = 0.05 stays above the factor's damping floor, kFactorDamping x the mean diagonal)
    J0, J = 72, base.num_joints
    trip = []
    for r in range(7 * J0):
        for k in range(base.pt_outer[r], base.pt_outer[r + 1]):
            trip.append((r, int(base.pt_inner[k]), float(base.pt_value[k])))
    names = list(base.param_names[:128])
    assert max(c for _, c, _ in trip) == 127
    for j in range(J0, J):
        for d in [RX, RX + 1, RZ, 0, 1, 2, 6][:extra_dofs]:  # rotations first, then translations, then the scale
            trip.append((j * 7 + d, len(names), 1.0))
            names.append(f"{base.joint_names[j]}_d{d}")
    return _build_rig(base.parent, base.pre_rotation, base.translation_offset, trip, len(names), list(base.joint_names), names)


def _upload(torch, pb, cons, B):
    dev = pb.device
    t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(dev)
    pb.set_constraints(
        t(cons.pos_offset, (B, cons.Kp, 3)), t(cons.pos_target, (B, cons.Kp, 3)), t(cons.pos_weight, (B, cons.Kp)),
        t(cons.ori_offset, (B, cons.Ko, 4)), t(cons.ori_target, (B, cons.Ko, 4)), t(cons.ori_weight, (B, cons.Ko)),
        cons.pos_function_weight, cons.ori_function_weight,
    )  # fmt: skip


@pytest.fixture(scope="module")
def problem812(torch_cuda, orc):
    torch = torch_cuda
    rig = _many_parameter_rig(3)  # 128 + 684 = 812 parameters
    assert rig.num_params == 812
    joints = np.arange(rig.num_joints, dtype=np.int32)  # a position and an orientation constraint on every joint: M = 3600 rows
    B = 3
    cons, th0, _ = make_problem(rig, joints, joints, B, seed=77, perturb=0.1)
    rh = capi.RigHandle(rig, 0)
    pb = capi.Problem(rh, B, cons.pos_parent, cons.ori_parent)
    _upload(torch, pb, cons, B)
    return torch, rig, cons, th0, pb, B


def test_normal_equations_of_812_parameters(problem812, orc):
    torch, rig, cons, th0, pb, B = problem812
    assert pb.n > 512
    jtj, jtr, err = pb.normal_equations(torch.from_numpy(th0.copy()).to(pb.device))
    jtj, jtr = jtj.cpu().numpy(), jtr.cpu().numpy()
    opt1 = GnOptions.make(min_iterations=1, max_iterations=1, threshold=1.0, regularization=0.05)
    for b in range(B):
        ref = orc.solve(rig, cons.subset(np.array([b])), th0[b], opt1, dtype="f64")
        H, g = ref["jtj"], ref["jtr"]
        assert H.shape == jtj[b].shape == (812, 812)
        scale = np.abs(H).max()
        assert np.abs(jtj[b] - H).max() <= 2e-5 * scale
        assert np.abs(jtr[b] - g).max() <= 2e-5 * np.abs(g).max()


def test_solve_with_812_parameters_matches_the_double_oracle(problem812, orc):
    torch, rig, cons, th0, pb, B = problem812
    opt = GnOptions.make(min_iterations=5, max_iterations=5, threshold=1.0, regularization=0.05)
    out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
    assert pb.last_route() == "explicit_jacobian"
    ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64", nthreads=4)
    th = out["theta"].cpu().numpy().astype(np.float64)
    rel = np.linalg.norm(th - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
    assert np.all((out["status"].cpu().numpy() & 3) == 0)
    assert np.array_equal(out["iterations"].cpu().numpy(), ref["iterations"])
    h, href = out["error_history"].cpu().numpy(), ref["error_history"]
    assert np.all(np.abs(h - href) <= 1e-4 * np.abs(href) + 1e-7 * href[:, :1])
    assert rel.max() <= 1e-5, rel


def test_solve_with_1496_parameters_near_the_limit(torch_cuda, orc):
    """128 + 228 x 6 parameters (rotations and translations of every extra joint): the normal equations' 16-row chunks,
    the factor's 94-tile panel and the refinement's chunk of J at the edge of one workgroup's LDS."""
    torch = torch_cuda
    rig = _many_parameter_rig(6)
    assert rig.num_params == 1496
    joints = np.arange(rig.num_joints, dtype=np.int32)
    cons, th0, _ = make_problem(rig, joints, joints, 1, seed=78, perturb=0.05)
    rh = capi.RigHandle(rig, 0)
    pb = capi.Problem(rh, 1, cons.pos_parent, cons.ori_parent)
    _upload(torch, pb, cons, 1)
    opt = GnOptions.make(min_iterations=3, max_iterations=3, threshold=1.0, regularization=0.05)
    out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
    assert pb.last_route() == "explicit_jacobian" and pb.n == 1496
    ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64")
    r32 = orc.solve_batch(rig, cons, th0, opt, dtype="f32")
    dist = lambda th: np.linalg.norm(th.astype(np.float64) - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
    rel, rel32 = dist(out["theta"].cpu().numpy()), dist(r32["theta"])
    print("rel", rel, "float oracle", rel32)
    assert np.all((out["status"].cpu().numpy() & 3) == 0)
    h, href = out["error_history"].cpu().numpy(), ref["error_history"]
    assert np.all(np.abs(h - href) <= 1e-4 * np.abs(href) + 1e-7 * href[:, :1])
    assert rel.max() <= 1e-5, (rel, rel32)


def _every_joint_dof_rig(skip=()):
    """make_rig300's skeleton with one model parameter per joint parameter (300 x 7 = 2100) except the rows in `skip`."""
    base = make_rig300(unit=0.01)
    J = base.num_joints
    rows = [r for r in range(7 * J) if r not in set(skip)]
    trip = [(r, c, 1.0) for c, r in enumerate(rows)]
    names = [f"{base.joint_names[r // 7]}_{'tx ty tz rx ry rz sc'.split()[r % 7]}" for r in rows]
    return _build_rig(base.parent, base.pre_rotation, base.translation_offset, trip, len(rows), list(base.joint_names), names)


@pytest.fixture(scope="module")
def problem2048(torch_cuda):
    torch = torch_cuda
    rig = _every_joint_dof_rig(skip=[40 * i + 6 for i in range(52)])  # without 52 of the scales: 2048 = kMaxModelParams
    assert rig.num_params == 2048
    joints = np.arange(rig.num_joints, dtype=np.int32)
    # (offsets of a few centimetres: with a point off the joint's origin its scale is observable)
    cons, th0, _ = make_problem(rig, joints, joints, 1, seed=79, perturb=0.03, random_offsets=True, offset_scale=0.03)
    rh = capi.RigHandle(rig, 0)
    pb = capi.Problem(rh, 1, cons.pos_parent, cons.ori_parent)
    _upload(torch, pb, cons, 1)
    return torch, rig, cons, th0, pb


def test_solve_with_2048_parameters_the_reference_maximum(problem2048, orc):
    """kMaxModelParams (momentum/math/types.h:426-429) solved parameters.  The factor's 128-tile panel leaves the refinement
    eight rows of J per chunk."""
    torch, rig, cons, th0, pb = problem2048
    assert pb.n == 2048
    opt = GnOptions.make(min_iterations=3, max_iterations=3, threshold=1.0, regularization=0.05)
    out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
    assert pb.last_route() == "explicit_jacobian"
    ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64")
    r32 = orc.solve_batch(rig, cons, th0, opt, dtype="f32")
    dist = lambda th: np.linalg.norm(th.astype(np.float64) - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
    rel, rel32 = dist(out["theta"].cpu().numpy()), dist(r32["theta"])
    print("rel", rel, "float oracle", rel32, "history", ref["error_history"])
    assert np.all((out["status"].cpu().numpy() & 3) == 0)
    h, href = out["error_history"].cpu().numpy(), ref["error_history"]
    assert np.all(np.abs(h - href) <= 1e-4 * np.abs(href) + 1e-7 * href[:, :1])
    assert rel.max() <= 1e-5, (rel, rel32)


def test_a_rig_of_more_than_2048_parameters_is_refused(torch_cuda):
    """... like the reference's own bound on a parameter transform (kMaxModelParams): no solve ever sees more."""
    with pytest.raises(capi.MmxError) as e:
        capi.RigHandle(_every_joint_dof_rig(), 0)  # 2100 parameters
    assert e.value.code == 1 and "kMaxModelParams" in str(e.value)  # MMX_ERR_INVALID_ARGUMENT


@pytest.mark.parametrize("line_search", [1, 2])
def test_line_search_with_812_parameters(problem812, orc, line_search):
    """stepUpdateKernel (both backtracking rules) on the explicit-Jacobian route beyond 512 solved parameters."""
    torch, rig, cons, th0, pb, B = problem812
    opt = GnOptions.make(min_iterations=4, max_iterations=4, threshold=1.0, regularization=0.05, do_line_search=line_search)
    out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
    assert pb.last_route() == "explicit_jacobian"
    ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64", nthreads=4)
    th = out["theta"].cpu().numpy().astype(np.float64)
    rel = np.linalg.norm(th - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
    h, href = out["error_history"].cpu().numpy(), ref["error_history"]
    assert np.all((out["status"].cpu().numpy() & 3) == 0)
    assert np.all(np.abs(h - href) <= 1e-4 * np.abs(href) + 1e-7 * href[:, :1])
    assert rel.max() <= 1e-5, rel
