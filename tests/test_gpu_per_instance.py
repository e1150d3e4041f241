"""Per-instance characters and constraint parents (the batched driver's `*characters[iBatch]`,
pymomentum/tensor_ik/tensor_ik.cpp:129,140, and per-element `parents`, tensor_marker_error_function.cpp:
97-98,186): every element of the batch is checked against the CPU oracle run on THAT element's own rig
and parent lists -- J / r / error elementwise, then the solve (1e-5 on the pose parameters), on both solver
paths and with the line search (whose trial errors re-run FK on the element's constants)."""
import copy

import numpy as np
import pytest

from momentum_amd import capi  # noqa: E402  (default_route: which kernels the problems of a test run)

from momentum_amd import humanoid72_landmark_joints, make_humanoid72, make_test_character
from momentum_amd._abi import MMX_PRECISION_MIXED, GnOptions
from tests.helpers import make_problem

pytestmark = pytest.mark.gpu
UNIT = 0.01


def _variants(rig, B, rng, scale=0.25):
    """B characters of rig's topology: per-subject bone lengths (offsets scaled per joint) and perturbed
    pre-rotations."""
    J = rig.num_joints
    off = np.repeat(rig.translation_offset[None], B, axis=0).astype(np.float32)
    off *= (1.0 + scale * rng.uniform(-1, 1, size=(B, J, 1))).astype(np.float32)
    pre = np.repeat(rig.pre_rotation[None], B, axis=0).astype(np.float64)
    pre += 0.2 * rng.normal(size=pre.shape)
    pre /= np.linalg.norm(pre, axis=2, keepdims=True)
    rigs = []
    for b in range(B):
        r = copy.copy(rig)
        r.translation_offset = np.ascontiguousarray(off[b])
        r.pre_rotation = np.ascontiguousarray(pre[b].astype(np.float32))
        rigs.append(r)
    return off, pre.astype(np.float32), rigs


def _upload(torch, pb, cons, B):
    dev = pb.device
    t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(dev)
    pb.set_constraints(
        t(cons.pos_offset, (B, cons.Kp, 3)), t(cons.pos_target, (B, cons.Kp, 3)), t(cons.pos_weight, (B, cons.Kp)),
        t(cons.ori_offset, (B, cons.Ko, 4)), t(cons.ori_target, (B, cons.Ko, 4)), t(cons.ori_weight, (B, cons.Ko)), 1.0, 1.0,
    )  # fmt: skip


def _instance_cons(orc, cons, b, pos_parent, ori_parent):
    return orc.Constraints(pos_parent, cons.pos_offset[b], cons.pos_target[b], cons.pos_weight[b],
                           ori_parent, cons.ori_offset[b], cons.ori_target[b], cons.ori_weight[b])  # fmt: skip


def _check_jacobian(torch, orc, pb, rigs, cons, theta, pos_parents, ori_parents):
    jac, res, err = pb.eval_jacobian(torch.from_numpy(theta).to(pb.device))
    jac, res, err = jac.cpu().numpy(), res.cpu().numpy(), err.cpu().numpy()
    for b in range(theta.shape[0]):
        J, r, e = orc.eval_jacobian(rigs[b], _instance_cons(orc, cons, b, pos_parents[b], ori_parents[b]), theta[b].astype(np.float64), dtype="f64")
        scale = max(1.0, np.abs(J).max())
        assert np.abs(jac[b].T - J).max() <= 2e-5 * scale, (b, np.abs(jac[b].T - J).max())
        assert np.abs(jac[b].T[:, np.abs(J).max(axis=0) == 0]).max(initial=0.0) == 0
        assert np.abs(res[b] - r).max() <= 2e-5 * max(1.0, np.abs(r).max())
        assert abs(err[b] - e) <= 2e-5 * max(1.0, e)


def _check_solve(torch, orc, pb, rigs, cons, th0, pos_parents, ori_parents, opt, tol=1e-5):
    out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
    th = out["theta"].cpu().numpy()
    for b in range(th0.shape[0]):
        ref = orc.solve(rigs[b], _instance_cons(orc, cons, b, pos_parents[b], ori_parents[b]), th0[b], opt, dtype="f64")
        rel = np.linalg.norm(th[b] - ref["theta"]) / np.linalg.norm(ref["theta"])
        assert rel <= tol, (b, rel)
        assert int(out["iterations"][b]) == ref["iterations"] and int(out["status"][b]) & 3 == ref["status"]
        href = np.asarray(ref["error_history"])
        h = out["error_history"][b].cpu().numpy()[: len(href)]
        assert np.abs(h - href).max() <= 1e-4 * max(1.0, np.abs(href).max())


@pytest.mark.parametrize("solver", ["fused", "v1"])
@pytest.mark.parametrize("memory", ["device", "host"])
def test_per_instance_characters(torch_cuda, orc, solver, memory, monkeypatch):

    torch = torch_cuda
    if solver == "v1":
        monkeypatch.setattr(capi, "default_route", "explicit_jacobian")
    rig = make_humanoid72(unit=UNIT)
    lm = humanoid72_landmark_joints(rig)
    B = 6
    rng = np.random.default_rng(31)
    off, pre, rigs = _variants(rig, B, rng)
    # targets: FK(theta*) on each element's OWN character
    conss = [make_problem(rigs[b], lm, lm, 1, seed=100 + b, perturb=0.3)[0] for b in range(B)]
    cat = lambda f: np.concatenate([getattr(c, f) for c in conss], axis=0)
    cons = orc.Constraints(lm, cat("pos_offset"), cat("pos_target"), cat("pos_weight"), lm, cat("ori_offset"), cat("ori_target"), cat("ori_weight"))
    rh = capi.RigHandle(rig, 0)
    pb = capi.Problem(rh, B, lm, lm)
    _upload(torch, pb, cons, B)
    if memory == "device":
        pb.set_instance_rig(torch.from_numpy(off).to(pb.device), torch.from_numpy(pre).to(pb.device))
    else:
        pb.set_instance_rig(off, pre)
    parents = [lm] * B
    theta = rng.uniform(-0.3, 0.3, size=(B, rig.num_params)).astype(np.float32)
    # skeleton state of every element on its own constants
    st = pb.skeleton_state(torch.from_numpy(theta).to(pb.device)).cpu().numpy()
    for b in range(B):
        ref = orc.skeleton_state(rigs[b], theta[b].astype(np.float64), "f64")["world"]
        assert np.abs(st[b] - ref).max() <= 5e-6 * max(1.0, np.abs(ref).max())
    _check_jacobian(torch, orc, pb, rigs, cons, theta, parents, parents)
    th0 = np.zeros((B, rig.num_params), np.float32)
    _check_solve(torch, orc, pb, rigs, cons, th0, parents, parents, GnOptions.make(min_iterations=10, max_iterations=10, regularization=0.05))
    _check_solve(torch, orc, pb, rigs, cons, th0, parents, parents, GnOptions.make(min_iterations=6, max_iterations=6, regularization=0.05, do_line_search=2))
    if solver == "fused":  # the mixed-precision instantiation on every element's own constants: the double run's answer to the float result's last bit
        _check_solve(torch, orc, pb, rigs, cons, th0, parents, parents, GnOptions.make(min_iterations=10, max_iterations=10, regularization=0.05, precision=MMX_PRECISION_MIXED), 2e-7)
    # offsets only (pre-rotations of the shared rig), then back to the shared rig
    pb.set_instance_rig(off, None)
    rigs2 = []
    for b in range(B):
        r = copy.copy(rig)
        r.translation_offset = np.ascontiguousarray(off[b])
        rigs2.append(r)
    _check_jacobian(torch, orc, pb, rigs2, cons, theta, parents, parents)
    pb.set_instance_rig(None, None)
    _check_jacobian(torch, orc, pb, [rig] * B, cons, theta, parents, parents)


@pytest.mark.parametrize("solver", ["fused", "v1"])
@pytest.mark.parametrize("which", ["humanoid72", "chain9"])
def test_per_instance_constraint_parents(torch_cuda, orc, solver, which, monkeypatch):

    torch = torch_cuda
    if solver == "v1":
        monkeypatch.setattr(capi, "default_route", "explicit_jacobian")
    rng = np.random.default_rng(57)
    if which == "humanoid72":
        rig = make_humanoid72(unit=UNIT)
        B, Kp, Ko = 7, 14, 9
    else:
        rig = make_test_character(9)
        B, Kp, Ko = 5, 4, 3
    J = rig.num_joints
    # every element constrains different joints (repeats allowed; element 0 uses a small subtree only)
    pos_parents = [rng.choice(J, size=Kp, replace=True).astype(np.int32) for _ in range(B)]
    ori_parents = [rng.choice(J, size=Ko, replace=True).astype(np.int32) for _ in range(B)]
    pos_parents[0][:] = pos_parents[0][0]
    conss = [make_problem(rig, pos_parents[b], ori_parents[b], 1, seed=300 + b, perturb=0.3, random_offsets=True, weights="random")[0] for b in range(B)]
    cat = lambda f: np.concatenate([getattr(c, f) for c in conss], axis=0)
    cons = orc.Constraints(pos_parents[0], cat("pos_offset"), cat("pos_target"), cat("pos_weight"), ori_parents[0], cat("ori_offset"), cat("ori_target"), cat("ori_weight"))
    rh = capi.RigHandle(rig, 0)
    pb = capi.Problem(rh, B, pos_parents[0], ori_parents[0])  # the batch-shared lists are overridden below
    _upload(torch, pb, cons, B)
    pp, op = np.stack(pos_parents), np.stack(ori_parents)
    pb.set_instance_parents(torch.from_numpy(pp).to(pb.device), torch.from_numpy(op).to(pb.device))
    theta = rng.uniform(-0.3, 0.3, size=(B, rig.num_params)).astype(np.float32)
    rigs = [rig] * B
    _check_jacobian(torch, orc, pb, rigs, cons, theta, pos_parents, ori_parents)
    th0 = np.zeros((B, rig.num_params), np.float32)
    tol = 1e-5 if which == "humanoid72" else 5e-5  # (the under-determined chain fixture amplifies fp32 input rounding, see test_gpu_parity)
    _check_solve(torch, orc, pb, rigs, cons, th0, pos_parents, ori_parents, GnOptions.make(min_iterations=10, max_iterations=10, regularization=0.05), tol)
    _check_solve(torch, orc, pb, rigs, cons, th0, pos_parents, ori_parents, GnOptions.make(min_iterations=6, max_iterations=6, regularization=0.05, do_line_search=1), tol)
    if solver == "fused":  # per-element unit tables (buildInstanceUnitTables) under the mixed-precision instantiation's double adjoint pass
        _check_solve(torch, orc, pb, rigs, cons, th0, pos_parents, ori_parents, GnOptions.make(min_iterations=10, max_iterations=10, regularization=0.05, precision=MMX_PRECISION_MIXED), 2e-7)
    # a disabled parameter set on top, then host lists for the positions only (orientation back to shared)
    en = np.ones(rig.num_params, np.uint8)
    en[[1, 4]] = 0
    pb.set_enabled(en)
    pb.set_instance_parents(pp, None)
    jac, res, err = pb.eval_jacobian(torch.from_numpy(theta).to(pb.device))
    for b in range(B):
        Jr, r, e = orc.eval_jacobian(rig, _instance_cons(orc, cons, b, pos_parents[b], ori_parents[0]), theta[b].astype(np.float64), enabled=en, dtype="f64")
        assert np.abs(jac[b].cpu().numpy().T - Jr).max() <= 2e-5 * max(1.0, np.abs(Jr).max())
    # out-of-range joint: loud error, nothing modified
    bad = pp.copy()
    bad[2, 1] = J
    with pytest.raises(capi.MmxError):
        pb.set_instance_parents(bad, None)
    jac2, _, _ = pb.eval_jacobian(torch.from_numpy(theta).to(pb.device))
    assert torch.equal(jac, jac2)


def test_per_instance_characters_on_the_wide_path(torch_cuda, orc):
    """Per-element rig constants on a problem that takes the wide path (P = 219, constraints on every joint: tree normal
    equations -> tiled factor -> tree refinement): every element against the oracle on its own character, with and
    without the line search (whose trial errors re-run FK on the element's constants)."""

    torch = torch_cuda
    rig = make_humanoid72(variant="p219", unit=UNIT)
    allj = np.arange(rig.num_joints, dtype=np.int32)
    B = 4
    rng = np.random.default_rng(37)
    off, pre, rigs = _variants(rig, B, rng, scale=0.15)
    conss = [make_problem(rigs[b], allj, allj, 1, seed=200 + b, perturb=0.25)[0] for b in range(B)]
    cat = lambda f: np.concatenate([getattr(c, f) for c in conss], axis=0)
    cons = orc.Constraints(allj, cat("pos_offset"), cat("pos_target"), cat("pos_weight"), allj, cat("ori_offset"), cat("ori_target"), cat("ori_weight"))
    pb = capi.Problem(capi.RigHandle(rig, 0), B, allj, allj)
    _upload(torch, pb, cons, B)
    pb.set_instance_rig(torch.from_numpy(off).to(pb.device), torch.from_numpy(pre).to(pb.device))
    parents = [allj] * B
    th0 = np.zeros((B, rig.num_params), np.float32)
    _check_solve(torch, orc, pb, rigs, cons, th0, parents, parents, GnOptions.make(min_iterations=8, max_iterations=8, regularization=0.05))
    _check_solve(torch, orc, pb, rigs, cons, th0, parents, parents, GnOptions.make(min_iterations=6, max_iterations=6, regularization=0.05, do_line_search=2))


@pytest.mark.parametrize("route", ["tree", "dense"])
def test_per_instance_constraint_parents_on_the_wide_path(torch_cuda, orc, route, monkeypatch):
    """Per-element constraint parents on a problem of 219 solved parameters (wide path): every element constrains its own
    60 + 30 joints; the tree kernels build the element's units-per-joint lists in LDS like the fused solve does
    (buildInstanceUnitTables), the dense route reads the per-element parents in the J assembly."""

    torch = torch_cuda
    if route == "dense":
        monkeypatch.setattr(capi, "default_route", "explicit_jacobian")
    rig = make_humanoid72(variant="p219", unit=UNIT)
    J, B, Kp, Ko = rig.num_joints, 4, 60, 30
    rng = np.random.default_rng(59)
    pos_parents = [rng.choice(J, size=Kp, replace=True).astype(np.int32) for _ in range(B)]
    ori_parents = [rng.choice(J, size=Ko, replace=True).astype(np.int32) for _ in range(B)]
    conss = [make_problem(rig, pos_parents[b], ori_parents[b], 1, seed=400 + b, perturb=0.25, random_offsets=True, weights="random")[0] for b in range(B)]
    cat = lambda f: np.concatenate([getattr(c, f) for c in conss], axis=0)
    cons = orc.Constraints(pos_parents[0], cat("pos_offset"), cat("pos_target"), cat("pos_weight"), ori_parents[0], cat("ori_offset"), cat("ori_target"), cat("ori_weight"))
    pb = capi.Problem(capi.RigHandle(rig, 0), B, pos_parents[0], ori_parents[0])
    _upload(torch, pb, cons, B)
    pb.set_instance_parents(torch.from_numpy(np.stack(pos_parents)).to(pb.device), torch.from_numpy(np.stack(ori_parents)).to(pb.device))
    rigs = [rig] * B
    th0 = np.zeros((B, rig.num_params), np.float32)
    _check_solve(torch, orc, pb, rigs, cons, th0, pos_parents, ori_parents, GnOptions.make(min_iterations=8, max_iterations=8, regularization=0.05), 2e-5)
    _check_solve(torch, orc, pb, rigs, cons, th0, pos_parents, ori_parents, GnOptions.make(min_iterations=6, max_iterations=6, regularization=0.05, do_line_search=2), 2e-5)
