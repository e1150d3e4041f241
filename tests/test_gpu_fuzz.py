"""Randomised rigs on the GPU: random trees (chains, stars, bushy), parameter transforms with shared
parameters, translation / scale dofs, rows with three entries (no two-slot ELL copy -> CSR walk),
non-zero transform offsets, random constraint sets, weights and enabled masks -- J / r / error and a
short solve against the CPU oracle.  Catches indexing mistakes in the host-built tables (DFS
intervals, column sources, term records, limit tables) that the fixed fixtures cannot."""
import os

import numpy as np
import pytest

from momentum_amd import capi  # noqa: E402  (default_route: which kernels the problems of a test run)

from momentum_amd._abi import GnOptions, ParameterLimit
from momentum_amd.rigs import _build_rig
from tests.helpers import make_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU (run with -m gpu on the MI355X box)")
    return torch


def random_rig(rng, J, shape):
    parent = [-1]
    for j in range(1, J):
        if shape == "chain":
            parent.append(j - 1)
        elif shape == "star":
            parent.append(0 if rng.uniform() < 0.7 else int(rng.integers(0, j)))
        else:
            parent.append(int(rng.integers(max(0, j - 6), j)))
    pre = np.zeros((J, 4), np.float32)
    for j in range(J):
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        ang = rng.uniform(-0.4, 0.4)
        pre[j] = [*(np.sin(ang / 2) * ax), np.cos(ang / 2)]
    off = rng.uniform(-0.3, 0.3, size=(J, 3)).astype(np.float32)
    trip, names = [], []

    def new_param(name):
        names.append(name)
        return len(names) - 1

    for d in range(6):  # root: rigid motion
        trip.append((d, new_param(f"root{d}"), 1.0))
    trip.append((6, new_param("scale"), 1.0))
    for j in range(1, J):
        for d in (3, 4, 5):
            if rng.uniform() < 0.75:
                trip.append((7 * j + d, new_param(f"j{j}r{d}"), float(rng.uniform(0.5, 1.5))))
        if rng.uniform() < 0.15:
            trip.append((7 * j + int(rng.integers(0, 3)), new_param(f"j{j}t"), 1.0))
        if rng.uniform() < 0.1:
            trip.append((7 * j + 6, new_param(f"j{j}s"), 0.5))
    # shared parameters: each drives one rotation dof of several joints (some rows end up with 2-3 entries)
    for s in range(max(1, J // 6)):
        p = new_param(f"shared{s}")
        for j in rng.choice(np.arange(1, J), size=min(J - 1, int(rng.integers(2, 6))), replace=False):
            trip.append((7 * int(j) + int(rng.integers(3, 6)), p, float(rng.uniform(-1.0, 1.0))))
    rig = _build_rig(parent, pre, off, trip, len(names), [f"j{j}" for j in range(J)], names)
    rig.pt_offsets[:] = (rng.uniform(-0.05, 0.05, size=7 * J) * (rng.uniform(size=7 * J) < 0.2)).astype(np.float32)
    return rig


@pytest.mark.parametrize("seed", range(int(os.environ.get("MMX_FUZZ_SEEDS", "48"))))
def test_random_rig_matches_oracle(torch_cuda, orc, seed, monkeypatch):

    torch = torch_cuda
    if seed % 4 == 3:
        monkeypatch.setattr(capi, "default_route", "explicit_jacobian")  # every fourth rig through the three-kernel path
    rng = np.random.default_rng(1000 + seed)
    J = int(rng.integers(2, int(os.environ.get("MMX_FUZZ_JMAX", "48"))))  # (MMX_FUZZ_JMAX=100: the mid-size instantiations and the route switch)
    rig = random_rig(rng, J, ["chain", "star", "bushy"][seed % 3])
    P = rig.num_params
    Kp, Ko = int(rng.integers(0, 9)), int(rng.integers(0, 6))
    if Kp + Ko == 0:
        Kp = 1
    pp = rng.integers(0, J, size=Kp).astype(np.int32)
    op = rng.integers(0, J, size=Ko).astype(np.int32)
    B = 3
    cons, th0, ths = make_problem(rig, pp, op, B, seed=seed, perturb=0.25, random_offsets=True, weights="random")
    limits = []
    if seed % 2 == 0 and P >= 4:
        a, b2 = rng.choice(P, size=2, replace=False)
        limits = [ParameterLimit.minmax(int(a), -0.05, 0.05, 1.5), ParameterLimit.linear(int(a), int(b2), 0.7, 0.02)]
        rows = [r for r in range(7 * J) if rig.pt_outer[r + 1] - rig.pt_outer[r] in (1, 2)]
        if rows:
            r0 = int(rng.choice(rows))
            limits.append(ParameterLimit.minmax_joint(r0 // 7, r0 % 7, -0.02, 0.03, 1.0))
    full = orc.Constraints(
        cons.pos_parent, cons.pos_offset, cons.pos_target, cons.pos_weight, cons.ori_parent, cons.ori_offset, cons.ori_target, cons.ori_weight,
        pos_function_weight=0.9, ori_function_weight=1.1, limits=limits, limit_function_weight=0.5,
    )  # fmt: skip
    rh = capi.RigHandle(rig, 0)
    pb = capi.Problem(rh, B, pp, op)
    dev = pb.device
    t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(dev)
    pb.set_constraints(
        t(cons.pos_offset, (B, Kp, 3)), t(cons.pos_target, (B, Kp, 3)), t(cons.pos_weight, (B, Kp)),
        t(cons.ori_offset, (B, Ko, 4)), t(cons.ori_target, (B, Ko, 4)), t(cons.ori_weight, (B, Ko)),
        0.9, 1.1, limits=limits, limit_function_weight=0.5,
    )  # fmt: skip
    en = (rng.uniform(size=P) < 0.8).astype(np.uint8)
    en[:3] = 1
    pb.set_enabled(en)
    theta = rng.uniform(-0.3, 0.3, size=(B, P)).astype(np.float32)
    jac, res, err = pb.eval_jacobian(torch.from_numpy(theta).to(dev))
    jac, res, err = jac.cpu().numpy(), res.cpu().numpy(), err.cpu().numpy()
    for b in range(B):
        Jo, ro, eo = orc.eval_jacobian(rig, full.instance(b), theta[b].astype(np.float64), enabled=en, dtype="f64")
        scale = max(1.0, np.abs(Jo).max())
        assert np.abs(jac[b].T - Jo).max() <= 3e-5 * scale, (seed, b)
        assert np.abs(res[b] - ro).max() <= 3e-5 * max(1.0, np.abs(ro).max())
        assert abs(err[b] - eo) <= 3e-5 * max(1.0, eo)
    # a short, well regularised solve (lambda = 0.5 keeps even the degenerate random rigs conditioned)
    # (every fifth rig with a line search: the trial evaluation hands its joint states to the next iteration)
    opt = GnOptions.make(min_iterations=5, max_iterations=5, regularization=0.5, do_line_search=(1 + seed % 2) if seed % 5 == 4 else 0,
                         step_rule=1 if os.environ.get("MMX_FUZZ_LM") and seed % 5 == 3 else 0)
    out = pb.solve(torch.from_numpy(th0.copy()).to(dev), opt, want_history=True)
    ref = orc.solve_batch(rig, full, th0, opt, enabled=en, dtype="f64")
    th = out["theta"].cpu().numpy()
    dnorm = np.maximum(np.linalg.norm(ref["theta"], axis=1), 1e-3)
    rel = np.linalg.norm(th - ref["theta"], axis=1) / dnorm
    tol = np.full(B, 2e-5)
    if np.any(rel > tol):  # (sweeps over larger rigs: the bound follows what the oracle's own float instantiation loses on the instance)
        ref32 = orc.solve_batch(rig, full, th0, opt, enabled=en, dtype="f32")
        tol = np.maximum(tol, 3.0 * np.linalg.norm(ref32["theta"] - ref["theta"], axis=1) / dnorm)
    assert np.all(rel <= tol), (seed, rel, tol)
    assert np.array_equal(out["iterations"].cpu().numpy(), ref["iterations"])
    assert np.array_equal(out["status"].cpu().numpy() & 3, ref["status"])
    h, href = out["error_history"].cpu().numpy(), ref["error_history"]
    assert np.abs(h - href).max() <= 1e-4 * max(1.0, np.abs(href).max())
    assert np.all(th[:, en == 0] == th0[:, en == 0])  # disabled parameters are never touched


@pytest.mark.parametrize("seed", range(int(os.environ.get("MMX_FUZZ_SEEDS", "48")) // 3))
def test_random_rig_with_joint_blocks_and_ellipsoids(torch_cuda, orc, seed):
    """The same random rigs with the further joint error functions and Ellipsoid limits: exercises the
    host tables (flattened constraint lists, DFS indices, the stop index of the ellipsoid walk, the compacted
    solve list) on arbitrary trees -- J / r through the explicit-Jacobian kernels, the solve through the fused
    kernel's general rows."""
    from momentum_amd._abi import EllipsoidLimit
    from tests.test_oracle_joint_blocks import TYPES, make_block

    torch = torch_cuda
    rng = np.random.default_rng(5000 + seed)
    J = int(rng.integers(2, int(os.environ.get("MMX_FUZZ_JMAX", "40"))))
    rig = random_rig(rng, J, ["chain", "star", "bushy"][seed % 3])
    P = rig.num_params
    Kp = int(rng.integers(0, 5))
    pp = rng.integers(0, J, size=Kp).astype(np.int32)
    op = np.zeros(0, np.int32)
    B = 3
    cons, th0, _ = make_problem(rig, pp, op, B, seed=seed, perturb=0.25, random_offsets=True, weights="random")
    types = list(TYPES.values())
    blocks = [make_block(types[int(k)], rng.integers(0, J, size=int(rng.integers(1, 4))), rng, weight=1.0, batch=B,
                         function_weight=float(rng.uniform(0.3, 1.2)), loss=(2.0, 1.0) if rng.uniform() < 0.7 else (0.0, 0.8))
              for k in rng.choice(len(types), size=int(rng.integers(1, 4)), replace=False)]  # fmt: skip
    ells = [EllipsoidLimit.make(int(rng.integers(0, J)), rng.uniform(-0.2, 0.2, 3), int(rng.integers(0, J)), rng.uniform(-0.2, 0.2, 3),
                                rng.uniform(-180, 180, 3), rng.uniform(0.1, 0.6, 3), float(rng.uniform(0.5, 2.0)))
            for _ in range(int(rng.integers(0, 3)))]  # fmt: skip
    wl = 50.0
    full = orc.Constraints(cons.pos_parent, cons.pos_offset, cons.pos_target, cons.pos_weight, cons.ori_parent, cons.ori_offset,
                           cons.ori_target, cons.ori_weight, joint_blocks=blocks, ellipsoid_limits=ells, limit_function_weight=wl)  # fmt: skip
    pb = capi.Problem(capi.RigHandle(rig, 0), B, pp, op)
    f = lambda a, shp: np.ascontiguousarray(a, np.float32).reshape(shp)
    pb.set_constraints(f(cons.pos_offset, (B, Kp, 3)), f(cons.pos_target, (B, Kp, 3)), f(cons.pos_weight, (B, Kp)), f(cons.ori_offset, (B, 0, 4)),
                       f(cons.ori_target, (B, 0, 4)), f(cons.ori_weight, (B, 0)), joint_blocks=blocks, ellipsoid_limits=ells,
                       limit_function_weight=wl)  # fmt: skip  (host payload: copied by the library)
    assert pb.M == full.rows
    en = (rng.uniform(size=P) < 0.85).astype(np.uint8)
    en[:3] = 1
    pb.set_enabled(en)
    theta = rng.uniform(-0.3, 0.3, size=(B, P)).astype(np.float32)
    jac, res, err = pb.eval_jacobian(torch.from_numpy(theta).to(pb.device))
    jac, res, err = jac.cpu().numpy(), res.cpu().numpy(), err.cpu().numpy()
    for b in range(B):
        Jo, ro, eo = orc.eval_jacobian(rig, full.instance(b), theta[b].astype(np.float64), enabled=en, dtype="f64")
        assert np.abs(jac[b].T - Jo).max() <= 5e-5 * max(1.0, np.abs(Jo).max()), (seed, b)
        assert np.abs(res[b] - ro).max() <= 5e-5 * max(1.0, np.abs(ro).max())
        assert abs(err[b] - eo) <= 5e-5 * max(1.0, eo)
    opt = GnOptions.make(min_iterations=4, max_iterations=4, regularization=0.5, do_line_search=bool(seed % 2))
    out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
    ref = orc.solve_batch(rig, full, th0, opt, enabled=en, dtype="f64")
    th = out["theta"].cpu().numpy()
    den = np.maximum(np.linalg.norm(ref["theta"], axis=1), 1e-3)
    rel = np.linalg.norm(th - ref["theta"], axis=1) / den
    # acos-type blocks (fixed-axis angle) amplify single-precision rounding: the bound follows what the oracle's own
    # float instantiation loses on the same instance (seed 15: 5.5e-5; explicit-Jacobian kernels 7e-5, fused solve 1.2e-4)
    ref32 = orc.solve_batch(rig, full, th0, opt, enabled=en, dtype="f32")
    tol = np.maximum(1e-4, 3.0 * np.linalg.norm(ref32["theta"] - ref["theta"], axis=1) / den)
    assert np.all(rel <= tol), (seed, rel, tol)
    h, href = out["error_history"].cpu().numpy(), ref["error_history"]
    assert np.abs(h - href).max() <= 1e-4 * max(1.0, np.abs(href).max())
    assert np.all(th[:, en == 0] == th0[:, en == 0])


@pytest.mark.parametrize("seed", range(int(os.environ.get("MMX_FUZZ_WIDE_SEEDS", "32"))))  # (a one-off sweep: MMX_FUZZ_WIDE_SEEDS=64)
def test_random_wide_rig_matches_oracle(torch_cuda, orc, seed, monkeypatch):
    """Random trees of 100-170 joints with shared parameters, translation / scale dofs and transform offsets, more than
    224 solved parameters: the wide path (tree normal equations incl. the term records of multi-source columns,
    paired-column factor, tree refinement) on shapes the 300-joint rig does not have; odd seeds keep some
    parameters disabled, every fourth one takes the explicit-Jacobian route (dense J^T J on the matrix cores, the
    refinement streaming J), seeds 2 and 6 the directional line search."""

    torch = torch_cuda
    if seed % 4 == 3:
        monkeypatch.setattr(capi, "default_route", "explicit_jacobian")
    rng = np.random.default_rng(9000 + seed)
    J = int(rng.integers(100, int(os.environ.get("MMX_FUZZ_WIDE_JMAX", "170"))))  # (MMX_FUZZ_WIDE_JMAX=195: up to the 512-parameter limit)
    rig = random_rig(rng, J, ["chain", "star", "bushy"][seed % 3])
    P = rig.num_params
    if P > 512:
        pytest.skip("more than 512 parameters: MMX_ERR_UNSUPPORTED")
    Kp, Ko = int(rng.integers(20, 70)), int(rng.integers(4, 24))
    pp = rng.integers(0, J, size=Kp).astype(np.int32)
    op = rng.integers(0, J, size=Ko).astype(np.int32)
    B = 3
    cons, th0, ths = make_problem(rig, pp, op, B, seed=seed, perturb=0.2, random_offsets=True, weights="random")
    pb = capi.Problem(capi.RigHandle(rig, 0), B, pp, op)
    t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(pb.device)
    pb.set_constraints(t(cons.pos_offset, (B, Kp, 3)), t(cons.pos_target, (B, Kp, 3)), t(cons.pos_weight, (B, Kp)),
                       t(cons.ori_offset, (B, Ko, 4)), t(cons.ori_target, (B, Ko, 4)), t(cons.ori_weight, (B, Ko)))  # fmt: skip
    en = np.ones(P, np.uint8)
    if seed % 2 == 1:
        en = (rng.uniform(size=P) < 0.9).astype(np.uint8)
        en[:3] = 1
    pb.set_enabled(en)
    opt = GnOptions.make(min_iterations=5, max_iterations=5, regularization=0.5, do_line_search=2 if seed % 4 == 2 else 0)
    out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
    ref = orc.solve_batch(rig, cons, th0, opt, enabled=en, dtype="f64")
    th = out["theta"].cpu().numpy()
    den = np.maximum(np.linalg.norm(ref["theta"], axis=1), 1e-3)
    rel = np.linalg.norm(th - ref["theta"], axis=1) / den
    # 2e-5 like the small random rigs; a chain of 100+ joints amplifies single-precision FK, so the bound follows what the
    # oracle's own float instantiation loses on the instance (seed 3, a 168-joint chain through the dense-J refinement: 3.2e-5)
    ref32 = orc.solve_batch(rig, cons, th0, opt, enabled=en, dtype="f32")
    tol = np.maximum(2e-5, 3.0 * np.linalg.norm(ref32["theta"] - ref["theta"], axis=1) / den)
    assert np.all(rel <= tol), (seed, J, P, rel, tol)
    assert np.array_equal(out["iterations"].cpu().numpy(), ref["iterations"])
    assert np.array_equal(out["status"].cpu().numpy() & 3, ref["status"])
    h, href = out["error_history"].cpu().numpy(), ref["error_history"]
    assert np.abs(h - href).max() <= 1e-4 * max(1.0, np.abs(href).max())
    assert np.all(th[:, en == 0] == th0[:, en == 0])


@pytest.mark.parametrize("seed", range(int(os.environ.get("MMX_FUZZ_WIDE_SEEDS", "32")) // 2))
def test_random_wide_rig_with_extra_rows(torch_cuda, orc, seed):
    """The wide path's generalisations on random trees: parameter limits (every seed), the model-parameter prior (every
    second), a plane block of a few rows (two of three), per-element constraint parents (every fourth) -- the tree kernels'
    extra-rows instantiation against the oracle."""
    from momentum_amd import _abi
    from tests.test_oracle_joint_blocks import make_block

    torch = torch_cuda
    rng = np.random.default_rng(11000 + seed)
    J = int(rng.integers(100, 160))
    rig = random_rig(rng, J, ["chain", "star", "bushy"][seed % 3])
    P = rig.num_params
    Kp, Ko = int(rng.integers(20, 60)), int(rng.integers(4, 20))
    B = 3
    inst = seed % 4 == 3
    pos_parents = [rng.integers(0, J, size=Kp).astype(np.int32) for _ in range(B if inst else 1)]
    ori_parents = [rng.integers(0, J, size=Ko).astype(np.int32) for _ in range(B if inst else 1)]
    if inst:
        conss = [make_problem(rig, pos_parents[b], ori_parents[b], 1, seed=seed * 7 + b, perturb=0.2, random_offsets=True, weights="random") for b in range(B)]
        cat = lambda f: np.concatenate([getattr(c[0], f) for c in conss], axis=0)
        th0 = np.concatenate([c[1] for c in conss], axis=0)
        base = dict(pos_offset=cat("pos_offset"), pos_target=cat("pos_target"), pos_weight=cat("pos_weight"),
                    ori_offset=cat("ori_offset"), ori_target=cat("ori_target"), ori_weight=cat("ori_weight"))
    else:
        c0, th0, _ = make_problem(rig, pos_parents[0], ori_parents[0], B, seed=seed, perturb=0.2, random_offsets=True, weights="random")
        base = dict(pos_offset=c0.pos_offset, pos_target=c0.pos_target, pos_weight=c0.pos_weight,
                    ori_offset=c0.ori_offset, ori_target=c0.ori_target, ori_weight=c0.ori_weight)
    a, b2 = rng.choice(P, size=2, replace=False)
    limits = [ParameterLimit.minmax(int(a), -0.05, 0.05, 1.5), ParameterLimit.linear(int(a), int(b2), 0.7, 0.02)]
    rows = [r for r in range(7 * J) if rig.pt_outer[r + 1] - rig.pt_outer[r] in (1, 2)]
    r0 = int(rng.choice(rows))
    limits.append(ParameterLimit.minmax_joint(r0 // 7, r0 % 7, -0.02, 0.03, 1.0))
    mt = mw = None
    if seed % 2 == 0:
        mt = rng.uniform(-0.2, 0.2, size=(B, P)).astype(np.float32)
        mw = rng.uniform(-0.3, 1.0, size=(B, P)).astype(np.float32)
    blocks = [make_block(_abi.MMX_JC_PLANE, rng.integers(0, J, size=int(rng.integers(2, 6))), rng, weight=1.0, batch=B)] if seed % 3 != 2 else []

    def constraints(b=None):
        s = (lambda x: x) if b is None else (lambda x: x[b])
        return orc.Constraints(pos_parents[0 if b is None or not inst else b], s(base["pos_offset"]), s(base["pos_target"]), s(base["pos_weight"]),
                               ori_parents[0 if b is None or not inst else b], s(base["ori_offset"]), s(base["ori_target"]), s(base["ori_weight"]),
                               limits=limits, limit_function_weight=0.5, model_target=None if mt is None else s(mt), model_weights=None if mw is None else s(mw),
                               model_function_weight=0.8, joint_blocks=blocks if b is None else [k.instance(b) for k in blocks])  # fmt: skip

    pb = capi.Problem(capi.RigHandle(rig, 0), B, pos_parents[0], ori_parents[0])
    t = lambda x, shp: torch.from_numpy(np.ascontiguousarray(x, np.float32).reshape(shp)).to(pb.device)
    gb = [_abi.JointBlock(k.type, k.parent, t(k.weight, (B, k.count)), t(k.global_, (B, k.count, 3)), t(k.local_point, (B, k.count, 3)), None,
                          t(k.plane_d, (B, k.count)), k.function_weight, k.loss) for k in blocks]  # fmt: skip
    pb.set_constraints(t(base["pos_offset"], (B, Kp, 3)), t(base["pos_target"], (B, Kp, 3)), t(base["pos_weight"], (B, Kp)),
                       t(base["ori_offset"], (B, Ko, 4)), t(base["ori_target"], (B, Ko, 4)), t(base["ori_weight"], (B, Ko)),
                       limits=limits, limit_function_weight=0.5, model_target=None if mt is None else t(mt, (B, P)),
                       model_weights=None if mw is None else t(mw, (B, P)), model_function_weight=0.8, joint_blocks=gb or None)  # fmt: skip
    if inst:
        pb.set_instance_parents(np.stack(pos_parents), np.stack(ori_parents))
    rigs = [rig] * B
    if seed % 4 == 1:  # per-element rig constants (bone lengths, pre-rotations) on top
        from tests.test_gpu_per_instance import _variants

        off, pre, rigs = _variants(rig, B, rng, scale=0.1)
        pb.set_instance_rig(off, pre)
    opt = GnOptions.make(min_iterations=5, max_iterations=5, regularization=0.5, do_line_search=2 if seed % 5 == 4 else 0)
    out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
    th = out["theta"].cpu().numpy()
    for b in range(B):
        cb = constraints(b)
        ref = orc.solve(rigs[b], cb, th0[b], opt, dtype="f64")
        r32 = orc.solve(rigs[b], cb, th0[b], opt, dtype="f32")
        den = max(np.linalg.norm(ref["theta"]), 1e-3)
        rel = np.linalg.norm(th[b] - ref["theta"]) / den
        tol = max(2e-5, 3.0 * np.linalg.norm(r32["theta"] - ref["theta"]) / den)
        assert rel <= tol, (seed, b, J, P, rel, tol)
        assert int(out["iterations"][b]) == ref["iterations"] and int(out["status"][b]) & 3 == ref["status"]
        href = np.asarray(ref["error_history"])
        h = out["error_history"][b].cpu().numpy()[: len(href)]
        assert np.abs(h - href).max() <= 1e-4 * max(1.0, np.abs(href).max())


@pytest.mark.parametrize("seed", range(int(os.environ.get("MMX_FUZZ_SEEDS", "48")) // 3))
def test_random_rig_double_solve_matches_oracle(torch_cuda, orc, seed):
    """mmx_solve_f64 on the random rigs (shared parameters, translation / scale dofs, transform offsets, enabled masks,
    every third with a line search): the oracle's double instantiation at 1e-8 (both run the same algorithm in the
    same precision; lambda = 0.5 keeps the degenerate rigs conditioned)."""

    torch = torch_cuda
    rng = np.random.default_rng(13000 + seed)
    J = int(rng.integers(2, int(os.environ.get("MMX_FUZZ_JMAX", "48"))))
    rig = random_rig(rng, J, ["chain", "star", "bushy"][seed % 3])
    P = rig.num_params
    Kp, Ko = int(rng.integers(1, 9)), int(rng.integers(0, 6))
    pp = rng.integers(0, J, size=Kp).astype(np.int32)
    op = rng.integers(0, J, size=Ko).astype(np.int32)
    B = 3
    cons, th0, _ = make_problem(rig, pp, op, B, seed=seed, perturb=0.25, random_offsets=True, weights="random")
    pb = capi.Problem(capi.RigHandle(rig, 0), B, pp, op)
    t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(pb.device)
    pb.set_constraints(t(cons.pos_offset, (B, Kp, 3)), t(cons.pos_target, (B, Kp, 3)), t(cons.pos_weight, (B, Kp)),
                       t(cons.ori_offset, (B, Ko, 4)), t(cons.ori_target, (B, Ko, 4)), t(cons.ori_weight, (B, Ko)))  # fmt: skip
    en = (rng.uniform(size=P) < 0.85).astype(np.uint8)
    en[:3] = 1
    pb.set_enabled(en)
    opt = GnOptions.make(min_iterations=5, max_iterations=5, regularization=0.5, do_line_search=(1 + seed % 2) if seed % 3 == 2 else 0)
    out = pb.solve_f64(torch.from_numpy(th0.astype(np.float64)).to(pb.device), opt, want_history=True)
    ref = orc.solve_batch(rig, cons, th0, opt, enabled=en, dtype="f64")
    th = out["theta"].cpu().numpy()
    rel = np.linalg.norm(th - ref["theta"], axis=1) / np.maximum(np.linalg.norm(ref["theta"], axis=1), 1e-3)
    assert np.all(rel <= 1e-8), (seed, J, P, rel)
    assert np.array_equal(out["iterations"].cpu().numpy(), ref["iterations"]) and np.array_equal(out["status"].cpu().numpy() & 3, ref["status"])
    assert np.all(th[:, en == 0] == th0[:, en == 0].astype(np.float64))
