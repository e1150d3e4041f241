"""ctypes wrapper of the CPU ORACLE (oracle/libmmx_oracle.so).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package momentum_amd (tests/test_no_oracle_in_product.py
enforces that).  See oracle/mmx_oracle.hpp for the pinning statement and reference citations.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

from momentum_amd._abi import ConstraintData, GnOptions, MMX_MEM_HOST, as_ptr, ellipsoid_array, joint_block_array, limit_array, void_p
from momentum_amd.rigs import Rig

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmmx_oracle.so")
_lib: Optional[C.CDLL] = None


_STAMP = os.path.join(_HERE, ".built_for")
# ISA / tuning candidates of the CPU baseline.  SURVEY.md 8d asks for the host's own ISA (-march=native); gcc 11 does not know
# the GPU box's Zen 5 (`native` resolves to znver3 + the AVX-512 feature flags), so build() compiles every candidate below,
# times a fixed single-thread solve workload with each (_selftime) and keeps the fastest: the baseline is the best this code
# does on the box at hand, and cpu_baseline's details say which flags won and what the others reached.  Since round 5 the
# dense kernels are written with explicit vector types (mmx_oracle.hpp syrkLower), so the candidates differ by < 15 % on a
# converging solve (EPYC 9575F: 1155 ... 1321 solves/s/thread; profiles/r05_cpu_baseline.txt has the per-phase split and
# why rounds 3-4 saw a 4-5 x spread).
_CANDIDATES = [
    "-march=native",
    "-march=native -mtune=generic",
    "-march=native -mtune=generic -mprefer-vector-width=256",
    "-march=x86-64-v3",
]
_BASE_FLAGS = "-O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter"


def host_cpu() -> str:
    """Model name + the widest vector ISA of the CPU this process runs on."""
    model, flags = "unknown", set()
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("flags") and not flags:
                flags = set(line.split(":", 1)[1].split())
            if model != "unknown" and flags:
                break
    except OSError:
        pass
    isa = "avx512" if "avx512f" in flags else ("avx2" if "avx2" in flags else "sse")
    return f"{model} ({isa})"


def build_info() -> str:
    """What the library in use was built with ("<cpu> | <flags> | timings of the candidates")."""
    try:
        return open(_STAMP).read().strip()
    except OSError:
        return "unknown"


def _selftime() -> float:
    """Seconds for a fixed single-thread workload with the library on disk: 32 solves of BASELINE configs[1]'s shape (72-joint
    humanoid, position + orientation on the 16 landmarks, targets = FK(theta*) like bench.py's batches, 10 iterations, float);
    run in a fresh interpreter so that every candidate is loaded from scratch.  (Until round 5 the workload had random,
    unreachable targets: its undamped iterations ran into Inf / NaN, and how slow THAT arithmetic is differs several-fold
    between the ISA candidates -- 312 against 1219 solves/s on the GPU box's EPYC 9575F -- while on a converging solve the same
    candidates are within 15 % of each other: the sweep picked by the pathology, profiles/r05_cpu_baseline.txt.)"""
    import sys

    code = (
        "import sys,time;sys.path.insert(0,%r);import numpy as np;"
        "from oracle import oracle as o;o._lib=__import__('ctypes').CDLL(o._LIB_PATH);"
        "from momentum_amd import humanoid72_landmark_joints as lj, make_humanoid72 as mk;from momentum_amd._abi import GnOptions as G;"
        "from tests.helpers import make_problem as mp;"
        "r=mk(seed=12345,variant='p128',unit=0.01);l=lj(r);B=32;c,th,_=mp(r,l,l,B,seed=1,perturb=0.3);"
        "op=G.make(10,10);o.solve_batch(r,c,th[:4],op,dtype='f32');"
        "t=time.perf_counter();o.solve_batch(r,c,th,op,dtype='f32');print(time.perf_counter()-t)"
    ) % os.path.dirname(_HERE)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True).stdout.strip().split("\n")[-1]
    return float(out)


def _make(flags: str) -> None:
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B", f"CXXFLAGS={_BASE_FLAGS} {flags}"])


def build(force: bool = False) -> str:
    """Compile the oracle with oracle/Makefile (gcc only).  A library built on another host (it travels with the snapshot)
    is rebuilt for this one, with the fastest of the ISA candidates above."""
    cpu = host_cpu()
    stamp = build_info()

    def stale():  # a source newer than the library (the library travels with the snapshot; its sources may have moved on)
        try:
            t = os.path.getmtime(_LIB_PATH)
            srcs = [os.path.join(_HERE, "mmx_oracle_capi.cpp"), os.path.join(_HERE, "mmx_oracle.hpp"), os.path.join(os.path.dirname(_HERE), "include", "mmx.h")]
            return any(os.path.getmtime(p) > t for p in srcs)
        except OSError:
            return True

    force = force or stale()
    if force or not os.path.exists(_LIB_PATH) or not stamp.startswith(cpu + " | "):
        import fcntl

        with open(os.path.join(_HERE, ".build_lock"), "w") as lock:  # two ranks of one test may get here together
            fcntl.flock(lock, fcntl.LOCK_EX)
            stamp = build_info()
            if force or not os.path.exists(_LIB_PATH) or not stamp.startswith(cpu + " | "):
                timings = []
                for flags in _CANDIDATES:
                    try:
                        _make(flags)
                        timings.append((_selftime(), flags))
                    except Exception:  # a candidate this compiler / CPU does not take
                        continue
                if not timings:
                    _make("-march=x86-64-v3")
                    best, note = "-march=x86-64-v3", "no candidate could be timed"
                else:
                    best = min(timings)[1]
                    note = "; ".join(f"{f}: {32 / t:.0f} solves/s/thread" for t, f in timings)
                    if best != timings[-1][1]:
                        _make(best)
                with open(_STAMP, "w") as f:
                    f.write(f"{cpu} | {_BASE_FLAGS.split()[0]} {best} | {note}\n")
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
    return _lib


def _np(dtype):
    return np.float32 if dtype in ("f32", np.float32) else np.float64


def _suf(dtype) -> str:
    return "f32" if _np(dtype) is np.float32 else "f64"


def _ct(dtype):
    return C.c_float if _np(dtype) is np.float32 else C.c_double


class Constraints:
    """One instance's (or a batch's) constraint payload, numpy-backed."""

    def __init__(
        self,
        pos_parent,
        pos_offset,
        pos_target,
        pos_weight,
        ori_parent,
        ori_offset,
        ori_target,
        ori_weight,
        pos_function_weight: float = 1.0,
        ori_function_weight: float = 1.0,
        limits=None,
        limit_function_weight: float = 1.0,
        model_target=None,
        model_weights=None,
        model_function_weight: float = 1.0,
        pos_loss=(2.0, 1.0),
        ori_loss=(2.0, 1.0),
        joint_blocks=None,
        ellipsoid_limits=None,
        function_weights=None,
    ):
        f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        # per-element error-function weights [B, C] (errorFunctionWeights[iBatch][weightsMap[iErr]] of solveTensorIKProblem):
        # columns position, orientation, limits, model parameters, joint block 0, ...
        self.function_weights = None if function_weights is None else np.ascontiguousarray(function_weights, dtype=np.float32).reshape(-1, np.asarray(function_weights).shape[-1])
        self.pos_parent = np.ascontiguousarray(pos_parent, dtype=np.int32).reshape(-1)
        self.ori_parent = np.ascontiguousarray(ori_parent, dtype=np.int32).reshape(-1)
        self.Kp = int(self.pos_parent.shape[0])
        self.Ko = int(self.ori_parent.shape[0])
        self.pos_offset, self.pos_target, self.pos_weight = f(pos_offset), f(pos_target), f(pos_weight)
        self.ori_offset, self.ori_target, self.ori_weight = f(ori_offset), f(ori_target), f(ori_weight)
        self.pos_function_weight = float(pos_function_weight)
        self.ori_function_weight = float(ori_function_weight)
        # parameter-space blocks: a list of momentum_amd._abi.ParameterLimit (batch-shared) and the
        # ModelParametersErrorFunction targets / weights ([P] or [B,P])
        self.limits = list(limits) if limits else []
        self._limit_array = limit_array(self.limits)
        self.limit_function_weight = float(limit_function_weight)
        self.model_target = None if model_target is None else f(model_target)
        self.model_weights = None if model_weights is None else f(model_weights)
        self.model_function_weight = float(model_function_weight)
        self.pos_loss = (float(pos_loss[0]), float(pos_loss[1]))  # GeneralizedLoss (alpha, c)
        self.ori_loss = (float(ori_loss[0]), float(ori_loss[1]))
        self.joint_blocks = list(joint_blocks) if joint_blocks else []  # momentum_amd._abi.JointBlock
        self.ellipsoid_limits = list(ellipsoid_limits) if ellipsoid_limits else []  # momentum_amd._abi.EllipsoidLimit (batch-shared)
        self._ellipsoid_array = ellipsoid_array(self.ellipsoid_limits)

    @property
    def P(self) -> int:
        return 0 if self.model_target is None else int(self.model_target.shape[-1])

    @property
    def rows(self) -> int:
        return 3 * self.Kp + 9 * self.Ko + sum(b.rows for b in self.joint_blocks) + 3 * len(self.ellipsoid_limits) + len(self.limits) + self.P

    def data(self) -> ConstraintData:
        self._keep = []
        self._block_array = joint_block_array(self.joint_blocks, self._keep)
        return ConstraintData(
            void_p(self.pos_offset if self.Kp else None),
            void_p(self.pos_target if self.Kp else None),
            void_p(self.pos_weight if self.Kp else None),
            void_p(self.ori_offset if self.Ko else None),
            void_p(self.ori_target if self.Ko else None),
            void_p(self.ori_weight if self.Ko else None),
            self.pos_function_weight,
            self.ori_function_weight,
            MMX_MEM_HOST,
            void_p(self.model_target),
            void_p(self.model_weights),
            self.model_function_weight,
            len(self.limits),
            C.cast(self._limit_array, C.c_void_p) if self.limits else None,
            self.limit_function_weight,
            self.pos_loss[0],
            self.pos_loss[1],
            self.ori_loss[0],
            self.ori_loss[1],
            len(self.joint_blocks),
            C.cast(self._block_array, C.c_void_p) if self.joint_blocks else None,
            len(self.ellipsoid_limits),
            C.cast(self._ellipsoid_array, C.c_void_p) if self.ellipsoid_limits else None,
            void_p(self.function_weights),
            0 if self.function_weights is None else int(self.function_weights.shape[1]),
        )

    def instance(self, b: int) -> "Constraints":
        """Slice instance b out of a batched payload ([B,K,...] arrays)."""
        return Constraints(
            self.pos_parent,
            self.pos_offset.reshape(-1, self.Kp, 3)[b] if self.Kp else self.pos_offset,
            self.pos_target.reshape(-1, self.Kp, 3)[b] if self.Kp else self.pos_target,
            self.pos_weight.reshape(-1, self.Kp)[b] if self.Kp else self.pos_weight,
            self.ori_parent,
            self.ori_offset.reshape(-1, self.Ko, 4)[b] if self.Ko else self.ori_offset,
            self.ori_target.reshape(-1, self.Ko, 4)[b] if self.Ko else self.ori_target,
            self.ori_weight.reshape(-1, self.Ko)[b] if self.Ko else self.ori_weight,
            self.pos_function_weight,
            self.ori_function_weight,
            self.limits,
            self.limit_function_weight,
            None if self.model_target is None else self.model_target.reshape(-1, self.P)[b if self.model_target.ndim > 1 else 0],
            None if self.model_weights is None else self.model_weights.reshape(-1, self.P)[b if self.model_weights.ndim > 1 else 0],
            self.model_function_weight,
            self.pos_loss,
            self.ori_loss,
            [blk.instance(b) for blk in self.joint_blocks],
            self.ellipsoid_limits,
            None if self.function_weights is None else self.function_weights[b],
        )


def _subset(self, idx) -> "Constraints":
    """Instances idx (an index array) of a batched payload, every block of it (joint blocks, limits, prior) included."""
    return self.instance(np.asarray(idx, dtype=np.int64))


Constraints.subset = _subset


def _enabled_ptr(enabled, P):
    if enabled is None:
        return None, C.POINTER(C.c_uint8)()
    e = np.ascontiguousarray(enabled, dtype=np.uint8).reshape(-1)
    assert e.shape[0] == P
    return e, as_ptr(e, C.c_uint8)


def skeleton_state(rig: Rig, theta, dtype="f64"):
    """A1+A2: returns dict(world[J,8], local[J,8], trans_axis[J,3,3], rot_axis[J,3,3], joint_params[7J])."""
    T = _np(dtype)
    J = rig.num_joints
    th = np.ascontiguousarray(theta, dtype=T).reshape(-1)
    assert th.shape[0] == rig.num_params
    world = np.zeros((J, 8), T)
    local = np.zeros((J, 8), T)
    ta = np.zeros((J, 3, 3), T)
    ra = np.zeros((J, 3, 3), T)
    jp = np.zeros(7 * J, T)
    d = rig.desc()
    ct = _ct(dtype)
    fn = getattr(lib(), f"orc_skeleton_state_{_suf(dtype)}")
    rc = fn(C.byref(d), as_ptr(th, ct), as_ptr(world, ct), as_ptr(local, ct), as_ptr(ta, ct), as_ptr(ra, ct), as_ptr(jp, ct))
    assert rc == 0
    return dict(world=world, local=local, trans_axis=ta, rot_axis=ra, joint_params=jp)


def eval_jacobian(rig: Rig, cons: Constraints, theta, enabled=None, dtype="f64"):
    """Returns (J [M,P] (numpy view of the column-major buffer, i.e. J[i,p]), r [M], error)."""
    T = _np(dtype)
    M, P = cons.rows, rig.num_params
    th = np.ascontiguousarray(theta, dtype=T).reshape(-1)
    jac = np.zeros((P, M), T)  # column-major M x P == C-order [P][M]
    res = np.zeros(M, T)
    err = C.c_double(0)
    d, cd = rig.desc(), cons.data()
    ekeep, eptr = _enabled_ptr(enabled, P)
    ct = _ct(dtype)
    fn = getattr(lib(), f"orc_eval_jacobian_{_suf(dtype)}")
    rc = fn(
        C.byref(d), cons.Kp, as_ptr(cons.pos_parent, C.c_int32), cons.Ko, as_ptr(cons.ori_parent, C.c_int32),
        C.byref(cd), eptr, as_ptr(th, ct), as_ptr(jac, ct), as_ptr(res, ct), C.byref(err),
    )  # fmt: skip
    assert rc == 0
    return jac.T, res, err.value


def get_error(rig: Rig, cons: Constraints, theta, dtype="f64") -> float:
    T = _np(dtype)
    th = np.ascontiguousarray(theta, dtype=T).reshape(-1)
    err = C.c_double(0)
    d, cd = rig.desc(), cons.data()
    fn = getattr(lib(), f"orc_get_error_{_suf(dtype)}")
    rc = fn(
        C.byref(d), cons.Kp, as_ptr(cons.pos_parent, C.c_int32), cons.Ko, as_ptr(cons.ori_parent, C.c_int32),
        C.byref(cd), as_ptr(th, _ct(dtype)), C.byref(err),
    )  # fmt: skip
    assert rc == 0
    return err.value


def solve(rig: Rig, cons: Constraints, theta0, options: GnOptions, enabled=None, dtype="f64", use_block_jtj=False, use_qr=False):
    """SolverT::solve with GaussNewtonSolverT for one instance.  Returns dict(theta, error,
    iterations, status, error_history, jtj, jtr) (jtj/jtr = compacted system of the last iteration).
    use_qr: GaussNewtonSolverQRT (gauss_newton_solver_qr.cpp:50-150) instead of GaussNewtonSolverT's Cholesky; a line search,
    when asked for, is then the directional rule whatever do_line_search says (the QR solver has no other)."""
    T = _np(dtype)
    P = rig.num_params
    th = np.array(theta0, dtype=T).reshape(-1).copy()
    ekeep, eptr = _enabled_ptr(enabled, P)
    n = P if enabled is None else int(np.count_nonzero(ekeep))
    err = C.c_double(0)
    iters = C.c_int32(0)
    status = C.c_int32(0)
    hist = np.zeros(max(1, options.max_iterations), np.float64)
    jtj = np.zeros((n, n), T)
    jtr = np.zeros(n, T)
    d, cd = rig.desc(), cons.data()
    ct = _ct(dtype)
    fn = getattr(lib(), f"orc_solve_{_suf(dtype)}")
    rc = fn(
        C.byref(d), cons.Kp, as_ptr(cons.pos_parent, C.c_int32), cons.Ko, as_ptr(cons.ori_parent, C.c_int32),
        C.byref(cd), eptr, C.byref(options), int(bool(use_block_jtj)) | (2 if use_qr else 0), as_ptr(th, ct), C.byref(err), C.byref(iters),
        C.byref(status), as_ptr(hist, C.c_double), as_ptr(jtj, ct), as_ptr(jtr, ct),
    )  # fmt: skip
    assert rc == 0
    # the oracle stores the lower triangle column-major: element (i,j), i>=j at [j*n+i]; as a C-order
    # numpy array that is the upper triangle -> mirror to a full symmetric matrix
    full = np.triu(jtj) + np.triu(jtj, 1).T
    return dict(
        theta=th, error=err.value, iterations=iters.value, status=status.value,
        error_history=hist[: iters.value].copy(), jtj=full, jtr=jtr,
    )  # fmt: skip


def solve_batch(rig: Rig, cons: Constraints, theta0, options: GnOptions, enabled=None, dtype="f32", nthreads=1, use_block_jtj=False, use_qr=False, step_history=False):
    """The reference's batched driver shape (pymomentum/tensor_ik/tensor_ik.cpp:127-177): one
    independent solver per instance, `nthreads` std::threads.  cons arrays are [B,K,...].
    step_history (LM schedule, step_rule 1): also `lambda_history`, `gain_ratio_history` [B][max_iterations] -- the damping
    iteration i factored with and the gain ratio its accept / scale decisions were taken on."""
    T = _np(dtype)
    P = rig.num_params
    th = np.array(theta0, dtype=T).reshape(-1, P).copy()
    B = th.shape[0]
    ekeep, eptr = _enabled_ptr(enabled, P)
    err = np.zeros(B, np.float64)
    iters = np.zeros(B, np.int32)
    status = np.zeros(B, np.int32)
    hist = np.zeros((B, max(1, options.max_iterations)), np.float64)
    d, cd = rig.desc(), cons.data()
    ct = _ct(dtype)
    args = (
        C.byref(d), B, cons.Kp, as_ptr(cons.pos_parent, C.c_int32), cons.Ko, as_ptr(cons.ori_parent, C.c_int32),
        C.byref(cd), eptr, C.byref(options), int(bool(use_block_jtj)) | (2 if use_qr else 0), as_ptr(th, ct), as_ptr(err, C.c_double),
        as_ptr(iters, C.c_int32), as_ptr(status, C.c_int32), as_ptr(hist, C.c_double), int(nthreads),
    )  # fmt: skip
    out = dict(theta=th, error=err, iterations=iters, status=status, error_history=hist)
    if step_history:
        lam = np.zeros_like(hist)
        rho = np.zeros_like(hist)
        rc = getattr(lib(), f"orc_solve_batch_steps_{_suf(dtype)}")(*args, as_ptr(lam, C.c_double), as_ptr(rho, C.c_double))
        out.update(lambda_history=lam, gain_ratio_history=rho)
    else:
        rc = getattr(lib(), f"orc_solve_batch_{_suf(dtype)}")(*args)
    assert rc == 0
    return out


def mock_solve(P: int, theta0, options: GnOptions, enabled=None, dtype="f64", use_block_jtj=False):
    """GaussNewtonSolverT on MockSolverFunction (J = I, r = theta);
    momentum/test/solver/gauss_newton_solver_test.cpp:19-106."""
    T = _np(dtype)
    th = np.array(theta0, dtype=T).reshape(-1).copy()
    ekeep, eptr = _enabled_ptr(enabled, P)
    err = C.c_double(0)
    iters = C.c_int32(0)
    hist = np.zeros(max(1, options.max_iterations), np.float64)
    fn = getattr(lib(), f"orc_mock_solve_{_suf(dtype)}")
    rc = fn(P, eptr, C.byref(options), int(use_block_jtj), as_ptr(th, _ct(dtype)), C.byref(err), C.byref(iters), as_ptr(hist, C.c_double))
    assert rc == 0
    return dict(theta=th, error=err.value, iterations=iters.value, error_history=hist[: iters.value].copy())


def ancestor_matrix(rig: Rig) -> np.ndarray:
    J = rig.num_joints
    out = np.zeros((J, J), np.uint8)
    d = rig.desc()
    assert lib().orc_ancestor_matrix(C.byref(d), as_ptr(out, C.c_uint8)) == 0
    return out


def active_joint_params(rig: Rig, enabled) -> np.ndarray:
    e = np.ascontiguousarray(enabled, dtype=np.uint8)
    out = np.zeros(7 * rig.num_joints, np.uint8)
    d = rig.desc()
    assert lib().orc_active_joint_params(C.byref(d), as_ptr(e, C.c_uint8), as_ptr(out, C.c_uint8)) == 0
    return out


def hardware_threads() -> int:
    return int(lib().orc_hardware_threads())
