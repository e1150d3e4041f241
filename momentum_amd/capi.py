"""ctypes binding of momentum_amd/libmmx_hip.so (the C ABI of include/mmx.h).

Plumbing only: torch owns device memory and streams, the HIP library does all compute.  There is
no CPU fallback -- loading fails loudly if the library is missing, and every compute call raises
MmxError if the device path cannot run.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import _abi
from ._abi import ConstraintData, GnOptions, RigDesc, as_ptr
from .rigs import Rig

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMX_LIB") or os.path.join(_HERE, "libmmx_hip.so")  # MMX_LIB: A/B experiment builds (momentum_amd/build.py)
_lib: Optional[C.CDLL] = None

# every symbol include/mmx.h declares (tests/test_abi.py checks the header against this list and
# the built library against both)
SYMBOLS = [
    "mmx_gn_options_default", "mmx_abi_version", "mmx_last_error", "mmx_device_count",
    "mmx_rig_create", "mmx_rig_destroy", "mmx_rig_num_joints", "mmx_rig_num_params",
    "mmx_problem_create", "mmx_problem_destroy", "mmx_problem_num_rows", "mmx_problem_batch",
    "mmx_problem_set_tuning", "mmx_problem_last_route",
    "mmx_problem_set_enabled", "mmx_problem_set_constraints", "mmx_problem_set_constraints_sized", "mmx_problem_set_instance_rig", "mmx_problem_set_instance_parents", "mmx_eval_jacobian", "mmx_eval_jacobian_timed", "mmx_debug_store_pattern",
    "mmx_eval_skeleton_state", "mmx_eval_normal_equations", "mmx_solve", "mmx_solve_with_history", "mmx_solve_with_step_history", "mmx_problem_solve_diagnostics", "mmx_solve_f64", "mmx_solve_f64_host", "mmx_solve_host",
    "mmx_eval_jacobian_host", "mmx_eval_skeleton_state_host", "mmx_host_tables", "mmx_debug_fused_normal_equations", "mmx_debug_tree_normal_equations",
    "mmx_host_elimination_order", "mmx_host_tile_structure", "mmx_host_tile_level_schedule", "mmx_host_f64_assembly_list", "mmx_problem_tile_structure",
    "mmx_comm_unique_id", "mmx_comm_create", "mmx_comm_create_all", "mmx_comm_world_size", "mmx_comm_rank",
    "mmx_comm_all_reduce_norms", "mmx_comm_all_reduce_norms_host", "mmx_residual_norms", "mmx_comm_destroy",
]  # fmt: skip


class MmxError(RuntimeError):
    """Non-zero status from the C ABI (the C++ shell maps it to std::runtime_error like MT_CHECK)."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"mmx error {code}: {msg}")
        self.code = code


def lib() -> C.CDLL:
    """Loads libmmx_hip.so.  torch is imported first so that both share one HIP runtime."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MmxError(-1, f"{LIB_PATH} is missing: build it with `python -m momentum_amd.build` (no CPU fallback exists)")
    try:
        import torch  # noqa: F401  (loads torch's libamdhip64 first; same soname, one runtime)
    except Exception:  # pragma: no cover - torch is plumbing, the library also works without it
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32 = C.c_void_p, C.c_int32
    L.mmx_last_error.restype = C.c_char_p
    L.mmx_gn_options_default.restype = None
    L.mmx_gn_options_default.argtypes = [C.POINTER(GnOptions)]
    L.mmx_rig_create.argtypes = [C.POINTER(RigDesc), i32, C.POINTER(vp)]
    L.mmx_rig_destroy.argtypes = [vp]
    L.mmx_rig_destroy.restype = None
    L.mmx_rig_num_joints.argtypes = [vp]
    L.mmx_rig_num_params.argtypes = [vp]
    L.mmx_problem_create.argtypes = [vp, i32, i32, _abi.c_int32_p, i32, _abi.c_int32_p, C.POINTER(vp)]
    L.mmx_problem_destroy.argtypes = [vp]
    L.mmx_problem_destroy.restype = None
    L.mmx_problem_num_rows.argtypes = [vp]
    L.mmx_problem_batch.argtypes = [vp]
    L.mmx_problem_set_enabled.argtypes = [vp, _abi.c_uint8_p]
    L.mmx_problem_set_tuning.argtypes = [vp, C.POINTER(_abi.Tuning)]
    L.mmx_problem_last_route.argtypes = [vp]
    L.mmx_problem_set_constraints.argtypes = [vp, C.POINTER(ConstraintData), vp]
    L.mmx_problem_set_constraints_sized.argtypes = [vp, C.POINTER(ConstraintData), C.c_size_t, vp]
    L.mmx_problem_set_instance_rig.argtypes = [vp, vp, vp, i32, vp]
    L.mmx_problem_set_instance_parents.argtypes = [vp, vp, vp, i32, vp]
    L.mmx_eval_jacobian.argtypes = [vp, vp, vp, vp, vp, i32, vp]
    L.mmx_eval_jacobian_timed.argtypes = [vp, vp, vp, vp, vp, i32, vp, C.POINTER(C.c_float)]
    L.mmx_debug_store_pattern.argtypes = [vp, vp, vp, C.POINTER(C.c_float)]
    L.mmx_eval_skeleton_state.argtypes = [vp, vp, vp, vp]
    L.mmx_eval_normal_equations.argtypes = [vp, vp, vp, vp, vp, vp]
    L.mmx_solve.argtypes = [vp, C.POINTER(GnOptions), vp, vp, vp, vp, vp, vp]
    L.mmx_solve_with_history.argtypes = [vp, C.POINTER(GnOptions), vp, vp, vp, vp, vp, vp, vp]
    L.mmx_solve_with_step_history.argtypes = [vp, C.POINTER(GnOptions), vp, vp, vp, vp, vp, vp, vp, vp]
    L.mmx_problem_solve_diagnostics.argtypes = [vp, vp, vp]
    L.mmx_solve_f64.argtypes = [vp, C.POINTER(GnOptions), vp, vp, vp, vp, vp, vp]
    L.mmx_solve_host.argtypes = [vp, C.POINTER(GnOptions), vp, vp, vp, vp]
    L.mmx_solve_f64_host.argtypes = [vp, C.POINTER(GnOptions), vp, vp, vp, vp]
    L.mmx_eval_jacobian_host.argtypes = [vp, vp, vp, vp, vp, i32]
    L.mmx_eval_skeleton_state_host.argtypes = [vp, vp, vp]
    L.mmx_debug_tree_normal_equations.argtypes = [vp, vp, vp, vp, vp]
    L.mmx_debug_fused_normal_equations.argtypes = [vp, vp, vp, vp, _abi.c_int32_p, _abi.c_int32_p, vp]
    L.mmx_comm_unique_id.argtypes = [vp]
    L.mmx_comm_create.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
    L.mmx_comm_create_all.argtypes = [i32, _abi.c_int32_p, C.POINTER(vp)]
    L.mmx_comm_world_size.argtypes = [vp]
    L.mmx_comm_rank.argtypes = [vp]
    L.mmx_comm_all_reduce_norms.argtypes = [vp, vp, vp]
    L.mmx_comm_all_reduce_norms_host.argtypes = [vp, vp]
    L.mmx_residual_norms.argtypes = [i32, vp, vp, vp, vp, vp]
    L.mmx_comm_destroy.argtypes = [vp]
    L.mmx_comm_destroy.restype = None
    L.mmx_host_tables.argtypes = [
        C.POINTER(RigDesc), _abi.c_uint8_p, _abi.c_int32_p, _abi.c_int32_p, _abi.c_int32_p,
        _abi.c_uint8_p, _abi.c_int32_p, _abi.c_int32_p,
    ]  # fmt: skip
    u32p, i64p = C.POINTER(C.c_uint32), C.POINTER(C.c_int64)
    L.mmx_host_elimination_order.argtypes = [C.POINTER(RigDesc), _abi.c_uint8_p, _abi.c_int32_p, _abi.c_int32_p]
    L.mmx_host_tile_structure.argtypes = [i32, _abi.c_uint8_p, u32p, u32p, i64p]
    L.mmx_host_tile_level_schedule.argtypes = [i32, _abi.c_uint8_p, C.POINTER(C.c_int32)]
    L.mmx_problem_tile_structure.argtypes = [vp, u32p, u32p, _abi.c_int32_p, _abi.c_int32_p, i64p]
    _lib = L
    return L


def _check(rc: int) -> None:
    if rc != 0:
        raise MmxError(rc, lib().mmx_last_error().decode("utf-8", "replace"))


def device_count() -> int:
    return int(lib().mmx_device_count())


def host_tables(rig: Rig, enabled=None) -> dict:
    """Integer bookkeeping of the path, computed by the library's host code (no GPU needed)."""
    J, P = rig.num_joints, rig.num_params
    level, tin, tout = (np.zeros(J, np.int32) for _ in range(3))
    active = np.zeros(7 * J, np.uint8)
    elist = np.zeros(P, np.int32)
    n = C.c_int32(0)
    d = rig.desc()
    if enabled is None:
        eptr = _abi.c_uint8_p()
    else:
        e = np.ascontiguousarray(enabled, dtype=np.uint8)
        eptr = as_ptr(e, C.c_uint8)
    _check(
        lib().mmx_host_tables(
            C.byref(d), eptr, as_ptr(level, C.c_int32), as_ptr(tin, C.c_int32), as_ptr(tout, C.c_int32),
            as_ptr(active, C.c_uint8), as_ptr(elist, C.c_int32), C.byref(n),
        )
    )  # fmt: skip
    order = np.zeros(P, np.int32)
    _check(lib().mmx_host_elimination_order(C.byref(d), eptr, as_ptr(order, C.c_int32), C.byref(n)))
    return dict(level=level, tin=tin, tout=tout, active_joint_params=active, enabled_list=elist[: n.value].copy(),
                elimination_order=order[: n.value].copy())  # fmt: skip


def host_tile_structure(related: np.ndarray):
    """Symbolic factorisation on the 16 x 16 tile grid (mmx_host_tile_structure): related [n, n] (lower triangle used)
    -> dict(row_mask [32], col_mask [32], products)."""
    rel = np.ascontiguousarray(related, dtype=np.uint8)
    n = rel.shape[0]
    assert rel.shape == (n, n)
    row, col = np.zeros(32, np.uint32), np.zeros(32, np.uint32)
    prod = C.c_int64(0)
    _check(lib().mmx_host_tile_structure(C.c_int32(n), as_ptr(rel, C.c_uint8), as_ptr(row, C.c_uint32), as_ptr(col, C.c_uint32), C.byref(prod)))
    return dict(row_mask=row, col_mask=col, products=prod.value)


def host_tile_level_schedule(related: np.ndarray):
    """The resident factor kernel's level schedule for that structure (mmx_host_tile_level_schedule): list of steps, a step =
    list of (block column, first wave, number of waves; 15 = the whole workgroup)."""
    rel = np.ascontiguousarray(related, dtype=np.uint8)
    n = rel.shape[0]
    assert rel.shape == (n, n)
    words = np.zeros(1 + 4 * 32, np.int32)
    _check(lib().mmx_host_tile_level_schedule(C.c_int32(n), as_ptr(rel, C.c_uint8), as_ptr(words, C.c_int32)))
    steps = []
    for s in range(int(words[0])):
        steps.append([(int(w) & 0xff, (int(w) >> 8) & 0xf, (int(w) >> 12) & 0xf) for w in words[1 + 4 * s : 5 + 4 * s] if w >= 0])
    return steps


def host_f64_assembly_list(rig: Rig, solve_list, pos_parent, ori_parent, units_per_chunk: int):
    """mmx_solve_f64's assembly list (mmx_host_f64_assembly_list): list of chunks, a chunk = dict(block_mask, entries) with
    entries = list of (column, unit in chunk, [source indices])."""
    sl = np.ascontiguousarray(solve_list, dtype=np.int32)
    pp = np.ascontiguousarray(pos_parent, dtype=np.int32)
    op = np.ascontiguousarray(ori_parent, dtype=np.int32)
    d = rig.desc()
    ng, ne, nc = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    i32p = C.POINTER(C.c_int32)
    args = lambda g, e, cs: (C.byref(d), as_ptr(sl, C.c_int32), C.c_int32(len(sl)), as_ptr(pp, C.c_int32), C.c_int32(len(pp)), as_ptr(op, C.c_int32),
                             C.c_int32(len(op)), C.c_int32(units_per_chunk), g, C.byref(ng), e, C.byref(ne), cs, C.byref(nc))  # fmt: skip
    f = lib().mmx_host_f64_assembly_list
    f.restype = C.c_int32
    _check(f(*args(None, None, None)))
    groups = np.zeros(2 * max(ng.value, 1), np.uint32)
    extra = np.zeros(max(ne.value, 1), np.int32)
    cstart = np.zeros(2 * nc.value + 1, np.int32)
    _check(f(*args(groups.ctypes.data_as(C.POINTER(C.c_uint32)), extra.ctypes.data_as(i32p), cstart.ctypes.data_as(i32p))))
    chunks = []
    for ch in range(nc.value):
        entries = []
        for g in range(cstart[ch], cstart[ch + 1]):
            w0, w1 = int(groups[2 * g]), int(groups[2 * g + 1])
            c, ul, count = w0 & 0xfff, (w0 >> 12) & 0x3f, w0 >> 18
            entries.append((c, ul, [w1] if count == 1 else [int(x) for x in extra[w1 : w1 + count]]))
        chunks.append(dict(block_mask=int(cstart[nc.value + 1 + ch]), entries=entries))
    return chunks


def _stream_ptr() -> C.c_void_p:
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


class RigHandle:
    """Device-resident Skeleton + ParameterTransform (mmx_rig)."""

    def __init__(self, rig: Rig, device: int = 0):
        self.rig = rig
        self.device = int(device)
        self._h = C.c_void_p(0)
        d = rig.desc()
        _check(lib().mmx_rig_create(C.byref(d), self.device, C.byref(self._h)))

    def close(self) -> None:
        if self._h:
            lib().mmx_rig_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


# Route every Problem created from here on starts with ("auto" | "fused" | "wide" | "explicit_jacobian"): the parity tests
# set it (monkeypatch.setattr) to send whole test bodies through one route; Problem.set_route changes it per handle.
# "prefer_wide" (sweeps: MMX_TEST_ROUTE=prefer_wide, tests/conftest.py): every solve tries the wide route first and falls
# back to the library's own choice where the problem is outside the tree kernels' scope.
default_route = "auto"


class Problem:
    """One batch of independent IK instances on one GPU (mmx_problem): the batched counterpart of
    one SkeletonSolverFunction + GaussNewtonSolver per element
    (pymomentum/tensor_ik/tensor_ik.cpp:127-177)."""

    def __init__(self, rig_handle: RigHandle, batch: int, pos_parent, ori_parent):
        import torch

        self.rh = rig_handle
        self.B = int(batch)
        self.P = rig_handle.rig.num_params
        self.J = rig_handle.rig.num_joints
        self.pos_parent = np.ascontiguousarray(pos_parent, dtype=np.int32).reshape(-1)
        self.ori_parent = np.ascontiguousarray(ori_parent, dtype=np.int32).reshape(-1)
        self.Kp, self.Ko = len(self.pos_parent), len(self.ori_parent)
        self.M = 3 * self.Kp + 9 * self.Ko
        self.n = self.P
        self.device = torch.device("cuda", rig_handle.device)
        self._h = C.c_void_p(0)
        self._keep = []
        _check(
            lib().mmx_problem_create(
                rig_handle._h, self.B, self.Kp, as_ptr(self.pos_parent, C.c_int32), self.Ko,
                as_ptr(self.ori_parent, C.c_int32), C.byref(self._h),
            )
        )  # fmt: skip
        self._prefer_wide = default_route == "prefer_wide"
        if default_route not in ("auto", "prefer_wide"):
            self.set_route(default_route)

    def close(self) -> None:
        if self._h:
            lib().mmx_problem_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # -- which kernels mmx_solve runs (mmx_tuning): "auto" | "fused" | "wide" | "explicit_jacobian"
    def set_route(self, route: str, max_refinement_steps: int = 0) -> None:
        # a caller that pins a route means it: the sweep default (MMX_TEST_ROUTE=prefer_wide) steps back for this handle
        # until the caller returns to "auto" (route-pinned tests then run what they name under a forced-route sweep)
        self._prefer_wide = default_route == "prefer_wide" and route == "auto"
        self._set_tuning(route, max_refinement_steps)

    def _set_tuning(self, route: str, max_refinement_steps: int = 0) -> None:
        t = _abi.Tuning()
        t.route = _abi.ROUTES[route]
        t.max_refinement_steps = int(max_refinement_steps)  # 0 default (up to three), -1 none, 1..3
        t.mixed_tolerance, t.mixed_max_cg = getattr(self, "_mixed", (0.0, 0))  # MMX_PRECISION_MIXED: 0 = defaults (1e-7, 12)
        self._route_args = (route, int(max_refinement_steps))
        _check(lib().mmx_problem_set_tuning(self._h, C.byref(t)))

    def set_mixed(self, tolerance: float = 0.0, max_cg: int = 0) -> None:
        """mmx_tuning::mixed_tolerance / mixed_max_cg (MMX_PRECISION_MIXED); the route setting is kept."""
        self._mixed = (float(tolerance), int(max_cg))
        self._set_tuning(*getattr(self, "_route_args", ("auto", 0)))

    def last_route(self) -> str:
        r = int(lib().mmx_problem_last_route(self._h))
        return {v: k for k, v in _abi.ROUTES.items()}[r]

    # -- SolverT::setEnabledParameters
    def set_enabled(self, enabled) -> None:
        e = np.ascontiguousarray(enabled, dtype=np.uint8).reshape(-1)
        assert e.shape[0] == self.P
        _check(lib().mmx_problem_set_enabled(self._h, as_ptr(e, C.c_uint8)))
        self.n = int(np.count_nonzero(e))

    # -- setConstraints for every batch element; device tensors are borrowed, numpy arrays copied
    def set_constraints(
        self, pos_offset, pos_target, pos_weight, ori_offset, ori_target, ori_weight,
        pos_function_weight: float = 1.0, ori_function_weight: float = 1.0,
        limits=None, limit_function_weight: float = 1.0,
        model_target=None, model_weights=None, model_function_weight: float = 1.0,
        pos_loss=(2.0, 1.0), ori_loss=(2.0, 1.0), joint_blocks=None, ellipsoid_limits=None, function_weights=None,
    ) -> None:  # fmt: skip
        """limits: list of _abi.ParameterLimit (batch-shared, LimitErrorFunction);
        model_target / model_weights: [B,P] (ModelParametersErrorFunction), same memory kind as the
        constraint arrays.  Adds len(limits) + (P if model_target is given) rows to J / r.
        pos_loss / ori_loss: GeneralizedLoss (alpha, c) of the two joint-constraint blocks
        (alpha 2 = L2, 1 = L1, 0 = Cauchy, _abi.MMX_LOSS_WELSCH = Welsch, else Barron's general form).
        joint_blocks: list of _abi.JointBlock (Plane / Aim / FixedAxis / Normal error functions), payload
        of the same memory kind as the constraint arrays; their rows follow the orientation rows.
        function_weights: [B, C] per-element error-function weights (errorFunctionWeights of solveTensorIKProblem), columns
        position, orientation, limits, model parameters, joint block 0, ...; same memory kind as the constraint arrays."""
        import torch

        arrs = [pos_offset, pos_target, pos_weight, ori_offset, ori_target, ori_weight]
        shapes = [(self.B, self.Kp, 3), (self.B, self.Kp, 3), (self.B, self.Kp), (self.B, self.Ko, 4), (self.B, self.Ko, 4), (self.B, self.Ko)]
        if model_target is not None:
            arrs += [model_target, model_weights]
            shapes += [(self.B, self.P), (self.B, self.P)]
        on_dev = all(isinstance(a, torch.Tensor) for a in arrs)
        keep, ptrs = [], []
        for a, shp in zip(arrs, shapes):
            if on_dev:
                assert a.is_cuda and a.dtype == torch.float32 and a.is_contiguous() and tuple(a.shape) == shp, (a.shape, shp)
                keep.append(a)
                ptrs.append(C.c_void_p(a.data_ptr() if a.numel() else 0))
            else:
                x = np.ascontiguousarray(a, dtype=np.float32).reshape(shp)
                keep.append(x)
                ptrs.append(C.c_void_p(x.ctypes.data if x.size else 0))
        if model_target is None:
            ptrs += [C.c_void_p(0), C.c_void_p(0)]
        fw_ptr, fw_cols = C.c_void_p(0), 0
        if function_weights is not None:
            if on_dev:
                assert function_weights.is_cuda and function_weights.dtype == torch.float32 and function_weights.is_contiguous() and function_weights.shape[0] == self.B
                keep.append(function_weights)
                fw_ptr, fw_cols = C.c_void_p(function_weights.data_ptr()), int(function_weights.shape[1])
            else:
                fwh = np.ascontiguousarray(function_weights, dtype=np.float32).reshape(self.B, -1)
                keep.append(fwh)
                fw_ptr, fw_cols = C.c_void_p(fwh.ctypes.data), int(fwh.shape[1])
        limits = list(limits) if limits else []
        larr = _abi.limit_array(limits)
        blocks = list(joint_blocks) if joint_blocks else []
        bkeep: list = []
        barr = _abi.joint_block_array(blocks, bkeep, self.B, on_dev)
        ells = list(ellipsoid_limits) if ellipsoid_limits else []  # _abi.EllipsoidLimit, batch-shared
        earr = _abi.ellipsoid_array(ells)
        cd = ConstraintData(
            *ptrs[:6], float(pos_function_weight), float(ori_function_weight), _abi.MMX_MEM_DEVICE if on_dev else _abi.MMX_MEM_HOST,
            ptrs[6], ptrs[7], float(model_function_weight), len(limits), C.cast(larr, C.c_void_p) if limits else None, float(limit_function_weight),
            float(pos_loss[0]), float(pos_loss[1]), float(ori_loss[0]), float(ori_loss[1]),
            len(blocks), C.cast(barr, C.c_void_p) if blocks else None,
            len(ells), C.cast(earr, C.c_void_p) if ells else None,
            fw_ptr, fw_cols,
        )  # fmt: skip
        _check(lib().mmx_problem_set_constraints_sized(self._h, C.byref(cd), C.sizeof(cd), _stream_ptr()))
        self._keep = keep + bkeep if on_dev else []
        self.M = int(lib().mmx_problem_num_rows(self._h))

    # -- per-instance characters of one topology (characters[iBatch] of solveTensorIKProblem)
    def set_instance_rig(self, translation_offset=None, pre_rotation=None) -> None:
        """translation_offset [B,J,3], pre_rotation [B,J,4] (x,y,z,w): float32 cuda tensors (borrowed) or
        numpy arrays (copied); None = the rig's own values.  Both None restores the shared rig."""
        self._keep_rig = self._instance_call(
            lib().mmx_problem_set_instance_rig, [(translation_offset, (self.B, self.J, 3)), (pre_rotation, (self.B, self.J, 4))], "float32"
        )

    # -- per-instance ConstraintData::parent
    def set_instance_parents(self, pos_parent=None, ori_parent=None) -> None:
        """pos_parent [B,Kp], ori_parent [B,Ko]: int32 cuda tensors (borrowed) or numpy arrays (copied);
        None = the batch-shared list of the constructor."""
        self._keep_parents = self._instance_call(
            lib().mmx_problem_set_instance_parents, [(pos_parent, (self.B, self.Kp)), (ori_parent, (self.B, self.Ko))], "int32"
        )

    def _instance_call(self, fn, arrays, dtype):
        import torch

        given = [a for a, _ in arrays if a is not None]
        on_dev = bool(given) and all(isinstance(a, torch.Tensor) for a in given)
        keep, ptrs = [], []
        for a, shp in arrays:
            if a is None:
                ptrs.append(C.c_void_p(0))
            elif on_dev:
                assert a.is_cuda and str(a.dtype) == "torch." + dtype and a.is_contiguous() and tuple(a.shape) == shp, (a.shape, shp, a.dtype)
                keep.append(a)
                ptrs.append(C.c_void_p(a.data_ptr() if a.numel() else 0))
            else:
                x = np.ascontiguousarray(a, dtype=dtype).reshape(shp)
                keep.append(x)
                ptrs.append(C.c_void_p(x.ctypes.data if x.size else 0))
        _check(fn(self._h, ptrs[0], ptrs[1], _abi.MMX_MEM_DEVICE if on_dev else _abi.MMX_MEM_HOST, _stream_ptr()))
        return keep if on_dev else []

    def _theta(self, theta):
        import torch

        assert isinstance(theta, torch.Tensor) and theta.is_cuda and theta.dtype == torch.float32
        assert theta.is_contiguous() and tuple(theta.shape) == (self.B, self.P), theta.shape
        return theta

    # -- the graded kernel: dense column-major Jacobian + residual (+ error)
    def eval_jacobian(self, theta, jac=None, res=None, err=None, want_err: bool = True, row_major: bool = False):
        """Returns (jac [B,P,M] (jac[b].T is the M x P Jacobian; [B,M,P] with row_major), res [B,M], err [B] float64)."""
        import torch

        theta = self._theta(theta)
        if jac is None:
            jac = torch.empty((self.B, self.M, self.P) if row_major else (self.B, self.P, self.M), dtype=torch.float32, device=self.device)
        if res is None:
            res = torch.empty((self.B, self.M), dtype=torch.float32, device=self.device)
        if err is None and want_err:
            err = torch.empty((self.B,), dtype=torch.float64, device=self.device)
        layout = _abi.MMX_LAYOUT_ROW_MAJOR if row_major else _abi.MMX_LAYOUT_COL_MAJOR
        _check(lib().mmx_eval_jacobian(self._h, _dev(theta), _dev(jac), _dev(res), _dev(err), layout, _stream_ptr()))
        return jac, res, err

    def eval_jacobian_kernel_ms(self, theta, jac, res, err=None) -> float:
        """Duration (ms) of the J-assembly kernel itself: events attached to its dispatch packet
        (mmx_eval_jacobian_timed); what a rocprofv3 kernel trace reports for it."""
        theta = self._theta(theta)
        ms = C.c_float(0.0)
        _check(lib().mmx_eval_jacobian_timed(self._h, _dev(theta), _dev(jac), _dev(res), _dev(err), _abi.MMX_LAYOUT_COL_MAJOR,
                                             _stream_ptr(), C.byref(ms)))  # fmt: skip
        return float(ms.value)

    def store_pattern_kernel_ms(self, jac) -> float:
        """Duration (ms) of the store-only counterpart of the J-assembly kernel (mmx_debug_store_pattern)."""
        ms = C.c_float(0.0)
        _check(lib().mmx_debug_store_pattern(self._h, _dev(jac), _stream_ptr(), C.byref(ms)))
        return float(ms.value)

    def skeleton_state(self, theta):
        import torch

        theta = self._theta(theta)
        st = torch.empty((self.B, self.J, 8), dtype=torch.float32, device=self.device)
        _check(lib().mmx_eval_skeleton_state(self._h, _dev(theta), _dev(st), _stream_ptr()))
        return st

    def normal_equations(self, theta):
        import torch

        theta = self._theta(theta)
        jtj = torch.empty((self.B, self.n, self.n), dtype=torch.float32, device=self.device)
        jtr = torch.empty((self.B, self.n), dtype=torch.float32, device=self.device)
        err = torch.empty((self.B,), dtype=torch.float64, device=self.device)
        _check(lib().mmx_eval_normal_equations(self._h, _dev(theta), _dev(jtj), _dev(jtr), _dev(err), _stream_ptr()))
        return jtj, jtr, err

    def jtj_history(self, theta_init, parameter_history, iterations, regularization, out=None):
        """GaussNewtonSolverT's iterationHistory_["jtj"] (gauss_newton_solver.cpp:262-279) for a finished solve: entry i of
        element b is hessianApprox_ of its iteration i -- the lower triangle of J^T J over the enabled parameters at the
        parameters BEFORE that iteration (theta_init for i = 0, row i - 1 of parameter_history after), with the
        regularisation on the diagonal (:249 adds it in place before the factorisation; the strict upper triangle is
        never written and stays zero).  Entries from an element's iteration count on are zero (the reference leaves them
        unset).  Rebuilt with mmx_eval_normal_equations, one launch per iteration, instead of being stored by the solve:
        n^2 floats per element and iteration ([B, K, n, n]; `out` to bring the storage).  The rebuilt system comes from the
        explicit Jacobian with the single-precision pointer-jumping forward kinematics of mmx_eval_normal_equations; the solve
        kernels carry the jump rounds' partial products in double, so an entry can differ from the system the solve
        actually factored by a few ulp -- the tolerance of tests/test_gpu_edge_cases.py, not bit-exact."""
        import torch

        theta_init = self._theta(theta_init)
        K = int(parameter_history.shape[1])
        assert tuple(parameter_history.shape) == (self.B, K, self.P) and parameter_history.is_cuda
        if out is None:
            out = torch.zeros((self.B, K, self.n, self.n), dtype=torch.float32, device=self.device)
        assert tuple(out.shape) == (self.B, K, self.n, self.n) and out.dtype == torch.float32 and out.is_cuda
        its = iterations.to(self.device).to(torch.int64)
        eye = torch.eye(self.n, dtype=torch.float32, device=self.device) * float(regularization)
        for i in range(K):
            th = theta_init if i == 0 else parameter_history[:, i - 1].contiguous()
            jtj, _, _ = self.normal_equations(th)
            h = torch.tril(jtj) + eye
            out[:, i] = torch.where((its > i)[:, None, None], h, torch.zeros_like(h))
        return out

    def tree_normal_equations(self, theta):
        """Parity hook: (JtJ [B,n,n] lower triangle, Jtr [B,n]) from the tree moments (the wide route's first stage)."""
        import torch

        theta = self._theta(theta)
        jtj = torch.empty((self.B, self.n, self.n), dtype=torch.float32, device=self.device)
        jtr = torch.empty((self.B, self.n), dtype=torch.float32, device=self.device)
        _check(lib().mmx_debug_tree_normal_equations(self._h, _dev(theta), _dev(jtj), _dev(jtr), _stream_ptr()))
        # the kernels number the columns in elimination order (mmx_host_tables.hpp); hand back the enabled-list order of
        # mmx_eval_normal_equations
        lst = np.zeros(self.P, np.int32)
        n = C.c_int32(0)
        _check(lib().mmx_debug_fused_normal_equations(self._h, None, None, None, as_ptr(lst, C.c_int32), C.byref(n), None))
        assert n.value == self.n
        order = torch.from_numpy(np.argsort(lst[: self.n], kind="stable")).to(self.device)
        full = jtj + jtj.transpose(1, 2) - torch.diag_embed(torch.diagonal(jtj, dim1=1, dim2=2))
        return torch.tril(full[:, order][:, :, order]).contiguous(), jtr[:, order].contiguous()

    def tile_structure(self):
        """The tile structure the wide route factors with (mmx_problem_tile_structure): dict(row_mask, col_mask, blocks,
        tiles, products, dense_tiles, dense_products)."""
        row, col = np.zeros(32, np.uint32), np.zeros(32, np.uint32)
        nb, nt, prod = C.c_int32(0), C.c_int32(0), C.c_int64(0)
        _check(lib().mmx_problem_tile_structure(self._h, as_ptr(row, C.c_uint32), as_ptr(col, C.c_uint32), C.byref(nb), C.byref(nt), C.byref(prod)))
        NB = nb.value
        return dict(row_mask=row, col_mask=col, blocks=NB, tiles=nt.value, products=prod.value, dense_tiles=NB * (NB + 1) // 2,
                    dense_products=NB * (NB * NB - 1) // 6)  # fmt: skip

    def fused_normal_equations(self, theta):
        """Parity hook: (solve_list [n], JtJ [B,n,n], Jtr [B,n]) as the fused kernel builds them."""
        import torch

        theta = self._theta(theta)
        lst = np.zeros(self.P, np.int32)
        n = C.c_int32(0)
        _check(lib().mmx_debug_fused_normal_equations(self._h, None, None, None, as_ptr(lst, C.c_int32), C.byref(n), None))
        n = n.value
        jtj = torch.zeros((self.B, n, n), dtype=torch.float32, device=self.device)
        jtr = torch.zeros((self.B, n), dtype=torch.float32, device=self.device)
        _check(lib().mmx_debug_fused_normal_equations(self._h, _dev(theta), _dev(jtj), _dev(jtr), as_ptr(lst, C.c_int32), C.byref(C.c_int32(0)), _stream_ptr()))
        return lst[:n].copy(), jtj, jtr

    def solve_f64(self, theta, options: GnOptions, want_history: bool = False):
        """In-place batched SolverT<double>::solve (mmx_solve_f64); theta: float64 cuda tensor [B, P]."""
        import torch

        assert isinstance(theta, torch.Tensor) and theta.is_cuda and theta.dtype == torch.float64 and theta.is_contiguous() and tuple(theta.shape) == (self.B, self.P)
        out = dict(
            error=torch.empty((self.B,), dtype=torch.float64, device=self.device),
            iterations=torch.empty((self.B,), dtype=torch.int32, device=self.device),
            status=torch.empty((self.B,), dtype=torch.int32, device=self.device),
        )
        if want_history:
            out["error_history"] = torch.empty((self.B, max(1, options.max_iterations)), dtype=torch.float64, device=self.device)
        _check(lib().mmx_solve_f64(self._h, C.byref(options), _dev(theta), _dev(out["error"]), _dev(out["iterations"]), _dev(out["status"]),
                                   _dev(out.get("error_history")), _stream_ptr()))  # fmt: skip
        out["theta"] = theta
        return out

    def solve_diagnostics(self):
        """[B, 4] float tensor of the last single-precision solve (mmx_problem_solve_diagnostics): precision estimate,
        smallest pivot ratio, largest refinement ratio, |theta|."""
        import torch

        out = torch.empty((self.B, 4), dtype=torch.float32, device=self.device)
        _check(lib().mmx_problem_solve_diagnostics(self._h, _dev(out), _stream_ptr()))
        return out

    def solve(self, theta, options: GnOptions, want_history: bool = False, outputs=None, want_parameter_history: bool = False, want_step_history: bool = False):
        """In-place batched SolverT::solve.  Returns dict(theta, error, iterations, status[, error_history]
        [, parameter_history [B, max_iterations, P]][, step_history [B, max_iterations, 2]: (lambda, gain ratio) per iteration
        of the LM schedule])."""
        import torch

        theta = self._theta(theta)
        if getattr(self, "_prefer_wide", False):  # sweeps (MMX_TEST_ROUTE=prefer_wide): the wide route wherever it applies
            self._prefer_wide = False
            keep = theta.clone()
            try:
                self._set_tuning("wide")
                return self.solve(theta, options, want_history, outputs, want_parameter_history, want_step_history)
            except MmxError as e:
                if e.code != 4:  # MMX_ERR_UNSUPPORTED: outside the tree kernels' scope -> the library's own choice
                    raise
                theta.copy_(keep)
                self._set_tuning("auto")
            finally:
                self._prefer_wide = True
                import sys

                if sys.exc_info()[0] is None:  # (an exception on its way out is not to be masked by a second library call)
                    if self.last_route() == "wide":
                        self._set_tuning("auto")
                else:
                    try:
                        self._set_tuning("auto")
                    except Exception:
                        pass
        if outputs is None:
            outputs = dict(
                error=torch.empty((self.B,), dtype=torch.float64, device=self.device),
                iterations=torch.empty((self.B,), dtype=torch.int32, device=self.device),
                status=torch.empty((self.B,), dtype=torch.int32, device=self.device),
            )
            if want_history:
                outputs["error_history"] = torch.empty((self.B, max(1, options.max_iterations)), dtype=torch.float64, device=self.device)
        if want_parameter_history and "parameter_history" not in outputs:
            outputs["parameter_history"] = torch.empty((self.B, max(1, options.max_iterations), self.P), dtype=torch.float32, device=self.device)
        if want_step_history and "step_history" not in outputs:
            outputs["step_history"] = torch.empty((self.B, max(1, options.max_iterations), 2), dtype=torch.float64, device=self.device)
        if outputs.get("step_history") is not None:
            _check(
                lib().mmx_solve_with_step_history(
                    self._h, C.byref(options), _dev(theta), _dev(outputs["error"]), _dev(outputs["iterations"]),
                    _dev(outputs["status"]), _dev(outputs.get("error_history")), _dev(outputs.get("parameter_history")),
                    _dev(outputs["step_history"]), _stream_ptr(),
                )
            )  # fmt: skip
            outputs["theta"] = theta
            return outputs
        if outputs.get("parameter_history") is not None:
            _check(
                lib().mmx_solve_with_history(
                    self._h, C.byref(options), _dev(theta), _dev(outputs["error"]), _dev(outputs["iterations"]),
                    _dev(outputs["status"]), _dev(outputs.get("error_history")), _dev(outputs["parameter_history"]), _stream_ptr(),
                )
            )  # fmt: skip
            outputs["theta"] = theta
            return outputs
        _check(
            lib().mmx_solve(
                self._h, C.byref(options), _dev(theta), _dev(outputs["error"]), _dev(outputs["iterations"]),
                _dev(outputs["status"]), _dev(outputs.get("error_history")), _stream_ptr(),
            )
        )  # fmt: skip
        outputs["theta"] = theta
        return outputs


COMM_ID_BYTES = 128


class Comm:
    """The path's one multi-GPU exchange (mmx_comm): an RCCL all-reduce of the three residual norms per solve,
    one rank per GPU.  Rank 0 makes the id with Comm.unique_id() and hands the 128 bytes to the other ranks
    through any side channel (bench.py: a torch.distributed broadcast)."""

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        _check(lib().mmx_comm_unique_id(C.cast(buf, C.c_void_p)))
        return bytes(buf)

    def __init__(self, comm_id: bytes, world_size: int, rank: int, device: int):
        assert len(comm_id) == COMM_ID_BYTES
        self._h = C.c_void_p(0)
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(comm_id)
        _check(lib().mmx_comm_create(C.cast(buf, C.c_void_p), int(world_size), int(rank), int(device), C.byref(self._h)))

    @property
    def world_size(self) -> int:
        return int(lib().mmx_comm_world_size(self._h))

    @property
    def rank(self) -> int:
        return int(lib().mmx_comm_rank(self._h))

    def all_reduce_norms(self, norms) -> None:
        """In-place sum over the ranks of a float64[3] cuda tensor, on torch's current stream."""
        assert norms.is_cuda and str(norms.dtype) == "torch.float64" and norms.numel() == 3 and norms.is_contiguous()
        _check(lib().mmx_comm_all_reduce_norms(self._h, _dev(norms), _stream_ptr()))

    def close(self) -> None:
        if self._h:
            lib().mmx_comm_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def residual_norms(outputs: dict, norms) -> None:
    """(sum final error, sum iterations, #failed) of a solve's outputs -> norms (float64[3] cuda tensor): one
    small kernel on torch's current stream, fixed summation order."""
    e, it, st = outputs["error"], outputs["iterations"], outputs["status"]
    _check(lib().mmx_residual_norms(int(e.numel()), _dev(e), _dev(it), _dev(st), _dev(norms), _stream_ptr()))
