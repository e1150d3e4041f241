"""ctypes mirrors of the POD structs in include/mmx.h (the C-ABI boundary).

Pure plumbing: no compute here.  Field order and types must match include/mmx.h exactly;
tests/test_abi.py checks sizes and that the built library exports every declared symbol.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

c_float_p = C.POINTER(C.c_float)
c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)

MMX_ABI_VERSION = 11
MMX_OK = 0
MMX_SOLVE_OK, MMX_SOLVE_NONFINITE, MMX_SOLVE_NOT_PD = 0, 1, 2
MMX_SOLVE_DAMPING_FLOORED = 4  # informational bit of status[] (include/mmx.h)
MMX_SOLVE_PRECISION_SUSPECT = 8  # informational: the single-precision solve's precision estimate exceeds options.precision_bound
MMX_SOLVE_ESCALATED_F64 = 16  # informational: MMX_PRECISION_AUTO re-solved the element in double
MMX_SOLVE_MIXED = 32  # informational: the element was solved by the mixed-precision instantiation (MMX_PRECISION_MIXED / AUTO)
MMX_SOLVE_ERROR_MASK = 3
MMX_PRECISION_F32, MMX_PRECISION_F64, MMX_PRECISION_AUTO, MMX_PRECISION_MIXED = 0, 1, 2, 3
MMX_MEM_HOST, MMX_MEM_DEVICE = 0, 1
MMX_LAYOUT_COL_MAJOR, MMX_LAYOUT_ROW_MAJOR = 0, 1
MMX_STEP_GN_FIXED_LAMBDA, MMX_STEP_LM_SCHEDULE, MMX_STEP_TRUST_REGION = 0, 1, 2


class RigDesc(C.Structure):
    _fields_ = [
        ("num_joints", C.c_int32),
        ("num_params", C.c_int32),
        ("parent", c_int32_p),
        ("pre_rotation", c_float_p),
        ("translation_offset", c_float_p),
        ("pt_outer", c_int32_p),
        ("pt_inner", c_int32_p),
        ("pt_value", c_float_p),
        ("pt_offsets", c_float_p),
    ]


MMX_LOSS_WELSCH = float(np.finfo(np.float32).min)  # GeneralizedLossT::kWelsch
MMX_LIMIT_MINMAX = 0  # momentum::LimitType values (character/parameter_limits.h:20-31)
MMX_LIMIT_MINMAX_JOINT = 1
MMX_LIMIT_MINMAX_JOINT_PASSIVE = 2  # accepted and ignored, like LimitErrorFunctionT does
MMX_LIMIT_LINEAR = 3
MMX_LIMIT_LINEAR_JOINT = 4
MMX_LIMIT_HALFPLANE = 6


class ParameterLimit(C.Structure):
    """mmx_parameter_limit: one model-parameter entry of Character::parameterLimits."""

    _fields_ = [
        ("type", C.c_int32),
        ("index0", C.c_int32),
        ("index1", C.c_int32),
        ("weight", C.c_float),
        ("v", C.c_float * 4),
    ]

    @classmethod
    def minmax(cls, param, lo, hi, weight=1.0):
        return cls(MMX_LIMIT_MINMAX, int(param), 0, float(weight), (C.c_float * 4)(lo, hi, 0.0, 0.0))

    @classmethod
    def linear(cls, reference, target, scale, offset, range_min=0.0, range_max=0.0, weight=1.0):
        return cls(MMX_LIMIT_LINEAR, int(reference), int(target), float(weight), (C.c_float * 4)(scale, offset, range_min, range_max))

    @classmethod
    def minmax_joint(cls, joint, joint_parameter, lo, hi, weight=1.0):
        return cls(MMX_LIMIT_MINMAX_JOINT, 7 * int(joint) + int(joint_parameter), 0, float(weight), (C.c_float * 4)(lo, hi, 0.0, 0.0))

    @classmethod
    def minmax_joint_passive(cls, joint, joint_parameter, lo, hi, weight=1.0):
        return cls(MMX_LIMIT_MINMAX_JOINT_PASSIVE, 7 * int(joint) + int(joint_parameter), 0, float(weight), (C.c_float * 4)(lo, hi, 0.0, 0.0))

    @classmethod
    def linear_joint(cls, ref_joint, ref_parameter, tgt_joint, tgt_parameter, scale, offset, range_min=0.0, range_max=0.0, weight=1.0):
        return cls(MMX_LIMIT_LINEAR_JOINT, 7 * int(ref_joint) + int(ref_parameter), 7 * int(tgt_joint) + int(tgt_parameter),
                   float(weight), (C.c_float * 4)(scale, offset, range_min, range_max))  # fmt: skip

    @classmethod
    def halfplane(cls, param1, param2, n0, n1, offset, weight=1.0):
        return cls(MMX_LIMIT_HALFPLANE, int(param1), int(param2), float(weight), (C.c_float * 4)(n0, n1, offset, 0.0))


def limit_array(limits):
    """ctypes array (kept alive by the caller) of ParameterLimit."""
    arr = (ParameterLimit * max(len(limits), 1))()
    for i, l in enumerate(limits):
        arr[i] = l
    return arr


class EllipsoidLimit(C.Structure):
    """mmx_ellipsoid_limit: LimitType::Ellipsoid entry (character/parameter_limits.h:77-84)."""

    _fields_ = [
        ("ellipsoid", C.c_float * 12),
        ("ellipsoid_inv", C.c_float * 12),
        ("offset", C.c_float * 3),
        ("weight", C.c_float),
        ("ellipsoid_parent", C.c_int32),
        ("parent", C.c_int32),
    ]

    @classmethod
    def make(cls, parent, offset, ellipsoid_parent, translation, euler_zyx_deg, scale, weight=1.0):
        """The text form's arguments (parseEllipsoid, io/skeleton/parameter_limits_io.cpp:580-610):
        linear part = R(extrinsic XYZ of the reversed, radian angles) * diag(scale)."""
        ez = np.radians(np.asarray(euler_zyx_deg, np.float32).astype(np.float64))
        ax, ay, az = ez[2], ez[1], ez[0]
        cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
        R = np.array([[cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx],
                      [sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx],
                      [-sy, cy * sx, cy * cx]])  # Rz Ry Rx = eulerXYZToRotationMatrix(., Extrinsic), math/utility.cpp:315-380
        A = np.eye(4)
        A[:3, :3] = R @ np.diag(np.asarray(scale, np.float64))
        A[:3, 3] = np.asarray(translation, np.float64)
        return cls.from_affine(parent, offset, ellipsoid_parent, A[:3], weight)

    @classmethod
    def from_affine(cls, parent, offset, ellipsoid_parent, affine34, weight=1.0):
        A = np.eye(4)
        A[:3] = np.asarray(affine34, np.float64).reshape(3, 4)
        Ai = np.linalg.inv(A)
        f12 = lambda m: (C.c_float * 12)(*[float(x) for x in np.asarray(m[:3], np.float32).reshape(-1)])
        return cls(f12(A), f12(Ai), (C.c_float * 3)(*[float(x) for x in offset]), float(weight), int(ellipsoid_parent), int(parent))


def ellipsoid_array(items):
    arr = (EllipsoidLimit * max(len(items), 1))()
    for i, e in enumerate(items):
        arr[i] = e
    return arr


MMX_JC_PLANE, MMX_JC_HALF_PLANE, MMX_JC_AIM_DIST, MMX_JC_AIM_DIR = 0, 1, 2, 3
MMX_JC_FIXED_AXIS_DIFF, MMX_JC_FIXED_AXIS_COS, MMX_JC_FIXED_AXIS_ANGLE, MMX_JC_NORMAL = 4, 5, 6, 7
MMX_MAX_JOINT_BLOCKS = 8


def jc_func_dim(type_: int) -> int:
    """FuncDim of the block's error function (rows per constraint)."""
    return 3 if type_ in (MMX_JC_AIM_DIST, MMX_JC_AIM_DIR, MMX_JC_FIXED_AXIS_DIFF) else 1


class JointConstraintBlock(C.Structure):
    """mmx_joint_constraint_block."""

    _fields_ = [
        ("type", C.c_int32),
        ("count", C.c_int32),
        ("parent", C.c_void_p),
        ("local_point", C.c_void_p),
        ("local_dir", C.c_void_p),
        ("global_", C.c_void_p),
        ("plane_d", C.c_void_p),
        ("weight", C.c_void_p),
        ("function_weight", C.c_float),
        ("loss_alpha", C.c_float),
        ("loss_c", C.c_float),
    ]


class JointBlock:
    """Python-side description of one further joint-constraint block (Plane / Aim / FixedAxis /
    Normal error function).  Payload arrays are numpy ([K,..] for one instance or [B,K,..]) or,
    for the device path, contiguous float32 cuda tensors [B,K,..]."""

    FIELDS = (("local_point", 3), ("local_dir", 3), ("global_", 3), ("plane_d", 0), ("weight", 0))

    def __init__(self, type, parent, weight, global_, local_point=None, local_dir=None, plane_d=None,
                 function_weight: float = 1.0, loss=(2.0, 1.0)):  # fmt: skip
        self.type = int(type)
        self.parent = np.ascontiguousarray(parent, dtype=np.int32).reshape(-1)
        self.count = int(self.parent.shape[0])
        self.weight, self.global_ = weight, global_
        self.local_point, self.local_dir, self.plane_d = local_point, local_dir, plane_d
        self.function_weight = float(function_weight)
        self.loss = (float(loss[0]), float(loss[1]))

    @property
    def rows(self) -> int:
        return jc_func_dim(self.type) * self.count

    def instance(self, b: int) -> "JointBlock":
        def cut(a, d):
            if a is None:
                return None
            a = np.asarray(a, dtype=np.float32)
            shp = (-1, self.count, d) if d else (-1, self.count)
            return a.reshape(shp)[b]

        return JointBlock(self.type, self.parent, cut(self.weight, 0), cut(self.global_, 3), cut(self.local_point, 3),
                          cut(self.local_dir, 3), cut(self.plane_d, 0), self.function_weight, self.loss)  # fmt: skip

    def struct(self, keep: list, batch=None, device: bool = False) -> JointConstraintBlock:
        """ctypes struct; arrays it points to are appended to `keep`.  batch = B checks [B,K,..] shapes."""
        ptrs = {}
        for name, d in self.FIELDS:
            a = getattr(self, name)
            if a is None:
                ptrs[name] = None
                continue
            if device:
                shp = (batch, self.count, d) if d else (batch, self.count)
                assert a.is_cuda and a.is_contiguous() and tuple(a.shape) == shp, (name, tuple(a.shape), shp)
                keep.append(a)
                ptrs[name] = C.c_void_p(a.data_ptr() if a.numel() else 0)
            else:
                x = np.ascontiguousarray(a, dtype=np.float32)
                if batch is not None:
                    x = x.reshape((batch, self.count, d) if d else (batch, self.count))
                keep.append(x)
                ptrs[name] = C.c_void_p(x.ctypes.data if x.size else 0)
        keep.append(self.parent)
        return JointConstraintBlock(
            self.type, self.count, C.c_void_p(self.parent.ctypes.data if self.count else 0), ptrs["local_point"], ptrs["local_dir"],
            ptrs["global_"], ptrs["plane_d"], ptrs["weight"], self.function_weight, self.loss[0], self.loss[1],
        )  # fmt: skip


def joint_block_array(blocks, keep: list, batch=None, device: bool = False):
    arr = (JointConstraintBlock * max(len(blocks), 1))()
    for i, blk in enumerate(blocks):
        arr[i] = blk.struct(keep, batch, device)
    return arr


class ConstraintData(C.Structure):
    _fields_ = [
        ("pos_offset", C.c_void_p),
        ("pos_target", C.c_void_p),
        ("pos_weight", C.c_void_p),
        ("ori_offset", C.c_void_p),
        ("ori_target", C.c_void_p),
        ("ori_weight", C.c_void_p),
        ("pos_function_weight", C.c_float),
        ("ori_function_weight", C.c_float),
        ("memory", C.c_int32),
        # optional parameter-space blocks
        ("model_target", C.c_void_p),
        ("model_weights", C.c_void_p),
        ("model_function_weight", C.c_float),
        ("num_limits", C.c_int32),
        ("limits", C.c_void_p),
        ("limit_function_weight", C.c_float),
        # GeneralizedLoss(alpha, c) of the position / orientation blocks; c <= 0 = default L2
        ("pos_loss_alpha", C.c_float),
        ("pos_loss_c", C.c_float),
        ("ori_loss_alpha", C.c_float),
        ("ori_loss_c", C.c_float),
        # further joint-constraint blocks (host array of JointConstraintBlock)
        ("num_joint_blocks", C.c_int32),
        ("joint_blocks", C.c_void_p),
        # Ellipsoid entries of the limit block (host array of EllipsoidLimit)
        ("num_ellipsoid_limits", C.c_int32),
        ("ellipsoid_limits", C.c_void_p),
        # per-element error-function weights [B][num_function_weights] (columns: position, orientation, limits, model, blocks...)
        ("function_weights", C.c_void_p),
        ("num_function_weights", C.c_int32),
    ]


class GnOptions(C.Structure):
    """POD mirror of SolverOptions + GaussNewtonSolverOptions
    (momentum/solver/solver.h:19-34, gauss_newton_solver.h:17-59)."""

    _fields_ = [
        ("min_iterations", C.c_int32),
        ("max_iterations", C.c_int32),
        ("threshold", C.c_float),
        ("regularization", C.c_float),
        ("do_line_search", C.c_int32),
        ("step_rule", C.c_int32),
        ("lm_lambda_min", C.c_float),
        ("lm_lambda_max", C.c_float),
        ("lm_up", C.c_float),
        ("lm_down", C.c_float),
        ("trust_region_radius", C.c_float),
        ("precision", C.c_int32),
        ("precision_bound", C.c_float),
    ]

    @classmethod
    def make(
        cls,
        min_iterations=1,
        max_iterations=2,
        threshold=1.0,
        regularization=0.05,
        do_line_search=False,
        step_rule=MMX_STEP_GN_FIXED_LAMBDA,
        lm_lambda_min=1e-6,
        lm_lambda_max=1e6,
        lm_up=4.0,
        lm_down=0.5,
        trust_region_radius=1.0,
        precision=MMX_PRECISION_F32,
        precision_bound=1e-5,
    ) -> "GnOptions":
        return cls(
            int(min_iterations),
            int(max_iterations),
            float(threshold),
            float(regularization),
            int(do_line_search),  # MMX_LINE_SEARCH_*: False/0 none, True/1 GaussNewtonSolverT rule, 2 SubsetGN / GN-QR rule
            int(step_rule),
            float(lm_lambda_min),
            float(lm_lambda_max),
            float(lm_up),
            float(lm_down),
            float(trust_region_radius),
            int(precision),
            float(precision_bound),
        )


ROUTES = {"auto": 0, "fused": 1, "wide": 2, "explicit_jacobian": 3}  # MMX_ROUTE_*


class Tuning(C.Structure):
    """mmx_tuning: which kernels mmx_solve runs (include/mmx.h)."""

    _fields_ = [("route", C.c_int32), ("max_refinement_steps", C.c_int32), ("mixed_tolerance", C.c_float), ("mixed_max_cg", C.c_int32), ("reserved", C.c_int32 * 4)]


def as_ptr(a: np.ndarray, ctype):
    """Typed pointer to a C-contiguous numpy array (caller keeps `a` alive)."""
    assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
    return a.ctypes.data_as(C.POINTER(ctype))


def void_p(a) -> C.c_void_p:
    """void* of a numpy array, a raw device address (int) or None."""
    if a is None:
        return C.c_void_p(0)
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return C.c_void_p(a.ctypes.data)
    return C.c_void_p(int(a))
