"""pymomentum.solver2-shaped Python surface over the C ABI.

Same class and argument names as the reference's binding (pymomentum/solver2/solver2_pybind.cpp:
432-560 SkeletonSolverFunction, :656-740 options, :876-916 Solver / GaussNewtonSolver;
solver2_error_functions.cpp:340-442 PositionErrorFunction, :1094-1200 OrientationErrorFunction,
:263-283 ModelParametersErrorFunction; solver2_distance_error_functions.cpp:93-201 NormalErrorFunction,
:430-530 PlaneErrorFunction; solver2_aim_axis_error_functions.cpp:57-282 Aim* / FixedAxis*), so a
call site written against pymomentum.solver2 reads the same here.  Differences, all additive:

* everything is BATCHED: `solve` takes model parameters [P] or [B, P]; constraint payloads
  (targets, offsets, weights) may carry a leading batch dimension [B, K, ...] and are broadcast
  otherwise.  Parents are shared by the batch.
* compute runs on the GPU through include/mmx.h (momentum_amd/libmmx_hip.so).  There is no CPU
  path: without the library or a GPU every compute call raises.

Host-side plumbing only; no arithmetic of the hot path lives here.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _abi
from ._abi import EllipsoidLimit, GnOptions, JointBlock, ParameterLimit
from .rigs import Rig


# ---------------------------------------------------------------------------------------------
# Character: the two members the solver reads (skeleton + parameter transform)
# ---------------------------------------------------------------------------------------------
class _Sized:
    def __init__(self, size: int, names: Sequence[str]):
        self.size = int(size)
        self.names = list(names)


class Character:
    """momentum::Character restricted to `skeleton` and `parameter_transform`
    (momentum/character/character.h:32-125; the solver reads only these two,
    skeleton_solver_function.cpp:30-33)."""

    def __init__(self, rig: Rig, device: int = 0):
        self.rig = rig
        self.device = int(device)
        self.skeleton = _Sized(rig.num_joints, rig.joint_names)
        self.parameter_transform = _Sized(rig.num_params, rig.param_names)
        self.parameter_limits: List[ParameterLimit] = []
        self._handle = None

    def handle(self):
        from . import capi

        if self._handle is None:
            self._handle = capi.RigHandle(self.rig, self.device)
        return self._handle


def model_parameters_to_skeleton_state(character: Character, model_parameters):
    """pymomentum.geometry.model_parameters_to_skeleton_state: [.., J, 8] = (tx,ty,tz, rx,ry,rz,rw, s)
    world transforms (SkeletonStateT::set, character/skeleton_state.cpp:87-121), on the GPU."""
    import torch

    from . import capi

    mp = np.ascontiguousarray(model_parameters, dtype=np.float32)
    single = mp.ndim == 1
    mp = mp.reshape(-1, character.parameter_transform.size)
    pb = capi.Problem(character.handle(), mp.shape[0], [], [])
    st = pb.skeleton_state(torch.from_numpy(mp).to(pb.device)).cpu().numpy()
    pb.close()
    return st[0] if single else st


# ---------------------------------------------------------------------------------------------
# error functions
# ---------------------------------------------------------------------------------------------
class SkeletonErrorFunction:
    """Base: SkeletonErrorFunction::weight_ and the character it was built for
    (character_solver/skeleton_error_function.h:44-141)."""

    def __init__(self, character: Character, weight: float = 1.0):
        self.character = character
        self.weight = float(weight)


@dataclass
class _Constraint:
    parent: int
    weight: float
    name: str
    data: dict  # field name -> array ([..] or [B, ..])


class _JointErrorFunction(SkeletonErrorFunction):
    """JointErrorFunctionT<T, Data, FuncDim, NumVec, NumPos> with GeneralizedLoss(alpha, c)
    (character_solver/joint_error_function.h:44-225)."""

    FIELDS: Sequence = ()  # (name, length) of the per-constraint payload

    def __init__(self, character: Character, alpha: float = 2.0, c: float = 1.0, weight: float = 1.0):
        super().__init__(character, weight)
        self.alpha, self.c = float(alpha), float(c)
        self._constraints: List[_Constraint] = []

    @property
    def constraints(self):
        return list(self._constraints)

    def clear_constraints(self) -> None:
        self._constraints.clear()

    def _add(self, parent, weight, name, **data) -> None:
        parent = int(parent)
        if parent < 0 or parent >= self.character.skeleton.size:
            raise RuntimeError(f"Invalid parent index {parent}")  # validateJointIndex in the reference binding
        self._constraints.append(_Constraint(parent, weight, name, data))

    def _add_many(self, parent, weight, name, **arrays) -> None:
        parent = np.asarray(parent, dtype=np.int64).reshape(-1)
        K = parent.shape[0]
        w = np.ones(K, np.float32) if weight is None else np.asarray(weight, dtype=np.float32)
        dims = dict(self.FIELDS)
        for k in range(K):
            row = {}
            for f, a in arrays.items():
                a = np.asarray(a, dtype=np.float32)
                row[f] = a[..., k, :] if dims[f] else a[..., k]
            self._add(parent[k], w[..., k], "" if name is None else name[k], **row)

    # batch payload: parents [K], weights [B, K], fields [B, K, d]
    def _stack(self, B: int):
        K = len(self._constraints)
        parents = np.array([c.parent for c in self._constraints], np.int32)
        weights = np.zeros((B, K), np.float32)
        out = {f: np.zeros((B, K, d) if d else (B, K), np.float32) for f, d in self.FIELDS}
        for k, c in enumerate(self._constraints):
            weights[:, k] = np.broadcast_to(np.asarray(c.weight, np.float32), (B,))
            for f, d in self.FIELDS:
                out[f][:, k] = np.broadcast_to(np.asarray(c.data[f], np.float32), (B, d) if d else (B,))
        return parents, weights, out


class PositionErrorFunction(_JointErrorFunction):
    """PositionErrorFunctionT (character_solver/position_error_function.h:34-70)."""

    FIELDS = (("offset", 3), ("target", 3))

    def add_constraint(self, parent, target, offset=None, weight: float = 1.0, name: str = "") -> None:
        self._add(parent, weight, name, offset=np.zeros(3, np.float32) if offset is None else offset, target=target)

    def add_constraints(self, parent, target, offset=None, weight=None, name=None) -> None:
        target = np.asarray(target, dtype=np.float32)
        off = np.zeros_like(target) if offset is None else offset
        self._add_many(parent, weight, name, offset=off, target=target)


class OrientationErrorFunction(_JointErrorFunction):
    """OrientationErrorFunctionT (orientation_error_function.h:41-66); quaternions (x, y, z, w)."""

    FIELDS = (("offset", 4), ("target", 4))

    def add_constraint(self, target, parent, offset=None, weight: float = 1.0, name: str = "") -> None:
        self._add(parent, weight, name, offset=np.array([0, 0, 0, 1], np.float32) if offset is None else offset, target=target)

    def add_constraints(self, target, parent, offset=None, weight=None, name=None) -> None:
        target = np.asarray(target, dtype=np.float32)
        if offset is None:
            offset = np.zeros_like(target)
            offset[..., 3] = 1.0
        self._add_many(parent, weight, name, offset=offset, target=target)


class _BlockErrorFunction(_JointErrorFunction):
    """Error functions that map to one mmx_joint_constraint_block."""

    TYPE = -1
    MAP = {}  # block field -> constraint field

    def block(self, B: int) -> JointBlock:
        parents, weights, data = self._stack(B)
        kw = {bf: data[cf] for bf, cf in self.MAP.items()}
        return JointBlock(self._type(), parents, weights, function_weight=self.weight, loss=(self.alpha, self.c), **kw)

    def _type(self) -> int:
        return self.TYPE


class PlaneErrorFunction(_BlockErrorFunction):
    """PlaneErrorFunctionT (plane_error_function.h:47-101); above=True is the half-plane variant."""

    FIELDS = (("offset", 3), ("normal", 3), ("d", 0))
    MAP = {"local_point": "offset", "global_": "normal", "plane_d": "d"}

    def __init__(self, character, above: bool = False, alpha: float = 2.0, c: float = 1.0, weight: float = 1.0):
        super().__init__(character, alpha, c, weight)
        self.above = bool(above)

    def _type(self) -> int:
        return _abi.MMX_JC_HALF_PLANE if self.above else _abi.MMX_JC_PLANE

    def add_constraint(self, offset, normal, d, parent, weight: float = 1.0, name: str = "") -> None:
        self._add(parent, weight, name, offset=offset, normal=normal, d=d)

    def add_constraints(self, normal, d, parent, offset=None, weight=None, name=None) -> None:
        normal = np.asarray(normal, dtype=np.float32)
        self._add_many(parent, weight, name, offset=np.zeros_like(normal) if offset is None else offset, normal=normal, d=d)


class _AimErrorFunction(_BlockErrorFunction):
    FIELDS = (("local_point", 3), ("local_dir", 3), ("global_target", 3))
    MAP = {"local_point": "local_point", "local_dir": "local_dir", "global_": "global_target"}

    def add_constraint(self, local_point, local_dir, global_target, parent, weight: float = 1.0, name: str = "") -> None:
        self._add(parent, weight, name, local_point=local_point, local_dir=local_dir, global_target=global_target)

    def add_constraints(self, local_point, local_dir, global_target, parent_index, weight=None, name=None) -> None:
        self._add_many(parent_index, weight, name, local_point=local_point, local_dir=local_dir, global_target=global_target)


class AimDistErrorFunction(_AimErrorFunction):
    """AimDistErrorFunctionT (aim_error_function.h:44-81)."""

    TYPE = _abi.MMX_JC_AIM_DIST


class AimDirErrorFunction(_AimErrorFunction):
    """AimDirErrorFunctionT (aim_error_function.h:83-114)."""

    TYPE = _abi.MMX_JC_AIM_DIR


class _FixedAxisErrorFunction(_BlockErrorFunction):
    FIELDS = (("local_axis", 3), ("global_axis", 3))
    MAP = {"local_dir": "local_axis", "global_": "global_axis"}

    def add_constraint(self, local_axis, global_axis, parent, weight: float = 1.0, name: str = "") -> None:
        self._add(parent, weight, name, local_axis=local_axis, global_axis=global_axis)

    def add_constraints(self, local_axis, global_axis, parent_index, weight=None, name=None) -> None:
        self._add_many(parent_index, weight, name, local_axis=local_axis, global_axis=global_axis)


class FixedAxisDiffErrorFunction(_FixedAxisErrorFunction):
    TYPE = _abi.MMX_JC_FIXED_AXIS_DIFF


class FixedAxisCosErrorFunction(_FixedAxisErrorFunction):
    TYPE = _abi.MMX_JC_FIXED_AXIS_COS


class FixedAxisAngleErrorFunction(_FixedAxisErrorFunction):
    TYPE = _abi.MMX_JC_FIXED_AXIS_ANGLE


class NormalErrorFunction(_BlockErrorFunction):
    """NormalErrorFunctionT (normal_error_function.h:42-73)."""

    TYPE = _abi.MMX_JC_NORMAL
    FIELDS = (("local_point", 3), ("local_normal", 3), ("global_point", 3))
    MAP = {"local_point": "local_point", "local_dir": "local_normal", "global_": "global_point"}

    def add_constraint(self, local_normal, global_point, parent, local_point=None, weight: float = 1.0, name: str = "") -> None:
        self._add(parent, weight, name, local_point=np.zeros(3, np.float32) if local_point is None else local_point,
                  local_normal=local_normal, global_point=global_point)  # fmt: skip

    def add_constraints(self, local_normal, global_point, parent, local_point=None, weight=None, name=None) -> None:
        local_normal = np.asarray(local_normal, dtype=np.float32)
        self._add_many(parent, weight, name, local_point=np.zeros_like(local_normal) if local_point is None else local_point,
                       local_normal=local_normal, global_point=global_point)  # fmt: skip


class LimitErrorFunction(SkeletonErrorFunction):
    """LimitErrorFunctionT on the character's parameter limits (limit_error_function.h:45-110);
    limit types MinMax / MinMaxJoint / Linear / LinearJoint / HalfPlane (ParameterLimit) and Ellipsoid
    (EllipsoidLimit) in one list, like Character::parameterLimits."""

    def __init__(self, character: Character, limits: Optional[Sequence[ParameterLimit]] = None, weight: float = 1.0):
        super().__init__(character, weight)
        self.limits = list(character.parameter_limits if limits is None else limits)

    def set_limits(self, limits: Sequence[ParameterLimit]) -> None:
        self.limits = list(limits)


class ModelParametersErrorFunction(SkeletonErrorFunction):
    """ModelParametersErrorFunctionT (model_parameters_error_function.h:24-66)."""

    def __init__(self, character: Character, target_parameters=None, weights=None, weight: float = 1.0):
        super().__init__(character, weight)
        P = character.parameter_transform.size
        self.target_parameters = np.zeros(P, np.float32)
        self.target_weights = np.zeros(P, np.float32)
        if target_parameters is not None:
            self.set_target_parameters(target_parameters, weights)

    def set_target_parameters(self, target_parameters, weights=None) -> None:
        P = self.character.parameter_transform.size
        t = np.asarray(target_parameters, dtype=np.float32)
        if t.shape[-1] != P:
            raise RuntimeError(f"Expected target parameters of size {P}")
        self.target_parameters = t
        self.target_weights = np.ones_like(t) if weights is None else np.asarray(weights, dtype=np.float32)


# ---------------------------------------------------------------------------------------------
# solver function and solver
# ---------------------------------------------------------------------------------------------
class SolverFunction:
    pass


class SkeletonSolverFunction(SolverFunction):
    """SkeletonSolverFunctionT (character_solver/skeleton_solver_function.h:21-95) for a batch."""

    def __init__(self, character: Character, error_functions: Sequence[SkeletonErrorFunction] = ()):
        self.character = character
        self._efs: List[SkeletonErrorFunction] = []
        self._cache = None
        for e in error_functions:
            self.add_error_function(e)

    def add_error_function(self, error_function: SkeletonErrorFunction) -> None:
        if error_function.character.rig is not self.character.rig:  # validateErrorFunctionMatchesCharacter
            raise RuntimeError("Error function was created for a different character")
        self._efs.append(error_function)

    def clear_error_functions(self) -> None:
        self._efs.clear()

    @property
    def error_functions(self):
        return list(self._efs)

    @error_functions.setter
    def error_functions(self, efs) -> None:
        self.clear_error_functions()
        for e in efs:
            self.add_error_function(e)

    def get_num_parameters(self) -> int:
        return self.character.parameter_transform.size

    # ---- lowering to one mmx_problem
    def _merged(self, cls, B: int, width: int):
        """all error functions of class `cls` as one constraint list; the function weight is folded
        into the constraint weights (w = c.weight * weight_, joint_error_function-inl.h:205)"""
        fns = [e for e in self._efs if isinstance(e, cls) and not isinstance(e, _BlockErrorFunction)]  # user subclasses included
        if len({(e.alpha, e.c) for e in fns}) > 1:
            raise RuntimeError(f"{cls.__name__}: error functions with different losses cannot share a problem")
        parents, weights, offs, tgts = [], [], [], []
        for e in fns:
            p, w, d = e._stack(B)
            parents.append(p)
            weights.append(w * np.float32(e.weight))
            offs.append(d["offset"])
            tgts.append(d["target"])
        cat = lambda xs, shp: np.concatenate(xs, axis=1) if xs else np.zeros(shp, np.float32)
        loss = (fns[0].alpha, fns[0].c) if fns else (2.0, 1.0)
        return (np.concatenate(parents) if parents else np.zeros(0, np.int32), cat(offs, (B, 0, width)), cat(tgts, (B, 0, width)),
                cat(weights, (B, 0)), loss)  # fmt: skip

    def lower(self, B: int):
        """(capi.Problem, keepalive) for a batch of B; rebuilt when the structure changed."""
        import torch

        from . import capi

        pp, po, pt, pw, ploss = self._merged(PositionErrorFunction, B, 3)
        op, oo, ot, ow, oloss = self._merged(OrientationErrorFunction, B, 4)
        blocks = [e.block(B) for e in self._efs if isinstance(e, _BlockErrorFunction) and e._constraints]
        key = (B, pp.tobytes(), op.tobytes(), tuple((b.type, b.parent.tobytes()) for b in blocks))
        if self._cache is None or self._cache[0] != key:
            if self._cache is not None:
                self._cache[1].close()
            self._cache = (key, capi.Problem(self.character.handle(), B, pp, op))
        pb = self._cache[1]
        limits, ells, wl = [], [], 1.0
        mt = mw = None
        wm = 1.0
        for e in self._efs:
            if isinstance(e, LimitErrorFunction):
                if limits or ells:
                    raise RuntimeError("only one LimitErrorFunction per solver function")
                limits = [l for l in e.limits if isinstance(l, ParameterLimit)]
                ells = [l for l in e.limits if isinstance(l, EllipsoidLimit)]
                wl = e.weight
            elif isinstance(e, ModelParametersErrorFunction):
                if mt is not None:
                    raise RuntimeError("only one ModelParametersErrorFunction per solver function")
                P = self.get_num_parameters()
                mt = np.ascontiguousarray(np.broadcast_to(e.target_parameters, (B, P)), np.float32)
                mw = np.ascontiguousarray(np.broadcast_to(e.target_weights, (B, P)), np.float32)
                wm = e.weight
            elif not isinstance(e, _JointErrorFunction):
                raise RuntimeError(f"{type(e).__name__} is not available on the GPU path")
        pb.set_constraints(po, pt, pw, oo, ot, ow, 1.0, 1.0, limits=limits, limit_function_weight=wl, model_target=mt,
                           model_weights=mw, model_function_weight=wm, pos_loss=ploss, ori_loss=oloss, joint_blocks=blocks,
                           ellipsoid_limits=ells)  # fmt: skip
        return pb, torch

    def _params(self, model_parameters):
        mp = np.ascontiguousarray(model_parameters, dtype=np.float32)
        P = self.get_num_parameters()
        if mp.shape[-1] != P:
            raise RuntimeError(f"Expected parameters to be of size {P}")
        return mp.ndim == 1, mp.reshape(-1, P)

    def get_error(self, model_parameters):
        single, mp = self._params(model_parameters)
        pb, torch = self.lower(mp.shape[0])
        _, _, err = pb.eval_jacobian(torch.from_numpy(mp).to(pb.device))
        e = err.cpu().numpy()
        return float(e[0]) if single else e

    def get_jacobian(self, model_parameters):
        """(residual [.., M], jacobian [.., M, P]) like SolverFunctionT::getJacobian.

        Row order: all position constraints (3 rows each, in the order their error functions were added), all
        orientation constraints (9 rows each), the further joint error functions block by block, ellipsoid limits,
        parameter limits, model-parameter rows -- grouped by kind, NOT interleaved in add_error_function order like the
        reference's per-function offsets (skeleton_solver_function.cpp:97-133).  J^T J, J^T r and the solve do not
        depend on the row order."""
        single, mp = self._params(model_parameters)
        pb, torch = self.lower(mp.shape[0])
        jac, res, _ = pb.eval_jacobian(torch.from_numpy(mp).to(pb.device))
        J = jac.cpu().numpy().transpose(0, 2, 1)
        r = res.cpu().numpy()
        return (r[0], J[0]) if single else (r, J)

    def get_gradient(self, model_parameters):
        """2 J^T r (error_function_helpers.cpp:220)."""
        r, J = self.get_jacobian(model_parameters)
        return 2.0 * np.einsum("...mp,...m->...p", J, r)


class SolverOptions:
    """momentum::SolverOptions (solver/solver.h:19-34)."""

    def __init__(self):
        self.min_iterations = 1
        self.max_iterations = 2
        self.threshold = 1.0
        self.verbose = False


class GaussNewtonSolverBaseOptions(SolverOptions):
    """gauss_newton_solver.h:17-33"""

    def __init__(self):
        super().__init__()
        self.regularization = 0.05
        self.do_line_search = False


class GaussNewtonSolverOptions(GaussNewtonSolverBaseOptions):
    """gauss_newton_solver.h:36-59 (use_block_jtj / direct_sparse_jtj select the reference's CPU
    assembly strategy; the GPU path has one)."""

    def __init__(self):
        super().__init__()
        self.use_block_jtj = False
        self.direct_sparse_jtj = False
        self.sparse_matrix_threshold = 200


class GaussNewtonSolverQROptions(GaussNewtonSolverBaseOptions):
    """character_solver/gauss_newton_solver_qr.h:20-25"""


class SubsetGaussNewtonSolverOptions(GaussNewtonSolverBaseOptions):
    """solver/subset_gauss_newton_solver.h:19-26"""


class Solver:
    #: MMX_LINE_SEARCH_* rule used when options.do_line_search is set
    _line_search_rule = 1  # GaussNewtonSolverT::updateParameters (gauss_newton_solver.cpp:283-313)

    def __init__(self, solver_function: SkeletonSolverFunction, options: Optional[SolverOptions] = None):
        self.solver_function = solver_function
        self.options = options if options is not None else GaussNewtonSolverOptions()
        self._enabled = None
        self._history = None
        self._store_history = False
        self._iteration_history = {}

    def set_store_history(self, b: bool) -> None:
        """SolverT::setStoreHistory (solver.h / solver.cpp:53-72)"""
        self._store_history = bool(b)

    def get_history(self):
        """SolverT::getHistory: {"parameters" [B, K, P], "error" [B, K], "iterations" [B], and for the Gauss-Newton solver
        "jtj" [B, K, n, n]} of the last solve (gauss_newton_solver.cpp:262-279: the damped lower triangle of every
        iteration's normal equations); empty unless set_store_history(True) was called before it.  Numpy arrays, without
        the batch axis for a single parameter vector."""
        return self._iteration_history

    def set_enabled_parameters(self, active_parameters) -> None:
        a = np.asarray(active_parameters).astype(bool).reshape(-1)
        if a.shape[0] != self.solver_function.get_num_parameters():
            raise RuntimeError("active_parameters has the wrong size")
        self._enabled = a.astype(np.uint8)

    @property
    def per_iteration_errors(self):
        """SolverT::getErrorHistory: list (one instance) or list of lists (batch)."""
        return self._history

    def solve(self, model_parameters):
        fn = self.solver_function
        single, mp = fn._params(model_parameters)
        pb, torch = fn.lower(mp.shape[0])
        pb.set_enabled(self._enabled if self._enabled is not None else np.ones(fn.get_num_parameters(), np.uint8))
        o = self.options
        opt = GnOptions.make(min_iterations=o.min_iterations, max_iterations=o.max_iterations, threshold=o.threshold,
                             regularization=getattr(o, "regularization", 0.05), do_line_search=self._line_search_rule if getattr(o, "do_line_search", False) else 0)  # fmt: skip
        t0 = torch.from_numpy(mp.copy()).to(pb.device)
        out = pb.solve(t0.clone(), opt, want_history=True, want_parameter_history=self._store_history)
        it = out["iterations"].cpu().numpy()
        h = out["error_history"].cpu().numpy()
        self._iteration_history = {}
        if self._store_history:
            jtj = pb.jtj_history(t0, out["parameter_history"], out["iterations"], opt.regularization)
            full = {"parameters": out["parameter_history"].cpu().numpy(), "error": h, "iterations": it, "jtj": jtj.cpu().numpy()}
            self._iteration_history = {k: (v[0] if single else v) for k, v in full.items()}
        hist = [[float(x) for x in h[b, : it[b]]] for b in range(mp.shape[0])]
        self._history = hist[0] if single else hist
        th = out["theta"].cpu().numpy()
        return th[0] if single else th


class GaussNewtonSolver(Solver):
    """GaussNewtonSolverT<float> (solver/gauss_newton_solver.h:67-137) for every element of the batch."""


class GaussNewtonSolverQR(Solver):
    """Solves the same regularised least-squares problem as GaussNewtonSolverQRT
    (character_solver/gauss_newton_solver_qr.cpp:50-150: QR of [J; sqrt(lambda) I]) -- on the GPU by
    the refined Cholesky step -- with that solver's line search (:126-149, against the directional
    derivative J^T r . delta, c_1 = 1e-4)."""

    _line_search_rule = 2


class SubsetGaussNewtonSolver(Solver):
    """SubsetGaussNewtonSolverT (solver/subset_gauss_newton_solver.cpp:72-145) on the subset given to
    `set_enabled_parameters`: same normal equations as GaussNewtonSolverT, line search against the
    directional derivative (:117-142)."""

    _line_search_rule = 2
