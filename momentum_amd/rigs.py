"""Host-side rig data model + synthetic rig generators.

`Rig` carries exactly what the hot path reads from a momentum::Character: the Skeleton
(parents, pre-rotations, translation offsets; momentum/character/joint.h:18-36,
skeleton.h:22-25) and the ParameterTransform as CSR (momentum/character/
parameter_transform.h:62-95).  Generators:

  make_test_character(n)   the reference's own fixture createTestCharacter(n)
                           (momentum/test/character/character_helpers.cpp:38-55,106-149)
  make_humanoid72(...)     BASELINE.json configs 2-4 (72-joint humanoid, P=128 or P=219)
  make_rig300(...)         BASELINE.json config 5 (300-joint hand+body rig, P=300)

No compute lives here (integer/array bookkeeping only).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

import numpy as np

from ._abi import RigDesc, as_ptr

PARAMS_PER_JOINT = 7  # momentum/character/types.h:21
TX, TY, TZ, RX, RY, RZ, SC = range(7)  # momentum/character/types.h:21-27


@dataclass
class Rig:
    """Skeleton + ParameterTransform (the two Character members the solver reads,
    momentum/character_solver/skeleton_solver_function.cpp:30-33)."""

    parent: np.ndarray  # [J] int32, -1 = kInvalidIndex
    pre_rotation: np.ndarray  # [J,4] float32 (x,y,z,w)
    translation_offset: np.ndarray  # [J,3] float32
    pt_outer: np.ndarray  # [7J+1] int32
    pt_inner: np.ndarray  # [nnz] int32
    pt_value: np.ndarray  # [nnz] float32
    pt_offsets: np.ndarray  # [7J] float32
    num_params: int
    joint_names: List[str] = field(default_factory=list)
    param_names: List[str] = field(default_factory=list)

    @property
    def num_joints(self) -> int:
        return int(self.parent.shape[0])

    def desc(self) -> RigDesc:
        """mmx_rig_desc pointing into this object's arrays (keep `self` alive)."""
        return RigDesc(
            self.num_joints,
            self.num_params,
            as_ptr(self.parent, C.c_int32),
            as_ptr(self.pre_rotation, C.c_float),
            as_ptr(self.translation_offset, C.c_float),
            as_ptr(self.pt_outer, C.c_int32),
            as_ptr(self.pt_inner, C.c_int32),
            as_ptr(self.pt_value, C.c_float),
            as_ptr(self.pt_offsets, C.c_float),
        )

    def depth(self) -> np.ndarray:
        d = np.zeros(self.num_joints, dtype=np.int32)
        for j in range(self.num_joints):
            p = int(self.parent[j])
            d[j] = 0 if p < 0 else d[p] + 1
        return d

    def dense_transform(self) -> np.ndarray:
        """7J x P dense copy of the parameter transform (tests only)."""
        A = np.zeros((PARAMS_PER_JOINT * self.num_joints, self.num_params), dtype=np.float32)
        for r in range(A.shape[0]):
            for k in range(self.pt_outer[r], self.pt_outer[r + 1]):
                A[r, self.pt_inner[k]] += self.pt_value[k]
        return A


def _build_rig(
    parent: Sequence[int],
    pre_rotation: np.ndarray,
    translation_offset: np.ndarray,
    triplets: Sequence[Tuple[int, int, float]],
    num_params: int,
    joint_names: List[str],
    param_names: List[str],
) -> Rig:
    """Triplets -> CSR with ascending columns per row (what Eigen's setFromTriplets yields,
    character_helpers.cpp:146)."""
    J = len(parent)
    rows = PARAMS_PER_JOINT * J
    parent = np.asarray(parent, dtype=np.int32)
    for j in range(J):  # Skeleton invariant parent < child (skeleton.cpp:16-22)
        assert parent[j] < j, "joints must be listed parent-before-child"
    acc: Dict[Tuple[int, int], float] = {}
    for r, c, v in triplets:
        assert 0 <= r < rows and 0 <= c < num_params
        acc[(r, c)] = acc.get((r, c), 0.0) + float(v)
    keys = sorted(acc.keys())
    outer = np.zeros(rows + 1, dtype=np.int32)
    for r, _ in keys:
        outer[r + 1] += 1
    outer = np.cumsum(outer).astype(np.int32)
    inner = np.array([c for _, c in keys], dtype=np.int32).reshape(-1)
    value = np.array([acc[k] for k in keys], dtype=np.float32).reshape(-1)
    return Rig(
        parent=parent,
        pre_rotation=np.ascontiguousarray(pre_rotation, dtype=np.float32),
        translation_offset=np.ascontiguousarray(translation_offset, dtype=np.float32),
        pt_outer=outer,
        pt_inner=inner,
        pt_value=value,
        pt_offsets=np.zeros(rows, dtype=np.float32),
        num_params=int(num_params),
        joint_names=list(joint_names),
        param_names=list(param_names),
    )


def make_test_character(num_joints: int = 3) -> Rig:
    """createTestCharacter(numJoints): straight chain, joint i offset UnitY, identity
    pre-rotations; parameters root_tx..root_rz, scale_global, joint1_rx, shared_rz (0.5 on
    joint1.rz and joint2.rz), jointK_rx for K >= 2
    (momentum/test/character/character_helpers.cpp:38-55,106-149)."""
    n = int(num_joints)
    assert n >= 3
    parent = [-1] + list(range(0, n - 1))
    pre = np.zeros((n, 4), dtype=np.float32)
    pre[:, 3] = 1.0
    off = np.zeros((n, 3), dtype=np.float32)
    off[1:, 1] = 1.0
    names = ["root_tx", "root_ty", "root_tz", "root_rx", "root_ry", "root_rz", "scale_global", "joint1_rx", "shared_rz"]
    rx_start = len(names)
    names += [f"joint{j}_rx" for j in range(2, n)]
    trip: List[Tuple[int, int, float]] = []
    for d in range(7):
        trip.append((0 * 7 + d, d, 1.0))
    trip.append((1 * 7 + RX, 7, 1.0))
    trip.append((1 * 7 + RZ, 8, 0.5))
    trip.append((2 * 7 + RZ, 8, 0.5))
    for j in range(2, n):
        trip.append((j * 7 + RX, rx_start + j - 2, 1.0))
    jn = ["root"] + [f"joint{i}" for i in range(1, n)]
    return _build_rig(parent, pre, off, trip, len(names), jn, names)


def _rand_quat_small(rng: np.random.Generator, max_angle: float) -> np.ndarray:
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    ang = rng.uniform(-max_angle, max_angle)
    return np.array([*(np.sin(0.5 * ang) * axis), np.cos(0.5 * ang)], dtype=np.float32)


# joint table of the 72-joint humanoid: (name, parent name, limb axis used for the offset)
def _humanoid72_topology() -> List[Tuple[str, str, Tuple[float, float, float]]]:
    up, down, fwd = (0.0, 1.0, 0.0), (0.0, -1.0, 0.0), (0.0, 0.0, 1.0)
    t: List[Tuple[str, str, Tuple[float, float, float]]] = [("pelvis", "", (0.0, 0.0, 0.0))]
    prev = "pelvis"
    for n in ("spine1", "spine2", "spine3", "spine4", "neck", "head"):
        t.append((n, prev, up))
        prev = n
    t += [("jaw", "head", fwd), ("eye_l", "head", fwd), ("eye_r", "head", fwd)]
    for s, sx in (("l", 1.0), ("r", -1.0)):
        side = (sx, 0.0, 0.0)
        t.append((f"hip_{s}", "pelvis", side))
        t.append((f"knee_{s}", f"hip_{s}", down))
        t.append((f"ankle_{s}", f"knee_{s}", down))
        t.append((f"ball_{s}", f"ankle_{s}", fwd))
        t.append((f"toe_{s}", f"ball_{s}", fwd))
        t.append((f"toe_end_{s}", f"toe_{s}", fwd))
    for s, sx in (("l", 1.0), ("r", -1.0)):
        side = (sx, 0.0, 0.0)
        t.append((f"clavicle_{s}", "spine4", side))
        t.append((f"shoulder_{s}", f"clavicle_{s}", side))
        t.append((f"elbow_{s}", f"shoulder_{s}", side))
        t.append((f"forearm_twist_{s}", f"elbow_{s}", side))
        t.append((f"wrist_{s}", f"elbow_{s}", side))
        for f in ("thumb", "index", "middle", "ring", "pinky"):
            prevf = f"wrist_{s}"
            for k in range(4):
                nm = f"{f}{k}_{s}"
                t.append((nm, prevf, side))
                prevf = nm
    assert len(t) == 72
    return t


def make_humanoid72(seed: int = 12345, variant: str = "p128", unit: float = 1.0) -> Rig:
    """72-joint humanoid of BASELINE.json configs 2-4: pelvis root;
    spine 4 + neck + head + jaw/eyes 3; 2 x leg 6; 2 x (arm 4 + forearm twist 1 + 5 fingers x 4);
    depth 12.  Offsets U[2,30] (cm, times `unit`) along limb axes, pre-rotations random <= 0.3 rad.

    variant "p128": root 6 DOF + scale_global + rotations on articulated joints, a jaw
                    translation, finger curls shared across joints and three shared parameters
                    (fist_l, fist_r, spine_twist) like the fixture's shared_rz  -> P = 128
    variant "p219": 6 root DOF + 3 rotations on all 71 non-root joints      -> P = 219
    """
    rng = np.random.default_rng(seed)
    topo = _humanoid72_topology()
    names = [n for n, _, _ in topo]
    idx = {n: i for i, n in enumerate(names)}
    J = len(topo)
    parent = [-1 if p == "" else idx[p] for _, p, _ in topo]
    pre = np.zeros((J, 4), dtype=np.float32)
    off = np.zeros((J, 3), dtype=np.float32)
    for j, (_, p, axis) in enumerate(topo):
        pre[j] = _rand_quat_small(rng, 0.3) if j > 0 else np.array([0, 0, 0, 1], dtype=np.float32)
        if j > 0:
            length = rng.uniform(2.0, 30.0) * unit
            if topo[j][0].startswith(("thumb", "index", "middle", "ring", "pinky")):
                length = rng.uniform(2.0, 5.0) * unit  # finger segments
            a = np.asarray(axis, dtype=np.float64) + 0.15 * rng.normal(size=3)
            off[j] = (length * a / np.linalg.norm(a)).astype(np.float32)

    pnames: List[str] = []
    trip: List[Tuple[int, int, float]] = []

    def add_param(name: str, entries: Sequence[Tuple[str, int, float]]) -> None:
        col = len(pnames)
        pnames.append(name)
        for jn, dof, w in entries:
            trip.append((idx[jn] * 7 + dof, col, w))

    if variant == "p219":
        for d, nm in enumerate(("tx", "ty", "tz", "rx", "ry", "rz")):
            add_param(f"root_{nm}", [("pelvis", d, 1.0)])
        for n in names[1:]:
            for d, nm in ((RX, "rx"), (RY, "ry"), (RZ, "rz")):
                add_param(f"{n}_{nm}", [(n, d, 1.0)])
        assert len(pnames) == 219
    elif variant == "p128":
        for d, nm in enumerate(("tx", "ty", "tz", "rx", "ry", "rz")):
            add_param(f"root_{nm}", [("pelvis", d, 1.0)])
        add_param("scale_global", [("pelvis", SC, 1.0)])
        def rot3(n, dofs=((RX, "rx"), (RY, "ry"), (RZ, "rz"))):
            for d, nm in dofs:
                add_param(f"{n}_{nm}", [(n, d, 1.0)])

        # spine: rx/rz everywhere, ry individually on spine1/spine4 only; spine2/spine3 twist is
        # driven ONLY by the shared spine_twist parameter (no exactly-redundant parameters: a
        # redundant direction is a null space of J that fp32 noise / lambda drifts along)
        rot3("spine1")
        rot3("spine2", ((RX, "rx"), (RZ, "rz")))
        rot3("spine3", ((RX, "rx"), (RZ, "rz")))
        rot3("spine4")
        add_param("spine_twist", [("spine2", RY, 0.5), ("spine3", RY, 0.5)])
        for n in ("neck", "head", "jaw", "eye_l", "eye_r"):
            rot3(n)
        add_param("jaw_tz", [("jaw", TZ, 1.0)])  # a translation DOF below a non-root parent
        for s in ("l", "r"):
            for n in (f"hip_{s}", f"knee_{s}", f"ankle_{s}", f"ball_{s}", f"toe_{s}"):
                rot3(n)
        for s in ("l", "r"):
            for n in (f"clavicle_{s}", f"shoulder_{s}", f"elbow_{s}", f"wrist_{s}"):
                rot3(n)
            rot3(f"forearm_twist_{s}", ((RX, "rx"), (RY, "ry")))
        for s in ("l", "r"):
            for f in ("thumb", "index", "middle", "ring", "pinky"):
                dofs = ((RX, "rx"), (RY, "ry"), (RZ, "rz")) if f in ("thumb", "index") else ((RX, "rx"), (RZ, "rz"))
                rot3(f"{f}0_{s}", dofs)
                add_param(f"{f}_{s}_curl", [(f"{f}1_{s}", RZ, 0.6), (f"{f}2_{s}", RZ, 0.4)])
        for s in ("l", "r"):
            ent: List[Tuple[str, int, float]] = []
            for f in ("thumb", "index", "middle", "ring", "pinky"):
                ent += [(f"{f}0_{s}", RZ, 0.3), (f"{f}1_{s}", RZ, 0.2), (f"{f}2_{s}", RZ, 0.2)]
            add_param(f"fist_{s}", ent)
        assert len(pnames) == 128, len(pnames)
    else:
        raise ValueError(f"unknown variant {variant!r}")
    return _build_rig(parent, pre, off, trip, len(pnames), names, pnames)


# the 16 end-effector / landmark joints of BASELINE.json config 2
HUMANOID72_LANDMARKS = [
    "spine4", "head", "knee_l", "knee_r", "ankle_l", "ankle_r", "toe_l", "toe_r",
    "elbow_l", "elbow_r", "wrist_l", "wrist_r", "index3_l", "index3_r", "thumb3_l", "thumb3_r",
]  # fmt: skip


def humanoid72_landmark_joints(rig: Rig) -> np.ndarray:
    idx = {n: i for i, n in enumerate(rig.joint_names)}
    return np.array([idx[n] for n in HUMANOID72_LANDMARKS], dtype=np.int32)


def make_rig300(seed: int = 12345, unit: float = 1.0) -> Rig:
    """300-joint hand+body rig of BASELINE.json config 5: the 72-joint body plus 228 extra
    joints (dense hand / twist / leaf chains) attached by a seeded procedure, depth <= 16.
    Parameters: the 128 of the p128 body + one rotation on 172 of the extra joints -> P = 300."""
    base = make_humanoid72(seed, "p128", unit)
    rng = np.random.default_rng(seed + 1)
    J0 = base.num_joints
    parent = list(map(int, base.parent))
    names = list(base.joint_names)
    depth = list(map(int, base.depth()))
    pre = [base.pre_rotation[j] for j in range(J0)]
    off = [base.translation_offset[j] for j in range(J0)]
    while len(parent) < 300:
        # grow short chains off joints that still have depth head-room
        cand = [j for j in range(len(parent)) if depth[j] < 15]
        p = int(rng.choice(cand))
        chain = int(rng.integers(1, 4))
        for _ in range(chain):
            if len(parent) >= 300 or depth[p] >= 16:
                break
            j = len(parent)
            parent.append(p)
            depth.append(depth[p] + 1)
            names.append(f"extra{j}")
            pre.append(_rand_quat_small(rng, 0.3))
            a = rng.normal(size=3)
            off.append((rng.uniform(1.0, 8.0) * unit * a / np.linalg.norm(a)).astype(np.float32))
            p = j
    J = len(parent)
    assert J == 300 and max(depth) <= 16
    trip: List[Tuple[int, int, float]] = []
    for r in range(7 * J0):
        for k in range(base.pt_outer[r], base.pt_outer[r + 1]):
            trip.append((r, int(base.pt_inner[k]), float(base.pt_value[k])))
    pnames = list(base.param_names)
    extra = list(range(J0, J))
    chosen = sorted(rng.choice(extra, size=172, replace=False).tolist())
    for j in chosen:
        d = int(rng.integers(RX, RZ + 1))
        trip.append((j * 7 + d, len(pnames), 1.0))
        pnames.append(f"{names[j]}_r{'xyz'[d - RX]}")
    assert len(pnames) == 300
    return _build_rig(parent, np.stack(pre), np.stack(off), trip, len(pnames), names, pnames)
