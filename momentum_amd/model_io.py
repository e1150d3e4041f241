"""Rig ingestion: momentum's text formats for the two Character members the
solver reads, into the C-ABI rig descriptor (`Rig`) and `mmx_parameter_limit` entries.

* `.model` text ("Momentum Model Definition V1.0"): sections [ParameterTransform], [Limits]
  ([ParameterSets] / [PoseConstraints] are read as raw text) -- momentum/io/skeleton/
  parameter_transform_io.cpp:40-125 (sections; duplicate sections are concatenated),
  :157-240,311-372 (channel expressions `joint.attr = w*param + w*joint.attr + const`),
  parameter_limits_io.cpp:296-343 (minmax, minmax_passive), :345-445 (piecewise `linear`),
  :447-545 (`linear` on joint parameters), :558-578 (`halfplane`, normalised), :622-690 (dispatch).
* legacy JSON skeleton ("Skeleton" / "BodySkeleton" / "skeleton" -> "Bones": Name, Parent,
  PreRotation (x,y,z,w), TranslationOffset) -- momentum/io/legacy_json/legacy_json_io.cpp:82-86,
  121-156,591.
* glTF / GLB: joints from the node hierarchy, parameter transform and limits from the FB_momentum
  extension (see `load_gltf` at the end of this file).

Host-side text handling only.  `ellipsoid` lines become `EllipsoidLimit` (mmx_ellipsoid_limit);
`minmax_passive` entries are kept as dicts so that a caller sees them -- LimitErrorFunction itself
ignores that type (limit_error_function.cpp:1136-1150) and `limits_for_solver` drops it.
"""
from __future__ import annotations

import json
import math
import re
from typing import Dict, List, Sequence, Tuple

import numpy as np

from ._abi import (
    MMX_LIMIT_HALFPLANE, MMX_LIMIT_LINEAR, MMX_LIMIT_LINEAR_JOINT, MMX_LIMIT_MINMAX, MMX_LIMIT_MINMAX_JOINT, EllipsoidLimit, ParameterLimit,
)  # fmt: skip
from .rigs import Rig

JOINT_PARAMETER_NAMES = ("tx", "ty", "tz", "rx", "ry", "rz", "sc")  # character/types.h:24
KNOWN_SECTIONS = ("ParameterTransform", "ParameterSets", "PoseConstraints", "Limits")
FLT_MAX = float(np.finfo(np.float32).max)
HEADER = "Momentum Model Definition V1.0"


class ModelFormatError(RuntimeError):
    """MT_THROW of the reference's parsers."""


def _strip(line: str) -> str:
    return line.split("#", 1)[0].strip()


def load_momentum_model(text: str) -> Dict[str, str]:
    """loadMomentumModelCommon: {section name: content}; a repeated section is concatenated."""
    lines = text.splitlines()
    i = 0
    while i < len(lines):
        s = _strip(lines[i])
        i += 1
        if not s:
            continue
        if s == HEADER:
            break
        raise ModelFormatError(f"Invalid model definition file; expected '{HEADER}', got {s}")
    out: Dict[str, str] = {}
    name = ""
    for raw in lines[i:]:
        s = _strip(raw)
        if not s:
            continue
        m = re.fullmatch(r"\[(\w+)\]", s)
        if m:
            name = m.group(1)
        elif name in KNOWN_SECTIONS:
            out[name] = out.get(name, "") + s + "\n"
    return out


# ---------------------------------------------------------------------------------------------
# [ParameterTransform]
# ---------------------------------------------------------------------------------------------
def parse_parameter_transform(text: str, joint_names: Sequence[str]):
    """parseParameterTransform: (parameter names, triplets [(row, col, value)], offsets [7J])."""
    J = len(joint_names)
    jid = {n: k for k, n in enumerate(joint_names)}
    names: List[str] = []
    triplets: List[Tuple[int, int, float]] = []
    offsets = np.zeros(7 * J, np.float32)
    for ln, raw in enumerate(text.splitlines(), 1):
        line = raw.split("#", 1)[0]
        if not line.strip():
            continue
        if line.startswith("limit") or line.startswith("parameterset") or line.startswith("poseconstraints"):
            continue  # old single-section files mix these lines in; their own parsers read them
        sides = line.split("=")
        if len(sides) != 2:
            continue  # "Ignoring invalid line" in the reference
        lhs = sides[0].strip().split(".")
        if len(lhs) != 2:
            raise ModelFormatError(f"Unknown joint name in expression at line {ln}: {line}")
        joint, attr = lhs[0].strip(), lhs[1].strip()
        if joint not in jid:
            raise ModelFormatError(f"Unknown joint name in expression at line {ln}: {line}")
        if attr not in JOINT_PARAMETER_NAMES:
            raise ModelFormatError(f"Unknown channel name in expression at line {ln}: {line}")
        row = 7 * jid[joint] + JOINT_PARAMETER_NAMES.index(attr)
        for term in sides[1].split("+"):
            parts = term.split("*")
            if len(parts) == 1:
                if parts[0].strip():
                    offsets[row] = np.float32(float(parts[0]))  # additional constant
                continue
            if len(parts) != 2:
                continue
            weight = float(np.float32(float(parts[0])))
            pname = parts[1].strip()
            ref_joint = pname.split(".", 1)[0]
            ref_attr = pname.split(".", 1)[1] if "." in pname else ""
            if pname in names:
                triplets.append((row, names.index(pname), weight))
            elif ref_joint in jid and ref_attr in JOINT_PARAMETER_NAMES:
                # reference to an earlier joint channel: copy its parameters, scaled (:213-226)
                ref = 7 * jid[ref_joint] + JOINT_PARAMETER_NAMES.index(ref_attr)
                for r, c, v in list(triplets):
                    if r == ref:
                        triplets.append((row, c, float(np.float32(v * weight))))
            elif ref_joint in jid:
                raise ModelFormatError(f"Could not parse channel expression : {line}")
            else:
                names.append(pname)
                triplets.append((row, len(names) - 1, weight))
    triplets = [t for t in triplets if t[2] != 0.0]
    return names, triplets, offsets


def _csr(triplets, rows: int):
    """Eigen setFromTriplets into a row-major matrix: duplicates add up, columns ascending per row."""
    acc: Dict[Tuple[int, int], float] = {}
    for r, c, v in triplets:
        acc[(r, c)] = float(np.float32(acc.get((r, c), 0.0) + v))
    outer = np.zeros(rows + 1, np.int32)
    inner, value = [], []
    for r, c in sorted(acc):
        outer[r + 1] += 1
        inner.append(c)
        value.append(acc[(r, c)])
    return np.cumsum(outer).astype(np.int32), np.array(inner, np.int32), np.array(value, np.float32)


# ---------------------------------------------------------------------------------------------
# [Limits]
# ---------------------------------------------------------------------------------------------
_TOKEN = re.compile(r"\s*(?:(\[)|(\])|(,)|([A-Za-z_][A-Za-z0-9_.]*)|([-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?))")


class _Tokens:
    def __init__(self, line: str, ln: int):
        self.line, self.ln, self.toks = line, ln, []
        pos = 0
        while pos < len(line):
            if not line[pos:].strip():
                break
            m = _TOKEN.match(line, pos)
            if not m:
                raise ModelFormatError(f"Unexpected token at character {pos} of line {ln}: {line}")
            kind = "[" if m.group(1) else "]" if m.group(2) else "," if m.group(3) else "id" if m.group(4) else "num"
            self.toks.append((kind, m.group(m.lastindex)))
            pos = m.end()
        self.i = 0

    def eof(self) -> bool:
        return self.i >= len(self.toks)

    def peek(self) -> str:
        return "eof" if self.eof() else self.toks[self.i][0]

    def take(self, kind: str) -> str:
        if self.peek() != kind:
            raise ModelFormatError(f"Expected {kind} in line {self.ln}: {self.line}")
        self.i += 1
        return self.toks[self.i - 1][1]

    def number(self) -> float:
        return float(np.float32(float(self.take("num"))))

    def vec(self) -> List[float]:
        self.take("[")
        out = [self.number()]
        while self.peek() == ",":
            self.take(",")
            out.append(self.number())
        self.take("]")
        return out


def parse_parameter_limits(text: str, joint_names: Sequence[str], param_names: Sequence[str]) -> list:
    """parseParameterLimits: list of ParameterLimit (+ dicts for the types the GPU path does not take)."""
    jid = {n: k for k, n in enumerate(joint_names)}

    def model_index(name: str, t: _Tokens) -> int:
        if name not in param_names:
            raise ModelFormatError(f"Unknown parameter name {name} in line {t.ln}: {t.line}")
        return list(param_names).index(name)

    def joint_row(name: str, t: _Tokens) -> Tuple[int, int]:
        j, _, a = name.partition(".")
        if j not in jid or a not in JOINT_PARAMETER_NAMES:
            raise ModelFormatError(f"Unknown joint parameter {name} in line {t.ln}: {t.line}")
        return jid[j], JOINT_PARAMETER_NAMES.index(a)

    out: list = []
    for ln, raw in enumerate(text.splitlines(), 1):
        line = raw.split("#", 1)[0]
        if not line.strip() or not line.startswith("limit"):
            continue
        t = _Tokens(line, ln)
        t.take("id")  # "limit"
        pname, kind = t.take("id"), t.take("id")
        if kind in ("minmax", "minmax_passive"):
            lo, hi = t.vec()
            w = 1.0 if t.eof() else t.number()
            if kind == "minmax" and "." not in pname:
                out.append(ParameterLimit.minmax(model_index(pname, t), lo, hi, w))
            else:
                j, a = joint_row(pname, t)
                lim = ParameterLimit.minmax_joint(j, a, lo, hi, w)
                out.append(lim if kind == "minmax" else dict(type="minmax_passive", joint=j, joint_parameter=a, limits=(lo, hi), weight=w))
        elif kind == "linear":
            on_joint = "." in pname
            target = t.take("id")
            segs = []
            prev = -FLT_MAX
            while t.peek() == "[":
                seg = t.vec()
                if len(seg) not in (2, 3):
                    raise ModelFormatError(f"Expected 2 or 3 values for linear segment in line {ln}: {line}")
                if prev == FLT_MAX and len(seg) == 3:
                    raise ModelFormatError(f"Only the last linear segment can have unrestricted range in line {ln}: {line}")
                cur = seg[2] if len(seg) == 3 else FLT_MAX
                segs.append((seg[0], seg[1], prev, cur))
                prev = cur
            if not segs:
                raise ModelFormatError(f"Expected [ in line {ln}: {line}")
            for (s0, o0, _, hi0), (s1, o1, _, _) in zip(segs, segs[1:]):
                if abs((s0 * hi0 - o0) - (s1 * hi0 - o1)) > 1e-3:
                    raise ModelFormatError(f"Mismatch between function values between two linear segments in line {ln}: {line}")
            w = 1.0 if t.eof() else t.number()
            for sc, off, lo, hi in segs:
                rng = (lo, hi)  # a single unrestricted segment gets (-FLT_MAX, FLT_MAX) like the reference (:376,398)
                if on_joint:
                    (rj, ra), (tj, ta) = joint_row(pname, t), joint_row(target, t)
                    out.append(ParameterLimit.linear_joint(rj, ra, tj, ta, sc, off, rng[0], rng[1], w))
                else:
                    out.append(ParameterLimit.linear(model_index(pname, t), model_index(target, t), sc, off, rng[0], rng[1], w))
        elif kind == "halfplane":
            p2 = t.take("id")
            n = t.vec()
            off = t.number()
            ln2 = math.hypot(n[0], n[1])
            w = 1.0 if t.eof() else t.number()
            out.append(ParameterLimit.halfplane(model_index(pname, t), model_index(p2, t), n[0] / ln2, n[1] / ln2, off / ln2, w))
        elif kind in ("ellipsoid", "elipsoid"):  # "[offset] parent [translation] [rotation zyx, degrees] [scale] <weight>" (:580-610)
            if pname not in jid:
                raise ModelFormatError(f"Unknown joint name {pname} in line {ln}: {line}")
            offset = t.vec()
            ep = t.take("id")
            if ep not in jid:
                raise ModelFormatError(f"Unknown joint name {ep} in line {ln}: {line}")
            translation, euler_zyx, scale = t.vec(), t.vec(), t.vec()
            w = 1.0 if t.eof() else t.number()
            out.append(EllipsoidLimit.make(jid[pname], offset, jid[ep], translation, euler_zyx, scale, w))
        else:
            raise ModelFormatError(f"Unexpected parameter limit type {kind} in line {ln}: {line}")
        if not t.eof():
            raise ModelFormatError(f"Unexpected token in line {ln}: {line}")
    return out


def limits_for_solver(limits: list) -> List[ParameterLimit]:
    """the parameter-space entries LimitErrorFunction evaluates (mmx_parameter_limit)"""
    return [l for l in limits if isinstance(l, ParameterLimit)]


def ellipsoids_for_solver(limits: list) -> List[EllipsoidLimit]:
    """the LimitType::Ellipsoid entries (mmx_ellipsoid_limit)"""
    return [l for l in limits if isinstance(l, EllipsoidLimit)]


def _num(x: float) -> str:
    return repr(float(np.float32(x)))


def write_parameter_limits(limits: Sequence, joint_names: Sequence[str], param_names: Sequence[str]) -> str:
    """writeParameterLimits for the types above; consecutive Linear entries of one (reference,
    target) pair whose ranges chain are written as one piecewise line like the reference does."""
    jp = lambda row: f"{joint_names[row // 7]}.{JOINT_PARAMETER_NAMES[row % 7]}"
    out = []
    i = 0
    limits = list(limits)
    while i < len(limits):
        l = limits[i]
        if not isinstance(l, ParameterLimit):
            i += 1
            continue
        v = list(l.v)
        if l.type == MMX_LIMIT_MINMAX:
            out.append(f"limit {param_names[l.index0]} minmax [{_num(v[0])}, {_num(v[1])}] {_num(l.weight)}")
        elif l.type == MMX_LIMIT_MINMAX_JOINT:
            out.append(f"limit {jp(l.index0)} minmax [{_num(v[0])}, {_num(v[1])}] {_num(l.weight)}")
        elif l.type == MMX_LIMIT_HALFPLANE:
            out.append(f"limit {param_names[l.index0]} halfplane {param_names[l.index1]} [{_num(v[0])}, {_num(v[1])}] {_num(v[2])} {_num(l.weight)}")
        elif l.type in (MMX_LIMIT_LINEAR, MMX_LIMIT_LINEAR_JOINT):
            name = (lambda k: param_names[k]) if l.type == MMX_LIMIT_LINEAR else jp
            segs = [l]
            while (i + 1 < len(limits) and isinstance(limits[i + 1], ParameterLimit) and limits[i + 1].type == l.type
                   and (limits[i + 1].index0, limits[i + 1].index1) == (l.index0, l.index1)
                   and not (segs[-1].v[2] == 0 and segs[-1].v[3] == 0) and limits[i + 1].v[2] == segs[-1].v[3]):  # fmt: skip
                i += 1
                segs.append(limits[i])
            body = " ".join(
                f"[{_num(s.v[0])}, {_num(s.v[1])}]" if (s.v[2] == 0 and s.v[3] == 0) or s.v[3] >= FLT_MAX else f"[{_num(s.v[0])}, {_num(s.v[1])}, {_num(s.v[3])}]"
                for s in segs
            )
            out.append(f"limit {name(l.index0)} linear {name(l.index1)} {body} {_num(l.weight)}")
        i += 1
    return "\n".join(out) + ("\n" if out else "")


# ---------------------------------------------------------------------------------------------
# skeleton (legacy JSON) and the assembled rig
# ---------------------------------------------------------------------------------------------
def parse_legacy_skeleton(doc) -> Tuple[List[str], np.ndarray, np.ndarray, np.ndarray]:
    """legacySkeletonToMomentum: (names, parent [-1 = root], pre_rotation [J,4] xyzw, translation_offset [J,3])."""
    if isinstance(doc, str):
        doc = json.loads(doc)
    for key in ("Skeleton", "BodySkeleton", "skeleton"):
        if key in doc:
            doc = doc[key]
            break
    if "Bones" not in doc:
        raise ModelFormatError("Legacy skeleton JSON missing 'Bones' field")
    bones = doc["Bones"]
    J = len(bones)
    names = [b["Name"] for b in bones]
    parent = np.array([int(b["Parent"]) if 0 <= int(b["Parent"]) < J else -1 for b in bones], np.int32)  # kInvalidIndex = SIZE_MAX
    pre = np.array([b.get("PreRotation", [0.0, 0.0, 0.0, 1.0]) for b in bones], np.float32).reshape(J, 4)
    off = np.array([b.get("TranslationOffset", [0.0, 0.0, 0.0]) for b in bones], np.float32).reshape(J, 3)
    return names, parent, pre, off


def skeleton_to_legacy_json(rig: Rig) -> dict:
    """momentumSkeletonToLegacy (legacy_json_io.cpp:158-190)."""
    bones = []
    for j in range(rig.num_joints):
        bones.append({
            "Name": rig.joint_names[j] if rig.joint_names else f"joint{j}",
            "Parent": int(rig.parent[j]) if rig.parent[j] >= 0 else 2**64 - 1,
            "PreRotation": [float(x) for x in rig.pre_rotation[j]],
            "TranslationOffset": [float(x) for x in rig.translation_offset[j]],
            "JointType": "Root" if rig.parent[j] < 0 else "Limb",
            "RotationOrder": "XYZ",
        })  # fmt: skip
    return {"Skeleton": {"Bones": bones}}


def write_parameter_transform(rig: Rig) -> str:
    """one channel expression per driven joint parameter (writeParameterTransform's shape)"""
    lines = []
    for r in range(7 * rig.num_joints):
        terms = [f"{_num(rig.pt_value[k])}*{rig.param_names[rig.pt_inner[k]]}" for k in range(rig.pt_outer[r], rig.pt_outer[r + 1])]
        if rig.pt_offsets[r] != 0:
            terms.append(_num(rig.pt_offsets[r]))
        if terms:
            lines.append(f"{rig.joint_names[r // 7]}.{JOINT_PARAMETER_NAMES[r % 7]} = " + " + ".join(terms))
    return "\n".join(lines) + "\n"


def load_character(skeleton_json, model_text: str) -> Tuple[Rig, list]:
    """(Rig, parameter limits) from a legacy JSON skeleton and a `.model` definition: what
    loadCharacter + loadModelDefinition hand to the solver (character_io.cpp, parameter_transform_io.cpp:126-155)."""
    names, parent, pre, off = parse_legacy_skeleton(skeleton_json)
    for j, p in enumerate(parent):
        if p >= j:
            raise ModelFormatError("skeleton joints must be stored parent-before-child")  # skeleton.cpp:16-22
    sections = load_momentum_model(model_text)
    pnames, triplets, offsets = parse_parameter_transform(sections.get("ParameterTransform", ""), names)
    outer, inner, value = _csr(triplets, 7 * len(names))
    rig = Rig(parent, pre, off, outer, inner, value, offsets, len(pnames), list(names), list(pnames))
    limits = parse_parameter_limits(sections.get("Limits", ""), names, pnames)
    return rig, limits


# ---------------------------------------------------------------------------------------------
# glTF / GLB characters (skeleton from the node hierarchy, parameter transform and limits from the
# FB_momentum extension) -- momentum/io/gltf/gltf_skeleton_io.cpp:79-175 (loadHierarchyRecursive),
# :267-278 (createJoint, isHierarchyNode), :280-384 (gatherSkeletonRoots), :389-418 (loadHierarchy);
# gltf_io.cpp:57-80 (loadGlobalExtensions); io/common/json_utils.cpp:204-285 (parameterTransformFromJson),
# :519-674 (limits).  Lengths in the file are metres, momentum works in centimetres
# (gltf/utils/coordinate_utils.h:13-26); matrices are stored as lists of columns (json_utils.h:30-41).
# ---------------------------------------------------------------------------------------------
_M_TO_CM = 100.0


def _gltf_document(data) -> dict:
    if isinstance(data, dict):
        return data
    if isinstance(data, str):
        return json.loads(data)
    b = bytes(data)
    if b[:4] != b"glTF":
        return json.loads(b.decode("utf-8"))
    import struct

    version, _ = struct.unpack("<II", b[4:12])
    if version != 2:
        raise ModelFormatError(f"unsupported GLB version {version}")
    clen, ctype = struct.unpack("<I4s", b[12:20])
    if ctype != b"JSON":
        raise ModelFormatError("GLB: first chunk is not JSON")
    return json.loads(b[20 : 20 + clen].decode("utf-8"))


def _momentum_ext(obj: dict) -> dict:
    return obj.get("extensions", {}).get("FB_momentum", {}) if isinstance(obj, dict) else {}


def _gltf_skeleton_roots(doc: dict) -> List[List[int]]:
    nodes = doc.get("nodes", [])
    scene = doc["scenes"][doc.get("scene", 0)]
    if not doc.get("meshes") or not doc.get("skins"):
        return [list(scene.get("nodes", []))]
    parent = [-1] * len(nodes)
    stack = list(scene.get("nodes", []))
    while stack:
        nxt = []
        for n in stack:
            for c in nodes[n].get("children", []):
                parent[c] = n
                nxt.append(c)
        stack = nxt

    def ancestors(n):
        out = []
        while n != -1:
            out.append(n)
            n = parent[n]
        return out[::-1]

    def common(a, b2):
        ca = -1
        for x, y in zip(ancestors(a), ancestors(b2)):
            if x != y:
                break
            ca = x
        return ca

    per_skin = []
    for skin in doc["skins"]:
        roots: List[int] = []
        for j in skin["joints"]:
            if parent[j] not in skin["joints"]:
                for k, r in enumerate(roots):
                    ca = common(j, r)
                    if ca != -1:
                        roots[k] = ca
                        break
                else:
                    roots.append(j)
        per_skin.append(roots)
    result = []
    for i, roots in enumerate(per_skin):
        if len(roots) > 1:
            result.append(roots)
            continue
        anc = ancestors(roots[0])
        if not any(k != i and len(o) == 1 and o[0] in anc for k, o in enumerate(per_skin)):
            result.append([roots[0]])
    return result


def load_gltf(data) -> Tuple[Rig, list]:
    """(Rig, parameter limits) of the first skeleton in a glTF / GLB document (bytes, str or dict)."""
    doc = _gltf_document(data)
    nodes = doc.get("nodes", [])
    if not nodes:
        raise ModelFormatError("No valid node found in the gltf file.")
    use_ext = "FB_momentum" in doc.get("extensions", {})
    names: List[str] = []
    parents: List[int] = []
    pre: List[List[float]] = []
    off: List[List[float]] = []

    def walk(nid: int, pj: int) -> None:
        if nid < 0 or nid >= len(nodes):
            raise ModelFormatError(f"Invalid node id found in the gltf hierarchy: {nid}")
        node = nodes[nid]
        if "skin" in node or "camera" in node:  # isHierarchyNode
            return
        typ = _momentum_ext(node).get("type", "")
        npj = pj
        if typ in ("collision_capsule", "collision_ellipsoid", "collision_box", "locator"):
            pass  # end nodes the solver path does not read
        elif (not use_ext and "mesh" not in node) or typ == "skeleton_joint":
            names.append(node.get("name", ""))
            parents.append(pj)
            pre.append(list(node.get("rotation", [0.0, 0.0, 0.0, 1.0])))
            off.append([_M_TO_CM * float(x) for x in node.get("translation", [0.0, 0.0, 0.0])])
            npj = len(names) - 1
        for c in node.get("children", []):
            walk(c, npj)

    roots = _gltf_skeleton_roots(doc)
    if len(roots) != 1:
        raise ModelFormatError(f"expected one skeleton in the file, found {len(roots)}")
    for r in roots[0]:
        walk(r, -1)
        if names:
            break
    J = len(names)
    jid = {n: k for k, n in enumerate(names)}
    ext = _momentum_ext(doc)
    pnames: List[str] = []
    triplets: List[Tuple[int, int, float]] = []
    if "transform" in ext:
        tj = ext["transform"]
        if "parameters" not in tj or "joints" not in tj:
            raise ModelFormatError("No 'parameters' / 'joints' found in parameter transform.")
        pnames = list(tj["parameters"])
        for jn, attrs in tj["joints"].items():
            if jn not in jid:
                raise ModelFormatError(f"Unknown joint name in expression : {jn}")
            for an, terms in attrs.items():
                if an not in JOINT_PARAMETER_NAMES:
                    raise ModelFormatError(f"Unknown channel name in expression : {an}")
                for pn, w in terms.items():
                    if pn not in pnames:
                        raise ModelFormatError(f"Unknown parameter name in expression : {pn}")
                    triplets.append((7 * jid[jn] + JOINT_PARAMETER_NAMES.index(an), pnames.index(pn), float(np.float32(w))))
    outer, inner, value = _csr(triplets, 7 * J)
    rig = Rig(np.array(parents, np.int32), np.array(pre, np.float32).reshape(J, 4), np.array(off, np.float32).reshape(J, 3),
              outer, inner, value, np.zeros(7 * J, np.float32), len(pnames), names, pnames)  # fmt: skip
    limits: list = []
    pid = lambda n: pnames.index(n)
    for el in ext.get("parameterLimits", []):
        t, w = el.get("type", ""), float(el.get("weight", 0.0))
        if t == "minmax":
            lo, hi = np.ravel(np.asarray(el["limits"], np.float64))[:2]  # [lo, hi]; older files nest it: [[lo, hi]]
            limits.append(ParameterLimit.minmax(pid(el["parameter"]), float(lo), float(hi), w))
        elif t in ("minmax_joint", "minmax_joint_passive"):
            j, a = jid[el["jointIndex"]], JOINT_PARAMETER_NAMES.index(el["jointParameter"])
            lim = ParameterLimit.minmax_joint(j, a, el["limits"][0], el["limits"][1], w)
            limits.append(lim if t == "minmax_joint" else dict(type="minmax_passive", joint=j, joint_parameter=a, limits=tuple(el["limits"]), weight=w))
        elif t == "linear":
            limits.append(ParameterLimit.linear(pid(el["referenceParameter"]), pid(el["targetParameter"]), el["scale"], el["offset"],
                                                el.get("rangeMin", -FLT_MAX), el.get("rangeMax", FLT_MAX), w))  # fmt: skip
        elif t == "linear_joint":
            limits.append(ParameterLimit.linear_joint(jid[el["referenceJoint"]], int(el["referenceJointParameter"]), jid[el["targetJoint"]],
                                                      int(el["targetJointParameter"]), el["scale"], el["offset"],
                                                      el.get("rangeMin", -FLT_MAX), el.get("rangeMax", FLT_MAX), w))  # fmt: skip
        elif t == "half_plane":
            limits.append(ParameterLimit.halfplane(pid(el["param1"]), pid(el["param2"]), el["normal"][0], el["normal"][1], el["offset"], w))
        elif t in ("ellipsoid", "elipsoid"):
            A = np.array(el[t], np.float64).reshape(4, 4).T  # list of columns
            A[:3, 3] *= _M_TO_CM
            epk = "ellipsoidParent" if t == "ellipsoid" else "elipsoidParent"
            limits.append(EllipsoidLimit.from_affine(jid[el["parent"]], [_M_TO_CM * float(x) for x in el["offset"]], jid[el[epk]], A[:3], w))
        else:
            raise ModelFormatError(f"Unknown parameter limit type '{t}'")
    return rig, limits


def _glb_binary_chunk(data) -> bytes:
    """The BIN chunk of a GLB container (b"" when the document has none / is plain JSON)."""
    if isinstance(data, (dict, str)):
        return b""
    b = bytes(data)
    if b[:4] != b"glTF":
        return b""
    import struct

    if len(b) < 20:
        raise ModelFormatError("truncated GLB header")
    total, clen = struct.unpack("<I", b[8:12])[0], struct.unpack("<I", b[12:16])[0]
    if total > len(b) or 20 + clen > len(b):  # (an untrusted file: every declared length is held against the bytes at hand)
        raise ModelFormatError("GLB chunk lengths exceed the file")
    pos = 20 + clen
    while pos + 8 <= len(b):
        n, ctype = struct.unpack("<I4s", b[pos : pos + 8])
        if pos + 8 + n > len(b):
            raise ModelFormatError("GLB chunk lengths exceed the file")
        if ctype == b"BIN\0":
            return b[pos + 8 : pos + 8 + n]
        pos += 8 + n
    return b""


def _float_accessor(doc: dict, binary: bytes, index: int) -> np.ndarray:
    """Flat float32 contents of accessor `index` (copyAccessorBuffer<float>, momentum/io/gltf/utils/accessor_utils.h):
    component type FLOAT, tightly packed or strided, from the GLB's own buffer."""
    acc = doc["accessors"][index]
    if acc.get("componentType") != 5126:
        raise ModelFormatError("accessor is not of component type FLOAT")
    ncomp = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT2": 4, "MAT3": 9, "MAT4": 16}[acc.get("type", "SCALAR")]
    view = doc["bufferViews"][acc["bufferView"]]
    if view.get("buffer", 0) != 0 or "uri" in doc["buffers"][0]:
        raise ModelFormatError("only the GLB's embedded buffer is supported")
    start = int(view.get("byteOffset", 0)) + int(acc.get("byteOffset", 0))
    count, stride = int(acc["count"]), int(view.get("byteStride", 0)) or 4 * ncomp
    # the file is untrusted input: count / stride / offsets are held against the BIN chunk before any strided view is formed
    # (the reference's copyAccessorBuffer goes through fx::gltf's own size checks)
    if count < 0 or start < 0 or stride < 4 * ncomp:
        raise ModelFormatError("accessor with a negative count / offset or a stride below its element size")
    if count == 0:
        return np.zeros(0, np.float32)
    if start + (count - 1) * stride + 4 * ncomp > len(binary):
        raise ModelFormatError("accessor reaches past the end of the GLB's binary chunk")
    view_len = view.get("byteLength")
    if view_len is not None and int(acc.get("byteOffset", 0)) + (count - 1) * stride + 4 * ncomp > int(view_len):
        raise ModelFormatError("accessor reaches past the end of its buffer view")
    raw = np.frombuffer(binary, np.uint8)
    rows = np.lib.stride_tricks.as_strided(raw[start:], shape=(count, 4 * ncomp), strides=(stride, 1))
    return np.ascontiguousarray(rows).view("<f4").reshape(-1).astype(np.float32)


def load_gltf_motion(data):
    """The motion a momentum GLB stores in its FB_momentum extension (getMotionFromModel,
    momentum/io/gltf/gltf_animation_io.cpp:72-112): dict(parameter_names, poses [nframes, nparams] (the reference's
    column-major nparams x nframes matrix, one row per frame here), joint_names, identity [7 J] joint-parameter offsets,
    fps).  Empty dict when the file stores none."""
    doc = _gltf_document(data)
    ext = _momentum_ext(doc)
    motion = ext.get("motion", {})
    nframes = int(motion.get("nframes", 0))
    pose_acc, off_acc = int(motion.get("poses", -1)), int(motion.get("offsets", -1))
    if nframes == 0 or (pose_acc < 0 and off_acc < 0):
        return {}
    binary = _glb_binary_chunk(data)
    out = dict(parameter_names=[], poses=np.zeros((nframes, 0), np.float32), joint_names=[], identity=np.zeros(0, np.float32),
               fps=float(ext.get("fps", 0.0)))  # fmt: skip
    if pose_acc >= 0:
        names = list(motion.get("parameterNames", []))
        v = _float_accessor(doc, binary, pose_acc)
        if v.size != len(names) * nframes:
            return {}
        out["parameter_names"], out["poses"] = names, v.reshape(nframes, len(names))
    if off_acc >= 0:
        jn = list(motion.get("jointNames", []))
        v = _float_accessor(doc, binary, off_acc)
        if v.size != 7 * len(jn):
            return {}
        out["joint_names"], out["identity"] = jn, v
    return out


def write_gltf(rig: Rig, limits: Sequence = ()) -> dict:
    """glTF document (dict) with the FB_momentum extension: joints as `skeleton_joint` nodes in the
    hierarchy, `transform` (parameterTransformToJson, json_utils.cpp:168-200) and `parameterLimits`."""
    J = rig.num_joints
    nodes = []
    for j in range(J):
        nodes.append({"name": rig.joint_names[j], "rotation": [float(x) for x in rig.pre_rotation[j]],
                      "translation": [float(x) / _M_TO_CM for x in rig.translation_offset[j]],
                      "extensions": {"FB_momentum": {"type": "skeleton_joint"}}})  # fmt: skip
    for j in range(J):
        p = int(rig.parent[j])
        if p >= 0:
            nodes[p].setdefault("children", []).append(j)
    joints: Dict[str, dict] = {}
    for r in range(7 * J):
        for k in range(rig.pt_outer[r], rig.pt_outer[r + 1]):
            joints.setdefault(rig.joint_names[r // 7], {}).setdefault(JOINT_PARAMETER_NAMES[r % 7], {})[rig.param_names[rig.pt_inner[k]]] = float(rig.pt_value[k])
    lj = []
    for l in limits:
        if isinstance(l, EllipsoidLimit):
            A = np.eye(4)
            A[:3] = np.array(list(l.ellipsoid)).reshape(3, 4)
            A[:3, 3] /= _M_TO_CM
            lj.append({"type": "ellipsoid", "weight": l.weight, "parent": rig.joint_names[l.parent], "ellipsoidParent": rig.joint_names[l.ellipsoid_parent],
                       "offset": [float(x) / _M_TO_CM for x in l.offset], "ellipsoid": [[float(A[r2, c]) for r2 in range(4)] for c in range(4)]})  # fmt: skip
        elif isinstance(l, ParameterLimit):
            v = list(l.v)
            if l.type == MMX_LIMIT_MINMAX:
                lj.append({"type": "minmax", "weight": l.weight, "parameter": rig.param_names[l.index0], "limits": v[:2]})
            elif l.type == MMX_LIMIT_MINMAX_JOINT:
                lj.append({"type": "minmax_joint", "weight": l.weight, "jointIndex": rig.joint_names[l.index0 // 7],
                           "jointParameter": JOINT_PARAMETER_NAMES[l.index0 % 7], "limits": v[:2]})  # fmt: skip
            elif l.type == MMX_LIMIT_LINEAR:
                lj.append({"type": "linear", "weight": l.weight, "referenceParameter": rig.param_names[l.index0],
                           "targetParameter": rig.param_names[l.index1], "scale": v[0], "offset": v[1], "rangeMin": v[2], "rangeMax": v[3]})  # fmt: skip
            elif l.type == MMX_LIMIT_LINEAR_JOINT:
                lj.append({"type": "linear_joint", "weight": l.weight, "referenceJoint": rig.joint_names[l.index0 // 7],
                           "referenceJointParameter": l.index0 % 7, "targetJoint": rig.joint_names[l.index1 // 7],
                           "targetJointParameter": l.index1 % 7, "scale": v[0], "offset": v[1], "rangeMin": v[2], "rangeMax": v[3]})  # fmt: skip
            elif l.type == MMX_LIMIT_HALFPLANE:
                lj.append({"type": "half_plane", "weight": l.weight, "param1": rig.param_names[l.index0], "param2": rig.param_names[l.index1],
                           "normal": v[:2], "offset": v[2]})  # fmt: skip
    roots = [j for j in range(J) if rig.parent[j] < 0]
    return {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": roots}], "nodes": nodes,
            "extensionsUsed": ["FB_momentum"],
            "extensions": {"FB_momentum": {"transform": {"parameters": list(rig.param_names), "joints": joints}, "parameterLimits": lj}}}  # fmt: skip


def to_glb(doc: dict) -> bytes:
    """GLB container with only a JSON chunk."""
    import struct

    payload = json.dumps(doc).encode("utf-8")
    payload += b" " * ((4 - len(payload) % 4) % 4)
    return b"glTF" + struct.pack("<II", 2, 12 + 8 + len(payload)) + struct.pack("<I4s", len(payload), b"JSON") + payload
