"""Rig ingestion (momentum_amd/model_io.py) against the reference's own parser
tests: section handling of io_model_parser_test.cpp:16-117 (duplicate sections are concatenated),
the write -> parse round trip of io_parameter_limits_test.cpp (createCharacterWithLimits: MinMax,
MinMaxJoint, Linear, the four-segment piecewise Linear, LinearJoint), the documented piecewise
example of parameter_limits_io.cpp:355-365, and a whole-character round trip of createTestCharacter."""
import json

import numpy as np
import pytest

from momentum_amd import make_test_character, model_io
from momentum_amd._abi import ParameterLimit

FLT_MAX = float(np.finfo(np.float32).max)


def _same(a: ParameterLimit, b: ParameterLimit) -> bool:
    return (a.type, a.index0, a.index1) == (b.type, b.index0, b.index1) and a.weight == pytest.approx(b.weight) and list(a.v) == pytest.approx(list(b.v))


def test_duplicate_sections_are_concatenated():
    text = """Momentum Model Definition V1.0

[Limits]
limit param1 minmax [-1.0, 1.0] 1.0

[ParameterTransform]
# Some parameter transforms
root.tx = 1.0 * param1

[Limits]
# Second set of limits
limit param2 minmax [-2.0, 2.0] 1.0

[PoseConstraints]
poseconstraints tpose param1=0.0

[Limits]
limit param3 minmax [-3.0, 3.0] 1.0
"""
    sec = model_io.load_momentum_model(text)
    assert "Limits" in sec
    for k in (1, 2, 3):
        assert f"limit param{k} minmax" in sec["Limits"]
    assert sec["ParameterTransform"].strip() == "root.tx = 1.0 * param1"
    with pytest.raises(model_io.ModelFormatError):
        model_io.load_momentum_model("Not a model file\n[Limits]\n")


def test_channel_expressions():
    joints = ["root", "child", "leaf"]
    text = """
root.tx = 1.0*tx
root.rz = 0.5*shared + 0.25
child.rz = 0.5*shared + 2.0*twist
leaf.rx = 0.5*child.rz          # copies child.rz's parameters, scaled
leaf.ry = 0.0*unused_zero
"""
    names, trip, offsets = model_io.parse_parameter_transform(text, joints)
    assert names == ["tx", "shared", "twist", "unused_zero"]
    A = np.zeros((21, 4))
    for r, c, v in trip:
        A[r, c] += v
    assert A[0, 0] == 1 and A[5, 1] == 0.5 and offsets[5] == np.float32(0.25)
    assert A[7 + 5, 1] == 0.5 and A[7 + 5, 2] == 2.0
    assert A[14 + 3, 1] == 0.25 and A[14 + 3, 2] == 1.0
    assert not A[:, 3].any()  # zero weights are dropped (:361-366) but the parameter exists
    with pytest.raises(model_io.ModelFormatError):
        model_io.parse_parameter_transform("nojoint.rx = 1.0*a", joints)
    with pytest.raises(model_io.ModelFormatError):
        model_io.parse_parameter_transform("root.qq = 1.0*a", joints)


def test_documented_piecewise_linear_example():
    lim = model_io.parse_parameter_limits("limit param1 linear param2 [-1, 3, -3] [1, -3, 0] [-2, -3] 4.0\n", [], ["param1", "param2"])
    assert len(lim) == 3 and all(l.weight == 4.0 and (l.index0, l.index1) == (0, 1) for l in lim)
    assert [list(l.v) for l in lim] == [[-1, 3, -FLT_MAX, -3], [1, -3, -3, 0], [-2, -3, 0, FLT_MAX]]
    with pytest.raises(model_io.ModelFormatError):  # segments must be continuous (:414-436)
        model_io.parse_parameter_limits("limit param1 linear param2 [-1, 3, -3] [1, 5, 0] [-2, -3]\n", [], ["param1", "param2"])
    with pytest.raises(model_io.ModelFormatError):
        model_io.parse_parameter_limits("limit param1 bogus [0, 1]\n", [], ["param1"])
    hp = model_io.parse_parameter_limits("limit a halfplane b [3, 4] 10 2.0\n", [], ["a", "b"])[0]
    assert list(hp.v)[:3] == pytest.approx([0.6, 0.8, 2.0]) and hp.weight == 2.0  # normalised (:568-571)


def test_limits_write_parse_round_trip_like_the_reference_test():
    rig = make_test_character(5)
    P = rig.param_names
    limits = [
        ParameterLimit.minmax(1, -0.2, 0.1, 1.5),
        ParameterLimit.minmax_joint(1, 2, -0.3, 0.0, 2.0),
        ParameterLimit.linear(2, 1, 3.0, 2.0, -FLT_MAX, FLT_MAX, 2.0),
        # f(x) = -x-3 (x<-3) ; x+3 (-3<=x<0) ; -2x+3 (0<=x<3) ; 0.5x-4.5 (x>=3)   [io_parameter_limits_test.cpp]
        ParameterLimit.linear(2, 1, -1.0, 3.0, -FLT_MAX, -3.0, 2.5),
        ParameterLimit.linear(2, 1, 1.0, -3.0, -3.0, 0.0, 2.5),
        ParameterLimit.linear(2, 1, -2.0, -3.0, 0.0, 3.0, 2.5),
        ParameterLimit.linear(2, 1, 0.5, 4.5, 3.0, FLT_MAX, 2.5),
        ParameterLimit.linear(2, 1, 1.2, 0.3, -FLT_MAX, FLT_MAX, 2.5),
        ParameterLimit.linear_joint(2, 3, 1, 0, 1.0, 0.0, -FLT_MAX, 0.0, 2.5),
        ParameterLimit.linear_joint(2, 3, 1, 0, 2.0, 0.0, 0.0, FLT_MAX, 2.5),
        ParameterLimit.halfplane(0, 3, 0.6, 0.8, 0.25, 1.0),
    ]
    text = model_io.write_parameter_limits(limits, rig.joint_names, P)
    assert text.count("\n") == 7  # the piecewise entries share one line each
    back = model_io.limits_for_solver(model_io.parse_parameter_limits(text, rig.joint_names, P))
    assert len(back) == len(limits)
    for a, b in zip(limits, back):
        assert _same(a, b), (a.type, list(a.v), list(b.v))
    passive = model_io.parse_parameter_limits(f"limit {rig.joint_names[2]}.ry minmax_passive [0, 0.5] 2.0\n", rig.joint_names, P)
    assert isinstance(passive[0], dict) and model_io.limits_for_solver(passive) == []


@pytest.mark.parametrize("n", [3, 24])
def test_character_round_trip(n):
    rig = make_test_character(n)
    doc = json.dumps(model_io.skeleton_to_legacy_json(rig))
    model = "Momentum Model Definition V1.0\n\n[ParameterTransform]\n" + model_io.write_parameter_transform(rig) + "\n[Limits]\nlimit root_rx minmax [-1, 1]\n"
    back, limits = model_io.load_character(doc, model)
    assert back.num_params == rig.num_params and back.param_names == rig.param_names and back.joint_names == rig.joint_names
    for f in ("parent", "pre_rotation", "translation_offset", "pt_outer", "pt_inner", "pt_value", "pt_offsets"):
        assert np.array_equal(getattr(back, f), getattr(rig, f)), f
    assert len(limits) == 1 and limits[0].index0 == rig.param_names.index("root_rx") and limits[0].weight == 1.0


@pytest.mark.gpu
def test_loaded_character_solves_like_the_original(orc):
    import torch

    from momentum_amd import capi
    from momentum_amd._abi import GnOptions
    from tests.helpers import make_problem

    rig = make_test_character(8)
    back, _ = model_io.load_character(
        model_io.skeleton_to_legacy_json(rig), "Momentum Model Definition V1.0\n[ParameterTransform]\n" + model_io.write_parameter_transform(rig)
    )
    cons, th0, _ = make_problem(rig, [7, 3], [6], 4, seed=5, perturb=0.3)
    opt = GnOptions.make(min_iterations=6, max_iterations=6, regularization=0.05)
    out = []
    for r in (rig, back):
        pb = capi.Problem(capi.RigHandle(r, 0), 4, cons.pos_parent, cons.ori_parent)
        pb.set_constraints(cons.pos_offset.reshape(4, 2, 3), cons.pos_target.reshape(4, 2, 3), cons.pos_weight.reshape(4, 2),
                           cons.ori_offset.reshape(4, 1, 4), cons.ori_target.reshape(4, 1, 4), cons.ori_weight.reshape(4, 1))  # fmt: skip
        out.append(pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt)["theta"].cpu().numpy())
    assert np.array_equal(out[0], out[1])


def test_gltf_round_trip_with_momentum_extension():
    """Character -> glTF (FB_momentum extension: skeleton_joint nodes, transform, parameterLimits) -> GLB
    container -> Character: arrays identical (lengths travel in metres), limits identical."""
    from momentum_amd._abi import EllipsoidLimit

    rig = make_test_character(7)
    limits = [
        ParameterLimit.minmax(1, -0.2, 0.1, 1.5),
        ParameterLimit.minmax_joint(1, 2, -0.3, 0.0, 2.0),
        ParameterLimit.linear(2, 1, 3.0, 2.0, -FLT_MAX, FLT_MAX, 2.0),
        ParameterLimit.linear_joint(2, 3, 1, 0, 1.0, 0.0, -FLT_MAX, 0.0, 2.5),
        ParameterLimit.halfplane(0, 3, 0.6, 0.8, 0.25, 1.0),
        EllipsoidLimit.make(5, [1.0, 2.0, 3.0], 2, [2.0, 3.0, 4.0], [-60.0, 45.0, 90.0], [0.8, 0.9, 1.3], 4.0),
    ]
    glb = model_io.to_glb(model_io.write_gltf(rig, limits))
    assert glb[:4] == b"glTF"
    back, lim2 = model_io.load_gltf(glb)
    assert back.joint_names == rig.joint_names and back.param_names == rig.param_names
    for f in ("parent", "pre_rotation", "pt_outer", "pt_inner", "pt_value", "pt_offsets"):
        assert np.array_equal(getattr(back, f), getattr(rig, f)), f
    assert np.abs(back.translation_offset - rig.translation_offset).max() <= 1e-6  # cm -> m -> cm
    assert len(lim2) == len(limits)
    for a, b in zip(limits[:-1], lim2[:-1]):
        assert _same(a, b)
    ea, eb = limits[-1], lim2[-1]
    assert (ea.parent, ea.ellipsoid_parent, ea.weight) == (eb.parent, eb.ellipsoid_parent, eb.weight)
    assert np.abs(np.array(list(ea.ellipsoid)) - np.array(list(eb.ellipsoid))).max() <= 1e-5
    assert np.abs(np.array(list(ea.offset)) - np.array(list(eb.offset))).max() <= 1e-5


def test_gltf_without_extension_every_unskinned_node_is_a_joint():
    """No FB_momentum extension: every hierarchy node without a mesh becomes a joint, in depth-first
    order, translations converted from metres (gltf_skeleton_io.cpp:79-175,267-274)."""
    doc = {
        "asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}],
        "nodes": [
            {"name": "root", "children": [1, 3], "translation": [0.0, 1.0, 0.0]},
            {"name": "spine", "children": [2], "rotation": [0.0, 0.0, 0.7071068, 0.7071068], "translation": [0.0, 0.25, 0.0]},
            {"name": "head", "translation": [0.0, 0.1, 0.0]},
            {"name": "body_mesh", "mesh": 0},
            {"name": "unreachable"},
        ],
        "meshes": [{"primitives": []}],
    }  # fmt: skip
    rig, limits = model_io.load_gltf(json.dumps(doc))
    assert rig.joint_names == ["root", "spine", "head"] and list(rig.parent) == [-1, 0, 1] and limits == []
    assert np.allclose(rig.translation_offset, [[0, 100, 0], [0, 25, 0], [0, 10, 0]])
    assert np.allclose(rig.pre_rotation[1], [0, 0, 0.7071068, 0.7071068]) and np.allclose(rig.pre_rotation[0], [0, 0, 0, 1])
    assert rig.num_params == 0


@pytest.mark.parametrize("name,joints", [("blender_simple_armature.glb", None), ("skeleton_non_joint_root.glb", None), ("sort_joints.glb", None)])
def test_reference_glb_resources_load(name, joints):
    """The reference's own GLB test resources (read in place when the checkout is present; they carry
    no FB_momentum extension): a parent-before-child skeleton with unit pre-rotations comes out."""
    import os

    path = os.path.join("/root/reference/momentum/test/resources", name)
    if not os.path.exists(path):
        pytest.skip("reference checkout not present")
    rig, limits = model_io.load_gltf(open(path, "rb").read())
    assert rig.num_joints >= 2 and limits == []
    assert rig.parent[0] == -1 and all(rig.parent[j] < j for j in range(rig.num_joints))
    assert np.allclose(np.linalg.norm(rig.pre_rotation, axis=1), 1.0, atol=1e-4)
    assert len(set(rig.joint_names)) == rig.num_joints


def _glb_with_motion(poses: np.ndarray, count=None, stride=None, byte_offset=0, truncate=0):
    """A minimal GLB whose FB_momentum extension stores `poses` (nframes x nparams floats) in its binary chunk; the accessor's
    count / stride / offset can be forged, the binary chunk truncated."""
    import json
    import struct

    nframes, nparams = poses.shape
    blob = np.ascontiguousarray(poses, "<f4").tobytes()
    doc = {
        "asset": {"version": "2.0"},
        "buffers": [{"byteLength": len(blob)}],
        "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": len(blob), **({"byteStride": stride} if stride else {})}],
        "accessors": [{"bufferView": 0, "byteOffset": byte_offset, "componentType": 5126, "type": "SCALAR", "count": poses.size if count is None else count}],
        "extensions": {"FB_momentum": {"fps": 30.0, "motion": {"nframes": nframes, "poses": 0, "parameterNames": [f"p{i}" for i in range(nparams)]}}},
    }
    js = json.dumps(doc).encode()
    js += b" " * (-len(js) % 4)
    blob = blob[: len(blob) - truncate]
    body = struct.pack("<I4s", len(js), b"JSON") + js + struct.pack("<I4s", len(blob), b"BIN\0") + blob
    return b"glTF" + struct.pack("<II", 2, 12 + len(body)) + body


def test_glb_accessors_are_bounds_checked():
    """An accessor of an untrusted GLB that reaches past the binary chunk is an error, not an out-of-bounds read (the
    reference's copyAccessorBuffer goes through fx::gltf's size checks, momentum/io/gltf/utils/accessor_utils.h)."""
    poses = np.arange(12, dtype=np.float32).reshape(3, 4)
    ok = model_io.load_gltf_motion(_glb_with_motion(poses))
    assert np.array_equal(ok["poses"], poses) and ok["parameter_names"] == ["p0", "p1", "p2", "p3"]
    for forged in (
        _glb_with_motion(poses, count=10**6),  # far more elements than the chunk holds
        _glb_with_motion(poses, count=13),  # one element past the end
        _glb_with_motion(poses, byte_offset=8),  # the same count from a later start
        _glb_with_motion(poses, stride=2),  # a stride below the element size
        _glb_with_motion(poses, stride=64),  # a stride that walks out of the chunk
        _glb_with_motion(poses, count=-3),
    ):
        with pytest.raises(model_io.ModelFormatError):
            model_io.load_gltf_motion(forged)
    cut = _glb_with_motion(poses, truncate=0)[:-16]  # the file ends before its BIN chunk does
    with pytest.raises(model_io.ModelFormatError):
        model_io.load_gltf_motion(cut)
