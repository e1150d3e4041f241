"""momentum_amd -- MI355X-native batched inverse kinematics (the FK -> Jacobian -> Gauss-Newton
hot path of facebookresearch/momentum) behind a C ABI (include/mmx.h).

The compute lives in momentum_amd/csrc (hand-written HIP for gfx950, built into
momentum_amd/libmmx_hip.so).  This Python package is host-side plumbing only: rig data model,
ctypes binding, torch device buffers/streams.  There is NO CPU fallback: every compute entry
point raises if the HIP library or a GPU is missing.
"""
from .rigs import (  # noqa: F401
    Rig,
    make_test_character,
    make_humanoid72,
    make_rig300,
    humanoid72_landmark_joints,
)

__all__ = ["Rig", "make_test_character", "make_humanoid72", "make_rig300", "humanoid72_landmark_joints"]
