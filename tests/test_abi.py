"""CPU checks of the C-ABI boundary: the built library loads, exports every symbol include/mmx.h
declares, struct layouts match the ctypes mirrors, and the product path fails LOUDLY without a GPU."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from momentum_amd import _abi, make_test_character
from momentum_amd import build as mbuild
from momentum_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mmx.h")


@pytest.fixture(scope="module")
def L():
    mbuild.build()
    return capi.lib()


def _declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mmx_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_match_binding_list():
    assert _declared_symbols() == sorted(capi.SYMBOLS)


def test_library_exports_every_declared_symbol(L):
    for name in _declared_symbols():
        assert hasattr(L, name), f"libmmx_hip.so does not export {name}"
    assert L.mmx_abi_version() == _abi.MMX_ABI_VERSION


def test_struct_layouts_match_header():
    prog = r"""
    #include <stdio.h>
    #include <stddef.h>
    #include "mmx.h"
    int main(void) {
      printf("%zu %zu %zu\n", sizeof(mmx_rig_desc), sizeof(mmx_constraint_data), sizeof(mmx_gn_options));
      printf("%zu %zu %zu\n", offsetof(mmx_rig_desc, pt_offsets), offsetof(mmx_constraint_data, memory), offsetof(mmx_gn_options, lm_down));
      printf("%zu %zu %zu %zu\n", sizeof(mmx_parameter_limit), offsetof(mmx_constraint_data, limits), offsetof(mmx_constraint_data, model_weights),
             offsetof(mmx_constraint_data, ori_loss_c));
      printf("%zu %zu %zu %zu\n", sizeof(mmx_joint_constraint_block), offsetof(mmx_joint_constraint_block, plane_d),
             offsetof(mmx_joint_constraint_block, loss_c), offsetof(mmx_constraint_data, joint_blocks));
      printf("%zu %zu %zu\n", sizeof(mmx_ellipsoid_limit), offsetof(mmx_ellipsoid_limit, parent), offsetof(mmx_constraint_data, ellipsoid_limits));
      return 0;
    }"""
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "t.c")
        open(c, "w").write(prog)
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe]).decode().split()
    sizes = [int(x) for x in out]
    assert sizes[:3] == [C.sizeof(_abi.RigDesc), C.sizeof(_abi.ConstraintData), C.sizeof(_abi.GnOptions)]
    assert sizes[3:6] == [_abi.RigDesc.pt_offsets.offset, _abi.ConstraintData.memory.offset, _abi.GnOptions.lm_down.offset]
    assert sizes[14:] == [C.sizeof(_abi.EllipsoidLimit), _abi.EllipsoidLimit.parent.offset, _abi.ConstraintData.ellipsoid_limits.offset]
    assert sizes[10:14] == [C.sizeof(_abi.JointConstraintBlock), _abi.JointConstraintBlock.plane_d.offset,
                          _abi.JointConstraintBlock.loss_c.offset, _abi.ConstraintData.joint_blocks.offset]  # fmt: skip
    assert sizes[6:10] == [C.sizeof(_abi.ParameterLimit), _abi.ConstraintData.limits.offset, _abi.ConstraintData.model_weights.offset,
                         _abi.ConstraintData.ori_loss_c.offset]  # fmt: skip


def test_default_options_match_reference_structs(L):
    # SolverOptions{1,2,1} (solver.h:19-34), GaussNewtonSolverBaseOptions{0.05,false} (gauss_newton_solver.h:17-33)
    o = _abi.GnOptions()
    L.mmx_gn_options_default(C.byref(o))
    assert (o.min_iterations, o.max_iterations, o.threshold, o.do_line_search, o.step_rule) == (1, 2, 1.0, 0, 0)
    assert abs(o.regularization - 0.05) < 1e-9


def test_invalid_rig_is_rejected_like_mt_check(L):
    rig = make_test_character(4)
    rig.parent[2] = 3  # child before parent violates the Skeleton invariant (skeleton.cpp:16-22)
    with pytest.raises(capi.MmxError) as ei:
        capi.host_tables(rig)
    assert "parent-before-child" in str(ei.value)
    rig = make_test_character(4)
    rig.pt_inner[0] = 99  # column index beyond numAllModelParameters
    with pytest.raises(capi.MmxError):
        capi.host_tables(rig)


def test_no_gpu_means_loud_failure_not_cpu_fallback(L):
    if capi.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(capi.MmxError) as ei:
        capi.RigHandle(make_test_character(3))
    assert ei.value.code == 6  # MMX_ERR_NO_DEVICE
    assert "no CPU fallback" in str(ei.value)


def test_product_package_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under momentum_amd/ (nor include/) may import,
    include or link it."""
    bad = []
    for base in ("momentum_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                    txt = open(os.path.join(dp, fn), errors="replace").read()
                    if re.search(r"^\s*(from|import)\s+oracle|#include\s+\"[^\"]*oracle|libmmx_oracle", txt, flags=re.M):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad
    out = subprocess.check_output(["readelf", "-d", capi.LIB_PATH]).decode()
    assert "oracle" not in out


def test_build_recipe_flags_per_translation_unit():
    """momentum_amd/build.py: the solve kernels' translation units (mmx_fused.hip groups 0-3, mmx_f64.hip) are compiled without
    machine-level LICM and loop strength reduction (the kernels spill otherwise: profiles/r05_exp_fused.txt item 14), the wide
    route's (mmx_kernels.hip, the tree kernels = group 4 of mmx_fused.hip) and the host-side files with the default pipeline;
    every group the source dispatches to is built."""
    import re

    from momentum_amd import build as mbuild

    solve = ["-mllvm", "-disable-machine-licm", "-mllvm", "-disable-lsr"]
    for g in (0, 1, 2, 3, 5, 6):
        assert mbuild._extra_flags("mmx_fused.hip", g) == solve
    assert mbuild._extra_flags("mmx_f64.hip", None) == solve
    for src, g in (("mmx_fused.hip", 4), ("mmx_kernels.hip", None), ("mmx_capi.hip", None), ("mmx_comm.hip", None), ("mmx_host_tables.cpp", None)):
        assert mbuild._extra_flags(src, g) == []
    text = open(os.path.join(os.path.dirname(mbuild.__file__), "csrc", "mmx_fused.hip")).read()
    groups = {int(g) for g in re.findall(r"MMX_FUSED_GROUP == (\d+)", text)} - {8, 9}  # (8, 9: the one-instantiation compile probes)
    assert groups == set(range(mbuild.FUSED_GROUPS))


def test_production_instantiations_do_not_spill():
    """The RESULT the recipe's two -mllvm switches exist for, not the flag list: the three production instantiations of the
    one-launch solve (BASELINE configs[1]'s six blocks: plain Gauss-Newton, LM schedule, generic rule), compiled the way
    build.py compiles them on this toolchain (scripts/probes/fused_one.sh), keep at most 8 spilled vector registers at 128
    registers = four workgroups per CU.  A ROCm that drops, renames or re-tunes either switch fails HERE instead of silently
    giving the speed back (default pipeline: 54 / 354 spilled, 6-13 % slower; profiles/r05_exp_fused.txt item 14).  When the
    compiler rejects the switches build.py falls back to the default pipeline and says so in build_info.json: the test then
    checks that the stamp admits it."""
    import json
    from concurrent.futures import ThreadPoolExecutor

    from momentum_amd import build as mbuild

    flags = mbuild.SOLVE_KERNEL_FLAGS if mbuild.solve_flags_accepted() else []
    probe = os.path.join(ROOT, "scripts", "probes", "fused_one.sh")

    def one(rule):
        out = subprocess.run(["bash", probe, f"-DMMX_PROBE_RULE={rule}"] + flags, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900).stdout.decode()
        m = re.search(r"VGPRs: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+)", out, flags=re.S)  # (the first kernel listed: the solve)
        assert m, out
        return rule, tuple(int(x) for x in m.groups())

    with ThreadPoolExecutor(max_workers=3) as ex:
        got = dict(ex.map(one, (0, 1, -1)))
    info = mbuild.build_info()
    if not flags:
        assert info.get("solve_kernel_pipeline") == "default", info
        pytest.skip(f"hipcc rejects {mbuild.SOLVE_KERNEL_FLAGS}: default pipeline (stamped), figures {got}")
    for rule, (vgprs, occ, sspill, vspill) in got.items():
        assert vgprs <= 128 and occ == 4, (rule, got)
        assert vspill <= (10 if rule >= 0 else 12), (rule, got)  # (round 6: 9 / 6 / 12 -- the ride-along forward substitution, the per-iteration estimate; default pipeline: 37 / 54 / 94)
        assert sspill <= 340, (rule, got)
    if os.path.exists(mbuild.LIB) and info:
        assert info.get("solve_kernel_pipeline") == "no-machine-licm,no-lsr", info
        json.dumps(info)
