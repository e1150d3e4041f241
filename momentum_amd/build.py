"""Builds momentum_amd/libmmx_hip.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc.

In-tree artefact: the .so is git-ignored but travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# A/B experiments on one GPU box: MMX_BUILD_VARIANT=name builds momentum_amd/libmmx_hip_<name>.so with
# -DMMX_EXP_<NAME> (objects kept apart); MMX_LIB=<path> makes capi.py load that library instead.
VARIANT = os.environ.get("MMX_BUILD_VARIANT", "")
# ... and MMX_BUILD_FLAGS="..." appends compiler flags to such a variant build (e.g. -mllvm -amdgpu-sched-strategy=max-ilp)
VARIANT_FLAGS = os.environ.get("MMX_BUILD_FLAGS", "").split() if VARIANT else []
LIB = os.path.join(HERE, f"libmmx_hip_{VARIANT}.so" if VARIANT else "libmmx_hip.so")
SOURCES = ["mmx_kernels.hip", "mmx_fused.hip", "mmx_capi.hip", "mmx_comm.hip", "mmx_f64.hip", "mmx_host_tables.cpp"]
FUSED_GROUPS = 7  # mmx_fused.hip is compiled once per group of template instantiations, in parallel (4: the wide route's tree kernels; 5, 6: the mixed-precision instantiations)
# The solve kernels (one-launch solve, double solve) are compiled WITHOUT the machine-level loop-invariant code motion and
# WITHOUT loop strength reduction: at their register budgets the per-lane address arithmetic the first hoists out of the
# iteration loop and the induction pointers the second creates are what gets spilled.  Spilled VGPRs of the BASELINE
# configs[1] instantiation 37 -> 4 -> 0, LM schedule 54 -> 16 -> 0, generic rule 94 -> 49 -> 0, double solve 186 -> 40 -> 7
# (scripts/probes/fused_one.sh prints them in seconds).  Measured, one box each: first flag headline + 1.7 %, cfg3 + 4.6 %,
# line search + 13 %, double solve + 3.9 %; second flag on top + 4 % / + 2.5 % / + 1-4 % / + 0.7 %.  The wide route's kernels
# (mmx_kernels.hip, the tree kernels = group 4) lose 0.2-0.3 % with either and keep the default pipeline
# (profiles/r05_exp_fused.txt).
SOLVE_KERNEL_FLAGS = ["-mllvm", "-disable-machine-licm", "-mllvm", "-disable-lsr"]
# The two switches are internal LLVM options, not a stable interface: they are PROBED once per build (a one-line kernel
# compiled with them).  A toolchain that rejects them gets the default pipeline with a loud warning instead of a failed build
# (the same source then spills 54 vector registers in the headline instantiation and runs 6-13 % slower, r05_exp_fused.txt);
# which pipeline a library was built with is recorded in momentum_amd/build_info.json, printed by bench.py in its line, and
# tests/test_abi.py holds the RESULT (spilled registers and occupancy of the production instantiations), not the flag list.
BUILD_INFO = os.path.join(HERE, "build_info.json")
_probe_cache = {}


def solve_flags_accepted() -> bool:
    """Does this hipcc take SOLVE_KERNEL_FLAGS?  (cached; MMX_BUILD_DEFAULT_PIPELINE=1 answers no without asking)"""
    if os.environ.get("MMX_BUILD_DEFAULT_PIPELINE"):
        return False
    if "ok" not in _probe_cache:
        import tempfile

        with tempfile.TemporaryDirectory() as td:
            src = os.path.join(td, "probe.hip")
            with open(src, "w") as f:
                f.write("#include <hip/hip_runtime.h>\n__global__ void k(float* p) { for (int i = 0; i < 8; ++i) p[i * 3] += 1.f; }\n")
            cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-c", src, "-o", os.path.join(td, "probe.o")] + SOLVE_KERNEL_FLAGS
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            _probe_cache["ok"] = r.returncode == 0
            if r.returncode != 0:
                print(
                    "WARNING momentum_amd/build.py: hipcc rejects " + " ".join(SOLVE_KERNEL_FLAGS) + " -- the solve kernels are built with the "
                    "DEFAULT pipeline (they spill vector registers and run 6-13 % slower; build_info.json says so):\n" + r.stderr.decode(errors="replace")[-400:],
                    file=sys.stderr,
                )
    return _probe_cache["ok"]


def _wants_solve_flags(src: str, group) -> bool:
    return src in ("mmx_f64.hip",) or (src == "mmx_fused.hip" and group != 4)


def _extra_flags(src: str, group) -> list:
    """The flags the recipe WANTS for a translation unit (build() drops them when the compiler does not take them)."""
    if _wants_solve_flags(src, group):
        return SOLVE_KERNEL_FLAGS
    return []


def build_info() -> dict:
    """What the library on disk was built with ({} when there is no stamp)."""
    import json

    try:
        with open(BUILD_INFO) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}
HEADERS = ["mmx_device.hpp", "mmx_device_d.hpp", "mmx_kernels.hpp", "mmx_tree.hpp", "mmx_host_tables.hpp", os.path.join("..", "..", "include", "mmx.h")]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build the gfx950 kernels)")


def needs_build() -> bool:
    if not os.path.exists(LIB) or (not VARIANT and not os.path.exists(BUILD_INFO)):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    jobs, keep = [], []
    flags_ok = solve_flags_accepted()
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    for src in SOURCES:
        groups = range(FUSED_GROUPS) if src == "mmx_fused.hip" else [None]
        for g in groups:
            stem = os.path.splitext(src)[0] + ("" if g is None else f"_g{g}")
            obj = os.path.join(CSRC, stem + (f".{VARIANT}" if VARIANT else "") + ".o")
            cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", os.path.join(CSRC, src), "-o", obj]
            if VARIANT:
                cmd.insert(1, f"-DMMX_EXP_{VARIANT.upper()}")
                cmd[1:1] = VARIANT_FLAGS
            if g is not None:
                cmd.insert(1, f"-DMMX_FUSED_GROUP={g}")
            if flags_ok:  # (A/B: MMX_BUILD_DEFAULT_PIPELINE=1 compiles everything with the default pipeline)
                cmd[1:1] = _extra_flags(src, g)
            if src.endswith(".cpp"):
                cmd.insert(1, "-x")
                cmd.insert(2, "hip")
            # incremental: an object newer than its source and every header is reused (force rebuilds everything)
            if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(hdr_time, os.path.getmtime(os.path.join(CSRC, src))):
                keep.append((None, obj))
                continue
            jobs.append((cmd, obj))
    from concurrent.futures import ThreadPoolExecutor

    def run(job):
        if verbose:
            print(" ".join(job[0]), file=sys.stderr)
        subprocess.check_call(job[0])
        return job[1]

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    objs = []
    for src in SOURCES:  # link in the fixed source order whatever was recompiled
        groups = range(FUSED_GROUPS) if src == "mmx_fused.hip" else [None]
        for g in groups:
            stem = os.path.splitext(src)[0] + ("" if g is None else f"_g{g}")
            objs.append(os.path.join(CSRC, stem + (f".{VARIANT}" if VARIANT else "") + ".o"))
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    if not VARIANT:
        import json

        ver = subprocess.run([_hipcc(), "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode(errors="replace")
        m = [ln for ln in ver.splitlines() if "HIP version" in ln or "clang version" in ln]
        with open(BUILD_INFO, "w") as f:
            json.dump(
                {
                    "solve_kernel_pipeline": "no-machine-licm,no-lsr" if flags_ok else "default",
                    "solve_kernel_flags": SOLVE_KERNEL_FLAGS if flags_ok else [],
                    "flags_rejected_by_compiler": (not flags_ok) and not os.environ.get("MMX_BUILD_DEFAULT_PIPELINE"),
                    "arch": ARCH,
                    "compiler": " | ".join(m),
                },
                f,
            )
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
