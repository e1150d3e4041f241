"""The C++ shell that keeps momentum's class surface (include/momentum_amd/momentum_amd.hpp):
compiles everywhere (CPU check), runs its smoke program on the GPU box."""
import os
import subprocess

import pytest

from momentum_amd import build as mbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_shell.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "test_shell")


PARITY_SRC = os.path.join(ROOT, "tests", "cpp", "test_shell_parity.cpp")
PARITY_EXE = os.path.join(ROOT, "tests", "cpp", "test_shell_parity")
GOLDEN_INC = os.path.join(ROOT, "tests", "cpp", "golden_cfg2.inc")


def _compile(src=SRC, exe=EXE):
    mbuild.build()
    libdir = os.path.join(ROOT, "momentum_amd")
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-L", libdir, "-lmmx_hip",
           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]  # fmt: skip
    subprocess.check_call(cmd)


MULTI_SRC = os.path.join(ROOT, "tests", "cpp", "test_multi_gpu.cpp")
MULTI_EXE = os.path.join(ROOT, "tests", "cpp", "test_multi_gpu")


def _golden_header():
    import sys

    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "cpp", "make_golden_header.py"),
                           os.path.join(ROOT, "tests", "golden", "cfg2_humanoid72.npz"), GOLDEN_INC])  # fmt: skip


def _compile_multi():
    _golden_header()
    mbuild.build()
    libdir = os.path.join(ROOT, "momentum_amd")
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", "-pthread", "-I", os.path.join(ROOT, "include"), MULTI_SRC, "-L", libdir, "-lmmx_hip",
           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", MULTI_EXE]  # fmt: skip
    subprocess.check_call(cmd)


def _compile_parity():
    # the committed golden fixture (inputs + the oracle's double-precision answers) as a C++ include
    import sys

    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "cpp", "make_golden_header.py"),
                           os.path.join(ROOT, "tests", "golden", "cfg2_humanoid72.npz"), GOLDEN_INC])  # fmt: skip
    _compile(PARITY_SRC, PARITY_EXE)


def test_cpp_shell_compiles_and_links():
    _compile()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_shell_solves_on_gpu():
    _compile()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("OK")


def test_cpp_shell_parity_program_compiles_and_links():
    _compile_parity()
    assert os.path.exists(PARITY_EXE)


@pytest.mark.gpu
def test_cpp_shell_matches_golden_fixture_on_gpu():
    """Character -> DeviceCharacter -> BatchedSkeletonSolverFunction -> BatchedGaussNewtonSolver on the
    numbers of tests/golden/cfg2_humanoid72.npz: 1e-5 on the pose parameters against the oracle's stored
    double-precision solve; per-element characters / parents reproduce it bit for bit."""
    _compile_parity()
    out = subprocess.run([PARITY_EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("OK"), out.stdout


REAL_SRC = os.path.join(ROOT, "tests", "cpp", "test_real_rig.cpp")
REAL_EXE = os.path.join(ROOT, "tests", "cpp", "test_real_rig")
REAL_INC = os.path.join(ROOT, "tests", "cpp", "golden_real_rig.inc")


def _compile_real_rig():
    import sys

    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "cpp", "make_golden_header.py"),
                           os.path.join(ROOT, "tests", "golden", "real_rig_character_with_motion.npz"), REAL_INC])  # fmt: skip
    _compile(REAL_SRC, REAL_EXE)


def test_cpp_real_rig_program_compiles_and_links():
    _compile_real_rig()
    assert os.path.exists(REAL_EXE)


@pytest.mark.gpu
def test_cpp_real_rig_skeleton_state_and_solve_on_gpu():
    """The reference's character_with_motion.glb through the C++ shell: SkeletonState at the stored motion frames against the
    oracle's double FK, then the solve towards them (tests/cpp/test_real_rig.cpp; the Python twin is tests/test_real_rig.py)."""
    _compile_real_rig()
    out = subprocess.run([REAL_EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("OK"), out.stdout


ADAPTER_EXE = os.path.join(ROOT, "integration", "adapter_check")


def _compile_adapter():
    """integration/tensor_ik_mmx_adapter.cpp (the binding a momentum maintainer adds next to
    solveTensorIKProblem) against a stub of the reference types it touches: a drift of include/mmx.h breaks this build."""
    mbuild.build()
    libdir = os.path.join(ROOT, "momentum_amd")
    idir = os.path.join(ROOT, "integration")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", idir,
                           os.path.join(idir, "tensor_ik_mmx_adapter.cpp"), os.path.join(idir, "adapter_check.cpp"), "-L", libdir, "-lmmx_hip",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", ADAPTER_EXE])  # fmt: skip


def test_reference_side_adapter_compiles_and_converts_a_character():
    _compile_adapter()
    out = subprocess.run([ADAPTER_EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("OK"), out.stdout


@pytest.mark.gpu
def test_reference_side_adapter_solves_on_gpu():
    _compile_adapter()
    out = subprocess.run([ADAPTER_EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "four elements on their targets" in out.stdout and out.stdout.strip().endswith("OK"), out.stdout


STUB_SRC = os.path.join(ROOT, "tests", "cpp", "test_multi_gpu_stub.cpp")
STUB_EXE = os.path.join(ROOT, "tests", "cpp", "test_multi_gpu_stub")


def test_cpp_multi_gpu_host_logic_with_a_stubbed_exchange():
    """BatchedMultiGpuSolverT with stub device classes and an in-process all-reduce (no GPU): shard routing of
    per-element calls over 1 ... 8 ranks, ragged and empty shards, the reduced norms, a failing rank (the others finish,
    every rank still joins the collective, the exception is rethrown after the join) -- the host logic of the N > 1 path,
    which the one-GPU box can only ever run with a single rank."""
    mbuild.build()
    libdir = os.path.join(ROOT, "momentum_amd")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-pthread", "-I", os.path.join(ROOT, "include"), STUB_SRC, "-L", libdir,
                           "-lmmx_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", STUB_EXE])  # fmt: skip
    out = subprocess.run([STUB_EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("OK"), out.stdout


def test_cpp_multi_gpu_program_compiles_and_links():
    _compile_multi()
    assert os.path.exists(MULTI_EXE)


@pytest.mark.gpu
def test_cpp_multi_gpu_solver_on_every_visible_gpu():
    """BatchedMultiGpuSolver: one host thread + one RCCL rank per visible device (one on the test box, eight
    on a full node), ragged shards, per-element parity with the golden fixture, all-reduced residual norms."""
    _compile_multi()
    out = subprocess.run([MULTI_EXE], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("OK"), out.stdout
