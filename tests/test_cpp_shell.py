"""The C++ shell that keeps momentum's class surface (include/momentum_amd/momentum_amd.hpp):
compiles everywhere (CPU check), runs its smoke program on the GPU box."""
import os
import subprocess

import pytest

from momentum_amd import build as mbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_shell.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "test_shell")


def _compile():
    mbuild.build()
    libdir = os.path.join(ROOT, "momentum_amd")
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-L", libdir, "-lmmx_hip",
           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", EXE]  # fmt: skip
    subprocess.check_call(cmd)


def test_cpp_shell_compiles_and_links():
    _compile()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_shell_solves_on_gpu():
    _compile()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("OK")
