"""Where the oracle's time goes per phase of a solve, for each ISA candidate of oracle.py's build sweep (ORC_PHASE_TIMERS build:
time-stamp-counter cycles per phase, single thread, cfg2 shape, float and double).  Explains the several-fold spread between
the candidates on the GPU box's EPYC 9575F (profiles/r05_cpu_baseline.txt).  python scripts/probes/oracle_phase_probe.py"""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from momentum_amd import humanoid72_landmark_joints as lj, make_humanoid72 as mk  # noqa: E402
from momentum_amd._abi import GnOptions as G  # noqa: E402
from oracle import oracle as o  # noqa: E402
from tests.helpers import make_problem  # noqa: E402

o.build()  # (the default library, for make_problem)
r = mk(seed=12345, variant="p128", unit=0.01)
l = lj(r)
B = 64
cons, th0, _ = make_problem(r, l, l, B, seed=1, perturb=0.3)
names = ["getJacobian", "compact+zero", "JtJ/Jtr", "copy system", "factor+solve+update", "loop"]
for flags in ("-march=x86-64-v3", "-march=x86-64-v4", "-march=native", "-march=native -mtune=generic", "-march=native -mno-avx512f"):
    lib = "/tmp/liborc_phase.so"
    subprocess.check_call(["g++", "-O3", "-std=c++17", "-fPIC", *flags.split(), "-DORC_PHASE_TIMERS", "-shared", "-pthread", "-o", lib,
                           os.path.join(ROOT, "oracle", "mmx_oracle_capi.cpp")])  # fmt: skip
    # a fresh interpreter per candidate: dlopen caches by path
    code = f"""
import sys, time, ctypes as C
sys.path.insert(0, {ROOT!r})
import numpy as np
from oracle import oracle as o
o._lib = C.CDLL({lib!r})
from momentum_amd import humanoid72_landmark_joints as lj, make_humanoid72 as mk
from momentum_amd._abi import GnOptions as G
from tests.helpers import make_problem
r = mk(seed=12345, variant='p128', unit=0.01); l = lj(r)
cons, th0, _ = make_problem(r, l, l, 64, seed=1, perturb=0.3)
for dt in ('f32', 'f64'):
    op = G.make(10, 10, 1.0, 0.05)
    o.solve_batch(r, cons, th0[:4], op, dtype=dt)
    buf = (C.c_ulonglong * 8)()
    o._lib.orc_phase_cycles(buf)
    t = time.perf_counter(); o.solve_batch(r, cons, th0, op, dtype=dt, nthreads=1); t1 = time.perf_counter() - t
    o._lib.orc_phase_cycles(buf); tot = sum(buf)
    print('  %s %7.1f solves/s (%.3f ms / iteration): ' % (dt, len(th0) / t1, 1e3 * t1 / len(th0) / 10) + '  '.join('%s %.1f%%' % (n, 100 * buf[i] / tot) for i, n in enumerate({names!r})))
"""
    print("==", flags, flush=True)
    print(subprocess.run([sys.executable, "-c", code], capture_output=True, text=True).stdout, end="", flush=True)
