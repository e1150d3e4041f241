#!/bin/bash
cd "$GRAFT_REPO_ROOT/oracle" || exit 1
lscpu | grep -E "Model name|MHz|^CPU\(s\)|Thread|L3" | head -8
gcc --version | head -1
echo "cpufreq:"; cat /sys/devices/system/cpu/cpu0/cpufreq/scaling_cur_freq 2>/dev/null; grep MHz /proc/cpuinfo | sort -t: -k2 -n | tail -2
nproc; cat /sys/fs/cgroup/cpu.max
t() { # lib
python - "$1" <<'PY'
import sys,time,ctypes,os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from oracle import oracle as o
o._lib = ctypes.CDLL(sys.argv[1])
from momentum_amd import humanoid72_landmark_joints as lj, make_humanoid72 as mk
from momentum_amd._abi import GnOptions as G
r=mk(seed=12345,variant='p128',unit=0.01);l=lj(r);rng=np.random.default_rng(0);B=64
z=np.zeros
c=o.Constraints(l,z((B,16,3)),rng.uniform(-1,1,(B,16,3)),np.ones((B,16)),l,np.tile([0,0,0,1.],(B,16,1)),np.tile([0,0,0,1.],(B,16,1)),np.ones((B,16)))
op=G.make(10,10);th=z((B,r.num_params),np.float32);o.solve_batch(r,c,th[:4],op,dtype='f32')
ts=[]
for _ in range(3):
    t=time.perf_counter();o.solve_batch(r,c,th,op,dtype='f32');ts.append(time.perf_counter()-t)
print(sys.argv[1].split('/')[-1], "%.0f solves/s/thread (best of 3), all: %s" % (B/min(ts), ["%.0f"%(B/x) for x in ts]))
PY
}
t $PWD/libmmx_oracle_here_v3.so
for fl in "-march=x86-64-v3" "-march=x86-64-v3 -mtune=generic" "-march=native" "-march=znver3" "-march=znver4"; do
  g++ -O3 $fl -std=c++17 -fPIC -shared -pthread -o /tmp/o.so mmx_oracle_capi.cpp 2>/dev/null && { echo "box build $fl:"; t /tmp/o.so; } || echo "box build $fl: compile failed"
done
t $PWD/libmmx_oracle_here_v3.so
