#!/bin/bash
# Register / spill / scratch / occupancy figures of the kernels the bench times, from the compiler
# (-Rpass-analysis=kernel-resource-usage), as a table: bash scripts/resource_usage.sh > profiles/rNN_kernel_resource_usage.txt
cd "$(dirname "$0")/../momentum_amd/csrc" || exit 1
tmp=$(mktemp -d)
# (the flags of momentum_amd/build.py: the solve kernels' groups without machine-level loop-invariant code motion / loop strength reduction)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -disable-machine-licm -mllvm -disable-lsr -DMMX_FUSED_GROUP=1 -c mmx_fused.hip -o $tmp/g1.o -Rpass-analysis=kernel-resource-usage 2> $tmp/g1.txt &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMMX_FUSED_GROUP=4 -c mmx_fused.hip -o $tmp/g0.o -Rpass-analysis=kernel-resource-usage 2> $tmp/g0.txt &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c mmx_kernels.hip -o $tmp/k.o -Rpass-analysis=kernel-resource-usage 2> $tmp/k.txt &
wait
python3 - $tmp <<'PY'
import re,sys,subprocess,glob
rows=[]
for f in sorted(glob.glob(sys.argv[1]+"/*.txt")):
    cur=None
    for line in open(f):
        m=re.search(r"remark: Function Name: (\S+)", line)
        if m: cur={"name":m.group(1)}; rows.append(cur); continue
        m=re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur is not None: cur[m.group(1)]=int(m.group(2))
names=[r["name"] for r in rows]
dem=subprocess.run([__import__("shutil").which("c++filt") or "c++filt"],input="\n".join(names),capture_output=True,text=True).stdout.split("\n")
want=("fusedSolveKernel<6, 0,","fusedSolveKernel<6, 2,","fusedSolveKernel<4, 0, false, false, 0","treeNormalEquationsKernel","treeRefineKernel","fkJacobianKernel<true","choleskyFactorTiledKernel","choleskyFactorResidentKernel","choleskyFinishTiledKernel","choleskyStepKernel","choleskyStepTiledKernel","stepUpdateKernel","normalEquationsMfmaKernel","trustDecideKernel")
print("kernel resource usage (hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage)")
print(f"{'kernel':<78} {'VGPR':>5} {'SGPR':>5} {'vspill':>6} {'sspill':>6} {'scratch B':>9} {'waves/SIMD':>10}")
for r,d in zip(rows,dem):
    short=re.sub(r"\(.*","",d).replace("void mmx::","")
    if any(w in short for w in want):
        print(f"{short[:78]:<78} {r.get('VGPRs',0):>5} {r.get('TotalSGPRs',0):>5} {r.get('VGPRs Spill',0):>6} {r.get('SGPRs Spill',0):>6} {r.get('ScratchSize [bytes/lane]',0):>9} {r.get('Occupancy [waves/SIMD]',0):>10}")
PY
rm -rf $tmp
