#!/bin/bash
# PMC passes (counters only, no tracing) over an arbitrary bench.py command line; prints per-kernel averages.
# usage: pmc_kernel.sh <outdir-name> <bench args...>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --output-format csv -d $O -o pass$i -- python $R/bench.py "$@" > $O/pass$i.log 2>&1
done
python $R/scripts/pmc_summary.py $O/*_counter_collection.csv > $O/summary.txt 2>&1
cat $O/summary.txt
