#!/bin/bash
# PMC passes over the fused solve kernel (counters only, no tracing): instruction mix and issue utilisation.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_fused
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_F32" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --output-format csv -d $O -o pass$i -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $O/pass$i.log 2>&1
  tail -2 $O/pass$i.log | cut -c1-200
done
python $R/scripts/pmc_summary.py $O/*_counter_collection.csv > $O/summary.txt 2>&1
cat $O/summary.txt
