#!/bin/bash
# round 5, call L: the three-workgroup instantiations with their LDS copy of the CSR back: the whole suite + the lines they serve
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05l; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=line < /dev/null 2>&1 | tail -8 > $out/pytest_gpu.txt; tail -8 $out/pytest_gpu.txt
for a in "" "--line-search 2" "--config cfg2_tracker" "--config cfg3 --batch 65536 --steps 4"; do
  timeout 300 python bench.py $a --no-extra-configs --no-cpu-baseline --check-instances 256 < /dev/null 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$a', '%.4g' % d['value'], d['check'].get('max_rel'))" | tee -a $out/lines.txt
done
