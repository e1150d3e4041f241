#!/bin/bash
# every BASELINE config through bench.py on one box (no CPU baseline): one JSON line each
for c in cfg2 cfg3 cfg2_all cfg5; do
  python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']
    print('$c', 'B=%d'%d['config']['batch_per_gpu'], '%.4g solves/s'%d['value'], '%.3f ms/step'%d['ms_per_step'], 'J-assembly %.1f us %.0f GB/s frac %.3f'%(r['ms_per_launch']*1e3, r['achieved'], r['frac']))
"
done
MMX_SOLVER=v1 python bench.py --config cfg2 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('cfg2 via explicit-J path', '%.4g solves/s'%d['value'])
"
