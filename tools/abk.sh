#!/bin/bash
# usage: abk.sh K1 K2 ... ; runs bench with VSG_FAST_K
for k in "$@"; do
  for rep in 1 2; do
    VSG_FAST_K=$k python bench.py --cpu-seconds 0 --no-extras 2>>gpurun_out/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
s=d['roofline']['stage_ms']
print('K=$k', 'fps=%.0f'%d['value'], 'parity=%s'%d['parity']['bit_exact_vs_oracle'], ' '.join('%s=%.3f'%(k,v) for k,v in s.items()))
"
  done
done
