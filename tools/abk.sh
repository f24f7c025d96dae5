#!/bin/bash
# In-run A/B of cells per FAST workgroup (VSG_FAST_K; 0 = the launcher's own choice): tools/abk.sh "<bench args>" K1 K2 ...
args="$1"; shift
for k in "$@"; do
  for rep in 1 2; do
    VSG_FAST_K=$k python bench.py --cpu-seconds 0 --no-extras $args 2>>gpurun_out/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
s=d['roofline']['stage_ms']
print('K=$k', 'fps=%.0f'%d['value'], 'parity=%s'%d['parity']['bit_exact_vs_oracle'], ' '.join('%s=%.3f'%(k,v) for k,v in s.items()))
"
  done
done
