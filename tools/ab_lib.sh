#!/bin/bash
# In-run A/B of library variants built into tools/_bin/libvsg_<name>.so (box-to-box timing varies ~25 %, so variants
# must be compared inside ONE gpurun call).  The variant is loaded through the VSG_LIB override of
# visual_sgraphs_amd/orb.py: the in-tree library is never touched.  "cur" = the in-tree build.
# Usage on the GPU box: tools/ab_lib.sh "<bench args>" name1 name2 ...
args="$1"; shift
mkdir -p gpurun_out
for v in "$@"; do
  if [ "$v" = cur ]; then unset VSG_LIB; else export VSG_LIB="$PWD/tools/_bin/libvsg_$v.so"; fi
  for rep in 1 2; do
    python bench.py --cpu-seconds 0 --no-extras $args 2>>gpurun_out/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
s=d['roofline']['stage_ms']
print('$v', 'fps=%.0f'%d['value'], 'parity=%s'%d['parity']['bit_exact_vs_oracle'], ' '.join('%s=%.3f'%(k,v) for k,v in s.items()))
" | tee -a gpurun_out/ab.txt
  done
done
unset VSG_LIB
