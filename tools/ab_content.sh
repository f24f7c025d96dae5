#!/bin/bash
# In-run A/B of library variants (tools/_bin/libvsg_<name>.so, "cur" = the in-tree build) on tools/content_sweep.py:
# the FAST launch's duration and frames/s per content class.  Usage on the GPU box:
#   tools/ab_content.sh "<classes,comma,separated>" <reps> name1 name2 ...
classes="$1"; reps="$2"; shift 2
mkdir -p gpurun_out
for rep in $(seq 1 $reps); do
  for v in "$@"; do
    if [ "$v" = cur ]; then unset VSG_LIB; else export VSG_LIB="$PWD/tools/_bin/libvsg_$v.so"; fi
    python tools/content_sweep.py ${AB_BATCH:-1024} 12 "$classes" 2>>gpurun_out/ab_content.err | python -c "
import json,sys
for line in sys.stdin:
    line=line.strip()
    if not line or line.startswith('{'): continue
    k, rest = line.split(None, 1)
    d = json.loads(rest)
    print('$v', k, 'fast_ms=%s' % d.get('fast_ms'), 'fps=%s' % d.get('frames_per_s'), 'parity=%s' % d.get('parity'))
"
  done
done
unset VSG_LIB
