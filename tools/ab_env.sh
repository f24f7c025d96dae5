#!/bin/bash
# In-run A/B of ENVIRONMENT settings on tools/content_sweep.py (FAST launch duration + frames/s per content class).
# Usage on the GPU box: tools/ab_env.sh "<classes>" <reps> "VAR=val" "VAR=val2" ...   ("-" = no setting)
classes="$1"; reps="$2"; shift 2
for rep in $(seq 1 $reps); do
  for e in "$@"; do
    if [ "$e" = "-" ]; then pre=""; else pre="$e"; fi
    env $pre python tools/content_sweep.py ${AB_BATCH:-1024} 12 "$classes" 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    line=line.strip()
    if not line or line.startswith('{'): continue
    k, rest = line.split(None, 1)
    d = json.loads(rest)
    print('$e', k, 'fast_ms=%s' % d.get('fast_ms'), 'fps=%s' % d.get('frames_per_s'), 'parity=%s' % d.get('parity'))
"
  done
done
