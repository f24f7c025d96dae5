#!/bin/bash
# In-run A/B of the BASELINE config chains (tools/_bin/config_chain: C3 pair, C5 streams, one-frame latency) over library variants
# (tools/build_variant.sh; "cur" = in-tree).  The binary finds libvsg_orb.so through its RUNPATH, which LD_LIBRARY_PATH precedes.
# Usage: tools/ab_chain.sh <reps> name1 name2 ...
reps="$1"; shift
for v in "$@"; do
  if [ "$v" != cur ]; then mkdir -p tools/_bin/var_$v; cp tools/_bin/libvsg_$v.so tools/_bin/var_$v/libvsg_orb.so; fi
done
for rep in $(seq 1 $reps); do
  for v in "$@"; do
    if [ "$v" = cur ]; then P=""; else P="$PWD/tools/_bin/var_$v"; fi
    LD_LIBRARY_PATH="$P:$LD_LIBRARY_PATH" tools/_bin/config_chain 2>/dev/null | python3 -c "
import json, sys
d = json.load(sys.stdin)
c3, c5, fl = d['C3'], d['C5'], d['frame_latency']
print('$v', 'C3 ms_per_pair', c3['ms_per_pair'], 'extract_2_eyes', c3['fused_stage_ms']['extract_2_eyes'], 'pipelines', c3['pairs_per_s_all_pipelines'],
      '| C5 ms/frame', c5['ms_per_frame_one_stream'], 'fps', c5['frames_per_s'], '| extract_ms', fl['extract_ms'], 'parity', c3['parity'], c5['parity'], fl['parity'])"
  done
done
