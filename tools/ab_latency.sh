#!/bin/bash
# In-run A/B of the one-frame latency path (tools/latency_probe.py: blocking single-frame operator(), wall per call and the
# HIP-event spans of its kernels) over library variants.  Usage: tools/ab_latency.sh <reps> name1 name2 ...   ("cur" = in-tree)
reps="$1"; shift
for rep in $(seq 1 $reps); do
  for v in "$@"; do
    if [ "$v" = cur ]; then unset VSG_LIB; else export VSG_LIB="$PWD/tools/_bin/libvsg_$v.so"; fi
    echo "$v $(python tools/latency_probe.py 2>/dev/null | tr '\n' ' ')"
  done
done
unset VSG_LIB
