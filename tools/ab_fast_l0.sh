#!/bin/bash
# TIMING experiment for VERDICT r5 #6a: what the one-frame FAST launch costs without level 0's cells / with them alone
# (libraries built with -DVSG_EXP_FAST_LEVELS=1 / 2: those cells are treated as empty; keypoints are wrong, the launch shape is
# not).  Usage: tools/ab_fast_l0.sh <reps>
for rep in $(seq 1 $1); do
  for v in cur nol0 onlyl0; do
    if [ "$v" = cur ]; then unset VSG_LIB; else export VSG_LIB="$PWD/tools/_bin/libvsg_$v.so"; fi
    echo "$v $(python tools/latency_probe.py 2>&1 | tail -2 | tr '\n' ' ')"
  done
done
