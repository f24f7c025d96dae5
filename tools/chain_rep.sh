#!/bin/bash
# How often rocprofv3 --kernel-trace brings down tools/_bin/config_chain (see profiles/r05_q_rocprofv3_kernel_trace_c5_segfault.txt):
# N plain runs and N runs under the tool per leg.   Usage (GPU box): bash tools/chain_rep.sh [N]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
N=${1:-6}
mkdir -p gpurun_out/chainrep
for leg in c3 c5; do
  for i in $(seq $N); do tools/_bin/config_chain 2 1 $leg > /dev/null 2> gpurun_out/chainrep/plain_${leg}_$i.err; echo "plain $leg $i rc=$?"; done
  for i in $(seq $N); do
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/chainrep/prof_${leg}_$i -- tools/_bin/config_chain 2 1 $leg > /dev/null 2> gpurun_out/chainrep/prof_${leg}_$i.err
    echo "rocprofv3 --kernel-trace $leg $i rc=$?"; rm -rf gpurun_out/chainrep/prof_${leg}_$i
  done
done
