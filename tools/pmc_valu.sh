#!/bin/bash
# SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_INSTS_LDS / SQ_WAVES per kernel and launch (millions) for library variants
# (tools/_bin/libvsg_<name>.so, "cur" = the in-tree build), one rocprofv3 pass each, 512-frame C2 batches.
# Usage on the GPU box: bash tools/pmc_valu.sh name1 name2 ...     (extra bench args in $BENCH_ARGS)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  if [ "$v" = cur ]; then unset VSG_LIB; else export VSG_LIB="$PWD/tools/_bin/libvsg_$v.so"; fi
  rm -rf gpurun_out/pmcv_$v
  VSG_NO_OVERLAP=1 timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d gpurun_out/pmcv_$v -- \
    python3 bench.py --cpu-seconds 0 --no-stage-timing --steps 2 --warmup 1 --no-extras $BENCH_ARGS > /dev/null 2>>gpurun_out/pmc_valu.err
  echo "== $v"
  python3 tools/pmc_summary.py gpurun_out/pmcv_$v | tee gpurun_out/pmc_valu_$v.txt
  rm -rf gpurun_out/pmcv_$v
done
unset VSG_LIB
