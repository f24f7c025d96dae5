#!/bin/bash
# PMC A/B of library variants (tools/_bin/libvsg_<name>.so, or "cur" = the in-tree build): SQ instruction mix and LDS
# counters per kernel, one counter set per pass.  Usage on the GPU box: tools/pmc_ab.sh name1 name2 ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  if [ "$v" = cur ]; then unset VSG_LIB; else export VSG_LIB="$PWD/tools/_bin/libvsg_$v.so"; fi  # never overwrite the in-tree library
  rm -rf gpurun_out/pmc_$v
  B="python3 bench.py --batch ${PMC_BATCH:-64} --cpu-seconds 0 --no-stage-timing --steps 2 --warmup 1 --no-extras"
  VSG_NO_OVERLAP=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d gpurun_out/pmc_$v/a -- $B > /dev/null 2>>gpurun_out/pmc_ab.err
  VSG_NO_OVERLAP=1 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc_$v/b -- $B > /dev/null 2>&1
  VSG_NO_OVERLAP=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_$v/c -- $B > /dev/null 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmc_$v/a gpurun_out/pmc_$v/b gpurun_out/pmc_$v/c > gpurun_out/pmc_ab_$v.txt 2>&1
  echo "== $v"; cat gpurun_out/pmc_ab_$v.txt
  rm -rf gpurun_out/pmc_$v
done
