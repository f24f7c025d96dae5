#!/bin/bash
# Memory-side PMC passes of library variants (tools/_bin/libvsg_<name>.so, or "cur" = the in-tree build): vector-memory
# instruction cycles, L1 (TCP) / TA / TD / L2 (TCC) activity and stalls per kernel, one counter set per pass
# (--pmc only, no trace domains; the TA_* / TD_* sets hang rocprofv3 on this pool and are left out).  Values are averages per launch in millions.
# Usage on the GPU box: tools/pmc_mem.sh name1 name2 ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SETS=(
 "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES"
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
 "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum"
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
)
for v in "$@"; do
  if [ "$v" = cur ]; then unset VSG_LIB; else export VSG_LIB="$PWD/tools/_bin/libvsg_$v.so"; fi
  rm -rf gpurun_out/pmcm_$v
  B="python3 bench.py --cpu-seconds 0 --no-stage-timing --steps 2 --warmup 1 --no-extras"
  dirs=""
  i=0
  for s in "${SETS[@]}"; do
    VSG_NO_OVERLAP=1 timeout 150 rocprofv3 --pmc $s --output-format csv -d gpurun_out/pmcm_$v/$i -- $B > /dev/null 2>>gpurun_out/pmc_mem.err
    dirs="$dirs gpurun_out/pmcm_$v/$i"
    i=$((i+1))
  done
  python3 tools/pmc_summary.py $dirs > gpurun_out/pmc_mem_$v.txt 2>&1
  echo "== $v"; cat gpurun_out/pmc_mem_$v.txt
  rm -rf gpurun_out/pmcm_$v
done
