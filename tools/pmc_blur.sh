#!/bin/bash
# Counters of the blur launch (serialized stages) for library variants: tools/pmc_blur.sh name1 name2 ...   (GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  if [ "$v" = cur ]; then unset VSG_LIB; else export VSG_LIB="$PWD/tools/_bin/libvsg_$v.so"; fi
  rm -rf gpurun_out/pmcb_$v
  B="python3 bench.py --batch ${PMC_BATCH:-512} --cpu-seconds 0 --no-stage-timing --steps 2 --warmup 1 --no-extras"
  VSG_NO_OVERLAP=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d gpurun_out/pmcb_$v/a -- $B > /dev/null 2>>gpurun_out/pmc_blur.err
  VSG_NO_OVERLAP=1 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/pmcb_$v/b -- $B > /dev/null 2>>gpurun_out/pmc_blur.err
  VSG_NO_OVERLAP=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmcb_$v/c -- $B > /dev/null 2>>gpurun_out/pmc_blur.err
  VSG_NO_OVERLAP=1 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/pmcb_$v/d -- $B > /dev/null 2>>gpurun_out/pmc_blur.err
  VSG_NO_OVERLAP=1 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmcb_$v/e -- $B > /dev/null 2>>gpurun_out/pmc_blur.err
  python3 tools/pmc_summary.py gpurun_out/pmcb_$v/a gpurun_out/pmcb_$v/b gpurun_out/pmcb_$v/c gpurun_out/pmcb_$v/d gpurun_out/pmcb_$v/e > gpurun_out/pmc_blur_$v.txt 2>&1
  echo "== $v"; grep -E "kernel|k_blur" gpurun_out/pmc_blur_$v.txt
  rm -rf gpurun_out/pmcb_$v
done
