#!/bin/bash
# Issue-stall breakdown of the blur launch: tools/pmc_blur2.sh name1 ...   (GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  if [ "$v" = cur ]; then unset VSG_LIB; else export VSG_LIB="$PWD/tools/_bin/libvsg_$v.so"; fi
  rm -rf gpurun_out/pmcb_$v
  B="python3 bench.py --batch ${PMC_BATCH:-512} --cpu-seconds 0 --no-stage-timing --steps 2 --warmup 1 --no-extras"
  VSG_NO_OVERLAP=1 rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmcb_$v/a -- $B > /dev/null 2>>gpurun_out/pmc_blur.err
  VSG_NO_OVERLAP=1 rocprofv3 --pmc SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --output-format csv -d gpurun_out/pmcb_$v/b -- $B > /dev/null 2>>gpurun_out/pmc_blur.err
  VSG_NO_OVERLAP=1 rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_LDS_UNALIGNED_STALL SQ_IFETCH --output-format csv -d gpurun_out/pmcb_$v/c -- $B > /dev/null 2>>gpurun_out/pmc_blur.err
  VSG_NO_OVERLAP=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA --output-format csv -d gpurun_out/pmcb_$v/d -- $B > /dev/null 2>>gpurun_out/pmc_blur.err
  python3 tools/pmc_summary.py gpurun_out/pmcb_$v/a gpurun_out/pmcb_$v/b gpurun_out/pmcb_$v/c gpurun_out/pmcb_$v/d > gpurun_out/pmc_blur2_$v.txt 2>&1
  echo "== $v"; grep -E "kernel|k_blur" gpurun_out/pmc_blur2_$v.txt
  rm -rf gpurun_out/pmcb_$v
done
