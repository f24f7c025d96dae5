cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 1 2 3 0; do
  rm -rf gpurun_out/pmcf$v
  VSG_FAST_DBG=$v rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmcf$v -- python3 bench.py --batch 64 --steps 2 --warmup 1 --cpu-seconds 0 --no-stage-timing --no-match > /dev/null 2>&1
  echo "== VSG_FAST_DBG=$v"; python3 tools/pmc_summary.py gpurun_out/pmcf$v | grep -E "kernel|fast"
done
