#!/bin/bash
# Counters of the C3 stereo pair chain (tools/config_chain.cpp: k_stereo, k_bow_descend, k_search_by_bow, k_window_search,
# k_frame_grid_build + the one-frame extractor chain) per call: FETCH_SIZE / WRITE_SIZE / SQ instruction / SQ wait passes, one
# counter set per run.  Usage on the GPU box: bash tools/pmc_c3_chain.sh <out dir>
OUT=${1:-gpurun_out/c3_pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
C3="tools/_bin/config_chain 2 1"
pass() { local name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -- $C3 > /dev/null 2>>$OUT/rocprof.err; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES
pass wait SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_wait > $OUT/c3_chain_pmc.txt 2>&1
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_wait
cat $OUT/c3_chain_pmc.txt
