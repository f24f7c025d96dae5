#!/bin/bash
# Round profile: un-profiled bench line, rocprofv3 kernel stats of the SAME command (serialized so kernel
# durations are interference-free), PMC passes (FETCH_SIZE / WRITE_SIZE / SQ instruction counts, one counter set per
# run, --kernel-trace only) for HBM traffic, and the kernel stats of the C-ABI matcher probe.
# Usage (on the GPU box): bash tools/profile_round.sh <tag>
set -u
TAG=${1:-rXX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
B="python3 bench.py --cpu-seconds 0 --no-stage-timing --no-extras"
VSG_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/bench_rocprof.json 2> $OUT/rocprof.err
VSG_NO_OVERLAP=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B --steps 3 --warmup 1 > /dev/null 2>&1
VSG_NO_OVERLAP=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B --steps 3 --warmup 1 > /dev/null 2>&1
VSG_NO_OVERLAP=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/pmc_sq -- $B --steps 3 --warmup 1 > /dev/null 2>&1
cp $OUT/stats/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.txt 2>&1
python3 tools/pmc_summary.py $OUT/pmc_sq > $OUT/pmc_sq.txt 2>&1
python3 tools/make_traffic_json.py C2/512 $OUT/traffic.json $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_abi -- tools/_bin/abi_latency 300 > $OUT/abi_latency.json 2>> $OUT/rocprof.err
cp $OUT/stats_abi/*/*kernel_stats.csv $OUT/abi_kernel_stats.csv 2>/dev/null
rm -rf $OUT/stats/*/*kernel_trace.csv $OUT/stats_abi/*/*kernel_trace.csv $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
cat $OUT/bench.json | cut -c1-1500
cat $OUT/pmc_traffic.txt $OUT/pmc_sq.txt
cat $OUT/traffic.json | head -50
