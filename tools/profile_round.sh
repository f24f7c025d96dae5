#!/bin/bash
# Round profile: un-profiled bench line, rocprofv3 kernel stats of the SAME command (serialized so kernel durations are
# interference-free), PMC passes -- FETCH_SIZE / WRITE_SIZE for HBM traffic, SQ instruction counts, SQ wave / wait /
# active cycles, L1 (TCP) activity; one counter set per run, --pmc only -- for EVERY kernel of the step, the kernel
# stats of the C-ABI matcher probe, the BASELINE config chains and the single-frame timeline.
# Usage (on the GPU box): bash tools/profile_round.sh <tag>     -> gpurun_out/<tag>/ (copy what is to be kept to profiles/)
set -u
TAG=${1:-rXX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --cpu-seconds 0 --no-stage-timing --no-extras"
VSG_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/bench_rocprof.json 2> $OUT/rocprof.err
# the same command WITHOUT the serialisation: the kernels of the timed region as the bench runs them (the blur's workgroups
# ride in k_octree_blur; the counters below describe the separate-launch forms k_octree / k_blur)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_timed -- $B > $OUT/bench_rocprof_timed.json 2>> $OUT/rocprof.err
P="--steps 3 --warmup 1"
pass() {  # name, counters...
  local name=$1; shift
  VSG_NO_OVERLAP=1 timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -- $B $P > /dev/null 2>>$OUT/rocprof.err
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES
pass wait SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass active SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
# one counter pass of the un-serialised step: the fused k_octree_blur launch the timed region really runs
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/pmc_fused -- $B $P > /dev/null 2>>$OUT/rocprof.err
python3 tools/pmc_summary.py $OUT/pmc_fused > $OUT/pmc_sq_timed_region_fused_octree_blur.txt 2>&1
cp $OUT/stats/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
cp $OUT/stats_timed/*/*kernel_stats.csv $OUT/kernel_stats_timed_region.csv 2>/dev/null
python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.txt 2>&1
python3 tools/pmc_summary.py $OUT/pmc_sq $OUT/pmc_wait $OUT/pmc_active $OUT/pmc_tcp > $OUT/pmc_sq.txt 2>&1
python3 tools/make_traffic_json.py C2/1024 $OUT/traffic.json $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_wait $OUT/pmc_active $OUT/pmc_tcp > /dev/null 2>&1
# ---- the other configurations a scaling curve / the stereo chain are quoted on: C4 = 1280x720 / 2000 in
# 256-frame batches (kernel stats of the timed region and of the serialised step, FETCH / WRITE / SQ / wait passes ->
# traffic.json["C4/256"]), C3 = the stereo pair chain of tools/config_chain.cpp (k_stereo, k_bow_descend, k_search_by_bow)
B4="python3 bench.py --cpu-seconds 0 --no-stage-timing --no-extras --workload C4 --batch 256"
VSG_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4_stats -- $B4 > $OUT/c4_bench_rocprof.json 2>> $OUT/rocprof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4_stats_timed -- $B4 > $OUT/c4_bench_rocprof_timed.json 2>> $OUT/rocprof.err
pass4() {
  local name=$1; shift
  VSG_NO_OVERLAP=1 timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/c4_pmc_$name -- $B4 $P > /dev/null 2>>$OUT/rocprof.err
}
pass4 fetch FETCH_SIZE
pass4 write WRITE_SIZE
pass4 sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES
pass4 wait SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
cp $OUT/c4_stats/*/*kernel_stats.csv $OUT/c4_kernel_stats.csv 2>/dev/null
cp $OUT/c4_stats_timed/*/*kernel_stats.csv $OUT/c4_kernel_stats_timed_region.csv 2>/dev/null
python3 tools/pmc_summary.py $OUT/c4_pmc_fetch $OUT/c4_pmc_write $OUT/c4_pmc_sq $OUT/c4_pmc_wait > $OUT/c4_pmc.txt 2>&1
python3 tools/make_traffic_json.py C4/256 $OUT/traffic.json $OUT/c4_pmc_fetch $OUT/c4_pmc_write $OUT/c4_pmc_sq $OUT/c4_pmc_wait > /dev/null 2>&1
# ---- a real photograph at the headline geometry (round 6): the building photo, ~3.5 x the level-0 candidates of the default
# content, iniThFAST / minThFAST / empty cells inside one frame -> kernel stats of the timed region and of the serialised step,
# SQ counters for FAST and the octree (traffic.json["C2/1024/photo_china"])
BP="python3 bench.py --cpu-seconds 0 --no-stage-timing --no-extras --content photo_china"
VSG_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/photo_stats -- $BP > $OUT/photo_bench_rocprof.json 2>> $OUT/rocprof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/photo_stats_timed -- $BP > $OUT/photo_bench_rocprof_timed.json 2>> $OUT/rocprof.err
passp() {
  local name=$1; shift
  VSG_NO_OVERLAP=1 timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/photo_pmc_$name -- $BP $P > /dev/null 2>>$OUT/rocprof.err
}
passp sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES
passp wait SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
cp $OUT/photo_stats/*/*kernel_stats.csv $OUT/photo_kernel_stats.csv 2>/dev/null
cp $OUT/photo_stats_timed/*/*kernel_stats.csv $OUT/photo_kernel_stats_timed_region.csv 2>/dev/null
python3 tools/pmc_summary.py $OUT/photo_pmc_sq $OUT/photo_pmc_wait > $OUT/photo_pmc.txt 2>&1
python3 tools/make_traffic_json.py C2/1024/photo_china $OUT/traffic.json $OUT/photo_pmc_sq $OUT/photo_pmc_wait > /dev/null 2>&1
# (the C3 leg alone: with the four host threads of the C5 leg in the same process rocprofv3 --kernel-trace itself
# segfaults in 2 of 6 runs -- inside hipStreamSynchronize, below the HIP runtime, in the tool's HSA queue interception;
# un-profiled and --pmc runs of the same binary never do: profiles/r05_q_rocprofv3_kernel_trace_c5_segfault.txt --
# and up to three attempts)
C3="tools/_bin/config_chain 2 1 c3"
for attempt in 1 2 3; do
  rm -rf $OUT/c3_stats
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_stats -- $C3 > $OUT/c3_chain_rocprof.json 2>> $OUT/rocprof.err && break
done
cp $OUT/c3_stats/*/*kernel_stats.csv $OUT/c3_chain_kernel_stats.csv 2>/dev/null
pass3() {
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/c3_pmc_$name -- $C3 > /dev/null 2>>$OUT/rocprof.err
}
pass3 fetch FETCH_SIZE
pass3 write WRITE_SIZE
pass3 sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES
pass3 wait SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
python3 tools/pmc_summary.py $OUT/c3_pmc_fetch $OUT/c3_pmc_write $OUT/c3_pmc_sq $OUT/c3_pmc_wait > $OUT/c3_chain_pmc.txt 2>&1
# the un-profiled bench line LAST among the bench runs, with this build's own counters in place (bench.py only quotes a
# traffic file whose source hash is the hash of the sources it runs on): the copies under profiles/ on this box are scratch
cp $OUT/traffic.json profiles/traffic_r06.json
python3 tools/isa_mix.py > $OUT/isa_mix.json 2>> $OUT/rocprof.err && cp $OUT/isa_mix.json profiles/r06_isa_mix.json
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_abi -- tools/_bin/abi_latency 300 > $OUT/abi_latency.json 2>> $OUT/rocprof.err
cp $OUT/stats_abi/*/*kernel_stats.csv $OUT/abi_kernel_stats.csv 2>/dev/null
tools/_bin/config_chain 2 4 > $OUT/config_chain.json 2>> $OUT/rocprof.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/lt -- tools/_bin/extract_latency 300 > $OUT/extract_latency.json 2>> $OUT/rocprof.err
python3 tools/latency_timeline.py $OUT/lt > $OUT/frame_timeline.txt 2>&1
for e in "" VSG_GRAPH=1; do echo "== ${e:-default}"; env $e tools/_bin/extract_latency 1000; done > $OUT/frame_latency_ab.txt 2>&1
rm -rf $OUT/c4_stats/*/*kernel_trace.csv $OUT/c4_stats_timed/*/*kernel_trace.csv $OUT/c3_stats/*/*kernel_trace.csv $OUT/c4_pmc_* $OUT/c3_pmc_* $OUT/photo_pmc_* $OUT/photo_stats/*/*kernel_trace.csv $OUT/photo_stats_timed/*/*kernel_trace.csv $OUT/stats/*/*kernel_trace.csv $OUT/stats_timed/*/*kernel_trace.csv $OUT/pmc_fused $OUT/stats_abi/*/*kernel_trace.csv $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_wait $OUT/pmc_active $OUT/pmc_tcp $OUT/lt
cat $OUT/bench.json | cut -c1-1200
cat $OUT/pmc_traffic.txt $OUT/pmc_sq.txt $OUT/c4_pmc.txt $OUT/photo_pmc.txt $OUT/c3_chain_pmc.txt
