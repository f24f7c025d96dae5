#!/bin/bash
# Copy what tools/profile_round.sh <tag> left under gpurun_out/<tag>/ into profiles/ under the names the index uses.
# Usage: tools/copy_profile.sh <tag> [gpu test log]
set -e
T=$1; O=gpurun_out/$T; P=profiles/$T
cd "$(dirname "$0")/.."
R=r06  # the round whose traffic / ISA-mix files bench.py reads (bench.py: TRAFFIC_FILE, ISA_MIX_FILE)
# (an `[ -f x ] && cp` body returns 1 for an absent file and `set -e` then ends the script: ADVICE r5)
c() { if [ -f "$O/$1" ]; then cp "$O/$1" "${P}_$2"; fi; }
c bench.json bench.json; c bench_rocprof.json bench_under_rocprof_serialized.json
c kernel_stats.csv kernel_stats_serialized.csv; c kernel_stats_timed_region.csv kernel_stats_timed_region.csv
c pmc_traffic.txt pmc_traffic_fetch_write_all_kernels.txt; c pmc_sq.txt pmc_sq_wait_l1_all_kernels.txt
c pmc_sq_timed_region_fused_octree_blur.txt pmc_sq_timed_region_fused_octree_blur.txt
c c4_bench_rocprof_timed.json c4_bench_under_rocprof_timed_region.json; c c4_kernel_stats.csv c4_kernel_stats_serialized.csv
c c4_kernel_stats_timed_region.csv c4_kernel_stats_timed_region.csv; c c4_pmc.txt c4_pmc_fetch_write_sq_wait.txt
c c3_chain_kernel_stats.csv c3_chain_kernel_stats.csv; c c3_chain_pmc.txt c3_chain_pmc.txt; c c3_chain_rocprof.json c3_chain_under_rocprof.json
c config_chain.json config_chain.json; c extract_latency.json extract_latency.json; c frame_latency_ab.txt frame_latency_ab.txt
c photo_kernel_stats_timed_region.csv photo_china_kernel_stats_timed_region.csv; c photo_kernel_stats.csv photo_china_kernel_stats_serialized.csv
c photo_pmc.txt photo_china_pmc_sq_wait.txt; c photo_bench_rocprof_timed.json photo_china_bench_under_rocprof_timed_region.json
c frame_timeline.txt frame_timeline.txt; c abi_latency.json abi_latency.json; c abi_kernel_stats.csv abi_kernel_stats.csv
# merge: keys of earlier runs of the SAME sources (other batch sizes) stay, this run's keys replace their namesakes
python3 - $O/traffic.json profiles/traffic_$R.json <<'PY'
import json, sys
new = json.load(open(sys.argv[1]))
try:
    old = json.load(open(sys.argv[2]))
except Exception:
    old = {}
if old.get('source_hash') == new.get('source_hash'):
    for k, v in old.items():
        new.setdefault(k, v)
json.dump(new, open(sys.argv[2], 'w'), indent=1)
PY
cp $O/isa_mix.json profiles/${R}_isa_mix.json
if [ -n "${2:-}" ]; then grep -E "passed|failed" "$2" > ${P}_gpu_tests.txt || true; fi
python3 -c "import json;print('traffic source hash', json.load(open('profiles/traffic_$R.json'))['source_hash'])"; python3 tools/source_hash.py
