#!/bin/bash
# Per-kernel average durations of the bench step on a content class (timed-region launch forms), for the in-tree library or a
# variant (VSG_LIB).  Usage on the GPU box: bash tools/kstats_content.sh <content> [variant]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
c=${1:-photo_china}; v=${2:-cur}
if [ "$v" != cur ]; then export VSG_LIB="$PWD/tools/_bin/libvsg_$v.so"; fi
rm -rf gpurun_out/kstats_$v
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats_$v -- python3 bench.py --cpu-seconds 0 --no-stage-timing --no-extras --steps 10 --warmup 2 --ramp-steps 20 --content $c > /dev/null 2> gpurun_out/kstats_$v.err
python3 - <<PY
import csv,glob
for f in glob.glob("gpurun_out/kstats_$v/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) < 1: continue
        n=r["Name"].split("(")[0][-40:]
        print("$c $v %-42s calls %5s avg_us %9.1f pct %5s" % (n, r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
rm -rf gpurun_out/kstats_$v
