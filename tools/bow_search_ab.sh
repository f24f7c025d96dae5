cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06_c
for v in ${@:-cur}; do
  if [ "$v" = cur ]; then unset VSG_LIB; else export VSG_LIB="$PWD/tools/_bin/libvsg_$v.so"; fi
  rm -rf gpurun_out/r06_c/st_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_c/st_$v -- python3 tools/bow_search_probe.py 300 > gpurun_out/r06_c/probe_$v.txt 2>&1
  echo "== $v"; grep "per call" gpurun_out/r06_c/probe_$v.txt
  python3 - <<PY
import csv,glob
for f in glob.glob("gpurun_out/r06_c/st_$v/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "bow" in r["Name"]: print(r["Name"].split("(")[1][-30:] if r["Name"].startswith("(") else r["Name"][:40], "calls", r["Calls"], "avg_us %.1f min %.1f max %.1f" % (float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
  rm -rf gpurun_out/r06_c/st_$v
done
