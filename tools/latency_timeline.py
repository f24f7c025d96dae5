#!/usr/bin/env python3
"""GPU timeline of the blocking single-frame operator(): from a rocprofv3 --kernel-trace CSV of tools/_bin/extract_latency,
the average start offset, duration and gap of every kernel of a call (the calls are the repeating kernel sequence)."""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0][-28:]))
rows.sort()
first = rows[0][2] if rows else ""
names = []
for r in rows[-400:]:
    if r[2] not in names:
        names.append(r[2])
# a call starts at each occurrence of the first kernel of the chain (the ingest kernel or the pyramid)
starts = [i for i, r in enumerate(rows) if "k_ingest" in r[2]] or [i for i, r in enumerate(rows) if "k_pyramid" in r[2]]
calls = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])][-100:]
n = len(calls[0])
calls = [c for c in calls if len(c) == n]
print("calls analysed", len(calls), "kernels per call", n)
t_prev_end = None
tot_busy = 0.0
for k in range(n):
    st = sum(c[k][0] - c[0][0] for c in calls) / len(calls) / 1e3
    du = sum(c[k][1] - c[k][0] for c in calls) / len(calls) / 1e3
    print("%-30s start %7.1f us  dur %6.1f us" % (calls[0][k][2], st, du))
    tot_busy += du
span = sum(max(x[1] for x in c) - c[0][0] for c in calls) / len(calls) / 1e3
period = sum(b[0][0] - a[0][0] for a, b in zip(calls[:-1], calls[1:])) / max(1, len(calls) - 1) / 1e3
print("GPU span of a call %.1f us, sum of kernel durations %.1f us, call period %.1f us" % (span, tot_busy, period))
