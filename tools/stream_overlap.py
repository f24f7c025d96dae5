#!/usr/bin/env python3
"""Kernel trace (rocprofv3 --kernel-trace csv) of concurrent streams -> per kernel name: count, mean duration, and how much
of the traced time had 1, 2, 3, 4+ kernels running at once; per stream (queue): busy fraction and mean gap between
consecutive kernels.   python tools/stream_overlap.py <dir> [t_from_frac t_to_frac]"""
import collections, csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0][-40:], r.get("Queue_Id", r.get("Stream_Id", "?"))))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
a, b = t0 + lo * (t1 - t0), t0 + hi * (t1 - t0)
rows = [r for r in rows if r[0] >= a and r[1] <= b]
print("window %.1f ms, %d kernels" % ((b - a) / 1e6, len(rows)))
ev = []
for s, e, n, q in rows:
    ev.append((s, 1)), ev.append((e, -1))
ev.sort()
depth, last, hist = 0, ev[0][0], collections.Counter()
for t, d in ev:
    hist[min(depth, 5)] += t - last
    depth += d
    last = t
tot = sum(hist.values())
print("concurrency: " + "  ".join("%d: %.1f %%" % (k, 100.0 * v / tot) for k, v in sorted(hist.items())))
by = collections.defaultdict(list)
for s, e, n, q in rows:
    by[n].append(e - s)
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print("  %-42s n %6d  mean %7.1f us  total %8.1f ms" % (n, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
byq = collections.defaultdict(list)
for s, e, n, q in rows:
    byq[q].append((s, e))
for q, v in sorted(byq.items()):
    busy = sum(e - s for s, e in v)
    gaps = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
    print("  queue %s: %d kernels, busy %.1f %% of the window, mean gap %.1f us" % (q, len(v), 100.0 * busy / (b - a), sum(gaps) / max(len(gaps), 1) / 1e3))
