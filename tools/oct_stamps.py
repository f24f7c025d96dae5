#!/usr/bin/env python3
"""Where the one-frame octree spends its time: a diagnostic build (tools/build_oct_stamps.py) stamps
s_memtime at the phase boundaries of DistributeOctTree (tid 0 of every level's workgroup, frame 0); this script runs
blocking single-frame operator() calls through that build and prints the deltas per level.  Shares, not lengths: the
stamps' waits forbid overlaps the real kernel has.
    VSG_LIB=tools/_bin/libvsg_octstamp.so python tools/oct_stamps.py [content class [width height nFeatures]]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
from visual_sgraphs_amd import orb, synth  # noqa: E402

TAGS = {0: "enter", 1: "prefix scan + compaction", 2: "points loaded", 3: "initial nodes", 10: "main pass (labels)", 12: "hist: counting sweep", 13: "hist: coarser counts",
        14: "hist: main pass (counts)", 15: "hist: cell table", 11: "hist: relabel sweep", 40: "  pass: node read", 41: "  pass: child counts read",
        42: "  pass: block scan", 43: "  pass: children written", 44: "  pass: barrier",
        20: "sort keys built", 50: "  sort: one round of ranges",
        21: "partition phase", 22: "stable ranks", 23: "careful pass", 30: "best point per node", 31: "exit"}

kind = sys.argv[1] if len(sys.argv) > 1 else "rectangles"
GW, GH, NF = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (640, 480, 1000)
img = synth.content_frame(kind, GW, GH, NF, 3)
ex = orb.ORBextractor(NF, 1.2, 8, 20, 7)
print(f"{kind} {GW}x{GH} / {NF}")
L = orb.load_library()
L.vsg_debug_oct_stamps.argtypes = [C.c_void_p, C.c_int]
for _ in range(30):
    ex(img)
acc = {}
REPS = 20
for rep in range(REPS):
    L.vsg_debug_oct_stamps(None, 1)
    mono, kps, desc = ex(img)
    buf = np.zeros(8 * 32 * 2, np.uint64)
    L.vsg_debug_oct_stamps(buf.ctypes.data_as(C.c_void_p), 0)
    st = buf.reshape(8, 32, 2)
    for lv in range(8):
        rows = [(int(t), int(tag)) for t, tag in st[lv] if t]
        ends = [i for i, (_, tag) in enumerate(rows) if tag == 31]
        if ends:
            rows = rows[:ends[0] + 1]  # the slots behind a call's last stamp hold an earlier call's
        seq = []
        for (t0, _), (t1, tag) in zip(rows, rows[1:]):
            seq.append((tag, t1 - t0))
        acc.setdefault(lv, []).append(seq)
for lv in range(8):
    runs = acc[lv]
    n = min(len(r) for r in runs)
    print(f"level {lv}: {len(kps[kps['octave'] == lv])} keypoints")
    tot = 0
    for i in range(n):
        d = np.median([r[i][1] for r in runs])
        tot += d
        print(f"   -> {TAGS.get(runs[0][i][0], runs[0][i][0]):24s} {d:9.0f} cycles")
    print(f"   total {tot:9.0f} cycles")
