#!/usr/bin/env python3
"""Single-frame operator() latency breakdown (host image -> host keypoints): wall time per call next to the
HIP-event spans of the kernels inside it.  Runs on the GPU box."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from visual_sgraphs_amd import orb, synth  # noqa: E402

GW, GH, NF = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (640, 480, 1000)
KIND = sys.argv[4] if len(sys.argv) > 4 else "rectangles"
ex = orb.ORBextractor(NF, 1.2, 8, 20, 7)
img = synth.content_frame(KIND, GW, GH, NF, 1)
for _ in range(20):
    ex(img)
n = 300
t0 = time.perf_counter()
for _ in range(n):
    ex(img)
wall = (time.perf_counter() - t0) / n * 1e3
ex.enable_timing(True)
for _ in range(n):
    ex(img)
print(f"wall per call: {wall:.3f} ms")
print("kernel spans (ms):", {k: round(v, 4) for k, v in ex.timing_ms().items()})
