#!/usr/bin/env python3
"""Single-frame operator() latency breakdown (host image -> host keypoints): wall time per call next to the
HIP-event spans of the kernels inside it.  Runs on the GPU box."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from visual_sgraphs_amd import orb, synth  # noqa: E402

ex = orb.ORBextractor(1000, 1.2, 8, 20, 7)
img = synth.frame(640, 480, 1)
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
