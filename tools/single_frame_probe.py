"""Wall time of one blocking single-frame operator() (host image -> host keypoints) through the ctypes binding; run it
under `rocprofv3 --kernel-trace --memory-copy-trace` to see the GPU timeline of a call.
Usage on the GPU box: python3 tools/single_frame_probe.py"""
import numpy as np, time, sys
sys.path.insert(0, '.')
from visual_sgraphs_amd import orb, synth
ex = orb.ORBextractor(1000, 1.2, 8, 20, 7)
img = synth.sequence_frame(640, 480, 1000, 0)
for _ in range(20): ex(img)
t=time.perf_counter()
for _ in range(200): ex(img)
print("ms per call", (time.perf_counter()-t)/200*1e3)
