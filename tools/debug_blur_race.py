#!/usr/bin/env python3
"""Repeat a resident 512-frame C2 batch and compare the blurred levels of sampled frames with the oracle's GaussianBlur of the
GPU's own pyramid levels: prints where (level, frame, rows, columns) they differ.  A debugging aid for the matrix-core blur."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
from visual_sgraphs_amd import orb, synth
import oracle_lib as ol

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
frames = np.stack([synth.frame(640, 480, i % 16) for i in range(B)])
dev = torch.device("cuda:0")
d_in = torch.from_numpy(frames).to(dev)
ex = orb.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
cap = ex.capacity(480, 640)
d_kps = torch.zeros((B, cap, 28), dtype=torch.uint8, device=dev)
d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
d_counts = torch.zeros((B, 2), dtype=torch.int32, device=dev)
bad = 0
for rep in range(reps):
    ex.extract_batch_device(d_in.data_ptr(), B, 640 * 480, 480, 640, 640, d_kps.data_ptr(), d_desc.data_ptr(), d_counts.data_ptr(), cap)
    torch.cuda.synchronize()
    for f in sorted(set([0, 1, B // 2, B - 2, B - 1] + list(np.random.default_rng(rep).integers(0, B, 6)))):
        for l in range(8):
            got = ex.blurred_level(l, frame=int(f))
            src = frames[f] if l == 0 else ex.image_pyramid(l, frame=int(f))
            want = ol.gaussian_blur7(np.ascontiguousarray(src))
            if not np.array_equal(got, want):
                ys, xs = np.nonzero(got != want)
                bad += 1
                print(f"rep {rep} frame {f} level {l}: {len(ys)} px differ, rows {ys.min()}..{ys.max()} cols {xs.min()}..{xs.max()}; "
                      f"row set {sorted(set(ys))[:12]} col%32 set {sorted(set(xs % 32))[:12]} strips {sorted(set(xs // 32))[:8]}", flush=True)
print("mismatching (frame, level) pairs:", bad)
