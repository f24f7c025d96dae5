#!/usr/bin/env python3
"""bench.py's content_sweep leg on its own (A/B of FAST changes): C2 geometry, extract + match, 512-frame batches, one line
per content class -- frames/s, the FAST launch's duration, parity, the CPU oracle's rate.
usage: python tools/content_sweep.py [batch=512] [steps=12] [class,class,...]"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import bench  # noqa: E402

if __name__ == "__main__":
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    if len(sys.argv) > 3:  # a subset of the classes (A/B runs)
        from visual_sgraphs_amd import synth
        synth.CONTENT_CLASSES = tuple(c for c in synth.CONTENT_CLASSES if c in sys.argv[3].split(","))
    r = bench.content_sweep_leg(0, batch=batch, steps=steps, cpu_seconds=0.3)
    for k, v in r["classes"].items():
        print(f"{k:12s} {json.dumps(v)}")
    print(json.dumps({k: v for k, v in r.items() if k != "classes"}))
