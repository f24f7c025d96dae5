#!/usr/bin/env python3
"""Kernel-time probe of SearchByBoW on two resident frames with resident FeatureVectors (k = 10, L = 6 vocabulary, levelsup 4:
~100 nodes of ~12 features).  Run under `rocprofv3 --kernel-trace --stats`; VSG_LIB selects a library variant.
usage: python tools/bow_search_probe.py [reps=300]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib as ol  # noqa: E402
from visual_sgraphs_amd import orb, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
voc = orb.ORBVocabulary(synth.synthetic_vocabulary(10, 6, seed=7))
e = ol.OracleExtractor(1200, 1.2, 8, 20, 7)
_, k0, d0 = e(synth.sequence_frame(752, 480, 13, 0))
d1 = d0.copy()
rng = np.random.default_rng(5)
flip = rng.integers(0, 256, (len(d1), 6))
for j in range(flip.shape[1]):
    d1[np.arange(len(d1)), flip[:, j] >> 3] ^= (1 << (flip[:, j] & 7)).astype(np.uint8)
b = (0.0, 0.0, 752.0, 480.0)
f0, f1 = orb.Frame(1300).upload(k0, d0, b), orb.Frame(1300).upload(k0, d1, b)
fv0, fv1 = f0.ComputeBoW(voc, 4)["fv"], f1.ComputeBoW(voc, 4)["fv"]
valid = np.ones(len(k0), np.uint8)
want = f0.SearchByBoW_KF_F(valid, fv0, f1, fv1, 0.7, True)
for name, a, c in (("resident", None, None), ("host_fv", fv0, fv1)):
    t0 = time.perf_counter()
    for _ in range(reps):
        got = f0.SearchByBoW_KF_F(valid, a, f1, c, 0.7, True)
    dt = (time.perf_counter() - t0) / reps
    print(f"{name}: {dt * 1e6:.1f} us per call, matches {got[0]}, same as first {got[0] == want[0] and np.array_equal(got[1], want[1])}")
t0 = time.perf_counter()
for _ in range(reps):
    bow = f0.ComputeBoW(voc, 4)
dt = (time.perf_counter() - t0) / reps
ref = ol.OracleVocabulary(synth.synthetic_vocabulary(10, 6, seed=7)).transform(d0, 4)
same = np.array_equal(bow["bow_vals"].view(np.uint64), ref["bow_vals"].view(np.uint64)) and all(
    np.array_equal(x, y) for x, y in zip(bow["fv"], ref["fv"]))
print(f"ComputeBoW (resident frame, {len(d0)} features): {dt * 1e6:.1f} us per call, bit-identical to the oracle: {same}")
