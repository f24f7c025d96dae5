#!/usr/bin/env python3
"""Wall time per host-side matcher call (host arrays in, host arrays out) on realistic sizes.  GPU box."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from visual_sgraphs_amd import orb, synth  # noqa: E402

ex = orb.ORBextractor(1000, 1.2, 8, 20, 7)
_, k0, d0 = ex(synth.sequence_frame(640, 480, 3, 0))
_, k1, d1 = ex(synth.sequence_frame(640, 480, 3, 1))
m = orb.ORBmatcher(0.7, True)


def bench(name, fn, n=100):
    for _ in range(5):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    print(f"{name}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per call")


bench("block_best2 1006x1006", lambda: m.block_best2(d0, d1))
g = orb.FrameGrid(k1, 0.0, 0.0, 640.0, 480.0)
qx, qy = k0["x"] - 3, k0["y"] - 2
qr = np.full(len(k0), 15.0, np.float32)
bench("grid build", lambda: orb.FrameGrid(k1, 0.0, 0.0, 640.0, 480.0))
off, idx = g.GetFeaturesInArea(qx, qy, qr)
bench("grid query 1006 windows", lambda: g.GetFeaturesInArea(qx, qy, qr))
bench("search_window 1006 queries", lambda: orb.search_window(d0, None, off, idx, d1, None, 100))
bench("SearchByProjection_Last", lambda: m.SearchByProjection_Last(d0, k0["angle"], np.ones(len(d0), np.uint8), off, idx,
                                                                    d1, k1["angle"], np.zeros(len(d1), np.uint8)))

# ---- the same calls on the CPU oracle (one thread), and the BoW / initialization searches
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import oracle_lib as ol  # noqa: E402

bench("  oracle block_best2", lambda: ol.block_best2(d0, d1), 20)
bench("  oracle search_window", lambda: ol.search_window(d0, None, off, idx, d1, None, 100))
bench("  oracle SearchByProjection_Last", lambda: ol.search_by_projection_last(
    d0, k0["angle"], np.ones(len(d0), np.uint8), off, idx, d1, k1["angle"], np.zeros(len(d1), np.uint8), 100, True))
blob = synth.synthetic_vocabulary(k=10, L=3, seed=4)
voc, ovoc = orb.ORBVocabulary(blob), ol.OracleVocabulary(blob)
bench("bow transform 1006", lambda: voc.transform(d0, 2))
bench("  oracle bow transform", lambda: ovoc.transform(d0, 2))
t0, t1 = ovoc.transform(d0, 2), ovoc.transform(d1, 2)
valid = np.ones(len(d0), np.uint8)
bench("SearchByBoW KF-F", lambda: m.SearchByBoW_KF_F(d0, k0["angle"], valid, t0["fv"], d1, k1["angle"], t1["fv"]))
bench("  oracle SearchByBoW KF-F", lambda: ol.search_by_bow_kf_f(d0, k0["angle"], valid, t0["fv"], d1, k1["angle"],
                                                                   t1["fv"], 0.7, True))
bench("SearchForInitialization", lambda: m.SearchForInitialization(d0, k0["angle"], k0["octave"], off, idx, d1, k1["angle"]))
bench("  oracle SearchForInitialization", lambda: ol.search_for_initialization(d0, k0["angle"], k0["octave"], off, idx,
                                                                               d1, k1["angle"], 0.7, True))

# ---- stereo and distinctive descriptors
exl, exr = orb.ORBextractor(1200, 1.2, 8, 20, 7), orb.ORBextractor(1200, 1.2, 8, 20, 7)
rl, rr = ol.OracleExtractor(1200, 1.2, 8, 20, 7), ol.OracleExtractor(1200, 1.2, 8, 20, 7)
L_, R_ = synth.sequence_frame(752, 480, 8, 2), synth.sequence_frame(752, 480, 8, 0)
(_, kl, dl), (_, kr, dr) = exl(L_), exr(R_)
rl(L_), rr(R_)
bench("ComputeStereoMatches 1200", lambda: orb.ComputeStereoMatches(exl, 0, exr, 0, kl, dl, kr, dr, 0.11, 47.9))
bench("  oracle ComputeStereoMatches", lambda: ol.stereo_matches(rl, rr, kl, dl, kr, dr, 0.11, 47.9), 20)
goff = np.arange(0, len(d0) + 1, 8, dtype=np.int32)
bench("ComputeDistinctiveDescriptors 125 x 8", lambda: orb.ComputeDistinctiveDescriptors(d0[:goff[-1]], goff))
bench("  oracle ComputeDistinctiveDescriptors", lambda: ol.distinctive_descriptors(d0[:goff[-1]], goff))
