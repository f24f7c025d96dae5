#!/usr/bin/env python3
"""Wall time per host-side matcher call on realistic sizes (C2: 640x480 / 1000 features), three ways:
  resident  -- vsg_frame_* : the frame's keypoints / descriptors / grid live on the device, a call uploads the
               projected positions + map-point descriptors only (what Tracking.cc:2955,3493 would call per frame);
  host      -- the round-1 entry points (host candidate lists + both descriptor sets in, results out);
  oracle    -- the CPU oracle on one host thread.
GPU box.  `python tools/matcher_latency_probe.py [--json]` (bench.py imports `measure()` for its matcher_latency key)."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def _time(fn, n=200, warm=10):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e3


def measure(n=200, with_oracle=True):
    from visual_sgraphs_amd import orb, synth
    import oracle_lib as ol
    out = {}
    ex = orb.ORBextractor(1000, 1.2, 8, 20, 7)
    _, k0, d0 = ex(synth.sequence_frame(640, 480, 3, 0))
    _, k1, d1 = ex(synth.sequence_frame(640, 480, 3, 1))
    bounds = (0.0, 0.0, 640.0, 480.0)
    sf = ex.GetScaleFactors()
    m = orb.ORBmatcher(0.7, True)
    f0 = orb.Frame(ex.capacity(480, 640)).upload(k0, d0, bounds)
    f1 = orb.Frame(ex.capacity(480, 640)).from_extractor(ex, 0, k1, bounds)  # straight out of the extractor
    o0, o1 = ol.OracleFrame(k0, d0, bounds), ol.OracleFrame(k1, d1, bounds)
    nq = len(k0)
    u, v = (k0["x"] - 3).astype(np.float32), (k0["y"] - 2).astype(np.float32)
    obs, blk = np.ones(nq, np.uint8), np.zeros(len(k1), np.uint8)
    oct0, ang0 = k0["octave"].astype(np.int32), k0["angle"].astype(np.float32)

    def row(name, resident=None, host=None, oracle=None):
        r = {}
        if resident:
            r["resident_ms"] = round(_time(resident, n), 4)
        if host:
            r["host_ms"] = round(_time(host, n), 4)
        if oracle and with_oracle:
            r["oracle_1thread_ms"] = round(_time(oracle, max(20, n // 4)), 4)
        out[name] = r

    # --- SearchByProjection(CurrentFrame, LastFrame) (Tracking.cc:2955), th = 15
    rad = (np.float32(15.0) * sf[oct0]).astype(np.float32)
    off, idx = f1.GetFeaturesInArea(u, v, rad, oct0 - 1, oct0 + 1)
    row("SearchByProjection_last_frame",
        lambda: f1.SearchByProjection_Last(d0, obs, u, v, None, oct0, ang0, 15.0, 0, sf, True, blk),
        lambda: m.SearchByProjection_Last(d0, ang0, obs, off, idx, d1, k1["angle"], blk),
        lambda: o1.search_by_projection_last(d0, obs, u, v, None, oct0, ang0, 15.0, 0, sf, True, blk))
    out["SearchByProjection_last_frame"]["queries"] = int(nq)
    out["SearchByProjection_last_frame"]["candidates"] = int(off[-1])
    # --- SearchByProjection(F, local map points) (Tracking.cc:3493), th = 1
    mp = dict(desc=d0, observed=obs, in_view=obs, proj_x=u, proj_y=v, proj_xr=u, scale_level=oct0,
              view_cos=np.full(nq, 0.9, np.float32))
    win = (np.float32(4.0) * sf[oct0]).astype(np.float32)
    offl, idxl = f1.GetFeaturesInArea(u, v, win, oct0 - 1, oct0)
    row("SearchByProjection_local_map",
        lambda: f1.SearchByProjection(mp, 1.0, 0.8, sf, blk),
        lambda: m.SearchByProjection_Local(d0, obs, offl, idxl, d1, k1["octave"], blk),
        lambda: o1.search_by_projection(mp, 1.0, 0.8, sf, blk))
    # --- SearchByProjection(KeyFrame, Sim3) / (Frame, KeyFrame) / SearchBySim3 / Fuse
    rad10 = (np.float32(10.0) * sf[oct0]).astype(np.float32)
    matched = np.full(len(k1), -1, np.int32)
    offk, idxk = f1.GetFeaturesInArea(u, v, rad10)
    row("SearchByProjection_keyframe_sim3",
        lambda: f1.SearchByProjection_Sim3(d0, u, v, rad10, oct0, 1.0, matched),
        lambda: orb.search_window(d0, obs, offk, idxk, d1, blk, 50),
        lambda: o1.search_by_projection_sim3(d0, u, v, rad10, oct0, 1.0, matched))
    row("SearchByProjection_frame_keyframe",
        lambda: f1.SearchByProjection_KF(d0, u, v, rad10, oct0, ang0, 64, True, blk), None,
        lambda: o1.search_by_projection_kf(d0, u, v, rad10, oct0, ang0, 64, True, blk))
    q1 = dict(idx=np.arange(nq, dtype=np.int32), desc=d0, u=u, v=v, radius=rad10, level=oct0)
    n1 = len(k1)
    q2 = dict(idx=np.arange(n1, dtype=np.int32), desc=d1, u=(k1["x"] + 3).astype(np.float32),
              v=(k1["y"] + 2).astype(np.float32), radius=(np.float32(10.0) * sf[k1["octave"]]).astype(np.float32),
              level=k1["octave"].astype(np.int32))
    row("SearchBySim3", lambda: orb.SearchBySim3(f0, f1, q1, q2), None, lambda: ol.search_by_sim3(o0, o1, q1, q2))
    inv2 = ex.GetInverseScaleSigmaSquares()
    rad3 = (np.float32(3.0) * sf[oct0]).astype(np.float32)
    slot, ob, bad = np.full(n1, -1, np.int32), np.ones(nq + n1, np.int32), np.zeros(nq + n1, np.uint8)
    qmp = np.arange(nq, dtype=np.int32)

    def fuse_gpu():
        _, bi, bd = f1.Fuse(d0, u, v, u, rad3, oct0, inv2)
        return orb.fuse_decide(qmp, bi, bd, False, slot, ob, bad)
    row("Fuse", fuse_gpu, None, lambda: o1.fuse(qmp, d0, u, v, u, rad3, oct0, inv2, slot, ob, bad))
    # --- SearchForInitialization (Tracking.cc:2556), windowSize = 100
    kx, ky = k0["x"].astype(np.float32), k0["y"].astype(np.float32)
    offi, idxi = f1.GetFeaturesInArea(kx, ky, np.full(nq, 100.0, np.float32), np.zeros(nq, np.int32),
                                      np.zeros(nq, np.int32))
    lvl0 = k0["octave"] == 0
    offi2 = np.concatenate([[0], np.cumsum(np.where(lvl0, np.diff(offi), 0))]).astype(np.int32)
    idxi2 = np.concatenate([idxi[offi[i]:offi[i + 1]] for i in range(nq) if lvl0[i]] or [np.zeros(0, np.int32)])
    row("SearchForInitialization",
        lambda: f0.SearchForInitialization(f1, kx, ky, 100, 0.9, True),
        lambda: m.SearchForInitialization(d0, ang0, oct0, offi2, idxi2, d1, k1["angle"]),
        lambda: o0.search_for_initialization(o1, kx, ky, 100, 0.9, True))
    # --- grid: build (upload / device) and 1000 windows
    row("frame_upload (keypoints + descriptors + grid)", lambda: f0.upload(k0, d0, bounds))
    row("frame_from_extractor (device to device + grid kernel)", lambda: f1.from_extractor(ex, 0, k1, bounds))
    g = orb.FrameGrid(k1, *bounds)
    og = ol.OracleGrid(k1, *bounds)
    row("GetFeaturesInArea_1000_windows", lambda: f1.GetFeaturesInArea(u, v, rad, oct0 - 1, oct0 + 1),
        lambda: g.GetFeaturesInArea(u, v, rad, oct0 - 1, oct0 + 1),
        lambda: [og.query(u[i], v[i], rad[i], oct0[i] - 1, oct0[i] + 1) for i in range(0, nq, 10)])
    out["GetFeaturesInArea_1000_windows"]["oracle_note"] = "oracle timed on every 10th window (python call overhead)"
    # --- BoW chain
    blob = synth.synthetic_vocabulary(k=10, L=3, seed=4)
    voc, ovoc = orb.ORBVocabulary(blob), ol.OracleVocabulary(blob)
    row("ComputeBoW", lambda: f0.ComputeBoW(voc, 2), lambda: voc.transform(d0, 2), lambda: ovoc.transform(d0, 2))
    t0, t1 = ovoc.transform(d0, 2), ovoc.transform(d1, 2)
    valid = np.ones(nq, np.uint8)
    row("SearchByBoW_KF_F", lambda: f0.SearchByBoW_KF_F(valid, t0["fv"], f1, t1["fv"], 0.7, True),
        lambda: m.SearchByBoW_KF_F(d0, ang0, valid, t0["fv"], d1, k1["angle"], t1["fv"]),
        lambda: ol.search_by_bow_kf_f(d0, ang0, valid, t0["fv"], d1, k1["angle"], t1["fv"], 0.7, True))
    row("block_best2_1006x1006", None, lambda: m.block_best2(d0, d1), lambda: ol.block_best2(d0, d1))
    goff = np.arange(0, nq + 1, 8, dtype=np.int32)
    row("ComputeDistinctiveDescriptors_125x8", None, lambda: orb.ComputeDistinctiveDescriptors(d0[:goff[-1]], goff),
        lambda: ol.distinctive_descriptors(d0[:goff[-1]], goff))
    # --- stereo (C3)
    exl, exr = orb.ORBextractor(1200, 1.2, 8, 20, 7), orb.ORBextractor(1200, 1.2, 8, 20, 7)
    rl, rr = ol.OracleExtractor(1200, 1.2, 8, 20, 7), ol.OracleExtractor(1200, 1.2, 8, 20, 7)
    L_, R_ = synth.sequence_frame(752, 480, 8, 2), synth.sequence_frame(752, 480, 8, 0)
    (_, kl, dl), (_, kr, dr) = exl(L_), exr(R_)
    rl(L_), rr(R_)
    b3 = (0.0, 0.0, 752.0, 480.0)
    fl = orb.Frame(exl.capacity(480, 752)).from_extractor(exl, 0, kl, b3)
    fr = orb.Frame(exr.capacity(480, 752)).from_extractor(exr, 0, kr, b3)
    row("ComputeStereoMatches_1200", lambda: orb.ComputeStereoMatches_resident(exl, 0, exr, 0, fl, fr, 0.11, 47.9),
        lambda: orb.ComputeStereoMatches(exl, 0, exr, 0, kl, dl, kr, dr, 0.11, 47.9),
        lambda: ol.stereo_matches(rl, rr, kl, dl, kr, dr, 0.11, 47.9))
    out["arena_growths_after_warmup"] = orb.thread_arena_growths(0)
    return out


if __name__ == "__main__":
    res = measure()
    if "--json" in sys.argv:
        print(json.dumps(res))
    else:
        for k, v in res.items():
            print(f"{k:55s} {v}")
