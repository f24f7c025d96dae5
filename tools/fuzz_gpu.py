#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity sweep (runs on the GPU box; not part of the pytest suites because it is open
ended).  Every case draws an entry point and random, reference-processable parameters; any mismatch prints the
case and exits 1.

    python tools/fuzz_gpu.py --cases 300 --seed 1
"""
import argparse
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import oracle_lib as ol  # noqa: E402
from visual_sgraphs_amd import orb, synth  # noqa: E402


def same(got, want):
    return got[0] == want[0] and got[1].tobytes() == want[1].tobytes() and np.array_equal(got[2], want[2])


def geometry(rng):
    while True:
        w, h = int(rng.integers(150, 1000)), int(rng.integers(120, 760))
        nl = int(rng.integers(1, 9))
        sc = float(rng.choice([1.1, 1.2, 1.25, 1.3, 1.5, 2.0, 2.5]))
        nf = int(rng.integers(50, 2400))
        top = sc ** (nl - 1)
        if w / top < 80 or h / top < 80 or w / h > 3.5 or w < 0.6 * h:
            continue
        ini = int(rng.integers(8, 60))
        return w, h, nf, sc, nl, ini, int(rng.integers(2, ini + 1))


def case_batch(rng):
    w, h, nf, sc, nl, ini, mn = geometry(rng)
    B = int(rng.integers(1, 7))
    lap = (int(rng.integers(-10, w)), int(rng.integers(-10, w + 50)))
    # half of the cases: the default rectangles + noise frames; the other half: a random content class per frame
    # (synth.CONTENT_CLASSES: value noise, checkerboards, gratings, defocus, saturation, ramps, salt and pepper)
    kinds = [str(rng.choice(synth.CONTENT_CLASSES)) if rng.random() < 0.5 else "" for _ in range(B)]
    imgs = np.stack([synth.content_frame(k, w, h, int(rng.integers(0, 1 << 20)), int(rng.integers(0, 40))) if k else
                     synth.frame(w, h, int(rng.integers(0, 1 << 20)), amplitude_div=int(rng.choice([1, 1, 4])))
                     for k in kinds])
    ex = orb.ORBextractor(nf, sc, nl, ini, mn, max_batch=B)
    ref = ol.OracleExtractor(nf, sc, nl, ini, mn)
    outs = ex.extract_batch(imgs, lap)
    return all(same(outs[i], ref(imgs[i], lap)) for i in range(B)), ("batch", w, h, nf, sc, nl, ini, mn, B, lap, kinds)


def case_colour(rng):
    w, h, nf, sc, nl, ini, mn = geometry(rng)
    ch, rgb = int(rng.choice([3, 4])), bool(rng.integers(0, 2))
    base = synth.frame(w, h, int(rng.integers(0, 1 << 20)))
    col = np.stack([np.clip(base.astype(np.int32) + rng.integers(-25, 26, base.shape), 0, 255).astype(np.uint8)
                    for _ in range(ch)], axis=-1)
    gray = ol.cvt_gray(col, rgb)
    ex = orb.ORBextractor(nf, sc, nl, ini, mn, max_batch=1)
    got = ex.extract_batch_color(col[None], rgb)[0]
    return same(got, ol.OracleExtractor(nf, sc, nl, ini, mn)(gray)), ("colour", w, h, nf, sc, nl, ini, mn, ch, rgb)


def case_best2(rng):
    na, nb = int(rng.integers(0, 700)), int(rng.integers(0, 700))
    base = rng.integers(0, 256, (int(rng.integers(1, 40)), 32), dtype=np.uint8)
    a = base[rng.integers(0, len(base), na)] if na else np.zeros((0, 32), np.uint8)
    b = base[rng.integers(0, len(base), nb)] if nb else np.zeros((0, 32), np.uint8)
    flips = int(rng.integers(0, 30))
    for arr in (a, b):
        for r in range(len(arr)):
            for bit in rng.integers(0, 256, flips):
                arr[r, bit >> 3] ^= np.uint8(1 << (bit & 7))
    if na == 0:
        return True, ("best2", na, nb)
    g, w_ = orb.ORBmatcher().block_best2(a, b), ol.block_best2(a, b)
    return all(np.array_equal(x, y) for x, y in zip(g, w_)), ("best2", na, nb, flips)


def case_stereo(rng):
    w, h = int(rng.integers(400, 800)), int(rng.integers(300, 520))
    nf = int(rng.integers(300, 1500))
    seq, t = int(rng.integers(0, 1 << 16)), int(rng.integers(0, 10))
    L, R = synth.sequence_frame(w, h, seq, t + 2), synth.sequence_frame(w, h, seq, t)
    exl, exr = orb.ORBextractor(nf, 1.2, 8, 20, 7), orb.ORBextractor(nf, 1.2, 8, 20, 7)
    rl, rr = ol.OracleExtractor(nf, 1.2, 8, 20, 7), ol.OracleExtractor(nf, 1.2, 8, 20, 7)
    (_, kl, dl), (_, kr, dr) = exl(L), exr(R)
    rl(L), rr(R)
    mb, mbf = 0.11, 47.9
    gu, gd = orb.ComputeStereoMatches(exl, 0, exr, 0, kl, dl, kr, dr, mb, mbf)
    wu, wd = ol.stereo_matches(rl, rr, kl, dl, kr, dr, mb, mbf)
    return gu.tobytes() == wu.tobytes() and gd.tobytes() == wd.tobytes(), ("stereo", w, h, nf, seq, t)


def case_bow(rng):
    k, L = int(rng.choice([3, 6, 10])), int(rng.integers(2, 5))
    blob = synth.synthetic_vocabulary(k=k, L=L, seed=int(rng.integers(0, 1000)), scoring=int(rng.integers(0, 2)),
                                      weighting=int(rng.integers(0, 4)))
    n = int(rng.integers(0, 1500))
    desc = synth.random_descriptors(max(n, 1), int(rng.integers(0, 1 << 16)))[:n]
    lv = int(rng.integers(0, L + 2))
    got = orb.ORBVocabulary(blob).transform(desc, lv)
    want = ol.OracleVocabulary(blob).transform(desc, lv)
    ok = np.array_equal(got["bow_ids"], want["bow_ids"]) and got["bow_vals"].tobytes() == want["bow_vals"].tobytes()
    ok = ok and all(np.array_equal(x, y) for x, y in zip(got["fv"], want["fv"]))
    return ok, ("bow", k, L, n, lv)


def case_bow_search(rng):
    """SearchByBoW(KF, F) / (KF, KF) on two resident frames (ORBmatcher.cc:226-428, 758-900), FeatureVectors host-side and
    resident: clustered descriptors (many features fight for the same candidates: the ordered walk's fixed point needs several
    rounds), random validity, nodes from a handful of features to the whole frame (the large-node form), fisheye Nleft."""
    k, L = int(rng.choice([3, 6, 10])), int(rng.integers(2, 4))
    blob = synth.synthetic_vocabulary(k=k, L=L, seed=int(rng.integers(0, 1000)))
    voc, ref = orb.ORBVocabulary(blob), ol.OracleVocabulary(blob)
    lv = int(rng.integers(0, L + 2))
    nA, nB = int(rng.integers(1, 420)), int(rng.integers(1, 420))
    base = synth.random_descriptors(int(rng.integers(1, 40)), int(rng.integers(0, 1 << 16)))

    def cloud(n):
        d = base[rng.integers(0, len(base), n)].copy()
        nflip = int(rng.integers(0, 12))
        for _ in range(nflip):
            bit = rng.integers(0, 256, n)
            d[np.arange(n), bit >> 3] ^= (1 << (bit & 7)).astype(np.uint8)
        return d

    def keys(n):
        kp = np.zeros(n, orb.KP_DTYPE)
        kp["x"], kp["y"] = rng.uniform(20, 600, n), rng.uniform(20, 440, n)
        kp["angle"] = rng.choice([0.0, 15.0, 29.9, 30.0, 45.0, 180.0, 359.0], n) if rng.random() < 0.5 else rng.uniform(0, 360, n)
        kp["octave"] = rng.integers(0, 8, n)
        kp["size"] = 31.0
        return kp
    dA, dB, kA, kB = cloud(nA), cloud(nB), keys(nA), keys(nB)
    b = (0.0, 0.0, 640.0, 480.0)
    nleft = int(rng.integers(0, nB + 1)) if rng.random() < 0.3 else -1
    fA, fB = orb.Frame(nA + 2).upload(kA, dA, b), orb.Frame(nB + 2).upload(kB, dB, b, nleft=nleft)
    fvA, fvB = fA.ComputeBoW(voc, lv)["fv"], fB.ComputeBoW(voc, lv)["fv"]
    ofvA, ofvB = ref.transform(dA, lv)["fv"], ref.transform(dB, lv)["fv"]
    ok = all(np.array_equal(x, y) for x, y in zip(fvA + fvB, ofvA + ofvB))
    vA = (rng.random(nA) < 0.8).astype(np.uint8)
    vB = (rng.random(nB) < 0.8).astype(np.uint8)
    ratio, ori = float(rng.choice([0.6, 0.7, 0.8, 0.9, 1.0])), bool(rng.integers(0, 2))
    want = ol.search_by_bow_kf_f(dA, kA["angle"], vA, ofvA, dB, kB["angle"], ofvB, ratio, ori, nleft)
    for fv in ((None, None), (fvA, fvB)):
        got = fA.SearchByBoW_KF_F(vA, fv[0], fB, fv[1], ratio, ori)
        ok = ok and got[0] == want[0] and np.array_equal(got[1], want[1])
    if nleft < 0:
        want = ol.search_by_bow_kf_kf(dA, kA["angle"], vA, ofvA, dB, kB["angle"], vB, ofvB, ratio, ori)
        for fv in ((None, None), (fvA, fvB)):
            got = fA.SearchByBoW_KF_KF(vA, fv[0], fB, vB, fv[1], ratio, ori)
            ok = ok and got[0] == want[0] and np.array_equal(got[1], want[1])
    return ok, ("bow_search", k, L, lv, nA, nB, len(base), nleft, ratio, ori)


def _rand_camera(rng, w, h):
    """None (mDistCoef(0) == 0: bounds = the image) a third of the time, otherwise a random pinhole camera with 4 or 5
    distortion coefficients in the range of the reference's settings files (TUM1: k1 0.26, k2 -0.95, k3 1.16; D435i:
    k1 0.125, k2 -0.25): Frame::ComputeImageBounds then gives fractional bounds, negative or inside the image."""
    if rng.random() < 0.34:
        return None
    f = float(rng.uniform(0.6, 1.1)) * w
    dist = [float(rng.uniform(-0.3, 0.3)) or 0.1, float(rng.uniform(-1.0, 1.0)), float(rng.uniform(-0.008, 0.008)),
            float(rng.uniform(-0.008, 0.008))]
    if rng.random() < 0.5:
        dist.append(float(rng.uniform(-1.2, 1.2)))
    return dict(K4=(f, f * float(rng.uniform(0.98, 1.02)), w / 2 + float(rng.uniform(-12, 12)),
                    h / 2 + float(rng.uniform(-12, 12))), dist=tuple(dist), size=(w, h))


def _camera_view(cam, keys, w, h):
    """(mvKeysUn, bounds) -- Frame::UndistortKeyPoints / ComputeImageBounds (Frame.cc:891-955) through the oracle harness;
    a camera whose bounds come out degenerate (wild coefficients) falls back to the image rectangle"""
    if cam is None:
        return keys, (0.0, 0.0, float(w), float(h))
    b = ol.image_bounds(cam)
    if not (np.all(np.isfinite(b)) and b[2] - b[0] > w / 4 and b[3] - b[1] > h / 4):
        return keys, (0.0, 0.0, float(w), float(h))
    return ol.undistort_keypoints(keys, cam), b


def case_window(rng):
    """Frame grid -> candidate lists -> the four windowed searches, random blocking state and thresholds; the target
    frame seen through a random (often distorted) camera."""
    w, h = int(rng.integers(300, 800)), int(rng.integers(240, 600))
    nf = int(rng.integers(100, 1500))
    seq = int(rng.integers(0, 1 << 16))
    ref = ol.OracleExtractor(nf, 1.2, 8, 20, 7)
    _, k0, d0 = ref(synth.sequence_frame(w, h, seq, 0))
    _, k1, d1 = ref(synth.sequence_frame(w, h, seq, 1))
    if len(k0) == 0 or len(k1) == 0:
        return True, ("window-empty",)
    cam = _rand_camera(rng, w, h)
    k1, bounds = _camera_view(cam, k1, w, h)
    k0, _ = _camera_view(cam, k0, w, h)
    g = orb.FrameGrid(k1, *bounds)
    og = ol.OracleGrid(k1, *bounds)
    r = rng.uniform(3, 40, len(k0)).astype(np.float32)
    qx = (k0["x"] + rng.uniform(-8, 8, len(k0))).astype(np.float32)
    qy = (k0["y"] + rng.uniform(-8, 8, len(k0))).astype(np.float32)
    lo = rng.integers(-1, 4, len(k0)).astype(np.int32)
    hi = (lo + rng.integers(-1, 4, len(k0))).astype(np.int32)
    off, idx = g.GetFeaturesInArea(qx, qy, r, lo, hi)
    woff, widx = [0], []
    for i in range(len(k0)):
        widx += og.query(qx[i], qy[i], r[i], int(lo[i]), int(hi[i])).tolist()
        woff.append(len(widx))
    if not (np.array_equal(off, woff) and np.array_equal(idx, widx)):
        return False, ("grid", w, h, nf, seq, cam)
    qb = (rng.random(len(k0)) < 0.7).astype(np.uint8)
    tb = (rng.random(len(k1)) < 0.2).astype(np.uint8)
    th = int(rng.choice([50, 100, 255]))  # 256 would make the reference index [-1] when every candidate is blocked
    ratio = float(rng.choice([0.6, 0.8, 0.9]))
    ori = bool(rng.integers(0, 2))
    m = orb.ORBmatcher(ratio, ori)
    a = orb.search_window(d0, qb, off, idx, d1, tb, th)
    b = ol.search_window(d0, qb, off, idx, d1, tb, th)
    ok = a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1:], b[1:]))
    a = m.SearchByProjection_Last(d0, k0["angle"], qb, off, idx, d1, k1["angle"], tb, th)
    b = ol.search_by_projection_last(d0, k0["angle"], qb, off, idx, d1, k1["angle"], tb, th, ori)
    ok = ok and a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1:], b[1:]))
    a = m.SearchByProjection_Local(d0, qb, off, idx, d1, k1["octave"], tb)
    b = ol.search_by_projection_local(d0, qb, off, idx, d1, k1["octave"], tb, ratio)
    ok = ok and a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1:], b[1:]))
    a = m.SearchForInitialization(d0, k0["angle"], k0["octave"], off, idx, d1, k1["angle"])
    b = ol.search_for_initialization(d0, k0["angle"], k0["octave"], off, idx, d1, k1["angle"], ratio, ori)
    ok = ok and a[0] == b[0] and np.array_equal(a[1], b[1])
    return ok, ("window", w, h, nf, seq, th, ratio, ori, cam)


def _rand_frame_pair(rng, stereo2):
    """(keys, desc, nleft, u_right, bounds) of a random target frame + (k0, d0) of the source frame it is searched from"""
    w, h = int(rng.integers(300, 800)), int(rng.integers(240, 600))
    nf = int(rng.integers(100, 1500))
    seq = int(rng.integers(0, 1 << 16))
    ref = ol.OracleExtractor(nf, 1.2, 8, 20, 7)
    _, k0, d0 = ref(synth.sequence_frame(w, h, seq, 0))
    _, k1, d1 = ref(synth.sequence_frame(w, h, seq, 1))
    bounds = (float(rng.choice([0.0, -7.5])), float(rng.choice([0.0, -3.25])), float(w), float(h))
    if stereo2:
        _, k2, d2 = ref(synth.sequence_frame(w, h, seq, 2))
        keys, desc, nleft, ur = np.concatenate([k1, k2]), np.concatenate([d1, d2]), len(k1), None
    else:
        cam = _rand_camera(rng, w, h)
        if cam is not None:  # Nleft == -1: the frame holds mvKeysUn and Frame::ComputeImageBounds' bounds
            k1, bounds = _camera_view(cam, k1, w, h)
            k0, _ = _camera_view(cam, k0, w, h)
        keys, desc, nleft = k1, d1, -1
        ur = np.where(rng.random(len(k1)) < 0.5, k1["x"] - rng.uniform(1, 30, len(k1)), -1).astype(np.float32)
        if rng.random() < 0.3:
            ur = None
    return keys, desc, nleft, ur, bounds, k0, d0


def case_resident(rng):
    """Device-resident frames: every routine-level search against the routine-level oracle, random geometry state."""
    import scenarios as sc
    stereo2 = bool(rng.integers(0, 2))
    keys, desc, nleft, ur, bounds, k0, d0 = _rand_frame_pair(rng, stereo2)
    n = len(k0)
    if n == 0 or len(keys) == 0:
        return True, ("resident-empty",)
    f = orb.Frame(len(keys)).upload(keys, desc, bounds, ur, nleft)
    o = ol.OracleFrame(keys, desc, bounds, ur, nleft)
    sf, inv2 = sc.SCALE_FACTORS, sc.INV_SIGMA2
    u = (k0["x"] - 3 + rng.normal(0, 2, n)).astype(np.float32)
    v = (k0["y"] - 2 + rng.normal(0, 2, n)).astype(np.float32)
    qd = sc.noisy_desc(rng, d0, int(rng.integers(0, 20)))
    lvl = np.clip(k0["octave"] + rng.integers(-1, 2, n), 0, 7).astype(np.int32)
    obs = (rng.random(n) < 0.7).astype(np.uint8)
    blocked = (rng.random(len(keys)) < 0.15).astype(np.uint8)
    th = float(rng.choice([1.0, 3.0, 7.0, 15.0]))
    ori = bool(rng.integers(0, 2))
    which = int(rng.integers(0, 6))
    tag = ("resident", which, stereo2, len(keys), n, th, ori)
    if which == 0:  # SearchByProjection(F, vpMapPoints)
        mp = dict(desc=qd, observed=obs, in_view=(rng.random(n) < 0.9).astype(np.uint8), proj_x=u, proj_y=v,
                  proj_xr=(u - rng.uniform(1, 30, n)).astype(np.float32), scale_level=lvl,
                  view_cos=rng.choice(np.array([0.9, 0.9985], np.float32), n))
        ltr = rtl = None
        if stereo2:
            lr = lvl.copy()
            lr[rng.random(n) < 0.05] = -1
            mp.update(in_view_r=(rng.random(n) < 0.7).astype(np.uint8), proj_x_r=(u + 8).astype(np.float32), proj_y_r=v,
                      scale_level_r=lr, view_cos_r=rng.choice(np.array([0.9, 0.9999], np.float32), n))
            nr = len(keys) - nleft
            ltr, rtl = np.full(nleft, -1, np.int32), np.full(nr, -1, np.int32)
            m = min(nleft, nr) // 3
            a = rng.choice(min(nleft, nr), m, replace=False)
            b = rng.permutation(a)
            ltr[a], rtl[b] = b, a
        ratio = float(rng.choice([0.6, 0.8, 0.9]))
        g = f.SearchByProjection(mp, th, ratio, sf, blocked, ltr, rtl)
        r = o.search_by_projection(mp, th, ratio, sf, blocked, ltr, rtl)
    elif which == 1:  # SearchByProjection(CurrentFrame, LastFrame)
        d = int(rng.integers(0, 3))
        args = (qd, obs, u, v, (u - rng.uniform(1, 30, n)).astype(np.float32), k0["octave"].astype(np.int32),
                k0["angle"].astype(np.float32), th, d, sf, ori, blocked)
        kw = dict(u_r=(u + 8).astype(np.float32), v_r=v) if stereo2 else {}
        g, r = f.SearchByProjection_Last(*args, **kw), o.search_by_projection_last(*args, **kw)
    elif which == 2 and not stereo2:  # SearchByProjection(KF, Sim3) / (Frame, KF)
        rad = (np.float32(th) * sf[lvl]).astype(np.float32)
        matched = np.where(rng.random(len(keys)) < 0.15, 99999, -1).astype(np.int32)
        ratio = float(rng.choice([0.5, 1.0, 1.5]))
        g1, r1 = f.SearchByProjection_Sim3(qd, u, v, rad, lvl, ratio, matched), o.search_by_projection_sim3(qd, u, v, rad, lvl, ratio, matched)
        od = int(rng.choice([50, 64, 100]))
        g2 = f.SearchByProjection_KF(qd, u, v, rad, lvl, k0["angle"], od, ori, blocked)
        r2 = o.search_by_projection_kf(qd, u, v, rad, lvl, k0["angle"], od, ori, blocked)
        g, r = (g1[0] + g2[0], g1[1], g2[1], g2[2]), (r1[0] + r2[0], r1[1], r2[1], r2[2])
    elif which == 3:  # Fuse x2 (bRight on fisheye-stereo frames)
        rad = (np.float32(th) * sf[lvl]).astype(np.float32)
        right = stereo2 and bool(rng.integers(0, 2))
        ur_q = (u - rng.uniform(1, 30, n)).astype(np.float32)
        nfu, bi, bd = f.Fuse(qd, u, v, ur_q, rad, lvl, inv2, right=right)
        nk = len(keys)
        slot = np.where(rng.random(nk) < 0.5, n + np.arange(nk), -1).astype(np.int32)
        mobs = rng.integers(1, 6, n + nk).astype(np.int32)
        bad = (rng.random(n + nk) < 0.1).astype(np.uint8)
        qmp = np.arange(n, dtype=np.int32)
        rr = o.fuse(qmp, qd, u, v, ur_q, rad, lvl, inv2, slot, mobs, bad, right=right)
        dd = orb.fuse_decide(qmp, bi, bd, False, slot, mobs, bad)
        g = (nfu, bi, bd, dd[1], dd[3])
        r = (rr[0], rr[1], rr[2], rr[3], rr[5])
        if not stereo2:
            nf2, bi2, bd2 = f.Fuse_Sim3(qd, u, v, rad, lvl)
            r2 = o.fuse_sim3(qmp, qd, u, v, rad, lvl, slot, mobs, bad)
            g, r = g + (nf2, bi2, bd2), r + (r2[0], r2[1], r2[2])
    elif which == 4 and not stereo2:  # SearchBySim3 + SearchForInitialization (two resident frames)
        f0, o0 = orb.Frame(n).upload(k0, d0, bounds), ol.OracleFrame(k0, d0, bounds)
        i1 = np.sort(rng.choice(n, max(1, n // 2), replace=False)).astype(np.int32)
        q1 = dict(idx=i1, desc=qd[i1], u=u[i1], v=v[i1], radius=(np.float32(th) * sf[lvl[i1]]).astype(np.float32), level=lvl[i1])
        n1 = len(keys)
        i2 = np.sort(rng.choice(n1, max(1, n1 // 2), replace=False)).astype(np.int32)
        l2 = keys["octave"][i2].astype(np.int32)
        q2 = dict(idx=i2, desc=desc[i2], u=(keys["x"][i2] + 3).astype(np.float32), v=(keys["y"][i2] + 2).astype(np.float32),
                  radius=(np.float32(th) * sf[l2]).astype(np.float32), level=l2)
        g1, r1 = orb.SearchBySim3(f0, f, q1, q2), ol.search_by_sim3(o0, o, q1, q2)
        ws = int(rng.choice([20, 100]))
        g2 = f0.SearchForInitialization(f, k0["x"], k0["y"], ws, 0.9, ori)
        r2 = o0.search_for_initialization(o, k0["x"], k0["y"], ws, 0.9, ori)
        g, r = (g1[0] + g2[0], g1[1], g2[1]), (r1[0] + r2[0], r1[1], r2[1])
    else:  # windows only, Frame and KeyFrame forms, both grids
        nq = min(n, 400)
        rr_ = rng.uniform(0.5, 120, nq).astype(np.float32)
        lo, hi = rng.integers(-1, 8, nq).astype(np.int32), rng.integers(-1, 8, nq).astype(np.int32)
        right = stereo2 and bool(rng.integers(0, 2))
        off, idx = f.GetFeaturesInArea(u[:nq], v[:nq], rr_, lo, hi, bRight=right)
        want = [o.features_in_area(u[q], v[q], rr_[q], lo[q], hi[q], right) for q in range(nq)]
        g = (len(idx), off, idx)
        r = (sum(len(x) for x in want), np.concatenate([[0], np.cumsum([len(x) for x in want])]).astype(np.int32),
             np.concatenate(want).astype(np.int32) if want else np.zeros(0, np.int32))
    ok = g[0] == r[0] and all(np.array_equal(x, y) for x, y in zip(g[1:], r[1:]))
    if not ok and which == 5:  # keep the failing window case for a look at it off the box
        np.savez_compressed(ROOT / "gpurun_out" / "fuzz_fail_windows.npz", keys=keys, desc=desc, bounds=np.array(bounds),
                            ur=ur if ur is not None else np.zeros(0, np.float32), nleft=nleft, u=u[:nq], v=v[:nq], r=rr_,
                            lo=lo, hi=hi, right=right, got_off=g[1], got_idx=g[2], want_off=r[1], want_idx=r[2])
    return ok, tag


def case_undistort(rng):
    """A frame straight out of the extractor through a random distorted camera: vsg_frame_from_extractor_undistort
    (Frame::UndistortKeyPoints on the device, FP64) against the oracle harness -- mvKeysUn bit for bit, the library's own
    ComputeImageBounds, the device-built grid, and one windowed search on the resident frame."""
    w, h = int(rng.integers(300, 800)), int(rng.integers(240, 600))
    nf = int(rng.integers(100, 1500))
    seq = int(rng.integers(0, 1 << 16))
    cam = _rand_camera(rng, w, h) or dict(K4=(0.8 * w, 0.8 * w, w / 2, h / 2), dist=(0.0, 0.3, 0.0, 0.0), size=(w, h))
    img = synth.sequence_frame(w, h, seq, 1)
    ex = orb.ORBextractor(nf, 1.2, 8, 20, 7)
    _, k, d = ex(img)
    if len(k) == 0:
        return True, ("undistort-empty",)
    bounds = orb.camera_image_bounds(w, h, cam["K4"], cam["dist"])
    tag = ("undistort", w, h, nf, seq, cam)
    if bounds != ol.image_bounds(cam):
        return False, tag
    if not (np.all(np.isfinite(bounds)) and bounds[2] - bounds[0] > w / 4 and bounds[3] - bounds[1] > h / 4):
        return True, ("undistort-degenerate",)
    f = orb.Frame(ex.capacity(h, w)).from_extractor_undistort(ex, 0, k, cam["K4"], cam["dist"], bounds)
    kun = ol.undistort_keypoints(k, cam)
    if f.kps.tobytes() != kun.tobytes():
        return False, tag
    o = ol.OracleFrame(kun, d, bounds)
    ok = all(np.array_equal(a, b) for a, b in zip(f.grid(), o.grid()))
    import scenarios as sc
    n = len(kun)
    u = (kun["x"] - 3 + rng.normal(0, 2, n)).astype(np.float32)
    v = (kun["y"] - 2 + rng.normal(0, 2, n)).astype(np.float32)
    lvl = np.clip(kun["octave"] + rng.integers(-1, 2, n), 0, 7).astype(np.int32)
    rad = (np.float32(rng.choice([3.0, 7.0])) * sc.SCALE_FACTORS[lvl]).astype(np.float32)
    qd = sc.noisy_desc(rng, d, 8)
    m0 = np.full(n, -1, np.int32)
    a, b = f.SearchByProjection_Sim3(qd, u, v, rad, lvl, 1.0, m0), o.search_by_projection_sim3(qd, u, v, rad, lvl, 1.0, m0)
    return bool(ok and a[0] == b[0] and np.array_equal(a[1], b[1])), tag


TRACE = False
DIRECT_REGISTERED = False  # --direct-registered: registered arrays are read / written in place (opt-in route)
ALLOC_ONLY = False  # --alloc-only: every pinned `async` case uses vsg_host_alloc memory
ALLOC = True  # --registered-only: every pinned `async` case registers numpy arrays (the form profiles/r04_q_* is about)


def case_async(rng):
    """vsg_orb_submit_batch / vsg_orb_wait with random batch sizes, strides, pinned / pageable buffers, lapping areas."""
    w, h, nf, sc_, nl, ini, mn = geometry(rng)
    # one case in eight: batches of 16 frames and more, which pageable input stages with the handle's helper threads
    B = int(rng.integers(16, 21)) if rng.integers(0, 8) == 0 else int(rng.integers(1, 6))
    ex = orb.ORBextractor(nf, sc_, nl, ini, mn, max_batch=B)
    ref = ol.OracleExtractor(nf, sc_, nl, ini, mn)
    cap = ex.capacity(h, w)
    lap = (int(rng.integers(-10, w)), int(rng.integers(-10, w + 50)))
    pinned = bool(rng.integers(0, 2))
    nb = int(rng.integers(1, 6))
    # pinned buffers: registered numpy arrays, or (ALLOC, every second pinned case) arrays over vsg_host_alloc memory.
    # Registered arrays take the library's DEFAULT route (staged like pageable memory, the device never touches them)
    # unless --direct-registered opts the handle in to in-place access (the hunt mode of profiles/r04_q_* / r05_*)
    alloc = pinned and ALLOC and (bool(rng.integers(0, 2)) or ALLOC_ONLY)
    if DIRECT_REGISTERED:
        ex.set_direct_registered(True)
    tickets, bufs = [], []
    ok = True
    if TRACE:
        print("  async", w, h, nf, sc_, nl, "B", B, "nb", nb, "pinned", pinned, "lap", lap, "cap", cap, flush=True)
    for k in range(nb):
        b = int(rng.integers(max(1, B - 4), B + 1))
        pad = int(rng.integers(0, 9))
        big = np.zeros((b, h, w + pad), np.uint8)
        imgs = np.stack([synth.frame(w, h, int(rng.integers(0, 1 << 20))) for _ in range(b)])
        big[:, :, :w] = imgs
        view = big[:, :, :w]
        kps, desc = np.zeros((b, cap), orb.KP_DTYPE), np.zeros((b, cap, 32), np.uint8)
        owners = ()
        if alloc:
            owners = (orb.PinnedArray(big.shape), orb.PinnedArray(kps.shape, orb.KP_DTYPE), orb.PinnedArray(desc.shape))
            owners[0].a[...] = big
            big, kps, desc = owners[0].a, owners[1].a, owners[2].a
            view = big[:, :, :w]
        elif pinned:
            big, kps, desc = orb.pin(big), orb.pin(kps), orb.pin(desc)
            view = big[:, :, :w]
        if len(tickets) == ex.slots():
            ok &= _finish(ex, ref, tickets.pop(0), bufs.pop(0), lap, pinned)
        if TRACE:
            print("    submit", k, "b", b, "pad", pad, hex(big.ctypes.data), big.nbytes, hex(kps.ctypes.data), kps.nbytes, hex(desc.ctypes.data), desc.nbytes, flush=True)
        tickets.append(ex.submit_batch(view, kps, desc, lap))
        bufs.append((imgs, big, kps, desc, owners))
    while tickets:
        ok &= _finish(ex, ref, tickets.pop(0), bufs.pop(0), lap, pinned)
    return ok, ("async", w, h, nf, sc_, nl, B, nb, pinned, lap)


def _finish(ex, ref, ticket, buf, lap, pinned):
    imgs, big, kps, desc, owners = buf
    n, mono = ex.wait(ticket)
    ok = True
    for i in range(len(imgs)):
        rm, rk, rd = ref(imgs[i], lap)
        ok &= n[i] == len(rk) and mono[i] == rm and kps[i, :n[i]].tobytes() == rk.tobytes() and np.array_equal(desc[i, :n[i]], rd)
    if owners:
        del big, kps, desc
        for o in owners:
            o.free()
    elif pinned:
        orb.unpin(big), orb.unpin(kps), orb.unpin(desc)
    return bool(ok)


CASES = {"undistort": case_undistort, "resident": case_resident, "async": case_async, "window": case_window, "batch": case_batch, "colour": case_colour, "best2": case_best2, "stereo": case_stereo, "bow": case_bow, "bow_search": case_bow_search}


def run_child(args, argv):
    """One run = one FRESH child process (never a re-exec of a process that has touched the GPU): a GPU memory access
    fault kills the child, and the parent reports it as a failed run with the child's last lines."""
    import subprocess
    failed = 0
    for r in range(args.runs):
        child_args, skip = [], False
        for a in argv:  # everything but --runs N / --runs=N
            if skip:
                skip = False
            elif a == "--runs":
                skip = True
            elif not a.startswith("--runs="):
                child_args.append(a)
        cmd = [sys.executable, str(Path(__file__).resolve()), "--child"] + child_args
        if "--trace" not in cmd:
            cmd.append("--trace")  # the last line names the case a fault happened in
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        lines = p.stdout.splitlines()
        tail = [x for x in lines if not x.startswith(("case ", "  async", "    submit", "REFUSED"))][-5:]
        if p.returncode == 0:
            print(f"run {r}: " + (tail[-1] if tail else "ok"), flush=True)
        else:
            failed += 1
            last_case = [x for x in lines if x.startswith("case ")][-1:] or ["(no case printed)"]
            how = f"signal {-p.returncode}" if p.returncode < 0 else f"exit code {p.returncode}"
            print(f"run {r}: CHILD DIED ({how}) in / after '{last_case[0]}'", flush=True)
            for x in lines[-12:]:
                print("    | " + x, flush=True)
    print(f"fuzz runs: {args.runs - failed} of {args.runs} clean", flush=True)
    return 1 if failed else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--only", default="")
    ap.add_argument("--trace", action="store_true", help="print every case's index and name before it runs (a GPU fault kills the process: the last line names the case)")
    ap.add_argument("--registered-only", action="store_true", help="pinned `async` cases never use vsg_host_alloc memory")
    ap.add_argument("--alloc-only", action="store_true", help="pinned `async` cases always use vsg_host_alloc memory")
    ap.add_argument("--direct-registered", action="store_true",
                    help="hunt mode: handles opt in to in-place device access of registered numpy arrays "
                         "(vsg_orb_set_direct_registered); default = the library's default route, which stages them")
    ap.add_argument("--stop-at", type=int, default=-1, help="run cases up to this index only (the random stream stays the same)")
    ap.add_argument("--runs", type=int, default=1, help="repeat the run this many times, each in a fresh child process")
    ap.add_argument("--child", action="store_true", help="(internal) this process runs the cases itself")
    args = ap.parse_args()
    if not args.child:
        sys.exit(run_child(args, sys.argv[1:]))
    global TRACE, ALLOC, ALLOC_ONLY, DIRECT_REGISTERED
    TRACE = args.trace
    ALLOC = not args.registered_only
    ALLOC_ONLY = args.alloc_only
    DIRECT_REGISTERED = args.direct_registered
    rng = np.random.default_rng(args.seed)
    names = [n for n in CASES if not args.only or n in args.only.split(",")]
    counts = {n: 0 for n in names}
    for i in range(args.cases):
        name = names[int(rng.integers(0, len(names)))]
        if args.trace:
            print("case", i, name, flush=True)
        if 0 <= args.stop_at < i:
            break
        try:
            ok, desc = CASES[name](rng)
        except orb.VsgError as e:
            print("REFUSED", name, e)
            continue
        counts[name] += 1
        if not ok:
            print("MISMATCH", desc)
            sys.exit(1)
    print("fuzz ok:", counts)


if __name__ == "__main__":
    main()
