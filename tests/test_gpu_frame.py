"""GPU parity tests of the device-resident frame path (vsg_frame_*, include/vsg_orb.h) against the routine-level CPU
oracle (oracle/routines_oracle.cpp), through the C ABI.  Everything is integer / index work: the bar is bit-exact."""
import threading

import numpy as np
import pytest

import oracle_lib as ol
import scenarios as sc
from visual_sgraphs_amd import orb, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=sc.CAMERA_NAMES)
def camera(request):
    """Every test of this file runs under three cameras: no distortion (bounds = the image rectangle) and
    the two BASELINE cameras whose Frame constructor undistorts -- TUM1 (C1) and RealSense D435i (C5): the frame's keys
    are mvKeysUn, the grid bounds Frame::ComputeImageBounds' fractional / negative values, mfGridElementWidthInv no
    longer 64 / W (Frame.cc:378-379, 870-880, 891-955)."""
    sc.use_camera(request.param)
    yield request.param
    sc.use_camera("image")


def _frame(keys, desc, u_right=None, nleft=-1, cap=None):
    return orb.Frame(cap or max(len(keys), 1)).upload(keys, desc, sc.BOUNDS, u_right, nleft)


# ---------------------------------------------------------------------------------------------- grid + windows
@pytest.mark.parametrize("seed", [1, 2])
def test_uploaded_grid_equals_assign_features_to_grid(seed):
    keys, desc, nleft = sc.stereo_pair(seed)
    f = _frame(keys, desc, nleft=nleft)
    o = ol.OracleFrame(keys, desc, sc.BOUNDS, nleft=nleft)
    for right in (False, True):
        cs, en = f.grid(right)
        ocs, oen = o.grid(right)
        assert np.array_equal(cs, ocs) and np.array_equal(en, oen)


def test_device_built_grid_from_extractor_equals_oracle():
    """vsg_frame_from_extractor: keypoints / descriptors device to device, grid built by k_frame_grid_build."""
    imgs = np.stack([synth.sequence_frame(320, 240, 11, t) for t in range(3)])
    ex = orb.ORBextractor(600, 1.2, 8, 20, 7, max_batch=3)
    outs = ex.extract_batch(imgs)
    for i, (_, k, d) in enumerate(outs):
        if sc.CAM is None:
            f = orb.Frame(ex.capacity(240, 320)).from_extractor(ex, i, k, sc.BOUNDS)
        else:
            # distorted camera: Frame::UndistortKeyPoints runs on the device inside the grid launch; the bounds come
            # from the library's own ComputeImageBounds and must be the oracle's bit for bit
            bounds = orb.camera_image_bounds(320, 240, sc.CAM["K4"], sc.CAM["dist"])
            assert bounds == sc.BOUNDS
            f = orb.Frame(ex.capacity(240, 320)).from_extractor_undistort(ex, i, k, sc.CAM["K4"], sc.CAM["dist"], bounds)
            k = ol.undistort_keypoints(k, sc.CAM)   # mvKeysUn by the oracle
            assert f.kps.tobytes() == k.tobytes()   # ... equals what the device computed, bit for bit
            # and the upload route (host undistortion, then vsg_frame_upload) builds the same grid
            fu = orb.Frame(ex.capacity(240, 320)).upload(k, d, sc.BOUNDS)
            assert all(np.array_equal(a, b) for a, b in zip(f.grid(), fu.grid()))
        o = ol.OracleFrame(k, d, sc.BOUNDS)
        cs, en = f.grid()
        ocs, oen = o.grid()
        assert f.N == len(k) and np.array_equal(cs, ocs) and np.array_equal(en, oen)
        # the descriptors really are resident: a window search against them equals the oracle
        s = sc.kf_projection_scenario(3)
        n, m = f.SearchByProjection_Sim3(s["q_desc"], s["u"], s["v"], s["radius"], s["level"], 1.0,
                                         np.full(len(k), -1, np.int32))
        n2, m2 = o.search_by_projection_sim3(s["q_desc"], s["u"], s["v"], s["radius"], s["level"], 1.0,
                                             np.full(len(k), -1, np.int32))
        assert n == n2 and np.array_equal(m, m2)


def test_extract_into_frame_is_operator_plus_resident_frame():
    """vsg_orb_extract_to_frame (the Frame constructor's front end in one call and one wait: operator() -> UndistortKeyPoints
    -> AssignFeaturesToGrid) against the oracle and against the two separate calls, under the current camera; repeated calls
    on the same frame object (sizes change from call to call) and the empty frame."""
    ex = orb.ORBextractor(600, 1.2, 8, 20, 7)
    ref = ol.OracleExtractor(600, 1.2, 8, 20, 7)
    f = orb.Frame(ex.capacity(240, 320))
    K4, dist = (sc.CAM["K4"], sc.CAM["dist"]) if sc.CAM is not None else (None, None)
    for t, img in enumerate([synth.sequence_frame(320, 240, 12, 0), synth.content_frame("ramp", 320, 240, 12, 1),
                             synth.sequence_frame(320, 240, 12, 2), np.full((240, 320), 90, np.uint8)]):
        mono, k, d = f.extract_into(ex, img, sc.BOUNDS, K4, dist)
        rm, rk, rd = ref(img)
        assert mono == rm and k.tobytes() == rk.tobytes() and np.array_equal(d, rd), t
        kun = ol.undistort_keypoints(rk, sc.CAM) if sc.CAM is not None else rk
        assert f.N == len(rk) and f.kps.tobytes() == kun.tobytes(), t
        o = ol.OracleFrame(kun, rd, sc.BOUNDS)
        assert all(np.array_equal(a, b) for a, b in zip(f.grid(), o.grid())), t
        if len(rk):
            s = sc.kf_projection_scenario(3)
            m0 = np.full(len(rk), -1, np.int32)
            got = f.SearchByProjection_Sim3(s["q_desc"], s["u"], s["v"], s["radius"], s["level"], 1.0, m0)
            want = o.search_by_projection_sim3(s["q_desc"], s["u"], s["v"], s["radius"], s["level"], 1.0, m0)
            assert got[0] == want[0] and np.array_equal(got[1], want[1]), t
    # a lapping area (the mono case: everything from the back) goes through k_slots and the same hook
    mono, k, d = f.extract_into(ex, synth.sequence_frame(320, 240, 12, 3), sc.BOUNDS, K4, dist, vLappingArea=(0, 1000))
    rm, rk, rd = ref(synth.sequence_frame(320, 240, 12, 3), (0, 1000))
    assert mono == rm and k.tobytes() == rk.tobytes() and np.array_equal(d, rd)


@pytest.mark.parametrize("seed", [1, 2])
def test_features_in_area_order_and_filters(seed):
    keys, desc, nleft = sc.stereo_pair(seed)
    f = _frame(keys, desc, nleft=nleft)
    o = ol.OracleFrame(keys, desc, sc.BOUNDS, nleft=nleft)
    rng = np.random.default_rng(seed)
    nq = 300
    x, y = rng.uniform(-30, 350, nq).astype(np.float32), rng.uniform(-30, 270, nq).astype(np.float32)
    r = rng.uniform(0.5, 90, nq).astype(np.float32)
    lo, hi = rng.integers(-1, 8, nq).astype(np.int32), rng.integers(-1, 8, nq).astype(np.int32)
    for right in (False, True):
        off, idx = f.GetFeaturesInArea(x, y, r, lo, hi, bRight=right)
        offk, idxk = f.GetFeaturesInArea(x, y, r, bRight=right)
        for q in range(nq):
            assert np.array_equal(idx[off[q]:off[q + 1]], o.features_in_area(x[q], y[q], r[q], lo[q], hi[q], right))
            assert np.array_equal(idxk[offk[q]:offk[q + 1]], o.features_in_area(x[q], y[q], r[q], right=right, kf_form=True))
    assert off[-1] > 5 * nq  # windows of up to 180 px: the lists overflow the first stride and are re-run


# ---------------------------------------------------------------------------------------------- SearchByProjection x5
@pytest.mark.parametrize("seed,stereo2", [(1, False), (2, False), (3, False), (1, True), (2, True), (4, True)])
def test_search_by_projection_local_map(seed, stereo2):
    s = sc.local_map_scenario(seed, stereo2)
    f = _frame(s["keys"], s["desc"], s["u_right"], s["nleft"])
    o = ol.OracleFrame(s["keys"], s["desc"], sc.BOUNDS, s["u_right"], s["nleft"])
    got = f.SearchByProjection(s["mp"], s["th"], s["nnratio"], sc.SCALE_FACTORS, s["blocked"], s["ltr"], s["rtl"])
    ref = o.search_by_projection(s["mp"], s["th"], s["nnratio"], sc.SCALE_FACTORS, s["blocked"], s["ltr"], s["rtl"])
    assert ref[0] > 40 and got[0] == ref[0]
    assert np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
    if stereo2:
        assert np.any(ref[1][s["nleft"]:] >= 0)  # the right-camera block produced matches


@pytest.mark.parametrize("seed,stereo2", [(1, False), (2, False), (3, False), (5, False), (1, True), (2, True), (3, True)])
def test_search_by_projection_last_frame(seed, stereo2):
    s = sc.last_frame_scenario(seed, stereo2)
    f = _frame(s["keys"], s["desc"], s["u_right"], s["nleft"])
    o = ol.OracleFrame(s["keys"], s["desc"], sc.BOUNDS, s["u_right"], s["nleft"])
    for check_ori in (True, False):
        args = (s["q_desc"], s["observed"], s["u"], s["v"], s["ur"], s["octave"], s["angle"], s["th"], s["direction"],
                sc.SCALE_FACTORS, check_ori, s["blocked"])
        got = f.SearchByProjection_Last(*args, u_r=s["u_r"], v_r=s["v_r"])
        ref = o.search_by_projection_last(*args, u_r=s["u_r"], v_r=s["v_r"])
        assert ref[0] > 40 and got[0] == ref[0]
        assert np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])


def test_last_frame_quirks_on_device():
    """the two known answers of tests/test_routines_oracle.py through the HIP path"""
    d0 = np.zeros((1, 32), np.uint8)
    k = np.zeros(2, orb.KP_DTYPE)
    k["x"], k["y"] = [50, 200], [50, 200]
    f = _frame(k, np.concatenate([d0, d0]), nleft=1)
    common = dict(last_octave=[0], last_angle=[0.0], th=7.0, direction=0, scale_factors=sc.SCALE_FACTORS,
                  check_orientation=False, train_blocked=np.zeros(2, np.uint8))
    n, tm, _ = f.SearchByProjection_Last(d0, [1], [120.0], [120.0], None, u_r=[200.0], v_r=[200.0], **common)
    assert n == 0 and tm.tolist() == [-1, -1]  # empty left window: the right block is skipped (:1727)
    n, tm, _ = f.SearchByProjection_Last(d0, [1], [50.0], [50.0], None, u_r=[200.0], v_r=[200.0], **common)
    assert n == 2 and tm.tolist() == [0, 0]


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("ratio", [1.0, 1.5, 0.5])
def test_search_by_projection_keyframe_sim3(seed, ratio):
    s = sc.kf_projection_scenario(seed)
    f, o = _frame(s["keys"], s["desc"]), ol.OracleFrame(s["keys"], s["desc"], sc.BOUNDS)
    matched = np.full(len(s["keys"]), -1, np.int32)
    matched[s["rng"].random(len(matched)) < 0.15] = 12345  # vpMatched already holds a map point
    got = f.SearchByProjection_Sim3(s["q_desc"], s["u"], s["v"], s["radius"], s["level"], ratio, matched)
    ref = o.search_by_projection_sim3(s["q_desc"], s["u"], s["v"], s["radius"], s["level"], ratio, matched)
    assert ref[0] > 10 and got[0] == ref[0] and np.array_equal(got[1], ref[1])


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("orb_dist", [64, 100])
def test_search_by_projection_frame_keyframe(seed, orb_dist):
    s = sc.kf_projection_scenario(seed)
    f, o = _frame(s["keys"], s["desc"]), ol.OracleFrame(s["keys"], s["desc"], sc.BOUNDS)
    occ = (s["rng"].random(len(s["keys"])) < 0.15).astype(np.uint8)
    for check_ori in (True, False):
        got = f.SearchByProjection_KF(s["q_desc"], s["u"], s["v"], s["radius"], s["level"], s["angle"], orb_dist,
                                      check_ori, occ)
        ref = o.search_by_projection_kf(s["q_desc"], s["u"], s["v"], s["radius"], s["level"], s["angle"], orb_dist,
                                        check_ori, occ)
        assert ref[0] > 20 and got[0] == ref[0]
        assert np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])


# ---------------------------------------------------------------------------------------------- SearchBySim3, Fuse
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_search_by_sim3_two_directions_and_agreement(seed):
    rng = np.random.default_rng(seed + 77)
    k1, d1 = sc.features(seed, 0)
    k2, d2 = sc.features(seed, 1)
    f1, f2 = _frame(k1, d1), _frame(k2, d2)
    o1, o2 = ol.OracleFrame(k1, d1, sc.BOUNDS), ol.OracleFrame(k2, d2, sc.BOUNDS)

    def direction(src_k, src_d, shift):
        idx = np.sort(rng.choice(len(src_k), int(0.8 * len(src_k)), replace=False)).astype(np.int32)
        u, v = sc.projections(rng, src_k[idx], shift=shift)
        lvl = np.clip(src_k["octave"][idx] + rng.integers(-1, 2, len(idx)), 0, 7).astype(np.int32)
        return dict(idx=idx, desc=sc.noisy_desc(rng, src_d[idx], 6), u=u, v=v,
                    radius=(np.float32(7.5) * sc.SCALE_FACTORS[lvl]).astype(np.float32), level=lvl)
    q1, q2 = direction(k1, d1, (-3.0, -2.0)), direction(k2, d2, (3.0, 2.0))
    got = orb.SearchBySim3(f1, f2, q1, q2)
    ref = ol.search_by_sim3(o1, o2, q1, q2)
    assert ref[0] > 30 and got[0] == ref[0] and np.array_equal(got[1], ref[1])
    # one empty direction: nothing can agree
    e = dict(idx=[], desc=np.zeros((0, 32), np.uint8), u=[], v=[], radius=[], level=[])
    assert orb.SearchBySim3(f1, f2, q1, e)[0] == 0


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fuse_searches_and_decisions(seed):
    s = sc.kf_projection_scenario(seed)
    rng = s["rng"]
    f = _frame(s["keys"], s["desc"], s["u_right"])
    o = ol.OracleFrame(s["keys"], s["desc"], sc.BOUNDS, u_right=s["u_right"])
    nq, nk = len(s["u"]), len(s["keys"])
    slot = np.where(rng.random(nk) < 0.5, nq + np.arange(nk), -1).astype(np.int32)
    obs = rng.integers(1, 6, nq + nk).astype(np.int32)
    bad = (rng.random(nq + nk) < 0.1).astype(np.uint8)
    qmp = np.arange(nq, dtype=np.int32)
    # Fuse(pKF, vpMapPoints, th, bRight = false): chi-square gate with and without mvuRight
    nf, bi, bd = f.Fuse(s["q_desc"], s["u"], s["v"], s["ur"], s["radius"], s["level"], sc.INV_SIGMA2)
    r = o.fuse(qmp, s["q_desc"], s["u"], s["v"], s["ur"], s["radius"], s["level"], sc.INV_SIGMA2, slot, obs, bad)
    assert r[0] >= 10 and nf == r[0] and np.array_equal(bi, r[1]) and np.array_equal(bd, r[2])
    d = orb.fuse_decide(qmp, bi, bd, False, slot, obs, bad)
    assert d[0] == r[0] and np.array_equal(d[1], r[3]) and np.array_equal(d[3], r[5])
    # Fuse(pKF, Scw, vpPoints, th, vpReplacePoint)
    nf, bi, bd = f.Fuse_Sim3(s["q_desc"], s["u"], s["v"], s["radius"], s["level"])
    r = o.fuse_sim3(qmp, s["q_desc"], s["u"], s["v"], s["radius"], s["level"], slot, obs, bad)
    assert r[0] >= 20 and nf == r[0] and np.array_equal(bi, r[1]) and np.array_equal(bd, r[2])
    d = orb.fuse_decide(qmp, bi, bd, True, slot, obs, bad)
    assert d[0] == r[0] and np.array_equal(d[1], r[3]) and np.array_equal(d[2], r[4])


@pytest.mark.parametrize("seed", [1, 2])
def test_fuse_right_camera(seed):
    """Fuse(..., bRight = true) on a KeyFrame with NLeft != -1: mGridRight, mvKeysRight, indices + NLeft (:1294)."""
    keys, desc, nleft = sc.stereo_pair(seed)
    s = sc.kf_projection_scenario(seed)
    f, o = _frame(keys, desc, nleft=nleft), ol.OracleFrame(keys, desc, sc.BOUNDS, nleft=nleft)
    nq = len(s["u"])
    nf, bi, bd = f.Fuse(s["q_desc"], s["u"], s["v"], s["ur"], s["radius"], s["level"], sc.INV_SIGMA2, right=True)
    z = np.zeros(nq + len(keys), np.int32)
    r = o.fuse(np.arange(nq), s["q_desc"], s["u"], s["v"], s["ur"], s["radius"], s["level"], sc.INV_SIGMA2,
               np.full(len(keys), -1, np.int32), z, z.astype(np.uint8), right=True)
    assert r[0] >= 10 and nf == r[0] and np.array_equal(bi, r[1]) and np.array_equal(bd, r[2])
    assert bi[bi >= 0].min() >= nleft


# ---------------------------------------------------------------------------------------------- init, BoW, stereo
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_search_for_initialization_resident(seed):
    k1, d1 = sc.features(seed, 0)
    k2, d2 = sc.features(seed, 1)
    f1, f2 = _frame(k1, d1), _frame(k2, d2)
    o1, o2 = ol.OracleFrame(k1, d1, sc.BOUNDS), ol.OracleFrame(k2, d2, sc.BOUNDS)
    for check_ori in (True, False):
        got = f1.SearchForInitialization(f2, k1["x"], k1["y"], 100, 0.9, check_ori)
        ref = o1.search_for_initialization(o2, k1["x"], k1["y"], 100, 0.9, check_ori)
        assert ref[0] > 30 and got[0] == ref[0] and np.array_equal(got[1], ref[1])


@pytest.mark.parametrize("seed", [1, 2])
def test_bow_chain_on_resident_frames(seed):
    blob = synth.synthetic_vocabulary(k=10, L=3, seed=seed)
    voc, ovoc = orb.ORBVocabulary(blob), ol.OracleVocabulary(blob)
    k1, d1 = sc.features(seed, 0)
    k2, d2 = sc.features(seed, 1)
    f1, f2 = _frame(k1, d1), _frame(k2, d2)
    t1, t2 = f1.ComputeBoW(voc, 2), f2.ComputeBoW(voc, 2)
    r1, r2 = ovoc.transform(d1, 2), ovoc.transform(d2, 2)
    for t, r in ((t1, r1), (t2, r2)):
        assert np.array_equal(t["bow_ids"], r["bow_ids"]) and np.array_equal(t["bow_vals"], r["bow_vals"])
        assert all(np.array_equal(a, b) for a, b in zip(t["fv"], r["fv"]))
    valid = (np.random.default_rng(seed).random(len(k1)) < 0.9).astype(np.uint8)
    got = f1.SearchByBoW_KF_F(valid, t1["fv"], f2, t2["fv"], 0.7, True)
    ref = ol.search_by_bow_kf_f(d1, k1["angle"], valid, r1["fv"], d2, k2["angle"], r2["fv"], 0.7, True)
    assert ref[0] > 20 and got[0] == ref[0] and np.array_equal(got[1], ref[1])
    v2 = np.ones(len(k2), np.uint8)
    got = f1.SearchByBoW_KF_KF(valid, t1["fv"], f2, v2, t2["fv"], 0.8, True)
    ref = ol.search_by_bow_kf_kf(d1, k1["angle"], valid, r1["fv"], d2, k2["angle"], v2, r2["fv"], 0.8, True)
    assert ref[0] > 20 and got[0] == ref[0] and np.array_equal(got[1], ref[1])


@pytest.mark.parametrize("seed,p_ok,ori", [(1, 0.6, True), (2, None, False), (3, 1.0, True)])
def test_search_for_triangulation_resident(seed, p_ok, ori):
    """ORBmatcher::SearchForTriangulation (ORBmatcher.cc:902-1146) with both KeyFrames resident == the host-array form
    == the oracle."""
    from test_gpu_match import near_duplicates, shared_pair_bits
    blob = synth.synthetic_vocabulary(k=10, L=3, seed=seed)
    voc, ovoc = orb.ORBVocabulary(blob), ol.OracleVocabulary(blob)
    k1, d1 = sc.features(seed, 0)
    rng = np.random.default_rng(40 + seed)
    perm = rng.permutation(len(d1))
    k2, d2 = k1[perm], near_duplicates(d1[perm], rng, 3)
    f1, f2 = _frame(k1, d1), _frame(k2, d2)
    fv1, fv2 = f1.ComputeBoW(voc, 2)["fv"], f2.ComputeBoW(voc, 2)["fv"]
    assert all(np.array_equal(a, b) for a, b in zip(fv1 + fv2, ovoc.transform(d1, 2)["fv"] + ovoc.transform(d2, 2)["fv"]))
    e1 = (rng.random(len(d1)) > 0.2).astype(np.uint8)
    e2 = (rng.random(len(d2)) > 0.2).astype(np.uint8)
    ok, off = shared_pair_bits(fv1, fv2, rng, p_ok) if p_ok is not None else (None, None)
    got = f1.SearchForTriangulation(e1, fv1, f2, e2, fv2, ori, ok, off)
    host = orb.ORBmatcher(0.6, ori).SearchForTriangulation(d1, k1["angle"], e1, fv1, d2, k2["angle"], e2, fv2, ok, off)
    ref = ol.search_for_triangulation(d1, k1["angle"], e1, fv1, d2, k2["angle"], e2, fv2, ok, off, ori)
    assert ref[0] > 20 and got[0] == ref[0] == host[0] and np.array_equal(got[1], ref[1]) and np.array_equal(host[1], ref[1])


def test_created_but_never_uploaded_frame_is_empty():
    """vsg_frame_create zeroes the device block: searching a frame that holds nothing yet finds nothing (instead of
    walking whatever cell_start[] the allocation contained)."""
    s = sc.last_frame_scenario(1)
    f = orb.Frame(1200)
    assert f.N == 0
    cs, en = f.grid()
    assert not cs.any() and len(en) == 0
    off, idx = f.GetFeaturesInArea(s["u"][:50], s["v"][:50], np.full(50, 30.0, np.float32))
    assert not off.any() and len(idx) == 0
    nm, tm, _ = f.SearchByProjection_Last(s["q_desc"], s["observed"], s["u"], s["v"], s["ur"], s["octave"], s["angle"],
                                          s["th"], s["direction"], sc.SCALE_FACTORS, True, np.zeros(0, np.uint8))
    assert nm == 0 and len(tm) == 0
    k = sc.kf_projection_scenario(2)
    nm, m = f.SearchByProjection_Sim3(k["q_desc"], k["u"], k["v"], k["radius"], k["level"], 1.0, np.zeros(0, np.int32))
    assert nm == 0
    nf, bi, bd = f.Fuse_Sim3(k["q_desc"], k["u"], k["v"], k["radius"], k["level"])
    assert nf == 0 and np.all(bi == -1)


def test_frame_upload_rejects_octaves_the_packed_entries_cannot_hold():
    k, d = sc.features(1, 0)
    k = k.copy()
    k["octave"][3] = 16
    with pytest.raises(orb.VsgError) as e:
        orb.Frame(len(k) + 4).upload(k, d, sc.BOUNDS)
    assert e.value.code == -3


def test_stereo_matches_resident():
    exl, exr = orb.ORBextractor(1200, 1.2, 8, 20, 7), orb.ORBextractor(1200, 1.2, 8, 20, 7)
    rl, rr = ol.OracleExtractor(1200, 1.2, 8, 20, 7), ol.OracleExtractor(1200, 1.2, 8, 20, 7)
    from test_gpu_stereo import rectified_pair
    L_, R_ = rectified_pair(752, 480, 77, 17)
    (_, kl, dl), (_, kr, dr) = exl(L_), exr(R_)
    rl(L_), rr(R_)
    b = (0.0, 0.0, 752.0, 480.0)
    fl = orb.Frame(exl.capacity(480, 752)).from_extractor(exl, 0, kl, b)
    fr = orb.Frame(exr.capacity(480, 752)).from_extractor(exr, 0, kr, b)
    ur, dep = orb.ComputeStereoMatches_resident(exl, 0, exr, 0, fl, fr, 0.11, 47.9)
    our, odep = ol.stereo_matches(rl, rr, kl, dl, kr, dr, 0.11, 47.9)
    assert (our >= 0).sum() > 100
    assert ur.tobytes() == our.tobytes() and dep.tobytes() == odep.tobytes()
    ur2, dep2 = orb.ComputeStereoMatches(exl, 0, exr, 0, kl, dl, kr, dr, 0.11, 47.9)  # host-array form
    assert ur2.tobytes() == our.tobytes() and dep2.tobytes() == odep.tobytes()


# ---------------------------------------------------------------------------------------------- threads / arenas
def test_steady_state_allocates_nothing():
    s = sc.last_frame_scenario(2)
    f = _frame(s["keys"], s["desc"], s["u_right"])
    args = (s["q_desc"], s["observed"], s["u"], s["v"], s["ur"], s["octave"], s["angle"], s["th"], s["direction"],
            sc.SCALE_FACTORS, True, s["blocked"])
    first = f.SearchByProjection_Last(*args)
    g0 = orb.thread_arena_growths(0)
    for _ in range(20):
        again = f.SearchByProjection_Last(*args)
        orb.ORBmatcher(0.7, True).block_best2(s["q_desc"][:200], s["desc"][:200])
    assert orb.thread_arena_growths(0) == g0  # no hipMalloc / hipHostMalloc after the first calls
    assert again[0] == first[0] and np.array_equal(again[1], first[1])


def test_concurrent_host_threads_share_frames():
    """Tracking / LocalMapping / LoopClosing call the matcher concurrently: per-thread streams + arenas,
    immutable shared frames."""
    s = sc.last_frame_scenario(1)
    k = sc.kf_projection_scenario(2)
    f = _frame(s["keys"], s["desc"], s["u_right"])
    fk = _frame(k["keys"], k["desc"])
    o = ol.OracleFrame(s["keys"], s["desc"], sc.BOUNDS, s["u_right"])
    ok = ol.OracleFrame(k["keys"], k["desc"], sc.BOUNDS)
    args = (s["q_desc"], s["observed"], s["u"], s["v"], s["ur"], s["octave"], s["angle"], s["th"], s["direction"],
            sc.SCALE_FACTORS, True, s["blocked"])
    ref_a = o.search_by_projection_last(*args)
    m0 = np.full(len(k["keys"]), -1, np.int32)
    ref_b = ok.search_by_projection_sim3(k["q_desc"], k["u"], k["v"], k["radius"], k["level"], 1.0, m0)
    errors = []

    def worker(kind):
        try:
            for _ in range(30):
                if kind == 0:
                    g = f.SearchByProjection_Last(*args)
                    assert g[0] == ref_a[0] and np.array_equal(g[1], ref_a[1])
                else:
                    g = fk.SearchByProjection_Sim3(k["q_desc"], k["u"], k["v"], k["radius"], k["level"], 1.0, m0)
                    assert g[0] == ref_b[0] and np.array_equal(g[1], ref_b[1])
        except Exception as e:  # noqa: BLE001
            errors.append(e)
    ts = [threading.Thread(target=worker, args=(i % 2,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors


# ---------------------------------------------------------------------------------------------- edge cases
def test_empty_frames_zero_queries_and_wild_coordinates():
    k, d = sc.features(1, 1)
    f, o = _frame(k, d), ol.OracleFrame(k, d, sc.BOUNDS)
    empty = orb.Frame(8).upload(np.zeros(0, orb.KP_DTYPE), np.zeros((0, 32), np.uint8), sc.BOUNDS)
    z = np.zeros(0, np.float32)
    zi = np.zeros(0, np.int32)
    zd = np.zeros((0, 32), np.uint8)
    # no queries
    n, m = f.SearchByProjection_Sim3(zd, z, z, z, zi, 1.0, np.full(len(k), -1, np.int32))
    assert n == 0 and np.all(m == -1)
    assert f.Fuse_Sim3(zd, z, z, z, zi)[0] == 0
    # queries against a frame without features
    q = sc.kf_projection_scenario(1)
    n, m = empty.SearchByProjection_Sim3(q["q_desc"], q["u"], q["v"], q["radius"], q["level"], 1.0, np.zeros(0, np.int32))
    assert n == 0
    nf, bi, bd = empty.Fuse_Sim3(q["q_desc"], q["u"], q["v"], q["radius"], q["level"])
    assert nf == 0 and np.all(bi == -1) and np.all(bd == 0x7FFFFFFF)
    assert empty.SearchForInitialization(f, z, z, 100, 0.9, True)[0] == 0
    # NaN / infinite / huge coordinates and radii: empty windows on both sides, like the reference's int conversions
    wild = np.array([np.nan, np.inf, -np.inf, 1e12, -1e12, 3e38, 50.0, 50.0, 50.0], np.float32)
    rad = np.array([10, 10, 10, 10, 10, 10, np.nan, np.inf, 1e30], np.float32)
    lv = np.zeros(len(wild), np.int32)
    off, idx = f.GetFeaturesInArea(wild, np.full(len(wild), 60.0, np.float32), rad)
    for i in range(len(wild)):
        assert np.array_equal(idx[off[i]:off[i + 1]], o.features_in_area(wild[i], 60.0, rad[i], kf_form=True)), i
    qd = np.tile(d[:1], (len(wild), 1))
    got = f.SearchByProjection_Sim3(qd, wild, np.full(len(wild), 60.0, np.float32), rad, lv + 3, 1.0,
                                    np.full(len(k), -1, np.int32))
    ref = o.search_by_projection_sim3(qd, wild, np.full(len(wild), 60.0, np.float32), rad, lv + 3, 1.0,
                                      np.full(len(k), -1, np.int32))
    assert got[0] == ref[0] and np.array_equal(got[1], ref[1])


def test_every_keypoint_in_one_cell_and_outside_the_grid():
    """All features in a single grid cell (a cell with far more than 32 entries takes the kernel's fallback path) and
    features that PosInGrid rejects (undistorted coordinates outside the bounds, Frame.cc:877-878)."""
    rng = np.random.default_rng(3)
    n = 300
    k = np.zeros(n, orb.KP_DTYPE)
    k["x"] = rng.uniform(100.0, 104.0, n)
    k["y"] = rng.uniform(100.0, 104.0, n)
    k["octave"] = rng.integers(0, 4, n)
    k["x"][:20] = rng.uniform(-50, -10, 20)  # outside the grid: never indexed
    k["y"][20:40] = rng.uniform(300, 400, 20)
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    f, o = _frame(k, d), ol.OracleFrame(k, d, sc.BOUNDS)
    for right in (False,):
        cs, en = f.grid(right)
        ocs, oen = o.grid(right)
        assert np.array_equal(cs, ocs) and np.array_equal(en, oen) and len(en) == n - 40
    q = rng.integers(0, 256, (64, 32), dtype=np.uint8)
    u = rng.uniform(90, 115, 64).astype(np.float32)
    v = rng.uniform(90, 115, 64).astype(np.float32)
    r = rng.uniform(1, 30, 64).astype(np.float32)
    lvl = rng.integers(0, 4, 64).astype(np.int32)
    m0 = np.full(n, -1, np.int32)
    got = f.SearchByProjection_Sim3(q, u, v, r, lvl, 5.0, m0)
    ref = o.search_by_projection_sim3(q, u, v, r, lvl, 5.0, m0)
    assert ref[0] > 5 and got[0] == ref[0] and np.array_equal(got[1], ref[1])
    off, idx = f.GetFeaturesInArea(u, v, r)
    assert max(np.diff(off)) > 64  # lists longer than the inline slot and than 32 entries per cell
    for i in range(64):
        assert np.array_equal(idx[off[i]:off[i + 1]], o.features_in_area(u[i], v[i], r[i], kf_form=True))
