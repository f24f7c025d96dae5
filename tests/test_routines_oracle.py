"""CPU tests of the routine-level oracle (oracle/routines_oracle.cpp): the second, more literal restatement of the
ORBmatcher projection searches is checked against the first one (oracle/match_oracle.cpp: candidate lists in, same
reference lines) on seeded scenarios, and against hand-made known answers for the reference's quirks.  The host-only
pieces of the C ABI (vsg_fuse_decide) are checked here as well: they need no GPU."""
import numpy as np
import pytest

import oracle_lib as ol
import scenarios as sc
from visual_sgraphs_amd import orb


@pytest.fixture(autouse=True, params=sc.CAMERA_NAMES)
def camera(request):
    """The two restatements are compared under three cameras: no distortion, TUM1 and D435i (mvKeysUn + the fractional
    bounds of Frame::ComputeImageBounds; see scenarios.use_camera)."""
    sc.use_camera(request.param)
    yield request.param
    sc.use_camera("image")


def _lists(grid, xs, ys, rs, lo=None, hi=None):
    off, idx = [0], []
    for i in range(len(xs)):
        c = grid.query(xs[i], ys[i], rs[i], -1 if lo is None else lo[i], -1 if hi is None else hi[i])
        idx.extend(c.tolist())
        off.append(len(idx))
    return np.array(off, np.int32), np.array(idx, np.int32)


@pytest.mark.parametrize("seed", [1, 2])
def test_frame_grid_and_windows_match_first_restatement(seed):
    keys, desc = sc.features(seed, 1)
    fr = ol.OracleFrame(keys, desc, sc.BOUNDS)
    g = ol.OracleGrid(keys, *sc.BOUNDS)
    rng = np.random.default_rng(seed)
    cs, en = fr.grid()
    if sc.CAMERA != "tum1":
        assert cs[-1] == len(en) == len(keys)  # every keypoint of these frames lies inside the grid
    else:  # TUM1's bounds lie INSIDE the image: undistorted keypoints near the edge are dropped by PosInGrid (Frame.cc:877)
        assert cs[-1] == len(en) <= len(keys)
    assert np.all(np.diff(cs) >= 0)
    for _ in range(200):
        x, y, r = rng.uniform(-20, 340), rng.uniform(-20, 260), rng.uniform(1, 60)
        lo, hi = int(rng.integers(-1, 8)), int(rng.integers(-1, 8))
        assert np.array_equal(fr.features_in_area(x, y, r, lo, hi), g.query(x, y, r, lo, hi))
        assert np.array_equal(fr.features_in_area(x, y, r, kf_form=True), g.query(x, y, r, -1, -1))


def test_stereo_frame_grids_split_at_nleft():
    keys, desc, nleft = sc.stereo_pair(3)
    fr = ol.OracleFrame(keys, desc, sc.BOUNDS, nleft=nleft)
    csl, enl = fr.grid(False)
    csr, enr = fr.grid(True)
    assert len(enl) == nleft and len(enr) == len(keys) - nleft
    assert enr.max() < len(keys) - nleft  # right entries are i - Nleft (Frame.cc:549)
    gl = ol.OracleGrid(keys[:nleft], *sc.BOUNDS)
    gr = ol.OracleGrid(keys[nleft:], *sc.BOUNDS)
    assert np.array_equal(fr.features_in_area(160, 120, 40, 1, 3), gl.query(160, 120, 40, 1, 3))
    assert np.array_equal(fr.features_in_area(160, 120, 40, 1, 3, right=True), gr.query(160, 120, 40, 1, 3))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_last_frame_routine_equals_list_form(seed):
    s = sc.last_frame_scenario(seed)
    fr = ol.OracleFrame(s["keys"], s["desc"], sc.BOUNDS)  # no mvuRight: the list form leaves that gate to its caller
    n, tm, tb = fr.search_by_projection_last(s["q_desc"], s["observed"], s["u"], s["v"], None, s["octave"], s["angle"],
                                             s["th"], s["direction"], sc.SCALE_FACTORS, True, s["blocked"])
    g = ol.OracleGrid(s["keys"], *sc.BOUNDS)
    oct_ = s["octave"]
    d = s["direction"]
    lo = oct_ if d == 1 else np.zeros_like(oct_) if d == 2 else oct_ - 1
    hi = np.full_like(oct_, -1) if d == 1 else oct_ if d == 2 else oct_ + 1
    rad = (np.float32(s["th"]) * sc.SCALE_FACTORS[oct_]).astype(np.float32)
    off, idx = _lists(g, s["u"], s["v"], rad, lo, hi)
    n2, tm2, tb2 = ol.search_by_projection_last(s["q_desc"], s["angle"], s["observed"], off, idx, s["desc"],
                                                s["keys"]["angle"], s["blocked"], 100, True)
    assert n > 50 and n == n2 and np.array_equal(tm, tm2) and np.array_equal(tb, tb2)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_local_map_routine_equals_list_form(seed):
    s = sc.local_map_scenario(seed)
    mp = dict(s["mp"])
    mp["in_view"] = np.ones_like(mp["in_view"])  # the list form has no inactive queries
    fr = ol.OracleFrame(s["keys"], s["desc"], sc.BOUNDS)
    n, tm, tb = fr.search_by_projection(mp, s["th"], s["nnratio"], sc.SCALE_FACTORS, s["blocked"])
    g = ol.OracleGrid(s["keys"], *sc.BOUNDS)
    r = np.where(mp["view_cos"] > np.float32(0.998), np.float32(2.5), np.float32(4.0)).astype(np.float32)
    if s["th"] != 1.0:
        r = r * np.float32(s["th"])
    win = (r * sc.SCALE_FACTORS[mp["scale_level"]]).astype(np.float32)
    off, idx = _lists(g, mp["proj_x"], mp["proj_y"], win, mp["scale_level"] - 1, mp["scale_level"])
    n2, tm2, tb2 = ol.search_by_projection_local(mp["desc"], mp["observed"], off, idx, s["desc"],
                                                 s["keys"]["octave"], s["blocked"], s["nnratio"])
    assert n > 50 and n == n2 and np.array_equal(tm, tm2) and np.array_equal(tb, tb2)


def _kp(x, y, octave=0, angle=0.0):
    k = np.zeros(1, ol.KP_DTYPE)
    k["x"], k["y"], k["octave"], k["angle"], k["size"], k["class_id"] = x, y, octave, angle, 31, -1
    return k


def test_local_map_ratio_continue_skips_right_block():
    """ORBmatcher.cc:125-126: a failed ratio test on the LEFT block `continue`s past the right-camera block."""
    d0 = np.zeros((1, 32), np.uint8)
    d1 = d0.copy()
    d1[0, 0] = 0x01  # distance 1 to the map point
    d2 = d0.copy()
    d2[0, 0] = 0x03  # distance 2: second best on the same level -> 1 > 0.4 * 2 fails the ratio
    dr = d0.copy()   # right camera: a perfect match
    keys = np.concatenate([_kp(100, 100), _kp(102, 100), _kp(100, 100)])
    desc = np.concatenate([d1, d2, dr])
    fr = ol.OracleFrame(keys, desc, sc.BOUNDS, nleft=2)
    mp = dict(desc=d0, observed=[1], in_view=[1], proj_x=[101.0], proj_y=[100.0], proj_xr=[0.0], scale_level=[0],
              view_cos=[0.9], in_view_r=[1], proj_x_r=[100.0], proj_y_r=[100.0], scale_level_r=[0], view_cos_r=[0.9])
    args = (1.0, 0.4, sc.SCALE_FACTORS, np.zeros(3, np.uint8), np.full(2, -1, np.int32), np.full(1, -1, np.int32))
    n, tm, _ = fr.search_by_projection(mp, *args)
    assert n == 0 and np.all(tm == -1)  # the right block never ran
    mp["in_view"] = [0]  # without the left block the right one matches
    n, tm, _ = fr.search_by_projection(mp, *args)
    assert n == 1 and tm.tolist() == [-1, -1, 0]


def test_last_frame_empty_left_window_skips_right_block():
    """ORBmatcher.cc:1727-1728: `if (vIndices2.empty()) continue;` also skips the right-camera block (:1786)."""
    d0 = np.zeros((1, 32), np.uint8)
    keys = np.concatenate([_kp(50, 50), _kp(200, 200)])
    fr = ol.OracleFrame(keys, np.concatenate([d0, d0]), sc.BOUNDS, nleft=1)
    common = dict(last_octave=[0], last_angle=[0.0], th=7.0, direction=0, scale_factors=sc.SCALE_FACTORS,
                  check_ori=False, train_blocked=np.zeros(2, np.uint8))
    n, tm, _ = fr.search_by_projection_last(d0, [1], [120.0], [120.0], None, u_r=[200.0], v_r=[200.0], **common)
    assert n == 0 and tm.tolist() == [-1, -1]
    n, tm, _ = fr.search_by_projection_last(d0, [1], [50.0], [50.0], None, u_r=[200.0], v_r=[200.0], **common)
    assert n == 2 and tm.tolist() == [0, 0]


def test_last_frame_rotation_drop_frees_the_feature():
    """ORBmatcher.cc:1869: the rotation filter sets mvpMapPoints[i] = NULL -- the feature is not blocked any more."""
    rng = np.random.default_rng(5)
    nk = 12
    keys = np.concatenate([_kp(20 + 20 * i, 100, 0, 0.0) for i in range(nk)])
    desc = rng.integers(0, 256, (nk, 32), dtype=np.uint8)
    fr = ol.OracleFrame(keys, desc, sc.BOUNDS)
    angle = np.zeros(nk, np.float32)
    angle[0] = 180.0  # the odd one out lands in a bin of its own, far below 10 % of the main bin... with 12 entries
    n, tm, tb = fr.search_by_projection_last(desc, np.ones(nk, np.uint8), keys["x"], keys["y"], None,
                                             np.zeros(nk, np.int32), angle, 7.0, 0, sc.SCALE_FACTORS, True,
                                             np.zeros(nk, np.uint8))
    assert tm[0] == -1 and tb[0] == 0 and n == nk - 1 and np.all(tb[1:] == 1)


@pytest.mark.parametrize("seed", [1, 2])
def test_sim3_projection_equals_window_core(seed):
    s = sc.kf_projection_scenario(seed)
    fr = ol.OracleFrame(s["keys"], s["desc"], sc.BOUNDS)
    matched = np.full(len(s["keys"]), -1, np.int32)
    matched[s["rng"].random(len(matched)) < 0.1] = 7777
    n, m = fr.search_by_projection_sim3(s["q_desc"], s["u"], s["v"], s["radius"], s["level"], 1.0, matched)
    # first restatement: or_search_window over level-filtered KeyFrame windows, blocked = already matched
    off, idx = [0], []
    for i in range(len(s["u"])):
        c = fr.features_in_area(s["u"][i], s["v"][i], s["radius"][i], kf_form=True)
        o = s["keys"]["octave"][c]
        idx.extend(c[(o >= s["level"][i] - 1) & (o <= s["level"][i])].tolist())
        off.append(len(idx))
    n2, qi, qd, tm, tb = ol.search_window(s["q_desc"], np.ones(len(s["u"]), np.uint8), np.array(off, np.int32),
                                          np.array(idx, np.int32), s["desc"], (matched != -1).astype(np.uint8), 50)
    assert n > 20 and n == n2
    new = m != matched
    assert np.array_equal(np.nonzero(new)[0], np.nonzero(tm >= 0)[0]) and np.array_equal(m[new], tm[tm >= 0])


def test_fuse_decide_matches_oracle_and_known_answers():
    # slots: 0 empty, 1 holds mp 5 (3 obs), 2 holds mp 6 (bad), 3 holds mp 7 (1 obs)
    slot = np.array([-1, 5, 6, 7], np.int32)
    obs = np.array([2, 2, 2, 2, 2, 3, 4, 1], np.int32)
    bad = np.array([0, 0, 0, 0, 0, 0, 1, 0], np.uint8)
    q = np.array([0, 1, 2, 3, 4], np.int32)
    bi = np.array([0, 1, 2, 3, 0], np.int32)  # the last query hits the slot the first one filled
    bd = np.array([10, 50, 20, 30, 51], np.int32)
    nf, act, oth, sm, ob, bd2 = orb.fuse_decide(q, bi, bd, False, slot, obs, bad)
    assert nf == 4 and act.tolist() == [1, 2, 4, 3, 0]
    assert oth.tolist() == [-1, 5, 6, 7, -1]
    assert sm.tolist() == [0, 5, 6, 3] and ob.tolist() == [3, 2, 2, 3, 2, 5, 4, 1] and bd2.tolist() == [0, 1, 0, 0, 0, 0, 1, 1]
    bd[4] = 40  # now the last query is a match too: it finds mp 0 (3 obs > 2) in slot 0 and is replaced by it
    nf, act, oth, sm, ob, bd2 = orb.fuse_decide(q, bi, bd, False, slot, obs, bad)
    assert nf == 5 and act[4] == 2 and oth[4] == 0 and bd2[4] == 1
    nf, act, oth, sm, _, _ = orb.fuse_decide(q, bi, bd, True, slot, obs, bad)  # Sim3 form: vpReplacePoint
    assert act.tolist() == [1, 5, 4, 5, 5] and sm.tolist() == [0, 5, 6, 7]


@pytest.mark.parametrize("seed", [1, 2])
def test_fuse_decide_equals_oracle_walk(seed):
    s = sc.kf_projection_scenario(seed)
    rng = s["rng"]
    fr = ol.OracleFrame(s["keys"], s["desc"], sc.BOUNDS, u_right=s["u_right"])
    nq, nk = len(s["u"]), len(s["keys"])
    n_mp = nq + nk
    slot = np.where(rng.random(nk) < 0.5, nq + np.arange(nk), -1).astype(np.int32)
    obs = rng.integers(1, 6, n_mp).astype(np.int32)
    bad = (rng.random(n_mp) < 0.1).astype(np.uint8)
    qmp = np.arange(nq, dtype=np.int32)
    for sim3 in (False, True):
        if sim3:
            nf, bi, bd, act, oth, sm, ob, b2 = fr.fuse_sim3(qmp, s["q_desc"], s["u"], s["v"], s["radius"], s["level"],
                                                            slot, obs, bad)
        else:
            nf, bi, bd, act, oth, sm, ob, b2 = fr.fuse(qmp, s["q_desc"], s["u"], s["v"], s["ur"], s["radius"],
                                                       s["level"], sc.INV_SIGMA2, slot, obs, bad)
        nf2, act2, oth2, sm2, ob2, b22 = orb.fuse_decide(qmp, bi, bd, sim3, slot, obs, bad)
        assert nf >= 10 and nf == nf2
        assert np.array_equal(act, act2) and np.array_equal(oth, oth2)
        assert np.array_equal(sm, sm2) and np.array_equal(ob, ob2) and np.array_equal(b2, b22)
        assert len(set(act.tolist())) >= 3  # add, replace and none all occur
