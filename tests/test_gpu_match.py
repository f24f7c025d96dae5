"""GPU parity tests of the Hamming searches (ORBmatcher) through the C ABI against the CPU oracle:
index pairs and distances identical, including tie-breaking and the greedy, order-dependent state."""
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as ol
from visual_sgraphs_amd import orb, synth

pytestmark = pytest.mark.gpu

GOLDEN = np.load(Path(__file__).parent / "golden" / "orb_golden_v1.npz")


@pytest.fixture(scope="module")
def frames():
    """Two consecutive synthetic frames extracted by the oracle (matcher inputs)."""
    e = ol.OracleExtractor(1000, 1.2, 8, 20, 7)
    _, k0, d0 = e(synth.sequence_frame(640, 480, 3, 0))
    _, k1, d1 = e(synth.sequence_frame(640, 480, 3, 1))
    return (k0, d0), (k1, d1)


def make_fv(n, n_nodes, rng, drop=0.0):
    """Synthetic DBoW2 FeatureVector: node id per feature -> CSR with ascending node ids and, inside a node,
    ascending feature indices (FeatureVector::addFeature appends in feature order)."""
    node = rng.integers(0, n_nodes, n) * 7 + 3
    keep = rng.random(n) >= drop
    ids = np.unique(node[keep])
    off = [0]
    idx = []
    for i in ids:
        m = np.nonzero((node == i) & keep)[0]
        idx += m.tolist()
        off.append(len(idx))
    return ids.astype(np.int32), np.array(off, np.int32), np.array(idx, np.int32)


def near_duplicates(desc, rng, flips):
    """Descriptors at small Hamming distance from `desc` (so ratio tests and ties actually trigger)."""
    out = desc.copy()
    for r in range(len(out)):
        for b in rng.integers(0, 256, flips):
            out[r, b >> 3] ^= np.uint8(1 << (b & 7))
    return out


def test_descriptor_distance_kats_and_random():
    m = orb.ORBmatcher()
    z = np.zeros(32, np.uint8)
    assert m.DescriptorDistance(z, z) == 0 and m.DescriptorDistance(z, ~z) == 256
    rng = np.random.default_rng(0)
    a, b = rng.integers(0, 256, (2, 500, 32), dtype=np.uint8)
    want = np.unpackbits(a ^ b, axis=1).sum(axis=1)
    assert np.array_equal(m.DescriptorDistance(a, b), want)
    ia, ib = rng.integers(0, 500, (2, 3000))
    got = m.hamming_pairs(a, b, ia, ib)
    assert np.array_equal(got, np.unpackbits(a[ia] ^ b[ib], axis=1).sum(axis=1))
    assert got.tolist()[:50] == [ol.descriptor_distance(a[i], b[j]) for i, j in zip(ia[:50], ib[:50])]


def test_block_best2_real_frames_and_golden(frames):
    (k0, d0), (k1, d1) = frames
    m = orb.ORBmatcher()
    got = m.block_best2(d0, d1)
    want = ol.block_best2(d0, d1)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    g = m.block_best2(GOLDEN["match/desc0"], GOLDEN["match/desc1"])
    assert np.array_equal(g[0], GOLDEN["match/best"]) and np.array_equal(g[1], GOLDEN["match/second"])
    assert np.array_equal(g[2], GOLDEN["match/arg"])


def test_block_best2_ties_and_ragged_sizes():
    rng = np.random.default_rng(1)
    m = orb.ORBmatcher()
    for na, nb in [(1, 1), (3, 700), (257, 256), (256, 257), (1000, 3), (513, 1025)]:
        # few distinct rows => many exact ties; first index must win
        base = rng.integers(0, 256, (5, 32), dtype=np.uint8)
        a = base[rng.integers(0, 5, na)]
        b = near_duplicates(base[rng.integers(0, 5, nb)], rng, 1)
        for g, w in zip(m.block_best2(a, b), ol.block_best2(a, b)):
            assert np.array_equal(g, w), (na, nb)
    best, second, arg = m.block_best2(np.zeros((4, 32), np.uint8), np.zeros((0, 32), np.uint8))
    assert best.tolist() == [256] * 4 and second.tolist() == [256] * 4 and arg.tolist() == [-1] * 4


@pytest.mark.parametrize("seed,n_nodes,ratio,ori", [(0, 60, 0.7, True), (1, 8, 0.75, True), (2, 300, 0.9, False),
                                                    (3, 1, 0.6, True)])
def test_search_by_bow_kf_f(frames, seed, n_nodes, ratio, ori):
    (k0, d0), (k1, d1) = frames
    rng = np.random.default_rng(seed)
    # F descriptors = KF descriptors with a few bit flips + shuffled, so many candidates pass TH_LOW / the ratio test
    perm = rng.permutation(len(d0))
    f_desc = near_duplicates(d0[perm], rng, 6)
    f_angle = (k0["angle"][perm] + rng.choice([0, 0, 0, 45, 200], len(perm))).astype(np.float32) % np.float32(360)
    kf_valid = (rng.random(len(d0)) > 0.2).astype(np.uint8)
    # same vocabulary node for a feature and its perturbed copy (most of the time)
    node_of_kf = rng.integers(0, n_nodes, len(d0))
    node_of_f = node_of_kf[perm].copy()
    flip = rng.random(len(perm)) < 0.1
    node_of_f[flip] = rng.integers(0, n_nodes, flip.sum())

    def csr(node):
        ids = np.unique(node)
        idx = np.concatenate([np.nonzero(node == i)[0] for i in ids])
        off = np.concatenate([[0], np.cumsum([np.sum(node == i) for i in ids])])
        return ids.astype(np.int32), off.astype(np.int32), idx.astype(np.int32)

    kf_fv, f_fv = csr(node_of_kf), csr(node_of_f)
    m = orb.ORBmatcher(ratio, ori)
    n_got, got = m.SearchByBoW_KF_F(d0, k0["angle"], kf_valid, kf_fv, f_desc, f_angle, f_fv)
    n_want, want = ol.search_by_bow_kf_f(d0, k0["angle"], kf_valid, kf_fv, f_desc, f_angle, f_fv, ratio, ori)
    assert n_got == n_want and np.array_equal(got, want)
    assert n_want > 50


@pytest.mark.parametrize("seed,n_nodes", [(0, 40), (1, 5), (2, 200)])
def test_search_by_bow_kf_kf(frames, seed, n_nodes):
    (k0, d0), (k1, d1) = frames
    rng = np.random.default_rng(100 + seed)
    perm = rng.permutation(len(d0))
    d2 = near_duplicates(d0[perm], rng, 5)
    a2 = k0["angle"][perm]
    v1 = (rng.random(len(d0)) > 0.15).astype(np.uint8)
    v2 = (rng.random(len(d2)) > 0.15).astype(np.uint8)
    fv1 = make_fv(len(d0), n_nodes, rng, drop=0.05)
    node2 = rng.integers(0, n_nodes, len(d2))
    fv2 = make_fv(len(d2), n_nodes, rng)
    m = orb.ORBmatcher(0.8, True)
    n_got, got = m.SearchByBoW_KF_KF(d0, k0["angle"], v1, fv1, d2, a2, v2, fv2)
    n_want, want = ol.search_by_bow_kf_kf(d0, k0["angle"], v1, fv1, d2, a2, v2, fv2, 0.8, True)
    assert n_got == n_want and np.array_equal(got, want)


def window_candidates(kps_q, kps_t, radius, rng, level_window=True):
    """Candidate lists exactly as the C++ adaptor would build them: Frame::GetFeaturesInArea on the grid."""
    grid = ol.OracleGrid(kps_t, 0.0, 0.0, 640.0, 480.0)
    off, idx = [0], []
    for kp in kps_q:
        lo, hi = (int(kp["octave"]) - 1, int(kp["octave"]) + 1) if level_window else (-1, -1)
        c = grid.query(kp["x"] + 3.0, kp["y"] + 2.0, radius * (1.2 ** int(kp["octave"])), lo, hi)
        idx += c.tolist()
        off.append(len(idx))
    return np.array(off, np.int32), np.array(idx, np.int32)


@pytest.mark.parametrize("seed,ori,th", [(0, True, 100), (1, False, 100), (2, True, 60)])
def test_search_by_projection_last(frames, seed, ori, th):
    (k0, d0), (k1, d1) = frames
    rng = np.random.default_rng(seed)
    nq = 700
    q = rng.choice(len(d0), nq, replace=False)
    q.sort()
    cand_off, cand_idx = window_candidates(k0[q], k1, 15.0, rng)
    q_blocks = (rng.random(nq) > 0.3).astype(np.uint8)       # some temporal points with 0 observations
    t_blocked = (rng.random(len(d1)) < 0.05).astype(np.uint8)
    m = orb.ORBmatcher(0.9, ori)
    n_got, tm_got, tb_got = m.SearchByProjection_Last(d0[q], k0["angle"][q], q_blocks, cand_off, cand_idx, d1,
                                                      k1["angle"], t_blocked, th)
    n_want, tm_want, tb_want = ol.search_by_projection_last(d0[q], k0["angle"][q], q_blocks, cand_off, cand_idx, d1,
                                                            k1["angle"], t_blocked, th, ori)
    assert n_got == n_want and np.array_equal(tm_got, tm_want) and np.array_equal(tb_got, tb_want)
    assert n_want > 100


@pytest.mark.parametrize("seed,ratio", [(0, 0.8), (1, 0.6)])
def test_search_by_projection_local(frames, seed, ratio):
    (k0, d0), (k1, d1) = frames
    rng = np.random.default_rng(10 + seed)
    q = np.sort(rng.choice(len(d0), 800, replace=False))
    cand_off, cand_idx = window_candidates(k0[q], k1, 12.0, rng)
    q_blocks = (rng.random(len(q)) > 0.2).astype(np.uint8)
    t_blocked = (rng.random(len(d1)) < 0.1).astype(np.uint8)
    m = orb.ORBmatcher(ratio, True)
    n_got, tm_got, tb_got = m.SearchByProjection_Local(d0[q], q_blocks, cand_off, cand_idx, d1, k1["octave"], t_blocked)
    n_want, tm_want, tb_want = ol.search_by_projection_local(d0[q], q_blocks, cand_off, cand_idx, d1, k1["octave"],
                                                             t_blocked, ratio)
    assert n_got == n_want and np.array_equal(tm_got, tm_want) and np.array_equal(tb_got, tb_want)
    assert n_want > 100


@pytest.mark.parametrize("seed,ori", [(0, True), (1, False)])
def test_search_for_initialization(frames, seed, ori):
    (k0, d0), (k1, d1) = frames
    rng = np.random.default_rng(20 + seed)
    grid = ol.OracleGrid(k1, 0.0, 0.0, 640.0, 480.0)
    off, idx = [0], []
    for kp in k0:
        if kp["octave"] == 0:
            idx += grid.query(kp["x"] + 3.0, kp["y"] + 2.0, 30.0, 0, 0).tolist()  # windowSize, level1, level1
        off.append(len(idx))
    m = orb.ORBmatcher(0.9, ori)
    n_got, got = m.SearchForInitialization(d0, k0["angle"], k0["octave"], off, idx, d1, k1["angle"])
    n_want, want = ol.search_for_initialization(d0, k0["angle"], k0["octave"], off, idx, d1, k1["angle"], 0.9, ori)
    assert n_got == n_want and np.array_equal(got, want)
    assert n_want > 30


def test_matchers_on_empty_inputs():
    m = orb.ORBmatcher(0.7, True)
    e32 = np.zeros((0, 32), np.uint8)
    fv0 = (np.zeros(0, np.int32), np.zeros(1, np.int32), np.zeros(0, np.int32))
    assert m.SearchByBoW_KF_F(e32, [], [], fv0, e32, [], fv0)[0] == 0
    n, tm, tb = m.SearchByProjection_Last(e32, [], [], [0], [], np.zeros((5, 32), np.uint8), np.zeros(5), np.zeros(5))
    assert n == 0 and tm.tolist() == [-1] * 5


# ----------------------------------------------------------------------------- the Frame grid on the device
def _camera_view(name, keys, w=640, h=480):
    """(mvKeysUn, bounds) of `keys` seen through camera `name`: "image" = no distortion, "tum1" / "d435i" = the BASELINE
    cameras whose Frame constructor undistorts (Frame.cc:891-955; C1 and C5)."""
    if name == "image":
        return keys, (0.0, 0.0, float(w), float(h))
    cam = ol.scaled_camera(name, w, h)
    return ol.undistort_keypoints(keys, cam), ol.image_bounds(cam)


@pytest.mark.parametrize("camera", ["image", "tum1", "d435i"])
@pytest.mark.parametrize("seed", [0, 1])
def test_device_grid_query_equals_reference_order(frames, seed, camera):
    (k0, d0), (k1, d1) = frames
    rng = np.random.default_rng(seed)
    k1, bounds = _camera_view(camera, k1)
    g_ref = ol.OracleGrid(k1, *bounds)
    g = orb.FrameGrid(k1, *bounds)
    nq = 400
    x = rng.uniform(-20, 660, nq).astype(np.float32)
    y = rng.uniform(-20, 500, nq).astype(np.float32)
    r = rng.uniform(1, 60, nq).astype(np.float32)
    lo = rng.integers(-1, 5, nq).astype(np.int32)
    hi = rng.integers(-1, 8, nq).astype(np.int32)
    off, idx = g.GetFeaturesInArea(x, y, r, lo, hi)
    for q in range(nq):
        want = g_ref.query(x[q], y[q], r[q], int(lo[q]), int(hi[q]))
        assert np.array_equal(idx[off[q]:off[q + 1]], want), q  # same elements in the same ORDER
    off2, idx2 = g.GetFeaturesInArea(x, y, r)  # KeyFrame::GetFeaturesInArea: no level filter
    for q in range(0, nq, 7):
        assert np.array_equal(idx2[off2[q]:off2[q + 1]], g_ref.query(x[q], y[q], r[q], -1, -1))
    assert off[-1] > 1000


@pytest.mark.parametrize("camera", ["image", "tum1", "d435i"])
def test_device_grid_edge_cases(camera):
    e32 = np.zeros(0, orb.KP_DTYPE)
    _, bounds = _camera_view(camera, e32)
    g = orb.FrameGrid(e32, *bounds)
    off, idx = g.GetFeaturesInArea([10.0, 700.0], [10.0, 10.0], [5.0, 5.0])
    assert off.tolist() == [0, 0, 0] and len(idx) == 0
    kp = np.zeros(3, orb.KP_DTYPE)
    kp["x"], kp["y"], kp["octave"] = [5.0, 5.0, 639.9], [5.0, 5.0, 479.9], [0, 3, 1]
    kp, bounds = _camera_view(camera, kp)  # TUM1: (5, 5) undistorts to outside the grid -- PosInGrid drops it
    g = orb.FrameGrid(kp, *bounds)
    ref = ol.OracleGrid(kp, *bounds)
    for q in [(5.0, 5.0, 1.0, -1, -1), (5.0, 5.0, 1.0, 2, 4), (639.0, 479.0, 3.0, -1, -1), (-50.0, 5.0, 10.0, -1, -1),
              (12.0, 15.0, 9.0, -1, -1), (bounds[0], bounds[1], 0.5, -1, -1), (bounds[2], bounds[3], 2.0, -1, -1)]:
        off, idx = g.GetFeaturesInArea([q[0]], [q[1]], [q[2]], [q[3]], [q[4]])
        assert idx.tolist() == ref.query(*q).tolist()


@pytest.mark.parametrize("seed,blocking", [(0, True), (1, False), (2, None)])
def test_search_window_generic(frames, seed, blocking):
    """Common core of the Sim3 / relocalisation projection searches, SearchBySim3 and Fuse (per-query best)."""
    (k0, d0), (k1, d1) = frames
    rng = np.random.default_rng(40 + seed)
    q = np.sort(rng.choice(len(d0), 600, replace=False))
    grid = orb.FrameGrid(k1, 0.0, 0.0, 640.0, 480.0)
    cand_off, cand_idx = grid.GetFeaturesInArea(k0["x"][q] + 3.0, k0["y"][q] + 2.0, np.full(len(q), 14.0, np.float32))
    if blocking is None:
        q_blocks, t_blocked = None, None
    else:
        q_blocks = np.full(len(q), 1 if blocking else 0, np.uint8)
        t_blocked = (rng.random(len(d1)) < 0.1).astype(np.uint8)
    th = 50 if seed else 100
    got = orb.search_window(d0[q], q_blocks, cand_off, cand_idx, d1, t_blocked, th)
    want = ol.search_window(d0[q], q_blocks, cand_off, cand_idx, d1, t_blocked, th)
    assert got[0] == want[0] > 50
    for a, b in zip(got[1:4], want[1:4]):
        assert np.array_equal(a, b)
    if blocking is not None:
        assert np.array_equal(got[4], want[4])


def test_distinctive_descriptors(frames):
    """MapPoint::ComputeDistinctiveDescriptors (N^2 Hamming + per-row median), many map points."""
    (k0, d0), (k1, d1) = frames
    rng = np.random.default_rng(77)
    sizes = [1, 2, 3, 4, 7, 0, 16, 33, 64, 100, 128, 5, 5, 2]
    groups, off = [], [0]
    for n in sizes:
        base = d0[rng.integers(0, len(d0))]
        g = near_duplicates(np.repeat(base[None], n, 0), rng, 4) if n else np.zeros((0, 32), np.uint8)
        if n >= 4:
            g[1] = g[0]  # exact duplicates: equal medians -> the first row must win
        groups.append(g)
        off.append(off[-1] + n)
    desc = np.concatenate(groups)
    got = orb.ComputeDistinctiveDescriptors(desc, off)
    want = ol.distinctive_descriptors(desc, off)
    assert np.array_equal(got, want)
    assert want[5] == -1


def shared_pair_bits(fv1, fv2, rng, p_ok):
    """Random geometric-predicate bits for every pair of every SHARED node, in the layout of
    vsg_search_for_triangulation: pair_off[s] + i1 * n2(s) + i2."""
    ids1, off1, _ = fv1
    ids2, off2, _ = fv2
    shared = np.intersect1d(ids1, ids2)
    pair_off = [0]
    for s in shared:
        a, b = int(np.searchsorted(ids1, s)), int(np.searchsorted(ids2, s))
        pair_off.append(pair_off[-1] + int(off1[a + 1] - off1[a]) * int(off2[b + 1] - off2[b]))
    nbits = pair_off[-1]
    bits = rng.random(max(nbits, 1)) < p_ok
    words = np.zeros((nbits + 31) // 32 + 1, np.uint32)
    for i in np.nonzero(bits[:nbits])[0]:
        words[i >> 5] |= np.uint32(1 << (i & 31))
    return words, np.array(pair_off, np.int32)


@pytest.mark.parametrize("seed,n_nodes,p_ok,ori", [(0, 40, 0.7, True), (1, 5, 0.3, True), (2, 200, 1.0, False),
                                                   (3, 60, None, True)])
def test_search_for_triangulation(frames, seed, n_nodes, p_ok, ori):
    """ORBmatcher::SearchForTriangulation (ORBmatcher.cc:902-1146): predicate bits from the adaptor, dist <= TH_LOW,
    LAST minimum wins -- the duplicated descriptors make equal distances common."""
    (k0, d0), _ = frames
    rng = np.random.default_rng(300 + seed)
    perm = rng.permutation(len(d0))
    d2 = near_duplicates(d0[perm], rng, 4)
    d2[::3] = d2[1::3][:len(d2[::3])]  # exact duplicates inside KF2: ties on the distance
    a2 = k0["angle"][perm]
    e1 = (rng.random(len(d0)) > 0.2).astype(np.uint8)
    e2 = (rng.random(len(d2)) > 0.2).astype(np.uint8)
    # a feature and its near-duplicate land in the same vocabulary node (as similar descriptors do)
    node1 = rng.integers(0, n_nodes, len(d0)) * 7 + 3

    def fv_of(node):
        ids = np.unique(node)
        off_, idx_ = [0], []
        for i in ids:
            idx_ += np.nonzero(node == i)[0].tolist()
            off_.append(len(idx_))
        return ids.astype(np.int32), np.array(off_, np.int32), np.array(idx_, np.int32)
    fv1, fv2 = fv_of(node1), fv_of(node1[perm])
    ok, off = shared_pair_bits(fv1, fv2, rng, p_ok) if p_ok is not None else (None, None)
    m = orb.ORBmatcher(0.6, ori)
    n_got, got = m.SearchForTriangulation(d0, k0["angle"], e1, fv1, d2, a2, e2, fv2, ok, off)
    n_want, want = ol.search_for_triangulation(d0, k0["angle"], e1, fv1, d2, a2, e2, fv2, ok, off, ori)
    assert n_got == n_want and np.array_equal(got, want)
    assert n_want > 20
    # matches only between eligible features of the same node, inside TH_LOW, with the predicate bit set
    for i1 in np.nonzero(want >= 0)[0][:50]:
        assert e1[i1] and e2[want[i1]] and ol.descriptor_distance(d0[i1], d2[want[i1]]) <= 50


def test_block_best2_tile_boundaries_of_the_mfma_kernel():
    """The matrix-core matcher works on 32-row train tiles, 32-query column blocks, 128-query workgroups and an
    8-tile prefetch group: sizes on and around every one of those boundaries, tie-heavy data."""
    rng = np.random.default_rng(5)
    m = orb.ORBmatcher()
    sizes = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 287, 288, 289, 511, 513]
    base = rng.integers(0, 256, (9, 32), dtype=np.uint8)
    for na in sizes:
        for nb in (sizes if na in (1, 33, 129, 257) else (1, 31, 33, 256, 289)):
            a = near_duplicates(base[rng.integers(0, 9, na)], rng, 2)
            b = near_duplicates(base[rng.integers(0, 9, nb)], rng, 1)
            for g, w in zip(m.block_best2(a, b), ol.block_best2(a, b)):
                assert np.array_equal(g, w), (na, nb)
    # all-equal descriptors: distance 0 everywhere, the first train row must win and the second-best is 0 too
    z = np.zeros((70, 32), np.uint8)
    best, second, arg = m.block_best2(z, z)
    assert not best.any() and not second.any() and not arg.any()
    # complementary descriptors: distance 256 (the accumulator's extreme value) is never "better than 256": like
    # the reference's strict '<' scan from bestDist = 256, no best index
    comp = np.full((40, 32), 255, np.uint8)
    for g, w in zip(m.block_best2(z[:5], comp), ol.block_best2(z[:5], comp)):
        assert np.array_equal(g, w)
    assert m.block_best2(z[:5], comp)[2].tolist() == [-1] * 5
    off = np.array([0, 40], np.int32)
    n, qi, qd, tm, _ = orb.search_window(z[:1], None, off, np.arange(40, dtype=np.int32), comp, None, 256)
    assert (n, qi.tolist(), qd.tolist()) == (0, [-1], [256])
    assert ol.search_window(z[:1], None, off, np.arange(40, dtype=np.int32), comp, None, 256)[:3][0] == 0


@pytest.mark.parametrize("seed,n_nodes,ratio,ori", [(0, 60, 0.7, True), (1, 8, 0.9, True), (2, 1, 0.6, False)])
def test_search_by_bow_kf_f_fisheye_stereo(frames, seed, n_nodes, ratio, ori):
    """SearchByBoW(KeyFrame*, Frame&) with F.Nleft != -1 (ORBmatcher.cc:277-326, 362-389): the frame holds the left
    camera's features followed by the right camera's; a KF feature can claim one of each, the right one without a
    ratio test but only if the left best is within TH_LOW."""
    (k0, d0), _ = frames
    rng = np.random.default_rng(400 + seed)
    n = len(d0)
    permL, permR = rng.permutation(n), rng.permutation(n)
    keepR = rng.random(n) < 0.7
    f_desc = np.concatenate([near_duplicates(d0[permL], rng, 6), near_duplicates(d0[permR][keepR], rng, 9)])
    f_angle = np.concatenate([k0["angle"][permL], k0["angle"][permR][keepR]]).astype(np.float32)
    nleft = n
    kf_valid = (rng.random(n) > 0.2).astype(np.uint8)
    node_of_kf = rng.integers(0, n_nodes, n)
    node_of_f = np.concatenate([node_of_kf[permL], node_of_kf[permR][keepR]])

    def csr(node):
        ids = np.unique(node)
        idx = np.concatenate([np.nonzero(node == i)[0] for i in ids])
        off = np.concatenate([[0], np.cumsum([np.sum(node == i) for i in ids])])
        return ids.astype(np.int32), off.astype(np.int32), idx.astype(np.int32)

    kf_fv, f_fv = csr(node_of_kf), csr(node_of_f)
    m = orb.ORBmatcher(ratio, ori)
    n_got, got = m.SearchByBoW_KF_F(d0, k0["angle"], kf_valid, kf_fv, f_desc, f_angle, f_fv, f_nleft=nleft)
    n_want, want = ol.search_by_bow_kf_f(d0, k0["angle"], kf_valid, kf_fv, f_desc, f_angle, f_fv, ratio, ori, nleft)
    assert n_got == n_want and np.array_equal(got, want)
    assert (want[:nleft] >= 0).sum() > 50 and (want[nleft:] >= 0).sum() > 50  # both cameras really matched
    # with Nleft = -1 the same arrays are one block: different result, and it equals the mono entry point
    n_m, got_m = m.SearchByBoW_KF_F(d0, k0["angle"], kf_valid, kf_fv, f_desc, f_angle, f_fv)
    n_w, want_m = ol.search_by_bow_kf_f(d0, k0["angle"], kf_valid, kf_fv, f_desc, f_angle, f_fv, ratio, ori)
    assert n_m == n_w and np.array_equal(got_m, want_m) and not np.array_equal(want_m, want)
