"""CPU tests pinning the oracle (oracle/) against self-derived known answers and
definition-level restatements written independently in numpy.
The reference ships no golden vectors for this path (parity unpinned), so these
are the strongest pins available without OpenCV 4.2."""
import hashlib
import re
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as ol
from visual_sgraphs_amd import synth

ROOT = Path(__file__).resolve().parent.parent


# ----------------------------------------------------------------------------- data tables
@pytest.mark.parametrize("path", ["oracle/brief_pattern_data.inc", "visual_sgraphs_amd/csrc/brief_pattern_data.inc"])
def test_pattern_table_sha256(path):
    txt = (ROOT / path).read_text()
    body = "\n".join(l for l in txt.splitlines() if not l.startswith("//"))
    vals = [int(v) for v in re.findall(r"-?\d+", body)]
    assert len(vals) == 1024
    assert vals[:4] == [8, -3, 9, 5] and vals[-4:] == [-1, -6, 0, -11]
    assert min(vals) == -13 and max(vals) <= 13
    blob = bytes((v + 256) % 256 for v in vals)
    assert hashlib.sha256(blob).hexdigest() == "2164181aea6ff9ac426ca512d5130d15e1f6e3cd47b1cbdd568bbe1e55d49023"


CONFIGS = {
    # name: (W, H, nfeatures, quotas, sizes)
    "C2": (640, 480, 1000, [217, 181, 151, 126, 105, 87, 73, 60],
           [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]),
    "C3": (752, 480, 1200, [261, 217, 181, 151, 126, 105, 87, 72],
           [(752, 480), (627, 400), (522, 333), (435, 278), (363, 231), (302, 193), (252, 161), (210, 134)]),
    "C4": (1280, 720, 2000, [434, 362, 302, 251, 209, 175, 145, 122],
           [(1280, 720), (1067, 600), (889, 500), (741, 417), (617, 347), (514, 289), (429, 241), (357, 201)]),
    "C5": (640, 480, 1250, [271, 226, 189, 157, 131, 109, 91, 76], None),
}


@pytest.mark.parametrize("name", list(CONFIGS))
def test_constructor_tables(name):
    w, h, nf, quotas, sizes = CONFIGS[name]
    e = ol.OracleExtractor(nf, 1.2, 8, 20, 7)
    t = e.tables()
    assert t["features_per_level"].tolist() == quotas
    assert t["umax"].tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    # float chain s[i] = float(s[i-1] * double(float(1.2)))
    s = [np.float32(1.0)]
    for _ in range(7):
        s.append(np.float32(np.float64(s[-1]) * np.float64(np.float32(1.2))))
    assert np.array_equal(t["scale"], np.array(s, np.float32))
    assert np.array_equal(t["inv_scale"], np.float32(1.0) / np.array(s, np.float32))
    assert np.array_equal(t["sigma2"], np.array(s, np.float32) ** 2)
    if sizes:
        e(synth.constant_frame(w, h))
        assert [e.level_size(l) for l in range(8)] == sizes


def test_cv_round_half_even():
    for v, r in [(0.5, 0), (1.5, 2), (2.5, 2), (-0.5, 0), (-1.5, -2), (2.4999, 2), (2.5001, 3), (-2.5, -2)]:
        assert ol.cv_round_f(v) == r


# ----------------------------------------------------------------------------- pyramid
def _np_resize_linear(src, dw, dh):
    """Independent numpy restatement of cv::resize INTER_LINEAR 8UC1."""
    sh, sw = src.shape

    def tables(dn, sn):
        scale = 1.0 / (np.float64(dn) / sn)
        d = np.arange(dn, dtype=np.float64)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        return s, f

    sx, fx = tables(dw, sw)
    sy, fy = tables(dh, sh)
    fx = np.where(sx < 0, np.float32(0), fx)
    sx = np.maximum(sx, 0)
    fx = np.where(sx >= sw - 1, np.float32(0), fx)
    sx = np.minimum(sx, sw - 1)
    sx1 = np.minimum(sx + 1, sw - 1)
    a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int64)
    a1 = np.rint(fx * np.float32(2048)).astype(np.int64)
    b0 = np.rint((np.float32(1) - fy) * np.float32(2048)).astype(np.int64)
    b1 = np.rint(fy * np.float32(2048)).astype(np.int64)
    assert np.all(a0 + a1 == 2048) and np.all(b0 + b1 == 2048)
    y0 = np.clip(sy, 0, sh - 1)
    y1 = np.clip(sy + 1, 0, sh - 1)
    S = src.astype(np.int64)
    H = S[:, sx] * a0[None, :] + S[:, sx1] * a1[None, :]
    out = (((b0[:, None] * (H[y0] >> 4)) >> 16) + ((b1[:, None] * (H[y1] >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


@pytest.mark.parametrize("shape", [(640, 480, 533, 400), (533, 400, 444, 333), (179, 134, 149, 112), (97, 61, 81, 51)])
def test_resize_matches_independent_numpy(shape):
    sw, sh, dw, dh = shape
    src = synth.frame(sw, sh, 3)
    assert np.array_equal(ol.resize_linear(src, dw, dh), _np_resize_linear(src, dw, dh))


def test_resize_constant_image_is_constant():
    for v in (0, 1, 127, 128, 254, 255):
        out = ol.resize_linear(np.full((100, 120), v, np.uint8), 100, 83)
        assert np.all(out == v)


def test_border_reflect101():
    src = np.arange(7 * 9, dtype=np.uint8).reshape(7, 9)
    out = ol.copy_make_border101(src, 3)
    ref = np.pad(src, 3, mode="reflect")  # numpy 'reflect' == BORDER_REFLECT_101
    assert np.array_equal(out, ref)


def test_pyramid_chain_and_border():
    e = ol.OracleExtractor(500, 1.2, 4, 20, 7)
    img = synth.frame(320, 240, 5)
    e(img)
    prev = img
    for l in range(4):
        w, h = e.level_size(l)
        lvl = e.pyramid_level(l)
        if l > 0:
            assert np.array_equal(lvl, _np_resize_linear(prev, w, h))  # chained, not from level 0
        else:
            assert np.array_equal(lvl, img)
        assert np.array_equal(e.pyramid_level(l, with_border=True), np.pad(lvl, 19, mode="reflect"))
        prev = lvl


# ----------------------------------------------------------------------------- FAST
RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
        (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def np_fast_score_map(img):
    """score(p) = max over the 16 contiguous 9-arcs of min(+-(v - ring)) - 1 by definition; <0 where not defined."""
    h, w = img.shape
    I = img.astype(np.int32)
    v = I[3:h - 3, 3:w - 3]
    d = np.stack([v - I[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in RING])  # 16 x H x W
    best = np.full(v.shape, -10 ** 6, np.int32)
    for s in range(16):
        idx = [(s + k) % 16 for k in range(9)]
        best = np.maximum(best, d[idx].min(axis=0))
        best = np.maximum(best, (-d[idx]).min(axis=0))
    score = np.full((h, w), -1, np.int32)
    score[3:h - 3, 3:w - 3] = best - 1
    return score


def np_fast(img, t, nonmax=True):
    h, w = img.shape
    s = np_fast_score_map(img)
    corner = s >= t  # corner at t <=> exists arc with all |d| > t <=> score >= t
    sc = np.where(corner, s, 0)
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            if not corner[y, x]:
                continue
            if nonmax:
                nb = sc[y - 1:y + 2, x - 1:x + 2].copy()
                nb[1, 1] = -1
                if not np.all(sc[y, x] > nb):
                    continue
            out.append((x, y, int(s[y, x]) if nonmax else 0))
    return out


@pytest.mark.parametrize("seed,t", [(0, 20), (1, 7), (2, 20), (3, 7), (4, 40)])
def test_fast_matches_definition(seed, t):
    img = synth.frame(64, 48, 100 + seed)
    x, y, s = ol.fast9_16(img, t, True)
    ref = np_fast(img, t, True)
    assert list(zip(x.tolist(), y.tolist(), s.tolist())) == ref
    assert len(ref) > 0


def test_fast_without_nonmax_and_tiny_images():
    img = synth.frame(40, 30, 7)
    x, y, _ = ol.fast9_16(img, 20, False)
    assert list(zip(x.tolist(), y.tolist())) == [(a, b) for a, b, _ in np_fast(img, 20, False)]
    for shp in [(6, 6), (7, 6), (6, 7), (1, 1)]:
        assert len(ol.fast9_16(np.zeros(shp, np.uint8), 7)[0]) == 0
    one = np.full((7, 7), 100, np.uint8)
    one[3, 3] = 200
    assert [tuple(int(v[0]) for v in ol.fast9_16(one, 20))] == [(3, 3, 99)]


def np_cell_candidates(level_img, ini_th, min_th):
    """Score-map formulation of the per-cell FAST of ComputeKeyPointsOctTree:
    global score map, NMS confined to each cell's valid region, per-cell threshold fallback."""
    h, w = level_img.shape
    smap = np_fast_score_map(level_img)
    minB, maxBX, maxBY = 16, w - 16, h - 16
    width, height = np.float32(maxBX - minB), np.float32(maxBY - minB)
    n_cols, n_rows = int(width / np.float32(35)), int(height / np.float32(35))
    w_cell, h_cell = int(np.ceil(width / n_cols)), int(np.ceil(height / n_rows))
    out = []
    for i in range(n_rows):
        ini_y = minB + i * h_cell
        max_y = min(ini_y + h_cell + 6, maxBY)
        if ini_y >= maxBY - 3:
            continue
        for j in range(n_cols):
            ini_x = minB + j * w_cell
            max_x = min(ini_x + w_cell + 6, maxBX)
            if ini_x >= maxBX - 6:
                continue
            y0, y1, x0, x1 = ini_y + 3, max_y - 3, ini_x + 3, max_x - 3  # valid region
            if y1 <= y0 or x1 <= x0:
                continue
            s = np.zeros((y1 - y0 + 2, x1 - x0 + 2), np.int32)  # zero frame = "outside the cell counts as 0"
            s[1:-1, 1:-1] = np.maximum(smap[y0:y1, x0:x1], 0)
            c = s[1:-1, 1:-1]
            is_max = np.ones_like(c, bool)
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    if dx or dy:
                        is_max &= c > s[1 + dy:s.shape[0] - 1 + dy, 1 + dx:s.shape[1] - 1 + dx]
            # below min_th scores are irrelevant: a survivor needs score >= th > any sub-threshold neighbour
            for th in (ini_th, min_th):
                ys, xs = np.nonzero(is_max & (c >= th))
                if len(ys):
                    out += [(x0 + xx - 16, y0 + yy - 16, int(c[yy, xx])) for yy, xx in zip(ys, xs)]
                    break
    return out


@pytest.mark.parametrize("div", [1, 8])
def test_cell_candidates_equal_score_map_formulation(div):
    """The GPU design computes one score map + cell-confined NMS + threshold fallback; prove on the CPU
    that this equals the reference's per-cell cv::FAST(20) else cv::FAST(7) (order included)."""
    e = ol.OracleExtractor(400, 1.2, 3, 20, 7)
    img = synth.frame(320, 240, 11, amplitude_div=div)
    e(img)
    used_fallback = False
    for l in range(3):
        x, y, r = e.candidates(l)
        ref = np_cell_candidates(e.pyramid_level(l), 20, 7)
        assert list(zip(x.tolist(), y.tolist(), r.tolist())) == ref
        used_fallback |= bool(np.any(r < 20))
    if div == 8:
        assert used_fallback


# ----------------------------------------------------------------------------- blur / angle / descriptor
def test_blur_separable_definition():
    img = synth.frame(61, 47, 9)
    taps = np.array(ol.DEFAULT_TAPS, np.int64)
    P = np.pad(img.astype(np.int64), 3, mode="reflect")
    H = sum(taps[k] * P[3:-3, k:k + 61] for k in range(7))
    Hp = np.pad(H, ((3, 3), (0, 0)), mode="reflect")
    V = sum(taps[k] * Hp[k:k + 47, :] for k in range(7))
    ref = np.minimum((V + 32768) >> 16, 255).astype(np.uint8)
    assert np.array_equal(ol.gaussian_blur7(img), ref)
    ed = (18, 34, 48, 56, 48, 34, 18)  # the sum-256 variant keeps constants constant
    assert np.all(ol.gaussian_blur7(np.full((20, 20), 77, np.uint8), ed) == 77)
    assert np.all(ol.gaussian_blur7(np.full((20, 20), 255, np.uint8)) == 255)  # saturating


def test_fast_atan2_accuracy_and_kats():
    rng = np.random.default_rng(1)
    for _ in range(2000):
        y, x = rng.integers(-40000, 40000, 2)
        if x == 0 and y == 0:
            continue
        a = float(ol.fast_atan2(y, x))
        ref = np.degrees(np.arctan2(float(y), float(x))) % 360.0
        assert min(abs(a - ref), 360 - abs(a - ref)) < 0.02
    assert ol.fast_atan2(0, 0) == 0.0
    assert ol.fast_atan2(0, 5) == 0.0
    assert abs(float(ol.fast_atan2(5, 0)) - 90.0) < 1e-4
    assert abs(float(ol.fast_atan2(0, -5)) - 180.0) < 1e-4
    assert abs(float(ol.fast_atan2(-5, 0)) - 270.0) < 1e-4


def test_ic_angle_moments():
    img = synth.frame(64, 64, 21)
    cx, cy = 30, 31
    umax = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    m10 = m01 = 0
    for v in range(-15, 16):
        d = umax[abs(v)]
        for u in range(-d, d + 1):
            p = int(img[cy + v, cx + u])
            m10 += u * p
            m01 += v * p
    assert ol.ic_angle(img, cx, cy) == ol.fast_atan2(m01, m10)
    assert sum(2 * u + 1 for u in umax) * 2 - (2 * umax[0] + 1) == 749


def test_descriptor_definition():
    txt = (ROOT / "oracle/brief_pattern_data.inc").read_text()
    body = "\n".join(l for l in txt.splitlines() if not l.startswith("//"))
    pat = np.array([int(v) for v in re.findall(r"-?\d+", body)], np.int32).reshape(512, 2)
    img = ol.gaussian_blur7(synth.frame(80, 80, 31))
    cx, cy = 40, 39
    for ang in (0.0, 37.5, 153.59795, 359.99):
        ang = np.float32(ang)
        rad = np.float32(ang * np.float32(np.pi / np.float32(180.0)))
        a, b = np.float32(np.cos(np.float64(rad))), np.float32(np.sin(np.float64(rad)))
        bits = []
        for k in range(256):
            vals = []
            for px, py in (pat[2 * k], pat[2 * k + 1]):
                yy = int(np.rint(np.float32(np.float32(px) * b) + np.float32(np.float32(py) * a)))
                xx = int(np.rint(np.float32(np.float32(px) * a) - np.float32(np.float32(py) * b)))
                vals.append(int(img[cy + yy, cx + xx]))
            bits.append(vals[0] < vals[1])
        ref = np.packbits(np.array(bits, np.uint8), bitorder="little")
        got = ol.orb_descriptor(img, cx, cy, ang)
        # float cos/sin via float64 may differ from cosf/sinf by 1 ulp in rare cases: allow <=2 bit flips
        assert np.unpackbits(ref ^ got).sum() <= 2


# ----------------------------------------------------------------------------- octree
def test_octree_invariants():
    rng = np.random.default_rng(5)
    for trial in range(20):
        w, h = 608, 448
        n = int(rng.integers(1, 3000))
        pts = rng.choice(w * h, size=n, replace=False)
        x, y = (pts % w).astype(np.int32), (pts // w).astype(np.int32)
        r = rng.integers(7, 60, n).astype(np.int32)
        N = int(rng.integers(1, 300))
        sel = ol.distribute_octree(x, y, r, 16, 16 + w, 16, 16 + h, N)
        assert len(set(sel.tolist())) == len(sel)
        assert len(sel) <= max(N + 3, 1) or len(sel) <= n
        if n >= 4 * N:
            assert len(sel) >= N
        if n <= N:  # every point ends up alone in a node
            assert sorted(sel.tolist()) == list(range(n))
    assert len(ol.distribute_octree([], [], [], 16, 624, 16, 464, 100)) == 0
    assert ol.distribute_octree([5], [7], [30], 16, 624, 16, 464, 100).tolist() == [0]


def test_octree_response_tie_prefers_first():
    # A,B adjacent (stay in one node), C far away; N=2 -> one pass gives nodes {A,B},{C} and stops.
    # Equal responses inside {A,B}: the first in input order wins (strict '>' at ORBextractor.cc:774).
    # List order: children are push_front'ed, so the later-created node {C} (n4) comes first.
    xs, ys = [10, 11, 500], [10, 10, 400]
    assert ol.distribute_octree(xs, ys, [30, 30, 9], 16, 624, 16, 464, 2).tolist() == [2, 0]
    assert ol.distribute_octree(xs, ys, [30, 31, 9], 16, 624, 16, 464, 2).tolist() == [2, 1]
    assert ol.distribute_octree(xs[::-1], ys[::-1], [9, 30, 30], 16, 624, 16, 464, 2).tolist() == [0, 1]


# ----------------------------------------------------------------------------- operator() behaviour
def test_extract_empty_constant_and_slots():
    e = ol.OracleExtractor(500, 1.2, 4, 20, 7)
    assert e(None)[0] == -1
    mono, kps, desc = e(synth.constant_frame(320, 240))
    assert mono == 0 and len(kps) == 0 and desc.shape == (0, 32)
    img = synth.frame(320, 240, 2)
    mono0, k0, d0 = e(img, (0, 0))
    assert mono0 == len(k0) > 400
    assert np.all(np.diff(k0["octave"]) >= 0)  # level-major natural order
    mono1, k1, d1 = e(img, (0, 1000))  # monocular call site (Frame.cc:445): everything from the back
    assert mono1 == 0
    assert np.array_equal(k1[::-1], k0) and np.array_equal(d1[::-1], d0)
    mono2, k2, d2 = e(img, (100, 200))
    lap = (k0["x"] >= 100) & (k0["x"] <= 200)
    assert mono2 == int((~lap).sum())
    assert np.array_equal(k2[:mono2], k0[~lap]) and np.array_equal(k2[mono2:][::-1], k0[lap])
    assert np.array_equal(d2[:mono2], d0[~lap]) and np.array_equal(d2[mono2:][::-1], d0[lap])


def test_extract_keypoint_fields():
    e = ol.OracleExtractor(1000, 1.2, 8, 20, 7)
    mono, kps, desc = e(synth.frame(640, 480, 0))
    t = e.tables()
    assert 990 <= len(kps) <= 1024 and mono == len(kps)
    assert np.all(kps["class_id"] == -1)
    for l in range(8):
        m = kps["octave"] == l
        lk = e.level_keypoints(l)
        assert m.sum() == len(lk) <= t["features_per_level"][l] + 3
        assert np.all(kps["size"][m] == np.float32(int(np.float32(31) * t["scale"][l])))
        w, h = e.level_size(l)
        assert np.all((lk["x"] >= 19) & (lk["x"] < w - 19) & (lk["y"] >= 19) & (lk["y"] < h - 19))
        if l:
            assert np.array_equal(kps["x"][m], lk["x"] * t["scale"][l])
    assert np.all((kps["angle"] >= 0) & (kps["angle"] < 360))
    assert np.all(kps["response"] >= 7)


# ----------------------------------------------------------------------------- matcher
def test_descriptor_distance_is_popcount():
    rng = np.random.default_rng(3)
    for _ in range(200):
        a, b = rng.integers(0, 256, (2, 32), dtype=np.uint8)
        assert ol.descriptor_distance(a, b) == int(np.unpackbits(a ^ b).sum())
    z = np.zeros(32, np.uint8)
    assert ol.descriptor_distance(z, z) == 0 and ol.descriptor_distance(z, ~z) == 256


def test_three_maxima():
    s = [0] * 30
    s[3], s[7], s[9] = 10, 50, 4
    assert ol.three_maxima(s) == (7, 3, -1)  # third < 0.1 * max1 dropped
    s[9] = 5
    assert ol.three_maxima(s) == (7, 3, 9)
    assert ol.three_maxima([0] * 30) == (-1, -1, -1)
    s = [0] * 30
    s[2] = 100
    s[4] = 9
    assert ol.three_maxima(s) == (2, -1, -1)


def test_block_best2_ties_resolve_to_first():
    a = np.zeros((1, 32), np.uint8)
    b = np.zeros((4, 32), np.uint8)
    b[0, 0] = 0b111
    b[1, 0] = 0b1
    b[2, 1] = 0b1  # same distance as b[1] -> the earlier one stays best
    b[3, 0] = 0b11
    best, second, arg = ol.block_best2(a, b)
    assert (best[0], second[0], arg[0]) == (1, 1, 1)


def test_grid_query_matches_bruteforce_filter():
    e = ol.OracleExtractor(1000, 1.2, 8, 20, 7)
    _, kps, _ = e(synth.frame(640, 480, 4))
    g = ol.OracleGrid(kps, 0.0, 0.0, 640.0, 480.0)
    rng = np.random.default_rng(0)
    for _ in range(50):
        x, y = rng.uniform(0, 640), rng.uniform(0, 480)
        r = rng.uniform(5, 40)
        lo, hi = int(rng.integers(-1, 4)), int(rng.integers(-1, 8))
        got = g.query(x, y, r, lo, hi)
        x32, y32, r32 = np.float32(x), np.float32(y), np.float32(r)
        m = (np.abs(kps["x"] - x32) < r32) & (np.abs(kps["y"] - y32) < r32)
        if lo > 0 or hi >= 0:
            m &= kps["octave"] >= lo
            if hi >= 0:
                m &= kps["octave"] <= hi
        assert sorted(got.tolist()) == np.nonzero(m)[0].tolist()


def test_search_for_triangulation_last_minimum_and_predicate():
    """ORBmatcher.cc:1015 `dist > bestDist -> continue` lets a LATER equal distance win; the geometric predicate is
    consulted only through the pair bits; dist > TH_LOW never matches."""
    d1 = np.zeros((1, 32), np.uint8)
    d2 = np.zeros((4, 32), np.uint8)
    d2[1, 0] = 1          # distances 0, 1, 0, 64
    d2[3, :8] = 0xFF
    fv1 = (np.array([5], np.int32), np.array([0, 1], np.int32), np.array([0], np.int32))
    fv2 = (np.array([5], np.int32), np.array([0, 4], np.int32), np.array([0, 1, 2, 3], np.int32))
    a1, a2 = np.zeros(1, np.float32), np.zeros(4, np.float32)
    off = np.array([0, 4], np.int32)
    run = lambda bits, e2=(1, 1, 1, 1): ol.search_for_triangulation(
        d1, a1, [1], fv1, d2, a2, list(e2), fv2, None if bits is None else np.array([bits], np.uint32), off, False)
    assert run(None)[1].tolist() == [2]           # last of the two zero distances
    assert run(0b1011)[1].tolist() == [0]         # pair (0,2) fails the predicate
    assert run(0b1010)[1].tolist() == [1]         # only the dist-1 candidate passes
    assert run(0b1000)[1].tolist() == [-1]        # the only passing pair is beyond TH_LOW
    assert run(None, (1, 1, 0, 1))[1].tolist() == [0]  # KF2 feature 2 already has a MapPoint
    assert ol.search_for_triangulation(d1, a1, [0], fv1, d2, a2, [1, 1, 1, 1], fv2, None, None, False)[0] == 0


# ----------------------------------------------------------------------------- whole-batch checker (bench.py's gate)
def test_batch_checker_equals_the_per_frame_calls():
    """or_extract_batch_mt / or_block_best2_batch_mt (every frame of a batch on all host threads) against the plain
    per-frame or_extract / or_block_best2 they wrap, with and without a lapping area, incl. a constant frame (n = 0)."""
    W, H, nf = 320, 240, 300
    frames = [synth.sequence_frame(W, H, 77, t) for t in range(6)] + [synth.constant_frame(W, H)]
    frames += [synth.content_frame("value_noise", W, H, 78, t) for t in range(3)]
    frames = np.stack(frames)
    cap = nf + 3 * 8 + 64
    for lap in ((0, 0), (100, 200)):
        for threads in (1, 3, 16):
            counts, kps, desc = ol.extract_batch(frames, nf, cap, lapping=lap, nthreads=threads)
            ref = ol.OracleExtractor(nf, 1.2, 8, 20, 7)
            for f in range(len(frames)):
                mono, rk, rd = ref(frames[f], lap)
                assert counts[f].tolist() == [len(rk), mono]
                assert kps[f, :len(rk)].tobytes() == rk.tobytes() and np.array_equal(desc[f, :len(rk)], rd)
                assert not desc[f, len(rk):].any()
    assert counts[6].tolist() == [0, 0] or counts[6, 0] == 0
    assert ol.compare_batch(counts, kps, desc, counts, kps, desc) == []
    d2 = desc.copy()
    d2[4, 3, 7] ^= 1
    k2 = kps.copy()
    k2[8]["angle"][0] = np.nextafter(k2[8]["angle"][0], np.float32(400))
    assert ol.compare_batch(counts, kps, d2, counts, kps, desc) == [4]
    assert ol.compare_batch(counts, k2, desc, counts, kps, desc) == [8]
    # match rows: frame f against frame f - 1 (row 0 against the last frame)
    prev = np.roll(desc, 1, axis=0)
    nprev = np.roll(counts[:, 0], 1)
    best, second, arg = ol.block_best2_batch(desc, counts[:, 0], prev, nprev, nthreads=5)
    for f in range(len(frames)):
        n = int(counts[f, 0])
        rb, rs, ra = ol.block_best2(desc[f, :n], prev[f, :int(nprev[f])])
        assert np.array_equal(best[f, :n], rb) and np.array_equal(second[f, :n], rs) and np.array_equal(arg[f, :n], ra)
