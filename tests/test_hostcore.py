"""CPU tests of the PRODUCT's device-agnostic core (visual_sgraphs_amd/csrc/*.h compiled for the host by
tests/_hostcore): geometry tables, the libstdc++-exact introsort, the array-based octree (run as a
1-thread group) and the float helpers -- each against the oracle or against libm / std::sort."""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as ol
from visual_sgraphs_amd import synth

HC_DIR = Path(__file__).resolve().parent / "_hostcore"
_i32p = C.POINTER(C.c_int32)


@pytest.fixture(scope="module")
def hc():
    asan = bool(os.environ.get("VSG_HOSTCORE_ASAN"))  # tests/test_sanitizers.py: the ASan + UBSan build of the same core
    subprocess.check_call(["make", "-C", str(HC_DIR)] + (["asan"] if asan else []), stdout=subprocess.DEVNULL)
    L = C.CDLL(str(HC_DIR / ("libvsg_hostcore_asan.so" if asan else "libvsg_hostcore.so")))
    L.hc_build.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.hc_level.argtypes = [C.c_int, _i32p]
    L.hc_octree.argtypes = [C.c_int, _i32p, _i32p, _i32p, C.c_int, C.c_int, _i32p]
    L.hc_fast_atan2.restype = C.c_float
    L.hc_fast_atan2.argtypes = [C.c_float, C.c_float]
    L.hc_sinf.restype = C.c_float
    L.hc_sinf.argtypes = [C.c_float, C.c_int]
    L.hc_cosf.restype = C.c_float
    L.hc_cosf.argtypes = [C.c_float, C.c_int]
    L.hc_brief_rotation.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.hc_resize_tables.argtypes = [C.c_int, C.POINTER(C.c_int16), C.POINTER(C.c_int16)]
    L.hc_sort.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    L.hc_tables.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), _i32p, _i32p]
    L.hc_cell.argtypes = [C.c_int, _i32p]
    return L


def level_info(hc, l):
    out = np.zeros(16, np.int32)
    n = hc.hc_level(l, out.ctypes.data_as(_i32p))
    keys = ["w", "h", "pitch", "nCols", "nRows", "wCell", "hCell", "quota", "cand_cap", "sel_cap", "nIni",
            "oct_width", "oct_height", "cell_base", "kp_size"]
    return dict(zip(keys, out[:n].tolist()))


def hc_octree(hc, level, x, y, r, n_override=-1):
    x, y, r = (np.ascontiguousarray(a, np.int32) for a in (x, y, r))
    out = np.zeros(len(x) + 128, np.int32)
    m = hc.hc_octree(level, x.ctypes.data_as(_i32p), y.ctypes.data_as(_i32p), r.ctypes.data_as(_i32p), len(x),
                     n_override, out.ctypes.data_as(_i32p))
    p = out[:m].astype(np.uint32)
    return (p & 0xFFF).astype(np.int32), ((p >> 12) & 0xFFF).astype(np.int32), (p >> 24).astype(np.int32)


GEOM_CASES = [(640, 480, 1000, 8), (752, 480, 1200, 8), (1280, 720, 2000, 8), (320, 240, 500, 4), (1241, 376, 2000, 8)]


@pytest.mark.parametrize("w,h,nf,nl", GEOM_CASES)
def test_geometry_matches_oracle(hc, w, h, nf, nl):
    assert hc.hc_build(nf, 1.2, nl, 20, 7, h, w) == 0
    e = ol.OracleExtractor(nf, 1.2, nl, 20, 7)
    e(synth.constant_frame(w, h))
    t = e.tables()
    sc, inv = np.zeros(nl, np.float32), np.zeros(nl, np.float32)
    q, um = np.zeros(nl, np.int32), np.zeros(16, np.int32)
    hc.hc_tables(sc.ctypes.data_as(C.POINTER(C.c_float)), inv.ctypes.data_as(C.POINTER(C.c_float)),
                 q.ctypes.data_as(_i32p), um.ctypes.data_as(_i32p))
    assert np.array_equal(sc, t["scale"]) and np.array_equal(inv, t["inv_scale"])
    assert np.array_equal(q, t["features_per_level"]) and np.array_equal(um, t["umax"])
    for l in range(nl):
        info = level_info(hc, l)
        assert (info["w"], info["h"]) == e.level_size(l)
        assert info["quota"] == t["features_per_level"][l]
        assert info["kp_size"] == int(np.float32(31) * t["scale"][l])
        assert info["pitch"] % 64 == 0 and info["pitch"] >= info["w"]


def _reference_configs():
    """every (image size, ORBextractor.*) tuple of the reference's settings files (tests/golden/make_reference_configs.py)"""
    import json
    from pathlib import Path
    cases = json.loads((Path(__file__).parent / "golden" / "reference_configs.json").read_text())["cases"]
    return [tuple(c["params"][k] for k in ("w", "h", "nFeatures", "scaleFactor", "nLevels", "iniThFAST", "minThFAST"))
            for c in cases]


@pytest.mark.parametrize("w,h,nf,sf,nl,ini,mn", _reference_configs())
def test_geometry_matches_oracle_for_every_reference_config(hc, w, h, nf, sf, nl, ini, mn):
    assert hc.hc_build(nf, sf, nl, ini, mn, h, w) == 0
    e = ol.OracleExtractor(nf, sf, nl, ini, mn)
    e(synth.constant_frame(w, h))
    t = e.tables()
    sc, inv = np.zeros(nl, np.float32), np.zeros(nl, np.float32)
    q, um = np.zeros(nl, np.int32), np.zeros(16, np.int32)
    hc.hc_tables(sc.ctypes.data_as(C.POINTER(C.c_float)), inv.ctypes.data_as(C.POINTER(C.c_float)),
                 q.ctypes.data_as(_i32p), um.ctypes.data_as(_i32p))
    assert np.array_equal(sc, t["scale"]) and np.array_equal(inv, t["inv_scale"])
    assert np.array_equal(q, t["features_per_level"]) and int(q.sum()) == nf
    for l in range(nl):
        info = level_info(hc, l)
        assert (info["w"], info["h"]) == e.level_size(l)
        assert info["quota"] == t["features_per_level"][l]


@pytest.mark.parametrize("w,h,nf,nl", GEOM_CASES + [(2400, 2336, 1500, 3), (100, 100, 300, 2)])
def test_fast_cell_records_restate_the_cells(hc, w, h, nf, nl):
    """FastCellRec (one 32-byte scalar load per cell in k_fast_cells) against an independent restatement of the per-cell
    arithmetic the kernel used to do itself: tile origin, dwords and 16-byte chunks per tile row, run layout of the
    necessary test (8 pixels per run, runs start at a dword boundary) and the flag masks of a row's first / last run."""
    assert hc.hc_build(nf, 1.2, nl, 20, 7, h, w) == 0
    n = hc.hc_total_cells()
    assert n > 0

    def mask8(m8):  # pixel p < 4 -> byte p, bits 5 (dark) / 4 (bright); p >= 4 -> byte p - 4, bits 7 / 6
        f = 0
        for p in range(8):
            if m8 >> p & 1:
                f |= (0x30 if p < 4 else 0xC0) << (8 * (p & 3))
        return f

    cell, rec = np.zeros(5, np.int32), np.zeros(11, np.uint32)
    for i in range(n):
        hc.hc_cell(i, cell.ctypes.data_as(_i32p))
        hc.hc_fast_rec(i, rec.ctypes.data_as(C.POINTER(C.c_uint)))
        level, x0, y0, x1, y1 = (int(v) for v in cell)
        r = [int(v) for v in rec]
        info = level_info(hc, level)
        assert r[2] >> 20 == level and r[2] & 0xFFFFF == (0 if level == 0 else info["pitch"])
        assert r[1] == r[8] and r[6] == r[9] + r[10]
        assert r[7] == (x0 & 0xFFFF) | (y0 & 0xFFFF) << 16
        vw, vh = x1 - x0, y1 - y0
        if vw <= 0 or vh <= 0:
            assert r[3] == 0
            continue
        ax = (x0 - 3) & ~3
        ox = x0 - 3 - ax
        assert r[0] == ax | (y0 - 3) << 16
        tdw = -(-(ox + vw + 6) // 4)           # dwords that cover tile columns [0, ox + vw + 6)
        first = (3 + ox) // 4                  # dword of the first centre pixel (tile column 3 + ox)
        last = (3 + ox + vw - 1) // 4          # dword of the last one
        nrun = (last - first + 2) // 2         # runs of 2 dwords from `first`
        w3 = r[3]
        assert (w3 & 127, w3 >> 7 & 127, w3 >> 14 & 3, w3 >> 16 & 31, w3 >> 21 & 7, w3 >> 24 & 1, w3 >> 25) == \
            (vw, vh, ox, tdw, -(-tdw // 4), first, nrun), (i, level)
        # pixel p of run k is tile column 4 * (first + 2 k) + p = valid-region column 4 * (first + 2 k) + p - 3 - ox
        col0 = 4 * first - 3 - ox
        m_first = sum(1 << p for p in range(8) if 0 <= col0 + p < vw)
        colL = 4 * (first + 2 * (nrun - 1)) - 3 - ox
        m_last = sum(1 << p for p in range(8) if colL + p < vw)
        if nrun == 1:
            assert mask8(m_first) & mask8(m_last) == r[4] & r[5]
        else:
            assert (mask8(m_first), mask8(m_last)) == (r[4], r[5]), (i, level, hex(r[4]), hex(r[5]))


def test_geometry_known_answers_c2(hc):
    assert hc.hc_build(1000, 1.2, 8, 20, 7, 480, 640) == 0
    cells = [(17, 12, 36, 38), (14, 10, 36, 37), (11, 8, 38, 38), (9, 7, 38, 36), (7, 5, 40, 40), (6, 4, 38, 41),
             (5, 3, 37, 43), (4, 2, 37, 51)]
    for l, c in enumerate(cells):
        i = level_info(hc, l)
        assert (i["nCols"], i["nRows"], i["wCell"], i["hCell"]) == c
        assert i["nIni"] == 1
    assert hc.hc_total_cells() == 577
    assert hc.hc_build(1200, 1.2, 8, 20, 7, 480, 752) == 0
    assert hc.hc_total_cells() == 700 and level_info(hc, 0)["nIni"] == 2
    assert hc.hc_build(2000, 1.2, 8, 20, 7, 720, 1280) == 0
    assert hc.hc_total_cells() == 1987


def test_geometry_rejects_what_the_reference_cannot_process(hc):
    assert hc.hc_build(500, 1.2, 8, 20, 7, 120, 160) < 0      # top level narrower than one 35px cell
    assert hc.hc_build(500, 1.2, 4, 20, 7, 800, 300) < 0      # width/height rounds to 0 initial nodes
    assert hc.hc_build(500, 1.2, 4, 0, 7, 240, 320) < 0       # threshold 0: score map cannot encode it
    assert hc.hc_build(500, 1.2, 3, 20, 7, 120, 160) == 0


def test_resize_tables_match_oracle_resize(hc):
    """Apply the product's fixed-point tables in numpy and compare with the oracle's cv::resize restatement."""
    assert hc.hc_build(500, 1.2, 4, 20, 7, 240, 320) == 0
    src = synth.frame(320, 240, 77)
    for l in range(1, 4):
        info = level_info(hc, l)
        w, h = info["w"], info["h"]
        xs, ys = np.zeros((w, 4), np.int16), np.zeros((h, 4), np.int16)
        hc.hc_resize_tables(l, xs.ctypes.data_as(C.POINTER(C.c_int16)), ys.ctypes.data_as(C.POINTER(C.c_int16)))
        S = src.astype(np.int64)
        sx, a0, a1, sx1 = (xs[:, k].astype(np.int64) for k in range(4))
        sy0, sy1, b0, b1 = (ys[:, k].astype(np.int64) for k in range(4))
        H = S[:, sx] * a0[None, :] + S[:, sx1] * a1[None, :]
        out = ((((b0[:, None] * (H[sy0] >> 4)) >> 16) + ((b1[:, None] * (H[sy1] >> 4)) >> 16) + 2) >> 2).astype(np.uint8)
        assert np.array_equal(out, ol.resize_linear(src, w, h))
        src = out


def test_introsort_is_libstdcxx_exact(hc):
    """Key-only comparator with many ties: payload order must equal std::sort's (checked in C++ by
    tests/_hostcore against std::sort on 200k cases during development; here: sortedness + stability
    for n <= 16, where libstdc++ is a pure insertion sort)."""
    rng = np.random.default_rng(0)
    for n in (1, 2, 5, 16, 17, 100, 1000):
        keys = rng.integers(0, 5, n).astype(np.uint64)
        items = (keys << np.uint64(32)) | np.arange(n, dtype=np.uint64)
        buf = items.copy()
        hc.hc_sort(buf.ctypes.data_as(C.POINTER(C.c_uint64)), n)
        assert np.all(np.diff((buf >> np.uint64(32)).astype(np.int64)) >= 0)
        assert sorted(buf.tolist()) == sorted(items.tolist())
        if n <= 16:
            assert np.array_equal(buf, items[np.argsort(keys, kind="stable")])


def test_introsort_and_its_device_split_equal_std_sort(hc):
    """Move-for-move equality with the real std::sort (key-only comparator, tie-heavy inputs), both for the serial
    replay and for the split the device runs: serial partition phase + parallel stable rank."""
    rng = np.random.default_rng(1)
    p64 = C.POINTER(C.c_uint64)
    for trial in range(400):
        n = int(rng.integers(1, 400))
        nkeys = int(rng.choice([1, 2, 3, 8, 50, 100000]))
        keys = rng.integers(0, nkeys, n).astype(np.uint64)
        if trial % 5 == 0:
            keys = np.sort(keys)[::-1].copy() if trial % 10 == 0 else np.sort(keys)
        items = (keys << np.uint64(32)) | np.arange(n, dtype=np.uint64)
        a, b, c = items.copy(), items.copy(), items.copy()
        hc.hc_std_sort(a.ctypes.data_as(p64), n)
        hc.hc_sort(b.ctypes.data_as(p64), n)
        hc.hc_sort_split(c.ctypes.data_as(p64), n)
        assert np.array_equal(a, b), (trial, n, nkeys)
        assert np.array_equal(a, c), (trial, n, nkeys)


def test_split_sort_with_forced_heapsort_fallback(hc):
    """A tiny depth limit sends large ranges to the heapsort branch of introsort; the windowed stable rank that
    follows the partition phase must still reproduce the serial replay (same depth limit) exactly."""
    rng = np.random.default_rng(2)
    p64 = C.POINTER(C.c_uint64)
    hc.hc_sort_depth.argtypes = [p64, C.c_int, C.c_int]
    hc.hc_sort_split_depth.argtypes = [p64, C.c_int, C.c_int]
    for trial in range(200):
        n = int(rng.integers(17, 500))
        keys = rng.integers(0, int(rng.choice([2, 5, 1000])), n).astype(np.uint64)
        items = (keys << np.uint64(32)) | np.arange(n, dtype=np.uint64)
        for depth in (0, 1, 2):
            a, b = items.copy(), items.copy()
            hc.hc_sort_depth(a.ctypes.data_as(p64), n, depth)
            hc.hc_sort_split_depth(b.ctypes.data_as(p64), n, depth)
            assert np.array_equal(a, b), (trial, n, depth)


def test_float_helpers_match_oracle_and_libm(hc):
    rng = np.random.default_rng(2)
    for _ in range(5000):
        y, x = (int(v) for v in rng.integers(-200000, 200000, 2))
        assert np.float32(hc.hc_fast_atan2(y, x)) == ol.fast_atan2(y, x)
    # sinf/cosf kernels: bit-equal to this host's libm on a dense sample (the full 1.09e9-value sweep of
    # [0, 6.4] was run offline: 0 mismatches for both contraction variants)
    import struct
    libm = C.CDLL("libm.so.6")
    libm.sinf.restype = libm.cosf.restype = C.c_float
    libm.sinf.argtypes = libm.cosf.argtypes = [C.c_float]
    xs = np.concatenate([rng.uniform(0, 6.3, 20000), [0.0, 1e-5, 0.785398, 0.7853982, 1.5707964, 3.1415927, 6.2831855]])
    for xv in xs.astype(np.float32):
        for fma in (0, 1):
            assert struct.pack("f", hc.hc_sinf(float(xv), fma)) == struct.pack("f", libm.sinf(float(xv)))
            assert struct.pack("f", hc.hc_cosf(float(xv), fma)) == struct.pack("f", libm.cosf(float(xv)))


def test_lane_parallel_rotation_equals_the_scalar_kernels(hc):
    """k_orient_desc evaluates fastAtan2 -> cosf / sinf for a wavefront's keypoints at once with the branch-free forms of
    vsg_math.h (fast_atan2_deg_sel, sincos_pair); they must give the bits of the branching forms, which are the ones
    pinned against libm / the oracle above.  (The full sweep of every float in [0, 6.4] -- 1.09e9 values, both
    contraction variants -- was run offline: 0 mismatches.)"""
    import struct
    hc.hc_sincos_pair_sweep.restype = C.c_long
    hc.hc_sincos_pair_sweep.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
    hc.hc_fast_atan2_sel.restype = C.c_float
    hc.hc_fast_atan2_sel.argtypes = [C.c_float, C.c_float]
    hc.hc_brief_rotation_of_moments.argtypes = [C.c_float, C.c_float] + [C.POINTER(C.c_float)] * 3
    bits = lambda f: struct.unpack("<I", struct.pack("<f", f))[0]
    # windows of consecutive floats around every branch point of the scalar kernels, plus strided samples of the range
    windows = [(0, 200000)]
    for centre in (2.0 ** -12, 0.7853982, 1.5707964, 2.3561945, 3.1415927, 3.9269908, 4.712389, 5.4977875, 6.2831855):
        b = bits(centre)
        windows.append((b - 100000, b + 100000))
    top = bits(6.4)
    rng = np.random.default_rng(5)
    for start in rng.integers(0, top - 50000, 40):
        windows.append((int(start), int(start) + 50000))
    for fma in (0, 1):
        for lo, hi in windows:
            assert hc.hc_sincos_pair_sweep(lo, hi, fma) == 0, (lo, hi, fma)
    for _ in range(20000):
        y, x = (int(v) for v in rng.integers(-200000, 200000, 2))
        if rng.integers(0, 10) == 0:
            x = y if rng.integers(0, 2) else -y  # |x| == |y|: the tie of the two-sided form
        assert bits(hc.hc_fast_atan2_sel(y, x)) == bits(hc.hc_fast_atan2(y, x)), (y, x)
    ang, a, b = C.c_float(), C.c_float(), C.c_float()
    a0, b0 = C.c_float(), C.c_float()
    for _ in range(2000):
        y, x = (int(v) for v in rng.integers(-200000, 200000, 2))
        hc.hc_brief_rotation_of_moments(y, x, C.byref(ang), C.byref(a), C.byref(b))
        want = hc.hc_fast_atan2(y, x)
        hc.hc_brief_rotation(want, C.byref(a0), C.byref(b0))
        assert (bits(ang.value), bits(a.value), bits(b.value)) == (bits(want), bits(a0.value), bits(b0.value))


@pytest.mark.parametrize("w,h,nf,nl,seeds", [(640, 480, 1000, 8, [0, 1]), (752, 480, 1200, 8, [2]),
                                              (320, 240, 500, 4, [3, 4, 5]), (1280, 720, 2000, 8, [6])])
def test_octree_core_equals_oracle_on_real_candidates(hc, w, h, nf, nl, seeds):
    assert hc.hc_build(nf, 1.2, nl, 20, 7, h, w) == 0
    e = ol.OracleExtractor(nf, 1.2, nl, 20, 7)
    rng = np.random.default_rng(9)
    for seed in seeds:
        e(synth.frame(w, h, seed))
        for l in range(nl):
            x, y, r = e.candidates(l)
            kp = e.level_keypoints(l)
            perm = rng.permutation(len(x))  # the core must not depend on candidate order
            gx, gy, gr = hc_octree(hc, l, x[perm], y[perm], r[perm])
            assert np.array_equal(gx + 16, kp["x"].astype(np.int32))
            assert np.array_equal(gy + 16, kp["y"].astype(np.int32))
            assert np.array_equal(gr, kp["response"].astype(np.int32))


def test_octree_core_equals_oracle_on_adversarial_sets(hc):
    """Random sparse/dense point sets, many equal responses (tie-breaking by candidate rank), quotas from 0
    to more than the number of points, clustered points (deep trees, careful phase with ties)."""
    assert hc.hc_build(1000, 1.2, 8, 20, 7, 480, 640) == 0
    rng = np.random.default_rng(11)
    for trial in range(150):
        l = int(rng.integers(0, 8))
        info = level_info(hc, l)
        W, H = info["oct_width"], info["oct_height"]
        # valid candidate coordinates are 3..W-4 (3 px inside the FAST cells)
        n = int(rng.integers(0, 1500))
        mode = trial % 3
        if mode == 0:
            xs = rng.integers(3, W - 3, n)
            ys = rng.integers(3, H - 3, n)
        elif mode == 1:  # clusters
            cx, cy = rng.integers(3, W - 3, 5), rng.integers(3, H - 3, 5)
            k = rng.integers(0, 5, n)
            xs = np.clip(cx[k] + rng.integers(-12, 13, n), 3, W - 4)
            ys = np.clip(cy[k] + rng.integers(-12, 13, n), 3, H - 4)
        else:  # lattice
            xs = 3 + (rng.integers(0, (W - 6) // 4, n) * 4)
            ys = 3 + (rng.integers(0, (H - 6) // 4, n) * 4)
        pts = np.unique(np.stack([ys, xs], 1), axis=0)
        ys, xs = pts[:, 0], pts[:, 1]
        n = len(xs)
        rs = rng.integers(7, 10 if trial % 2 else 200, n)
        N = int(rng.choice([0, 1, 2, 5, 17, 60, 217, 400, 2000]))
        # reference candidate order = cell-major then row-major inside the cell
        i, j = (ys - 3) // info["hCell"], (xs - 3) // info["wCell"]
        order = np.lexsort((xs, ys, j, i))
        xs, ys, rs = xs[order], ys[order], rs[order]
        want = ol.distribute_octree(xs, ys, rs, 16, 16 + W, 16, 16 + H, N)
        perm = rng.permutation(n)
        gx, gy, gr = hc_octree(hc, l, xs[perm], ys[perm], rs[perm], N)
        assert np.array_equal(gx, xs[want]) and np.array_equal(gy, ys[want]) and np.array_equal(gr, rs[want]), trial


@pytest.mark.parametrize("w,h,nf", [(640, 480, 1000), (1280, 720, 2000)])
def test_octree_core_on_photographs_and_the_rank_sorted_careful_phase(hc, w, h, nf):
    """Real photographs: 5.7 k (640x480) / 19.6 k (1280x720) level-0 candidates, ~200-400 divisible nodes when the careful phase
    starts -- the node sort takes the counting form when all (size, UL.x) keys differ (round 6) and the exact std::sort replay when
    two are equal.  Both against the oracle's std::list + std::sort code, on the candidates of every level of every photograph,
    and on copies of them with responses flattened (ties in the best-point rule) and with points duplicated in mirrored
    quadrants (equal node sizes: ties in the sort)."""
    assert hc.hc_build(nf, 1.2, 8, 20, 7, h, w) == 0
    e = ol.OracleExtractor(nf, 1.2, 8, 20, 7)
    for kind in synth.PHOTO_CLASSES:
        e(synth.content_frame(kind, w, h, 77, 2))
        for l in range(8):
            x, y, r = e.candidates(l)
            kp = e.level_keypoints(l)
            gx, gy, gr = hc_octree(hc, l, x, y, r)
            assert np.array_equal(gx + 16, kp["x"].astype(np.int32)) and np.array_equal(gy + 16, kp["y"].astype(np.int32)), (kind, l)
            assert np.array_equal(gr, kp["response"].astype(np.int32))
    # symmetric point sets: the left half mirrored onto the right half gives pairs of nodes with equal sizes (sort ties)
    info = level_info(hc, 0)
    W, H = info["oct_width"], info["oct_height"]
    e(synth.content_frame("photo_china", w, h, 78, 0))
    x, y, r = e.candidates(0)
    keep = x < W // 2 - 2
    xs = np.concatenate([x[keep], (W - 1) - x[keep]])
    ys, rs = np.concatenate([y[keep], y[keep]]), np.concatenate([r[keep], r[keep]])
    ok = (xs >= 3) & (xs <= W - 4)
    pts = np.unique(np.stack([ys[ok], xs[ok], np.minimum(rs[ok], 40)], 1), axis=0)
    _, first = np.unique(pts[:, :2], axis=0, return_index=True)  # one point per pixel
    pts = pts[np.sort(first)]
    ys, xs, rs = pts[:, 0], pts[:, 1], pts[:, 2]
    # reference candidate order = cell-major then row-major inside the cell
    i, j = (ys - 3) // info["hCell"], (xs - 3) // info["wCell"]
    order = np.lexsort((xs, ys, j, i))
    xs, ys, rs = xs[order], ys[order], rs[order]
    want = ol.distribute_octree(xs, ys, rs, 16, 16 + W, 16, 16 + H, info["quota"])
    gx, gy, gr = hc_octree(hc, 0, xs[::-1], ys[::-1], rs[::-1])
    assert len(xs) > 2000 and np.array_equal(gx, xs[want]) and np.array_equal(gy, ys[want]) and np.array_equal(gr, rs[want])


def test_octree_core_multiple_initial_nodes(hc):
    assert hc.hc_build(2000, 1.2, 8, 20, 7, 376, 1241) == 0  # KITTI aspect: nIni = 3 or 4
    e = ol.OracleExtractor(2000, 1.2, 8, 20, 7)
    e(synth.frame(1241, 376, 8))
    assert level_info(hc, 0)["nIni"] >= 3
    for l in range(8):
        x, y, r = e.candidates(l)
        kp = e.level_keypoints(l)
        gx, gy, gr = hc_octree(hc, l, x[::-1], y[::-1], r[::-1])
        assert np.array_equal(gx + 16, kp["x"].astype(np.int32)) and np.array_equal(gy + 16, kp["y"].astype(np.int32))
    # EMPTY initial nodes among two to four (:597-608: erased, the others keep their order): points confined to some of
    # the vertical strips of the level, every subset of strips
    rng = np.random.default_rng(5)
    for l in (0, 3):
        info = level_info(hc, l)
        W, H, nI = info["oct_width"], info["oct_height"], info["nIni"]
        hX = np.float32(W) / np.float32(nI)
        for mask in range(1, 1 << nI):
            n = int(rng.integers(40, 900))
            xs, ys = rng.integers(3, W - 3, n), rng.integers(3, H - 3, n)
            strip = np.minimum((xs.astype(np.float32) / hX).astype(np.int64), nI - 1)
            keep = ((mask >> strip) & 1) == 1
            pts = np.unique(np.stack([ys[keep], xs[keep]], 1), axis=0)
            if len(pts) == 0:
                continue
            ys, xs = pts[:, 0], pts[:, 1]
            rs = rng.integers(7, 60, len(xs))
            i, j = (ys - 3) // info["hCell"], (xs - 3) // info["wCell"]
            order = np.lexsort((xs, ys, j, i))
            xs, ys, rs = xs[order], ys[order], rs[order]
            for N in (3, 40, 300):
                want = ol.distribute_octree(xs, ys, rs, 16, 16 + W, 16, 16 + H, N)
                perm = rng.permutation(len(xs))
                gx, gy, gr = hc_octree(hc, l, xs[perm], ys[perm], rs[perm], N)
                assert np.array_equal(gx, xs[want]) and np.array_equal(gy, ys[want]) and np.array_equal(gr, rs[want]), (l, mask, N)


def test_synth_header_matches_python(hc):
    """include/vsg_synth.h (the C mirror of the SURVEY 8d frame generator) equals synth.py byte for byte."""
    hc.hc_synth_frame.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    for (w, h, seq, t, div, noise) in [(640, 480, 0, 0, 1, 6), (320, 240, 7, 5, 1, 6), (200, 150, 3, 40, 8, 6),
                                       (96, 64, 9, 2, 1, 0), (177, 131, 11, 33, 3, 2)]:
        out = np.zeros((h, w + 5), np.uint8)
        assert hc.hc_synth_frame(w, h, seq, t, div, noise, out.ctypes.data, out.strides[0]) == 0
        want = synth.sequence_frame(w, h, seq, t, amplitude_div=div, noise=noise)
        assert np.array_equal(out[:, :w], want), (w, h, seq, t, div, noise)
        assert not out[:, w:].any()
    assert hc.hc_synth_frame(0, 10, 0, 0, 1, 6, None, 0) == -1


def test_synth_header_content_classes_match_python(hc):
    """vsg_synth_content_frame (C) == synth.content_frame (Python) for every GENERATED content class, byte for byte (the
    photo classes are data read from tests/golden/photos_v1.npz, not generated: no C mirror)."""
    hc.hc_synth_content_frame.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint, C.c_int, C.c_void_p, C.c_size_t]
    for kind, name in enumerate(synth.SYNTH_CLASSES):
        for (w, h, seq, t) in ((160, 120, 5, 0), (333, 241, 4099, 7)):
            out = np.zeros((h, w + 5), np.uint8)
            assert hc.hc_synth_content_frame(kind, w, h, seq, t, out.ctypes.data, out.strides[0]) == 0
            assert np.array_equal(out[:, :w], synth.content_frame(name, w, h, seq, t)), (name, w, h, seq, t)
    assert hc.hc_synth_content_frame(99, 10, 10, 0, 0, None, 0) == -1


def test_synth_header_vocabulary_matches_python(hc):
    """vsg_synth_vocabulary (C) == synth.synthetic_vocabulary (Python), byte for byte, incl. the reference-scale tree."""
    hc.hc_synth_vocabulary.restype = C.c_size_t
    hc.hc_synth_vocabulary.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_size_t]
    for (k, L, seed, sc_, wt, stop) in [(10, 3, 1, 0, 0, 0.02), (8, 3, 83, 1, 1, 0.02), (6, 4, 64, 2, 2, 0.3),
                                        (10, 6, 7, 0, 0, 0.02)]:
        want = synth.synthetic_vocabulary(k, L, seed=seed, scoring=sc_, weighting=wt, stop_fraction=stop)
        assert hc.hc_synth_vocabulary(k, L, seed, sc_, wt, stop, None, 0) == len(want)
        out = np.zeros(len(want), np.uint8)
        assert hc.hc_synth_vocabulary(k, L, seed, sc_, wt, stop, out.ctypes.data, len(out)) == len(want)
        assert out.tobytes() == want, (k, L, seed)
