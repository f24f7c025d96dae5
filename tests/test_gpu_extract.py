"""GPU parity tests of the extractor: every stage and the final operator() output of the HIP path, called
through the C ABI, must equal the CPU oracle bit for bit (integer/byte/index work: exact; the only float
fields -- angle, scaled coordinates -- are compared as raw bytes too)."""
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as ol
from visual_sgraphs_amd import orb, synth

pytestmark = pytest.mark.gpu

GOLDEN = np.load(Path(__file__).parent / "golden" / "orb_golden_v1.npz")


def assert_same_output(got, want, what=""):
    gm, gk, gd = got
    wm, wk, wd = want
    assert len(gk) == len(wk), f"{what}: keypoint count {len(gk)} != {len(wk)}"
    assert gm == wm, f"{what}: monoIndex {gm} != {wm}"
    for f in ("octave", "x", "y", "response", "size", "class_id", "angle"):
        bad = np.nonzero(gk[f].view(np.uint32) != wk[f].view(np.uint32))[0]
        assert len(bad) == 0, f"{what}: field {f} differs at {bad[:5]}: {gk[f][bad[:5]]} vs {wk[f][bad[:5]]}"
    assert gk.tobytes() == wk.tobytes()
    bad = np.nonzero((gd != wd).any(axis=1))[0]
    assert len(bad) == 0, f"{what}: descriptors differ at rows {bad[:5]}"


CASES = [
    # w, h, nfeatures, nlevels, seed, amplitude_div
    (640, 480, 1000, 8, 0, 1),     # C2
    (640, 480, 1000, 8, 1, 8),     # low contrast: minThFAST fallback cells
    (752, 480, 1200, 8, 2, 1),     # C3 geometry, two initial octree nodes
    (320, 240, 500, 4, 3, 1),
    (640, 480, 1250, 8, 4, 1),     # C5 parameters
]


@pytest.mark.parametrize("w,h,nf,nl,seed,div", CASES)
def test_stage_by_stage_parity(w, h, nf, nl, seed, div):
    img = synth.frame(w, h, seed, amplitude_div=div)
    ref = ol.OracleExtractor(nf, 1.2, nl, 20, 7)
    want = ref(img)
    ex = orb.ORBextractor(nf, 1.2, nl, 20, 7)
    got = ex(img)
    for l in range(nl):
        assert ex.level_size(l) == ref.level_size(l)
        assert np.array_equal(ex.image_pyramid(l), ref.pyramid_level(l)), f"pyramid level {l}"
        # FAST candidates: same multiset (the GPU list is unordered by design)
        gx, gy, gr = ex.candidates(l)
        rx, ry, rr = ref.candidates(l)
        assert len(gx) == len(rx), f"candidate count level {l}: {len(gx)} vs {len(rx)}"
        go, ro = np.lexsort((gx, gy)), np.lexsort((rx, ry))
        assert np.array_equal(gx[go], rx[ro]) and np.array_equal(gy[go], ry[ro]) and np.array_equal(gr[go], rr[ro])
        # octree selection: same keypoints in the same (list) order
        sx, sy, sr = ex.selected(l)
        lk = ref.level_keypoints(l)
        assert np.array_equal(sx + 16, lk["x"].astype(np.int32)) and np.array_equal(sy + 16, lk["y"].astype(np.int32))
        assert np.array_equal(sr, lk["response"].astype(np.int32))
        rb = ref.blurred_level(l)
        if rb is not None:
            assert np.array_equal(ex.blurred_level(l), rb), f"blurred level {l}"
    assert_same_output(got, want, f"{w}x{h}")
    if div == 8:
        assert np.any(want[1]["response"] < 20)  # the fallback branch really ran


def _reference_configs():
    """every (image size, ORBextractor.*) tuple of the reference's settings files (tests/golden/make_reference_configs.py)"""
    import json
    cases = json.loads((Path(__file__).parent / "golden" / "reference_configs.json").read_text())["cases"]
    return [tuple(c["params"][k] for k in ("w", "h", "nFeatures", "scaleFactor", "nLevels", "iniThFAST", "minThFAST"))
            for c in cases]


@pytest.mark.parametrize("w,h,nf,sf,nl,ini,mn", _reference_configs())
def test_every_reference_configuration(w, h, nf, sf, nl, ini, mn):
    """The 16 distinct extractor configurations of the reference's 49 settings files (EuRoC, KITTI, TUM, TUM-VI,
    RealSense D435i / T265, ...): single frame and a 3-frame batch, operator() output bit for bit."""
    ref = ol.OracleExtractor(nf, sf, nl, ini, mn)
    ex = orb.ORBextractor(nf, sf, nl, ini, mn, max_batch=3)
    seed = (w * 31 + h * 17 + nf + ini) % 1000
    imgs = [synth.frame(w, h, seed + i, amplitude_div=1 if i < 2 else 6) for i in range(3)]
    wants = [ref(im) for im in imgs]
    assert_same_output(ex(imgs[0]), wants[0], f"{w}x{h}/{nf}/{nl}/{ini}")
    for got, want in zip(ex.extract_batch(np.stack(imgs)), wants):
        assert_same_output(got, want, f"batch {w}x{h}/{nf}/{nl}/{ini}")


@pytest.mark.parametrize("w,h,nf,nl", [(1920, 1080, 3000, 8), (2400, 2336, 1500, 3)])
def test_large_images_take_the_other_candidate_list_forms(w, h, nf, nl):
    """1920x1080 / 3000 features: level 0 yields more FAST candidates than the octree keeps in registers, so the
    per-cell segments are compacted into the scratch list first.  2400x2336: 4 355 cells on level 0, more than the
    octree can prefix in LDS (vsg_common.h kOctMaxCells), so k_fast_cells appends to one list per level instead."""
    img = synth.frame(w, h, 4321)
    ref = ol.OracleExtractor(nf, 1.2, nl, 20, 7)
    want = ref(img)
    ex = orb.ORBextractor(nf, 1.2, nl, 20, 7)
    got = ex(img)
    gx, gy, gr = ex.candidates(0)
    rx, ry, rr = ref.candidates(0)
    go, ro = np.lexsort((gx, gy)), np.lexsort((rx, ry))
    assert len(gx) == len(rx) and np.array_equal(gx[go], rx[ro]) and np.array_equal(gy[go], ry[ro])
    assert np.array_equal(gr[go], rr[ro])
    if nl == 8:
        assert len(rx) > 2048
    assert_same_output(got, want, f"{w}x{h}")


def test_pyramid_with_border_matches_mvImagePyramid():
    img = synth.frame(320, 240, 12)
    ref = ol.OracleExtractor(500, 1.2, 4, 20, 7)
    ref(img)
    ex = orb.ORBextractor(500, 1.2, 4, 20, 7)
    ex(img)
    for l in range(4):
        assert np.array_equal(ex.image_pyramid(l, with_border=True), ref.pyramid_level(l, with_border=True))


@pytest.mark.parametrize("name", sorted({k.split("/")[0] for k in GOLDEN.files if k.endswith("/params")}))
def test_golden_fixtures(name):
    w, h, seed, div, nf, nl, lap0, lap1 = GOLDEN[name + "/params"].tolist()
    ex = orb.ORBextractor(nf, 1.2, nl, 20, 7)
    mono, kps, desc = ex(synth.frame(w, h, seed, amplitude_div=div), None, (lap0, lap1))
    assert mono == int(GOLDEN[name + "/mono"][0])
    assert kps.tobytes() == GOLDEN[name + "/kps"].tobytes()
    assert np.array_equal(desc, GOLDEN[name + "/desc"])


def test_empty_constant_and_lapping_slots():
    ex = orb.ORBextractor(500, 1.2, 4, 20, 7)
    ref = ol.OracleExtractor(500, 1.2, 4, 20, 7)
    assert ex(None)[0] == -1 and ex(np.zeros((0, 0), np.uint8))[0] == -1
    mono, kps, desc = ex(synth.constant_frame(320, 240))
    assert (mono, len(kps), desc.shape) == (0, 0, (0, 32))  # _descriptors.release() path
    img = synth.frame(320, 240, 2)
    for lap in [(0, 0), (0, 1000), (100, 200), (150, 150), (-5, 40)]:
        assert_same_output(ex(img, None, lap), ref(img, lap), f"lapping {lap}")


@pytest.mark.parametrize("w,h,nf", [(752, 480, 1200), (1241, 376, 2000), (1280, 720, 2000)])
def test_wide_frames_with_empty_initial_nodes(w, h, nf):
    """Two to four initial nodes (`round(width / height)`, ORBextractor.cc:566-586) of which some hold no candidate and are
    erased (`:597-608`): frames whose texture is confined to some of the vertical strips, every subset of strips; the
    whole frame textured is the case the other tests cover.  (Round 6: the two-to-four-node case has its own code path.)"""
    ex, ref = orb.ORBextractor(nf, 1.2, 8, 20, 7), ol.OracleExtractor(nf, 1.2, 8, 20, 7)
    nI = int(round(w / h))
    assert 2 <= nI <= 4
    img0 = synth.frame(w, h, 21)
    for mask in range(1, (1 << nI) - 1):
        img = np.full_like(img0, 97)
        for i in range(nI):
            if (mask >> i) & 1:
                a, b = i * w // nI, (i + 1) * w // nI
                img[:, a:b] = img0[:, a:b]
        got, want = ex(img), ref(img)
        assert len(want[1]) > 50
        assert_same_output(got, want, f"{w}x{h} strips {mask:0{nI}b}")


def test_same_handle_different_sizes_and_reuse():
    ex = orb.ORBextractor(800, 1.2, 6, 20, 7)
    ref = ol.OracleExtractor(800, 1.2, 6, 20, 7)
    for (w, h, seed) in [(480, 360, 1), (640, 480, 2), (480, 360, 3), (400, 300, 4)]:
        img = synth.frame(w, h, seed)
        assert_same_output(ex(img), ref(img), f"{w}x{h}")


def test_strided_input_and_unusual_thresholds():
    big = synth.frame(700, 500, 9)
    view = big[10:490, 30:670]  # non-contiguous rows
    ref = ol.OracleExtractor(1000, 1.2, 8, 12, 5)
    ex = orb.ORBextractor(1000, 1.2, 8, 12, 5)
    assert_same_output(ex(view), ref(np.ascontiguousarray(view)), "strided")
    ref2, ex2 = ol.OracleExtractor(300, 1.5, 3, 40, 10), orb.ORBextractor(300, 1.5, 3, 40, 10)
    img = synth.frame(512, 384, 10)
    assert_same_output(ex2(img), ref2(img), "scale 1.5")


def test_batch_equals_single_and_oracle():
    B = 6
    imgs = np.stack([synth.sequence_frame(640, 480, 5, t) for t in range(B)])
    ex = orb.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
    ref = ol.OracleExtractor(1000, 1.2, 8, 20, 7)
    outs = ex.extract_batch(imgs)
    single = orb.ORBextractor(1000, 1.2, 8, 20, 7)
    for t in range(B):
        want = ref(imgs[t])
        assert_same_output(outs[t], want, f"batch frame {t}")
        assert_same_output(single(imgs[t]), want, f"single frame {t}")
    # a partial batch after a full one must not see stale state
    outs2 = ex.extract_batch(imgs[3:5])
    assert_same_output(outs2[0], ref(imgs[3]), "partial batch 0")
    assert_same_output(outs2[1], ref(imgs[4]), "partial batch 1")


def test_full_size_c4_and_kitti_aspect():
    img = synth.frame(1280, 720, 21)
    ref = ol.OracleExtractor(2000, 1.2, 8, 20, 7)
    ex = orb.ORBextractor(2000, 1.2, 8, 20, 7)
    assert_same_output(ex(img), ref(img), "C4 1280x720")
    img = synth.frame(1241, 376, 22)  # 3-4 initial octree nodes
    assert_same_output(ex(img), ref(img), "1241x376")


def test_blur_taps_are_data():
    ed = (18, 34, 48, 56, 48, 34, 18)
    img = synth.frame(320, 240, 30)
    ref = ol.OracleExtractor(500, 1.2, 4, 20, 7)
    ref.set_blur_taps(ed)
    ex = orb.ORBextractor(500, 1.2, 4, 20, 7)
    ex.set_blur_taps(ed)
    assert_same_output(ex(img), ref(img), "ED taps")
    assert np.array_equal(ex.blurred_level(0), ol.gaussian_blur7(img, ed))


@pytest.mark.parametrize("taps", [(18, 34, 49, 55, 49, 34, 18), (18, 34, 48, 56, 48, 34, 18), (1, 0, 0, 0, 0, 0, 255),
                                  (255, 2, 0, 0, 0, 0, 0), (0, 0, 1, 255, 1, 0, 0), (128, 129, 0, 0, 0, 0, 0), (0, 0, 0, 0, 128, 129, 0),
                                  (37, 37, 37, 36, 37, 37, 36), (0, 0, 0, 0, 0, 0, 0)])
def test_blur_saturation_and_extreme_tap_tables(taps):
    """saturate_cast<uchar> of the 8.8 blur: a tap table that sums to 257 (OpenCV 4.2's plain rounding) takes a 255-valued
    area to (257 * 65535 + 32768) >> 16 = 257 -> 255.  The kernel forms the vertical sum 256 times too large and lets the
    clamp bit of v_dot2_u32_u16 saturate it; every level of a frame with white / black blocks, white borders and noise must
    equal the CPU blur for the shipped tables and for tables that put the whole weight on one end of the window."""
    w, h = 338, 262  # ragged: the last tile column and the last strip are partial
    rng = np.random.default_rng(sum(taps) * 7 + taps[0])
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    img[:40] = 255
    img[-9:] = 255
    img[:, :21] = 255
    img[:, -5:] = 255
    img[60:140, 100:260] = 255
    img[150:200, 30:90] = 0
    img[120:131, 280:291] = 254
    ex = orb.ORBextractor(400, 1.2, 4, 20, 7)
    ex.set_blur_taps(taps)
    ref = ol.OracleExtractor(400, 1.2, 4, 20, 7)
    ref.set_blur_taps(taps)
    assert_same_output(ex(img), ref(img), f"taps {taps}")
    for l in range(4):
        want = ol.gaussian_blur7(ref.pyramid_level(l), taps)
        got = ex.blurred_level(l)
        assert np.array_equal(got, want), (taps, l, int((got != want).sum()))
    if sum(taps) == 257:
        assert int(ex.blurred_level(0)[10, 100]) == 255  # the case the clamp exists for


@pytest.mark.parametrize("w,h,nl,sc", [(80, 80, 1, 1.2), (97, 83, 1, 1.2), (131, 80, 1, 1.2), (83, 131, 1, 1.2), (283, 167, 3, 1.2),
                                       (201, 149, 2, 1.5), (1241, 376, 8, 1.2), (644, 116, 2, 1.2), (100, 150, 2, 1.1)])
def test_blurred_levels_on_small_and_odd_geometries(w, h, nl, sc):
    """Every pixel of every blurred level (not only the ones a descriptor samples) on geometries at the small end: levels of
    80-120 rows are two or three 36-row strips whose REFLECT_101 folds at the top and at the bottom fall into the same
    strip or the next one, widths that leave a partial tile column, one-tile-column levels."""
    rng = np.random.default_rng(w * 1000 + h)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    img[: h // 5] = 255
    img[-3:] = 0
    ex = orb.ORBextractor(300, sc, nl, 20, 7)
    ref = ol.OracleExtractor(300, sc, nl, 20, 7)
    assert_same_output(ex(img), ref(img), f"{w}x{h}")
    for l in range(nl):
        want = ol.gaussian_blur7(ref.pyramid_level(l))
        got = ex.blurred_level(l)
        assert got.shape == want.shape and np.array_equal(got, want), (w, h, l, np.argwhere(got != want)[:4].tolist())


def test_unsupported_inputs_return_codes():
    ex = orb.ORBextractor(500, 1.2, 8, 20, 7)
    with pytest.raises(orb.VsgError) as ei:
        ex(synth.frame(160, 120, 1))  # top level narrower than one FAST cell: the reference divides by zero
    assert ei.value.code == -3


def test_two_handles_two_threads_like_stereo_ctor():
    """Frame.cc:129-132: left/right extractors run concurrently on two host threads."""
    import threading
    L, R = synth.sequence_frame(752, 480, 8, 0), synth.sequence_frame(752, 480, 8, 1)
    ref = ol.OracleExtractor(1200, 1.2, 8, 20, 7)
    want = [ref(L), ref(R)]
    exs = [orb.ORBextractor(1200, 1.2, 8, 20, 7), orb.ORBextractor(1200, 1.2, 8, 20, 7)]
    got = [None, None]

    def run(i, img):
        for _ in range(3):
            got[i] = exs[i](img)

    ts = [threading.Thread(target=run, args=(0, L)), threading.Thread(target=run, args=(1, R))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert_same_output(got[0], want[0], "left")
    assert_same_output(got[1], want[1], "right")


def test_roundtrip_properties_at_full_size():
    """Size-independent properties on the bench workload: determinism across repeated calls, quotas, ordering."""
    ex = orb.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=8)
    imgs = np.stack([synth.frame(640, 480, 100 + i) for i in range(8)])
    a = ex.extract_batch(imgs)
    b = ex.extract_batch(imgs)
    q = ex.features_per_level()
    for (m1, k1, d1), (m2, k2, d2) in zip(a, b):
        assert m1 == m2 and k1.tobytes() == k2.tobytes() and np.array_equal(d1, d2)
        assert np.all(np.diff(k1["octave"]) >= 0)
        cnt = np.bincount(k1["octave"], minlength=8)
        assert np.all(cnt <= q + 3) and cnt.sum() == len(k1) >= 900


@pytest.mark.parametrize("channels,rgb", [(3, True), (3, False), (4, True), (4, False)])
def test_colour_input_cvtcolor_fused(channels, rgb):
    """Tracking::GrabImage* cvtColor(RGB/BGR/RGBA/BGRA -> GRAY) fused into the level-0 staging."""
    rng = np.random.default_rng(7)
    base = synth.frame(640, 480, 60)
    col = np.stack([np.clip(base.astype(np.int32) + rng.integers(-20, 21, base.shape), 0, 255).astype(np.uint8)
                    for _ in range(channels)], axis=-1)
    gray = ol.cvt_gray(col, rgb)
    # the oracle's conversion equals the definition
    r, b = (col[..., 0], col[..., 2]) if rgb else (col[..., 2], col[..., 0])
    assert np.array_equal(gray, ((r.astype(np.int64) * 4899 + col[..., 1].astype(np.int64) * 9617 +
                                  b.astype(np.int64) * 1868 + 8192) >> 14).astype(np.uint8))
    ref = ol.OracleExtractor(1000, 1.2, 8, 20, 7)
    want = ref(gray)
    ex = orb.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=2)
    got = ex.extract_batch_color(np.stack([col, col]), rgb)
    assert np.array_equal(ex.image_pyramid(0), gray)
    assert_same_output(got[0], want, "colour frame 0")
    assert_same_output(got[1], want, "colour frame 1")


def test_device_sort_replay_equals_std_sort():
    """The octree's device sort (wave-parallel quicksort partitioning + stable rank) against the real std::sort
    (tests/_hostcore, same libstdc++ as the oracle) on tie-heavy inputs: the ORDER of equal keys must match."""
    import ctypes as C
    import subprocess
    hc_dir = Path(__file__).resolve().parent / "_hostcore"
    subprocess.check_call(["make", "-C", str(hc_dir)], stdout=subprocess.DEVNULL)
    hc = C.CDLL(str(hc_dir / "libvsg_hostcore.so"))
    hc.hc_std_sort.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    rng = np.random.default_rng(3)
    for trial in range(300):
        n = int(rng.integers(1, 300)) if trial % 20 else int(rng.integers(300, 2048))
        nkeys = int(rng.choice([1, 2, 3, 8, 50, 100000]))
        keys = rng.integers(0, nkeys, n).astype(np.uint64)
        if trial % 5 == 0:
            keys = np.sort(keys)[::-1].copy() if trial % 10 == 0 else np.sort(keys)
        if trial % 7 == 3:  # organ pipe / many-duplicates patterns that stress the median-of-3 pivot
            keys = np.concatenate([np.arange(n // 2), np.arange(n - n // 2)[::-1]]).astype(np.uint64) // np.uint64(3)
        items = (keys << np.uint64(32)) | np.arange(n, dtype=np.uint64)
        want = items.copy()
        hc.hc_std_sort(want.ctypes.data_as(C.POINTER(C.c_uint64)), n)
        got = orb.debug_device_sort(items)
        assert np.array_equal(got, want), (trial, n, nkeys)


@pytest.mark.parametrize("nsub", [None, "2", "3"])
def test_large_batch_and_sub_batches(nsub, monkeypatch):
    """A 259-frame batch (32 x 8 + 3: the XCD block remap covers 256 frames, the tail keeps the plain mapping), as
    one batch (default) and cut into 2 / 3 sub-batches on separate stream pairs (VSG_SUBBATCH, read when the handle
    is created): results per frame are the same as single-frame calls (checked against the oracle around the cuts
    and at both ends)."""
    if nsub is None:
        monkeypatch.delenv("VSG_SUBBATCH", raising=False)
    else:
        monkeypatch.setenv("VSG_SUBBATCH", nsub)
    B = 259
    imgs = np.stack([synth.sequence_frame(320, 240, 9, t % 40) for t in range(B)])
    ex = orb.ORBextractor(500, 1.2, 4, 20, 7, max_batch=B)
    ref = ol.OracleExtractor(500, 1.2, 4, 20, 7)
    outs = ex.extract_batch(imgs)
    assert len(outs) == B
    for t in (0, 1, 86, 87, 128, 129, 130, 131, 173, 174, 255, 256, 257, 258):
        assert_same_output(outs[t], ref(imgs[t]), f"frame {t} of {B}")
    # frames with identical content give identical output wherever they sit in the batch
    for t in range(40, B):
        assert outs[t][1].tobytes() == outs[t - 40][1].tobytes() and np.array_equal(outs[t][2], outs[t - 40][2])


def _fuzz_cases(n, seed):
    rng = np.random.default_rng(seed)
    cases = []
    while len(cases) < n:
        w, h = int(rng.integers(150, 1000)), int(rng.integers(120, 760))
        nl = int(rng.integers(1, 9))
        sc = float(rng.choice([1.1, 1.2, 1.25, 1.3, 1.5, 2.0, 2.5]))
        nf = int(rng.integers(50, 3000))
        ini = int(rng.integers(8, 60))
        mn = int(rng.integers(2, ini + 1))
        div = int(rng.choice([1, 1, 1, 2, 4, 8]))
        top = sc ** (nl - 1)
        # every level holds a FAST cell; 1..4 initial octree nodes (nIni = round(width / height) must be >= 1: the
        # reference divides by it, ORBextractor.cc:566-571)
        if nl == 1 and nf > 2400:  # one level would hold more than kMaxQuota features (refused, see the test below)
            continue
        if w / top < 80 or h / top < 80 or w / h > 3.5 or w < 0.6 * h:
            continue
        cases.append((w, h, nf, sc, nl, ini, mn, div, int(rng.integers(0, 1 << 20))))
    return cases


@pytest.mark.parametrize("case", _fuzz_cases(40, 2024), ids=lambda c: f"{c[0]}x{c[1]}_n{c[2]}_s{c[3]}_l{c[4]}_t{c[5]}-{c[6]}")
def test_fuzz_random_geometries(case):
    """Seeded random image sizes (odd widths, widths not divisible by 4), level counts, scale factors (incl. > 2: the
    per-level resize fallback), feature counts and thresholds: whatever the reference can process must come out
    bit-identical (the generator keeps every level large enough for one FAST cell, which the reference needs)."""
    w, h, nf, sc, nl, ini, mn, div, seed = case
    img = synth.frame(w, h, seed, amplitude_div=div)
    want = ol.OracleExtractor(nf, sc, nl, ini, mn)(img)
    got = orb.ORBextractor(nf, sc, nl, ini, mn)(img)
    assert_same_output(got, want, f"fuzz {case}")


def _contrast_ladder(w, h, seed):
    """Corners at every contrast from 1 to 255 gray levels (bright-on-dark and dark-on-bright squares on a noise-free
    background, plus a noisy band): thresholds right at a corner's contrast exercise the 6-bit necessary test's margin
    (it must never drop a pixel the exact 9-bit comparison passes) and both polarities of the one-sided score."""
    rng = np.random.default_rng(seed)
    img = np.full((h, w), 128, np.uint8)
    for i in range(256):
        x, y = 20 + (i % 24) * ((w - 60) // 24), 20 + (i // 24) * ((h - 60) // 11)
        base = int(rng.integers(0, 256 - i)) if i else 100
        img[y - 6:y + 14, x - 6:x + 14] = base                      # pedestal
        img[y:y + 7, x:x + 7] = base + i if (i & 1) else base       # brighter square ...
        if not (i & 1):
            img[y - 6:y + 14, x - 6:x + 14] = base + i              # ... or a darker one on a brighter pedestal
            img[y:y + 7, x:x + 7] = base
    band = img[h - 40:h - 10].astype(np.int32) + rng.integers(-9, 10, (30, w))
    img[h - 40:h - 10] = np.clip(band, 0, 255).astype(np.uint8)
    return img


@pytest.mark.parametrize("ini,mn", [(1, 1), (2, 1), (3, 2), (4, 3), (5, 5), (6, 2), (7, 7), (9, 8), (10, 3), (11, 10),
                                    (19, 18), (20, 7), (21, 20), (22, 21), (23, 5), (62, 61), (63, 62), (64, 63), (65, 64),
                                    (127, 126), (128, 127), (200, 199), (250, 249), (253, 252), (254, 253), (255, 254),
                                    (20, 25), (7, 20)])
def test_fast_thresholds_around_every_contrast_step(ini, mn):
    img = _contrast_ladder(640, 480, ini * 257 + mn)
    want = ol.OracleExtractor(1500, 1.2, 4, ini, mn)
    ex = orb.ORBextractor(1500, 1.2, 4, ini, mn)
    got = ex(img)
    assert_same_output(got, want(img), f"thresholds {ini}/{mn}")
    for l in range(4):  # the candidate multiset of every level, before the octree picks
        gx, gy, gr = ex.candidates(l)
        ox, oy, orr = want.candidates(l)
        assert sorted(zip(gx.tolist(), gy.tolist(), gr.tolist())) == sorted(zip(ox.tolist(), oy.tolist(), orr.tolist())), (ini, mn, l)


def test_single_level_with_large_quota_needs_more_than_64kb_of_lds():
    """nlevels = 1 puts the whole feature budget on one level: quota 2400 -> a 150 KB octree workspace (the
    launch raises the dynamic-LDS limit); beyond kMaxQuota the handle refuses the geometry."""
    img = synth.frame(640, 480, 77)
    ref = ol.OracleExtractor(2400, 1.2, 1, 20, 7)
    assert_same_output(orb.ORBextractor(2400, 1.2, 1, 20, 7)(img), ref(img), "quota 2400")
    with pytest.raises(orb.VsgError) as ei:
        orb.ORBextractor(3000, 1.2, 1, 20, 7)(img)
    assert ei.value.code == -3


@pytest.mark.parametrize("offset,pad", [(0, 0), (1, 5), (3, 2), (0, 6)])
def test_device_entry_with_aligned_and_unaligned_layouts(offset, pad):
    """vsg_orb_extract_batch_device reads level 0 in place when base / stride / frame stride are 4-byte aligned and
    restages otherwise: both must equal the oracle, on the caller's stream, outputs left on the device."""
    torch = pytest.importorskip("torch")
    B, H, W = 3, 240, 322
    imgs = np.stack([synth.frame(W, H, 500 + i) for i in range(B)])
    dev = torch.device("cuda", 0)
    big = torch.zeros((B, H, W + pad + offset + 2), dtype=torch.uint8, device=dev)
    big[:, :, offset:offset + W] = torch.from_numpy(imgs).to(dev)
    view = big[:, :, offset:offset + W]
    ex = orb.ORBextractor(600, 1.2, 5, 20, 7, max_batch=B)
    cap = ex.capacity(H, W)
    d_kps = torch.zeros((B, cap, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_counts = torch.zeros((B, 2), dtype=torch.int32, device=dev)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ex.extract_batch_device(view.data_ptr(), B, view.stride(0), H, W, view.stride(1), d_kps.data_ptr(),
                                d_desc.data_ptr(), d_counts.data_ptr(), cap, (0, 0), stream.cuda_stream)
    stream.synchronize()
    counts = d_counts.cpu().numpy()
    kps = d_kps.cpu().numpy().view(orb.KP_DTYPE).reshape(B, cap)
    desc = d_desc.cpu().numpy()
    ref = ol.OracleExtractor(600, 1.2, 5, 20, 7)
    for f in range(B):
        mono, rk, rd = ref(imgs[f])
        n = int(counts[f, 0])
        assert (n, int(counts[f, 1])) == (len(rk), mono)
        assert kps[f, :n].tobytes() == rk.tobytes() and np.array_equal(desc[f, :n], rd)


def test_stage_timing_modes_do_not_change_results():
    """vsg_orb_enable_timing: 1 = events around every stage (the blur runs as its own launch), 2 = events around the
    FAST launch only (the stage chain of an untimed call); results are the same, and mode 2 fills in "fast" alone."""
    imgs = np.stack([synth.sequence_frame(640, 480, 3, t) for t in range(4)])
    ex = orb.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=4)
    want = [ol.OracleExtractor(1000, 1.2, 8, 20, 7)(im) for im in imgs]
    for mode in (0, 1, 2, 0):
        ex.enable_timing(mode)
        for _ in range(3):  # a timed call's events are harvested by the next one
            outs = ex.extract_batch(imgs)
        for t in range(4):
            assert_same_output(outs[t], want[t], f"timing mode {mode} frame {t}")
        ms = ex.timing_ms()
        if mode == 1:
            assert all(ms[k] > 0 for k in ("pyramid", "fast", "octree", "blur", "orient_desc", "total")), ms
        elif mode == 2:
            assert ms["fast"] > 0 and all(v == 0 for k, v in ms.items() if k != "fast"), ms
