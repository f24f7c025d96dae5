"""Parity on more than "rectangles + noise" (VERDICT r3 #2): every content class of synth.CONTENT_CLASSES -- multi-octave
value noise, 1 / 2 px checkerboards, gratings, defocus, saturated blocks, smooth and steep ramps, salt and pepper --
stage by stage against the oracle on single frames and as batches large enough to take the several-cells-per-workgroup
form of k_fast_cells.  These classes reach what the default class leaves cold: cells where nearly every pixel passes the
necessary test (one-sided: checkerboards; two-sided: the steep ramp -> the queue's overflow / `single` re-unpack +
bright-side retry, vsg_kernels.hip k_fast_cells), levels where NO cell holds a corner at iniThFAST (the minThFAST second
pass for every cell, ORBextractor.cc:848-851), empty cells and nearly empty frames."""
import numpy as np
import pytest

import oracle_lib as ol
from test_gpu_extract import assert_same_output
from visual_sgraphs_amd import orb, synth

pytestmark = pytest.mark.gpu
CLASSES = [c for c in synth.CONTENT_CLASSES]


@pytest.mark.parametrize("kind", CLASSES)
@pytest.mark.parametrize("w,h,nf,ini,mn", [(640, 480, 1000, 20, 7), (752, 480, 1200, 12, 3)])
def test_stage_by_stage_parity_per_content_class(kind, w, h, nf, ini, mn):
    img = synth.content_frame(kind, w, h, 40, 1)
    ref = ol.OracleExtractor(nf, 1.2, 8, ini, mn)
    want = ref(img)
    ex = orb.ORBextractor(nf, 1.2, 8, ini, mn)
    got = ex(img)
    for l in range(8):
        assert np.array_equal(ex.image_pyramid(l), ref.pyramid_level(l)), f"pyramid level {l}"
        gx, gy, gr = ex.candidates(l)
        rx, ry, rr = ref.candidates(l)
        assert len(gx) == len(rx), f"{kind}: candidate count level {l}: {len(gx)} vs {len(rx)}"
        go, ro = np.lexsort((gx, gy)), np.lexsort((rx, ry))
        assert np.array_equal(gx[go], rx[ro]) and np.array_equal(gy[go], ry[ro]) and np.array_equal(gr[go], rr[ro])
        sx, sy, sr = ex.selected(l)
        lk = ref.level_keypoints(l)
        assert np.array_equal(sx + 16, lk["x"].astype(np.int32)) and np.array_equal(sy + 16, lk["y"].astype(np.int32))
        rb = ref.blurred_level(l)
        if rb is not None:
            assert np.array_equal(ex.blurred_level(l), rb), f"blurred level {l}"
    assert_same_output(got, want, f"{kind} {w}x{h}")


@pytest.mark.parametrize("kind", CLASSES)
def test_batches_per_content_class(kind):
    """96 frames of one class (55 392 cells: the launcher takes 3 cells per workgroup with the next tile in flight), ten
    of them compared with the oracle, and the class's candidate multiset of the last frame."""
    B, W, H = 96, 640, 480
    imgs = np.stack([synth.content_frame(kind, W, H, 41, t) for t in range(B)])
    ex = orb.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
    ref = ol.OracleExtractor(1000, 1.2, 8, 20, 7)
    outs = ex.extract_batch(imgs)
    for t in (0, 1, 2, 31, 32, 47, 63, 64, 94, 95):
        assert_same_output(outs[t], ref(imgs[t]), f"{kind} frame {t}")
    for l in range(8):
        gx, gy, gr = ex.candidates(l, frame=B - 1)
        ox, oy, orr = ref.candidates(l)
        assert sorted(zip(gx.tolist(), gy.tolist(), gr.tolist())) == sorted(zip(ox.tolist(), oy.tolist(), orr.tolist())), l


def test_the_cold_paths_are_really_reached():
    """What the classes are FOR, stated on the oracle's numbers: a level without a single corner at iniThFAST whose
    keypoints all come from the minThFAST pass (checker1 level 1+, value noise), and frames that fill no quota."""
    ref = ol.OracleExtractor(1000, 1.2, 8, 20, 7)
    _, k, _ = ref(synth.content_frame("value_noise", 640, 480, 40, 1))
    assert len(k) > 900 and (k["response"] < 20).mean() > 0.9      # nearly everything from the second pass
    _, k, _ = ref(synth.content_frame("ramp", 640, 480, 40, 1))
    assert len(k) < 800                                             # the quota is not met: short lists everywhere
    ref(synth.content_frame("sawtooth", 640, 480, 40, 1))
    assert len(ref.candidates(0)[0]) == 0 and len(ref.candidates(1)[0]) > 1500


def test_the_photographs_are_the_dense_mixed_regime():
    """What the photo classes are FOR (VERDICT r5 "missing" #3), stated on the oracle's numbers: several times the
    candidates of the generated classes at level 0 (the octree's memory-resident form: more than 2048 per level), and
    iniThFAST cells, minThFAST cells and empty cells inside ONE frame."""
    ref = ol.OracleExtractor(1000, 1.2, 8, 20, 7)
    _, k, _ = ref(synth.content_frame("photo_china", 640, 480, 40, 1))
    n_china = len(ref.candidates(0)[0])
    r0 = ref.candidates(0)[2]
    assert n_china > 4000 and (r0 < 20).any() and (r0 >= 20).mean() > 0.5   # both thresholds inside level 0
    ref(synth.sequence_frame(640, 480, 40, 1))
    assert n_china > 2.5 * len(ref.candidates(0)[0])
    _, k, _ = ref(synth.content_frame("photo_hopper", 640, 480, 40, 1))
    assert 0.1 < (k["response"] < 20).mean() < 0.6                           # a real share of retry keypoints
    ref4 = ol.OracleExtractor(2000, 1.2, 8, 20, 7)
    ref4(synth.content_frame("photo_china", 1280, 720, 40, 1))
    assert len(ref4.candidates(0)[0]) > 15000
