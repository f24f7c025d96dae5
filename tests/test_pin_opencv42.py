"""Pins the oracle (and, with -m gpu, the HIP path) against the REFERENCE's own ORBextractor.cc linked with a real
OpenCV 4.2 -- when tests/golden/opencv42_v1.npz exists.  That file is produced by tools/pin_with_opencv/ (a
five-minute job in the reference's Docker image, Dockerfile:1,20); it cannot be produced in the authoring image (no
OpenCV), so until someone runs the recipe these tests skip and parity stays "unpinned" (DESIGN.md section 2)."""
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as ol
from visual_sgraphs_amd import synth

PIN = Path(__file__).parent / "golden" / "opencv42_v1.npz"
pytestmark = pytest.mark.skipif(not PIN.exists(), reason="tests/golden/opencv42_v1.npz absent: run "
                                "tools/pin_with_opencv (needs OpenCV 4.2 + the reference checkout)")
TAPS_42, TAPS_43 = [18, 34, 49, 55, 49, 34, 18], [18, 34, 48, 56, 48, 34, 18]


def _pin():
    return np.load(PIN)


def _cases(P):
    return sorted({k.split("/")[0] for k in P.files if k.endswith("/params")})


def _taps(P):
    """the 8.8 taps the pinned OpenCV really used, from its response to a line image and to a flat image"""
    flat = int(P["gauss/flat200_response"].ravel()[3])
    line = P["gauss/line_response"].ravel().astype(int).tolist()
    for t in (TAPS_42, TAPS_43):
        s = sum(t)
        if [(s * k * 255 + 32768) >> 16 for k in t] == line and (200 * s * s + 32768) >> 16 == flat:
            return t
    raise AssertionError(f"neither known tap table explains the pinned blur: line {line}, flat {flat}")


def test_gaussian_taps_and_gray_coefficients_are_the_tables_in_use():
    P = _pin()
    taps = _taps(P)
    assert taps == TAPS_42, f"OpenCV {bytes(P['opencv_version']).decode()} uses {taps}: switch the default table"
    rgb = P["gray/rgb"].astype(np.int64)
    for name, order in (("rgb2gray", (0, 1, 2)), ("bgr2gray", (2, 1, 0))):
        want = P["gray/" + name].ravel()
        got = (rgb[:, order[0]] * 4899 + rgb[:, order[1]] * 9617 + rgb[:, order[2]] * 1868 + 8192) >> 14
        assert np.array_equal(got, want), "gray coefficients differ from 4899/9617/1868 >> 14"
        assert np.array_equal(ol.cvt_gray(P["gray/rgb"].reshape(64, 64, 3), rgb_order=(name == "rgb2gray")).ravel(), want)


def test_standalone_opencv_pieces():
    P = _pin()
    img = synth.frame(320, 240, 42)
    assert np.array_equal(ol.resize_linear(img, 267, 200), P["cv/resize_267x200"])
    for th in (20, 7):
        x, y, r = ol.fast9_16(img, th, True)
        assert np.array_equal(np.stack([x, y, r], 1), P[f"cv/fast{th}"])
    a = P["cv/fastatan2_args"]
    got = np.array([ol.fast_atan2(float(yy), float(xx)) for yy, xx in a], np.float32)
    assert got.tobytes() == P["cv/fastatan2"].tobytes()


@pytest.mark.parametrize("stage", ["pyramid", "blur", "keypoints"])
def test_oracle_equals_reference_extractor(stage):
    P = _pin()
    taps = _taps(P)
    for name in _cases(P):
        w, h, seed, div, nf, nl, lap0, lap1 = P[name + "/params"].tolist()
        e = ol.OracleExtractor(nf, 1.2, nl, 20, 7)
        e.set_blur_taps(taps)
        mono, kps, desc = e(synth.frame(w, h, seed, amplitude_div=div), (lap0, lap1))
        if stage == "pyramid":
            for l in range(nl):
                assert np.array_equal(e.pyramid_level(l, with_border=True), P[f"{name}/pyr{l}"]), (name, l)
        elif stage == "blur":
            for l in range(nl):
                b = e.blurred_level(l)
                if b is not None:
                    assert np.array_equal(b, P[f"{name}/blur{l}"]), (name, l)
        else:
            assert mono == int(P[name + "/mono"][0])
            assert kps.tobytes() == P[name + "/kps"].tobytes(), name
            assert np.array_equal(desc, P[name + "/desc"].reshape(-1, 32)), name


@pytest.mark.gpu
def test_hip_path_equals_reference_extractor():
    from visual_sgraphs_amd import orb
    P = _pin()
    taps = _taps(P)
    for name in _cases(P):
        w, h, seed, div, nf, nl, lap0, lap1 = P[name + "/params"].tolist()
        ex = orb.ORBextractor(nf, 1.2, nl, 20, 7)
        ex.capacity(h, w)
        ex.set_blur_taps(taps)
        mono, kps, desc = ex(synth.frame(w, h, seed, amplitude_div=div), vLappingArea=(lap0, lap1))
        assert mono == int(P[name + "/mono"][0])
        assert kps.tobytes() == P[name + "/kps"].tobytes() and np.array_equal(desc, P[name + "/desc"].reshape(-1, 32))
        for l in range(nl):
            assert np.array_equal(ex.image_pyramid(l, with_border=True), P[f"{name}/pyr{l}"])
            assert np.array_equal(ex.blurred_level(l), P[f"{name}/blur{l}"])
