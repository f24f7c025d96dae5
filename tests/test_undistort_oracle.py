"""oracle/undistort_oracle.cpp (Frame::UndistortKeyPoints / ComputeImageBounds + the cv::undistortPoints they call,
Frame.cc:891-955) against a numpy float64 restatement written from the definition, the forward distortion model, and
known answers for the two BASELINE cameras that take this path (TUM1: C1, D435i: C5).  CPU only."""
import numpy as np
import pytest

import oracle_lib as ol


def np_undistort(xy, cam, iters=5):
    """[OCV 4.2] cvUndistortPointsInternal, R = I, P = K, TermCriteria(MAX_ITER, 5): numpy float64, same operation order."""
    fx, fy, cx, cy = (np.float64(np.float32(v)) for v in cam["K4"])
    k = np.zeros(14, np.float64)
    k[:len(cam["dist"])] = np.asarray(cam["dist"], np.float32).astype(np.float64)
    ifx, ify = np.float64(1.0) / fx, np.float64(1.0) / fy
    out = np.zeros((len(xy), 2), np.float32)
    for i, (xi, yi) in enumerate(np.asarray(xy, np.float32)):
        u, v = np.float64(xi), np.float64(yi)
        x, y = (u - cx) * ifx, (v - cy) * ify
        x0, y0 = x, y
        for _ in range(iters):
            r2 = x * x + y * y
            icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
            if icdist < 0:
                x, y = (u - cx) * ifx, (v - cy) * ify
                break
            dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2
            dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2
            x, y = (x0 - dx) * icdist, (y0 - dy) * icdist
        out[i] = (np.float32(fx * x + cx), np.float32(fy * y + cy))
    return out


def np_distort(xy_un, cam):
    """The forward model cv::undistortPoints inverts: undistorted pixel -> distorted pixel."""
    fx, fy, cx, cy = (np.float64(np.float32(v)) for v in cam["K4"])
    d = np.zeros(5)
    d[:len(cam["dist"])] = np.asarray(cam["dist"], np.float32)
    k1, k2, p1, p2, k3 = d
    x, y = (xy_un[:, 0].astype(np.float64) - cx) / fx, (xy_un[:, 1].astype(np.float64) - cy) / fy
    r2 = x * x + y * y
    rad = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
    xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return np.stack([fx * xd + cx, fy * yd + cy], 1)


@pytest.mark.parametrize("name", ["tum1", "d435i"])
def test_oracle_equals_the_numpy_restatement_bit_for_bit(name):
    cam = ol.CAMERAS[name]
    rng = np.random.default_rng(5)
    xy = np.concatenate([rng.uniform(-20, 660, (4000, 2)), [[0, 0], [640, 0], [0, 480], [640, 480], [16.0, 16.0]]])
    xy = xy.astype(np.float32)
    got, want = ol.undistort_points(xy, cam), np_undistort(xy, cam)
    assert got.view(np.uint32).tolist() == want.view(np.uint32).tolist()
    half = ol.scaled_camera(name, 320, 240)
    assert np.array_equal(ol.undistort_points(xy / 2, half).view(np.uint32), np_undistort(xy / 2, half).view(np.uint32))


@pytest.mark.parametrize("name,tol", [("tum1", 0.06), ("d435i", 0.002)])
def test_five_iterations_invert_the_forward_model_inside_the_image(name, tol):
    """Definition-level check: distort(undistort(p)) = p up to what 5 fixed-point iterations leave (TUM1's strong k2 / k3
    converge slowly towards the corners; D435i is mild)."""
    cam = ol.CAMERAS[name]
    g = np.stack(np.meshgrid(np.arange(40, 601, 40), np.arange(40, 441, 40)), -1).reshape(-1, 2).astype(np.float32)
    back = np_distort(ol.undistort_points(g, cam), cam)
    assert np.abs(back - g).max() < tol


def test_known_bounds_of_the_baseline_cameras():
    """Frame::ComputeImageBounds for C1 / C5: fractional, not (0, 0, W, H) -- TUM1's corners move INWARDS (features near
    the image edge fall outside the grid and PosInGrid drops them), D435i's outwards (negative mins)."""
    t = ol.image_bounds(ol.CAMERAS["tum1"])
    d = ol.image_bounds(ol.CAMERAS["d435i"])
    assert t == pytest.approx((10.801185, 14.668615, 626.04785, 473.31189), abs=1e-4)
    assert d == pytest.approx((-1.5614729, -0.12478906, 633.60577, 479.47964), abs=1e-4)
    assert t[0] > 0 and t[1] > 0 and d[0] < 0 and d[1] < 0
    # mfGridElementWidthInv / HeightInv (Frame.cc:378-379) are no longer 64 / W, 48 / H
    assert np.float32(64) / (np.float32(d[2]) - np.float32(d[0])) != np.float32(64) / np.float32(640)


def test_undistort_keypoints_is_the_identity_without_distortion_and_keeps_the_other_fields():
    k = np.zeros(5, ol.KP_DTYPE)
    k["x"], k["y"] = [10, 200, 333.5, 600, 639], [5, 100, 240.25, 470, 479]
    k["octave"], k["angle"], k["response"], k["size"], k["class_id"] = [0, 1, 2, 3, 7], 33.0, 50.0, 31.0, -1
    flat = dict(K4=ol.CAMERAS["tum1"]["K4"], dist=(0.0, -0.9, 0.1, 0.2), size=(640, 480))  # mDistCoef(0) == 0 -> copy
    assert ol.undistort_keypoints(k, flat).tobytes() == k.tobytes()
    assert ol.image_bounds(flat) == (0.0, 0.0, 640.0, 480.0)
    un = ol.undistort_keypoints(k, ol.CAMERAS["tum1"])
    for f in ("octave", "angle", "response", "size", "class_id"):
        assert np.array_equal(un[f], k[f])
    xy = ol.undistort_points(np.stack([k["x"], k["y"]], 1), ol.CAMERAS["tum1"])
    assert np.array_equal(un["x"], xy[:, 0]) and np.array_equal(un["y"], xy[:, 1]) and not np.array_equal(un["x"], k["x"])
