"""Seeded search scenarios for the routine-level matcher tests (CPU: oracle vs oracle, GPU: HIP vs oracle).

Features come from the CPU oracle extractor on synthetic frames (cached per process); the "projections" are the
keypoints of a neighbouring frame moved by the known translation plus noise -- what Tracking's geometry would hand the
matcher -- with every knob of the routines exercised: blocked features, stereo coordinates, level windows, empty
windows, far-off points, duplicate targets."""
import functools

import numpy as np

import oracle_lib as ol
from visual_sgraphs_amd import synth

SCALE = np.float32(1.2) ** np.arange(8, dtype=np.float32)
_T = ol.OracleExtractor(1000, 1.2, 8, 20, 7).tables()
SCALE_FACTORS = _T["scale"]          # mvScaleFactors
INV_SIGMA2 = _T["inv_sigma2"]        # mvInvLevelSigma2
# The camera the scenario frames are seen through: "image" = no distortion (mDistCoef(0) == 0: mvKeysUn =
# mvKeys, bounds = the image rectangle), "tum1" / "d435i" = the two BASELINE cameras whose Frame constructor really
# undistorts (config/RGB-D/TUM1.yaml, config/RGB-D-Inertial/RealSense_D435i.yaml, scaled to the 320x240 scenario frames):
# keys = mvKeysUn through oracle/undistort_oracle.cpp, BOUNDS = Frame::ComputeImageBounds -- fractional, negative
# (D435i) or inside the image (TUM1: features near the edge fall outside the grid).  use_camera() switches all of it.
W, H = 320, 240
CAMERA_NAMES = ("image", "tum1", "d435i")
CAMERA = "image"
CAM = None                        # oracle_lib camera dict of the current camera (None: no distortion)
BOUNDS = (0.0, 0.0, float(W), float(H))


def use_camera(name):
    global CAMERA, CAM, BOUNDS
    assert name in CAMERA_NAMES
    CAMERA = name
    CAM = None if name == "image" else ol.scaled_camera(name, W, H)
    BOUNDS = (0.0, 0.0, float(W), float(H)) if CAM is None else ol.image_bounds(CAM)


@functools.lru_cache(maxsize=None)
def features_raw(seed, t, w=W, h=H, nfeat=600):
    """(mvKeys, mDescriptors) of the oracle extractor on a scenario frame."""
    ex = ol.OracleExtractor(nfeat, 1.2, 8, 20, 7)
    _, k, d = ex(synth.sequence_frame(w, h, seed, t))
    return k, d


@functools.lru_cache(maxsize=None)
def _features_un(camera, seed, t, w, h, nfeat):
    k, d = features_raw(seed, t, w, h, nfeat)
    return ol.undistort_keypoints(k, ol.scaled_camera(camera, w, h)), d


def features(seed, t, w=W, h=H, nfeat=600):
    """(mvKeysUn, mDescriptors) under the current camera (Frame::UndistortKeyPoints, Frame.cc:891-921)."""
    if CAMERA == "image":
        return features_raw(seed, t, w, h, nfeat)
    return _features_un(CAMERA, seed, t, w, h, nfeat)


def stereo_pair(seed):
    """Fisheye-stereo style frame: mvKeys || mvKeysRight, descriptors vconcat'ed, Nleft (Frame.cc:296)."""
    kl, dl = features(seed, 0)
    kr, dr = features(seed, 1)
    return np.concatenate([kl, kr]), np.concatenate([dl, dr]), len(kl)


def u_right_for(kps, rng, frac=0.6):
    """mvuRight: -1 for monocular points, x - disparity for the rest."""
    ur = np.full(len(kps), -1.0, np.float32)
    m = rng.random(len(kps)) < frac
    ur[m] = kps["x"][m] - rng.uniform(1.0, 30.0, m.sum()).astype(np.float32)
    return ur


def projections(rng, src_kps, shift=(-3.0, -2.0), noise=1.5, n_far=20, bounds=None):
    """(u, v) of the source keypoints in the target frame + a few points far from any feature / outside the image."""
    bounds = BOUNDS if bounds is None else bounds
    n = len(src_kps)
    u = src_kps["x"] + np.float32(shift[0]) + rng.normal(0, noise, n).astype(np.float32)
    v = src_kps["y"] + np.float32(shift[1]) + rng.normal(0, noise, n).astype(np.float32)
    far = rng.choice(n, min(n_far, n), replace=False)
    u[far] = rng.uniform(bounds[0] - 40, bounds[2] + 40, len(far)).astype(np.float32)
    v[far] = rng.uniform(bounds[1] - 40, bounds[3] + 40, len(far)).astype(np.float32)
    return u.astype(np.float32), v.astype(np.float32)


def noisy_desc(rng, desc, flip_bits=12):
    """Map-point descriptors: the source frame's descriptors with a few bits flipped."""
    d = desc.copy()
    for _ in range(flip_bits):
        rows = np.arange(len(d))
        byte = rng.integers(0, 32, len(d))
        bit = rng.integers(0, 8, len(d))
        flip = rng.random(len(d)) < 0.5
        d[rows[flip], byte[flip]] ^= (1 << bit[flip]).astype(np.uint8)
    return d


def local_map_scenario(seed, stereo2=False):
    """Inputs of SearchByProjection(F, vpMapPoints, th) (ORBmatcher.cc:42-216)."""
    rng = np.random.default_rng(seed)
    if stereo2:
        keys, desc, nleft = stereo_pair(seed)
        ur = None
    else:
        keys, desc = features(seed, 1)
        nleft = -1
        ur = u_right_for(keys, rng)
    src_k, src_d = features(seed, 0)
    n = len(src_k)
    u, v = projections(rng, src_k)
    mp = dict(desc=noisy_desc(rng, src_d), observed=(rng.random(n) < 0.8).astype(np.uint8),
              in_view=(rng.random(n) < 0.9).astype(np.uint8), proj_x=u, proj_y=v,
              proj_xr=(u - rng.uniform(1, 30, n)).astype(np.float32),
              scale_level=np.clip(src_k["octave"] + rng.integers(-1, 2, n), 0, 7).astype(np.int32),
              view_cos=rng.choice(np.array([0.9, 0.9985, 0.9999], np.float32), n))
    ltr = rtl = None
    if stereo2:
        ur2, vr2 = projections(rng, src_k, shift=(-3.0 + 8.0, -2.0))
        lvl_r = np.clip(src_k["octave"] + rng.integers(-1, 2, n), 0, 7).astype(np.int32)
        lvl_r[rng.random(n) < 0.05] = -1
        mp.update(in_view_r=(rng.random(n) < 0.7).astype(np.uint8), proj_x_r=ur2, proj_y_r=vr2, scale_level_r=lvl_r,
                  view_cos_r=rng.choice(np.array([0.9, 0.9999], np.float32), n))
        nr = len(keys) - nleft
        ltr = np.full(nleft, -1, np.int32)
        rtl = np.full(nr, -1, np.int32)
        pairs = rng.choice(min(nleft, nr), min(nleft, nr) // 3, replace=False)
        perm = rng.permutation(pairs)
        ltr[pairs] = perm
        rtl[perm] = pairs
    blocked = (rng.random(len(keys)) < 0.1).astype(np.uint8)
    th = float(rng.choice([1.0, 3.0, 5.0]))
    return dict(keys=keys, desc=desc, nleft=nleft, u_right=ur, mp=mp, th=th, nnratio=0.8, blocked=blocked, ltr=ltr,
                rtl=rtl)


def last_frame_scenario(seed, stereo2=False):
    """Inputs of SearchByProjection(CurrentFrame, LastFrame, th, bMono) (ORBmatcher.cc:1667-1878)."""
    rng = np.random.default_rng(seed + 1000)
    if stereo2:
        keys, desc, nleft = stereo_pair(seed)
        ur_frame = None
    else:
        keys, desc = features(seed, 1)
        nleft = -1
        ur_frame = u_right_for(keys, rng)
    src_k, src_d = features(seed, 0)
    n = len(src_k)
    u, v = projections(rng, src_k)
    out = dict(keys=keys, desc=desc, nleft=nleft, u_right=ur_frame, q_desc=noisy_desc(rng, src_d),
               observed=(rng.random(n) < 0.7).astype(np.uint8), u=u, v=v,
               ur=(u - rng.uniform(1, 30, n)).astype(np.float32), octave=src_k["octave"].astype(np.int32),
               angle=src_k["angle"].astype(np.float32), th=float(rng.choice([7.0, 15.0])),
               direction=int(rng.integers(0, 3)), blocked=(rng.random(len(keys)) < 0.1).astype(np.uint8), u_r=None,
               v_r=None)
    if stereo2:
        out["u_r"], out["v_r"] = projections(rng, src_k, shift=(-3.0 + 8.0, -2.0))
    return out


def kf_projection_scenario(seed):
    """Per-point inputs shared by the best-only routines: SearchByProjection(KF, Sim3) x2, (F, KF, sAlreadyFound),
    SearchBySim3, Fuse x2."""
    rng = np.random.default_rng(seed + 2000)
    keys, desc = features(seed, 1)
    src_k, src_d = features(seed, 0)
    n = len(src_k)
    u, v = projections(rng, src_k)
    level = np.clip(src_k["octave"] + rng.integers(-1, 2, n), 0, 7).astype(np.int32)
    th = float(rng.choice([3.0, 4.0, 10.0]))
    radius = (np.float32(th) * SCALE_FACTORS[level]).astype(np.float32)
    return dict(keys=keys, desc=desc, u_right=u_right_for(keys, rng), q_desc=noisy_desc(rng, src_d, 8), u=u, v=v,
                ur=(u - rng.uniform(1, 30, n)).astype(np.float32), level=level, radius=radius,
                angle=src_k["angle"].astype(np.float32), src_k=src_k, src_d=src_d, rng=rng)
