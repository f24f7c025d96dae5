"""ctypes binding of the CPU oracle (oracle/liborb_oracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"
LIB_PATH = ORACLE_DIR / "liborb_oracle.so"
if os.environ.get("VSG_ORACLE_LIB"):  # tests/test_sanitizers.py: the ASan + UBSan build of the same sources
    LIB_PATH = Path(os.environ["VSG_ORACLE_LIB"])

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28

_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)
_u16p = C.POINTER(C.c_uint16)


def _ptr(a, t):
    return a.ctypes.data_as(t)


def build():
    srcs = [ORACLE_DIR / n for n in ("orb_oracle.cpp", "match_oracle.cpp", "bow_oracle.cpp", "routines_oracle.cpp", "orb_oracle.h",
                                     "brief_pattern_data.inc")]
    if LIB_PATH.exists() and all(LIB_PATH.stat().st_mtime >= s.stat().st_mtime for s in srcs):
        return
    subprocess.check_call(["make", "-C", str(ORACLE_DIR)] + (["asan"] if "asan" in LIB_PATH.name else []),
                          stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(LIB_PATH))
        L.or_create.restype = C.c_void_p
        L.or_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.or_destroy.argtypes = [C.c_void_p]
        L.or_set_blur_taps.argtypes = [C.c_void_p, _u16p]
        L.or_get_tables.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, _f32p, _i32p, _i32p]
        L.or_extract.argtypes = [C.c_void_p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, _u8p,
                                 C.c_int, _i32p]
        L.or_level_size.argtypes = [C.c_void_p, C.c_int, _i32p, _i32p]
        L.or_get_pyramid_level.argtypes = [C.c_void_p, C.c_int, _u8p, C.c_int, C.c_int]
        L.or_get_blurred_level.argtypes = [C.c_void_p, C.c_int, _u8p, C.c_int]
        L.or_get_candidates.argtypes = [C.c_void_p, C.c_int, _i32p, _i32p, _i32p, C.c_int]
        L.or_get_level_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.or_cv_round_f.argtypes = [C.c_float]
        L.or_cv_round_d.argtypes = [C.c_double]
        L.or_fast_atan2.restype = C.c_float
        L.or_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.or_resize_linear_u8.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int, C.c_int, C.c_int]
        L.or_copy_make_border101.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int, C.c_int]
        L.or_gaussian_blur7_u8.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int, _u16p]
        L.or_fast9_16.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i32p, _i32p, _i32p, C.c_int]
        L.or_distribute_octree.argtypes = [_i32p, _i32p, _i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_int, _i32p, C.c_int]
        L.or_ic_angle.restype = C.c_float
        L.or_ic_angle.argtypes = [_u8p, C.c_int, C.c_int, C.c_int]
        L.or_orb_descriptor.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_float, _u8p]
        L.or_descriptor_distance.argtypes = [_u8p, _u8p]
        L.or_three_maxima.argtypes = [_i32p, C.c_int, _i32p, _i32p, _i32p]
        L.or_grid_build.restype = C.c_void_p
        L.or_grid_build.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]
        L.or_grid_destroy.argtypes = [C.c_void_p]
        L.or_grid_query.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, _i32p, C.c_int]
        L.or_search_by_bow_kf_f.argtypes = [_u8p, _f32p, _u8p, C.c_int, _i32p, _i32p, _i32p, C.c_int,
                                            _u8p, _f32p, C.c_int, _i32p, _i32p, _i32p, C.c_int,
                                            C.c_float, C.c_int, _i32p]
        L.or_search_by_bow_kf_f_stereo.argtypes = [_u8p, _f32p, _u8p, C.c_int, _i32p, _i32p, _i32p, C.c_int,
                                                   _u8p, _f32p, C.c_int, C.c_int, _i32p, _i32p, _i32p, C.c_int,
                                                   C.c_float, C.c_int, _i32p]
        L.or_search_by_bow_kf_kf.argtypes = [_u8p, _f32p, _u8p, C.c_int, _i32p, _i32p, _i32p, C.c_int,
                                             _u8p, _f32p, _u8p, C.c_int, _i32p, _i32p, _i32p, C.c_int,
                                             C.c_float, C.c_int, _i32p]
        L.or_search_for_triangulation.argtypes = [_u8p, _f32p, _u8p, C.c_int, _i32p, _i32p, _i32p, C.c_int,
                                                  _u8p, _f32p, _u8p, C.c_int, _i32p, _i32p, _i32p, C.c_int,
                                                  C.c_void_p, C.c_void_p, C.c_int, _i32p]
        L.or_search_by_projection_last.argtypes = [_u8p, _f32p, _u8p, C.c_int, _i32p, _i32p, _u8p, _f32p, _u8p,
                                                   C.c_int, C.c_int, C.c_int, _i32p]
        L.or_search_by_projection_local.argtypes = [_u8p, _u8p, C.c_int, _i32p, _i32p, _u8p, _i32p, _u8p,
                                                    C.c_int, C.c_float, _i32p]
        L.or_search_for_initialization.argtypes = [_u8p, _f32p, _i32p, C.c_int, _i32p, _i32p, _u8p, _f32p, C.c_int,
                                                   C.c_float, C.c_int, _i32p]
        L.or_block_best2.argtypes = [_u8p, C.c_int, _u8p, C.c_int, _i32p, _i32p, _i32p]
        L.or_distinctive_descriptors.argtypes = [_u8p, _i32p, C.c_int, _i32p]
        L.or_distinctive_descriptors.restype = None
        L.or_search_window.argtypes = [_u8p, _u8p, C.c_int, _i32p, _i32p, _u8p, _u8p, C.c_int, C.c_int, _i32p, _i32p,
                                       _i32p]
        L.or_vocab_load.restype = C.c_void_p
        L.or_vocab_load.argtypes = [_u8p, C.c_size_t]
        L.or_vocab_destroy.argtypes = [C.c_void_p]
        L.or_vocab_info.argtypes = [C.c_void_p] + [_i32p] * 6
        _f64p = C.POINTER(C.c_double)
        L.or_vocab_transform.argtypes = [C.c_void_p, _u8p, C.c_int, C.c_int, _i32p, _f64p, C.c_int, _i32p, _i32p, _i32p,
                                         _i32p, C.c_int, _i32p, _i32p, _i32p, _f64p]
        L.or_cvt_gray_u8.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, C.c_int, _i32p, C.c_int]
        L.or_cvt_gray_u8.restype = None
        L.or_stereo_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, _u8p, C.c_int, C.c_void_p, _u8p, C.c_int,
                                        C.c_float, C.c_float, _f32p, _f32p]
        L.or_stereo_matches.restype = None
        L.or_bench_throughput.restype = C.c_double
        L.or_bench_throughput.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_long)]
        L.or_extract_batch_mt.restype = None
        L.or_extract_batch_mt.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_int, _i32p, C.c_void_p, _u8p]
        L.or_block_best2_batch_mt.restype = None
        L.or_block_best2_batch_mt.argtypes = [_u8p, C.c_size_t, _i32p, _u8p, C.c_size_t, _i32p, C.c_int, C.c_int, C.c_int,
                                              _i32p, _i32p, _i32p]
        _lib = L
    return _lib


def _u8c(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a


def _i32c(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f32c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class OracleExtractor:
    """Mirror of VS_GRAPHS::ORBextractor over the oracle (ORBextractor.h:51-93)."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.L = lib()
        self.h = self.L.or_create(nfeatures, scale_factor, nlevels, ini_th, min_th)
        assert self.h
        self.nfeatures, self.nlevels = nfeatures, nlevels

    def __del__(self):
        if getattr(self, "h", None):
            self.L.or_destroy(self.h)
            self.h = None

    def set_blur_taps(self, taps):
        t = np.ascontiguousarray(taps, dtype=np.uint16)
        assert t.shape == (7,)
        self.L.or_set_blur_taps(self.h, _ptr(t, _u16p))

    def tables(self):
        n = self.nlevels
        sc, inv, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        fpl = np.zeros(n, np.int32)
        umax = np.zeros(16, np.int32)
        self.L.or_get_tables(self.h, _ptr(sc, _f32p), _ptr(inv, _f32p), _ptr(s2, _f32p), _ptr(is2, _f32p),
                             _ptr(fpl, _i32p), _ptr(umax, _i32p))
        return dict(scale=sc, inv_scale=inv, sigma2=s2, inv_sigma2=is2, features_per_level=fpl, umax=umax)

    def __call__(self, image, lapping=(0, 0)):
        """Returns (monoIndex, keypoints[KP_DTYPE], descriptors[n,32])."""
        if image is None or image.size == 0:
            return -1, np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        img = _u8c(image)
        rows, cols = img.shape
        cap = self.nfeatures + 3 * self.nlevels + 64
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int32(0)
        mono = self.L.or_extract(self.h, _ptr(img, _u8p), rows, cols, img.strides[0], int(lapping[0]),
                                 int(lapping[1]), kps.ctypes.data_as(C.c_void_p), _ptr(desc, _u8p), cap,
                                 C.byref(n))
        assert mono != -2, "oracle capacity too small"
        return mono, kps[:n.value].copy(), desc[:n.value].copy()

    def level_size(self, level):
        w, h = C.c_int32(), C.c_int32()
        assert self.L.or_level_size(self.h, level, C.byref(w), C.byref(h)) == 0
        return w.value, h.value

    def pyramid_level(self, level, with_border=False):
        w, h = self.level_size(level)
        if with_border:
            w, h = w + 38, h + 38
        out = np.zeros((h, w), np.uint8)
        assert self.L.or_get_pyramid_level(self.h, level, _ptr(out, _u8p), w, int(with_border)) == 0
        return out

    def blurred_level(self, level):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        rc = self.L.or_get_blurred_level(self.h, level, _ptr(out, _u8p), w)
        return out if rc == 0 else None

    def candidates(self, level):
        cap = 1 << 18
        x, y, r = (np.zeros(cap, np.int32) for _ in range(3))
        n = self.L.or_get_candidates(self.h, level, _ptr(x, _i32p), _ptr(y, _i32p), _ptr(r, _i32p), cap)
        assert 0 <= n <= cap
        return x[:n].copy(), y[:n].copy(), r[:n].copy()

    def level_keypoints(self, level):
        cap = self.nfeatures + 64
        kps = np.zeros(cap, KP_DTYPE)
        n = self.L.or_get_level_keypoints(self.h, level, kps.ctypes.data_as(C.c_void_p), cap)
        assert 0 <= n <= cap
        return kps[:n].copy()


def cv_round_f(v):
    return lib().or_cv_round_f(float(np.float32(v)))


def fast_atan2(y, x):
    return np.float32(lib().or_fast_atan2(float(np.float32(y)), float(np.float32(x))))


def resize_linear(src, dw, dh):
    src = _u8c(src)
    dst = np.zeros((dh, dw), np.uint8)
    lib().or_resize_linear_u8(_ptr(src, _u8p), src.shape[1], src.shape[0], src.strides[0], _ptr(dst, _u8p), dw, dh, dw)
    return dst


def copy_make_border101(src, b):
    src = _u8c(src)
    h, w = src.shape
    dst = np.zeros((h + 2 * b, w + 2 * b), np.uint8)
    lib().or_copy_make_border101(_ptr(src, _u8p), w, h, src.strides[0], _ptr(dst, _u8p), w + 2 * b, b)
    return dst


DEFAULT_TAPS = (18, 34, 49, 55, 49, 34, 18)


def gaussian_blur7(src, taps=DEFAULT_TAPS):
    src = _u8c(src)
    h, w = src.shape
    dst = np.zeros((h, w), np.uint8)
    t = np.asarray(taps, np.uint16)
    lib().or_gaussian_blur7_u8(_ptr(src, _u8p), w, h, src.strides[0], _ptr(dst, _u8p), w, _ptr(t, _u16p))
    return dst


def fast9_16(img, threshold, nonmax=True):
    img = _u8c(img)
    h, w = img.shape
    cap = max(16, w * h)
    x, y, s = (np.zeros(cap, np.int32) for _ in range(3))
    n = lib().or_fast9_16(_ptr(img, _u8p), w, h, img.strides[0], threshold, int(nonmax), _ptr(x, _i32p),
                          _ptr(y, _i32p), _ptr(s, _i32p), cap)
    return x[:n].copy(), y[:n].copy(), s[:n].copy()


def distribute_octree(x, y, response, min_x, max_x, min_y, max_y, n_target):
    x, y, r = _i32c(x), _i32c(y), _i32c(response)
    cap = len(x) + 8
    out = np.zeros(cap, np.int32)
    n = lib().or_distribute_octree(_ptr(x, _i32p), _ptr(y, _i32p), _ptr(r, _i32p), len(x), min_x, max_x, min_y,
                                   max_y, n_target, _ptr(out, _i32p), cap)
    return out[:n].copy()


def ic_angle(img, cx, cy):
    img = _u8c(img)
    return np.float32(lib().or_ic_angle(_ptr(img, _u8p), img.strides[0], cx, cy))


def orb_descriptor(blurred, cx, cy, angle_deg):
    img = _u8c(blurred)
    d = np.zeros(32, np.uint8)
    lib().or_orb_descriptor(_ptr(img, _u8p), img.strides[0], cx, cy, float(np.float32(angle_deg)), _ptr(d, _u8p))
    return d


def descriptor_distance(a, b):
    a, b = _u8c(a), _u8c(b)
    return lib().or_descriptor_distance(_ptr(a, _u8p), _ptr(b, _u8p))


def three_maxima(sizes):
    s = _i32c(sizes)
    a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
    lib().or_three_maxima(_ptr(s, _i32p), len(s), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


class OracleGrid:
    def __init__(self, kps, min_x, min_y, max_x, max_y):
        self.kps = np.ascontiguousarray(kps, dtype=KP_DTYPE)
        self.L = lib()
        self.h = self.L.or_grid_build(self.kps.ctypes.data_as(C.c_void_p), len(self.kps), min_x, min_y, max_x, max_y)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.or_grid_destroy(self.h)
            self.h = None

    def query(self, x, y, r, min_level=-1, max_level=-1):
        cap = len(self.kps) + 1
        out = np.zeros(cap, np.int32)
        n = self.L.or_grid_query(self.h, float(np.float32(x)), float(np.float32(y)), float(np.float32(r)),
                                 min_level, max_level, _ptr(out, _i32p), cap)
        return out[:n].copy()


def search_by_bow_kf_f(kf_desc, kf_angle, kf_valid, kf_fv, f_desc, f_angle, f_fv, nnratio, check_ori, f_nleft=-1):
    """fv = (node_ids, offsets, indices) CSR. Returns (nmatches, matchF)."""
    kd, fd = _u8c(kf_desc), _u8c(f_desc)
    ka, fa = _f32c(kf_angle), _f32c(f_angle)
    kv = _u8c(kf_valid)
    kn, ko, ki = (_i32c(a) for a in kf_fv)
    fn, fo, fi = (_i32c(a) for a in f_fv)
    out = np.zeros(len(fd), np.int32)
    n = lib().or_search_by_bow_kf_f_stereo(_ptr(kd, _u8p), _ptr(ka, _f32p), _ptr(kv, _u8p), len(kd), _ptr(kn, _i32p),
                                           _ptr(ko, _i32p), _ptr(ki, _i32p), len(kn), _ptr(fd, _u8p),
                                           _ptr(fa, _f32p), len(fd), int(f_nleft), _ptr(fn, _i32p), _ptr(fo, _i32p),
                                           _ptr(fi, _i32p), len(fn), float(nnratio), int(check_ori), _ptr(out, _i32p))
    return n, out


def search_by_bow_kf_kf(d1, a1, v1, fv1, d2, a2, v2, fv2, nnratio, check_ori):
    d1, d2 = _u8c(d1), _u8c(d2)
    a1, a2 = _f32c(a1), _f32c(a2)
    v1, v2 = _u8c(v1), _u8c(v2)
    n1, o1, i1 = (_i32c(a) for a in fv1)
    n2, o2, i2 = (_i32c(a) for a in fv2)
    out = np.zeros(len(d1), np.int32)
    n = lib().or_search_by_bow_kf_kf(_ptr(d1, _u8p), _ptr(a1, _f32p), _ptr(v1, _u8p), len(d1), _ptr(n1, _i32p),
                                     _ptr(o1, _i32p), _ptr(i1, _i32p), len(n1), _ptr(d2, _u8p), _ptr(a2, _f32p),
                                     _ptr(v2, _u8p), len(d2), _ptr(n2, _i32p), _ptr(o2, _i32p), _ptr(i2, _i32p),
                                     len(n2), float(nnratio), int(check_ori), _ptr(out, _i32p))
    return n, out


def search_for_triangulation(d1, a1, e1, fv1, d2, a2, e2, fv2, pair_ok, pair_off, check_ori):
    """pair_ok: uint32 bit array (or None = every pair passes), pair_off: int32 per shared node + 1."""
    d1, d2 = _u8c(d1), _u8c(d2)
    a1, a2 = _f32c(a1), _f32c(a2)
    e1, e2 = _u8c(e1), _u8c(e2)
    n1, o1, i1 = (_i32c(a) for a in fv1)
    n2, o2, i2 = (_i32c(a) for a in fv2)
    out = np.zeros(max(len(d1), 1), np.int32)
    ok = np.ascontiguousarray(pair_ok, np.uint32) if pair_ok is not None else None
    po = _i32c(pair_off) if pair_ok is not None else None
    n = lib().or_search_for_triangulation(_ptr(d1, _u8p), _ptr(a1, _f32p), _ptr(e1, _u8p), len(d1), _ptr(n1, _i32p),
                                          _ptr(o1, _i32p), _ptr(i1, _i32p), len(n1), _ptr(d2, _u8p), _ptr(a2, _f32p),
                                          _ptr(e2, _u8p), len(d2), _ptr(n2, _i32p), _ptr(o2, _i32p), _ptr(i2, _i32p),
                                          len(n2), ok.ctypes.data if ok is not None else None,
                                          po.ctypes.data if po is not None else None, int(check_ori), _ptr(out, _i32p))
    return n, out[:len(d1)]


def search_by_projection_last(q_desc, q_angle, q_blocks, cand_off, cand_idx, t_desc, t_angle, t_blocked, th_high,
                              check_ori):
    qd, td = _u8c(q_desc), _u8c(t_desc)
    qa, ta = _f32c(q_angle), _f32c(t_angle)
    qb = _u8c(q_blocks)
    co, ci = _i32c(cand_off), _i32c(cand_idx)
    if len(ci) == 0:
        ci = np.zeros(1, np.int32)
    tb = _u8c(t_blocked).copy()
    tm = np.full(len(td), -1, np.int32)
    n = lib().or_search_by_projection_last(_ptr(qd, _u8p), _ptr(qa, _f32p), _ptr(qb, _u8p), len(qd), _ptr(co, _i32p),
                                           _ptr(ci, _i32p), _ptr(td, _u8p), _ptr(ta, _f32p), _ptr(tb, _u8p), len(td),
                                           int(th_high), int(check_ori), _ptr(tm, _i32p))
    return n, tm, tb


def search_by_projection_local(q_desc, q_blocks, cand_off, cand_idx, t_desc, t_octave, t_blocked, nnratio):
    qd, td = _u8c(q_desc), _u8c(t_desc)
    qb = _u8c(q_blocks)
    co, ci = _i32c(cand_off), _i32c(cand_idx)
    if len(ci) == 0:
        ci = np.zeros(1, np.int32)
    to = _i32c(t_octave)
    tb = _u8c(t_blocked).copy()
    tm = np.full(len(td), -1, np.int32)
    n = lib().or_search_by_projection_local(_ptr(qd, _u8p), _ptr(qb, _u8p), len(qd), _ptr(co, _i32p), _ptr(ci, _i32p),
                                            _ptr(td, _u8p), _ptr(to, _i32p), _ptr(tb, _u8p), len(td), float(nnratio),
                                            _ptr(tm, _i32p))
    return n, tm, tb


def search_for_initialization(d1, a1, oct1, cand_off, cand_idx, d2, a2, nnratio, check_ori):
    d1, d2 = _u8c(d1), _u8c(d2)
    a1, a2 = _f32c(a1), _f32c(a2)
    o1 = _i32c(oct1)
    co, ci = _i32c(cand_off), _i32c(cand_idx)
    if len(ci) == 0:
        ci = np.zeros(1, np.int32)
    out = np.zeros(len(d1), np.int32)
    n = lib().or_search_for_initialization(_ptr(d1, _u8p), _ptr(a1, _f32p), _ptr(o1, _i32p), len(d1), _ptr(co, _i32p),
                                           _ptr(ci, _i32p), _ptr(d2, _u8p), _ptr(a2, _f32p), len(d2), float(nnratio),
                                           int(check_ori), _ptr(out, _i32p))
    return n, out


def block_best2(a, b):
    a, b = _u8c(a), _u8c(b)
    best, second, arg = (np.zeros(len(a), np.int32) for _ in range(3))
    lib().or_block_best2(_ptr(a, _u8p), len(a), _ptr(b, _u8p), len(b), _ptr(best, _i32p), _ptr(second, _i32p),
                         _ptr(arg, _i32p))
    return best, second, arg


def bench_throughput(frames, nfeatures, nthreads, seconds, do_match=True, scale=1.2, nlevels=8, ini_th=20, min_th=7):
    """Native multi-threaded CPU baseline (frames/s, frames processed)."""
    f = np.ascontiguousarray(frames, dtype=np.uint8)
    n, rows, cols = f.shape
    done = C.c_long(0)
    fps = lib().or_bench_throughput(_ptr(f, _u8p), n, rows, cols, nfeatures, scale, nlevels, ini_th, min_th,
                                    int(nthreads), float(seconds), int(do_match), C.byref(done))
    return fps, done.value


def host_threads():
    """Host threads this process may really use: min(affinity, cgroup CPU quota)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def extract_batch(frames, nfeatures, capacity, scale=1.2, nlevels=8, ini_th=20, min_th=7, lapping=(0, 0), nthreads=None):
    """operator() on EVERY frame of a batch (frames: [B, rows, cols] uint8) on all host threads.  Returns
    (counts [B, 2] = n / monoIndex, kps [B, capacity] KP_DTYPE, desc [B, capacity, 32]); rows beyond n are zero."""
    f = np.ascontiguousarray(frames, dtype=np.uint8)
    B, rows, cols = f.shape
    counts = np.zeros((B, 2), np.int32)
    kps = np.zeros((B, capacity), KP_DTYPE)
    desc = np.zeros((B, capacity, 32), np.uint8)
    lib().or_extract_batch_mt(_ptr(f, _u8p), B, rows, cols, nfeatures, scale, nlevels, ini_th, min_th, lapping[0],
                              lapping[1], nthreads or host_threads(), capacity, _ptr(counts, _i32p),
                              kps.ctypes.data_as(C.c_void_p), _ptr(desc, _u8p))
    return counts, kps, desc


def block_best2_batch(a_desc, na, b_desc, nb, nthreads=None):
    """Row f: brute-force best / second / argbest of a_desc[f, :na[f]] against b_desc[f, :nb[f]] (both [B, capacity, 32]),
    on all host threads.  Returns three [B, capacity] int32 arrays (rows beyond na[f] are zero)."""
    a, b = _u8c(a_desc), _u8c(b_desc)
    B, cap = a.shape[0], a.shape[1]
    assert b.shape[0] == B
    na, nb = _i32c(na), _i32c(nb)
    best, second, arg = (np.zeros((B, cap), np.int32) for _ in range(3))
    lib().or_block_best2_batch_mt(_ptr(a, _u8p), a.strides[0], _ptr(na, _i32p), _ptr(b, _u8p), b.strides[0],
                                  _ptr(nb, _i32p), B, cap, nthreads or host_threads(), _ptr(best, _i32p),
                                  _ptr(second, _i32p), _ptr(arg, _i32p))
    return best, second, arg


def compare_batch(counts_gpu, kps_gpu, desc_gpu, counts_ref, kps_ref, desc_ref):
    """Frames of a batch whose device output differs from the oracle's in any bit: list of frame indices.
    counts: [B, 2]; kps: [B, cap] records or [B, cap, 28] bytes; desc: [B, cap, 32]."""
    B = len(counts_ref)
    kg = np.ascontiguousarray(kps_gpu).view(np.uint8).reshape(B, -1, 28)
    kr = np.ascontiguousarray(kps_ref).view(np.uint8).reshape(B, -1, 28)
    bad = []
    for f in range(B):
        n = int(counts_ref[f, 0])
        if (int(counts_gpu[f, 0]) != n or int(counts_gpu[f, 1]) != int(counts_ref[f, 1]) or n < 0
                or not np.array_equal(kg[f, :n], kr[f, :n]) or not np.array_equal(desc_gpu[f, :n], desc_ref[f, :n])):
            bad.append(f)
    return bad


def search_window(q_desc, q_blocks, cand_off, cand_idx, t_desc, t_blocked, th_high):
    """Returns (nmatches, q_best_idx, q_best_dist, train_match, train_blocked)."""
    qd, td = _u8c(q_desc), _u8c(t_desc)
    co, ci = _i32c(cand_off), _i32c(cand_idx)
    if len(ci) == 0:
        ci = np.zeros(1, np.int32)
    qb = _u8c(q_blocks) if q_blocks is not None else None
    tb = _u8c(t_blocked).copy() if t_blocked is not None else None
    qi, qdist = np.zeros(len(qd), np.int32), np.zeros(len(qd), np.int32)
    tm = np.full(len(td), -1, np.int32)
    n = lib().or_search_window(_ptr(qd, _u8p), _ptr(qb, _u8p) if qb is not None else None, len(qd), _ptr(co, _i32p),
                               _ptr(ci, _i32p), _ptr(td, _u8p), _ptr(tb, _u8p) if tb is not None else None, len(td),
                               int(th_high), _ptr(qi, _i32p), _ptr(qdist, _i32p), _ptr(tm, _i32p))
    return n, qi, qdist, tm, tb


def stereo_matches(ex_left, ex_right, kps_l, desc_l, kps_r, desc_r, mb, mbf):
    """Frame::ComputeStereoMatches on two OracleExtractors that have just run.  Returns (uRight, depth)."""
    kl, kr = np.ascontiguousarray(kps_l, KP_DTYPE), np.ascontiguousarray(kps_r, KP_DTYPE)
    dl, dr = _u8c(desc_l), _u8c(desc_r)
    ur, dep = np.zeros(max(len(kl), 1), np.float32), np.zeros(max(len(kl), 1), np.float32)
    lib().or_stereo_matches(ex_left.h, ex_right.h, kl.ctypes.data_as(C.c_void_p), _ptr(dl, _u8p), len(kl),
                            kr.ctypes.data_as(C.c_void_p), _ptr(dr, _u8p), len(kr), float(mb), float(mbf),
                            _ptr(ur, _f32p), _ptr(dep, _f32p))
    return ur[:len(kl)], dep[:len(kl)]


GRAY_COEFFS = (4899, 9617, 1868)
GRAY_SHIFT = 14


def cvt_gray(img, rgb_order=True, coeffs=GRAY_COEFFS, shift=GRAY_SHIFT):
    img = _u8c(img)
    rows, cols, ch = img.shape
    out = np.zeros((rows, cols), np.uint8)
    c = np.asarray(coeffs, np.int32)
    lib().or_cvt_gray_u8(_ptr(img, _u8p), rows, cols, img.strides[0], ch, int(rgb_order), _ptr(out, _u8p), cols,
                         _ptr(c, _i32p), int(shift))
    return out


class OracleVocabulary:
    """DBoW2 TemplatedVocabulary restatement (loadFromBinFile image + transform)."""

    def __init__(self, blob):
        b = np.frombuffer(blob, dtype=np.uint8).copy()
        self.L_ = lib()
        self.h = self.L_.or_vocab_load(_ptr(b, _u8p), len(b))
        assert self.h, "not a vocabulary image"
        v = [C.c_int32() for _ in range(6)]
        self.L_.or_vocab_info(self.h, *[C.byref(x) for x in v])
        self.k, self.L, self.scoring, self.weighting, self.nnodes, self.nwords = [x.value for x in v]

    def __del__(self):
        if getattr(self, "h", None):
            self.L_.or_vocab_destroy(self.h)
            self.h = None

    def transform(self, desc, levelsup):
        """Returns dict(bow_ids, bow_vals, fv=(node, off, idx), word, node, weight)."""
        d = _u8c(desc).reshape(-1, 32)
        n = len(d)
        cap = n + 1
        bi, bv = np.zeros(cap, np.int32), np.zeros(cap, np.float64)
        fn, fo, fi = np.zeros(cap, np.int32), np.zeros(cap + 1, np.int32), np.zeros(cap, np.int32)
        w_of, n_of, wt = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.float64)
        nb, nf = C.c_int32(), C.c_int32()
        f64p = C.POINTER(C.c_double)
        self.L_.or_vocab_transform(self.h, _ptr(d, _u8p), n, int(levelsup), _ptr(bi, _i32p), _ptr(bv, f64p), cap,
                                   C.byref(nb), _ptr(fn, _i32p), _ptr(fo, _i32p), _ptr(fi, _i32p), cap, C.byref(nf),
                                   _ptr(w_of, _i32p), _ptr(n_of, _i32p), _ptr(wt, f64p))
        return dict(bow_ids=bi[:nb.value].copy(), bow_vals=bv[:nb.value].copy(),
                    fv=(fn[:nf.value].copy(), fo[:nf.value + 1].copy(), fi[:fo[nf.value]].copy()),
                    word=w_of[:n].copy(), node=n_of[:n].copy(), weight=wt[:n].copy())


def distinctive_descriptors(desc, off):
    d, o = _u8c(desc).reshape(-1, 32), _i32c(off)
    if len(d) == 0:
        d = np.zeros((1, 32), np.uint8)
    best = np.zeros(len(o) - 1, np.int32)
    lib().or_distinctive_descriptors(_ptr(d, _u8p), _ptr(o, _i32p), len(o) - 1, _ptr(best, _i32p))
    return best


# ---------------------------------------------------------------- routine-level oracle (oracle/routines_oracle.cpp)
# ---- cameras of the reference's settings files whose mDistCoef(0) != 0 (Frame::UndistortKeyPoints / ComputeImageBounds
# do real work): K4 = (fx, fy, cx, cy), dist = mDistCoef, size = (cols, rows)
CAMERAS = {
    # config/RGB-D/TUM1.yaml:11-23 (BASELINE config C1)
    "tum1": dict(K4=(517.306408, 516.469215, 318.643040, 255.313989),
                 dist=(0.262383, -0.953104, -0.005358, 0.002628, 1.163314), size=(640, 480)),
    # config/RGB-D-Inertial/RealSense_D435i.yaml:11-23 (BASELINE config C5)
    "d435i": dict(K4=(6.165911254882812e+02, 6.166796264648438e+02, 3.242193603515625e+02, 2.3942701721191406e+02),
                  dist=(1.25323e-01, -2.51452e-01, 7.12e-04, 6.217e-03), size=(640, 480)),
}


def scaled_camera(name, cols, rows):
    """The camera `name` for another image size (focal lengths and principal point scale with the image, the distortion
    coefficients live in normalised coordinates and stay): tests extract on small frames."""
    c = CAMERAS[name]
    sx, sy = cols / c["size"][0], rows / c["size"][1]
    fx, fy, cx, cy = c["K4"]
    return dict(K4=(fx * sx, fy * sy, cx * sx, cy * sy), dist=c["dist"], size=(cols, rows))


def _cam_args(cam):
    K4 = np.asarray(cam["K4"], np.float32)
    dist = np.asarray(cam["dist"], np.float32)
    return K4, dist


def _bind_undistort(L):
    if getattr(L, "_undistort_bound", False):
        return L
    L.or_undistort_points.argtypes = [_f32p, C.c_int, _f32p, _f32p, C.c_int, _f32p]
    L.or_undistort_points.restype = None
    L.or_undistort_keypoints.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p, C.c_int, C.c_void_p]
    L.or_undistort_keypoints.restype = None
    L.or_image_bounds.argtypes = [C.c_int, C.c_int, _f32p, _f32p, C.c_int, _f32p]
    L.or_image_bounds.restype = None
    L._undistort_bound = True
    return L


def undistort_points(xy, cam):
    """cv::undistortPoints(xy, K, D, Mat(), K) restated (oracle/undistort_oracle.cpp); xy [n, 2] float32."""
    L = _bind_undistort(lib())
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    K4, dist = _cam_args(cam)
    out = np.zeros_like(xy)
    L.or_undistort_points(_ptr(xy, _f32p), len(xy), _ptr(K4, _f32p), _ptr(dist, _f32p), len(dist), _ptr(out, _f32p))
    return out


def undistort_keypoints(kps, cam):
    """Frame::UndistortKeyPoints (Frame.cc:891-921): mvKeysUn."""
    L = _bind_undistort(lib())
    k = np.ascontiguousarray(kps, dtype=KP_DTYPE)
    out = k.copy()
    K4, dist = _cam_args(cam)
    L.or_undistort_keypoints(k.ctypes.data_as(C.c_void_p), len(k), _ptr(K4, _f32p), _ptr(dist, _f32p), len(dist),
                             out.ctypes.data_as(C.c_void_p))
    return out


def image_bounds(cam):
    """Frame::ComputeImageBounds (Frame.cc:924-955): (mnMinX, mnMinY, mnMaxX, mnMaxY) as float32 values."""
    L = _bind_undistort(lib())
    K4, dist = _cam_args(cam)
    out = np.zeros(4, np.float32)
    L.or_image_bounds(int(cam["size"][0]), int(cam["size"][1]), _ptr(K4, _f32p), _ptr(dist, _f32p), len(dist),
                      _ptr(out, _f32p))
    return tuple(float(v) for v in out)


def _rl():
    L = lib()
    if getattr(L, "_routines_bound", False):
        return L
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    L.or_frame_create.restype = vp
    L.or_frame_create.argtypes = [vp, _u8p, _f32p, ci, ci, cf, cf, cf, cf]
    L.or_frame_destroy.argtypes = [vp]
    L.or_frame_destroy.restype = None
    L.or_frame_grid.argtypes = [vp, ci, _i32p, _i32p]
    L.or_frame_features_in_area.argtypes = [vp, cf, cf, cf, ci, ci, ci, ci, _i32p, ci]
    L.or_frame_search_by_projection.argtypes = [vp, ci, _u8p, _u8p, _u8p, _f32p, _f32p, _f32p, _i32p, _f32p, _u8p, _f32p,
                                                _f32p, _i32p, _f32p, cf, cf, _f32p, _i32p, _i32p, _u8p, _i32p]
    L.or_frame_search_by_projection_last.argtypes = [vp, ci, _u8p, _u8p, _f32p, _f32p, _f32p, _f32p, _f32p, _i32p,
                                                     _f32p, cf, ci, ci, _f32p, ci, _u8p, _i32p]
    L.or_kf_search_by_projection_sim3.argtypes = [vp, ci, _u8p, _f32p, _f32p, _f32p, _i32p, cf, _i32p]
    L.or_frame_search_by_projection_kf.argtypes = [vp, ci, _u8p, _f32p, _f32p, _f32p, _i32p, _f32p, ci, ci, _u8p, _i32p]
    L.or_kf_search_by_sim3.argtypes = [vp, vp, ci, _i32p, _u8p, _f32p, _f32p, _f32p, _i32p, ci, _i32p, _u8p, _f32p,
                                       _f32p, _f32p, _i32p, _i32p]
    L.or_kf_fuse.argtypes = [vp, ci, _i32p, _u8p, _f32p, _f32p, _f32p, _f32p, _i32p, ci, _f32p, _i32p, _i32p, _u8p,
                             _i32p, _i32p, _i32p, _i32p]
    L.or_kf_fuse_sim3.argtypes = [vp, ci, _i32p, _u8p, _f32p, _f32p, _f32p, _i32p, _i32p, _i32p, _u8p, _i32p, _i32p,
                                  _i32p, _i32p]
    L.or_frame_search_for_initialization.argtypes = [vp, vp, _f32p, _f32p, ci, cf, ci, _i32p]
    L._routines_bound = True
    return L


def _o(a, t, conv):
    if a is None:
        return None, None
    x = conv(a)
    return x, _ptr(x, t)


class OracleFrame:
    """Flattened Frame / KeyFrame of the routine-level oracle: keys = mvKeysUn (or mvKeys || mvKeysRight), desc,
    mvuRight, Nleft, grid bounds."""

    def __init__(self, kps, desc, bounds, u_right=None, nleft=-1):
        self._L = _rl()
        self.kps = np.ascontiguousarray(kps, dtype=KP_DTYPE)
        self.desc = _u8c(desc).reshape(-1, 32)
        self.N, self.nleft = len(self.kps), int(nleft)
        ur = _f32c(u_right) if u_right is not None else None
        d = self.desc if len(self.desc) else np.zeros((1, 32), np.uint8)
        self._h = self._L.or_frame_create(self.kps.ctypes.data_as(C.c_void_p), _ptr(d, _u8p),
                                          _ptr(ur, _f32p) if ur is not None else None, self.N, self.nleft,
                                          *[float(b) for b in bounds])

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.or_frame_destroy(self._h)
            self._h = None

    def grid(self, right=False):
        cs, en = np.zeros(64 * 48 + 1, np.int32), np.zeros(max(self.N, 1), np.int32)
        ne = self._L.or_frame_grid(self._h, int(right), _ptr(cs, _i32p), _ptr(en, _i32p))
        return cs, en[:ne]

    def features_in_area(self, x, y, r, min_level=-1, max_level=-1, right=False, kf_form=False):
        out = np.zeros(max(self.N, 1), np.int32)
        n = self._L.or_frame_features_in_area(self._h, float(x), float(y), float(r), int(min_level), int(max_level),
                                              int(right), int(kf_form), _ptr(out, _i32p), len(out))
        return out[:n].copy()

    def search_by_projection(self, mp, th, nnratio, scale_factors, train_blocked, left_to_right=None,
                             right_to_left=None):
        d = _u8c(mp["desc"]).reshape(-1, 32)
        n = len(d)
        obs, inv = _u8c(mp["observed"]), _u8c(mp["in_view"])
        px, py, lvl, vc = _f32c(mp["proj_x"]), _f32c(mp["proj_y"]), _i32c(mp["scale_level"]), _f32c(mp["view_cos"])
        pxr = _f32c(mp["proj_xr"]) if mp.get("proj_xr") is not None else np.zeros(max(n, 1), np.float32)
        keep = []

        def opt(key, conv, t):
            if mp.get(key) is None:
                return None
            a = conv(mp[key])
            keep.append(a)
            return _ptr(a, t)
        sf = _f32c(scale_factors)
        ltr = _i32c(left_to_right) if left_to_right is not None else None
        rtl = _i32c(right_to_left) if right_to_left is not None else None
        tb = _u8c(train_blocked).copy()
        tm = np.full(max(len(tb), 1), -1, np.int32)
        nm = self._L.or_frame_search_by_projection(
            self._h, n, _ptr(d, _u8p), _ptr(obs, _u8p), _ptr(inv, _u8p), _ptr(px, _f32p), _ptr(py, _f32p),
            _ptr(pxr, _f32p), _ptr(lvl, _i32p), _ptr(vc, _f32p), opt("in_view_r", _u8c, _u8p),
            opt("proj_x_r", _f32c, _f32p), opt("proj_y_r", _f32c, _f32p), opt("scale_level_r", _i32c, _i32p),
            opt("view_cos_r", _f32c, _f32p), float(th), float(np.float32(nnratio)), _ptr(sf, _f32p),
            _ptr(ltr, _i32p) if ltr is not None else None, _ptr(rtl, _i32p) if rtl is not None else None,
            _ptr(tb, _u8p), _ptr(tm, _i32p))
        return nm, tm[:len(tb)], tb

    def search_by_projection_last(self, desc, observed, u, v, ur, last_octave, last_angle, th, direction,
                                  scale_factors, check_ori, train_blocked, u_r=None, v_r=None):
        d = _u8c(desc).reshape(-1, 32)
        obs = _u8c(observed)
        uu, vv, oc, an = _f32c(u), _f32c(v), _i32c(last_octave), _f32c(last_angle)
        ur_ = _f32c(ur) if ur is not None else np.zeros(max(len(d), 1), np.float32)
        ur2 = _f32c(u_r) if u_r is not None else None
        vr2 = _f32c(v_r) if v_r is not None else None
        sf = _f32c(scale_factors)
        tb = _u8c(train_blocked).copy()
        tm = np.full(max(len(tb), 1), -1, np.int32)
        nm = self._L.or_frame_search_by_projection_last(
            self._h, len(d), _ptr(d, _u8p), _ptr(obs, _u8p), _ptr(uu, _f32p), _ptr(vv, _f32p), _ptr(ur_, _f32p),
            _ptr(ur2, _f32p) if ur2 is not None else None, _ptr(vr2, _f32p) if vr2 is not None else None,
            _ptr(oc, _i32p), _ptr(an, _f32p), float(th), int(direction == 1), int(direction == 2), _ptr(sf, _f32p),
            int(check_ori), _ptr(tb, _u8p), _ptr(tm, _i32p))
        return nm, tm[:len(tb)], tb

    def search_by_projection_sim3(self, desc, u, v, radius, level, ratio_hamming, matched):
        d = _u8c(desc).reshape(-1, 32)
        m = _i32c(matched).copy()
        nm = self._L.or_kf_search_by_projection_sim3(self._h, len(d), _ptr(d, _u8p), _ptr(_f32c(u), _f32p),
                                                     _ptr(_f32c(v), _f32p), _ptr(_f32c(radius), _f32p),
                                                     _ptr(_i32c(level), _i32p), float(np.float32(ratio_hamming)),
                                                     _ptr(m, _i32p))
        return nm, m[:len(matched)]

    def search_by_projection_kf(self, desc, u, v, radius, level, kf_angle, orb_dist, check_ori, occupied):
        d = _u8c(desc).reshape(-1, 32)
        oc = _u8c(occupied).copy()
        tm = np.full(max(len(oc), 1), -1, np.int32)
        nm = self._L.or_frame_search_by_projection_kf(self._h, len(d), _ptr(d, _u8p), _ptr(_f32c(u), _f32p),
                                                      _ptr(_f32c(v), _f32p), _ptr(_f32c(radius), _f32p),
                                                      _ptr(_i32c(level), _i32p), _ptr(_f32c(kf_angle), _f32p),
                                                      int(orb_dist), int(check_ori), _ptr(oc, _u8p), _ptr(tm, _i32p))
        return nm, tm[:len(occupied)], oc

    def fuse(self, query_mp, desc, u, v, ur, radius, level, inv_sigma2, slot_mp, mp_obs, mp_bad, right=False):
        d = _u8c(desc).reshape(-1, 32)
        n = len(d)
        sm, ob, bad = _i32c(slot_mp).copy(), _i32c(mp_obs).copy(), _u8c(mp_bad).copy()
        bi, bd, act, oth = (np.zeros(max(n, 1), np.int32) for _ in range(4))
        nf = self._L.or_kf_fuse(self._h, n, _ptr(_i32c(query_mp), _i32p), _ptr(d, _u8p), _ptr(_f32c(u), _f32p),
                                _ptr(_f32c(v), _f32p), _ptr(_f32c(ur), _f32p), _ptr(_f32c(radius), _f32p),
                                _ptr(_i32c(level), _i32p), int(right), _ptr(_f32c(inv_sigma2), _f32p), _ptr(sm, _i32p),
                                _ptr(ob, _i32p), _ptr(bad, _u8p), _ptr(bi, _i32p), _ptr(bd, _i32p), _ptr(act, _i32p),
                                _ptr(oth, _i32p))
        return nf, bi[:n], bd[:n], act[:n], oth[:n], sm, ob, bad

    def fuse_sim3(self, query_mp, desc, u, v, radius, level, slot_mp, mp_obs, mp_bad):
        d = _u8c(desc).reshape(-1, 32)
        n = len(d)
        sm, ob, bad = _i32c(slot_mp).copy(), _i32c(mp_obs).copy(), _u8c(mp_bad).copy()
        bi, bd, act, oth = (np.zeros(max(n, 1), np.int32) for _ in range(4))
        nf = self._L.or_kf_fuse_sim3(self._h, n, _ptr(_i32c(query_mp), _i32p), _ptr(d, _u8p), _ptr(_f32c(u), _f32p),
                                     _ptr(_f32c(v), _f32p), _ptr(_f32c(radius), _f32p), _ptr(_i32c(level), _i32p),
                                     _ptr(sm, _i32p), _ptr(ob, _i32p), _ptr(bad, _u8p), _ptr(bi, _i32p), _ptr(bd, _i32p),
                                     _ptr(act, _i32p), _ptr(oth, _i32p))
        return nf, bi[:n], bd[:n], act[:n], oth[:n], sm, ob, bad

    def search_for_initialization(self, f2, prev_x, prev_y, window_size, nnratio, check_ori):
        out = np.full(max(self.N, 1), -1, np.int32)
        nm = self._L.or_frame_search_for_initialization(self._h, f2._h, _ptr(_f32c(prev_x), _f32p),
                                                        _ptr(_f32c(prev_y), _f32p), int(window_size),
                                                        float(np.float32(nnratio)), int(check_ori), _ptr(out, _i32p))
        return nm, out[:self.N]


def search_by_sim3(kf1, kf2, q1, q2):
    L = _rl()

    def unpack(q):
        d = _u8c(q["desc"]).reshape(-1, 32)
        return (len(d), _i32c(q["idx"]), d if len(d) else np.zeros((1, 32), np.uint8), _f32c(q["u"]), _f32c(q["v"]),
                _f32c(q["radius"]), _i32c(q["level"]))
    a, b = unpack(q1), unpack(q2)
    out = np.full(max(kf1.N, 1), -1, np.int32)
    nf = L.or_kf_search_by_sim3(kf1._h, kf2._h, a[0], _ptr(a[1], _i32p), _ptr(a[2], _u8p), _ptr(a[3], _f32p),
                                _ptr(a[4], _f32p), _ptr(a[5], _f32p), _ptr(a[6], _i32p), b[0], _ptr(b[1], _i32p),
                                _ptr(b[2], _u8p), _ptr(b[3], _f32p), _ptr(b[4], _f32p), _ptr(b[5], _f32p),
                                _ptr(b[6], _i32p), _ptr(out, _i32p))
    return nf, out[:kf1.N]
