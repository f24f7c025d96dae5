"""Host-side mirror of the reference's front-end classes over the HIP C ABI (include/vsg_orb.h).

`ORBextractor` mirrors VS_GRAPHS::ORBextractor (orb_slam3/include/ORBextractor.h:42-119):
same constructor arguments, `__call__` = operator() returning (monoIndex, keypoints, descriptors),
the scale-table getters and `image_pyramid` (mvImagePyramid).  `ORBmatcher` mirrors the Hamming
searches of VS_GRAPHS::ORBmatcher (orb_slam3/include/ORBmatcher.h:34-99) on flattened arrays.

There is no CPU fallback: importing works anywhere, but constructing an extractor or calling a
matcher without libvsg_orb.so and a HIP device raises.
"""
import ctypes as C
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libvsg_orb.so"

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28

VSG_OK = 0
ERRORS = {-1: "VSG_ERR_EMPTY_IMAGE", -2: "VSG_ERR_CAPACITY", -3: "VSG_ERR_UNSUPPORTED", -4: "VSG_ERR_NO_DEVICE",
          -5: "VSG_ERR_HIP", -6: "VSG_ERR_INVALID"}

_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)
_f32p = C.POINTER(C.c_float)
_u16p = C.POINTER(C.c_uint16)

EXPORTS = [
    "vsg_last_error", "vsg_device_count", "vsg_orb_create", "vsg_orb_destroy", "vsg_orb_get_tables",
    "vsg_orb_set_blur_taps", "vsg_orb_capacity", "vsg_orb_extract", "vsg_orb_extract_batch",
    "vsg_orb_extract_batch_device", "vsg_orb_level_size", "vsg_orb_copy_pyramid_level",
    "vsg_orb_copy_blurred_level", "vsg_orb_copy_candidates", "vsg_orb_copy_selected", "vsg_orb_enable_timing",
    "vsg_orb_get_timing", "vsg_orb_set_serialize", "vsg_hamming_pairs", "vsg_hamming_block_best2", "vsg_hamming_block_best2_device",
    "vsg_search_by_bow_kf_f", "vsg_search_by_bow_kf_kf", "vsg_search_by_projection_last",
    "vsg_search_by_projection_local", "vsg_search_for_initialization", "vsg_search_window", "vsg_grid_build",
    "vsg_grid_destroy", "vsg_grid_query", "vsg_stereo_matches", "vsg_orb_set_gray_coeffs",
    "vsg_orb_extract_batch_device_color", "vsg_orb_extract_batch_color", "vsg_vocab_load", "vsg_vocab_destroy",
    "vsg_vocab_info", "vsg_bow_transform", "vsg_distinctive_descriptors", "vsg_debug_device_sort",
    "vsg_search_for_triangulation", "vsg_search_by_bow_kf_f_stereo",
]


class VsgError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        super().__init__(f"{where}: {ERRORS.get(code, code)} {detail}".strip())


_lib = None


def debug_device_sort(items, device=0):
    """Test hook: the device's std::sort replay (see include/vsg_orb.h) on uint64 items (key = upper 32 bits)."""
    L = load_library()
    a = np.ascontiguousarray(items, dtype=np.uint64).copy()
    L.vsg_debug_device_sort.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.c_int]
    _check(L.vsg_debug_device_sort(device, a.ctypes.data_as(C.POINTER(C.c_uint64)), len(a)), "vsg_debug_device_sort")
    return a


def load_library():
    """dlopen libvsg_orb.so (built in-tree by visual_sgraphs_amd.build).  Fails loudly if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -m visual_sgraphs_amd.build` "
                           "(the ORB front-end has no CPU fallback)")
    L = C.CDLL(str(LIB_PATH))
    L.vsg_last_error.restype = C.c_char_p
    L.vsg_orb_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.vsg_orb_destroy.argtypes = [C.c_void_p]
    L.vsg_orb_destroy.restype = None
    L.vsg_orb_get_tables.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, _f32p, _i32p, _i32p]
    L.vsg_orb_set_blur_taps.argtypes = [C.c_void_p, _u16p]
    L.vsg_orb_capacity.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.vsg_orb_extract.argtypes = [C.c_void_p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, _u8p,
                                  C.c_int, _i32p]
    L.vsg_orb_extract_batch.argtypes = [C.c_void_p, _u8p, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_void_p, _u8p, C.c_int, _i32p, _i32p]
    L.vsg_orb_extract_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                               C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                               C.c_void_p]
    L.vsg_orb_level_size.argtypes = [C.c_void_p, C.c_int, _i32p, _i32p]
    L.vsg_orb_copy_pyramid_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int]
    L.vsg_orb_copy_blurred_level.argtypes = [C.c_void_p, C.c_int, C.c_int, _u8p, C.c_int]
    L.vsg_orb_copy_candidates.argtypes = [C.c_void_p, C.c_int, C.c_int, _u32p, C.c_int]
    L.vsg_orb_copy_selected.argtypes = [C.c_void_p, C.c_int, C.c_int, _u32p, C.c_int]
    L.vsg_orb_enable_timing.argtypes = [C.c_void_p, C.c_int]
    L.vsg_orb_get_timing.argtypes = [C.c_void_p, _f32p, C.c_int]
    L.vsg_orb_set_serialize.argtypes = [C.c_void_p, C.c_int]
    L.vsg_hamming_pairs.argtypes = [C.c_int, _u8p, C.c_int, _u8p, C.c_int, _i32p, _i32p, C.c_int, _i32p]
    L.vsg_hamming_block_best2.argtypes = [C.c_int, _u8p, C.c_int, _u8p, C.c_int, _i32p, _i32p, _i32p]
    L.vsg_hamming_block_best2_device.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                                 C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_void_p]
    L.vsg_search_by_bow_kf_f.argtypes = [C.c_int, _u8p, _f32p, _u8p, C.c_int, _i32p, _i32p, _i32p, C.c_int,
                                         _u8p, _f32p, C.c_int, _i32p, _i32p, _i32p, C.c_int, C.c_float, C.c_int, _i32p]
    L.vsg_search_by_bow_kf_f_stereo.argtypes = [C.c_int, _u8p, _f32p, _u8p, C.c_int, _i32p, _i32p, _i32p, C.c_int,
                                                _u8p, _f32p, C.c_int, C.c_int, _i32p, _i32p, _i32p, C.c_int, C.c_float,
                                                C.c_int, _i32p]
    L.vsg_search_by_bow_kf_kf.argtypes = [C.c_int, _u8p, _f32p, _u8p, C.c_int, _i32p, _i32p, _i32p, C.c_int,
                                          _u8p, _f32p, _u8p, C.c_int, _i32p, _i32p, _i32p, C.c_int, C.c_float,
                                          C.c_int, _i32p]
    L.vsg_search_for_triangulation.argtypes = [C.c_int, _u8p, _f32p, _u8p, C.c_int, _i32p, _i32p, _i32p, C.c_int,
                                               _u8p, _f32p, _u8p, C.c_int, _i32p, _i32p, _i32p, C.c_int, C.c_void_p,
                                               C.c_void_p, C.c_int, _i32p]
    L.vsg_search_by_projection_last.argtypes = [C.c_int, _u8p, _f32p, _u8p, C.c_int, _i32p, _i32p, _u8p, _f32p, _u8p,
                                                C.c_int, C.c_int, C.c_int, _i32p]
    L.vsg_search_by_projection_local.argtypes = [C.c_int, _u8p, _u8p, C.c_int, _i32p, _i32p, _u8p, _i32p, _u8p,
                                                 C.c_int, C.c_float, _i32p]
    L.vsg_search_for_initialization.argtypes = [C.c_int, _u8p, _f32p, _i32p, C.c_int, _i32p, _i32p, _u8p, _f32p,
                                                C.c_int, C.c_float, C.c_int, _i32p]
    L.vsg_search_window.argtypes = [C.c_int, _u8p, _u8p, C.c_int, _i32p, _i32p, _u8p, _u8p, C.c_int, C.c_int, _i32p,
                                    _i32p, _i32p]
    L.vsg_stereo_matches.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, _u8p, C.c_int, C.c_void_p,
                                     _u8p, C.c_int, C.c_float, C.c_float, _f32p, _f32p]
    L.vsg_orb_set_gray_coeffs.argtypes = [C.c_void_p, _i32p, C.c_int]
    L.vsg_orb_extract_batch_device_color.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t,
                                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                     C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.vsg_orb_extract_batch_color.argtypes = [C.c_void_p, _u8p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int,
                                              C.c_int, C.c_int, C.c_int, C.c_void_p, _u8p, C.c_int, _i32p, _i32p]
    _f64p = C.POINTER(C.c_double)
    L.vsg_vocab_load.argtypes = [C.c_int, _u8p, C.c_size_t, C.POINTER(C.c_void_p)]
    L.vsg_vocab_destroy.argtypes = [C.c_void_p]
    L.vsg_vocab_destroy.restype = None
    L.vsg_vocab_info.argtypes = [C.c_void_p] + [_i32p] * 6
    L.vsg_bow_transform.argtypes = [C.c_void_p, _u8p, C.c_int, C.c_int, _i32p, _f64p, C.c_int, _i32p, _i32p, _i32p,
                                    _i32p, C.c_int, _i32p, _i32p, _i32p, _f64p]
    L.vsg_distinctive_descriptors.argtypes = [C.c_int, _u8p, _i32p, C.c_int, _i32p]
    L.vsg_grid_build.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                 C.POINTER(C.c_void_p)]
    L.vsg_grid_destroy.argtypes = [C.c_void_p]
    L.vsg_grid_destroy.restype = None
    L.vsg_grid_query.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, _i32p, _i32p, C.c_int, _i32p, _i32p, C.c_int]
    _lib = L
    return L


def _check(rc, where):
    if rc < 0:
        raise VsgError(rc, where, load_library().vsg_last_error().decode(errors="replace"))
    return rc


def device_count():
    return load_library().vsg_device_count()


def _p(a, t):
    return a.ctypes.data_as(t)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a if a.size else np.zeros(1, np.int32)


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a if a.size else np.zeros(1, np.float32)


class ORBextractor:
    """VS_GRAPHS::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) on one MI355X."""

    STAGES = ("pyramid", "fast", "octree", "blur", "slots", "orient_desc", "total")

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device=0, max_batch=1):
        self._L = load_library()
        self._h = C.c_void_p()
        _check(self._L.vsg_orb_create(int(nfeatures), float(scaleFactor), int(nlevels), int(iniThFAST), int(minThFAST),
                                      int(device), int(max_batch), C.byref(self._h)), "vsg_orb_create")
        self.nfeatures, self.nlevels, self.scaleFactor = int(nfeatures), int(nlevels), float(np.float32(scaleFactor))
        self.device, self.max_batch = int(device), int(max_batch)
        self._shape = None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.vsg_orb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    # ---- getters (ORBextractor.h:63-91)
    def _tables(self):
        n = self.nlevels
        out = [np.zeros(n, np.float32) for _ in range(4)] + [np.zeros(n, np.int32), np.zeros(16, np.int32)]
        self._L.vsg_orb_get_tables(self._h, _p(out[0], _f32p), _p(out[1], _f32p), _p(out[2], _f32p), _p(out[3], _f32p),
                                   _p(out[4], _i32p), _p(out[5], _i32p))
        return out

    def GetLevels(self):
        return self.nlevels

    def GetScaleFactor(self):
        return self.scaleFactor

    def GetScaleFactors(self):
        return self._tables()[0]

    def GetInverseScaleFactors(self):
        return self._tables()[1]

    def GetScaleSigmaSquares(self):
        return self._tables()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[3]

    def features_per_level(self):
        return self._tables()[4]

    def umax(self):
        return self._tables()[5]

    def set_blur_taps(self, taps):
        t = np.ascontiguousarray(taps, dtype=np.uint16)
        assert t.shape == (7,)
        _check(self._L.vsg_orb_set_blur_taps(self._h, _p(t, _u16p)), "vsg_orb_set_blur_taps")

    def capacity(self, rows, cols):
        return _check(self._L.vsg_orb_capacity(self._h, int(rows), int(cols)), "vsg_orb_capacity")

    # ---- operator() (ORBextractor.h:59-61).  mask is ignored, like the reference (ORBextractor.h:58).
    def __call__(self, image, mask=None, vLappingArea=(0, 0)):
        """Returns (monoIndex, keypoints[KP_DTYPE], descriptors[n,32] uint8); monoIndex == -1 for an empty image."""
        if image is None or image.size == 0:
            return -1, np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        img = np.asarray(image)
        assert img.dtype == np.uint8 and img.ndim == 2, "CV_8UC1 expected (ORBextractor.cc:1091)"
        if img.strides[1] != 1:
            img = np.ascontiguousarray(img)
        rows, cols = img.shape
        cap = self.capacity(rows, cols)
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int32(0)
        mono = self._L.vsg_orb_extract(self._h, _p(img, _u8p), rows, cols, img.strides[0], int(vLappingArea[0]),
                                       int(vLappingArea[1]), kps.ctypes.data_as(C.c_void_p), _p(desc, _u8p), cap,
                                       C.byref(n))
        _check(mono, "vsg_orb_extract")
        self._shape = (rows, cols)
        return mono, kps[:n.value].copy(), desc[:n.value].copy()

    def extract_batch(self, images, vLappingArea=(0, 0)):
        """images: [B,H,W] uint8.  Returns list of (monoIndex, keypoints, descriptors)."""
        imgs = np.ascontiguousarray(images, dtype=np.uint8)
        assert imgs.ndim == 3 and imgs.shape[0] <= self.max_batch
        B, rows, cols = imgs.shape
        cap = self.capacity(rows, cols)
        kps = np.zeros((B, cap), KP_DTYPE)
        desc = np.zeros((B, cap, 32), np.uint8)
        n = np.zeros(B, np.int32)
        mono = np.zeros(B, np.int32)
        _check(self._L.vsg_orb_extract_batch(self._h, _p(imgs, _u8p), B, imgs.strides[0], rows, cols, imgs.strides[1],
                                             int(vLappingArea[0]), int(vLappingArea[1]),
                                             kps.ctypes.data_as(C.c_void_p), _p(desc, _u8p), cap, _p(n, _i32p),
                                             _p(mono, _i32p)), "vsg_orb_extract_batch")
        self._shape = (rows, cols)
        return [(int(mono[i]), kps[i, :n[i]].copy(), desc[i, :n[i]].copy()) for i in range(B)]

    def extract_batch_device(self, d_gray, nframes, frame_stride, rows, cols, stride, d_kps, d_desc, d_counts, capacity,
                             vLappingArea=(0, 0), stream=None):
        """Raw device-pointer entry (ints / torch .data_ptr()); asynchronous on `stream`."""
        _check(self._L.vsg_orb_extract_batch_device(self._h, C.c_void_p(d_gray), int(nframes), int(frame_stride),
                                                    int(rows), int(cols), int(stride), int(vLappingArea[0]),
                                                    int(vLappingArea[1]), C.c_void_p(d_kps), C.c_void_p(d_desc),
                                                    C.c_void_p(d_counts), int(capacity),
                                                    C.c_void_p(stream) if stream else None),
               "vsg_orb_extract_batch_device")
        self._shape = (rows, cols)

    def set_gray_coeffs(self, coeffs, shift):
        c = np.ascontiguousarray(coeffs, dtype=np.int32)
        _check(self._L.vsg_orb_set_gray_coeffs(self._h, _p(c, _i32p), int(shift)), "vsg_orb_set_gray_coeffs")

    def extract_batch_color(self, images, rgb_order=True, vLappingArea=(0, 0)):
        """images: [B,H,W,3|4] uint8 colour frames (Tracking::mbRGB = rgb_order).  Returns [(mono, kps, desc)]."""
        imgs = np.ascontiguousarray(images, dtype=np.uint8)
        assert imgs.ndim == 4 and imgs.shape[3] in (3, 4) and imgs.shape[0] <= self.max_batch
        B, rows, cols, ch = imgs.shape
        cap = self.capacity(rows, cols)
        kps = np.zeros((B, cap), KP_DTYPE)
        desc = np.zeros((B, cap, 32), np.uint8)
        n, mono = np.zeros(B, np.int32), np.zeros(B, np.int32)
        _check(self._L.vsg_orb_extract_batch_color(self._h, _p(imgs, _u8p), ch, int(rgb_order), B, imgs.strides[0],
                                                   rows, cols, imgs.strides[1], int(vLappingArea[0]),
                                                   int(vLappingArea[1]), kps.ctypes.data_as(C.c_void_p),
                                                   _p(desc, _u8p), cap, _p(n, _i32p), _p(mono, _i32p)),
               "vsg_orb_extract_batch_color")
        self._shape = (rows, cols)
        return [(int(mono[i]), kps[i, :n[i]].copy(), desc[i, :n[i]].copy()) for i in range(B)]

    def extract_batch_device_color(self, d_img, channels, rgb_order, nframes, frame_stride, rows, cols, stride, d_kps,
                                   d_desc, d_counts, capacity, vLappingArea=(0, 0), stream=None):
        """Interleaved 8-bit RGB(A)/BGR(A) frames on the device: cvtColor fused into the level-0 staging."""
        _check(self._L.vsg_orb_extract_batch_device_color(
            self._h, C.c_void_p(d_img), int(channels), int(rgb_order), int(nframes), int(frame_stride), int(rows),
            int(cols), int(stride), int(vLappingArea[0]), int(vLappingArea[1]), C.c_void_p(d_kps), C.c_void_p(d_desc),
            C.c_void_p(d_counts), int(capacity), C.c_void_p(stream) if stream else None),
            "vsg_orb_extract_batch_device_color")
        self._shape = (rows, cols)

    # ---- mvImagePyramid and stage read-back
    def level_size(self, level):
        w, h = C.c_int32(), C.c_int32()
        _check(self._L.vsg_orb_level_size(self._h, level, C.byref(w), C.byref(h)), "vsg_orb_level_size")
        return w.value, h.value

    def image_pyramid(self, level, frame=0, with_border=False):
        w, h = self.level_size(level)
        if with_border:
            w, h = w + 38, h + 38
        out = np.zeros((h, w), np.uint8)
        _check(self._L.vsg_orb_copy_pyramid_level(self._h, frame, level, int(with_border), _p(out, _u8p), w),
               "vsg_orb_copy_pyramid_level")
        return out

    def blurred_level(self, level, frame=0):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        _check(self._L.vsg_orb_copy_blurred_level(self._h, frame, level, _p(out, _u8p), w), "vsg_orb_copy_blurred_level")
        return out

    @staticmethod
    def _unpack(p):
        return (p & 0xFFF).astype(np.int32), ((p >> 12) & 0xFFF).astype(np.int32), (p >> 24).astype(np.int32)

    def candidates(self, level, frame=0):
        cap = 1 << 18
        buf = np.zeros(cap, np.uint32)
        n = _check(self._L.vsg_orb_copy_candidates(self._h, frame, level, _p(buf, _u32p), cap), "vsg_orb_copy_candidates")
        return self._unpack(buf[:n])

    def selected(self, level, frame=0):
        cap = 1 << 16
        buf = np.zeros(cap, np.uint32)
        n = _check(self._L.vsg_orb_copy_selected(self._h, frame, level, _p(buf, _u32p), cap), "vsg_orb_copy_selected")
        return self._unpack(buf[:n])

    def set_serialize(self, on=True):
        _check(self._L.vsg_orb_set_serialize(self._h, int(on)), "vsg_orb_set_serialize")

    def enable_timing(self, on=True):
        _check(self._L.vsg_orb_enable_timing(self._h, int(on)), "vsg_orb_enable_timing")

    def timing_ms(self):
        out = np.zeros(len(self.STAGES), np.float32)
        self._L.vsg_orb_get_timing(self._h, _p(out, _f32p), len(out))
        return dict(zip(self.STAGES, out.tolist()))


class ORBmatcher:
    """VS_GRAPHS::ORBmatcher(nnratio=0.6, checkOri=true) on flattened arrays (ORBmatcher.h:37)."""

    TH_HIGH = 100
    TH_LOW = 50
    HISTO_LENGTH = 30

    def __init__(self, nnratio=0.6, checkOri=True, device=0):
        self._L = load_library()
        self.mfNNratio = float(np.float32(nnratio))
        self.mbCheckOrientation = bool(checkOri)
        self.device = int(device)

    @staticmethod
    def DescriptorDistance(a, b, device=0):
        """ORBmatcher::DescriptorDistance for one pair (or row-wise for [n,32] arrays)."""
        a2, b2 = np.atleast_2d(_u8(a)), np.atleast_2d(_u8(b))
        n = len(a2)
        idx = np.arange(n, dtype=np.int32)
        out = np.zeros(n, np.int32)
        _check(load_library().vsg_hamming_pairs(device, _p(a2, _u8p), n, _p(b2, _u8p), len(b2), _p(idx, _i32p),
                                                _p(idx, _i32p), n, _p(out, _i32p)), "vsg_hamming_pairs")
        return int(out[0]) if np.ndim(a) == 1 else out

    def hamming_pairs(self, a, b, ia, ib):
        a, b, ia, ib = _u8(a), _u8(b), _i32(ia), _i32(ib)
        n = 0 if len(a) == 0 or len(b) == 0 else len(ia)
        out = np.zeros(n, np.int32)
        if n:
            _check(self._L.vsg_hamming_pairs(self.device, _p(a, _u8p), len(a), _p(b, _u8p), len(b), _p(ia, _i32p),
                                             _p(ib, _i32p), n, _p(out, _i32p)), "vsg_hamming_pairs")
        return out

    def block_best2(self, a, b):
        a, b = _u8(a).reshape(-1, 32), _u8(b).reshape(-1, 32)
        best, second, arg = (np.zeros(len(a), np.int32) for _ in range(3))
        if len(a):
            bb = b if len(b) else np.zeros((1, 32), np.uint8)
            _check(self._L.vsg_hamming_block_best2(self.device, _p(a, _u8p), len(a), _p(bb, _u8p), len(b),
                                                   _p(best, _i32p), _p(second, _i32p), _p(arg, _i32p)),
                   "vsg_hamming_block_best2")
        return best, second, arg

    def SearchByBoW_KF_F(self, kf_desc, kf_angle, kf_valid, kf_fv, f_desc, f_angle, f_fv, f_nleft=-1):
        """SearchByBoW(KeyFrame*, Frame&, ...).  fv = (node_ids, offsets, indices); f_nleft = F.Nleft (fisheye
        stereo: right-camera features start there).  Returns (nmatches, matchF)."""
        kd, fd = _u8(kf_desc).reshape(-1, 32), _u8(f_desc).reshape(-1, 32)
        ka, fa, kv = _f32(kf_angle), _f32(f_angle), _u8(kf_valid)
        kn, ko, ki = (_i32(x) for x in kf_fv)
        fn, fo, fi = (_i32(x) for x in f_fv)
        out = np.full(max(len(fd), 1), -1, np.int32)
        n = _check(self._L.vsg_search_by_bow_kf_f_stereo(
            self.device, _p(kd, _u8p), _p(ka, _f32p), _p(kv, _u8p), len(kd), _p(kn, _i32p), _p(ko, _i32p),
            _p(ki, _i32p), len(kf_fv[0]), _p(fd, _u8p), _p(fa, _f32p), len(fd), int(f_nleft), _p(fn, _i32p),
            _p(fo, _i32p), _p(fi, _i32p), len(f_fv[0]), self.mfNNratio, int(self.mbCheckOrientation),
            _p(out, _i32p)), "vsg_search_by_bow_kf_f_stereo")
        return n, out[:len(fd)]

    def SearchByBoW_KF_KF(self, d1, a1, v1, fv1, d2, a2, v2, fv2):
        d1, d2 = _u8(d1).reshape(-1, 32), _u8(d2).reshape(-1, 32)
        a1, a2, v1, v2 = _f32(a1), _f32(a2), _u8(v1), _u8(v2)
        n1, o1, i1 = (_i32(x) for x in fv1)
        n2, o2, i2 = (_i32(x) for x in fv2)
        out = np.full(max(len(d1), 1), -1, np.int32)
        n = _check(self._L.vsg_search_by_bow_kf_kf(self.device, _p(d1, _u8p), _p(a1, _f32p), _p(v1, _u8p), len(d1),
                                                   _p(n1, _i32p), _p(o1, _i32p), _p(i1, _i32p), len(fv1[0]),
                                                   _p(d2, _u8p), _p(a2, _f32p), _p(v2, _u8p), len(d2), _p(n2, _i32p),
                                                   _p(o2, _i32p), _p(i2, _i32p), len(fv2[0]), self.mfNNratio,
                                                   int(self.mbCheckOrientation), _p(out, _i32p)),
                   "vsg_search_by_bow_kf_kf")
        return n, out[:len(d1)]

    def SearchForTriangulation(self, d1, a1, eligible1, fv1, d2, a2, eligible2, fv2, pair_ok=None, pair_off=None):
        """ORBmatcher::SearchForTriangulation (ORBmatcher.cc:902-1146) on flattened views; pair_ok / pair_off carry
        the adaptor-evaluated geometric predicate per pair of every shared node (see include/vsg_orb.h).
        Returns (nmatches, matches12)."""
        d1, d2 = _u8(d1).reshape(-1, 32), _u8(d2).reshape(-1, 32)
        a1, a2, e1, e2 = _f32(a1), _f32(a2), _u8(eligible1), _u8(eligible2)
        n1, o1, i1 = (_i32(x) for x in fv1)
        n2, o2, i2 = (_i32(x) for x in fv2)
        ok = np.ascontiguousarray(pair_ok, np.uint32) if pair_ok is not None else None
        po = _i32(pair_off) if pair_ok is not None else None
        out = np.full(max(len(d1), 1), -1, np.int32)
        n = _check(self._L.vsg_search_for_triangulation(
            self.device, _p(d1, _u8p), _p(a1, _f32p), _p(e1, _u8p), len(d1), _p(n1, _i32p), _p(o1, _i32p),
            _p(i1, _i32p), len(fv1[0]), _p(d2, _u8p), _p(a2, _f32p), _p(e2, _u8p), len(d2), _p(n2, _i32p),
            _p(o2, _i32p), _p(i2, _i32p), len(fv2[0]), ok.ctypes.data if ok is not None else None,
            po.ctypes.data if po is not None else None, int(self.mbCheckOrientation), _p(out, _i32p)),
            "vsg_search_for_triangulation")
        return n, out[:len(d1)]

    def SearchByProjection_Last(self, q_desc, q_angle, q_blocks, cand_off, cand_idx, t_desc, t_angle, t_blocked,
                                th_high=TH_HIGH):
        qd, td = _u8(q_desc).reshape(-1, 32), _u8(t_desc).reshape(-1, 32)
        qa, ta, qb = _f32(q_angle), _f32(t_angle), _u8(q_blocks)
        co, ci = _i32(cand_off), _i32(cand_idx)
        tb = _u8(t_blocked).copy()
        tm = np.full(max(len(td), 1), -1, np.int32)
        n = _check(self._L.vsg_search_by_projection_last(self.device, _p(qd, _u8p), _p(qa, _f32p), _p(qb, _u8p),
                                                         len(qd), _p(co, _i32p), _p(ci, _i32p), _p(td, _u8p),
                                                         _p(ta, _f32p), _p(tb, _u8p), len(td), int(th_high),
                                                         int(self.mbCheckOrientation), _p(tm, _i32p)),
                   "vsg_search_by_projection_last")
        return n, tm[:len(td)], tb

    def SearchByProjection_Local(self, q_desc, q_blocks, cand_off, cand_idx, t_desc, t_octave, t_blocked):
        qd, td = _u8(q_desc).reshape(-1, 32), _u8(t_desc).reshape(-1, 32)
        qb, to = _u8(q_blocks), _i32(t_octave)
        co, ci = _i32(cand_off), _i32(cand_idx)
        tb = _u8(t_blocked).copy()
        tm = np.full(max(len(td), 1), -1, np.int32)
        n = _check(self._L.vsg_search_by_projection_local(self.device, _p(qd, _u8p), _p(qb, _u8p), len(qd),
                                                          _p(co, _i32p), _p(ci, _i32p), _p(td, _u8p), _p(to, _i32p),
                                                          _p(tb, _u8p), len(td), self.mfNNratio, _p(tm, _i32p)),
                   "vsg_search_by_projection_local")
        return n, tm[:len(td)], tb

    def SearchForInitialization(self, d1, a1, oct1, cand_off, cand_idx, d2, a2):
        d1, d2 = _u8(d1).reshape(-1, 32), _u8(d2).reshape(-1, 32)
        a1, a2, o1 = _f32(a1), _f32(a2), _i32(oct1)
        co, ci = _i32(cand_off), _i32(cand_idx)
        out = np.full(max(len(d1), 1), -1, np.int32)
        n = _check(self._L.vsg_search_for_initialization(self.device, _p(d1, _u8p), _p(a1, _f32p), _p(o1, _i32p),
                                                         len(d1), _p(co, _i32p), _p(ci, _i32p), _p(d2, _u8p),
                                                         _p(a2, _f32p), len(d2), self.mfNNratio,
                                                         int(self.mbCheckOrientation), _p(out, _i32p)),
                   "vsg_search_for_initialization")
        return n, out[:len(d1)]


class FrameGrid:
    """Frame::AssignFeaturesToGrid + GetFeaturesInArea on the device (Frame.cc:521-553, 802-880)."""

    def __init__(self, kps, min_x, min_y, max_x, max_y, device=0):
        self._L = load_library()
        self._kps = np.ascontiguousarray(kps, dtype=KP_DTYPE)
        self._h = C.c_void_p()
        _check(self._L.vsg_grid_build(int(device), self._kps.ctypes.data_as(C.c_void_p), len(self._kps), float(min_x),
                                      float(min_y), float(max_x), float(max_y), C.byref(self._h)), "vsg_grid_build")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.vsg_grid_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def GetFeaturesInArea(self, x, y, r, minLevel=None, maxLevel=None):
        """Batched query.  Returns (cand_off[nq+1], cand_idx) in the reference's candidate order."""
        x, y, r = _f32(np.atleast_1d(x)), _f32(np.atleast_1d(y)), _f32(np.atleast_1d(r))
        nq = len(x)
        lo = _i32(np.atleast_1d(minLevel)) if minLevel is not None else None
        hi = _i32(np.atleast_1d(maxLevel)) if maxLevel is not None else None
        off = np.zeros(nq + 1, np.int32)
        cap = max(1, 64 * nq)
        while True:
            idx = np.zeros(cap, np.int32)
            total = _check(self._L.vsg_grid_query(self._h, _p(x, _f32p), _p(y, _f32p), _p(r, _f32p),
                                                  _p(lo, _i32p) if lo is not None else None,
                                                  _p(hi, _i32p) if hi is not None else None, nq, _p(off, _i32p),
                                                  _p(idx, _i32p), cap), "vsg_grid_query")
            if total <= cap:
                return off, idx[:total]
            cap = total


def search_window(q_desc, q_blocks, cand_off, cand_idx, t_desc, t_blocked, th_high, device=0):
    """vsg_search_window.  Returns (nmatches, q_best_idx, q_best_dist, train_match, train_blocked)."""
    qd, td = _u8(q_desc).reshape(-1, 32), _u8(t_desc).reshape(-1, 32)
    co, ci = _i32(cand_off), _i32(cand_idx)
    qb = _u8(q_blocks) if q_blocks is not None else None
    tb = _u8(t_blocked).copy() if t_blocked is not None else None
    qi, qdist = np.zeros(max(len(qd), 1), np.int32), np.zeros(max(len(qd), 1), np.int32)
    tm = np.full(max(len(td), 1), -1, np.int32)
    n = _check(load_library().vsg_search_window(int(device), _p(qd, _u8p), _p(qb, _u8p) if qb is not None else None,
                                                len(qd), _p(co, _i32p), _p(ci, _i32p), _p(td, _u8p),
                                                _p(tb, _u8p) if tb is not None else None, len(td), int(th_high),
                                                _p(qi, _i32p), _p(qdist, _i32p), _p(tm, _i32p)), "vsg_search_window")
    return n, qi[:len(qd)], qdist[:len(qd)], tm[:len(td)], tb


def ComputeStereoMatches(ex_left, frame_l, ex_right, frame_r, kps_l, desc_l, kps_r, desc_r, mb, mbf):
    """Frame::ComputeStereoMatches (Frame.cc:957-1127).  Returns (mvuRight, mvDepth)."""
    kl, kr = np.ascontiguousarray(kps_l, KP_DTYPE), np.ascontiguousarray(kps_r, KP_DTYPE)
    dl, dr = _u8(desc_l).reshape(-1, 32), _u8(desc_r).reshape(-1, 32)
    ur, dep = np.zeros(max(len(kl), 1), np.float32), np.zeros(max(len(kl), 1), np.float32)
    _check(load_library().vsg_stereo_matches(ex_left.handle, int(frame_l), ex_right.handle, int(frame_r),
                                             kl.ctypes.data_as(C.c_void_p), _p(dl, _u8p), len(kl),
                                             kr.ctypes.data_as(C.c_void_p), _p(dr, _u8p), len(kr), float(mb), float(mbf),
                                             _p(ur, _f32p), _p(dep, _f32p)), "vsg_stereo_matches")
    return ur[:len(kl)], dep[:len(kl)]


class ORBVocabulary:
    """DBoW2 ORBVocabulary on the device: loadFromBinFile image + transform (TemplatedVocabulary.h)."""

    def __init__(self, blob, device=0):
        self._L = load_library()
        b = np.frombuffer(blob, dtype=np.uint8).copy()
        self._h = C.c_void_p()
        _check(self._L.vsg_vocab_load(int(device), _p(b, _u8p), len(b), C.byref(self._h)), "vsg_vocab_load")
        v = [C.c_int32() for _ in range(6)]
        self._L.vsg_vocab_info(self._h, *[C.byref(x) for x in v])
        self.k, self.L, self.scoring, self.weighting, self.nnodes, self.nwords = [x.value for x in v]

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.vsg_vocab_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def transform(self, desc, levelsup=4):
        """Returns dict(bow_ids, bow_vals, fv=(node_ids, offsets, indices), word, node, weight)."""
        d = _u8(desc).reshape(-1, 32)
        n = len(d)
        cap = n + 1
        bi, bv = np.zeros(cap, np.int32), np.zeros(cap, np.float64)
        fn, fo, fi = np.zeros(cap, np.int32), np.zeros(cap + 1, np.int32), np.zeros(cap, np.int32)
        w_of, n_of, wt = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.float64)
        nb, nf = C.c_int32(), C.c_int32()
        f64p = C.POINTER(C.c_double)
        _check(self._L.vsg_bow_transform(self._h, _p(d, _u8p), n, int(levelsup), _p(bi, _i32p), _p(bv, f64p), cap,
                                         C.byref(nb), _p(fn, _i32p), _p(fo, _i32p), _p(fi, _i32p), cap, C.byref(nf),
                                         _p(w_of, _i32p), _p(n_of, _i32p), _p(wt, f64p)), "vsg_bow_transform")
        return dict(bow_ids=bi[:nb.value].copy(), bow_vals=bv[:nb.value].copy(),
                    fv=(fn[:nf.value].copy(), fo[:nf.value + 1].copy(), fi[:fo[nf.value]].copy()),
                    word=w_of[:n].copy(), node=n_of[:n].copy(), weight=wt[:n].copy())


def ComputeDistinctiveDescriptors(desc, off, device=0):
    """MapPoint::ComputeDistinctiveDescriptors for many map points: returns the chosen row index per group."""
    d, o = _u8(desc).reshape(-1, 32), _i32(off)
    if len(d) == 0:
        d = np.zeros((1, 32), np.uint8)
    best = np.zeros(max(len(o) - 1, 1), np.int32)
    _check(load_library().vsg_distinctive_descriptors(int(device), _p(d, _u8p), _p(o, _i32p), len(o) - 1,
                                                      _p(best, _i32p)), "vsg_distinctive_descriptors")
    return best[:len(o) - 1]
