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

import os

_PKG = Path(__file__).resolve().parent
# VSG_LIB=<path> loads an experimental build instead of the in-tree library (A/B scripts under tools/): the in-tree
# file is never overwritten, so later runs cannot silently pick up a variant
LIB_PATH = Path(os.environ["VSG_LIB"]) if os.environ.get("VSG_LIB") else _PKG / "libvsg_orb.so"

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28

VSG_OK = 0
ERRORS = {-1: "VSG_ERR_EMPTY_IMAGE", -2: "VSG_ERR_CAPACITY", -3: "VSG_ERR_UNSUPPORTED", -4: "VSG_ERR_NO_DEVICE",
          -5: "VSG_ERR_HIP", -6: "VSG_ERR_INVALID", -7: "VSG_ERR_BUSY"}

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
    # round 2: threads / staging, async host pipeline, device-resident frames, routine-level searches
    "vsg_thread_release", "vsg_thread_arena_growths", "vsg_debug_call_profile", "vsg_orb_time_stats", "vsg_host_register", "vsg_host_unregister", "vsg_host_alloc", "vsg_host_free", "vsg_orb_slots",
    "vsg_orb_submit_batch", "vsg_orb_wait", "vsg_orb_copy_pyramid", "vsg_frame_create", "vsg_frame_destroy",
    "vsg_frame_upload", "vsg_frame_from_extractor", "vsg_frame_size", "vsg_frame_copy_grid",
    "vsg_frame_features_in_area", "vsg_frame_search_by_projection", "vsg_frame_search_by_projection_last",
    "vsg_frame_search_by_projection_sim3", "vsg_frame_search_by_projection_kf", "vsg_frame_search_by_sim3",
    "vsg_frame_fuse", "vsg_frame_fuse_sim3", "vsg_fuse_decide", "vsg_frame_search_for_initialization",
    "vsg_frame_search_by_bow_kf_f", "vsg_frame_search_by_bow_kf_kf", "vsg_frame_bow_transform", "vsg_frame_stereo_bow_search",
    "vsg_frame_search_for_triangulation",
    "vsg_frame_stereo_matches",
    "vsg_shard_last_error", "vsg_shard_record_bytes", "vsg_shard_record_desc_offset", "vsg_shard_frame_owner",
    "vsg_shard_stream_owner", "vsg_shard_unique_id", "vsg_shard_create", "vsg_shard_destroy", "vsg_shard_all_gather",
    "vsg_shard_record", "vsg_shard_world", "vsg_shard_send_recv_boundary", "vsg_shard_boundary_record",
    "vsg_copy_d2d_async", "vsg_orb_chain_graph_launches",
    # round 4
    "vsg_orb_set_pyramid_tiling", "vsg_shard_rank", "vsg_camera_image_bounds", "vsg_frame_from_extractor_undistort",
    "vsg_orb_extract_to_frame",
    # round 5
    "vsg_host_kind", "vsg_orb_set_direct_registered",
]


class VsgError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        super().__init__(f"{where}: {ERRORS.get(code, code)} {detail}".strip())


_lib = None


def camera_image_bounds(cols, rows, K4, dist):
    """Frame::ComputeImageBounds (Frame.cc:924-955): (mnMinX, mnMinY, mnMaxX, mnMaxY); host arithmetic in the library."""
    L = load_library()
    K4, dist = np.ascontiguousarray(K4, np.float32), np.ascontiguousarray(dist, np.float32)
    out = np.zeros(4, np.float32)
    _check(L.vsg_camera_image_bounds(int(cols), int(rows), _p(K4, _f32p), _p(dist, _f32p), len(dist), _p(out, _f32p)),
           "vsg_camera_image_bounds")
    return tuple(float(v) for v in out)


def debug_device_sort(items, device=0):
    """Test hook: the device's std::sort replay (see include/vsg_orb.h) on uint64 items (key = upper 32 bits)."""
    L = load_library()
    a = np.ascontiguousarray(items, dtype=np.uint64).copy()
    L.vsg_debug_device_sort.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.c_int]
    _check(L.vsg_debug_device_sort(device, a.ctypes.data_as(C.POINTER(C.c_uint64)), len(a)), "vsg_debug_device_sort")
    return a


def _share_torch_hip_runtime():
    """PyTorch-ROCm ships its own libamdhip64; libvsg_orb.so links the system one under the same soname.  Whichever
    is loaded first serves both, and torch only finds its GPUs through its own copy -- so when torch is installed its
    copy is loaded here, by path and without importing torch, before libvsg_orb.so resolves the soname.  One HIP
    runtime per process then holds no matter whether `import torch` or this module comes first (streams and device
    pointers are handed between the two, bench.py)."""
    import importlib.util
    import os
    import sys
    if "torch" in sys.modules or os.environ.get("VSG_SYSTEM_HIP"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if not spec or not spec.origin:
        return
    cand = Path(spec.origin).parent / "lib" / "libamdhip64.so"
    if cand.exists():
        try:
            C.CDLL(str(cand), mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library():
    """dlopen libvsg_orb.so (built in-tree by visual_sgraphs_amd.build).  Fails loudly if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -m visual_sgraphs_amd.build` "
                           "(the ORB front-end has no CPU fallback)")
    _share_torch_hip_runtime()
    L = C.CDLL(str(LIB_PATH))
    L.vsg_last_error.restype = C.c_char_p
    L.vsg_orb_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.vsg_orb_destroy.argtypes = [C.c_void_p]
    L.vsg_orb_destroy.restype = None
    L.vsg_orb_get_tables.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, _f32p, _i32p, _i32p]
    L.vsg_orb_set_blur_taps.argtypes = [C.c_void_p, _u16p]
    L.vsg_orb_capacity.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.vsg_orb_set_pyramid_tiling.argtypes = [C.c_void_p, C.c_int]
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
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    L.vsg_thread_arena_growths.argtypes = [ci]
    L.vsg_orb_time_stats.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), ci]
    L.vsg_host_register.argtypes = [vp, C.c_size_t]
    L.vsg_host_unregister.argtypes = [vp]
    L.vsg_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    L.vsg_host_free.argtypes = [vp]
    L.vsg_host_kind.argtypes = [vp, C.c_size_t]
    L.vsg_orb_set_direct_registered.argtypes = [vp, ci]
    L.vsg_orb_slots.argtypes = [vp]
    L.vsg_orb_chain_graph_launches.argtypes = [vp]
    L.vsg_orb_chain_graph_launches.restype = C.c_long
    L.vsg_orb_submit_batch.argtypes = [vp, vp, ci, C.c_size_t, ci, ci, ci, ci, ci, vp, vp, ci]
    L.vsg_orb_wait.argtypes = [vp, ci, _i32p, _i32p]
    L.vsg_orb_copy_pyramid.argtypes = [vp, ci, _u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.vsg_frame_create.argtypes = [ci, ci, C.POINTER(vp)]
    L.vsg_frame_destroy.argtypes = [vp]
    L.vsg_frame_destroy.restype = None
    L.vsg_frame_upload.argtypes = [vp, vp, _u8p, _f32p, ci, ci, cf, cf, cf, cf]
    L.vsg_frame_from_extractor.argtypes = [vp, vp, ci, vp, ci, cf, cf, cf, cf]
    L.vsg_frame_from_extractor_undistort.argtypes = [vp, vp, ci, vp, ci, _f32p, _f32p, ci, cf, cf, cf, cf, vp]
    L.vsg_camera_image_bounds.argtypes = [ci, ci, _f32p, _f32p, ci, _f32p]
    L.vsg_orb_extract_to_frame.argtypes = [vp, _u8p, ci, ci, ci, ci, ci, vp, _u8p, ci, _i32p, vp, _f32p, _f32p, ci, cf, cf,
                                           cf, cf, vp]
    L.vsg_frame_size.argtypes = [vp]
    L.vsg_frame_copy_grid.argtypes = [vp, ci, _i32p, _i32p]
    L.vsg_frame_features_in_area.argtypes = [vp, _f32p, _f32p, _f32p, _i32p, _i32p, ci, ci, _i32p, _i32p, ci]
    L.vsg_frame_search_by_projection.argtypes = [vp, ci, _u8p, _u8p, _u8p, _f32p, _f32p, _f32p, _i32p, _f32p, _u8p,
                                                 _f32p, _f32p, _i32p, _f32p, cf, cf, _f32p, ci, _i32p, _i32p, _u8p,
                                                 _i32p]
    L.vsg_frame_search_by_projection_last.argtypes = [vp, ci, _u8p, _u8p, _f32p, _f32p, _f32p, _f32p, _f32p, _i32p,
                                                      _f32p, cf, ci, _f32p, ci, ci, _u8p, _i32p]
    L.vsg_frame_search_by_projection_sim3.argtypes = [vp, ci, _u8p, _f32p, _f32p, _f32p, _i32p, cf, _i32p]
    L.vsg_frame_search_by_projection_kf.argtypes = [vp, ci, _u8p, _f32p, _f32p, _f32p, _i32p, _f32p, ci, ci, _u8p,
                                                    _i32p]
    L.vsg_frame_search_by_sim3.argtypes = [vp, vp, ci, _i32p, _u8p, _f32p, _f32p, _f32p, _i32p, ci, _i32p, _u8p, _f32p,
                                           _f32p, _f32p, _i32p, _i32p]
    L.vsg_frame_fuse.argtypes = [vp, ci, _u8p, _f32p, _f32p, _f32p, _f32p, _i32p, ci, _f32p, ci, _i32p, _i32p]
    L.vsg_frame_fuse_sim3.argtypes = [vp, ci, _u8p, _f32p, _f32p, _f32p, _i32p, _i32p, _i32p]
    L.vsg_fuse_decide.argtypes = [ci, _i32p, _i32p, _i32p, ci, _i32p, ci, _i32p, _u8p, ci, _i32p, _i32p]
    L.vsg_frame_search_for_initialization.argtypes = [vp, vp, _f32p, _f32p, ci, cf, ci, _i32p]
    L.vsg_frame_search_by_bow_kf_f.argtypes = [vp, _u8p, _i32p, _i32p, _i32p, ci, vp, _i32p, _i32p, _i32p, ci, cf, ci,
                                               _i32p]
    L.vsg_frame_search_by_bow_kf_kf.argtypes = [vp, _u8p, _i32p, _i32p, _i32p, ci, vp, _u8p, _i32p, _i32p, _i32p, ci,
                                                cf, ci, _i32p]
    L.vsg_frame_search_for_triangulation.argtypes = [vp, _u8p, _i32p, _i32p, _i32p, ci, vp, _u8p, _i32p, _i32p, _i32p,
                                                     ci, vp, vp, ci, _i32p]
    L.vsg_frame_bow_transform.argtypes = [vp, vp, ci, _i32p, _f64p, ci, _i32p, _i32p, _i32p, _i32p, ci, _i32p, _i32p,
                                          _i32p, _f64p]
    L.vsg_frame_stereo_matches.argtypes = [vp, ci, vp, ci, vp, vp, cf, cf, _f32p, _f32p]
    L.vsg_frame_stereo_bow_search.argtypes = [vp, ci, vp, ci, vp, vp, cf, cf, _f32p, _f32p, _i32p, vp, ci, _i32p, _f64p, ci,
                                              _i32p, _i32p, _i32p, _i32p, ci, _i32p, vp, _u8p, cf, ci, _i32p, _i32p]
    L.vsg_shard_last_error.restype = C.c_char_p
    L.vsg_shard_record_bytes.restype = C.c_size_t
    L.vsg_shard_record_bytes.argtypes = [ci]
    L.vsg_shard_record_desc_offset.restype = C.c_size_t
    L.vsg_shard_record_desc_offset.argtypes = [ci]
    L.vsg_shard_frame_owner.argtypes = [ci, ci]
    L.vsg_shard_stream_owner.argtypes = [ci, ci, ci, ci]
    L.vsg_shard_unique_id.argtypes = [_u8p]
    L.vsg_shard_create.argtypes = [ci, ci, ci, _u8p, ci, ci, C.POINTER(vp)]
    L.vsg_shard_destroy.argtypes = [vp]
    L.vsg_shard_destroy.restype = None
    L.vsg_shard_all_gather.argtypes = [vp, vp, vp, vp, ci, ci, vp]
    L.vsg_shard_record.argtypes = [vp, ci, ci, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.vsg_shard_world.argtypes = [vp]
    L.vsg_shard_rank.argtypes = [vp]
    L.vsg_shard_send_recv_boundary.argtypes = [vp, vp, vp, vp, ci, ci, vp]
    L.vsg_shard_boundary_record.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.vsg_copy_d2d_async.argtypes = [ci, vp, vp, C.c_size_t, vp]
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

    def set_pyramid_tiling(self, which):
        """ComputePyramid's launch form: -1 automatic (default), 0 / 1 / 2 the fused tilings (32 / 36 / 16 px), 3 one launch
        per level."""
        _check(self._L.vsg_orb_set_pyramid_tiling(self._h, int(which)), "vsg_orb_set_pyramid_tiling")

    def set_direct_registered(self, on=True):
        """Opt in to in-place device access of hipHostRegister-ed caller memory (`pin()`); by default such memory is staged
        like pageable memory and only vsg_host_alloc / hipHostMalloc memory (`PinnedArray`) is touched in place."""
        _check(self._L.vsg_orb_set_direct_registered(self._h, int(on)), "vsg_orb_set_direct_registered")

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

    # ---- asynchronous host pipeline (vsg_orb_submit_batch / vsg_orb_wait)
    def slots(self):
        return self._L.vsg_orb_slots(self._h)

    def chain_graph_launches(self):
        """Blocking calls served by one hipGraphLaunch so far (latency mode, include/vsg_orb.h)."""
        return self._L.vsg_orb_chain_graph_launches(self._h)

    def submit_batch(self, images, kps_out, desc_out, vLappingArea=(0, 0)):
        """images [B,H,W] uint8 (pinned via `pin()` => DMA straight from it), kps_out [B,cap] KP_DTYPE, desc_out
        [B,cap,32] uint8 (pinned => the device writes them directly).  Returns a ticket for wait()."""
        assert images.dtype == np.uint8 and images.ndim == 3 and images.strides[2] == 1
        B, rows, cols = images.shape
        cap = kps_out.shape[1]
        assert kps_out.dtype == KP_DTYPE and kps_out.flags.c_contiguous and desc_out.flags.c_contiguous
        t = _check(self._L.vsg_orb_submit_batch(self._h, C.c_void_p(images.ctypes.data), B, images.strides[0], rows, cols,
                                                images.strides[1], int(vLappingArea[0]), int(vLappingArea[1]),
                                                C.c_void_p(kps_out.ctypes.data), C.c_void_p(desc_out.ctypes.data), cap),
                   "vsg_orb_submit_batch")
        self._shape = (rows, cols)
        self._inflight = getattr(self, "_inflight", {})
        self._inflight[t] = (B, images, kps_out, desc_out)  # keep the buffers alive until the wait
        return t

    def wait(self, ticket):
        """Blocks until the batch of `ticket` is complete.  Returns (n[B], monoIndex[B])."""
        B = self._inflight[ticket][0]
        n, mono = np.zeros(B, np.int32), np.zeros(B, np.int32)
        _check(self._L.vsg_orb_wait(self._h, int(ticket), _p(n, _i32p), _p(mono, _i32p)), "vsg_orb_wait")
        del self._inflight[ticket]
        return n, mono

    def copy_pyramid(self, frame=0):
        """mvImagePyramid of one frame in ONE D2H: list of bordered levels ((h+38) x (w+38) uint8 views)."""
        offs = (C.c_size_t * self.nlevels)()
        need = _check(self._L.vsg_orb_copy_pyramid(self._h, frame, None, 0, offs), "vsg_orb_copy_pyramid")
        buf = np.zeros(need, np.uint8)
        _check(self._L.vsg_orb_copy_pyramid(self._h, frame, _p(buf, _u8p), need, offs), "vsg_orb_copy_pyramid")
        out = []
        for l in range(self.nlevels):
            w, h = self.level_size(l)
            out.append(buf[offs[l]:offs[l] + (w + 38) * (h + 38)].reshape(h + 38, w + 38))
        return out

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

    def time_stats(self, reset=False):
        """REGISTER_TIMES analogue: (calls, mean_ms, std_ms) of the blocking operator() calls so far."""
        m, sd = C.c_double(), C.c_double()
        n = _check(self._L.vsg_orb_time_stats(self._h, C.byref(m), C.byref(sd), int(reset)), "vsg_orb_time_stats")
        return n, m.value, sd.value

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
        """ORBmatcher::DescriptorDistance (ORBmatcher.cc:2047-2063).  One pair is eight popcounts: host arithmetic
        (a kernel launch would cost four orders of magnitude more); [n,32] arrays go row-wise through the device."""
        if np.ndim(a) == 1:
            x = np.bitwise_xor(_u8(a), _u8(b))
            return int(np.unpackbits(x).sum())
        a2, b2 = np.atleast_2d(_u8(a)), np.atleast_2d(_u8(b))
        n = len(a2)
        idx = np.arange(n, dtype=np.int32)
        out = np.zeros(n, np.int32)
        _check(load_library().vsg_hamming_pairs(device, _p(a2, _u8p), n, _p(b2, _u8p), len(b2), _p(idx, _i32p),
                                                _p(idx, _i32p), n, _p(out, _i32p)), "vsg_hamming_pairs")
        return out

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


HOST_KINDS = {0: "pageable", 1: "vsg_host_alloc", 2: "hipHostMalloc", 3: "registered"}


def host_kind(array):
    """How the library classifies the whole memory range of a (contiguous) numpy array: one of HOST_KINDS' values."""
    a = np.asarray(array)
    return HOST_KINDS[_check(load_library().vsg_host_kind(C.c_void_p(a.ctypes.data), max(a.nbytes, 1)), "vsg_host_kind")]


def pin(array):
    """hipHostRegister a numpy array (vsg_host_register).  Registered heap memory is only touched in place by handles
    that opted in (ORBextractor.set_direct_registered); prefer PinnedArray."""
    a = np.ascontiguousarray(array)
    _check(load_library().vsg_host_register(C.c_void_p(a.ctypes.data), a.nbytes), "vsg_host_register")
    return a


def unpin(array):
    _check(load_library().vsg_host_unregister(C.c_void_p(array.ctypes.data)), "vsg_host_unregister")


def _host_free(ptr):
    lib = _lib
    if lib is not None and ptr:
        lib.vsg_host_free(C.c_void_p(ptr))


class PinnedArray:
    """A numpy array over hipHostMalloc memory (vsg_host_alloc): pinned without a user-pointer mapping of heap pages
    underneath; the device reads / writes it in place.  `.a` is the array.  The memory lives as long as this object OR any
    numpy view derived from `.a` (the views hold the buffer object whose finalizer frees the allocation), so dropping the
    PinnedArray while a view is still in use -- or in flight -- is safe.  free() releases this object's own references;
    the allocation itself goes when the last view does."""

    class _Buf:  # owns the allocation; exposes it through the buffer protocol via a ctypes array
        def __init__(self, nbytes):
            import weakref
            p = C.c_void_p()
            _check(load_library().vsg_host_alloc(C.c_size_t(nbytes), C.byref(p)), "vsg_host_alloc")
            self.ptr = p.value
            self.raw = (C.c_uint8 * nbytes).from_address(self.ptr)
            # the ctypes array is what numpy keeps alive (ndarray.base chain): its death frees the memory
            self._fin = weakref.finalize(self.raw, _host_free, self.ptr)

    def __init__(self, shape, dtype=np.uint8):
        dt = np.dtype(dtype)
        count = int(np.prod(shape))
        n = max(count * dt.itemsize, 1)
        buf = PinnedArray._Buf(n)
        self._ptr = buf.ptr
        self._fin = buf._fin
        self.a = np.frombuffer(buf.raw, dtype=dt, count=count).reshape(shape)

    @property
    def ptr(self):
        return self._ptr

    def free(self):
        """Drop this object's array.  The pinned memory is released as soon as no numpy view of it is left -- it is NOT
        returned to the runtime here if a slice, a reshape or an exception traceback still holds one: use
        release() where the pinned footprint has to be bounded."""
        self.a = None

    @property
    def alive(self):
        """True while the hipHostMalloc block behind this object is still allocated (this object or a view holds it)."""
        return self._fin.alive

    def release(self):
        """free() + a check that the allocation really went: raises if a view of the array is still referenced
        somewhere (the memory then stays pinned until that view dies)."""
        self.a = None
        import gc
        if self._fin.alive:
            gc.collect()
        if self._fin.alive:
            raise RuntimeError("PinnedArray.release(): a numpy view of the array is still alive; the block stays pinned until "
                               "it is dropped")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.free()


def copy_d2d_async(dst, src, nbytes, stream, device=0):
    """Device-to-device copy of raw pointers (ints) on `stream` through the library's own HIP runtime."""
    _check(load_library().vsg_copy_d2d_async(int(device), C.c_void_p(dst), C.c_void_p(src), int(nbytes),
                                            C.c_void_p(stream) if stream else None), "vsg_copy_d2d_async")


def thread_arena_growths(device=0):
    return load_library().vsg_thread_arena_growths(int(device))


def _opt(a, t, conv):
    return (None, None) if a is None else (lambda x: (x, _p(x, t)))(conv(a))


class Frame:
    """What the searches read of a VS_GRAPHS::Frame / KeyFrame, resident on the device (include/vsg_orb.h: vsg_frame):
    mvKeysUn (or mvKeys || mvKeysRight with Nleft), mDescriptors, mvuRight, mGrid / mGridRight (Frame.h:280-290)."""

    def __init__(self, capacity, device=0):
        self._L = load_library()
        self._h = C.c_void_p()
        self.device = int(device)
        _check(self._L.vsg_frame_create(self.device, int(capacity), C.byref(self._h)), "vsg_frame_create")
        self.kps = np.zeros(0, KP_DTYPE)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.vsg_frame_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    @property
    def N(self):
        return self._L.vsg_frame_size(self._h)

    def upload(self, kps, desc, bounds, u_right=None, nleft=-1):
        """bounds = (mnMinX, mnMinY, mnMaxX, mnMaxY)."""
        k = np.ascontiguousarray(kps, dtype=KP_DTYPE)
        d = _u8(desc).reshape(-1, 32)
        assert len(k) == len(d)
        ur = _f32(u_right) if u_right is not None else None
        _check(self._L.vsg_frame_upload(self._h, k.ctypes.data_as(C.c_void_p), _p(d, _u8p) if len(d) else None,
                                        _p(ur, _f32p) if ur is not None else None, len(k), int(nleft),
                                        *[float(b) for b in bounds]), "vsg_frame_upload")
        self.kps, self.nleft = k, int(nleft)
        return self

    def from_extractor(self, ex, index, kps, bounds):
        """Device-to-device from frame `index` of the extractor's last call; kps = the records that call returned."""
        k = np.ascontiguousarray(kps, dtype=KP_DTYPE)
        _check(self._L.vsg_frame_from_extractor(self._h, ex.handle, int(index), k.ctypes.data_as(C.c_void_p), len(k),
                                                *[float(b) for b in bounds]), "vsg_frame_from_extractor")
        self.kps, self.nleft = k, -1
        return self

    def from_extractor_undistort(self, ex, index, kps, K4, dist, bounds):
        """The same for a distorted pinhole camera: Frame::UndistortKeyPoints (Frame.cc:891-921) on the device inside the
        grid launch.  self.kps = mvKeysUn afterwards."""
        k = np.ascontiguousarray(kps, dtype=KP_DTYPE)
        K4, dist = _f32(np.asarray(K4)), _f32(np.asarray(dist))
        un = np.zeros(len(k), KP_DTYPE)
        _check(self._L.vsg_frame_from_extractor_undistort(
            self._h, ex.handle, int(index), k.ctypes.data_as(C.c_void_p), len(k), _p(K4, _f32p), _p(dist, _f32p), len(dist),
            *[float(b) for b in bounds], un.ctypes.data_as(C.c_void_p)), "vsg_frame_from_extractor_undistort")
        self.kps, self.nleft = un, -1
        return self

    def extract_into(self, ex, image, bounds, K4=None, dist=None, vLappingArea=(0, 0)):
        """Frame::Frame front end in one call and one wait: operator() -> UndistortKeyPoints -> AssignFeaturesToGrid
        (vsg_orb_extract_to_frame).  Returns (monoIndex, mvKeys, descriptors); self.kps = mvKeysUn."""
        img = np.ascontiguousarray(image, dtype=np.uint8)
        rows, cols = img.shape
        cap = ex.capacity(rows, cols)
        kps, desc, un = np.zeros(cap, KP_DTYPE), np.zeros((cap, 32), np.uint8), np.zeros(cap, KP_DTYPE)
        n = C.c_int32(0)
        k4 = _f32(np.asarray(K4)) if K4 is not None else None
        d = _f32(np.asarray(dist)) if dist is not None else None
        mono = _check(self._L.vsg_orb_extract_to_frame(
            ex.handle, _p(img, _u8p), rows, cols, img.strides[0], int(vLappingArea[0]), int(vLappingArea[1]),
            kps.ctypes.data_as(C.c_void_p), _p(desc, _u8p), cap, C.byref(n), self._h, _p(k4, _f32p) if k4 is not None else None,
            _p(d, _f32p) if d is not None else None, len(d) if d is not None else 0, *[float(b) for b in bounds],
            un.ctypes.data_as(C.c_void_p)), "vsg_orb_extract_to_frame")
        self.kps, self.nleft = un[:n.value].copy(), -1
        return mono, kps[:n.value].copy(), desc[:n.value].copy()

    def grid(self, right=False):
        cs, en = np.zeros(64 * 48 + 1, np.int32), np.zeros(max(self.N, 1), np.int32)
        ne = _check(self._L.vsg_frame_copy_grid(self._h, int(right), _p(cs, _i32p), _p(en, _i32p)), "vsg_frame_copy_grid")
        return cs, en[:ne]

    def GetFeaturesInArea(self, x, y, r, minLevel=None, maxLevel=None, bRight=False):
        x, y, r = _f32(np.atleast_1d(x)), _f32(np.atleast_1d(y)), _f32(np.atleast_1d(r))
        nq = len(x)
        lo = _i32(np.atleast_1d(minLevel)) if minLevel is not None else None
        hi = _i32(np.atleast_1d(maxLevel)) if maxLevel is not None else None
        off = np.zeros(nq + 1, np.int32)
        cap = max(1, 64 * nq)
        while True:
            idx = np.zeros(cap, np.int32)
            total = _check(self._L.vsg_frame_features_in_area(
                self._h, _p(x, _f32p), _p(y, _f32p), _p(r, _f32p), _p(lo, _i32p) if lo is not None else None,
                _p(hi, _i32p) if hi is not None else None, int(bRight), nq, _p(off, _i32p), _p(idx, _i32p), cap),
                "vsg_frame_features_in_area")
            if total <= cap:
                return off, idx[:total]
            cap = total

    # ---- ORBmatcher::SearchByProjection(Frame &F, vpMapPoints, th, ...)  (ORBmatcher.cc:42-216)
    def SearchByProjection(self, mp, th, nnratio, scale_factors, train_blocked, left_to_right=None, right_to_left=None):
        """mp: dict of per-map-point arrays (desc, observed, in_view, proj_x, proj_y, proj_xr, scale_level, view_cos
        and, for Nleft != -1, in_view_r, proj_x_r, proj_y_r, scale_level_r, view_cos_r).
        Returns (nmatches, train_match, train_blocked)."""
        d = _u8(mp["desc"]).reshape(-1, 32)
        n = len(d)
        obs, inv = _u8(mp["observed"]), _u8(mp["in_view"])
        px, py, lvl, vc = _f32(mp["proj_x"]), _f32(mp["proj_y"]), _i32(mp["scale_level"]), _f32(mp["view_cos"])
        pxr = _f32(mp["proj_xr"]) if mp.get("proj_xr") is not None else None
        R = {}
        for key, conv, t in (("in_view_r", _u8, _u8p), ("proj_x_r", _f32, _f32p), ("proj_y_r", _f32, _f32p),
                             ("scale_level_r", _i32, _i32p), ("view_cos_r", _f32, _f32p)):
            R[key] = conv(mp[key]) if mp.get(key) is not None else None
        sf = _f32(scale_factors)
        ltr = _i32(left_to_right) if left_to_right is not None else None
        rtl = _i32(right_to_left) if right_to_left is not None else None
        tb = _u8(train_blocked).copy()
        tm = np.full(max(len(tb), 1), -1, np.int32)

        def o(a, t):
            return _p(a, t) if a is not None else None
        nm = _check(self._L.vsg_frame_search_by_projection(
            self._h, n, _p(d, _u8p), _p(obs, _u8p), _p(inv, _u8p), _p(px, _f32p), _p(py, _f32p), o(pxr, _f32p),
            _p(lvl, _i32p), _p(vc, _f32p), o(R["in_view_r"], _u8p), o(R["proj_x_r"], _f32p), o(R["proj_y_r"], _f32p),
            o(R["scale_level_r"], _i32p), o(R["view_cos_r"], _f32p), float(th), float(np.float32(nnratio)), _p(sf, _f32p),
            len(scale_factors), o(ltr, _i32p), o(rtl, _i32p), _p(tb, _u8p), _p(tm, _i32p)),
            "vsg_frame_search_by_projection")
        return nm, tm[:len(tb)], tb

    # ---- SearchByProjection(CurrentFrame, LastFrame, th, bMono)  (ORBmatcher.cc:1667-1878)
    def SearchByProjection_Last(self, desc, observed, u, v, ur, last_octave, last_angle, th, direction, scale_factors,
                                check_orientation, train_blocked, u_r=None, v_r=None):
        d = _u8(desc).reshape(-1, 32)
        obs = _u8(observed)
        uu, vv, oc, an = _f32(u), _f32(v), _i32(last_octave), _f32(last_angle)
        ur_ = _f32(ur) if ur is not None else None
        ur2, vr2 = (_f32(u_r), _f32(v_r)) if u_r is not None else (None, None)
        sf = _f32(scale_factors)
        tb = _u8(train_blocked).copy()
        tm = np.full(max(len(tb), 1), -1, np.int32)
        nm = _check(self._L.vsg_frame_search_by_projection_last(
            self._h, len(d), _p(d, _u8p), _p(obs, _u8p), _p(uu, _f32p), _p(vv, _f32p),
            _p(ur_, _f32p) if ur_ is not None else None, _p(ur2, _f32p) if ur2 is not None else None,
            _p(vr2, _f32p) if vr2 is not None else None, _p(oc, _i32p), _p(an, _f32p), float(th), int(direction),
            _p(sf, _f32p), len(scale_factors), int(check_orientation), _p(tb, _u8p), _p(tm, _i32p)),
            "vsg_frame_search_by_projection_last")
        return nm, tm[:len(tb)], tb

    # ---- SearchByProjection(KeyFrame*, Sim3, vpPoints, vpMatched, th, ratioHamming)  (ORBmatcher.cc:430-641)
    def SearchByProjection_Sim3(self, desc, u, v, radius, predicted_level, ratio_hamming, matched):
        d = _u8(desc).reshape(-1, 32)
        m = _i32(matched).copy()
        nm = _check(self._L.vsg_frame_search_by_projection_sim3(
            self._h, len(d), _p(d, _u8p), _p(_f32(u), _f32p), _p(_f32(v), _f32p), _p(_f32(radius), _f32p),
            _p(_i32(predicted_level), _i32p), float(np.float32(ratio_hamming)), _p(m, _i32p)),
            "vsg_frame_search_by_projection_sim3")
        return nm, m[:len(matched)]

    # ---- SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist)  (ORBmatcher.cc:1880-2000)
    def SearchByProjection_KF(self, desc, u, v, radius, predicted_level, kf_angle, orb_dist, check_orientation, occupied):
        d = _u8(desc).reshape(-1, 32)
        oc = _u8(occupied).copy()
        tm = np.full(max(len(oc), 1), -1, np.int32)
        nm = _check(self._L.vsg_frame_search_by_projection_kf(
            self._h, len(d), _p(d, _u8p), _p(_f32(u), _f32p), _p(_f32(v), _f32p), _p(_f32(radius), _f32p),
            _p(_i32(predicted_level), _i32p), _p(_f32(kf_angle), _f32p), int(orb_dist), int(check_orientation),
            _p(oc, _u8p), _p(tm, _i32p)), "vsg_frame_search_by_projection_kf")
        return nm, tm[:len(occupied)], oc

    # ---- Fuse x2 (ORBmatcher.cc:1148-1446): the search part
    def Fuse(self, desc, u, v, ur, radius, predicted_level, inv_level_sigma2, right=False):
        d = _u8(desc).reshape(-1, 32)
        bi, bd = np.zeros(max(len(d), 1), np.int32), np.zeros(max(len(d), 1), np.int32)
        s2 = _f32(inv_level_sigma2)
        nf = _check(self._L.vsg_frame_fuse(self._h, len(d), _p(d, _u8p), _p(_f32(u), _f32p), _p(_f32(v), _f32p),
                                           _p(_f32(ur), _f32p), _p(_f32(radius), _f32p),
                                           _p(_i32(predicted_level), _i32p), int(right), _p(s2, _f32p),
                                           len(inv_level_sigma2), _p(bi, _i32p), _p(bd, _i32p)), "vsg_frame_fuse")
        return nf, bi[:len(d)], bd[:len(d)]

    def Fuse_Sim3(self, desc, u, v, radius, predicted_level):
        d = _u8(desc).reshape(-1, 32)
        bi, bd = np.zeros(max(len(d), 1), np.int32), np.zeros(max(len(d), 1), np.int32)
        nf = _check(self._L.vsg_frame_fuse_sim3(self._h, len(d), _p(d, _u8p), _p(_f32(u), _f32p), _p(_f32(v), _f32p),
                                                _p(_f32(radius), _f32p), _p(_i32(predicted_level), _i32p),
                                                _p(bi, _i32p), _p(bd, _i32p)), "vsg_frame_fuse_sim3")
        return nf, bi[:len(d)], bd[:len(d)]

    def SearchForInitialization(self, f2, prev_x, prev_y, window_size, nnratio, check_orientation):
        """self = F1, f2 = F2 (both resident)."""
        out = np.full(max(self.N, 1), -1, np.int32)
        nm = _check(self._L.vsg_frame_search_for_initialization(
            self._h, f2.handle, _p(_f32(prev_x), _f32p), _p(_f32(prev_y), _f32p), int(window_size),
            float(np.float32(nnratio)), int(check_orientation), _p(out, _i32p)), "vsg_frame_search_for_initialization")
        return nm, out[:self.N]

    def SearchByBoW_KF_F(self, kf_valid, kf_fv, f, f_fv, nnratio, check_orientation):
        """self = KeyFrame, f = Frame (both resident).  kf_fv = f_fv = None: the FeatureVectors both frames keep resident
        since their ComputeBoW are joined on the device.  Returns (nmatches, matchF)."""
        kv = _u8(kf_valid)
        out = np.full(max(f.N, 1), -1, np.int32)
        if kf_fv is None and f_fv is None:
            nm = _check(self._L.vsg_frame_search_by_bow_kf_f(
                self._h, _p(kv, _u8p), None, None, None, 0, f.handle, None, None, None, 0, float(np.float32(nnratio)),
                int(check_orientation), _p(out, _i32p)), "vsg_frame_search_by_bow_kf_f")
            return nm, out[:f.N]
        kn, ko, ki = (_i32(x) for x in kf_fv)
        fn, fo, fi = (_i32(x) for x in f_fv)
        nm = _check(self._L.vsg_frame_search_by_bow_kf_f(
            self._h, _p(kv, _u8p), _p(kn, _i32p), _p(ko, _i32p), _p(ki, _i32p), len(kf_fv[0]), f.handle, _p(fn, _i32p),
            _p(fo, _i32p), _p(fi, _i32p), len(f_fv[0]), float(np.float32(nnratio)), int(check_orientation),
            _p(out, _i32p)), "vsg_frame_search_by_bow_kf_f")
        return nm, out[:f.N]

    def SearchByBoW_KF_KF(self, valid1, fv1, kf2, valid2, fv2, nnratio, check_orientation):
        v1, v2 = _u8(valid1), _u8(valid2)
        out = np.full(max(self.N, 1), -1, np.int32)
        if fv1 is None and fv2 is None:  # both FeatureVectors resident
            nm = _check(self._L.vsg_frame_search_by_bow_kf_kf(
                self._h, _p(v1, _u8p), None, None, None, 0, kf2.handle, _p(v2, _u8p), None, None, None, 0,
                float(np.float32(nnratio)), int(check_orientation), _p(out, _i32p)), "vsg_frame_search_by_bow_kf_kf")
            return nm, out[:self.N]
        n1, o1, i1 = (_i32(x) for x in fv1)
        n2, o2, i2 = (_i32(x) for x in fv2)
        nm = _check(self._L.vsg_frame_search_by_bow_kf_kf(
            self._h, _p(v1, _u8p), _p(n1, _i32p), _p(o1, _i32p), _p(i1, _i32p), len(fv1[0]), kf2.handle, _p(v2, _u8p),
            _p(n2, _i32p), _p(o2, _i32p), _p(i2, _i32p), len(fv2[0]), float(np.float32(nnratio)),
            int(check_orientation), _p(out, _i32p)), "vsg_frame_search_by_bow_kf_kf")
        return nm, out[:self.N]

    def SearchForTriangulation(self, eligible1, fv1, kf2, eligible2, fv2, check_orientation, pair_ok=None, pair_off=None):
        """ORBmatcher::SearchForTriangulation (ORBmatcher.cc:902-1146), self = pKF1, kf2 = pKF2, both resident."""
        e1, e2 = _u8(eligible1), _u8(eligible2)
        n1, o1, i1 = (_i32(x) for x in fv1)
        n2, o2, i2 = (_i32(x) for x in fv2)
        ok = np.ascontiguousarray(pair_ok, np.uint32) if pair_ok is not None else None
        po = _i32(pair_off) if pair_ok is not None else None
        out = np.full(max(self.N, 1), -1, np.int32)
        nm = _check(self._L.vsg_frame_search_for_triangulation(
            self._h, _p(e1, _u8p), _p(n1, _i32p), _p(o1, _i32p), _p(i1, _i32p), len(fv1[0]), kf2.handle, _p(e2, _u8p),
            _p(n2, _i32p), _p(o2, _i32p), _p(i2, _i32p), len(fv2[0]), ok.ctypes.data if ok is not None else None,
            po.ctypes.data if po is not None else None, int(check_orientation), _p(out, _i32p)),
            "vsg_frame_search_for_triangulation")
        return nm, out[:self.N]

    def ComputeBoW(self, voc, levelsup=4):
        """Frame::ComputeBoW on the resident descriptors; same dict as ORBVocabulary.transform."""
        n = self.N
        cap = n + 1
        bi, bv = np.zeros(cap, np.int32), np.zeros(cap, np.float64)
        fn, fo, fi = np.zeros(cap, np.int32), np.zeros(cap + 1, np.int32), np.zeros(cap, np.int32)
        w_of, n_of, wt = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.float64)
        nb, nf = C.c_int32(), C.c_int32()
        f64p = C.POINTER(C.c_double)
        _check(self._L.vsg_frame_bow_transform(voc._h, self._h, int(levelsup), _p(bi, _i32p), _p(bv, f64p), cap,
                                               C.byref(nb), _p(fn, _i32p), _p(fo, _i32p), _p(fi, _i32p), cap,
                                               C.byref(nf), _p(w_of, _i32p), _p(n_of, _i32p), _p(wt, f64p)),
               "vsg_frame_bow_transform")
        return dict(bow_ids=bi[:nb.value].copy(), bow_vals=bv[:nb.value].copy(),
                    fv=(fn[:nf.value].copy(), fo[:nf.value + 1].copy(), fi[:fo[nf.value]].copy()),
                    word=w_of[:n].copy(), node=n_of[:n].copy(), weight=wt[:n].copy())


def SearchBySim3(kf1, kf2, q1, q2):
    """ORBmatcher::SearchBySim3 (ORBmatcher.cc:1448-1665) on two resident KeyFrames.  q1 / q2: dicts with idx, desc,
    u, v, radius, level of the points projected KF1 -> KF2 and KF2 -> KF1.  Returns (nFound, matches12)."""
    L = load_library()

    def unpack(q):
        d = _u8(q["desc"]).reshape(-1, 32)
        return (len(d), _i32(q["idx"]), d if len(d) else np.zeros((1, 32), np.uint8), _f32(q["u"]), _f32(q["v"]),
                _f32(q["radius"]), _i32(q["level"]))
    a, b = unpack(q1), unpack(q2)
    out = np.full(max(kf1.N, 1), -1, np.int32)
    nf = _check(L.vsg_frame_search_by_sim3(
        kf1.handle, kf2.handle, a[0], _p(a[1], _i32p), _p(a[2], _u8p), _p(a[3], _f32p), _p(a[4], _f32p), _p(a[5], _f32p),
        _p(a[6], _i32p), b[0], _p(b[1], _i32p), _p(b[2], _u8p), _p(b[3], _f32p), _p(b[4], _f32p), _p(b[5], _f32p),
        _p(b[6], _i32p), _p(out, _i32p)), "vsg_frame_search_by_sim3")
    return nf, out[:kf1.N]


def fuse_decide(query_mp, best_idx, best_dist, sim3_form, slot_mp, mp_obs, mp_bad):
    """The ordered replace-vs-add pass of Fuse (ORBmatcher.cc:1308-1327, :1429-1444) on flattened map-point ids.
    Returns (nFused, action, other_mp, slot_mp, mp_obs, mp_bad) -- the three state arrays are updated copies."""
    qm, bi, bd = _i32(query_mp), _i32(best_idx), _i32(best_dist)
    sm, ob, bad = _i32(slot_mp).copy(), _i32(mp_obs).copy(), _u8(mp_bad).copy()
    act, oth = np.zeros(max(len(qm), 1), np.int32), np.zeros(max(len(qm), 1), np.int32)
    nf = _check(load_library().vsg_fuse_decide(len(query_mp), _p(qm, _i32p), _p(bi, _i32p), _p(bd, _i32p),
                                               int(sim3_form), _p(sm, _i32p), len(slot_mp), _p(ob, _i32p),
                                               _p(bad, _u8p), len(mp_obs), _p(act, _i32p), _p(oth, _i32p)),
                "vsg_fuse_decide")
    return nf, act[:len(query_mp)], oth[:len(query_mp)], sm, ob, bad


def stereo_bow_search(ex_left, frame_l, ex_right, frame_r, fl, fr, mb, mbf, voc, levelsup=4, kf=None, kf_valid=None,
                      nnratio=0.7, check_orientation=True):
    """vsg_frame_stereo_bow_search: ComputeStereoMatches + ComputeBoW (+ SearchByBoW(kf, fl) when kf is given) of one stereo
    Frame in one enqueue and one wait.  Returns dict(u_right, depth, n_stereo, bow_ids, bow_vals, fv, n_match, match_f)."""
    L = load_library()
    n = fl.N
    cap = n + 1
    ur, dep = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
    bi, bv = np.zeros(cap, np.int32), np.zeros(cap, np.float64)
    fn, fo, fi = np.zeros(cap, np.int32), np.zeros(cap + 1, np.int32), np.zeros(cap, np.int32)
    match = np.full(cap, -1, np.int32)
    ns, nb, nf, nm = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    f64p = C.POINTER(C.c_double)
    kv = _u8(kf_valid) if kf is not None else None
    _check(L.vsg_frame_stereo_bow_search(
        ex_left.handle, int(frame_l), ex_right.handle, int(frame_r), fl.handle, fr.handle, float(mb), float(mbf), _p(ur, _f32p),
        _p(dep, _f32p), C.byref(ns), voc._h, int(levelsup), _p(bi, _i32p), _p(bv, f64p), cap, C.byref(nb), _p(fn, _i32p),
        _p(fo, _i32p), _p(fi, _i32p), cap, C.byref(nf), kf.handle if kf is not None else None,
        _p(kv, _u8p) if kf is not None else None, float(np.float32(nnratio)), int(check_orientation), _p(match, _i32p),
        C.byref(nm)), "vsg_frame_stereo_bow_search")
    return dict(u_right=ur[:n], depth=dep[:n], n_stereo=ns.value, bow_ids=bi[:nb.value].copy(), bow_vals=bv[:nb.value].copy(),
                fv=(fn[:nf.value].copy(), fo[:nf.value + 1].copy(), fi[:fo[nf.value]].copy()), n_match=nm.value,
                match_f=match[:n])


def ComputeStereoMatches_resident(ex_left, frame_l, ex_right, frame_r, fl, fr, mb, mbf):
    """Frame::ComputeStereoMatches on two resident feature sets (vsg_frame_stereo_matches)."""
    ur, dep = np.zeros(max(fl.N, 1), np.float32), np.zeros(max(fl.N, 1), np.float32)
    _check(load_library().vsg_frame_stereo_matches(ex_left.handle, int(frame_l), ex_right.handle, int(frame_r),
                                                   fl.handle, fr.handle, float(mb), float(mbf), _p(ur, _f32p),
                                                   _p(dep, _f32p)), "vsg_frame_stereo_matches")
    return ur[:fl.N], dep[:fl.N]
