// vsg_orb.hip -- host runtime + C ABI of the extractor (include/vsg_orb.h).
//
// A handle owns: the constructor tables, the geometry of the current image size, device buffers
// for `max_batch` frames (pyramid, blurred pyramid, candidates, selection, outputs), pinned host
// staging, two HIP streams (main chain + blur) and events.  One call enqueues
//   [H2D] -> resize x (L-1) -> { blur  ||  FAST -> octree -> slots } -> orient+desc -> [D2H]
// with the blur on its own stream so it overlaps the FAST/octree chain (both only read the pyramid).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/vsg_orb.h"
#include "vsg_common.h"
#include "vsg_geometry.h"
#include "vsg_kernels.h"

using namespace vsg;

static thread_local std::string g_err;
static void set_err(const std::string &s) { g_err = s; }

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      set_err(std::string(#expr) + ": " + hipGetErrorString(_e));                                  \
      return VSG_ERR_HIP;                                                                          \
    }                                                                                              \
  } while (0)

static const int8_t kPattern[1024] = {
#include "brief_pattern_data.inc"
};

static_assert(sizeof(KeyPointPOD) == 28 && sizeof(vsg_keypoint) == 28, "cv::KeyPoint layout");

enum { kStages = 7, kEv = 12, kMaxSub = 8 };

struct vsg_orb {
  ExtractorTables T;
  Geometry G;
  int device = 0, max_batch = 1;
  int rows = 0, cols = 0;  // geometry currently built for
  uint16_t taps[7] = {18, 34, 49, 55, 49, 34, 18};
  int gray_coeffs[3] = {4899, 9617, 1868};  // [OCV] 4.2 R2Y, G2Y, B2Y
  int gray_shift = 14;                        // yuv_shift
  int last_frames = 0;
  // device
  FrameGeom *d_fg = nullptr;
  Short4 *d_tab = nullptr;
  CellDesc *d_cells = nullptr;
  int pyr_tiling = 0;                   // the tiling calibration found faster for a full batch of this geometry
  PyrTile *d_ptiles[kPyrTilings] = {};  // Geometry::pyr[i].tiles
  Short4 *d_ptab[kPyrTilings] = {};     // Geometry::pyr[i].tab
  uint8_t *d_in = nullptr;  // level-0 staging for host images / unaligned device images, pitch in_pitch
  int in_pitch = 0;
  Src0 last_src0 = {nullptr, 0, 0};
  int8_t *d_pattern = nullptr;
  uint8_t *d_pyr = nullptr, *d_blur = nullptr;
  uint32_t *d_cand = nullptr, *d_sel = nullptr;
  uint16_t *d_nodeof = nullptr;
  int *d_counts2 = nullptr;  // [2][B][kMaxLevels]: cand_count then sel_count
  int *d_flags = nullptr, *d_slots = nullptr;
  FrameHeader *d_hdr = nullptr;
  KeyPointPOD *d_kps = nullptr;
  uint8_t *d_desc = nullptr;
  int *d_out_counts = nullptr;
  // pinned host staging
  uint8_t *h_in = nullptr;
  KeyPointPOD *h_kps = nullptr;
  uint8_t *h_desc = nullptr;
  int *h_out_counts = nullptr;
  hipStream_t s_main = nullptr, s_blur = nullptr;
  hipEvent_t ev_pyr = nullptr, ev_blur = nullptr, ev_fork = nullptr;
  // sub-batch pipelining
  int nsub = 0;  // sub-batches per call; 0 = auto
  bool serialize = false;  // every kernel on one stream (per-kernel timing without interference)
  hipStream_t sub_s[kMaxSub] = {}, sub_b[kMaxSub] = {};
  hipEvent_t sub_ev_pyr[kMaxSub] = {}, sub_ev_blur[kMaxSub] = {}, sub_ev_done[kMaxSub] = {};
  // timing
  bool timing = false;
  hipEvent_t ev[kEv] = {};
  double acc_ms[kStages] = {};
  int acc_n = 0;
  bool ev_pending = false;
};

// dynamic LDS the fused pyramid may ask for (the launcher raises the 64 KB default limit; gfx950 has 160 KB)
constexpr int kPyrLdsLimit = 150000;

static void free_image_buffers(vsg_orb *h) {
  hipFree(h->d_fg), hipFree(h->d_tab), hipFree(h->d_cells), hipFree(h->d_in);
  for (int i = 0; i < kPyrTilings; i++) {
    hipFree(h->d_ptiles[i]), hipFree(h->d_ptab[i]);
    h->d_ptiles[i] = nullptr, h->d_ptab[i] = nullptr;
  }
  hipFree(h->d_pyr), hipFree(h->d_blur), hipFree(h->d_cand), hipFree(h->d_sel), hipFree(h->d_nodeof);
  hipFree(h->d_counts2), hipFree(h->d_flags), hipFree(h->d_slots), hipFree(h->d_hdr);
  hipFree(h->d_kps), hipFree(h->d_desc), hipFree(h->d_out_counts);
  hipHostFree(h->h_in), hipHostFree(h->h_kps), hipHostFree(h->h_desc), hipHostFree(h->h_out_counts);
  h->d_fg = nullptr, h->d_tab = nullptr, h->d_cells = nullptr, h->d_in = nullptr;
  h->last_src0 = {nullptr, 0, 0};
  h->d_pyr = h->d_blur = nullptr;
  h->d_cand = h->d_sel = nullptr;
  h->d_nodeof = nullptr;
  h->d_counts2 = h->d_flags = h->d_slots = nullptr;
  h->d_hdr = nullptr;
  h->d_kps = nullptr, h->d_desc = nullptr, h->d_out_counts = nullptr;
  h->h_in = nullptr, h->h_kps = nullptr, h->h_desc = nullptr, h->h_out_counts = nullptr;
  h->rows = h->cols = 0;
}

// (re)build geometry + buffers for an image size
static int ensure_geometry(vsg_orb *h, int rows, int cols) {
  if (h->rows == rows && h->cols == cols) return VSG_OK;
  Geometry G;
  int rc = build_geometry(G, h->T, rows, cols, 0, 0, h->taps);
  if (rc != 0) {
    set_err("image size / parameters not processable (see vsg_geometry.h build_geometry)");
    return VSG_ERR_UNSUPPORTED;
  }
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->s_main));
  HIP_TRY(hipStreamSynchronize(h->s_blur));
  free_image_buffers(h);
  h->G = G;
  const FrameGeom &fg = h->G.fg;
  const size_t B = (size_t)h->max_batch;
  HIP_TRY(hipMalloc(&h->d_fg, sizeof(FrameGeom)));
  HIP_TRY(hipMalloc(&h->d_tab, sizeof(Short4) * (h->G.resizeTab.size() + 1)));
  HIP_TRY(hipMalloc(&h->d_cells, sizeof(CellDesc) * h->G.cells.size()));
  HIP_TRY(hipMemcpy(h->d_fg, &fg, sizeof(FrameGeom), hipMemcpyHostToDevice));
  if (!h->G.resizeTab.empty())
    HIP_TRY(hipMemcpy(h->d_tab, h->G.resizeTab.data(), sizeof(Short4) * h->G.resizeTab.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->d_cells, h->G.cells.data(), sizeof(CellDesc) * h->G.cells.size(), hipMemcpyHostToDevice));
  for (int i = 0; i < kPyrTilings; i++) {
    const PyrTiling &PT = h->G.pyr[i];
    HIP_TRY(hipMalloc(&h->d_ptiles[i], sizeof(PyrTile) * (PT.tiles.size() + 1)));
    if (!PT.tiles.empty())
      HIP_TRY(hipMemcpy(h->d_ptiles[i], PT.tiles.data(), sizeof(PyrTile) * PT.tiles.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&h->d_ptab[i], sizeof(Short4) * (PT.tab.size() + 1)));
    if (!PT.tab.empty())
      HIP_TRY(hipMemcpy(h->d_ptab[i], PT.tab.data(), sizeof(Short4) * PT.tab.size(), hipMemcpyHostToDevice));
  }
  HIP_TRY(hipMalloc(&h->d_pyr, B * fg.pyr_frame_bytes));
  HIP_TRY(hipMalloc(&h->d_blur, B * fg.pyr_frame_bytes));
  HIP_TRY(hipMemset(h->d_pyr, 0, B * fg.pyr_frame_bytes));
  HIP_TRY(hipMemset(h->d_blur, 0, B * fg.pyr_frame_bytes));
  HIP_TRY(hipMalloc(&h->d_cand, B * fg.cand_frame * sizeof(uint32_t)));
  HIP_TRY(hipMalloc(&h->d_nodeof, B * fg.cand_frame * sizeof(uint16_t)));
  HIP_TRY(hipMalloc(&h->d_sel, B * fg.sel_frame * sizeof(uint32_t)));
  HIP_TRY(hipMalloc(&h->d_counts2, 2 * B * kMaxLevels * sizeof(int)));
  HIP_TRY(hipMalloc(&h->d_flags, B * fg.out_cap * sizeof(int)));
  HIP_TRY(hipMalloc(&h->d_slots, B * fg.out_cap * sizeof(int)));
  HIP_TRY(hipMalloc(&h->d_hdr, B * sizeof(FrameHeader)));
  HIP_TRY(hipMalloc(&h->d_kps, B * fg.out_cap * sizeof(KeyPointPOD)));
  HIP_TRY(hipMalloc(&h->d_desc, B * fg.out_cap * 32));
  HIP_TRY(hipMalloc(&h->d_out_counts, B * 2 * sizeof(int)));
  h->in_pitch = (cols + 3) & ~3;
  HIP_TRY(hipMalloc(&h->d_in, B * (size_t)rows * h->in_pitch + 64));
  HIP_TRY(hipHostMalloc(&h->h_in, B * (size_t)rows * h->in_pitch));
  HIP_TRY(hipHostMalloc(&h->h_kps, B * fg.out_cap * sizeof(KeyPointPOD)));
  HIP_TRY(hipHostMalloc(&h->h_desc, B * fg.out_cap * 32));
  HIP_TRY(hipHostMalloc(&h->h_out_counts, B * 2 * sizeof(int)));
  h->rows = rows;
  h->cols = cols;
  // Which pyramid tiling is faster depends on how the top level happens to divide (measured: 36 wins at 640x480,
  // 752x480 and 1920x1080, 32 at 1280x720), so both are timed once on a full batch of this geometry (the duration
  // does not depend on the pixel values; the staging buffer's content is used as it is).
  h->pyr_tiling = 0;
  {
    const PyrTiling &P0 = h->G.pyr[0], &P1 = h->G.pyr[1];
    const int nf = h->max_batch;
    if (fg.nlevels > 1 && P0.ok && P1.ok && P0.lds_bytes() <= kPyrLdsLimit && P1.lds_bytes() <= 50 * 1024 &&
        (long long)P1.tiles.size() * nf >= 2048) {
      const Src0 s0 = {h->d_in, (size_t)rows * h->in_pitch, h->in_pitch};
      float best = 0.f;
      for (int ti = 0; ti < kPyrTilings; ti++) {
        const PyrTiling &PT = h->G.pyr[ti];
        float ms = 0.f;
        for (int rep = 0; rep < 2; rep++) {  // first launch warms up, second is timed
          HIP_TRY(hipEventRecord(h->ev[0], h->s_main));
          launch_pyramid(h->s_main, h->d_pyr, h->d_fg, h->d_ptab[ti], s0, h->d_ptiles[ti], (int)PT.tiles.size(), PT.ldsA,
                         PT.ldsB, PT.tabMax, nf);
          HIP_TRY(hipEventRecord(h->ev[1], h->s_main));
          HIP_TRY(hipEventSynchronize(h->ev[1]));
          HIP_TRY(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
        }
        if (ti == 0 || ms < best) {
          best = ms;
          h->pyr_tiling = ti;
        }
      }
    }
  }
  return VSG_OK;
}

static void harvest_timing(vsg_orb *h) {
  if (!h->ev_pending) return;
  h->ev_pending = false;
  if (hipEventSynchronize(h->ev[5]) != hipSuccess || hipEventSynchronize(h->ev[7]) != hipSuccess) return;
  float ms;
  // main chain: 0 start, 1 pyramid end, 8 fast begin, 2 fast end, 10 octree begin, 3 octree end, 4 slots end,
  // 9 orient begin, 5 orient end; blur (own stream unless serialised, then between FAST and the octree): 6 begin, 7 end
  const int pairs[kStages][2] = {{0, 1}, {8, 2}, {10, 3}, {6, 7}, {3, 4}, {9, 5}, {0, 5}};
  for (int i = 0; i < kStages; i++)
    if (hipEventElapsedTime(&ms, h->ev[pairs[i][0]], h->ev[pairs[i][1]]) == hipSuccess) h->acc_ms[i] += ms;
  h->acc_n++;
}

// Enqueue the stage chain for frames [f0, f0 + nf) on stream `s` (blur on `sb`, joined before orient+desc).
// Every buffer is frame-strided, so a sub-batch is just offset base pointers.
static int enqueue_range(vsg_orb *h, const Src0 &src, int f0, int nf, int lap0, int lap1, KeyPointPOD *d_kps,
                         uint8_t *d_desc, int *d_counts, int capacity, hipStream_t s, hipStream_t sb,
                         hipEvent_t ev_pyr, hipEvent_t ev_blur, bool tm) {
  const FrameGeom &fg = h->G.fg;
  const size_t F = (size_t)f0;
  const Src0 s0 = {src.base + F * src.frame_stride, src.frame_stride, src.pitch};
  uint8_t *pyr = h->d_pyr + F * fg.pyr_frame_bytes, *blur = h->d_blur + F * fg.pyr_frame_bytes;
  uint32_t *cand = h->d_cand + F * fg.cand_frame, *sel = h->d_sel + F * fg.sel_frame;
  uint16_t *nodeof = h->d_nodeof + F * fg.cand_frame;
  int *cand_count = h->d_counts2 + F * kMaxLevels;
  int *sel_count = h->d_counts2 + ((size_t)h->max_batch + F) * kMaxLevels;
  int *flags = h->d_flags + F * fg.out_cap, *slots = h->d_slots + F * fg.out_cap;
  FrameHeader *hdr = h->d_hdr + F;
  if (tm) HIP_TRY(hipEventRecord(h->ev[0], s));
  static const bool per_level = getenv("VSG_PYR_PER_LEVEL") != nullptr;  // A/B switch: 7 chained launches
  // tiling: the coarser one when it still gives the chip enough workgroups and leaves three of them per CU
  int ti = -1;
  if (!per_level && fg.nlevels > 1) {
    const PyrTiling &P0 = h->G.pyr[0], &P1 = h->G.pyr[1];
    static const int forced = getenv("VSG_PYR_TILING") ? atoi(getenv("VSG_PYR_TILING")) : -1;  // A/B switch
    if (forced == 0 || forced == 1)
      ti = h->G.pyr[forced].ok ? forced : -1;
    else if (h->pyr_tiling == 1 && (long long)P1.tiles.size() * nf >= 2048)  // calibrated choice, large launches only
      ti = 1;
    else if (P0.ok && P0.lds_bytes() <= kPyrLdsLimit)
      ti = 0;
  }
  if (ti < 0) {
    for (int l = 1; l < fg.nlevels; l++) launch_resize(s, pyr, h->d_fg, h->d_tab, s0, fg, l, nf);
  } else {
    const PyrTiling &PT = h->G.pyr[ti];
    launch_pyramid(s, pyr, h->d_fg, h->d_ptab[ti], s0, h->d_ptiles[ti], (int)PT.tiles.size(), PT.ldsA, PT.ldsB,
                   PT.tabMax, nf);
  }
  if (tm) HIP_TRY(hipEventRecord(h->ev[1], s));
  // The blur only needs the pyramid and runs on its own stream.  It is released AFTER FAST, beside the
  // latency-bound octree (+ slots): FAST and the blur are both issue-bound, so running them side by side only
  // shares the CUs, while the octree leaves most issue slots free.  Measured on MI355X (C2, one batch of 256):
  // 254 k frames/s against 250 k with the blur released right after the pyramid (VSG_BLUR_EARLY=1).
  static const bool blur_early = getenv("VSG_BLUR_EARLY") != nullptr && getenv("VSG_BLUR_LATE") == nullptr;
  if (!blur_early) {
    if (tm) HIP_TRY(hipEventRecord(h->ev[8], s));
    launch_fast(s, pyr, h->d_fg, h->d_cells, s0, cand, cand_count, fg, h->G.fastMaxVh, h->G.fastMaxVw, h->G.fastMaxArea, nf);
    if (tm) HIP_TRY(hipEventRecord(h->ev[2], s));
  }
  HIP_TRY(hipEventRecord(ev_pyr, s));
  HIP_TRY(hipStreamWaitEvent(sb, ev_pyr, 0));
  if (tm) HIP_TRY(hipEventRecord(h->ev[6], sb));
  launch_blur(sb, pyr, blur, h->d_fg, s0, fg, nf);
  if (tm) HIP_TRY(hipEventRecord(h->ev[7], sb));
  HIP_TRY(hipEventRecord(ev_blur, sb));
  if (blur_early) {
    if (tm) HIP_TRY(hipEventRecord(h->ev[8], s));
    launch_fast(s, pyr, h->d_fg, h->d_cells, s0, cand, cand_count, fg, h->G.fastMaxVh, h->G.fastMaxVw, h->G.fastMaxArea, nf);
    if (tm) HIP_TRY(hipEventRecord(h->ev[2], s));
  }
  if (tm) HIP_TRY(hipEventRecord(h->ev[10], s));
  launch_octree(s, h->d_fg, cand, cand_count, nodeof, sel, sel_count, fg, h->G.maxQuota, nf);
  if (tm) HIP_TRY(hipEventRecord(h->ev[3], s));
  launch_slots(s, h->d_fg, sel, sel_count, flags, slots, hdr, lap0, lap1, nf);
  if (tm) HIP_TRY(hipEventRecord(h->ev[4], s));
  HIP_TRY(hipStreamWaitEvent(s, ev_blur, 0));
  if (tm) HIP_TRY(hipEventRecord(h->ev[9], s));
  launch_orient_desc(s, pyr, blur, h->d_fg, s0, sel, slots, hdr, h->d_pattern, d_kps + F * capacity,
                     d_desc + F * capacity * 32, d_counts + F * 2, capacity, fg, nf);
  if (tm) HIP_TRY(hipEventRecord(h->ev[5], s));
  return VSG_OK;
}

// Enqueue the whole pipeline for `nframes` frames whose level-0 images are described by `s0`.
// The batch is cut into `h->nsub` sub-batches that run on their own stream pairs: the latency-bound stages of
// one sub-batch (octree, orient+desc, the small pyramid levels) then overlap the throughput-bound stages of
// another (FAST, blur).  VSG_NO_OVERLAP=1 serialises everything on `s` (used to time kernels in isolation).
static int enqueue_pipeline(vsg_orb *h, const Src0 &s0, int nframes, int lap0, int lap1, KeyPointPOD *d_kps,
                            uint8_t *d_desc, int *d_counts, int capacity, hipStream_t s) {
  h->last_src0 = s0;
  const bool no_overlap = h->serialize;
  // auto (VSG_SUBBATCH unset): ONE batch.  Sub-batches paid while the octree was a long latency-bound stage (+4 %,
  // then +1 %); with the current kernels -- and whole frames pinned to one XCD's L2 by the block remap -- two
  // sub-batches cost 2-4 % at 128-256 frames and nothing is gained at 512 (MI355X, C2-C4), so the cut stays a knob.
  int nsub = no_overlap ? 1 : h->nsub > 0 ? h->nsub : 1;
  if (nsub > nframes) nsub = nframes;
  const bool tm = h->timing && nsub == 1;
  if (tm) harvest_timing(h);
  launch_zero(s, h->d_counts2, 2 * h->max_batch * kMaxLevels);
  if (nsub == 1) {
    int rc = enqueue_range(h, s0, 0, nframes, lap0, lap1, d_kps, d_desc, d_counts, capacity, s,
                           no_overlap ? s : h->s_blur, h->ev_pyr, h->ev_blur, tm);
    if (rc != VSG_OK) return rc;
    if (tm) h->ev_pending = true;
  } else {
    HIP_TRY(hipEventRecord(h->ev_fork, s));
    const int per = (nframes + nsub - 1) / nsub;
    for (int j = 0; j < nsub; j++) {
      const int f0 = j * per, nf = nframes - f0 < per ? nframes - f0 : per;
      if (nf <= 0) break;
      HIP_TRY(hipStreamWaitEvent(h->sub_s[j], h->ev_fork, 0));
      int rc = enqueue_range(h, s0, f0, nf, lap0, lap1, d_kps, d_desc, d_counts, capacity, h->sub_s[j], h->sub_b[j],
                             h->sub_ev_pyr[j], h->sub_ev_blur[j], false);
      if (rc != VSG_OK) return rc;
      HIP_TRY(hipEventRecord(h->sub_ev_done[j], h->sub_s[j]));
      HIP_TRY(hipStreamWaitEvent(s, h->sub_ev_done[j], 0));
    }
  }
  HIP_TRY(hipGetLastError());
  h->last_frames = nframes;
  return VSG_OK;
}

extern "C" {

const char *vsg_last_error(void) { return g_err.c_str(); }

int vsg_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    set_err("no HIP device");
    return VSG_ERR_NO_DEVICE;
  }
  return n;
}

int vsg_orb_create(int nfeatures, float scale_factor, int nlevels, int ini_th_fast, int min_th_fast, int device,
                   int max_batch, vsg_orb **out) {
  if (!out || max_batch < 1) return VSG_ERR_INVALID;
  *out = nullptr;
  vsg_orb *h = new vsg_orb();
  if (!build_tables(h->T, nfeatures, scale_factor, nlevels, ini_th_fast, min_th_fast)) {
    delete h;
    set_err("invalid extractor parameters");
    return VSG_ERR_INVALID;
  }
  h->device = device;
  h->max_batch = max_batch;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    delete h;
    set_err("no usable HIP device (the extractor has no CPU fallback)");
    return VSG_ERR_NO_DEVICE;
  }
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->s_main, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&h->s_blur, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_pyr, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_blur, hipEventDisableTiming) != hipSuccess ||
      hipMalloc(&h->d_pattern, 1024) != hipSuccess ||
      hipMemcpy(h->d_pattern, kPattern, 1024, hipMemcpyHostToDevice) != hipSuccess) {
    set_err("HIP initialisation failed");
    vsg_orb_destroy(h);
    return VSG_ERR_NO_DEVICE;
  }
  for (int i = 0; i < kEv; i++) hipEventCreate(&h->ev[i]);
  hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
  for (int j = 0; j < kMaxSub; j++) {
    hipStreamCreateWithFlags(&h->sub_s[j], hipStreamNonBlocking);
    hipStreamCreateWithFlags(&h->sub_b[j], hipStreamNonBlocking);
    hipEventCreateWithFlags(&h->sub_ev_pyr[j], hipEventDisableTiming);
    hipEventCreateWithFlags(&h->sub_ev_blur[j], hipEventDisableTiming);
    hipEventCreateWithFlags(&h->sub_ev_done[j], hipEventDisableTiming);
  }
  {
    const char *e = getenv("VSG_SUBBATCH");
    int k = e ? atoi(e) : 0;  // 0 = auto
    h->nsub = k < 0 ? 0 : k > kMaxSub ? kMaxSub : k;
    h->serialize = getenv("VSG_NO_OVERLAP") != nullptr;
  }
  *out = h;
  return VSG_OK;
}

void vsg_orb_destroy(vsg_orb *h) {
  if (!h) return;
  hipSetDevice(h->device);
  if (h->s_main) hipStreamSynchronize(h->s_main);
  if (h->s_blur) hipStreamSynchronize(h->s_blur);
  free_image_buffers(h);
  hipFree(h->d_pattern);
  for (int i = 0; i < kEv; i++)
    if (h->ev[i]) hipEventDestroy(h->ev[i]);
  for (int j = 0; j < kMaxSub; j++) {
    if (h->sub_s[j]) hipStreamSynchronize(h->sub_s[j]), hipStreamDestroy(h->sub_s[j]);
    if (h->sub_b[j]) hipStreamSynchronize(h->sub_b[j]), hipStreamDestroy(h->sub_b[j]);
    if (h->sub_ev_pyr[j]) hipEventDestroy(h->sub_ev_pyr[j]);
    if (h->sub_ev_blur[j]) hipEventDestroy(h->sub_ev_blur[j]);
    if (h->sub_ev_done[j]) hipEventDestroy(h->sub_ev_done[j]);
  }
  if (h->ev_fork) hipEventDestroy(h->ev_fork);
  if (h->ev_pyr) hipEventDestroy(h->ev_pyr);
  if (h->ev_blur) hipEventDestroy(h->ev_blur);
  if (h->s_main) hipStreamDestroy(h->s_main);
  if (h->s_blur) hipStreamDestroy(h->s_blur);
  delete h;
}

int vsg_orb_get_tables(const vsg_orb *h, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2,
                       int *features_per_level, int *umax16) {
  if (!h) return VSG_ERR_INVALID;
  for (int i = 0; i < h->T.nlevels; i++) {
    if (scale) scale[i] = h->T.scale[i];
    if (inv_scale) inv_scale[i] = h->T.invScale[i];
    if (sigma2) sigma2[i] = h->T.sigma2[i];
    if (inv_sigma2) inv_sigma2[i] = h->T.invSigma2[i];
    if (features_per_level) features_per_level[i] = h->T.quota[i];
  }
  if (umax16)
    for (int i = 0; i < 16; i++) umax16[i] = h->T.umax[i];
  return h->T.nlevels;
}

int vsg_orb_set_blur_taps(vsg_orb *h, const uint16_t taps[7]) {
  if (!h || !taps) return VSG_ERR_INVALID;
  int tsum = 0;
  for (int k = 0; k < 7; k++) {
    if (taps[k] > 255) return VSG_ERR_UNSUPPORTED;  // 8-bit taps: the kernel uses v_dot4_u32_u8
    tsum += taps[k];
  }
  if (tsum > 257) return VSG_ERR_UNSUPPORTED;  // keeps the 8.8 row sums within 16 bits (no saturation anywhere)
  memcpy(h->taps, taps, sizeof(h->taps));
  if (h->rows) {  // refresh the device copy of the geometry
    for (int k = 0; k < 7; k++) h->G.fg.taps[k] = taps[k];
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->s_main));
    HIP_TRY(hipMemcpy(h->d_fg, &h->G.fg, sizeof(FrameGeom), hipMemcpyHostToDevice));
  }
  return VSG_OK;
}

int vsg_orb_capacity(vsg_orb *h, int rows, int cols) {
  if (!h) return VSG_ERR_INVALID;
  if (rows <= 0 || cols <= 0) return VSG_ERR_EMPTY_IMAGE;
  int rc = ensure_geometry(h, rows, cols);
  if (rc != VSG_OK) return rc;
  return h->G.fg.out_cap;
}

}  // extern "C"

// D2H of the handle's own output buffers + copy into the caller's [nframes][capacity] arrays
static int fetch_outputs(vsg_orb *h, int nframes, vsg_keypoint *kps, uint8_t *desc, int capacity, int *n,
                         int *mono_index, hipStream_t s) {
  const FrameGeom &fg = h->G.fg;
  HIP_TRY(hipMemcpyAsync(h->h_out_counts, h->d_out_counts, (size_t)nframes * 2 * sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(h->h_kps, h->d_kps, (size_t)nframes * fg.out_cap * sizeof(KeyPointPOD), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(h->h_desc, h->d_desc, (size_t)nframes * fg.out_cap * 32, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  int status = VSG_OK;
  for (int f = 0; f < nframes; f++) {
    const int nf = h->h_out_counts[2 * f];
    n[f] = nf;
    mono_index[f] = h->h_out_counts[2 * f + 1];
    if (nf > capacity) {
      set_err("caller capacity too small for the extracted keypoints");
      status = VSG_ERR_CAPACITY;
      continue;
    }
    if (nf > 0 && kps) memcpy(kps + (size_t)f * capacity, h->h_kps + (size_t)f * fg.out_cap, (size_t)nf * sizeof(KeyPointPOD));
    if (nf > 0 && desc) memcpy(desc + (size_t)f * capacity * 32, h->h_desc + (size_t)f * fg.out_cap * 32, (size_t)nf * 32);
  }
  return status;
}

extern "C" {

int vsg_orb_extract_batch_device(vsg_orb *h, const uint8_t *d_gray, int nframes, size_t frame_stride, int rows,
                                 int cols, int stride, int lap0, int lap1, vsg_keypoint *d_kps, uint8_t *d_desc,
                                 int *d_counts, int capacity, void *stream) {
  if (!h || !d_kps || !d_desc || !d_counts || nframes < 1 || nframes > h->max_batch) return VSG_ERR_INVALID;
  if (!d_gray || rows <= 0 || cols <= 0) return VSG_ERR_EMPTY_IMAGE;
  HIP_TRY(hipSetDevice(h->device));
  int rc = ensure_geometry(h, rows, cols);
  if (rc != VSG_OK) return rc;
  const FrameGeom &fg = h->G.fg;
  if (capacity < fg.out_cap) {
    set_err("capacity below vsg_orb_capacity()");
    return VSG_ERR_CAPACITY;
  }
  hipStream_t s = stream ? (hipStream_t)stream : h->s_main;
  Src0 s0;
  if (((uintptr_t)d_gray & 3) == 0 && (stride & 3) == 0 && (frame_stride & 3) == 0) {
    s0 = {d_gray, frame_stride, stride};  // level 0 is read in place: no ingest copy
  } else {                                 // unaligned layout: restage (rare)
    for (int f = 0; f < nframes; f++)
      HIP_TRY(hipMemcpy2DAsync(h->d_in + (size_t)f * rows * h->in_pitch, h->in_pitch, d_gray + (size_t)f * frame_stride,
                               stride, cols, rows, hipMemcpyDeviceToDevice, s));
    s0 = {h->d_in, (size_t)rows * h->in_pitch, h->in_pitch};
  }
  return enqueue_pipeline(h, s0, nframes, lap0, lap1, (KeyPointPOD *)d_kps, d_desc, d_counts, capacity, s);
}

int vsg_orb_set_gray_coeffs(vsg_orb *h, const int coeffs[3], int shift) {
  if (!h || !coeffs || shift < 1 || shift > 20) return VSG_ERR_INVALID;
  if (coeffs[0] + coeffs[1] + coeffs[2] != (1 << shift)) return VSG_ERR_INVALID;  // CV_Assert in RGB2Gray<uchar>
  memcpy(h->gray_coeffs, coeffs, sizeof(h->gray_coeffs));
  h->gray_shift = shift;
  return VSG_OK;
}

int vsg_orb_extract_batch_device_color(vsg_orb *h, const uint8_t *d_img, int channels, int rgb_order, int nframes,
                                       size_t frame_stride, int rows, int cols, int stride, int lap0, int lap1,
                                       vsg_keypoint *d_kps, uint8_t *d_desc, int *d_counts, int capacity,
                                       void *stream) {
  if (!h || !d_kps || !d_desc || !d_counts || nframes < 1 || nframes > h->max_batch) return VSG_ERR_INVALID;
  if (channels != 3 && channels != 4) return VSG_ERR_INVALID;
  if (!d_img || rows <= 0 || cols <= 0) return VSG_ERR_EMPTY_IMAGE;
  HIP_TRY(hipSetDevice(h->device));
  int rc = ensure_geometry(h, rows, cols);
  if (rc != VSG_OK) return rc;
  if (capacity < h->G.fg.out_cap) return VSG_ERR_CAPACITY;
  hipStream_t s = stream ? (hipStream_t)stream : h->s_main;
  // cvtColor of Tracking::GrabImage* straight into the gray level-0 staging buffer
  launch_cvt_gray(s, d_img, frame_stride, stride, channels, rgb_order, rows, cols, h->d_in,
                  (size_t)rows * h->in_pitch, h->in_pitch, h->gray_coeffs, h->gray_shift, nframes);
  const Src0 s0 = {h->d_in, (size_t)rows * h->in_pitch, h->in_pitch};
  return enqueue_pipeline(h, s0, nframes, lap0, lap1, (KeyPointPOD *)d_kps, d_desc, d_counts, capacity, s);
}

int vsg_orb_extract_batch(vsg_orb *h, const uint8_t *gray, int nframes, size_t frame_stride, int rows, int cols,
                          int stride, int lap0, int lap1, vsg_keypoint *kps, uint8_t *desc, int capacity, int *n,
                          int *mono_index) {
  if (!h || nframes < 1 || nframes > h->max_batch || !n || !mono_index) return VSG_ERR_INVALID;
  if (!gray || rows <= 0 || cols <= 0) return VSG_ERR_EMPTY_IMAGE;
  HIP_TRY(hipSetDevice(h->device));
  int rc = ensure_geometry(h, rows, cols);
  if (rc != VSG_OK) return rc;
  const FrameGeom &fg = h->G.fg;
  hipStream_t s = h->s_main;
  const int ip = h->in_pitch;
  for (int f = 0; f < nframes; f++) {
    uint8_t *dst = h->h_in + (size_t)f * rows * ip;
    const uint8_t *src = gray + (size_t)f * frame_stride;
    if (stride == ip)
      memcpy(dst, src, (size_t)rows * ip);
    else
      for (int y = 0; y < rows; y++) memcpy(dst + (size_t)y * ip, src + (size_t)y * stride, (size_t)cols);
  }
  HIP_TRY(hipMemcpyAsync(h->d_in, h->h_in, (size_t)nframes * rows * ip, hipMemcpyHostToDevice, s));  // one DMA
  const Src0 s0 = {h->d_in, (size_t)rows * ip, ip};
  rc = enqueue_pipeline(h, s0, nframes, lap0, lap1, h->d_kps, h->d_desc, h->d_out_counts, fg.out_cap, s);
  if (rc != VSG_OK) return rc;
  return fetch_outputs(h, nframes, kps, desc, capacity, n, mono_index, s);
}

int vsg_orb_extract_batch_color(vsg_orb *h, const uint8_t *img, int channels, int rgb_order, int nframes,
                                size_t frame_stride, int rows, int cols, int stride, int lap0, int lap1,
                                vsg_keypoint *kps, uint8_t *desc, int capacity, int *n, int *mono_index) {
  if (!h || nframes < 1 || nframes > h->max_batch || !n || !mono_index) return VSG_ERR_INVALID;
  if (channels != 3 && channels != 4) return VSG_ERR_INVALID;
  if (!img || rows <= 0 || cols <= 0) return VSG_ERR_EMPTY_IMAGE;
  HIP_TRY(hipSetDevice(h->device));
  int rc = ensure_geometry(h, rows, cols);
  if (rc != VSG_OK) return rc;
  const FrameGeom &fg = h->G.fg;
  hipStream_t s = h->s_main;
  const size_t row_bytes = (size_t)cols * channels, frame_bytes = row_bytes * rows;
  uint8_t *d_color = nullptr;
  HIP_TRY(hipMalloc(&d_color, frame_bytes * nframes));
  hipError_t e = hipSuccess;
  for (int f = 0; f < nframes && e == hipSuccess; f++)
    e = hipMemcpy2DAsync(d_color + f * frame_bytes, row_bytes, img + (size_t)f * frame_stride, stride, row_bytes, rows,
                         hipMemcpyHostToDevice, s);
  if (e == hipSuccess) {
    rc = vsg_orb_extract_batch_device_color(h, d_color, channels, rgb_order, nframes, frame_bytes, rows, cols,
                                            (int)row_bytes, lap0, lap1, (vsg_keypoint *)h->d_kps, h->d_desc,
                                            h->d_out_counts, fg.out_cap, s);
    if (rc == VSG_OK) rc = fetch_outputs(h, nframes, kps, desc, capacity, n, mono_index, s);
  }
  hipStreamSynchronize(s);
  hipFree(d_color);
  HIP_TRY(e);
  return rc;
}

int vsg_orb_extract(vsg_orb *h, const uint8_t *gray, int rows, int cols, int stride, int lap0, int lap1,
                    vsg_keypoint *kps, uint8_t *desc, int capacity, int *n) {
  if (n) *n = 0;
  if (!h) return VSG_ERR_INVALID;
  if (!gray || rows <= 0 || cols <= 0) return VSG_ERR_EMPTY_IMAGE;
  int nn = 0, mono = 0;
  int rc = vsg_orb_extract_batch(h, gray, 1, 0, rows, cols, stride, lap0, lap1, kps, desc, capacity, &nn, &mono);
  if (n) *n = nn;
  if (rc != VSG_OK) return rc;
  return mono;
}

int vsg_orb_level_size(vsg_orb *h, int level, int *w, int *ht) {
  if (!h || !h->rows || level < 0 || level >= h->T.nlevels) return VSG_ERR_INVALID;
  if (w) *w = h->G.fg.lv[level].w;
  if (ht) *ht = h->G.fg.lv[level].h;
  return VSG_OK;
}

static int copy_level(vsg_orb *h, const uint8_t *base, int frame, int level, int with_border, uint8_t *dst,
                      int dst_stride) {
  if (!h || !h->rows || !dst || level < 0 || level >= h->T.nlevels || frame < 0 || frame >= h->max_batch)
    return VSG_ERR_INVALID;
  HIP_TRY(hipSetDevice(h->device));
  const FrameGeom &fg = h->G.fg;
  const LevelGeom &L = fg.lv[level];
  const uint8_t *img = base + (size_t)frame * fg.pyr_frame_bytes + L.img_off;
  int ipitch = L.pitch;
  if (level == 0 && base == h->d_pyr) {  // level 0 lives in the caller's / staging buffer of the last call
    if (!h->last_src0.base) return VSG_ERR_INVALID;
    img = h->last_src0.base + (size_t)frame * h->last_src0.frame_stride;
    ipitch = h->last_src0.pitch;
  }
  HIP_TRY(hipStreamSynchronize(h->s_main));
  HIP_TRY(hipStreamSynchronize(h->s_blur));
  if (!with_border) {
    HIP_TRY(hipMemcpy2D(dst, dst_stride, img, ipitch, L.w, L.h, hipMemcpyDeviceToHost));
    return VSG_OK;
  }
  const int b = kEdgeThreshold, bw = L.w + 2 * b, bh = L.h + 2 * b;
  uint8_t *tmp = nullptr;
  HIP_TRY(hipMalloc(&tmp, (size_t)bw * bh));
  launch_border_copy(h->s_main, img, L.w, L.h, ipitch, tmp, bw, b);
  hipError_t e = hipStreamSynchronize(h->s_main);
  if (e == hipSuccess) e = hipMemcpy2D(dst, dst_stride, tmp, bw, bw, bh, hipMemcpyDeviceToHost);
  hipFree(tmp);
  HIP_TRY(e);
  return VSG_OK;
}

int vsg_orb_copy_pyramid_level(vsg_orb *h, int frame, int level, int with_border, uint8_t *dst, int dst_stride) {
  return copy_level(h, h ? h->d_pyr : nullptr, frame, level, with_border, dst, dst_stride);
}
int vsg_orb_copy_blurred_level(vsg_orb *h, int frame, int level, uint8_t *dst, int dst_stride) {
  return copy_level(h, h ? h->d_blur : nullptr, frame, level, 0, dst, dst_stride);
}

static int copy_list(vsg_orb *h, const uint32_t *list, size_t frame_elems, int off, int lcap, const int *counts,
                     int frame, int level, uint32_t *dst, int cap) {
  if (!h || !h->rows || level < 0 || level >= h->T.nlevels || frame < 0 || frame >= h->max_batch) return VSG_ERR_INVALID;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->s_main));
  int n = 0;
  HIP_TRY(hipMemcpy(&n, counts + frame * kMaxLevels + level, sizeof(int), hipMemcpyDeviceToHost));
  if (n > lcap) n = lcap;
  const int m = n < cap ? n : cap;
  if (m > 0 && dst) HIP_TRY(hipMemcpy(dst, list + (size_t)frame * frame_elems + off, (size_t)m * 4, hipMemcpyDeviceToHost));
  return n;
}

int vsg_orb_copy_candidates(vsg_orb *h, int frame, int level, uint32_t *dst, int cap) {
  if (!h || !h->rows || level < 0 || level >= h->T.nlevels) return VSG_ERR_INVALID;
  const LevelGeom &L = h->G.fg.lv[level];
  return copy_list(h, h->d_cand, h->G.fg.cand_frame, L.cand_off, L.cand_cap, h->d_counts2, frame, level, dst, cap);
}
int vsg_orb_copy_selected(vsg_orb *h, int frame, int level, uint32_t *dst, int cap) {
  if (!h || !h->rows || level < 0 || level >= h->T.nlevels) return VSG_ERR_INVALID;
  const LevelGeom &L = h->G.fg.lv[level];
  return copy_list(h, h->d_sel, h->G.fg.sel_frame, L.sel_off, L.sel_cap,
                   h->d_counts2 + (size_t)h->max_batch * kMaxLevels, frame, level, dst, cap);
}

static int pyr_view(vsg_orb *h, int frame, PyrView &v) {
  if (!h || !h->rows || frame < 0 || frame >= h->max_batch || !h->last_src0.base) return VSG_ERR_INVALID;
  const FrameGeom &fg = h->G.fg;
  memset(&v, 0, sizeof(v));
  for (int l = 0; l < fg.nlevels; l++) {
    v.w[l] = fg.lv[l].w;
    v.h[l] = fg.lv[l].h;
    if (l == 0) {
      v.lvl[0] = h->last_src0.base + (size_t)frame * h->last_src0.frame_stride;
      v.pitch[0] = h->last_src0.pitch;
    } else {
      v.lvl[l] = h->d_pyr + (size_t)frame * fg.pyr_frame_bytes + fg.lv[l].img_off;
      v.pitch[l] = fg.lv[l].pitch;
    }
  }
  return VSG_OK;
}

int vsg_stereo_matches(vsg_orb *hl, int frame_l, vsg_orb *hr, int frame_r, const vsg_keypoint *kps_l,
                       const uint8_t *desc_l, int n_l, const vsg_keypoint *kps_r, const uint8_t *desc_r, int n_r,
                       float mb, float mbf, float *u_right, float *depth) {
  if (!hl || !hr || !u_right || !depth || n_l < 0 || n_r < 0 || hl->device != hr->device) return VSG_ERR_INVALID;
  for (int i = 0; i < n_l; i++) u_right[i] = -1.0f, depth[i] = -1.0f;
  if (n_l == 0 || n_r == 0) return 0;
  PyrView pl, pr;
  int rc = pyr_view(hl, frame_l, pl);
  if (rc == VSG_OK) rc = pyr_view(hr, frame_r, pr);
  if (rc != VSG_OK) return rc;
  HIP_TRY(hipSetDevice(hl->device));
  HIP_TRY(hipStreamSynchronize(hl->s_main));
  HIP_TRY(hipStreamSynchronize(hr->s_main));
  KeyPointPOD *dkl = nullptr, *dkr = nullptr;
  uint8_t *ddl = nullptr, *ddr = nullptr;
  float *du = nullptr, *dd = nullptr;
  int *ds = nullptr;
  hipError_t e = hipMalloc(&dkl, sizeof(KeyPointPOD) * n_l);
  if (e == hipSuccess) e = hipMalloc(&dkr, sizeof(KeyPointPOD) * n_r);
  if (e == hipSuccess) e = hipMalloc(&ddl, 32 * (size_t)n_l);
  if (e == hipSuccess) e = hipMalloc(&ddr, 32 * (size_t)n_r);
  if (e == hipSuccess) e = hipMalloc(&du, 4 * (size_t)n_l);
  if (e == hipSuccess) e = hipMalloc(&dd, 4 * (size_t)n_l);
  if (e == hipSuccess) e = hipMalloc(&ds, 4 * (size_t)n_l);
  if (e == hipSuccess) e = hipMemcpy(dkl, kps_l, sizeof(KeyPointPOD) * n_l, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dkr, kps_r, sizeof(KeyPointPOD) * n_r, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(ddl, desc_l, 32 * (size_t)n_l, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(ddr, desc_r, 32 * (size_t)n_r, hipMemcpyHostToDevice);
  std::vector<int> sad((size_t)n_l, -1);
  if (e == hipSuccess) {
    launch_stereo(hl->s_main, pl, pr, mb, mbf, hl->T.scale.data(), hl->T.invScale.data(), hl->T.nlevels, dkl, ddl, n_l,
                  dkr, ddr, n_r, du, dd, ds);
    e = hipStreamSynchronize(hl->s_main);
  }
  if (e == hipSuccess) e = hipMemcpy(u_right, du, 4 * (size_t)n_l, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(depth, dd, 4 * (size_t)n_l, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(sad.data(), ds, 4 * (size_t)n_l, hipMemcpyDeviceToHost);
  hipFree(dkl), hipFree(dkr), hipFree(ddl), hipFree(ddr), hipFree(du), hipFree(dd), hipFree(ds);
  HIP_TRY(e);
  // median-based outlier cut (Frame.cc:1113-1126)
  std::vector<std::pair<int, int>> vDistIdx;
  for (int i = 0; i < n_l; i++)
    if (sad[i] >= 0) vDistIdx.push_back(std::pair<int, int>(sad[i], i));
  if (vDistIdx.empty()) return 0;
  std::sort(vDistIdx.begin(), vDistIdx.end());
  const float median = (float)vDistIdx[vDistIdx.size() / 2].first;
  const float thDist = 1.5f * 1.4f * median;
  int kept = (int)vDistIdx.size();
  for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
    if ((float)vDistIdx[i].first < thDist) break;
    u_right[vDistIdx[i].second] = -1;
    depth[vDistIdx[i].second] = -1;
    kept--;
  }
  return kept;
}

int vsg_orb_set_serialize(vsg_orb *h, int serialize) {
  if (!h) return VSG_ERR_INVALID;
  h->serialize = serialize != 0;
  return VSG_OK;
}

int vsg_orb_enable_timing(vsg_orb *h, int enable) {
  if (!h) return VSG_ERR_INVALID;
  h->timing = enable != 0;
  h->ev_pending = false;
  memset(h->acc_ms, 0, sizeof(h->acc_ms));
  h->acc_n = 0;
  return VSG_OK;
}

int vsg_debug_device_sort(int device, uint64_t *items, int n) {
  if (!items || n < 0 || n > 2048) return VSG_ERR_INVALID;
  if (vsg_device_count() <= device || device < 0) return VSG_ERR_NO_DEVICE;
  if (n == 0) return VSG_OK;
  HIP_TRY(hipSetDevice(device));
  uint64_t *d = nullptr;
  HIP_TRY(hipMalloc(&d, sizeof(uint64_t) * n));
  hipError_t e = hipMemcpy(d, items, sizeof(uint64_t) * n, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    launch_debug_sort(nullptr, d, n);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(items, d, sizeof(uint64_t) * n, hipMemcpyDeviceToHost);
  hipFree(d);
  if (e != hipSuccess) {
    set_err(std::string("vsg_debug_device_sort: ") + hipGetErrorString(e));
    return VSG_ERR_HIP;
  }
  return VSG_OK;
}

int vsg_orb_get_timing(vsg_orb *h, float *ms_out, int cap) {
  if (!h) return VSG_ERR_INVALID;
  harvest_timing(h);
  for (int i = 0; i < kStages && i < cap; i++) ms_out[i] = h->acc_n ? (float)(h->acc_ms[i] / h->acc_n) : 0.f;
  return kStages;
}

}  // extern "C"
