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
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <chrono>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/vsg_orb.h"
#include "../../include/vsg_orb_debug.h"
#include "vsg_common.h"
#include "vsg_ctx.h"
#include "vsg_frame_int.h"
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

enum { kStages = 7, kEv = 12, kMaxSub = 8, kSlots = 3 };

// One pipeline slot of the host API: level-0 staging and output records of ONE batch in flight, on the device and in
// pinned host memory, plus the events that chain  H2D -> kernels -> export  across the three streams.
struct Slot {
  uint8_t *d_in = nullptr, *h_in = nullptr;  // [B][rows][in_pitch]
  uint8_t *h_in_dev = nullptr;               // device alias of h_in (the ingest kernel reads it over PCIe)
  KeyPointPOD *d_kps = nullptr, *h_kps = nullptr;
  uint8_t *d_desc = nullptr, *h_desc = nullptr;
  int *d_counts = nullptr, *h_counts = nullptr;
  hipEvent_t ev_in = nullptr, ev_done = nullptr, ev_out = nullptr;
  bool allocated = false, busy = false;
  int ticket = -1, nframes = 0;
  // where vsg_orb_wait delivers (the caller's arrays); direct = the export kernel already wrote them
  vsg_keypoint *out_kps = nullptr;
  uint8_t *out_desc = nullptr;
  int out_cap = 0;
  bool direct = false;
};

// Helper threads for the pageable-input path of vsg_orb_submit_batch: the frames of a batch are copied into the slot's
// pinned staging by the calling thread AND three sleeping helpers (one core copies ~20 GB/s, a 64-frame C2 batch is
// 20 MB; the copy was the whole cost of that path: 63 k -> 113 k frames/s on the GPU box, flat beyond 3-4 helpers).  The helpers are created at the first pageable batch of a handle,
// sleep on a condition variable between batches and are joined by vsg_orb_destroy.
struct StagePool {
  struct Job {
    const uint8_t *src = nullptr;
    uint8_t *dst = nullptr;
    size_t frame_stride = 0, fbytes = 0;
    int stride = 0, ip = 0, rows = 0, cols = 0, nframes = 0;
  } job;
  std::vector<std::thread> workers;
  std::mutex m;
  std::condition_variable cv_work, cv_done;
  std::atomic<int> next{0};
  int generation = 0, pending = 0;
  bool stop = false;

  static void copy_frame(const Job &j, int f) {
    uint8_t *dst = j.dst + (size_t)f * j.fbytes;
    const uint8_t *src = j.src + (size_t)f * j.frame_stride;
    if (j.stride == j.ip)
      memcpy(dst, src, j.fbytes);
    else
      for (int y = 0; y < j.rows; y++) memcpy(dst + (size_t)y * j.ip, src + (size_t)y * j.stride, (size_t)j.cols);
  }
  void drain() {
    for (int f = next.fetch_add(1); f < job.nframes; f = next.fetch_add(1)) copy_frame(job, f);
  }
  void worker() {
    int seen = 0;
    std::unique_lock<std::mutex> lk(m);
    for (;;) {
      cv_work.wait(lk, [&] { return stop || generation != seen; });
      if (stop) return;
      seen = generation;
      lk.unlock();
      drain();
      lk.lock();
      if (--pending == 0) cv_done.notify_one();
    }
  }
  explicit StagePool(int n) {
    for (int i = 0; i < n; i++) workers.emplace_back([this] { worker(); });
  }
  ~StagePool() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv_work.notify_all();
    for (auto &t : workers) t.join();
  }
  // copies every frame of the job; returns when all of them are staged
  void run(const Job &j) {
    {
      std::lock_guard<std::mutex> lk(m);
      job = j;
      next.store(0);
      pending = (int)workers.size();
      generation++;
    }
    cv_work.notify_all();
    drain();
    std::unique_lock<std::mutex> lk(m);
    cv_done.wait(lk, [&] { return pending == 0; });
  }
};

// Latency mode of the blocking entry points: the stream work of one call -- ingest, the stage chain with its blur fork /
// join, the export -- captured once as a hipGraph and replayed with ONE hipGraphLaunch per operator() (the reference's
// call pattern is one frame per call: System::TrackRGBD -> Frame::ExtractORB, Frame.cc:555-563).  A graph is tied to
// everything its nodes hold by value: the slot, the frame count, the lapping area, the source (the slot's pinned
// staging or a pinned caller image) and the destination (the slot's pinned records or pinned caller arrays).
struct ChainKey {
  int slot = -1, nframes = 0, lap0 = 0, lap1 = 0, stride = 0, capacity = 0;
  size_t frame_stride = 0;
  const void *src = nullptr, *dk = nullptr, *dd = nullptr;
  bool operator==(const ChainKey &o) const {
    return slot == o.slot && nframes == o.nframes && lap0 == o.lap0 && lap1 == o.lap1 && stride == o.stride &&
           capacity == o.capacity && frame_stride == o.frame_stride && src == o.src && dk == o.dk && dd == o.dd;
  }
};
struct ChainGraph {
  ChainKey key;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  int seen = 0;  // eager runs with this key so far: the graph is captured on the second call
  long stamp = 0;
};
enum { kChainGraphs = 8 };

struct vsg_orb {
  ExtractorTables T;
  Geometry G;
  int device = 0, max_batch = 1;
  StagePool *pool = nullptr;  // pageable-input staging helpers (vsg_orb_submit_batch)
  // Set by the BLOCKING entry points around their submit: input copy, kernels and export all go to s_main.  The three
  // streams exist so that consecutive batches overlap; a call that waits for its own batch has nothing to overlap
  // with, and each cross-stream event wait costs ~15 us of idle GPU (0.175 -> ~0.15 ms per single-frame operator()).
  bool one_stream = false;
  bool direct_registered = false;  // vsg_orb_set_direct_registered: hipHostRegister-ed caller memory is used in place
  bool capturing = false;  // the calls below are being recorded into a ChainGraph (no timing events, no host waits)
  OutMirror mirror;        // set around an enqueue by the latency path: k_orient_desc also writes pinned host records
  ChainGraph chain[kChainGraphs];
  long chain_clock = 0, chain_launches = 0;
  int rows = 0, cols = 0;  // geometry currently built for
  uint16_t taps[7] = {18, 34, 49, 55, 49, 34, 18};
  int gray_coeffs[3] = {4899, 9617, 1868};  // [OCV] 4.2 R2Y, G2Y, B2Y
  int gray_shift = 14;                        // yuv_shift
  int last_frames = 0;
  // device
  FrameGeom *d_fg = nullptr;
  Short4 *d_tab = nullptr;
  CellDesc *d_cells = nullptr;
  FastCellRec *d_fast = nullptr;  // one record per cell for k_fast_cells (vsg_common.h)
  int *d_cell_count = nullptr;    // [max_batch][fg.total_cells] FAST survivors per cell (segmented candidate lists)
  uint32_t *d_cand2 = nullptr;    // [max_batch][fg.cand_frame] compacted candidates of levels too large for the octree's registers
  int pyr_tiling = 0;                   // the tiling calibration found faster for a full batch of this geometry
  int force_tiling = -1;                // vsg_orb_set_pyramid_tiling (tests: every launch form against the oracle)
  vsg_post_chain_fn post_fn = nullptr;  // vsg_orb_set_post_chain: enqueued behind the next submit's chain, then cleared
  void *post_ctx = nullptr;
  int cus = 256;                        // compute units of `device` (k_fast_cells' cells-per-workgroup choice)
  PyrTile *d_ptiles[kPyrTilings] = {};  // Geometry::pyr[i].tiles
  Short4 *d_ptab[kPyrTilings] = {};     // Geometry::pyr[i].tab
  uint8_t *d_in = nullptr;  // level-0 staging for unaligned / colour device images (device API), pitch in_pitch
  int in_pitch = 0;
  Slot slot[kSlots];
  int next_ticket = 0;
  hipStream_t s_h2d = nullptr, s_d2h = nullptr;
  // outputs of the last enqueue (any entry point): where they are and when they are complete
  const KeyPointPOD *last_kps = nullptr;
  const uint8_t *last_desc = nullptr;
  const int *last_counts = nullptr;
  int last_cap = 0;
  hipEvent_t ev_last = nullptr;    // recorded behind the last kernel of the last enqueue
  bool have_last = false;
  hipEvent_t ev_null_in = nullptr, ev_null_out = nullptr;  // ordering against the caller's NULL stream
  uint8_t *d_scratch = nullptr;  // grows: colour upload of the host colour entry, bordered level of copy_level
  size_t scratch_cap = 0;
  Src0 last_src0 = {nullptr, 0, 0};
  int8_t *d_pattern = nullptr;
  uint8_t *d_pyr = nullptr, *d_blur = nullptr;
  uint32_t *d_cand = nullptr, *d_sel = nullptr;
  uint16_t *d_nodeof = nullptr;
  int *d_counts2 = nullptr;  // [2][B][kMaxLevels]: cand_count then sel_count
  int *d_flags = nullptr;
  int4 *d_slots = nullptr;  // per output index: {selected candidate, output slot, level, 0} (k_slots -> k_orient_desc)
  FrameHeader *d_hdr = nullptr;
  hipStream_t s_main = nullptr, s_blur = nullptr;
  hipEvent_t ev_pyr = nullptr, ev_blur = nullptr, ev_fork = nullptr;
  // sub-batch pipelining
  int nsub = 0;  // sub-batches per call; 0 = auto
  bool serialize = false;  // every kernel on one stream (per-kernel timing without interference)
  hipStream_t sub_s[kMaxSub] = {}, sub_b[kMaxSub] = {};
  hipEvent_t sub_ev_pyr[kMaxSub] = {}, sub_ev_blur[kMaxSub] = {}, sub_ev_done[kMaxSub] = {};
  // host wall time of the blocking operator() calls (REGISTER_TIMES analogue: Frame::mTimeORB_Ext, Frame.cc:126-137)
  double host_ms_sum = 0, host_ms_sq = 0;
  long host_calls = 0;
  // timing
  bool timing = false;
  bool timing_fast = false;  // events around the FAST launch only: the stage chain keeps the shape of an untimed call
  hipEvent_t ev[kEv] = {};
  double acc_ms[kStages] = {};
  int acc_n = 0;
  bool ev_pending = false;
};

// dynamic LDS the fused pyramid may ask for (the launcher raises the 64 KB default limit; gfx950 has 160 KB)
constexpr int kPyrLdsLimit = 150000;

static void free_slot(Slot &S) {
  hipFree(S.d_in), hipFree(S.d_kps), hipFree(S.d_desc), hipFree(S.d_counts);
  hipHostFree(S.h_in), hipHostFree(S.h_kps), hipHostFree(S.h_desc), hipHostFree(S.h_counts);
  S.d_in = S.h_in = S.h_in_dev = nullptr;
  S.d_kps = S.h_kps = nullptr;
  S.d_desc = S.h_desc = nullptr;
  S.d_counts = S.h_counts = nullptr;
  S.allocated = S.busy = false;
  S.ticket = -1;
}

static void free_chain_graphs(vsg_orb *h) {
  for (ChainGraph &g : h->chain) {
    if (g.exec) hipGraphExecDestroy(g.exec);
    if (g.graph) hipGraphDestroy(g.graph);
    g = ChainGraph();
  }
}

static void free_image_buffers(vsg_orb *h) {
  free_chain_graphs(h);  // their nodes point into the buffers below
  hipFree(h->d_fg), hipFree(h->d_tab), hipFree(h->d_cells), hipFree(h->d_fast), hipFree(h->d_in);
  hipFree(h->d_cell_count), hipFree(h->d_cand2);
  h->d_cell_count = nullptr, h->d_cand2 = nullptr;
  for (int i = 0; i < kPyrTilings; i++) {
    hipFree(h->d_ptiles[i]), hipFree(h->d_ptab[i]);
    h->d_ptiles[i] = nullptr, h->d_ptab[i] = nullptr;
  }
  hipFree(h->d_pyr), hipFree(h->d_blur), hipFree(h->d_cand), hipFree(h->d_sel), hipFree(h->d_nodeof);
  hipFree(h->d_counts2), hipFree(h->d_flags), hipFree(h->d_slots), hipFree(h->d_hdr);
  for (int i = 0; i < kSlots; i++) free_slot(h->slot[i]);
  h->d_fg = nullptr, h->d_tab = nullptr, h->d_cells = nullptr, h->d_fast = nullptr, h->d_in = nullptr;
  h->last_src0 = {nullptr, 0, 0};
  h->have_last = false;
  h->d_pyr = h->d_blur = nullptr;
  h->d_cand = h->d_sel = nullptr;
  h->d_nodeof = nullptr;
  h->d_counts2 = h->d_flags = nullptr;
  h->d_slots = nullptr;
  h->d_hdr = nullptr;
  h->rows = h->cols = 0;
}

// one of the handle's auxiliary streams, created on first use (see vsg_orb_create)
static int need_stream(vsg_orb *h, hipStream_t *ps) {
  if (*ps) return VSG_OK;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamCreateWithFlags(ps, hipStreamNonBlocking));
  return VSG_OK;
}
// "the blur on a stream of the handle's own": resolved (and the stream created) only by a call that does not fuse the blur
static const hipStream_t kOwnBlurStream = (hipStream_t)(uintptr_t)1;

// every stream of the handle idle (before buffers are re-built or parameters change under running kernels)
static int quiesce(vsg_orb *h) {
  if (h->s_h2d) HIP_TRY(hipStreamSynchronize(h->s_h2d));
  HIP_TRY(hipStreamSynchronize(h->s_main));
  if (h->s_blur) HIP_TRY(hipStreamSynchronize(h->s_blur));
  if (h->s_d2h) HIP_TRY(hipStreamSynchronize(h->s_d2h));
  if (h->have_last) HIP_TRY(hipEventSynchronize(h->ev_last));  // the last enqueue may sit on a caller stream
  return VSG_OK;
}

// slot i's device + pinned buffers for the current geometry (slot 0 with the geometry, the others on first use)
static int ensure_slot(vsg_orb *h, int i) {
  Slot &S = h->slot[i];
  if (S.allocated) return VSG_OK;
  const FrameGeom &fg = h->G.fg;
  const size_t B = (size_t)h->max_batch;
  const size_t in_bytes = B * (size_t)h->rows * h->in_pitch + 64;
  // outputs live in mapped pinned memory: the export kernel stores the records there straight from the device
  const unsigned flags = hipHostMallocMapped | hipHostMallocPortable;
  HIP_TRY(hipMalloc(&S.d_in, in_bytes));
  HIP_TRY(hipHostMalloc(&S.h_in, in_bytes, hipHostMallocPortable | hipHostMallocMapped));
  HIP_TRY(hipHostGetDevicePointer((void **)&S.h_in_dev, S.h_in, 0));
  HIP_TRY(hipMalloc(&S.d_kps, B * fg.out_cap * sizeof(KeyPointPOD)));
  HIP_TRY(hipMalloc(&S.d_desc, B * fg.out_cap * 32));
  HIP_TRY(hipMalloc(&S.d_counts, B * 2 * sizeof(int)));
  HIP_TRY(hipHostMalloc(&S.h_kps, B * fg.out_cap * sizeof(KeyPointPOD), flags));
  HIP_TRY(hipHostMalloc(&S.h_desc, B * fg.out_cap * 32, flags));
  HIP_TRY(hipHostMalloc(&S.h_counts, B * 2 * sizeof(int), flags));
  if (!S.ev_in) {
    HIP_TRY(hipEventCreateWithFlags(&S.ev_in, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&S.ev_done, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&S.ev_out, hipEventDisableTiming));
  }
  S.allocated = true;
  return VSG_OK;
}

static int ensure_scratch(vsg_orb *h, size_t bytes) {
  if (bytes <= h->scratch_cap) return VSG_OK;
  int rc = quiesce(h);
  if (rc != VSG_OK) return rc;
  hipFree(h->d_scratch);
  h->d_scratch = nullptr;
  h->scratch_cap = 0;
  const size_t cap = bytes + bytes / 2;
  HIP_TRY(hipMalloc(&h->d_scratch, cap));
  h->scratch_cap = cap;
  return VSG_OK;
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
  // a new image size frees every pipeline slot: refuse while tickets of vsg_orb_submit_batch are still un-waited (their
  // pinned result buffers and counts would silently disappear and vsg_orb_wait could only say "unknown ticket")
  for (int i = 0; i < kSlots; i++)
    if (h->slot[i].busy) {
      set_err("image size changed while submitted batches have not been waited for (vsg_orb_wait them first)");
      return VSG_ERR_BUSY;
    }
  rc = quiesce(h);
  if (rc != VSG_OK) return rc;
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
  HIP_TRY(hipMalloc(&h->d_fast, sizeof(FastCellRec) * h->G.fastRecs.size()));
  HIP_TRY(hipMemcpy(h->d_fast, h->G.fastRecs.data(), sizeof(FastCellRec) * h->G.fastRecs.size(), hipMemcpyHostToDevice));
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
  HIP_TRY(hipMalloc(&h->d_blur, B * fg.blur_frame_bytes));
  HIP_TRY(hipMemset(h->d_pyr, 0, B * fg.pyr_frame_bytes));
  HIP_TRY(hipMemset(h->d_blur, 0, B * fg.blur_frame_bytes));
  HIP_TRY(hipMalloc(&h->d_cand, B * fg.cand_frame * sizeof(uint32_t)));
  HIP_TRY(hipMalloc(&h->d_cand2, B * fg.cand_frame * sizeof(uint32_t)));
  HIP_TRY(hipMalloc(&h->d_cell_count, B * fg.total_cells * sizeof(int)));
  HIP_TRY(hipMemset(h->d_cell_count, 0, B * fg.total_cells * sizeof(int)));
  // the fills above ran on the NULL stream; the handle's streams are non-blocking, i.e. NOT ordered against it
  HIP_TRY(hipStreamSynchronize(nullptr));
  HIP_TRY(hipMalloc(&h->d_nodeof, B * fg.cand_frame * sizeof(uint16_t)));
  HIP_TRY(hipMalloc(&h->d_sel, B * fg.sel_frame * sizeof(uint32_t)));
  HIP_TRY(hipMalloc(&h->d_counts2, 2 * B * kMaxLevels * sizeof(int)));
  HIP_TRY(hipMalloc(&h->d_flags, B * fg.out_cap * sizeof(int)));
  HIP_TRY(hipMalloc(&h->d_slots, B * fg.out_cap * sizeof(int4)));
  HIP_TRY(hipMalloc(&h->d_hdr, B * sizeof(FrameHeader)));
  h->in_pitch = (cols + 3) & ~3;
  HIP_TRY(hipMalloc(&h->d_in, B * (size_t)rows * h->in_pitch + 64));
  h->rows = rows;
  h->cols = cols;
  rc = ensure_slot(h, 0);
  if (rc != VSG_OK) return rc;
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
      for (int ti = 0; ti < 2; ti++) {  // the two throughput tilings; kPyrTilingSmall serves the small launches
        const PyrTiling &PT = h->G.pyr[ti];
        float ms = 0.f;
        for (int rep = 0; rep < 2; rep++) {  // first launch warms up, second is timed
          HIP_TRY(hipEventRecord(h->ev[0], h->s_main));
          launch_pyramid(h->s_main, h->d_pyr, h->d_fg, h->d_ptab[ti], s0, h->d_ptiles[ti], (int)PT.tiles.size(), PT.ldsA,
                         PT.ldsB, PT.tabMax, nf, nullptr);
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
  float ms;
  if (h->timing_fast) {
    if (hipEventSynchronize(h->ev[2]) != hipSuccess) return;
    if (hipEventElapsedTime(&ms, h->ev[8], h->ev[2]) == hipSuccess) h->acc_ms[1] += ms;
    h->acc_n++;
    return;
  }
  if (hipEventSynchronize(h->ev[5]) != hipSuccess || hipEventSynchronize(h->ev[7]) != hipSuccess) return;
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
                         hipEvent_t ev_pyr, hipEvent_t ev_blur, bool tm, bool tmf = false) {
  const FrameGeom &fg = h->G.fg;
  const size_t F = (size_t)f0;
  const Src0 s0 = {src.base + F * src.frame_stride, src.frame_stride, src.pitch};
  uint8_t *pyr = h->d_pyr + F * fg.pyr_frame_bytes, *blur = h->d_blur + F * fg.blur_frame_bytes;
  uint32_t *cand = h->d_cand + F * fg.cand_frame, *sel = h->d_sel + F * fg.sel_frame;
  uint16_t *nodeof = h->d_nodeof + F * fg.cand_frame;
  int *cell_count = h->d_cell_count + F * fg.total_cells;
  int *cand_count = h->d_counts2 + F * kMaxLevels;
  int *sel_count = h->d_counts2 + ((size_t)h->max_batch + F) * kMaxLevels;
  int *flags = h->d_flags + F * fg.out_cap;
  int4 *slots = h->d_slots + F * fg.out_cap;
  FrameHeader *hdr = h->d_hdr + F;
  Range r_all("vsg_orb: enqueue stage chain");
  if (tm) HIP_TRY(hipEventRecord(h->ev[0], s));
  // tiling: the coarser one when it still gives the chip enough workgroups and leaves three of them per CU; without a
  // usable tiling (scale factors > 2, LDS) the chain runs as one k_resize launch per level
  int ti = -1;
  if (fg.nlevels > 1) {
    const PyrTiling &P0 = h->G.pyr[0], &P1 = h->G.pyr[1];
    const PyrTiling &PS = h->G.pyr[kPyrTilingSmall];
    const int forced = h->force_tiling;  // vsg_orb_set_pyramid_tiling: -1 = automatic, 0 / 1 / 2 = a tiling, 3 = per-level launches
    if (forced == 3)
      ti = -1;
    else if (forced >= 0 && forced < kPyrTilings)
      ti = h->G.pyr[forced].ok && h->G.pyr[forced].lds_bytes() <= kPyrLdsLimit ? forced : -1;
    else if (PS.ok && PS.lds_bytes() <= kPyrLdsLimit && (long long)P0.tiles.size() * nf <= 256)
      ti = kPyrTilingSmall;  // the coarse tiling would leave most CUs without a workgroup (one or two frames per call)
    else if (h->pyr_tiling == 1 && (long long)P1.tiles.size() * nf >= 2048)  // calibrated choice, large launches only
      ti = 1;
    else if (P0.ok && P0.lds_bytes() <= kPyrLdsLimit)
      ti = 0;
  }
  range_push("ComputePyramid");
  if (ti < 0) {
    launch_zero(s, cand_count, nf * kMaxLevels);  // the fused kernel clears the candidate counters itself
    for (int l = 1; l < fg.nlevels; l++) launch_resize(s, pyr, h->d_fg, h->d_tab, s0, fg, l, nf);
  } else {
    const PyrTiling &PT = h->G.pyr[ti];
    launch_pyramid(s, pyr, h->d_fg, h->d_ptab[ti], s0, h->d_ptiles[ti], (int)PT.tiles.size(), PT.ldsA, PT.ldsB,
                   PT.tabMax, nf, cand_count);
  }
  range_pop();
  if (tm) HIP_TRY(hipEventRecord(h->ev[1], s));
  // The blur only needs the pyramid and runs on its own stream.  It is released AFTER FAST, beside the
  // latency-bound octree (+ slots): FAST and the blur are both issue-bound, so running them side by side only
  // shares the CUs, while the octree leaves most issue slots free.  Measured on MI355X (C2, one batch of 256):
  // 254 k frames/s against 250 k with the blur released right after the pyramid.
  if (tm || tmf) HIP_TRY(hipEventRecord(h->ev[8], s));
  launch_fast(s, pyr, h->d_fg, h->d_fast, s0, cand, cand_count, cell_count, fg, h->G.fastMaxVh, h->G.fastMaxVw,
              h->G.fastMaxArea, nf, h->cus);
  if (tm || tmf) HIP_TRY(hipEventRecord(h->ev[2], s));
  // The blur's workgroups ride in the octree's launch (k_octree_blur): both need only the pyramid, the octree is a
  // latency-bound handful of workgroups per frame and the blur issue-bound filler, and inside ONE kernel (one register
  // allocation) the two kinds of waves co-reside on a SIMD -- as two kernels on two streams the octree's 96-register
  // waves left no room for an 80-register blur wave, so the two shared the CUs in time.  Measured: + 1.6-2.7 % on the
  // 512-frame step (295.0 -> 299.8 k frames/s, 302.9 k on a second box), and on the one-frame latency path no fork /
  // join events (each cross-stream wait cost the chain 5-7 us of idle GPU): 0.127 -> 0.108 ms.  Timed / serialised runs
  // keep the two kernels apart.
  // Throughput batches only take the fused launch while five of its workgroups (the 5 waves per SIMD it is compiled
  // for) still fit a CU's 160 KB of LDS: every blur workgroup is charged the octree's workspace, and at 1280x720 /
  // 2000 features (33 KB) the fifth no longer fits -- 93.1 -> 90.1 k frames/s there, so that geometry keeps two streams.
  // (re-measured in round 5 with the launch compiled for 4 waves per SIMD, where four 33 KB workgroups WOULD fit: 1280x720 /
  // 2000 fused 113.2-114.6 k frames/s, two streams 115.4-117.2 k -- the threshold stays at five workgroups' worth of LDS)
  // (kFusedLdsWorkgroups is a MEASURED threshold, not the launch's residency: the fused launch is compiled for kOctBlurWaves = 4
  // workgroups per CU.  Geometries whose octree workspace -- vsg_octree_core.h work_bytes, + 1.8 KB per workgroup since round 5's
  // histogram / cell-position tables -- lies between 160 / 5 = 32 KB and 160 / 4 = 40 KB take the two-stream form: 1280x720 / 2000
  // at 33 KB is the measured case above; nothing between 29 and 33 KB occurs among the reference's configurations
  // (tests/golden/reference_configs.json: 640x480 / 1000-1500 = 17-23 KB, 752x480 / 1200 = 20 KB, 1241x376 / 2000 = 33 KB).)
  constexpr int kFusedLdsWorkgroups = 5;
  const bool lds_fits = kFusedLdsWorkgroups * octree_lds_bytes(fg, h->G.maxQuota, h->G.maxCellsPerLevel) <= 160 * 1024;
  const bool fused_blur = !tm && sb != s && (lds_fits || (h->one_stream && nf <= 8));
  // Without a lapping area (every keypoint has x >= 19, so lap1 < 19 -- the {0, 0} of the RGB-D / stereo callers,
  // Frame.cc:108,344 -- selects nothing): k_orient_desc derives slots and level starts itself, k_slots is not launched
  // (8 us of the one-frame chain; for 512-frame batches + 0.4 % frames/s in the same run, round 4; in the two-stream form of
  // the geometries whose octree workspace keeps the blur out of its launch -- 1280x720 / 2000 -- k_slots was 51 us of a
  // 128-frame step, 5 %: profiles/r05_v_c4_kernel_stats_timed_region.csv)
  const bool self_slots = !tm && (lap1 < kEdgeThreshold || lap0 > lap1);
  if (fused_blur) {
    Range r_tail("DistributeOctTree (+ blur workgroups) + slots + IC_Angle / rBRIEF");
    launch_octree(s, h->d_fg, cand, cand_count, h->d_cells, cell_count, h->d_cand2 + F * fg.cand_frame, nodeof, sel,
                  sel_count, fg, h->G.maxQuota, h->G.maxCellsPerLevel, nf, pyr, blur, &s0);
    if (!self_slots) launch_slots(s, h->d_fg, sel, sel_count, flags, slots, hdr, lap0, lap1, nf);
  } else {
  HIP_TRY(hipEventRecord(ev_pyr, s));
  // Host enqueue order: the latency-critical launch (the octree) goes out BEFORE the three calls that fork the blur onto
  // its stream.  Serialised runs (sb == s, per-stage timing) keep stream order = stage order: blur, then octree.
  if (sb == kOwnBlurStream) {
    const int rs = need_stream(h, &h->s_blur);
    if (rs != VSG_OK) return rs;
    sb = h->s_blur;
  }
  const bool octree_first = sb != s;
  if (octree_first) {
    if (tm) HIP_TRY(hipEventRecord(h->ev[10], s));
    launch_octree(s, h->d_fg, cand, cand_count, h->d_cells, cell_count, h->d_cand2 + F * fg.cand_frame, nodeof, sel,
                  sel_count, fg, h->G.maxQuota, h->G.maxCellsPerLevel, nf);
    if (tm) HIP_TRY(hipEventRecord(h->ev[3], s));
  }
  HIP_TRY(hipStreamWaitEvent(sb, ev_pyr, 0));
  if (tm) HIP_TRY(hipEventRecord(h->ev[6], sb));
  launch_blur(sb, pyr, blur, h->d_fg, s0, fg, nf);
  if (tm) HIP_TRY(hipEventRecord(h->ev[7], sb));
  HIP_TRY(hipEventRecord(ev_blur, sb));
  Range r_tail("DistributeOctTree + slots + IC_Angle / rBRIEF");
  if (!octree_first) {
    if (tm) HIP_TRY(hipEventRecord(h->ev[10], s));
    launch_octree(s, h->d_fg, cand, cand_count, h->d_cells, cell_count, h->d_cand2 + F * fg.cand_frame, nodeof, sel,
                  sel_count, fg, h->G.maxQuota, h->G.maxCellsPerLevel, nf);
    if (tm) HIP_TRY(hipEventRecord(h->ev[3], s));
  }
  if (!self_slots) launch_slots(s, h->d_fg, sel, sel_count, flags, slots, hdr, lap0, lap1, nf);
  if (tm) HIP_TRY(hipEventRecord(h->ev[4], s));
  HIP_TRY(hipStreamWaitEvent(s, ev_blur, 0));
  if (tm) HIP_TRY(hipEventRecord(h->ev[9], s));
  }
  OutMirror mir = h->mirror;
  if (mir.kps) mir.kps += F * mir.capacity, mir.desc += F * mir.capacity * 32, mir.counts += F * 2;
#ifdef VSG_OD_SPATIAL_ON
  if (self_slots) slots = (int4 *)nodeof;  // experiment: the processing order of k_orient_desc (vsg_kernels.hip VSG_OD_SPATIAL)
#endif
  launch_orient_desc(s, pyr, blur, h->d_fg, s0, sel, sel_count, slots, hdr, h->d_pattern, d_kps + F * capacity,
                     d_desc + F * capacity * 32, d_counts + F * 2, capacity, fg, nf, mir, self_slots);
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
  const bool tm = h->timing && nsub == 1 && !h->capturing;
  const bool tmf = !tm && h->timing_fast && nsub == 1 && !h->capturing;
  if (tm || tmf) harvest_timing(h);
  if (nsub == 1) {
    int rc = enqueue_range(h, s0, 0, nframes, lap0, lap1, d_kps, d_desc, d_counts, capacity, s,
                           no_overlap ? s : kOwnBlurStream, h->ev_pyr, h->ev_blur, tm, tmf);
    if (rc != VSG_OK) return rc;
    if (tm || tmf) h->ev_pending = true;
  } else {
    HIP_TRY(hipEventRecord(h->ev_fork, s));
    const int per = (nframes + nsub - 1) / nsub;
    for (int j = 0; j < nsub; j++) {
      const int f0 = j * per, nf = nframes - f0 < per ? nframes - f0 : per;
      if (nf <= 0) break;
      int rs = need_stream(h, &h->sub_s[j]);
      if (rs == VSG_OK) rs = need_stream(h, &h->sub_b[j]);
      if (rs != VSG_OK) return rs;
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
  // where this call's results are and when they are complete (stage read-back, stereo, vsg_frame_from_extractor)
  h->last_kps = d_kps, h->last_desc = d_desc, h->last_counts = d_counts, h->last_cap = capacity;
  HIP_TRY(hipEventRecord(h->ev_last, s));
  h->have_last = true;
  return VSG_OK;
}

// wait (on the host) for the last enqueue, whichever stream it went to
static int wait_last(vsg_orb *h) {
  if (h->have_last) HIP_TRY(hipEventSynchronize(h->ev_last));
  if (h->s_blur) HIP_TRY(hipStreamSynchronize(h->s_blur));
  return VSG_OK;
}

// `stream` argument of the device entry points: NULL means the caller's NULL stream.  The chain runs on the handle's
// own non-blocking stream, ordered behind everything the NULL stream holds at entry, and the NULL stream is made to
// wait for it at exit -- so producer -> extract -> consumer sequences on the default stream are ordered.
struct NullStreamBridge {
  vsg_orb *h;
  bool on;
  hipStream_t s;
  NullStreamBridge(vsg_orb *h_, void *stream) : h(h_), on(stream == nullptr), s(stream ? (hipStream_t)stream : h_->s_main) {}
  int enter() {
    if (!on) return VSG_OK;
    HIP_TRY(hipEventRecord(h->ev_null_in, nullptr));
    HIP_TRY(hipStreamWaitEvent(s, h->ev_null_in, 0));
    return VSG_OK;
  }
  int leave() {
    if (!on) return VSG_OK;
    HIP_TRY(hipEventRecord(h->ev_null_out, s));
    HIP_TRY(hipStreamWaitEvent(nullptr, h->ev_null_out, 0));
    return VSG_OK;
  }
};

// ---- caller host memory the device may touch IN PLACE -------------------------------------------------------------
// Three kinds of pinned caller memory exist: (1) ranges this library handed out (vsg_host_alloc = hipHostMalloc, kept in
// the registry below), (2) hipHostMalloc memory of somebody else (torch's pinned tensors, the caller's own ring), (3)
// ordinary heap pages pinned after the fact (vsg_host_register = hipHostRegister: a user-pointer mapping).  In-place DMA
// and kernel access to kind (3) is where profiles/r04_q_open_issue_gpu_fault.txt happened (a page of a live registered
// range lost its device mapping; tools/repro_hostregister.cpp is the library-free reproducer), so kind (3) is staged
// through the slot's own hipHostMalloc buffers like pageable memory unless the caller opts in per handle
// (vsg_orb_set_direct_registered).  The WHOLE range [p, p + bytes) is validated, not its first byte.
static std::mutex g_alloc_mu;
static std::map<uintptr_t, size_t> g_allocs;  // vsg_host_alloc: base -> bytes
static std::map<uintptr_t, size_t> g_regs;    // vsg_host_register: base -> bytes

// the registered / allocated range that holds byte a, if any: 0 = none, 1 = holds [a, a + bytes), -1 = holds a but not the end
static int in_ranges(const std::map<uintptr_t, size_t> &m, const void *p, size_t bytes) {
  const uintptr_t a = (uintptr_t)p;
  auto it = m.upper_bound(a);
  if (it == m.begin()) return 0;
  --it;
  if (a < it->first || a - it->first >= it->second) return 0;
  return bytes <= it->second && a - it->first <= it->second - bytes ? 1 : -1;
}

// What the runtime says about one byte this library knows nothing about: 0 = pageable / unknown, 2 = hipHostMalloc memory,
// 3 = anything else that is pinned.  FAIL CLOSED: "somebody else's hipHostMalloc" is only claimed when the
// runtime answers hipHostGetFlags for the pointer AND the device sees the byte at the host's own address -- hipHostMalloc
// memory is one allocation mapped at one address for both, a hipHostRegister mapping (hsa_amd_memory_lock) hands the
// device an address of its own.  hipHostGetFlags refusing registered memory is undocumented ROCclr behaviour and not
// relied on; whatever fails either test is REGISTERED: staged unless the handle opted in.
static int runtime_kind(const void *p, void **dev_alias) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // pageable memory: not an error
    return VSG_HOST_PAGEABLE;
  }
  if (a.type != hipMemoryTypeHost) return VSG_HOST_PAGEABLE;
  if (dev_alias) *dev_alias = a.devicePointer;
  unsigned int fl = 0;
  const bool flags_ok = hipHostGetFlags(&fl, const_cast<void *>(p)) == hipSuccess;
  if (!flags_ok) (void)hipGetLastError();
  return flags_ok && a.devicePointer == p ? VSG_HOST_HIPHOSTMALLOC : VSG_HOST_REGISTERED;
}

static int host_kind(const void *p, size_t bytes, void **dev_alias) {
  if (!p || !bytes) return VSG_HOST_PAGEABLE;
  // ranges this library pinned itself are classified by ITS books, whatever the runtime says about them; a span that starts
  // inside one and ends outside it is not pinned as a whole
  int lib, reg;
  {
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    lib = in_ranges(g_allocs, p, bytes), reg = in_ranges(g_regs, p, bytes);
  }
  if (lib < 0 || reg < 0) return VSG_HOST_PAGEABLE;
  void *a0 = nullptr, *a1 = nullptr;
  const int k0 = runtime_kind(p, &a0);
  if (k0 == VSG_HOST_PAGEABLE) return k0;
  const int k1 = bytes > 1 ? runtime_kind((const uint8_t *)p + bytes - 1, &a1) : k0;
  // both ends pinned the same way, one contiguous device alias (two registrations back to back would differ here)
  if (k1 != k0 || (bytes > 1 && (uintptr_t)a1 - (uintptr_t)a0 != bytes - 1)) return VSG_HOST_PAGEABLE;
  if (dev_alias) *dev_alias = a0;
  if (reg) return VSG_HOST_REGISTERED;
  if (lib) return VSG_HOST_LIB_ALLOC;
  return k0;
}

// may the device read / write [p, p + bytes) in place on behalf of handle h?
static bool host_direct(const vsg_orb *h, const void *p, size_t bytes, void **dev_alias) {
  void *alias = nullptr;
  const int k = host_kind(p, bytes, &alias);
  if (k == VSG_HOST_PAGEABLE || !alias) return false;
  if (k == VSG_HOST_REGISTERED && !h->direct_registered) return false;
  if (dev_alias) *dev_alias = alias;
  return true;
}

// bytes spanned by `nframes` images of rows x cols (row stride `stride`, frame stride `frame_stride`)
static size_t image_span(int nframes, size_t frame_stride, int rows, int cols, int stride) {
  return (size_t)(nframes - 1) * frame_stride + (size_t)(rows - 1) * stride + (size_t)cols;
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

int vsg_host_register(void *ptr, size_t bytes) {
  if (!ptr || !bytes) return VSG_ERR_INVALID;
  HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterMapped | hipHostRegisterPortable));
  std::lock_guard<std::mutex> lk(g_alloc_mu);
  g_regs[(uintptr_t)ptr] = bytes;
  return VSG_OK;
}
int vsg_host_unregister(void *ptr) {
  if (!ptr) return VSG_ERR_INVALID;
  {
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    g_regs.erase((uintptr_t)ptr);
  }
  HIP_TRY(hipHostUnregister(ptr));
  return VSG_OK;
}
int vsg_host_alloc(size_t bytes, void **out) {
  if (!out || !bytes) return VSG_ERR_INVALID;
  *out = nullptr;
  HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocMapped | hipHostMallocPortable));
  std::lock_guard<std::mutex> lk(g_alloc_mu);
  g_allocs[(uintptr_t)*out] = bytes;
  return VSG_OK;
}
int vsg_host_free(void *ptr) {
  if (!ptr) return VSG_ERR_INVALID;
  {
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    auto it = g_allocs.find((uintptr_t)ptr);
    if (it == g_allocs.end()) {
      set_err("vsg_host_free: not a pointer vsg_host_alloc returned");
      return VSG_ERR_INVALID;
    }
    g_allocs.erase(it);
  }
  HIP_TRY(hipHostFree(ptr));
  return VSG_OK;
}
int vsg_host_kind(const void *ptr, size_t bytes) {
  if (!ptr || !bytes) return VSG_ERR_INVALID;
  return host_kind(ptr, bytes, nullptr);
}
int vsg_orb_set_direct_registered(vsg_orb *h, int enable) {
  if (!h) return VSG_ERR_INVALID;
  h->direct_registered = enable != 0;
  free_chain_graphs(h);  // a recorded chain holds the source / destination route it was recorded with
  return VSG_OK;
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
  // ONE stream at creation; the blur / copy / sub-batch streams come when a path first needs them (need_stream).  The
  // runtime deals a process's streams over a handful of hardware queues: with 20 streams per handle created up front,
  // the main streams of four handles -- four camera streams, BASELINE's C5 -- all sat on the SAME queue and their
  // kernels ran one after the other (profiles/r04_n_c5_stream_overlap.txt).
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->s_main, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_pyr, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_blur, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_last, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_null_in, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_null_out, hipEventDisableTiming) != hipSuccess ||
      hipMalloc(&h->d_pattern, 1024) != hipSuccess ||
      hipMemcpy(h->d_pattern, kPattern, 1024, hipMemcpyHostToDevice) != hipSuccess) {
    set_err("HIP initialisation failed");
    vsg_orb_destroy(h);
    return VSG_ERR_NO_DEVICE;
  }
  if (hipDeviceGetAttribute(&h->cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || h->cus <= 0) {
    (void)hipGetLastError();
    h->cus = 256;
  }
  for (int i = 0; i < kEv; i++) hipEventCreate(&h->ev[i]);
  hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
  for (int j = 0; j < kMaxSub; j++) {
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
  if (h->s_main) quiesce(h);
  delete h->pool;
  h->pool = nullptr;
  free_image_buffers(h);
  hipFree(h->d_pattern);
  hipFree(h->d_scratch);
  for (int i = 0; i < kSlots; i++) {
    Slot &S = h->slot[i];
    if (S.ev_in) hipEventDestroy(S.ev_in), hipEventDestroy(S.ev_done), hipEventDestroy(S.ev_out);
  }
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
  if (h->ev_last) hipEventDestroy(h->ev_last);
  if (h->ev_null_in) hipEventDestroy(h->ev_null_in);
  if (h->ev_null_out) hipEventDestroy(h->ev_null_out);
  if (h->s_main) hipStreamDestroy(h->s_main);
  if (h->s_blur) hipStreamDestroy(h->s_blur);
  if (h->s_h2d) hipStreamDestroy(h->s_h2d);
  if (h->s_d2h) hipStreamDestroy(h->s_d2h);
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
  if (h->rows) {  // refresh the device copy of the geometry -- never under running kernels
    for (int k = 0; k < 7; k++) h->G.fg.taps[k] = taps[k];
    HIP_TRY(hipSetDevice(h->device));
    int rc = quiesce(h);
    if (rc != VSG_OK) return rc;
    HIP_TRY(hipMemcpy(h->d_fg, &h->G.fg, sizeof(FrameGeom), hipMemcpyHostToDevice));
  }
  return VSG_OK;
}

int vsg_orb_set_pyramid_tiling(vsg_orb *h, int which) {
  if (!h || which < -1 || which > kPyrTilings) return VSG_ERR_INVALID;
  h->force_tiling = which;
  free_chain_graphs(h);  // a recorded chain holds the launch form it was recorded with
  return VSG_OK;
}

int vsg_orb_capacity(vsg_orb *h, int rows, int cols) {
  if (!h) return VSG_ERR_INVALID;
  if (rows <= 0 || cols <= 0) return VSG_ERR_EMPTY_IMAGE;
  int rc = ensure_geometry(h, rows, cols);
  if (rc != VSG_OK) return rc;
  return h->G.fg.out_cap;
}

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
  NullStreamBridge br(h, stream);
  rc = br.enter();
  if (rc != VSG_OK) return rc;
  hipStream_t s = br.s;
  Src0 s0;
  if (((uintptr_t)d_gray & 3) == 0 && (stride & 3) == 0 && (frame_stride & 3) == 0) {
    s0 = {d_gray, frame_stride, stride};  // level 0 is read in place: no ingest copy
  } else {                                 // unaligned layout: restage (rare)
    for (int f = 0; f < nframes; f++)
      HIP_TRY(hipMemcpy2DAsync(h->d_in + (size_t)f * rows * h->in_pitch, h->in_pitch, d_gray + (size_t)f * frame_stride,
                               stride, cols, rows, hipMemcpyDeviceToDevice, s));
    s0 = {h->d_in, (size_t)rows * h->in_pitch, h->in_pitch};
  }
  rc = enqueue_pipeline(h, s0, nframes, lap0, lap1, (KeyPointPOD *)d_kps, d_desc, d_counts, capacity, s);
  if (rc != VSG_OK) return rc;
  return br.leave();
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
  NullStreamBridge br(h, stream);
  rc = br.enter();
  if (rc != VSG_OK) return rc;
  hipStream_t s = br.s;
  // cvtColor of Tracking::GrabImage* straight into the gray level-0 staging buffer
  launch_cvt_gray(s, d_img, frame_stride, stride, channels, rgb_order, rows, cols, h->d_in,
                  (size_t)rows * h->in_pitch, h->in_pitch, h->gray_coeffs, h->gray_shift, nframes);
  const Src0 s0 = {h->d_in, (size_t)rows * h->in_pitch, h->in_pitch};
  rc = enqueue_pipeline(h, s0, nframes, lap0, lap1, (KeyPointPOD *)d_kps, d_desc, d_counts, capacity, s);
  if (rc != VSG_OK) return rc;
  return br.leave();
}

// ---- host API: submit / wait over the pipeline slots ---------------------------------------------------------

int vsg_orb_slots(const vsg_orb *h) { return h ? kSlots : VSG_ERR_INVALID; }

long vsg_orb_chain_graph_launches(const vsg_orb *h) { return h ? h->chain_launches : VSG_ERR_INVALID; }

// where the export of slot S goes: the caller's own arrays when both are pinned (the device writes them directly, nothing
// is left for vsg_orb_wait to copy), the slot's pinned staging otherwise
struct ExportDst {
  bool direct = false;
  void *dk = nullptr, *dd = nullptr;
};
static ExportDst export_dst(const vsg_orb *h, int nframes, vsg_keypoint *kps, uint8_t *desc, int capacity) {
  ExportDst e;
  const size_t recs = (size_t)nframes * (capacity > 0 ? capacity : 0);
  e.direct = kps && desc && capacity > 0 && host_direct(h, kps, recs * sizeof(vsg_keypoint), &e.dk) &&
             host_direct(h, desc, recs * 32, &e.dd) && e.dk && e.dd;
  return e;
}

// the stage chain + the export of slot S on the handle's streams (what a ChainGraph records)
static int tail_stream_work(vsg_orb *h, Slot &S, int nframes, int lap0, int lap1, const ExportDst &E, int capacity,
                            bool mirror_out = false) {
  const FrameGeom &fg = h->G.fg;
  const Src0 s0 = {S.d_in, (size_t)h->rows * h->in_pitch, h->in_pitch};
  if (mirror_out) {
    // latency path: k_orient_desc writes the records into the pinned destination as it produces them
    h->mirror.kps = (KeyPointPOD *)(E.direct ? E.dk : (void *)S.h_kps), h->mirror.desc = (uint8_t *)(E.direct ? E.dd : (void *)S.h_desc);
    h->mirror.counts = S.h_counts, h->mirror.capacity = E.direct ? capacity : fg.out_cap;
  }
  int rc = enqueue_pipeline(h, s0, nframes, lap0, lap1, S.d_kps, S.d_desc, S.d_counts, fg.out_cap, h->s_main);
  h->mirror = OutMirror();
  if (rc != VSG_OK) return rc;
  if (mirror_out) return VSG_OK;
  if (!h->one_stream) {
    const int rs = need_stream(h, &h->s_d2h);
    if (rs != VSG_OK) return rs;
  }
  const hipStream_t s_out = h->one_stream ? h->s_main : h->s_d2h;
  if (!h->one_stream) {
    HIP_TRY(hipEventRecord(S.ev_done, h->s_main));
    HIP_TRY(hipStreamWaitEvent(h->s_d2h, S.ev_done, 0));
  }
  // export: n records per frame, written by the device into pinned host memory
  if (E.direct) {
    launch_export(s_out, S.d_kps, S.d_desc, S.d_counts, fg.out_cap, E.dk, E.dd, S.h_counts, capacity, nframes);
  } else {
    launch_export(s_out, S.d_kps, S.d_desc, S.d_counts, fg.out_cap, S.h_kps, S.h_desc, S.h_counts, fg.out_cap, nframes);
  }
  HIP_TRY(hipGetLastError());
  return VSG_OK;
}

// bookkeeping of a submitted batch; records the completion event behind whatever was enqueued / launched
static int finish_submit(vsg_orb *h, Slot &S, int nframes, vsg_keypoint *kps, uint8_t *desc, int capacity, bool direct) {
  if (h->post_fn) {  // e.g. the resident frame's grid launch: on the chain's stream, covered by the call's one wait
    const vsg_post_chain_fn fn = h->post_fn;
    void *ctx = h->post_ctx;
    h->post_fn = nullptr, h->post_ctx = nullptr;
    OrbOutputView v;
    int rc = vsg_orb_output_view(h, 0, &v);
    if (rc != VSG_OK) return rc;
    v.done = nullptr;  // same stream: ordered behind the chain already
    rc = fn(ctx, h->one_stream ? h->s_main : h->s_d2h, v);
    if (rc != VSG_OK) return rc;
  }
  S.direct = direct;
  S.out_kps = kps, S.out_desc = desc, S.out_cap = capacity;
  HIP_TRY(hipEventRecord(S.ev_out, h->one_stream ? h->s_main : h->s_d2h));
  S.busy = true;
  S.nframes = nframes;
  S.ticket = h->next_ticket++;
  return S.ticket;
}

// kernels of the chain + export for slot S, behind its ev_in
static int submit_tail(vsg_orb *h, Slot &S, int nframes, int lap0, int lap1, vsg_keypoint *kps, uint8_t *desc,
                       int capacity) {
  const ExportDst E = export_dst(h, nframes, kps, desc, capacity);
  int rc = tail_stream_work(h, S, nframes, lap0, lap1, E, capacity);
  if (rc != VSG_OK) return rc;
  return finish_submit(h, S, nframes, kps, desc, capacity, E.direct);
}

// ---- latency mode (see ChainGraph): ingest kernel + chain + export of a blocking call, replayed as one graph launch
static int chain_stream_work(vsg_orb *h, Slot &S, const uint8_t *src_dev, size_t sframe, int sstride, int nframes, int lap0,
                             int lap1, const ExportDst &E, int capacity) {
  launch_ingest(h->s_main, src_dev, sframe, sstride, S.d_in, (size_t)h->rows * h->in_pitch, h->in_pitch, h->rows, h->cols,
                nframes);
  HIP_TRY(hipGetLastError());
  return tail_stream_work(h, S, nframes, lap0, lap1, E, capacity, true);
}

static int run_chain(vsg_orb *h, Slot &S, int slot_index, const uint8_t *src_dev, size_t sframe, int sstride, int nframes,
                     int lap0, int lap1, const ExportDst &E, int capacity) {
  // Opt-in (VSG_GRAPH=1): measured on MI355X / ROCm 7.0 the graph launch of this 8-node chain is no faster than eight
  // eager launches from the calling thread (single-frame operator() 0.139 ms with the graph, 0.131 ms without:
  // profiles/r03_*_frame_latency*); the GPU-side dependency chain is what a call waits for, not the host enqueue.
  static const bool use_graph = getenv("VSG_GRAPH") != nullptr;
  const bool eligible = use_graph && !h->timing && !h->serialize && h->nsub <= 1 && nframes <= 4;
  if (!eligible) return chain_stream_work(h, S, src_dev, sframe, sstride, nframes, lap0, lap1, E, capacity);
  ChainKey key;
  key.slot = slot_index, key.nframes = nframes, key.lap0 = lap0, key.lap1 = lap1, key.stride = sstride;
  key.capacity = E.direct ? capacity : 0, key.frame_stride = sframe, key.src = src_dev;
  key.dk = E.direct ? E.dk : nullptr, key.dd = E.direct ? E.dd : nullptr;
  ChainGraph *g = nullptr, *lru = &h->chain[0];
  for (ChainGraph &c : h->chain) {
    if (c.seen && c.key == key) g = &c;
    if (c.stamp < lru->stamp) lru = &c;
  }
  if (!g) {  // first call with this key: run eagerly (this also raises the LDS limits, times the tilings, ...)
    if (lru->exec) hipGraphExecDestroy(lru->exec);
    if (lru->graph) hipGraphDestroy(lru->graph);
    *lru = ChainGraph();
    lru->key = key, lru->seen = 1, lru->stamp = ++h->chain_clock;
    return chain_stream_work(h, S, src_dev, sframe, sstride, nframes, lap0, lap1, E, capacity);
  }
  g->stamp = ++h->chain_clock;
  if (!g->exec) {  // second call: record the same calls instead of running them
    HIP_TRY(hipStreamBeginCapture(h->s_main, hipStreamCaptureModeThreadLocal));
    h->capturing = true;
    const int rc = chain_stream_work(h, S, src_dev, sframe, sstride, nframes, lap0, lap1, E, capacity);
    h->capturing = false;
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(h->s_main, &graph);
    if (rc != VSG_OK || e != hipSuccess || !graph) {
      if (graph) hipGraphDestroy(graph);
      (void)hipGetLastError();
      g->seen = 0;  // do not try again with this key
      g->key.slot = -2;
      return chain_stream_work(h, S, src_dev, sframe, sstride, nframes, lap0, lap1, E, capacity);
    }
    hipGraphExec_t exec = nullptr;
    if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
      hipGraphDestroy(graph);
      (void)hipGetLastError();
      g->seen = 0, g->key.slot = -2;
      return chain_stream_work(h, S, src_dev, sframe, sstride, nframes, lap0, lap1, E, capacity);
    }
    g->graph = graph, g->exec = exec;
  }
  HIP_TRY(hipGraphLaunch(g->exec, h->s_main));
  h->chain_launches++;
  // what enqueue_pipeline notes on the host when it runs (it only ran while being recorded)
  const FrameGeom &fg = h->G.fg;
  h->last_src0 = {S.d_in, (size_t)h->rows * h->in_pitch, h->in_pitch};
  h->last_frames = nframes;
  h->last_kps = S.d_kps, h->last_desc = S.d_desc, h->last_counts = S.d_counts, h->last_cap = fg.out_cap;
  HIP_TRY(hipEventRecord(h->ev_last, h->s_main));
  h->have_last = true;
  return VSG_OK;
}

static int acquire_slot(vsg_orb *h, Slot **out) {
  const int i = h->next_ticket % kSlots;
  Slot &S = h->slot[i];
  if (S.busy) {
    set_err("every pipeline slot holds a batch that has not been waited for (vsg_orb_wait)");
    return VSG_ERR_BUSY;
  }
  int rc = ensure_slot(h, i);
  if (rc != VSG_OK) return rc;
  *out = &S;
  return VSG_OK;
}

int vsg_orb_submit_batch(vsg_orb *h, const uint8_t *gray, int nframes, size_t frame_stride, int rows, int cols,
                         int stride, int lap0, int lap1, vsg_keypoint *kps, uint8_t *desc, int capacity) {
  if (!h || nframes < 1 || nframes > h->max_batch) return VSG_ERR_INVALID;
  if (!gray || rows <= 0 || cols <= 0) return VSG_ERR_EMPTY_IMAGE;
  HIP_TRY(hipSetDevice(h->device));
  int rc = ensure_geometry(h, rows, cols);
  if (rc != VSG_OK) return rc;
  Slot *Sp = nullptr;
  rc = acquire_slot(h, &Sp);
  if (rc != VSG_OK) return rc;
  Slot &S = *Sp;
  const int ip = h->in_pitch;
  const size_t fbytes = (size_t)rows * ip;
  const bool packed = stride == ip && (nframes == 1 || frame_stride == fbytes);
  if (!h->one_stream) {
    const int rs = need_stream(h, &h->s_h2d);
    if (rs != VSG_OK) return rs;
  }
  const hipStream_t s_in = h->one_stream ? h->s_main : h->s_h2d;
  if (h->one_stream && nframes <= 8) {
    // Blocking call, small batch (the reference's one frame per operator()): the latency path.  Pageable images are
    // staged into the slot's pinned buffer by this thread; the device then reads the pinned image itself (ingest
    // kernel) and the whole call's stream work is one graph launch.
    void *alias = nullptr;
    const uint8_t *src_dev;
    size_t sframe;
    int sstride;
    if (host_direct(h, gray, image_span(nframes, frame_stride, rows, cols, stride), &alias)) {
      src_dev = (const uint8_t *)alias, sframe = frame_stride, sstride = stride;
    } else {
      for (int f = 0; f < nframes; f++) {
        uint8_t *dst = S.h_in + f * fbytes;
        const uint8_t *src = gray + (size_t)f * frame_stride;
        if (stride == ip)
          memcpy(dst, src, fbytes);
        else
          for (int y = 0; y < rows; y++) memcpy(dst + (size_t)y * ip, src + (size_t)y * stride, (size_t)cols);
      }
      src_dev = S.h_in_dev, sframe = fbytes, sstride = ip;
    }
    const ExportDst E = export_dst(h, nframes, kps, desc, capacity);
    rc = run_chain(h, S, (int)(Sp - h->slot), src_dev, sframe, sstride, nframes, lap0, lap1, E, capacity);
    if (rc != VSG_OK) return rc;
    return finish_submit(h, S, nframes, kps, desc, capacity, E.direct);
  }
  if (host_direct(h, gray, image_span(nframes, frame_stride, rows, cols, stride), nullptr)) {
    // pinned caller memory the device may read in place: DMA straight from it (one descriptor for a packed batch, one 2-D copy per frame otherwise)
    if (packed) {
      HIP_TRY(hipMemcpyAsync(S.d_in, gray, fbytes * nframes, hipMemcpyHostToDevice, s_in));
    } else {
      for (int f = 0; f < nframes; f++)
        HIP_TRY(hipMemcpy2DAsync(S.d_in + f * fbytes, ip, gray + (size_t)f * frame_stride, stride, cols, rows,
                                 hipMemcpyHostToDevice, s_in));
    }
  } else {
    // pageable memory: bounce through the slot's pinned staging.  Small batches: in chunks, so that the copy engine
    // works on chunk i while this thread copies chunk i + 1.  Large batches: the frames are dealt to this thread and
    // the handle's helper threads, then ONE DMA (which overlaps the staging of the next batch, another slot).
    constexpr int kStageHelpers = 3;  // flat beyond 3-4 (63 k frames/s alone, 113 k with 3 helpers: DESIGN 8, round 2)
    if (nframes >= 16) {
      if (!h->pool) h->pool = new StagePool(kStageHelpers);
      StagePool::Job j;
      j.src = gray, j.dst = S.h_in, j.frame_stride = frame_stride, j.fbytes = fbytes;
      j.stride = stride, j.ip = ip, j.rows = rows, j.cols = cols, j.nframes = nframes;
      h->pool->run(j);
      HIP_TRY(hipMemcpyAsync(S.d_in, S.h_in, fbytes * nframes, hipMemcpyHostToDevice, s_in));
    } else {
      const int chunk = 8;
      for (int f0 = 0; f0 < nframes; f0 += chunk) {
        const int nf = nframes - f0 < chunk ? nframes - f0 : chunk;
        for (int f = f0; f < f0 + nf; f++) {
          uint8_t *dst = S.h_in + f * fbytes;
          const uint8_t *src = gray + (size_t)f * frame_stride;
          if (stride == ip)
            memcpy(dst, src, fbytes);
          else
            for (int y = 0; y < rows; y++) memcpy(dst + (size_t)y * ip, src + (size_t)y * stride, (size_t)cols);
        }
        HIP_TRY(hipMemcpyAsync(S.d_in + f0 * fbytes, S.h_in + f0 * fbytes, fbytes * nf, hipMemcpyHostToDevice, s_in));
      }
    }
  }
  if (!h->one_stream) {
    HIP_TRY(hipEventRecord(S.ev_in, s_in));
    HIP_TRY(hipStreamWaitEvent(h->s_main, S.ev_in, 0));
  }
  return submit_tail(h, S, nframes, lap0, lap1, kps, desc, capacity);
}

int vsg_orb_wait(vsg_orb *h, int ticket, int *n, int *mono_index) {
  if (!h || ticket < 0 || !n || !mono_index) return VSG_ERR_INVALID;
  Slot &S = h->slot[ticket % kSlots];
  if (!S.busy || S.ticket != ticket) {
    set_err("unknown ticket (already waited for, or never submitted)");
    return VSG_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipEventSynchronize(S.ev_out));
  S.busy = false;
  const FrameGeom &fg = h->G.fg;
  int status = VSG_OK;
  for (int f = 0; f < S.nframes; f++) {
    const int nf = S.h_counts[2 * f];
    n[f] = nf;
    mono_index[f] = S.h_counts[2 * f + 1];
    if (nf > S.out_cap && (S.out_kps || S.out_desc)) {
      set_err("caller capacity too small for the extracted keypoints");
      status = VSG_ERR_CAPACITY;
      continue;
    }
    if (S.direct || nf <= 0) continue;
    if (S.out_kps) memcpy(S.out_kps + (size_t)f * S.out_cap, S.h_kps + (size_t)f * fg.out_cap, (size_t)nf * sizeof(KeyPointPOD));
    if (S.out_desc) memcpy(S.out_desc + (size_t)f * S.out_cap * 32, S.h_desc + (size_t)f * fg.out_cap * 32, (size_t)nf * 32);
  }
  return status;
}

int vsg_orb_extract_batch(vsg_orb *h, const uint8_t *gray, int nframes, size_t frame_stride, int rows, int cols,
                          int stride, int lap0, int lap1, vsg_keypoint *kps, uint8_t *desc, int capacity, int *n,
                          int *mono_index) {
  if (!h || nframes < 1 || nframes > h->max_batch || !n || !mono_index) return VSG_ERR_INVALID;
  if (!gray || rows <= 0 || cols <= 0) return VSG_ERR_EMPTY_IMAGE;
  Range r("ORBextractor::operator() [blocking batch]");
  const auto t0 = std::chrono::steady_clock::now();
  h->one_stream = true;
  const int t = vsg_orb_submit_batch(h, gray, nframes, frame_stride, rows, cols, stride, lap0, lap1, kps, desc, capacity);
  h->one_stream = false;
  if (t < 0) return t;
  const int rc = vsg_orb_wait(h, t, n, mono_index);
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  h->host_ms_sum += ms, h->host_ms_sq += ms * ms, h->host_calls++;
  return rc;
}

// "ORB Extraction" mean / std in ms over the blocking operator() calls so far, as Tracking::PrintTimeStats reports it
// under REGISTER_TIMES (Tracking.cc:291-330, Settings.h:23).  Returns the number of calls; reset != 0 clears.
int vsg_orb_time_stats(vsg_orb *h, double *mean_ms, double *std_ms, int reset) {
  if (!h) return VSG_ERR_INVALID;
  const long n = h->host_calls;
  const double mean = n ? h->host_ms_sum / n : 0.0;
  if (mean_ms) *mean_ms = mean;
  if (std_ms) *std_ms = n > 1 ? std::sqrt(std::max(0.0, (h->host_ms_sq - n * mean * mean) / (n - 1))) : 0.0;
  if (reset) h->host_ms_sum = h->host_ms_sq = 0, h->host_calls = 0;
  return (int)n;
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
  const size_t row_bytes = (size_t)cols * channels, frame_bytes = row_bytes * rows;
  rc = ensure_scratch(h, frame_bytes * nframes);  // grows once; no allocation per call
  if (rc != VSG_OK) return rc;
  Slot *Sp = nullptr;
  rc = acquire_slot(h, &Sp);
  if (rc != VSG_OK) return rc;
  Slot &S = *Sp;
  // A blocking call: upload, cvtColor of Tracking::GrabImage* into the slot's gray level-0 staging, the stage chain
  // and the export all on s_main (see vsg_orb::one_stream).  The scratch buffer is shared by consecutive colour
  // batches; stream order on s_main is what keeps the previous conversion ahead of this upload.
  for (int f = 0; f < nframes; f++)
    HIP_TRY(hipMemcpy2DAsync(h->d_scratch + f * frame_bytes, row_bytes, img + (size_t)f * frame_stride, stride, row_bytes,
                             rows, hipMemcpyHostToDevice, h->s_main));
  launch_cvt_gray(h->s_main, h->d_scratch, frame_bytes, (int)row_bytes, channels, rgb_order, rows, cols, S.d_in,
                  (size_t)rows * h->in_pitch, h->in_pitch, h->gray_coeffs, h->gray_shift, nframes);
  h->one_stream = true;
  const int t = submit_tail(h, S, nframes, lap0, lap1, kps, desc, capacity);
  h->one_stream = false;
  if (t < 0) return t;
  return vsg_orb_wait(h, t, n, mono_index);
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

// Stage read-back of the LAST call.  Level 0 lives in the buffer that call read it from: the slot's staging for the
// host entry points, the CALLER's device buffer for vsg_orb_extract_batch_device -- which must therefore still be
// alive (image_pyramid(0), vsg_stereo_matches).
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
  int rc = wait_last(h);
  if (rc != VSG_OK) return rc;
  if (!with_border) {
    HIP_TRY(hipMemcpy2D(dst, dst_stride, img, ipitch, L.w, L.h, hipMemcpyDeviceToHost));
    return VSG_OK;
  }
  const int b = kEdgeThreshold, bw = L.w + 2 * b, bh = L.h + 2 * b;
  rc = ensure_scratch(h, (size_t)bw * bh);
  if (rc != VSG_OK) return rc;
  launch_border_copy(h->s_main, img, L.w, L.h, ipitch, h->d_scratch, bw, b);
  HIP_TRY(hipMemcpy2DAsync(dst, dst_stride, h->d_scratch, bw, bw, bh, hipMemcpyDeviceToHost, h->s_main));
  HIP_TRY(hipStreamSynchronize(h->s_main));
  return VSG_OK;
}

int vsg_orb_copy_pyramid_level(vsg_orb *h, int frame, int level, int with_border, uint8_t *dst, int dst_stride) {
  return copy_level(h, h ? h->d_pyr : nullptr, frame, level, with_border, dst, dst_stride);
}
// The blurred levels are stored as 16 x 4-pixel tiles (LevelGeom::btx / boff, vsg_common.h): the level comes down as it
// lies and is de-tiled into the caller's rows here (a test / debug read-back, not on any hot path).
int vsg_orb_copy_blurred_level(vsg_orb *h, int frame, int level, uint8_t *dst, int dst_stride) {
  if (!h || !h->rows || !dst || level < 0 || level >= h->T.nlevels || frame < 0 || frame >= h->max_batch)
    return VSG_ERR_INVALID;
  HIP_TRY(hipSetDevice(h->device));
  const FrameGeom &fg = h->G.fg;
  const LevelGeom &L = fg.lv[level];
  int rc = wait_last(h);
  if (rc != VSG_OK) return rc;
  const size_t bytes = (size_t)L.btx * L.bty * kBlurTileBytes;
  std::vector<uint8_t> tiled(bytes);
  HIP_TRY(hipMemcpy(tiled.data(), h->d_blur + (size_t)frame * fg.blur_frame_bytes + L.boff, bytes, hipMemcpyDeviceToHost));
  for (int y = 0; y < L.h; y++)
    for (int x = 0; x < L.w; x += kBlurTileW)
      memcpy(dst + (size_t)y * dst_stride + x, &tiled[(size_t)blur_tiled_offset(x, y, L.btx)],
             (size_t)std::min<int>(kBlurTileW, L.w - x));
  return VSG_OK;
}

// One D2H for mvImagePyramid (ORBextractor.h:93) of frame `frame`: every level WITH its 19 px REFLECT_101 border,
// packed back to back into dst (level l at offsets[l], row stride = level width + 38).  Frame::ComputeStereoMatches
// (Frame.cc:964,1054-1069) is the only host reader.  Returns the bytes written, or the bytes needed when dst is NULL
// or dst_bytes is too small (nothing is written then).
int vsg_orb_copy_pyramid(vsg_orb *h, int frame, uint8_t *dst, size_t dst_bytes, size_t *offsets) {
  if (!h || !h->rows || frame < 0 || frame >= h->max_batch) return VSG_ERR_INVALID;
  const FrameGeom &fg = h->G.fg;
  const int b = kEdgeThreshold;
  size_t total = 0;
  for (int l = 0; l < fg.nlevels; l++) {
    if (offsets) offsets[l] = total;
    total += (size_t)(fg.lv[l].w + 2 * b) * (fg.lv[l].h + 2 * b);
  }
  if (total > 0x7FFFFFFFu) return VSG_ERR_UNSUPPORTED;
  if (!dst || dst_bytes < total) return (int)total;
  if (!h->last_src0.base) return VSG_ERR_INVALID;
  HIP_TRY(hipSetDevice(h->device));
  int rc = wait_last(h);
  if (rc != VSG_OK) return rc;
  rc = ensure_scratch(h, total);
  if (rc != VSG_OK) return rc;
  size_t off = 0;
  for (int l = 0; l < fg.nlevels; l++) {
    const LevelGeom &L = fg.lv[l];
    const uint8_t *img = l == 0 ? h->last_src0.base + (size_t)frame * h->last_src0.frame_stride
                                : h->d_pyr + (size_t)frame * fg.pyr_frame_bytes + L.img_off;
    launch_border_copy(h->s_main, img, L.w, L.h, l == 0 ? h->last_src0.pitch : L.pitch, h->d_scratch + off, L.w + 2 * b, b);
    off += (size_t)(L.w + 2 * b) * (L.h + 2 * b);
  }
  HIP_TRY(hipMemcpyAsync(dst, h->d_scratch, total, hipMemcpyDeviceToHost, h->s_main));
  HIP_TRY(hipStreamSynchronize(h->s_main));
  return (int)total;
}

static int copy_list(vsg_orb *h, const uint32_t *list, size_t frame_elems, int off, int lcap, const int *counts,
                     int frame, int level, uint32_t *dst, int cap) {
  if (!h || !h->rows || level < 0 || level >= h->T.nlevels || frame < 0 || frame >= h->max_batch) return VSG_ERR_INVALID;
  HIP_TRY(hipSetDevice(h->device));
  int rc = wait_last(h);
  if (rc != VSG_OK) return rc;
  int n = 0;
  HIP_TRY(hipMemcpy(&n, counts + frame * kMaxLevels + level, sizeof(int), hipMemcpyDeviceToHost));
  if (n > lcap) n = lcap;
  const int m = n < cap ? n : cap;
  if (m > 0 && dst) HIP_TRY(hipMemcpy(dst, list + (size_t)frame * frame_elems + off, (size_t)m * 4, hipMemcpyDeviceToHost));
  return n;
}

int vsg_orb_copy_candidates(vsg_orb *h, int frame, int level, uint32_t *dst, int cap) {
  if (!h || !h->rows || level < 0 || level >= h->T.nlevels) return VSG_ERR_INVALID;
  const FrameGeom &fg = h->G.fg;
  const LevelGeom &L = fg.lv[level];
  if (!fg.cand_segmented)
    return copy_list(h, h->d_cand, fg.cand_frame, L.cand_off, L.cand_cap, h->d_counts2, frame, level, dst, cap);
  // per-cell segments: the level's slice and its cell counts come down, the list is assembled here (debug path)
  if (frame < 0 || frame >= h->max_batch) return VSG_ERR_INVALID;
  HIP_TRY(hipSetDevice(h->device));
  int rc = wait_last(h);
  if (rc != VSG_OK) return rc;
  const int ncells = L.nCols * L.nRows;
  std::vector<int> counts((size_t)ncells);
  std::vector<uint32_t> slice((size_t)L.cand_cap);
  HIP_TRY(hipMemcpy(counts.data(), h->d_cell_count + (size_t)frame * fg.total_cells + L.cell_base, (size_t)ncells * 4,
                    hipMemcpyDeviceToHost));
  if (L.cand_cap > 0)
    HIP_TRY(hipMemcpy(slice.data(), h->d_cand + (size_t)frame * fg.cand_frame + L.cand_off, (size_t)L.cand_cap * 4,
                      hipMemcpyDeviceToHost));
  int n = 0;
  for (int ci = 0; ci < ncells; ci++) {
    const CellDesc &c = h->G.cells[(size_t)L.cell_base + ci];
    for (int k = 0; k < counts[ci]; k++, n++)
      if (dst && n < cap) dst[n] = slice[(size_t)c.cand_off + k];
  }
  return n;
}
int vsg_orb_copy_selected(vsg_orb *h, int frame, int level, uint32_t *dst, int cap) {
  if (!h || !h->rows || level < 0 || level >= h->T.nlevels) return VSG_ERR_INVALID;
  const LevelGeom &L = h->G.fg.lv[level];
  return copy_list(h, h->d_sel, h->G.fg.sel_frame, L.sel_off, L.sel_cap,
                   h->d_counts2 + (size_t)h->max_batch * kMaxLevels, frame, level, dst, cap);
}

}  // extern "C"

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

void vsg_orb_set_post_chain(vsg_orb *h, vsg_post_chain_fn fn, void *ctx) {
  if (h) h->post_fn = fn, h->post_ctx = ctx;
}

int vsg_orb_device_of(const vsg_orb *h) { return h ? h->device : -1; }

int vsg_orb_output_view(vsg_orb *h, int index, OrbOutputView *v) {
  if (!h || !v || !h->have_last || index < 0 || index >= h->last_frames) return VSG_ERR_INVALID;
  v->d_kps = h->last_kps + (size_t)index * h->last_cap;
  v->d_desc = h->last_desc + (size_t)index * h->last_cap * 32;
  v->d_counts = h->last_counts + 2 * index;
  v->done = h->ev_last;
  v->device = h->device;
  return VSG_OK;
}

// median-based outlier cut of Frame::ComputeStereoMatches (Frame.cc:1113-1126)
static int stereo_median_cut(const int *sad, int n_l, float *u_right, float *depth) {
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

// Frame::ComputeStereoMatches on device-resident keypoints / descriptors: one launch on the calling thread's stream,
// results through its pinned arena.  d_kps* / d_desc* are device pointers.  Two halves (round 6): enqueue, and -- after the
// stream has been waited for, possibly together with other work of the same Frame -- the copy-out and the median cut.
static int stereo_enqueue(vsg_orb *hl, int frame_l, vsg_orb *hr, int frame_r, const KeyPointPOD *dkl, const uint8_t *ddl,
                          int n_l, const KeyPointPOD *dkr, const uint8_t *ddr, int n_r, float mb, float mbf, ThreadCtx *c,
                          size_t out_off) {
  PyrView pl, pr;
  int rc = pyr_view(hl, frame_l, pl);
  if (rc == VSG_OK) rc = pyr_view(hr, frame_r, pr);
  if (rc != VSG_OK) return rc;
  // both extractors' last enqueues must be complete before their pyramids are read
  if (hl->have_last) HIP_TRY(hipStreamWaitEvent(c->stream, hl->ev_last, 0));
  if (hr->have_last) HIP_TRY(hipStreamWaitEvent(c->stream, hr->ev_last, 0));
  const size_t N = (size_t)n_l;
  float *du = (float *)(c->d_pin + out_off), *dd = du + N;
  int *ds = (int *)(dd + N);
  launch_stereo(c->stream, pl, pr, mb, mbf, hl->T.scale.data(), hl->T.invScale.data(), hl->T.nlevels, dkl, ddl, n_l, dkr,
                ddr, n_r, du, dd, ds);
  HIP_TRY(hipGetLastError());
  return VSG_OK;
}
static int stereo_finish(ThreadCtx *c, size_t out_off, int n_l, float *u_right, float *depth) {
  const size_t N = (size_t)n_l;
  const float *hu = (const float *)(c->h_pin + out_off), *hd = hu + N;
  const int *hs = (const int *)(hd + N);
  memcpy(u_right, hu, N * 4);
  memcpy(depth, hd, N * 4);
  return stereo_median_cut(hs, n_l, u_right, depth);
}
static int stereo_run(vsg_orb *hl, int frame_l, vsg_orb *hr, int frame_r, const KeyPointPOD *dkl, const uint8_t *ddl,
                      int n_l, const KeyPointPOD *dkr, const uint8_t *ddr, int n_r, float mb, float mbf, ThreadCtx *c,
                      size_t out_off, float *u_right, float *depth) {
  const int rc = stereo_enqueue(hl, frame_l, hr, frame_r, dkl, ddl, n_l, dkr, ddr, n_r, mb, mbf, c, out_off);
  if (rc != VSG_OK) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return stereo_finish(c, out_off, n_l, u_right, depth);
}

extern "C" {

int vsg_stereo_matches(vsg_orb *hl, int frame_l, vsg_orb *hr, int frame_r, const vsg_keypoint *kps_l,
                       const uint8_t *desc_l, int n_l, const vsg_keypoint *kps_r, const uint8_t *desc_r, int n_r,
                       float mb, float mbf, float *u_right, float *depth) {
  if (!hl || !hr || !u_right || !depth || n_l < 0 || n_r < 0 || hl->device != hr->device) return VSG_ERR_INVALID;
  for (int i = 0; i < n_l; i++) u_right[i] = -1.0f, depth[i] = -1.0f;
  if (n_l == 0 || n_r == 0) return 0;
  int rc = VSG_OK;
  ThreadCtx *c = thread_ctx(hl->device, &rc);
  if (!c) return rc;
  // keypoints and descriptors of both eyes go up through the pinned arena (one DMA into the device arena)
  Stage st;
  const size_t oKL = st.add(sizeof(KeyPointPOD) * n_l), oKR = st.add(sizeof(KeyPointPOD) * n_r);
  const size_t oDL = st.add(32 * (size_t)n_l), oDR = st.add(32 * (size_t)n_r);
  const size_t in_bytes = st.total;
  const size_t oOut = st.add(12 * (size_t)n_l);
  rc = ctx_reserve(c, st.total, in_bytes);
  if (rc != VSG_OK) return rc;
  memcpy(c->h_pin + oKL, kps_l, sizeof(KeyPointPOD) * n_l);
  memcpy(c->h_pin + oKR, kps_r, sizeof(KeyPointPOD) * n_r);
  memcpy(c->h_pin + oDL, desc_l, 32 * (size_t)n_l);
  memcpy(c->h_pin + oDR, desc_r, 32 * (size_t)n_r);
  HIP_TRY(hipMemcpyAsync(c->d_buf, c->h_pin, in_bytes, hipMemcpyHostToDevice, c->stream));
  return stereo_run(hl, frame_l, hr, frame_r, (const KeyPointPOD *)(c->d_buf + oKL), c->d_buf + oDL, n_l,
                    (const KeyPointPOD *)(c->d_buf + oKR), c->d_buf + oDR, n_r, mb, mbf, c, oOut, u_right, depth);
}

int vsg_frame_stereo_matches(vsg_orb *hl, int frame_l, vsg_orb *hr, int frame_r, vsg_frame *fl, vsg_frame *fr,
                             float mb, float mbf, float *u_right, float *depth) {
  if (!hl || !hr || !fl || !fr || !u_right || !depth || hl->device != hr->device || fl->device != hl->device ||
      fr->device != hl->device)
    return VSG_ERR_INVALID;
  const int n_l = fl->n, n_r = fr->n;
  for (int i = 0; i < n_l; i++) u_right[i] = -1.0f, depth[i] = -1.0f;
  if (n_l == 0 || n_r == 0) return 0;
  int rc = VSG_OK;
  ThreadCtx *c = thread_ctx(hl->device, &rc);
  if (!c) return rc;
  rc = ctx_reserve(c, 12 * (size_t)n_l + 64, 0);
  if (rc != VSG_OK) return rc;
  return stereo_run(hl, frame_l, hr, frame_r, fl->d_kps, fl->d_desc, n_l, fr->d_kps, fr->d_desc, n_r, mb, mbf, c, 0,
                    u_right, depth);
}

// The part of a stereo Frame's construction and first tracking step that depends only on two resident frames, in ONE enqueue
// and ONE wait (round 6, VERDICT r5 #3b): ComputeStereoMatches (Frame.cc:957-1127) -> ComputeBoW (Frame.cc:882-889, the
// BowVector / FeatureVector assembled on the device) -> SearchByBoW(KF, F) (ORBmatcher.cc:226-428) against a KeyFrame whose
// FeatureVector is resident from ITS ComputeBoW.  Three blocking calls cost three stream round trips (~20 us each) for
// kernels of 16 + 6 + 37 us; here the kernels queue up behind each other and the host parts (median cut, rotation histogram)
// run after the one wait, each on the bytes its blocking form would have seen.
int vsg_frame_stereo_bow_search(vsg_orb *hl, int frame_l, vsg_orb *hr, int frame_r, vsg_frame *fl, vsg_frame *fr, float mb,
                                float mbf, float *u_right, float *depth, int *n_stereo, vsg_vocab *voc, int levelsup,
                                int32_t *bow_ids, double *bow_vals, int bow_cap, int *n_bow, int32_t *fv_node,
                                int32_t *fv_off, int32_t *fv_idx, int fv_cap, int *n_fv, vsg_frame *kf,
                                const uint8_t *kf_valid, float nnratio, int check_orientation, int32_t *match_f,
                                int *n_match) {
  if (!hl || !hr || !fl || !fr || !u_right || !depth || !n_stereo || !voc || !n_bow || !n_fv || !fv_off ||
      hl->device != hr->device || fl->device != hl->device || fr->device != hl->device || vsg::vocab_device(voc) != hl->device)
    return VSG_ERR_INVALID;
  if (kf && (!kf_valid || !match_f || !n_match || kf->device != hl->device || kf == fl)) return VSG_ERR_INVALID;
  const int n_l = fl->n, n_r = fr->n;
  for (int i = 0; i < n_l; i++) u_right[i] = -1.0f, depth[i] = -1.0f;
  *n_stereo = 0;
  if (n_match) *n_match = 0;
  int rc = VSG_OK;
  ThreadCtx *c = thread_ctx(hl->device, &rc);
  if (!c) return rc;
  // one arena layout for the three calls, reserved BEFORE anything is enqueued (growing an arena frees the old one)
  size_t pinB = 0, devB = 0, pinS = 0, devS = 0;
  vsg::bow_sizes(n_l, false, &pinB, &devB);
  if (kf) vsg::bow_search_sizes(kf->n, n_l, 0, &pinS, &devS);
  const size_t pin_stereo = (12 * (size_t)n_l + 127) & ~(size_t)63;
  rc = ctx_reserve(c, pin_stereo + pinB + pinS, devB + devS);
  if (rc != VSG_OK) return rc;
  const bool do_stereo = n_l > 0 && n_r > 0;
  if (do_stereo) {
    rc = stereo_enqueue(hl, frame_l, hr, frame_r, fl->d_kps, fl->d_desc, n_l, fr->d_kps, fr->d_desc, n_r, mb, mbf, c, 0);
    if (rc != VSG_OK) return rc;
  }
  vsg::BowCall b;
  rc = vsg::bow_enqueue(&b, voc, nullptr, fl->d_desc, n_l, levelsup, fl, c, pin_stereo, 0);
  if (rc != VSG_OK) return rc;
  vsg::BowSearchCall sc;
  const bool do_search = kf != nullptr && kf->fv_valid && fl->fv_valid;
  if (kf && !do_search && kf->n > 0 && n_l > 0 && b.active) return VSG_ERR_INVALID;  // the KeyFrame never had its ComputeBoW
  if (do_search) {
    rc = vsg::bow_search_enqueue(&sc, 0, kf, kf_valid, fl, nullptr, nnratio, c, pin_stereo + pinB, devB);
    if (rc != VSG_OK) return rc;
  }
  HIP_TRY(hipStreamSynchronize(c->stream));  // the ONE wait
  if (do_stereo) *n_stereo = stereo_finish(c, 0, n_l, u_right, depth);
  rc = vsg::bow_finish(&b, bow_ids, bow_vals, bow_cap, n_bow, fv_node, fv_off, fv_idx, fv_cap, n_fv, nullptr, nullptr, nullptr);
  if (kf) {
    for (int i = 0; i < n_l; i++) match_f[i] = -1;
    if (do_search) *n_match = vsg::bow_search_finish(&sc, check_orientation, match_f);
  }
  return rc;
}

int vsg_orb_set_serialize(vsg_orb *h, int serialize) {
  if (!h) return VSG_ERR_INVALID;
  h->serialize = serialize != 0;
  return VSG_OK;
}

int vsg_orb_enable_timing(vsg_orb *h, int enable) {
  if (!h) return VSG_ERR_INVALID;
  h->timing = enable == 1;
  h->timing_fast = enable == 2;
  h->ev_pending = false;
  memset(h->acc_ms, 0, sizeof(h->acc_ms));
  h->acc_n = 0;
  return VSG_OK;
}

int vsg_debug_device_sort(int device, uint64_t *items, int n) {
  if (!items || n < 0 || n > 2048) return VSG_ERR_INVALID;
  if (n == 0) return vsg_device_count() > device && device >= 0 ? VSG_OK : VSG_ERR_NO_DEVICE;
  int rc = VSG_OK;
  ThreadCtx *c = thread_ctx(device, &rc);
  if (!c) return rc;
  rc = ctx_reserve(c, sizeof(uint64_t) * n, sizeof(uint64_t) * n);
  if (rc != VSG_OK) return rc;
  memcpy(c->h_pin, items, sizeof(uint64_t) * n);
  HIP_TRY(hipMemcpyAsync(c->d_buf, c->h_pin, sizeof(uint64_t) * n, hipMemcpyHostToDevice, c->stream));
  launch_debug_sort(c->stream, (uint64_t *)c->d_buf, n);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(c->h_pin, c->d_buf, sizeof(uint64_t) * n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  memcpy(items, c->h_pin, sizeof(uint64_t) * n);
  return VSG_OK;
}

int vsg_orb_get_timing(vsg_orb *h, float *ms_out, int cap) {
  if (!h) return VSG_ERR_INVALID;
  harvest_timing(h);
  for (int i = 0; i < kStages && i < cap; i++) ms_out[i] = h->acc_n ? (float)(h->acc_ms[i] / h->acc_n) : 0.f;
  return kStages;
}

}  // extern "C"
