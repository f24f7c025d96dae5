// vsg_frame.hip -- device-resident Frame / KeyFrame features and the windowed ORBmatcher searches on them.
//
//   Frame::AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea      orb_slam3/src/Frame.cc:521-553, 870-880, 802-868
//   KeyFrame::GetFeaturesInArea                                      orb_slam3/src/KeyFrame.cc:834-874
//   ORBmatcher::SearchByProjection x5, SearchBySim3, Fuse x2,
//   SearchForInitialization                                          orb_slam3/src/ORBmatcher.cc (lines cited below)
//
// One kernel, k_window_search, does everything that is data-parallel in those routines for ALL queries of a call:
// wave = one projected map point: the GetFeaturesInArea window on the resident CSR grid (lanes over the window's
// cells, candidates kept in the reference's order: cells ix outer / iy inner, insertion order inside a cell), the
// level / stereo / chi-square gates, and the Hamming distance to every surviving candidate (query descriptor in
// SGPRs, v_xor + v_bcnt).  Inputs are read from, and results written to, the calling thread's pinned arena straight
// over PCIe: a call is  fill -> ONE launch -> sync -> ordered host pass (vsg_walks.h).  Nothing is allocated and
// nothing runs on the NULL stream.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>
#include <time.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "vsg_frame_int.h"
#include "vsg_math.h"
#include "vsg_undistort.h"

using namespace vsg;

namespace {

enum { VSG_RETRY = -100 };  // internal: candidate lists overflowed the compact array; the entry point runs again

__device__ __forceinline__ int wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
  return v;
}

struct WinLaunch {
  int nq, mode, gate_mode, best_init, cap;
  uint32_t counter_base;
  float inv_sigma2[16];
};

// ---- Frame::AssignFeaturesToGrid (Frame.cc:521-553) for keypoints [i0, i0 + n) -> CSR (cell_start, ent) with the
// entries of every cell in ascending keypoint order (= the push_back order of the reference), index = i - i0.
// One workgroup: per-cell counts by LDS atomics, one block scan, an unordered atomic append, then every cell's
// (handful of) entries are put in index order by the thread that owns the cell.
__global__ __launch_bounds__(1024) void k_frame_grid_build(const KeyPointPOD *__restrict__ kps, int i0, int n,
                                                            float minX, float minY, float invW, float invH,
                                                            int *__restrict__ cell_start, GridEnt *__restrict__ ent,
                                                            KeyPointPOD *__restrict__ kps_copy,
                                                            const uint8_t *__restrict__ desc_src,
                                                            uint8_t *__restrict__ desc_copy,
                                                            int *__restrict__ zero_cells, CamModel cam,
                                                            KeyPointPOD *__restrict__ kps_un_host,
                                                            const int *__restrict__ n_dev, int n_cap) {
  // n_dev: the keypoint count where the extractor's chain left it ({n, monoIndex} of the frame) -- the launch that rides
  // behind operator()'s chain (vsg_orb_extract_to_frame) is enqueued before the host knows it
  if (n_dev) n = min(*n_dev, n_cap);
  __shared__ int s_cnt[kGridCells + 1];
  __shared__ int s_fill[kGridCells];
  __shared__ int s_wtot[16];
  const int tid = threadIdx.x;
  // making a frame resident straight out of the extractor is ONE launch: the keypoint / descriptor records are copied
  // into the frame's block and the (absent) right-camera grid is emptied by the threads that build the grid
  // Frame::UndistortKeyPoints (Frame.cc:891-921) on the way: with a distorted camera the frame's keypoints are mvKeysUn
  // -- every pt through cv::undistortPoints' five double-precision iterations (vsg_undistort.h) -- and the grid below is
  // built from THEM; the host's copy of mvKeysUn is written to pinned memory by the same threads
  const bool undist = kps_copy && cam.distorted;
  if (undist) {
    for (int i = tid; i < n; i += 1024) {
      KeyPointPOD kp = kps[i0 + i];
      undistort_point(cam, kp.x, kp.y, &kp.x, &kp.y);
      kps_copy[i] = kp;
      if (kps_un_host) kps_un_host[i] = kp;
    }
  } else if (kps_copy) {
    for (int i = tid; i < n * 7; i += 1024) ((uint32_t *)kps_copy)[i] = ((const uint32_t *)(kps + i0))[i];
  }
  if (desc_copy)
    for (int i = tid; i < n * 8; i += 1024) ((uint32_t *)desc_copy)[i] = ((const uint32_t *)desc_src)[i];
  if (zero_cells)
    for (int c = tid; c <= kGridCells; c += 1024) zero_cells[c] = 0;
  for (int c = tid; c <= kGridCells; c += 1024) s_cnt[c] = 0;
  __syncthreads();  // (also: the undistorted records above are visible to the whole workgroup)
  const KeyPointPOD *gsrc = undist ? kps_copy : kps + i0;  // the keypoints the grid indexes (mvKeysUn)
  // PosInGrid (Frame.cc:870-880): round() = half away from zero
  auto cell_of = [&](const KeyPointPOD &kp) -> int {
    const int px = cvt_int_x86(roundf(fmul(fsub(kp.x, minX), invW)));
    const int py = cvt_int_x86(roundf(fmul(fsub(kp.y, minY), invH)));
    return (px < 0 || px >= kGridCols || py < 0 || py >= kGridRows) ? -1 : px * kGridRows + py;
  };
  for (int i = tid; i < n; i += 1024) {
    const int c = cell_of(gsrc[i]);
    if (c >= 0) atomicAdd(&s_cnt[c], 1);
  }
  __syncthreads();
  {  // exclusive scan of the 3072 counts: 3 cells per thread
    const int c0 = tid * 3;
    const int a = s_cnt[c0], b = s_cnt[c0 + 1], c = s_cnt[c0 + 2];
    const int s = a + b + c;
    const int incl = wave_incl_scan(s);
    if ((tid & 63) == 63) s_wtot[tid >> 6] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (tid >> 6); w++) base += s_wtot[w];
    const int start = base + incl - s;
    __syncthreads();
    s_cnt[c0] = start, s_cnt[c0 + 1] = start + a, s_cnt[c0 + 2] = start + a + b;
    s_fill[c0] = start, s_fill[c0 + 1] = start + a, s_fill[c0 + 2] = start + a + b;
    if (tid == 1023) s_cnt[kGridCells] = start + s;
  }
  __syncthreads();
  for (int c = tid; c <= kGridCells; c += 1024) cell_start[c] = s_cnt[c];
  // The reference's cells list their features in index order (mGrid[i][j].push_back in AssignFeaturesToGrid's loop,
  // Frame.cc:521-533): the scatter's atomics do not, so every cell is put in order afterwards.  Up to kGridLdsMax
  // keypoints that happens in LDS on the 4-byte (index | octave) words alone and the 12-byte entries are written once,
  // coalesced, from the sorted words (sorting the entries where they lie -- global memory, a dependent round trip per
  // comparison -- was a third of this launch's 18 us at 1250 keypoints).
  __shared__ uint32_t s_io[kGridLdsMax];
  if (n <= kGridLdsMax) {
    for (int i = tid; i < n; i += 1024) {
      const KeyPointPOD kp = gsrc[i];
      const int c = cell_of(kp);
      if (c < 0) continue;
      s_io[atomicAdd(&s_fill[c], 1)] = (uint32_t)i | ((uint32_t)(kp.octave & 0xFFFF) << 16);
    }
    __syncthreads();
    for (int c = tid; c < kGridCells; c += 1024) {  // insertion sort by index: cells hold a few entries
      const int e0 = s_cnt[c], e1 = s_cnt[c + 1];
      for (int a = e0 + 1; a < e1; a++) {
        const uint32_t v = s_io[a];
        int b = a - 1;
        while (b >= e0 && (s_io[b] & 0xFFFFu) > (v & 0xFFFFu)) {
          s_io[b + 1] = s_io[b];
          b--;
        }
        s_io[b + 1] = v;
      }
    }
    __syncthreads();
    const int nin = s_cnt[kGridCells];
    for (int e = tid; e < nin; e += 1024) {
      const uint32_t io = s_io[e];
      const KeyPointPOD kp = gsrc[io & 0xFFFFu];
      ent[e] = {kp.x, kp.y, io};
    }
    return;
  }
  for (int i = tid; i < n; i += 1024) {
    const KeyPointPOD kp = gsrc[i];
    const int c = cell_of(kp);
    if (c < 0) continue;
    const int slot = atomicAdd(&s_fill[c], 1);
    ent[slot] = {kp.x, kp.y, (uint32_t)i | ((uint32_t)(kp.octave & 0xFFFF) << 16)};
  }
  __threadfence_block();
  __syncthreads();
  for (int c = tid; c < kGridCells; c += 1024) {  // insertion sort by index: cells hold a few entries
    const int e0 = s_cnt[c], e1 = s_cnt[c + 1];
    for (int a = e0 + 1; a < e1; a++) {
      const GridEnt v = ent[a];
      int b = a - 1;
      while (b >= e0 && (ent[b].io & 0xFFFFu) > (v.io & 0xFFFFu)) {
        ent[b + 1] = ent[b];
        b--;
      }
      ent[b + 1] = v;
    }
  }
}

// ---- the window search (see the file header).  4 queries per 256-thread workgroup, one per wavefront.
// List mode: every query owns an inline slot of kInline entries (one 64-byte line) in the output array, so that the
// host's ordered pass streams through memory; a window with more candidates reserves a segment of the overflow area
// behind the slots with ONE atomic on a never-reset device counter (the host knows its value) and writes its whole
// list there.  {start, length} per query say where the list is.  Best mode: first minimum over the candidates.
// Completion is the stream's: a variant whose last workgroup stamped a pinned flag for the host to spin on needed a
// system-scope fence per wave and took 32 us instead of 13 (MI355X, 1004 queries).
enum { kInline = 16 };

__global__ __launch_bounds__(256) void k_window_search(FrameDev F, const WinQuery *__restrict__ Q,
                                                       const uint8_t *__restrict__ qdesc, WinLaunch W,
                                                       int *__restrict__ off, int *__restrict__ cnt,
                                                       uint32_t *__restrict__ out, int *__restrict__ best,
                                                       uint32_t *__restrict__ counter) {
  const int lane = threadIdx.x & 63;
  const int q = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (q < W.nq) {
    const WinQuery wq = Q[q];
    uint32_t qd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (qdesc) {
      const uint32_t *p = (const uint32_t *)(qdesc + (size_t)q * 32);
#pragma unroll
      for (int k = 0; k < 8; k++) qd[k] = p[k];
    }
    const int right = wq.flags & 1;
    const int *cs = F.cell_start[right];
    const GridEnt *en = F.ent[right];
    const int koff = right ? F.nleft : 0;
    const float x = wq.x, y = wq.y, r = wq.r;
    // (int)floor((x - mnMinX - factorX) * mfGridElementWidthInv) etc. (Frame.cc:810-832): float arithmetic
    // (the int conversions behave like the reference's x86 build for NaN / out-of-range values: cvt_int_x86)
    const int nMinCellX = max(0, cvt_int_x86(floorf(fmul(fsub(fsub(x, F.minX), r), F.invW))));
    const int nMaxCellX = min(kGridCols - 1, cvt_int_x86(ceilf(fmul(fadd(fsub(x, F.minX), r), F.invW))));
    const int nMinCellY = max(0, cvt_int_x86(floorf(fmul(fsub(fsub(y, F.minY), r), F.invH))));
    const int nMaxCellY = min(kGridRows - 1, cvt_int_x86(ceilf(fmul(fadd(fsub(y, F.minY), r), F.invH))));
    const bool active = !(wq.flags & 2) && nMinCellX < kGridCols && nMaxCellX >= 0 && nMinCellY < kGridRows &&
                        nMaxCellY >= 0 && nMaxCellX >= nMinCellX && nMaxCellY >= nMinCellY;
    const int ncy = active ? nMaxCellY - nMinCellY + 1 : 1, ncell = active ? (nMaxCellX - nMinCellX + 1) * ncy : 0;
    const bool bCheckLevels = (wq.minL > 0) || (wq.maxL >= 0);
    // does this grid entry survive GetFeaturesInArea and the routine's static gates?
    auto pass = [&](const GridEnt &g) -> bool {
      const int oct = (int)(int16_t)(g.io >> 16);
      if (bCheckLevels) {
        if (oct < wq.minL) return false;
        if (wq.maxL >= 0 && oct > wq.maxL) return false;
      }
      const float distx = fsub(g.x, x), disty = fsub(g.y, y);
      if (!(fabsf(distx) < r && fabsf(disty) < r)) return false;
      if (wq.hi >= 0 && (oct < wq.lo || oct > wq.hi)) return false;
      if (W.gate_mode == kGateUr) {
        // F.Nleft == -1 && F.mvuRight[idx] > 0: er = fabs(ur - mvuRight[idx]); er > gate -> skip
        // (ORBmatcher.cc:97-102, 1741-1747)
        if (F.uright && F.nleft == -1) {
          const float uR = F.uright[g.io & 0xFFFFu];
          if (uR > 0 && fabsf(fsub(wq.ur, uR)) > wq.gate) return false;
        }
      } else if (W.gate_mode == kGateChi2) {
        // Fuse (ORBmatcher.cc:1267-1292): reprojection error against the keypoint, chi-square at the keypoint's level
        const float uR = F.uright ? F.uright[g.io & 0xFFFFu] : -1.0f;
        const float ex = fsub(x, g.x), ey = fsub(y, g.y);
        const float inv = W.inv_sigma2[oct & 15];
        if (uR >= 0) {
          const float er = fsub(wq.ur, uR);
          const float e2 = fadd(fadd(fmul(ex, ex), fmul(ey, ey)), fmul(er, er));
          if ((double)fmul(e2, inv) > 7.8) return false;
        } else {
          const float e2 = fadd(fmul(ex, ex), fmul(ey, ey));
          if ((double)fmul(e2, inv) > 5.99) return false;
        }
      }
      return true;
    };
    uint32_t bestKey = 0xFFFFFFFFu;
    int bestIdx = -1;
    // One walk over the window in the reference's candidate order (cells ix outer / iy inner = ascending lane, entries
    // in cell order).  `emit(pos, i, dist, oct)` receives every surviving candidate with its list position.
    auto walk_window = [&](auto emit) -> int {
      int count = 0;
      for (int c0 = 0; c0 < ncell; c0 += 64) {
        const int c = c0 + lane;
        int e0 = 0, e1 = 0;
        if (c < ncell) {
          const int cx = c / ncy, cy = c - cx * ncy;
          const int cell = (nMinCellX + cx) * kGridRows + nMinCellY + cy;
          e0 = cs[cell];
          e1 = cs[cell + 1];
        }
        // the filter runs once per entry: survivors are remembered as a bit mask (cells hold a handful of entries;
        // a chunk with a cell of more than 32 falls back to filtering twice)
        const bool big = __ballot(e1 - e0 > 32) != 0;
        uint32_t mask = 0;
        int mine = 0;
        if (!big) {
          for (int e = e0; e < e1; e++)
            if (pass(en[e])) mask |= 1u << (e - e0);
          mine = __popc(mask);
        } else {
          for (int e = e0; e < e1; e++) mine += pass(en[e]) ? 1 : 0;
        }
        const int incl = wave_incl_scan(mine);
        const int tot = __builtin_amdgcn_readlane(incl, 63);
        if (tot == 0) continue;
        int pos = count + incl - mine;
        auto one = [&](int e) {
          const GridEnt g = en[e];
          const int i = (int)(g.io & 0xFFFFu);
          int dist = 0;
          if (qdesc) {
            const uint4 *d = (const uint4 *)(F.desc + (size_t)(i + koff) * 32);
            const uint4 b0 = d[0], b1 = d[1];
            dist = __popc(qd[0] ^ b0.x) + __popc(qd[1] ^ b0.y) + __popc(qd[2] ^ b0.z) + __popc(qd[3] ^ b0.w) +
                   __popc(qd[4] ^ b1.x) + __popc(qd[5] ^ b1.y) + __popc(qd[6] ^ b1.z) + __popc(qd[7] ^ b1.w);
          }
          emit(pos, i, dist, (int)((g.io >> 16) & 15u));
          pos++;
        };
        if (!big) {
          while (mask) {
            one(e0 + __builtin_ctz(mask));
            mask &= mask - 1;
          }
        } else {
          for (int e = e0; e < e1; e++)
            if (pass(en[e])) one(e);
        }
        count += tot;
      }
      return count;
    };
    if (W.mode == kWinList) {
      uint32_t *slot = out + (size_t)q * kInline;
      const int total = walk_window([&](int pos, int i, int dist, int oct) {
        if (pos < kInline) slot[pos] = (uint32_t)i | ((uint32_t)dist << 15) | ((uint32_t)oct << 24);
      });
      int start = q * kInline;
      if (total > kInline) {  // the whole list goes to the overflow area
        int base = 0;
        if (lane == 0) base = (int)(atomicAdd(counter, (uint32_t)total) - W.counter_base);
        base = __builtin_amdgcn_readfirstlane(base);
        start = W.nq * kInline + base;
        uint32_t *seg = out + start;
        const int room = W.cap - base;
        walk_window([&](int pos, int i, int dist, int oct) {
          if (pos < room) seg[pos] = (uint32_t)i | ((uint32_t)dist << 15) | ((uint32_t)oct << 24);
        });
      }
      if (lane == 0) {
        off[q] = start;
        cnt[q] = total;
      }
    } else {
      walk_window([&](int pos, int i, int dist, int) {
        if (dist < W.best_init) {
          const uint32_t key = ((uint32_t)dist << 16) | (uint32_t)pos;  // first minimum in candidate order
          if (key < bestKey) {
            bestKey = key;
            bestIdx = i;
          }
        }
      });
      uint32_t k = bestKey;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) k = min(k, __shfl_xor(k, d));
      const uint64_t owner = __ballot(bestKey == k && bestIdx >= 0);
      int idx = -1, dist = W.best_init;
      if (k != 0xFFFFFFFFu && owner) {
        idx = __builtin_amdgcn_readlane(bestIdx, __builtin_ctzll(owner));
        dist = (int)(k >> 16);
      }
      if (lane == 0) {
        best[2 * q] = idx < 0 ? -1 : idx + koff;  // index into mDescriptors (ORBmatcher.cc:1294-1295: idx += NLeft)
        best[2 * q + 1] = dist;
      }
    }
  }
}

thread_local int t_cap_hint = 0;  // entries per query the compact candidate array is sized for (sticky, grows)

// where the last window call of this thread spent its wall time (vsg_debug_call_profile): a handful of clock reads
struct CallProf {
  double t0 = 0, fill = 0, launch = 0, sync = 0, total = 0;
};
thread_local CallProf t_prof;
inline double now_us() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

}  // namespace

namespace vsg {

FrameDev frame_dev(const vsg_frame *f) {
  FrameDev d;
  d.kps = f->d_kps;
  d.desc = f->d_desc;
  d.uright = f->has_uright ? f->d_uright : nullptr;
  for (int g = 0; g < 2; g++) {
    d.cell_start[g] = f->d_cell_start[g];
    d.ent[g] = f->d_ent[g];
  }
  d.n = f->n;
  d.nleft = f->nleft;
  d.minX = f->minX, d.minY = f->minY, d.invW = f->invW, d.invH = f->invH;
  return d;
}

int WindowCall::begin(int device, int nq_, int mode_, bool with_desc_, size_t arena_base, size_t arena_extra) {
  int rc = VSG_OK;
  if (range_open) range_pop();  // a retry re-enters begin()
  range_push("ORBmatcher window search");
  range_open = true;
  if (arena_base == 0) t_prof.t0 = now_us();
  c = thread_ctx(device, &rc);
  if (!c) return rc;
  nq = nq_, mode = mode_, with_desc = with_desc_, base = arena_base;
  const size_t Q = (size_t)(nq > 0 ? nq : 1);
  if (t_cap_hint < 4) t_cap_hint = 4;
  cap = mode == kWinList ? (int)(Q * (size_t)t_cap_hint) : 0;  // entries of the overflow area behind the inline slots
  Stage st;
  oQ = st.add(Q * sizeof(WinQuery));
  oD = st.add(with_desc ? Q * 32 : 0);
  oOff = st.add(mode == kWinList ? Q * 4 : 0);
  oCnt = st.add(mode == kWinList ? Q * 4 : 0);
  oOut = st.add(mode == kWinList ? (Q * kInline + (size_t)cap) * 4 : Q * 8);
  return ctx_reserve(c, base + st.total + arena_extra, 0);
}

WindowCall::~WindowCall() {
  if (range_open) range_pop();  // an error return between begin() and finish()
}

size_t WindowCall::bytes() const {
  const size_t Q = (size_t)(nq > 0 ? nq : 1);
  return oOut + (((mode == kWinList ? (Q * kInline + (size_t)cap) * 4 : Q * 8) + 63) & ~(size_t)63);
}

int WindowCall::launch(const vsg_frame *f, int gate_mode_, int best_init_, const float *inv_sigma2_, int nlevels,
                       const uint8_t *qdesc_dev) {
  frame = f;
  gate_mode = gate_mode_, best_init = best_init_;
  const double tl = now_us();
  if (base == 0) t_prof.fill = tl - t_prof.t0;
  if (nq <= 0) return VSG_OK;
  WinLaunch W;
  W.nq = nq, W.mode = mode, W.gate_mode = gate_mode, W.best_init = best_init, W.cap = cap;
  W.counter_base = c->counter_base;
  for (int l = 0; l < 16; l++) W.inv_sigma2[l] = (inv_sigma2_ && l < nlevels) ? inv_sigma2_[l] : 0.f;
  uint8_t *d = c->d_pin + base;
  hipLaunchKernelGGL(k_window_search, dim3((nq + 3) / 4), dim3(256), 0, c->stream, frame_dev(f),
                     (const WinQuery *)(d + oQ),
                     qdesc_dev ? qdesc_dev : with_desc ? (const uint8_t *)(d + oD) : (const uint8_t *)nullptr, W,
                     (int *)(d + oOff), (int *)(d + oCnt), (uint32_t *)(d + oOut), (int *)(d + oOut), c->d_counter);
  t_prof.launch = now_us() - tl;
  return hipGetLastError() == hipSuccess ? VSG_OK : VSG_ERR_HIP;
}

int WindowCall::finish() {
  const double ts = now_us();
  if (nq > 0 && hipStreamSynchronize(c->stream) != hipSuccess) return VSG_ERR_HIP;
  t_prof.sync = now_us() - ts;
  if (range_open) range_pop(), range_open = false;
  if (mode != kWinList) return VSG_OK;
  const int32_t *cn = (const int32_t *)(c->h_pin + base + oCnt);
  long long total = 0;  // entries of the lists that went to the overflow area = what the waves added to the counter
  for (int q = 0; q < nq; q++) total += cn[q] > kInline ? cn[q] : 0;
  c->counter_base += (uint32_t)total;
  if (total <= cap) return VSG_OK;
  const long long per = (2 * total + nq - 1) / (nq > 0 ? nq : 1);
  t_cap_hint = (int)(per > t_cap_hint ? per : 2 * t_cap_hint);  // sticky: room for windows like these from now on
  return VSG_RETRY;
}

walk::CandView WindowCall::lists() const {
  walk::CandView cv;
  cv.ent = (const uint32_t *)(c->h_pin + base + oOut);
  cv.off = (const int32_t *)(c->h_pin + base + oOff);
  cv.cnt = (const int32_t *)(c->h_pin + base + oCnt);
  return cv;
}

}  // namespace vsg

namespace {

#define F_TRY(expr)                               \
  do {                                            \
    if ((expr) != hipSuccess) return VSG_ERR_HIP; \
  } while (0)

// device layout of a frame for `cap` features
struct FrameLayout {
  size_t oK, oD, oU, oCS0, oE0, oCS1, oE1, oFvH, oFvN, oFvO, oFvI, total;
  explicit FrameLayout(int cap) {
    Stage st;
    const size_t C = (size_t)cap + 1;
    oK = st.add(C * sizeof(KeyPointPOD));
    oD = st.add(C * 32);
    oU = st.add(C * 4);
    oCS0 = st.add((kGridCells + 1) * 4);
    oE0 = st.add(C * sizeof(GridEnt));
    oCS1 = st.add((kGridCells + 1) * 4);
    oE1 = st.add(C * sizeof(GridEnt));
    oFvH = st.add(64), oFvN = st.add(C * 4), oFvO = st.add((C + 1) * 4), oFvI = st.add(C * 4);  // Frame::mFeatVec
    total = st.total;
  }
};

// Frame::AssignFeaturesToGrid (Frame.cc:521-553) on the host for keys [i0, i0 + n): stable bucket fill with the
// reference's float operations (PosInGrid, Frame.cc:870-880; libm round = half away from zero)
void host_grid(const vsg_keypoint *kps, int i0, int n, float minX, float minY, float invW, float invH, int *cell_start,
               GridEnt *ent) {
  std::vector<int16_t> cell_of((size_t)n + 1);
  std::vector<int> cnt(kGridCells, 0);
  for (int i = 0; i < n; i++) {
    const int px = cvt_int_x86(roundf(fmul(fsub(kps[i0 + i].x, minX), invW)));
    const int py = cvt_int_x86(roundf(fmul(fsub(kps[i0 + i].y, minY), invH)));
    const bool in = !(px < 0 || px >= kGridCols || py < 0 || py >= kGridRows);
    cell_of[i] = in ? (int16_t)(px * kGridRows + py) : (int16_t)-1;
    if (in) cnt[px * kGridRows + py]++;
  }
  int run = 0;
  for (int c = 0; c < kGridCells; c++) {
    cell_start[c] = run;
    run += cnt[c];
    cnt[c] = cell_start[c];
  }
  cell_start[kGridCells] = run;
  for (int i = 0; i < n; i++)
    if (cell_of[i] >= 0) {  // insertion order == ascending keypoint index
      const vsg_keypoint &k = kps[i0 + i];
      ent[cnt[cell_of[i]]++] = {k.x, k.y, (uint32_t)i | ((uint32_t)(k.octave & 0xFFFF) << 16)};
    }
}

int frame_check(const vsg_frame *f) { return f && f->d_block ? VSG_OK : VSG_ERR_INVALID; }

void set_bounds(vsg_frame *f, float min_x, float min_y, float max_x, float max_y) {
  f->minX = min_x, f->minY = min_y, f->maxX = max_x, f->maxY = max_y;
  // mfGridElementWidthInv = FRAME_GRID_COLS / (mnMaxX - mnMinX)   (Frame.cc:378-379)
  f->invW = (float)kGridCols / (max_x - min_x);
  f->invH = (float)kGridRows / (max_y - min_y);
}

// retry loop around a window-search entry point body
template <class Body>
int with_retry(Body body) {
  for (int attempt = 0; attempt < 8; attempt++) {
    const int rc = body();
    if (rc != VSG_RETRY) {
      t_prof.total = now_us() - t_prof.t0;
      return rc;
    }
  }
  return VSG_ERR_CAPACITY;
}

inline float radius_by_viewing_cos(float viewCos) {  // ORBmatcher::RadiusByViewingCos (ORBmatcher.cc:218-224)
  if (viewCos > 0.998) return 2.5f;
  return 4.0f;
}

}  // namespace

extern "C" {

// debug: wall time (microseconds) of the calling thread's last windowed search: filling the pinned arena, the kernel
// launch call, the wait for completion (= kernel + PCIe), and the whole entry point (the rest is the ordered host pass)
int vsg_debug_call_profile(float us[4]) {
  if (!us) return VSG_ERR_INVALID;
  us[0] = (float)t_prof.fill, us[1] = (float)t_prof.launch, us[2] = (float)t_prof.sync, us[3] = (float)t_prof.total;
  return VSG_OK;
}

int vsg_frame_create(int device, int capacity, vsg_frame **out) {
  if (!out || capacity < 1 || capacity > 32767) return VSG_ERR_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return VSG_ERR_NO_DEVICE;
  F_TRY(hipSetDevice(device));
  vsg_frame *f = new vsg_frame();
  f->device = device;
  f->capacity = capacity;
  const FrameLayout L(capacity);
  // zeroed: a frame that was created but never uploaded has n = 0 AND all-zero cell_start arrays, so a search on it
  // walks empty [0, 0) entry ranges instead of whatever the allocation held.  The fill runs on the calling thread's own
  // stream and is WAITED for: uploads and searches use non-blocking streams, which the NULL stream's hipMemset is not
  // ordered against (it could land after an upload and wipe it).
  int rc = VSG_OK;
  ThreadCtx *c = thread_ctx(device, &rc);
  if (!c) {
    delete f;
    return rc;
  }
  if (hipMalloc((void **)&f->d_block, L.total) != hipSuccess ||
      hipMemsetAsync(f->d_block, 0, L.total, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
    if (f->d_block) hipFree(f->d_block);
    delete f;
    return VSG_ERR_HIP;
  }
  f->d_kps = (KeyPointPOD *)(f->d_block + L.oK);
  f->d_desc = f->d_block + L.oD;
  f->d_uright = (float *)(f->d_block + L.oU);
  f->d_cell_start[0] = (int *)(f->d_block + L.oCS0);
  f->d_ent[0] = (GridEnt *)(f->d_block + L.oE0);
  f->d_cell_start[1] = (int *)(f->d_block + L.oCS1);
  f->d_ent[1] = (GridEnt *)(f->d_block + L.oE1);
  f->d_fv_hdr = (int *)(f->d_block + L.oFvH), f->d_fv_node = (int *)(f->d_block + L.oFvN);
  f->d_fv_off = (int *)(f->d_block + L.oFvO), f->d_fv_idx = (int *)(f->d_block + L.oFvI);
  *out = f;
  return VSG_OK;
}

void vsg_frame_destroy(vsg_frame *f) {
  if (!f) return;
  hipSetDevice(f->device);
  hipFree(f->d_block);
  delete f;
}

int vsg_frame_size(const vsg_frame *f) { return f ? f->n : VSG_ERR_INVALID; }

int vsg_frame_upload(vsg_frame *f, const vsg_keypoint *keys, const uint8_t *desc, const float *u_right, int n,
                     int nleft, float min_x, float min_y, float max_x, float max_y) {
  if (frame_check(f) != VSG_OK || n < 0 || n > f->capacity || (n > 0 && (!keys || !desc)) || nleft < -1 || nleft > n)
    return VSG_ERR_INVALID;
  // packed candidate entries carry the octave in 4 bits (index : 15 | distance : 9 | octave : 4, vsg_walks.h)
  for (int i = 0; i < n; i++)
    if (keys[i].octave < 0 || keys[i].octave > 15) return VSG_ERR_UNSUPPORTED;
  int rc = VSG_OK;
  ThreadCtx *c = thread_ctx(f->device, &rc);
  if (!c) return rc;
  const FrameLayout L(f->capacity);
  rc = ctx_reserve(c, L.total, 0);
  if (rc != VSG_OK) return rc;
  set_bounds(f, min_x, min_y, max_x, max_y);
  f->n = n, f->nleft = nleft, f->has_uright = u_right != nullptr, f->fv_valid = false;
  f->h_kps.assign(keys, keys + n);
  // the whole device image of the frame is assembled in the pinned arena and goes up in ONE DMA
  uint8_t *h = c->h_pin;
  if (n) memcpy(h + L.oK, keys, (size_t)n * sizeof(vsg_keypoint));
  if (n) memcpy(h + L.oD, desc, (size_t)n * 32);
  if (u_right && n) memcpy(h + L.oU, u_right, (size_t)n * 4);
  const int nl = nleft == -1 ? n : nleft;
  host_grid(keys, 0, nl, f->minX, f->minY, f->invW, f->invH, (int *)(h + L.oCS0), (GridEnt *)(h + L.oE0));
  if (nleft != -1)
    host_grid(keys, nleft, n - nleft, f->minX, f->minY, f->invW, f->invH, (int *)(h + L.oCS1), (GridEnt *)(h + L.oE1));
  else
    memset(h + L.oCS1, 0, (kGridCells + 1) * 4);
  // only the used part of every block travels: [keys | desc | uright | grid] are contiguous up to the right grid
  const size_t used = nleft != -1 ? L.oE1 + (size_t)(n - nleft + 1) * sizeof(GridEnt) : L.oCS1 + (kGridCells + 1) * 4;
  F_TRY(hipMemcpyAsync(f->d_block, h, used, hipMemcpyHostToDevice, c->stream));
  F_TRY(hipStreamSynchronize(c->stream));  // the frame may be searched from any thread from now on
  return VSG_OK;
}

// one launch: [undistortion +] grid from the keypoints where they already are (the extractor's output) + the two record
// copies + an empty right-camera grid
static int frame_from_extractor(vsg_frame *f, vsg_orb *h, int index, const vsg_keypoint *kps_host, int n,
                                const CamModel &cam, float min_x, float min_y, float max_x, float max_y,
                                vsg_keypoint *keys_un_out) {
  OrbOutputView v;
  int rc = vsg_orb_output_view(h, index, &v);
  if (rc != VSG_OK) return rc;
  if (v.device != f->device) return VSG_ERR_INVALID;
  ThreadCtx *c = thread_ctx(f->device, &rc);
  if (!c) return rc;
  KeyPointPOD *un_pin = nullptr, *un_dev = nullptr;
  if (cam.distorted) {  // mvKeysUn comes back through the calling thread's pinned arena (written by the kernel itself)
    rc = ctx_reserve(c, (size_t)(n + 1) * sizeof(KeyPointPOD), 0);
    if (rc != VSG_OK) return rc;
    un_pin = (KeyPointPOD *)c->h_pin, un_dev = (KeyPointPOD *)c->d_pin;
  }
  set_bounds(f, min_x, min_y, max_x, max_y);
  f->n = n, f->nleft = -1, f->has_uright = false, f->fv_valid = false;
  if (v.done) F_TRY(hipStreamWaitEvent(c->stream, v.done, 0));
  hipLaunchKernelGGL(k_frame_grid_build, dim3(1), dim3(1024), 0, c->stream, v.d_kps, 0, n, f->minX, f->minY, f->invW,
                     f->invH, f->d_cell_start[0], f->d_ent[0], f->d_kps, v.d_desc, f->d_desc, f->d_cell_start[1], cam,
                     un_dev, (const int *)nullptr, 0);
  F_TRY(hipGetLastError());
  if (!cam.distorted) f->h_kps.assign(kps_host, kps_host + n);  // beside the kernel
  F_TRY(hipStreamSynchronize(c->stream));
  if (cam.distorted) {
    f->h_kps.assign((const vsg_keypoint *)un_pin, (const vsg_keypoint *)un_pin + n);
    if (keys_un_out && n) memcpy(keys_un_out, un_pin, (size_t)n * sizeof(vsg_keypoint));
  } else if (keys_un_out && n && keys_un_out != kps_host) {
    memcpy(keys_un_out, kps_host, (size_t)n * sizeof(vsg_keypoint));  // mvKeysUn = mvKeys (Frame.cc:893-897)
  }
  return VSG_OK;
}

int vsg_frame_from_extractor(vsg_frame *f, vsg_orb *h, int index, const vsg_keypoint *kps_host, int n, float min_x,
                             float min_y, float max_x, float max_y) {
  if (frame_check(f) != VSG_OK || !h || n < 0 || n > f->capacity || (n > 0 && !kps_host)) return VSG_ERR_INVALID;
  CamModel cam = {};
  return frame_from_extractor(f, h, index, kps_host, n, cam, min_x, min_y, max_x, max_y, nullptr);
}

// ---- Frame::Frame(...) in ONE call and ONE wait: ExtractORB (operator()) -> UndistortKeyPoints -> AssignFeaturesToGrid
// (Frame.cc:344-358 for RGB-D).  The grid launch is enqueued on the extractor's stream right behind its stage chain -- it
// reads the keypoint count from the device -- so the blocking call's single wait covers both, instead of
// operator() [wait] -> vsg_frame_from_extractor* [launch, wait].
namespace {
struct ToFrameHook {
  vsg_frame *f;
  CamModel cam;
  KeyPointPOD *un_dev;
};
int to_frame_hook(void *ctx, hipStream_t s, const OrbOutputView &v) {
  const ToFrameHook *H = (const ToFrameHook *)ctx;
  vsg_frame *f = H->f;
  hipLaunchKernelGGL(k_frame_grid_build, dim3(1), dim3(1024), 0, s, v.d_kps, 0, 0, f->minX, f->minY, f->invW, f->invH,
                     f->d_cell_start[0], f->d_ent[0], f->d_kps, v.d_desc, f->d_desc, f->d_cell_start[1], H->cam, H->un_dev,
                     v.d_counts, f->capacity);
  return hipGetLastError() == hipSuccess ? VSG_OK : VSG_ERR_HIP;
}
}  // namespace

int vsg_orb_extract_to_frame(vsg_orb *h, const uint8_t *gray, int rows, int cols, int stride, int lap0, int lap1,
                             vsg_keypoint *kps, uint8_t *desc, int capacity, int *n, vsg_frame *f, const float K4[4],
                             const float *dist, int ndist, float min_x, float min_y, float max_x, float max_y,
                             vsg_keypoint *keys_un_out) {
  if (n) *n = 0;
  if (frame_check(f) != VSG_OK || !h || !kps || !desc || !n) return VSG_ERR_INVALID;
  // the hook launches on the extractor's stream with the frame's device pointers: one device for both
  if (vsg_orb_device_of(h) != f->device) return VSG_ERR_INVALID;
  ToFrameHook H;
  H.f = f, H.un_dev = nullptr;
  H.cam = CamModel();
  if (K4 && !make_cam_model(K4, dist, ndist, &H.cam)) return VSG_ERR_INVALID;
  int rc = VSG_OK;
  ThreadCtx *c = thread_ctx(f->device, &rc);
  if (!c) return rc;
  KeyPointPOD *un_pin = nullptr;
  if (H.cam.distorted) {
    rc = ctx_reserve(c, (size_t)(f->capacity + 1) * sizeof(KeyPointPOD), 0);
    if (rc != VSG_OK) return rc;
    un_pin = (KeyPointPOD *)c->h_pin, H.un_dev = (KeyPointPOD *)c->d_pin;
  }
  set_bounds(f, min_x, min_y, max_x, max_y);
  vsg_orb_set_post_chain(h, to_frame_hook, &H);
  const int mono = vsg_orb_extract(h, gray, rows, cols, stride, lap0, lap1, kps, desc, capacity, n);
  vsg_orb_set_post_chain(h, nullptr, nullptr);  // (a call that failed before its submit leaves the hook unconsumed)
  if (mono < 0 || *n > f->capacity) {
    // the bounds are already the new ones and the hook may have rewritten the device arrays: the frame holds nothing
    // searchable any more, and says so
    f->n = 0, f->nleft = -1, f->has_uright = false, f->fv_valid = false;
    f->h_kps.clear();
    return mono < 0 ? mono : VSG_ERR_CAPACITY;
  }
  f->n = *n, f->nleft = -1, f->has_uright = false, f->fv_valid = false;
  if (H.cam.distorted) {
    f->h_kps.assign((const vsg_keypoint *)un_pin, (const vsg_keypoint *)un_pin + *n);
    if (keys_un_out && *n) memcpy(keys_un_out, un_pin, (size_t)*n * sizeof(vsg_keypoint));
  } else {
    f->h_kps.assign(kps, kps + *n);
    if (keys_un_out && *n && keys_un_out != kps) memcpy(keys_un_out, kps, (size_t)*n * sizeof(vsg_keypoint));
  }
  return mono;
}

int vsg_camera_image_bounds(int cols, int rows, const float K4[4], const float *dist, int ndist, float out[4]) {
  CamModel cam;
  if (!out || cols < 1 || rows < 1 || !make_cam_model(K4, dist, ndist, &cam)) return VSG_ERR_INVALID;
  if (!cam.distorted) {  // Frame.cc:948-954
    out[0] = 0.0f, out[1] = 0.0f, out[2] = (float)cols, out[3] = (float)rows;
    return VSG_OK;
  }
  // the four corners through cv::undistortPoints (Frame.cc:928-946): four points, once per camera -- host arithmetic,
  // the same source the device runs per keypoint
  float x[4], y[4];
  const float cx[4] = {0.f, (float)cols, 0.f, (float)cols}, cy[4] = {0.f, 0.f, (float)rows, (float)rows};
  for (int i = 0; i < 4; i++) undistort_point(cam, cx[i], cy[i], &x[i], &y[i]);
  out[0] = std::min(x[0], x[2]);  // mnMinX = min(mat(0,0), mat(2,0))
  out[2] = std::max(x[1], x[3]);  // mnMaxX = max(mat(1,0), mat(3,0))
  out[1] = std::min(y[0], y[1]);  // mnMinY = min(mat(0,1), mat(1,1))
  out[3] = std::max(y[2], y[3]);  // mnMaxY = max(mat(2,1), mat(3,1))
  return VSG_OK;
}

int vsg_frame_from_extractor_undistort(vsg_frame *f, vsg_orb *h, int index, const vsg_keypoint *kps_host, int n,
                                       const float K4[4], const float *dist, int ndist, float min_x, float min_y,
                                       float max_x, float max_y, vsg_keypoint *keys_un_out) {
  if (frame_check(f) != VSG_OK || !h || n < 0 || n > f->capacity || (n > 0 && !kps_host)) return VSG_ERR_INVALID;
  CamModel cam;
  if (!make_cam_model(K4, dist, ndist, &cam)) return VSG_ERR_INVALID;
  return frame_from_extractor(f, h, index, kps_host, n, cam, min_x, min_y, max_x, max_y, keys_un_out);
}

int vsg_frame_copy_grid(vsg_frame *f, int right, int32_t *cell_start, int32_t *entries) {
  if (frame_check(f) != VSG_OK || !cell_start || !entries || right < 0 || right > 1) return VSG_ERR_INVALID;
  int rc = VSG_OK;
  ThreadCtx *c = thread_ctx(f->device, &rc);
  if (!c) return rc;
  rc = ctx_reserve(c, (kGridCells + 1) * 4 + (size_t)(f->capacity + 1) * sizeof(GridEnt) + 128, 0);
  if (rc != VSG_OK) return rc;
  int *hcs = (int *)c->h_pin;
  GridEnt *he = (GridEnt *)(c->h_pin + (((kGridCells + 1) * 4 + 63) & ~63));
  F_TRY(hipMemcpyAsync(hcs, f->d_cell_start[right], (kGridCells + 1) * 4, hipMemcpyDeviceToHost, c->stream));
  F_TRY(hipMemcpyAsync(he, f->d_ent[right], (size_t)f->capacity * sizeof(GridEnt), hipMemcpyDeviceToHost, c->stream));
  F_TRY(hipStreamSynchronize(c->stream));
  memcpy(cell_start, hcs, (kGridCells + 1) * 4);
  const int ne = hcs[kGridCells];
  if (ne < 0 || ne > f->capacity) return VSG_ERR_HIP;
  for (int i = 0; i < ne; i++) entries[i] = (int)(he[i].io & 0xFFFFu);
  return ne;
}

int vsg_frame_features_in_area(vsg_frame *f, const float *x, const float *y, const float *r, const int32_t *min_level,
                               const int32_t *max_level, int right, int nq, int32_t *cand_off, int32_t *cand_idx,
                               int cap) {
  if (frame_check(f) != VSG_OK || !x || !y || !r || !cand_off || nq < 0 || cap < 0) return VSG_ERR_INVALID;
  cand_off[0] = 0;
  if (nq == 0) return 0;
  return with_retry([&]() -> int {
    WindowCall wc;
    int rc = wc.begin(f->device, nq, kWinList, false);
    if (rc != VSG_OK) return rc;
    WinQuery *Q = wc.queries();
    for (int q = 0; q < nq; q++) {
      WinQuery w = {x[q], y[q], r[q], min_level ? min_level[q] : -1, max_level ? max_level[q] : -1, 0, -1, 0.f, 0.f,
                    right ? 1 : 0, 0, 0};
      Q[q] = w;
    }
    rc = wc.launch(f, kGateNone, 256, nullptr, 0);
    if (rc != VSG_OK) return rc;
    rc = wc.finish();
    if (rc != VSG_OK) return rc;
    const walk::CandView cv = wc.lists();
    int total = 0;
    for (int q = 0; q < nq; q++) {
      const int n = cv.size(q);
      const uint32_t *e = cv.begin(q);
      for (int k = 0; k < n; k++, total++)
        if (cand_idx && total < cap) cand_idx[total] = walk::ent_idx(e[k]);
      cand_off[q + 1] = total;
    }
    return total;
  });
}

int vsg_frame_search_by_projection(vsg_frame *F, int n_mp, const uint8_t *mp_desc, const uint8_t *mp_observed,
                                   const uint8_t *in_view, const float *proj_x, const float *proj_y,
                                   const float *proj_xr, const int32_t *scale_level, const float *view_cos,
                                   const uint8_t *in_view_r, const float *proj_x_r, const float *proj_y_r,
                                   const int32_t *scale_level_r, const float *view_cos_r, float th, float nnratio,
                                   const float *scale_factors, int nlevels, const int32_t *left_to_right,
                                   const int32_t *right_to_left, uint8_t *train_blocked, int32_t *train_match) {
  if (frame_check(F) != VSG_OK || n_mp < 0 || !train_blocked || !train_match || !scale_factors || nlevels < 1)
    return VSG_ERR_INVALID;
  if (n_mp == 0) return 0;
  if (!mp_desc || !in_view || !proj_x || !proj_y || !scale_level || !view_cos) return VSG_ERR_INVALID;
  const bool stereo2 = F->nleft != -1;
  if (stereo2 && in_view_r && (!proj_x_r || !proj_y_r || !scale_level_r || !view_cos_r)) return VSG_ERR_INVALID;
  const bool bFactor = th != 1.0;  // :46
  return with_retry([&]() -> int {
    const int nq = stereo2 ? 2 * n_mp : n_mp;
    WindowCall wc;
    int rc = wc.begin(F->device, nq, kWinList, true);
    if (rc != VSG_OK) return rc;
    WinQuery *Q = wc.queries();
    uint8_t *D = wc.desc();
    for (int i = 0; i < n_mp; i++) {
      WinQuery w = {0, 0, 0, -1, -1, 0, -1, 0.f, 0.f, 2, 0, 0};
      if (in_view[i]) {
        const int lvl = scale_level[i];
        if (lvl < 0 || lvl >= nlevels) return VSG_ERR_INVALID;
        float r = radius_by_viewing_cos(view_cos[i]);  // :64
        if (bFactor) r *= th;                          // :66-67
        const float win = r * scale_factors[lvl];
        // GetFeaturesInArea(mTrackProjX, mTrackProjY, r * mvScaleFactors[level], level - 1, level)  (:69-70)
        w = {proj_x[i], proj_y[i], win, lvl - 1, lvl, 0, -1, proj_xr ? proj_xr[i] : 0.f, win, 0, 0, 0};
      }
      Q[i] = w;
      memcpy(D + (size_t)i * 32, mp_desc + (size_t)i * 32, 32);
      if (stereo2) {
        WinQuery wr = {0, 0, 0, -1, -1, 0, -1, 0.f, 0.f, 3, 0, 0};
        if (in_view_r && in_view_r[i] && scale_level_r[i] != -1) {
          const int lvl = scale_level_r[i];
          if (lvl < 0 || lvl >= nlevels) return VSG_ERR_INVALID;
          const float r = radius_by_viewing_cos(view_cos_r[i]);  // :151 (no th factor in the right block)
          wr = {proj_x_r[i], proj_y_r[i], r * scale_factors[lvl], lvl - 1, lvl, 0, -1, 0.f, 0.f, 1, 0, 0};
        }
        Q[n_mp + i] = wr;
        memcpy(D + (size_t)(n_mp + i) * 32, mp_desc + (size_t)i * 32, 32);
      }
    }
    // the stereo gate of :97-102 applies to F.Nleft == -1 frames with mvuRight
    rc = wc.launch(F, (!stereo2 && F->has_uright && proj_xr) ? kGateUr : kGateNone, 256, nullptr, 0);
    if (rc != VSG_OK) return rc;
    rc = wc.finish();
    if (rc != VSG_OK) return rc;
    return walk::search_local(wc.lists(), n_mp, F->nleft, in_view, in_view_r, scale_level_r, mp_observed, nnratio,
                              left_to_right, right_to_left, train_blocked, train_match);
  });
}

int vsg_frame_search_by_projection_last(vsg_frame *cur, int n_q, const uint8_t *mp_desc, const uint8_t *mp_observed,
                                        const float *u, const float *v, const float *ur, const float *u_r,
                                        const float *v_r, const int32_t *last_octave, const float *last_angle,
                                        float th, int direction, const float *scale_factors, int nlevels,
                                        int check_orientation, uint8_t *train_blocked, int32_t *train_match) {
  if (frame_check(cur) != VSG_OK || n_q < 0 || !train_blocked || !train_match || !scale_factors || nlevels < 1 ||
      direction < 0 || direction > 2)
    return VSG_ERR_INVALID;
  if (n_q == 0) return 0;
  if (!mp_desc || !u || !v || !last_octave || (check_orientation && !last_angle)) return VSG_ERR_INVALID;
  const bool stereo2 = cur->nleft != -1;
  if (stereo2 && (!u_r || !v_r)) return VSG_ERR_INVALID;
  return with_retry([&]() -> int {
    const int nq = stereo2 ? 2 * n_q : n_q;
    WindowCall wc;
    int rc = wc.begin(cur->device, nq, kWinList, true);
    if (rc != VSG_OK) return rc;
    WinQuery *Q = wc.queries();
    uint8_t *D = wc.desc();
    for (int i = 0; i < n_q; i++) {
      const int oct = last_octave[i];
      if (oct < 0 || oct >= nlevels) return VSG_ERR_INVALID;
      const float radius = th * scale_factors[oct];  // :1714
      // level window (:1718-1723): forward -> (nLastOctave, -1), backward -> (0, nLastOctave), else +-1
      const int minL = direction == 1 ? oct : direction == 2 ? 0 : oct - 1;
      const int maxL = direction == 1 ? -1 : direction == 2 ? oct : oct + 1;
      WinQuery w = {u[i], v[i], radius, minL, maxL, 0, -1, ur ? ur[i] : 0.f, radius, 0, 0, 0};
      Q[i] = w;
      memcpy(D + (size_t)i * 32, mp_desc + (size_t)i * 32, 32);
      if (stereo2) {
        WinQuery wr = {u_r[i], v_r[i], radius, minL, maxL, 0, -1, 0.f, 0.f, 1, 0, 0};  // :1797-1803
        Q[n_q + i] = wr;
        memcpy(D + (size_t)(n_q + i) * 32, mp_desc + (size_t)i * 32, 32);
      }
    }
    rc = wc.launch(cur, (!stereo2 && cur->has_uright && ur) ? kGateUr : kGateNone, 256, nullptr, 0);
    if (rc != VSG_OK) return rc;
    rc = wc.finish();
    if (rc != VSG_OK) return rc;
    const vsg_keypoint *hk = cur->h_kps.data();
    return walk::search_last(wc.lists(), n_q, cur->nleft, last_angle, mp_observed, [&](int i) { return hk[i].angle; },
                             walk::TH_HIGH, check_orientation != 0, train_blocked, train_match);
  });
}

int vsg_frame_search_by_projection_sim3(vsg_frame *kf, int n_q, const uint8_t *mp_desc, const float *u, const float *v,
                                        const float *radius, const int32_t *predicted_level, float ratio_hamming,
                                        int32_t *matched) {
  if (frame_check(kf) != VSG_OK || n_q < 0 || !matched) return VSG_ERR_INVALID;
  if (n_q == 0) return 0;
  if (!mp_desc || !u || !v || !radius || !predicted_level) return VSG_ERR_INVALID;
  return with_retry([&]() -> int {
    WindowCall wc;
    int rc = wc.begin(kf->device, n_q, kWinList, true);
    if (rc != VSG_OK) return rc;
    WinQuery *Q = wc.queries();
    for (int i = 0; i < n_q; i++) {
      // pKF->GetFeaturesInArea(u, v, radius) (:485); kpLevel in [nPredictedLevel - 1, nPredictedLevel] (:506-509)
      WinQuery w = {u[i], v[i], radius[i], -1, -1, predicted_level[i] - 1, predicted_level[i], 0.f, 0.f, 0, 0, 0};
      if (predicted_level[i] < 0) w.flags = 2;
      Q[i] = w;
    }
    memcpy(wc.desc(), mp_desc, (size_t)n_q * 32);
    rc = wc.launch(kf, kGateNone, 256, nullptr, 0);
    if (rc != VSG_OK) return rc;
    rc = wc.finish();
    if (rc != VSG_OK) return rc;
    return walk::search_sim3_projection(wc.lists(), n_q, ratio_hamming, matched);
  });
}

int vsg_frame_search_by_projection_kf(vsg_frame *cur, int n_q, const uint8_t *mp_desc, const float *u, const float *v,
                                      const float *radius, const int32_t *predicted_level, const float *kf_angle,
                                      int orb_dist, int check_orientation, uint8_t *occupied, int32_t *train_match) {
  if (frame_check(cur) != VSG_OK || n_q < 0 || !occupied || !train_match) return VSG_ERR_INVALID;
  if (n_q == 0) return 0;
  if (!mp_desc || !u || !v || !radius || !predicted_level || (check_orientation && !kf_angle)) return VSG_ERR_INVALID;
  return with_retry([&]() -> int {
    WindowCall wc;
    int rc = wc.begin(cur->device, n_q, kWinList, true);
    if (rc != VSG_OK) return rc;
    WinQuery *Q = wc.queries();
    for (int i = 0; i < n_q; i++) {
      // GetFeaturesInArea(u, v, radius, nPredictedLevel - 1, nPredictedLevel + 1) (:1934)
      WinQuery w = {u[i], v[i], radius[i], predicted_level[i] - 1, predicted_level[i] + 1, 0, -1, 0.f, 0.f, 0, 0, 0};
      Q[i] = w;
    }
    memcpy(wc.desc(), mp_desc, (size_t)n_q * 32);
    rc = wc.launch(cur, kGateNone, 256, nullptr, 0);
    if (rc != VSG_OK) return rc;
    rc = wc.finish();
    if (rc != VSG_OK) return rc;
    const vsg_keypoint *hk = cur->h_kps.data();
    return walk::search_kf_projection(wc.lists(), n_q, kf_angle, [&](int i) { return hk[i].angle; }, orb_dist,
                                      check_orientation != 0, occupied, train_match);
  });
}

int vsg_frame_search_by_sim3(vsg_frame *kf1, vsg_frame *kf2, int nq1, const int32_t *idx1, const uint8_t *desc1,
                             const float *u1, const float *v1, const float *radius1, const int32_t *level1, int nq2,
                             const int32_t *idx2, const uint8_t *desc2, const float *u2, const float *v2,
                             const float *radius2, const int32_t *level2, int32_t *matches12) {
  if (frame_check(kf1) != VSG_OK || frame_check(kf2) != VSG_OK || kf1->device != kf2->device || nq1 < 0 || nq2 < 0 ||
      !matches12)
    return VSG_ERR_INVALID;
  if ((nq1 > 0 && (!idx1 || !desc1 || !u1 || !v1 || !radius1 || !level1)) ||
      (nq2 > 0 && (!idx2 || !desc2 || !u2 || !v2 || !radius2 || !level2)))
    return VSG_ERR_INVALID;
  const int N1 = kf1->n, N2 = kf2->n;
  for (int i = 0; i < N1; i++) matches12[i] = -1;
  // both directions in one arena, two launches, one sync
  WindowCall a, b;
  int rc = a.begin(kf1->device, nq1, kWinBest, true, 0, 0);
  if (rc != VSG_OK) return rc;
  const size_t abytes = a.bytes();
  rc = b.begin(kf1->device, nq2, kWinBest, true, abytes, 0);
  if (rc != VSG_OK) return rc;
  a.c = b.c;  // b.begin may have grown (= re-allocated) the arena: a has not written anything yet
  auto fill = [](WindowCall &wc, int nq, const uint8_t *desc, const float *u, const float *v, const float *radius,
                 const int32_t *level) {
    WinQuery *Q = wc.queries();
    for (int i = 0; i < nq; i++) {
      // pKF->GetFeaturesInArea(u, v, radius) (:1531, :1609); kp.octave in [level - 1, level] (:1547-1548, :1625-1626)
      WinQuery w = {u[i], v[i], radius[i], -1, -1, level[i] - 1, level[i], 0.f, 0.f, level[i] < 0 ? 2 : 0, 0, 0};
      Q[i] = w;
    }
    if (nq) memcpy(wc.desc(), desc, (size_t)nq * 32);
  };
  fill(a, nq1, desc1, u1, v1, radius1, level1);  // KF1's points searched in KF2
  fill(b, nq2, desc2, u2, v2, radius2, level2);  // KF2's points searched in KF1
  rc = a.launch(kf2, kGateNone, 0x7FFFFFFF, nullptr, 0);
  if (rc == VSG_OK) rc = b.launch(kf1, kGateNone, 0x7FFFFFFF, nullptr, 0);
  if (rc != VSG_OK) return rc;
  rc = a.finish();  // one of the two directions may be empty: each waits for the stream it launched on
  if (rc == VSG_OK) rc = b.finish();
  if (rc != VSG_OK) return rc;
  std::vector<int> vnMatch1((size_t)N1, -1), vnMatch2((size_t)N2, -1);
  const int32_t *ba = a.best(), *bb = b.best();
  for (int k = 0; k < nq1; k++) {
    if (idx1[k] < 0 || idx1[k] >= N1) return VSG_ERR_INVALID;
    if (ba[2 * k] >= 0 && ba[2 * k + 1] <= walk::TH_HIGH) vnMatch1[idx1[k]] = ba[2 * k];  // :1562-1565
  }
  for (int k = 0; k < nq2; k++) {
    if (idx2[k] < 0 || idx2[k] >= N2) return VSG_ERR_INVALID;
    if (bb[2 * k] >= 0 && bb[2 * k + 1] <= walk::TH_HIGH) vnMatch2[idx2[k]] = bb[2 * k];  // :1640-1643
  }
  int nFound = 0;  // agreement (:1646-1662)
  for (int i1 = 0; i1 < N1; i1++) {
    const int i2 = vnMatch1[i1];
    if (i2 >= 0 && i2 < N2 && vnMatch2[i2] == i1) {
      matches12[i1] = i2;
      nFound++;
    }
  }
  return nFound;
}

static int fuse_search(vsg_frame *kf, int n_q, const uint8_t *mp_desc, const float *u, const float *v, const float *ur,
                       const float *radius, const int32_t *predicted_level, int right, int gate, int init,
                       const float *inv_level_sigma2, int nlevels, int32_t *best_idx, int32_t *best_dist) {
  if (frame_check(kf) != VSG_OK || n_q < 0 || !best_idx || !best_dist) return VSG_ERR_INVALID;
  if (n_q == 0) return 0;
  if (!mp_desc || !u || !v || !radius || !predicted_level) return VSG_ERR_INVALID;
  if (gate == kGateChi2 && (!ur || !inv_level_sigma2 || nlevels < 1 || nlevels > 16)) return VSG_ERR_INVALID;
  if (right && kf->nleft == -1) return VSG_ERR_INVALID;
  WindowCall wc;
  int rc = wc.begin(kf->device, n_q, kWinBest, true);
  if (rc != VSG_OK) return rc;
  WinQuery *Q = wc.queries();
  for (int i = 0; i < n_q; i++) {
    // pKF->GetFeaturesInArea(u, v, radius, bRight) (:1240 / :1394); kpLevel in [level - 1, level] (:1262-1265 / :1411-1414)
    WinQuery w = {u[i], v[i], radius[i], -1, -1, predicted_level[i] - 1, predicted_level[i], ur ? ur[i] : 0.f, 0.f,
                  (right ? 1 : 0) | (predicted_level[i] < 0 ? 2 : 0), 0, 0};
    Q[i] = w;
  }
  memcpy(wc.desc(), mp_desc, (size_t)n_q * 32);
  rc = wc.launch(kf, gate, init, inv_level_sigma2, nlevels);
  if (rc != VSG_OK) return rc;
  rc = wc.finish();
  if (rc != VSG_OK) return rc;
  const int32_t *b = wc.best();
  int nfused = 0;
  for (int k = 0; k < n_q; k++) {
    best_idx[k] = b[2 * k];
    best_dist[k] = b[2 * k] >= 0 ? b[2 * k + 1] : init;
    if (b[2 * k] >= 0 && b[2 * k + 1] <= walk::TH_LOW) nfused++;
  }
  return nfused;
}

int vsg_frame_fuse(vsg_frame *kf, int n_q, const uint8_t *mp_desc, const float *u, const float *v, const float *ur,
                   const float *radius, const int32_t *predicted_level, int right, const float *inv_level_sigma2,
                   int nlevels, int32_t *best_idx, int32_t *best_dist) {
  return fuse_search(kf, n_q, mp_desc, u, v, ur, radius, predicted_level, right, kGateChi2, 256, inv_level_sigma2,
                     nlevels, best_idx, best_dist);
}

int vsg_frame_fuse_sim3(vsg_frame *kf, int n_q, const uint8_t *mp_desc, const float *u, const float *v,
                        const float *radius, const int32_t *predicted_level, int32_t *best_idx, int32_t *best_dist) {
  return fuse_search(kf, n_q, mp_desc, u, v, nullptr, radius, predicted_level, 0, kGateNone, 0x7FFFFFFF, nullptr, 0,
                     best_idx, best_dist);
}

int vsg_fuse_decide(int n_q, const int32_t *query_mp, const int32_t *best_idx, const int32_t *best_dist, int sim3_form,
                    int32_t *slot_mp, int n_slots, int32_t *mp_obs, uint8_t *mp_bad, int n_mp, int32_t *action,
                    int32_t *other_mp) {
  if (n_q < 0 || !query_mp || !best_idx || !best_dist || !slot_mp || !mp_obs || !mp_bad || !action) return VSG_ERR_INVALID;
  int nFused = 0;
  for (int k = 0; k < n_q; k++) {
    action[k] = 0;
    if (other_mp) other_mp[k] = -1;
    const int idx = best_idx[k], pMP = query_mp[k];
    if (idx < 0 || best_dist[k] > walk::TH_LOW) continue;  // :1308 / :1429
    if (idx >= n_slots || pMP < 0 || pMP >= n_mp) return VSG_ERR_INVALID;
    const int pMPinKF = slot_mp[idx];  // pKF->GetMapPoint(bestIdx)
    if (pMPinKF >= 0) {
      if (pMPinKF >= n_mp) return VSG_ERR_INVALID;
      if (other_mp) other_mp[k] = pMPinKF;
      if (!mp_bad[pMPinKF]) {
        if (sim3_form) {
          action[k] = 5;  // vpReplacePoint[iMP] = pMPinKF (:1436)
        } else if (mp_obs[pMPinKF] > mp_obs[pMP]) {
          action[k] = 2;  // pMP->Replace(pMPinKF) (:1315): pMP turns bad, its observations move over
          mp_obs[pMPinKF] += mp_obs[pMP];
          mp_bad[pMP] = 1;
        } else {
          action[k] = 3;  // pMPinKF->Replace(pMP) (:1317): the slot now holds pMP
          mp_obs[pMP] += mp_obs[pMPinKF];
          mp_bad[pMPinKF] = 1;
          slot_mp[idx] = pMP;
        }
      } else {
        action[k] = 4;
      }
    } else {
      action[k] = 1;  // pMP->AddObservation(pKF, bestIdx); pKF->AddMapPoint(pMP, bestIdx) (:1321-1322 / :1440-1441)
      slot_mp[idx] = pMP;
      mp_obs[pMP] += 1;
    }
    nFused++;
  }
  return nFused;
}

int vsg_frame_search_for_initialization(vsg_frame *f1, vsg_frame *f2, const float *prev_x, const float *prev_y,
                                        int window_size, float nnratio, int check_orientation, int32_t *matches12) {
  if (frame_check(f1) != VSG_OK || frame_check(f2) != VSG_OK || f1->device != f2->device || !matches12 || !prev_x ||
      !prev_y)
    return VSG_ERR_INVALID;
  const int n1 = f1->n, n2 = f2->n;
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  if (n1 == 0 || n2 == 0) return 0;
  int rcx = VSG_OK;
  ThreadCtx *c = thread_ctx(f1->device, &rcx);
  if (!c) return rcx;
  return with_retry([&]() -> int {
    WindowCall wc;
    int rc = wc.begin(f1->device, n1, kWinList, false);
    if (rc != VSG_OK) return rc;
    WinQuery *Q = wc.queries();
    const vsg_keypoint *k1 = f1->h_kps.data(), *k2 = f2->h_kps.data();
    for (int i = 0; i < n1; i++) {
      // level1 > 0 -> continue (:659-661); F2.GetFeaturesInArea(vbPrevMatched[i1].x, .y, windowSize, level1, level1) (:663)
      const int level1 = k1[i].octave;
      WinQuery w = {prev_x[i], prev_y[i], (float)window_size, level1, level1, 0, -1, 0.f, 0.f, level1 > 0 ? 2 : 0, 0, 0};
      Q[i] = w;
    }
    // F1's descriptors are resident: the kernel reads the query descriptors where they are
    rc = wc.launch(f2, kGateNone, 256, nullptr, 0, f1->d_desc);
    if (rc != VSG_OK) return rc;
    rc = wc.finish();
    if (rc != VSG_OK) return rc;
    return walk::search_initialization(wc.lists(), n1, n2, nullptr, [&](int i) { return k1[i].angle; },
                                       [&](int i) { return k2[i].angle; }, nnratio, check_orientation != 0, matches12);
  });
}

}  // extern "C"
