// vsg_kernels.hip -- gfx950 kernels of the ORB extractor.  One launch per stage covers every pyramid
// level and every frame of the batch (grid.y / grid.z = frame), so a 64-frame batch fills 256 CUs.
//
//   k_resize        ORBextractor::ComputePyramid            ORBextractor.cc:1171-1195  ([OCV] resize INTER_LINEAR 8U)
//   k_fast_cells    per-cell cv::FAST(20) else cv::FAST(7)  ORBextractor.cc:787-876    ([OCV] fast.cpp / fast_score.cpp)
//   k_octree        DistributeOctTree                       ORBextractor.cc:562-785    (vsg_octree_core.h)
//   k_blur          GaussianBlur 7x7 sigma 2                ORBextractor.cc:1129-1130  ([OCV] 8.8 fixed point)
//   k_slots         output ordering / lapping area          ORBextractor.cc:1117-1168
//   k_orient_desc   IC_Angle + computeOrbDescriptor         ORBextractor.cc:73-149, 472-480, 1074-1081
//
// All image arithmetic is integer; the float pieces live in vsg_math.h with explicit non-contracting ops.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vsg_common.h"
#include "vsg_geometry.h"
#include "vsg_math.h"
#include "vsg_octree_core.h"
#include "vsg_kernels.h"

namespace vsg {

// ------------------------------------------------------------------------------------------------
// Level l of frame f: level 0 is read in place from the caller's buffer (Src0), levels >= 1 from the pyramid.
__device__ __forceinline__ const uint8_t *level_ptr(const FrameGeom *fg, const Src0 &s0, const uint8_t *pyr, int frame,
                                                    int level, int &pitch) {
  if (level == 0) {
    pitch = s0.pitch;
    return s0.base + (size_t)frame * s0.frame_stride;
  }
  pitch = fg->lv[level].pitch;
  return pyr + (size_t)frame * fg->pyr_frame_bytes + fg->lv[level].img_off;
}

// Pyramid: level `level` from level-1, chained like the reference (resize of the previous LEVEL).
// Thread = 4 horizontally adjacent destination pixels -> one aligned 32-bit store.
__global__ __launch_bounds__(256) void k_resize(uint8_t *__restrict__ pyr, const FrameGeom *__restrict__ fg,
                                                const Short4 *__restrict__ tab, Src0 s0, int level) {
  const LevelGeom &D = fg->lv[level];
  const int x4 = (blockIdx.x * 64 + threadIdx.x) * 4;
  const int y = blockIdx.y * 4 + threadIdx.y;
  if (x4 >= D.w || y >= D.h) return;
  uint8_t *frame = pyr + (size_t)blockIdx.z * fg->pyr_frame_bytes;
  int spitch;
  const uint8_t *S = level_ptr(fg, s0, pyr, blockIdx.z, level - 1, spitch);
  const Short4 ty = tab[D.tab_y_off + y];
  const uint8_t *S0 = S + (size_t)ty.a * spitch;
  const uint8_t *S1 = S + (size_t)ty.b * spitch;
  const int b0 = ty.c, b1 = ty.d;
  uint32_t out = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int x = x4 + k;
    if (x < D.w) {
      const Short4 tx = tab[D.tab_x_off + x];
      const int h0 = (int)S0[tx.a] * tx.b + (int)S0[tx.d] * tx.c;
      const int h1 = (int)S1[tx.a] * tx.b + (int)S1[tx.d] * tx.c;
      const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
      out |= (uint32_t)(v & 0xFF) << (8 * k);
    }
  }
  *(uint32_t *)(frame + D.img_off + (size_t)y * D.pitch + x4) = out;
}

// ------------------------------------------------------------------------------------------------
// FAST-9-16 ([OCV] fast.cpp FAST_t<16>, fast_score.cpp cornerScore<16>) on an LDS tile of row pitch kTileP.
//   score = max over the 16 contiguous 9-arcs of min(+-(v - ring)) - 1;  corner-at-t <=> score >= t, and the
//   score does not depend on t.  fast_quick() is a necessary condition on two opposing ring pairs (same idea as
//   the tab[] pre-test of FAST_t); fast_score() is exact and returns 0 below `floor_t`.
enum { kTileP = 84, kScoreP = 72 };

__device__ __forceinline__ bool fast_quick(const uint8_t *c, int floor_t) {
  const int P = kTileP;
  const int v = c[0];
  const int r0 = c[3 * P], r8 = c[-3 * P], r4 = c[3], r12 = c[-3];
  const int lo = v - floor_t, hi = v + floor_t;
  const bool dark = ((r0 < lo) | (r8 < lo)) & ((r4 < lo) | (r12 < lo));
  const bool bright = ((r0 > hi) | (r8 > hi)) & ((r4 > hi) | (r12 > hi));
  return dark | bright;
}

__device__ __forceinline__ int fast_score(const uint8_t *c, int floor_t) {
  const int P = kTileP;
  const int v = c[0];
  const int r0 = c[3 * P], r8 = c[-3 * P], r4 = c[3], r12 = c[-3];
  const int r2 = c[2 * P + 2], r10 = c[-2 * P - 2], r6 = c[-2 * P + 2], r14 = c[2 * P - 2];
  const int lo = v - floor_t, hi = v + floor_t;
  const bool dark = ((r0 < lo) | (r8 < lo)) & ((r4 < lo) | (r12 < lo)) & ((r2 < lo) | (r10 < lo)) & ((r6 < lo) | (r14 < lo));
  const bool bright = ((r0 > hi) | (r8 > hi)) & ((r4 > hi) | (r12 > hi)) & ((r2 > hi) | (r10 > hi)) & ((r6 > hi) | (r14 > hi));
  if (!(dark | bright)) return 0;
  int d[16];
  d[0] = v - r0;
  d[1] = v - c[3 * P + 1];
  d[2] = v - r2;
  d[3] = v - c[P + 3];
  d[4] = v - r4;
  d[5] = v - c[-P + 3];
  d[6] = v - r6;
  d[7] = v - c[-3 * P + 1];
  d[8] = v - r8;
  d[9] = v - c[-3 * P - 1];
  d[10] = v - r10;
  d[11] = v - c[-P - 3];
  d[12] = v - r12;
  d[13] = v - c[P - 3];
  d[14] = v - r14;
  d[15] = v - c[3 * P - 1];
  int mn2[16], mx2[16], mn4[16], mx4[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    mn2[k] = min(d[k], d[(k + 1) & 15]);
    mx2[k] = max(d[k], d[(k + 1) & 15]);
  }
#pragma unroll
  for (int k = 0; k < 16; k++) {
    mn4[k] = min(mn2[k], mn2[(k + 2) & 15]);
    mx4[k] = max(mx2[k], mx2[(k + 2) & 15]);
  }
  int A = -256, B = 256;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int mn9 = min(min(mn4[k], mn4[(k + 4) & 15]), d[(k + 8) & 15]);
    const int mx9 = max(max(mx4[k], mx4[(k + 4) & 15]), d[(k + 8) & 15]);
    A = max(A, mn9);
    B = min(B, mx9);
  }
  const int s = max(A, -B) - 1;
  return s >= floor_t ? s : 0;
}

// One workgroup per FAST cell.  The cell's valid region (3 px inside the reference's sub-image) is staged in LDS
// with aligned 32-bit loads; a cheap necessary test runs on every pixel and the few that pass are COMPACTED into an
// LDS queue, so the exact (expensive) score and the non-max suppression run on dense wavefronts.  NMS only looks at
// neighbours INSIDE the valid region (outside counts as 0, exactly like the zeroed row buffers of FAST_t); the cell
// emits survivors >= iniTh if any, else survivors >= minTh.  Order is irrelevant (the octree ranks candidates).
__global__ __launch_bounds__(256) void k_fast_cells(const uint8_t *__restrict__ pyr, const FrameGeom *__restrict__ fg,
                                                    const CellDesc *__restrict__ cells, Src0 s0,
                                                    uint32_t *__restrict__ cand, int *__restrict__ cand_count) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[(kCellMax + 6) * kTileP];
  __shared__ __attribute__((aligned(16))) uint8_t score[(kCellMax + 2) * kScoreP];
  __shared__ uint16_t queue[kCellMax * kCellMax];
  __shared__ int s_cnt[5];  // [0]=survivors>=iniTh [1]=survivors>=floor [2]=emit cursor [3]=global base [4]=queue length
  const CellDesc cell = cells[blockIdx.x];
  const int frame = blockIdx.y;
  const LevelGeom &L = fg->lv[cell.level];
  const int vw = cell.x1 - cell.x0, vh = cell.y1 - cell.y0;
  if (vw <= 0 || vh <= 0) return;
  const int iniTh = fg->iniTh, minTh = fg->minTh;
  const int floor_t = min(iniTh, minTh);
  const int tid = threadIdx.x, lane = tid & 63;
  int pitch;
  const uint8_t *img = level_ptr(fg, s0, pyr, frame, cell.level, pitch);
  // tile column 0 <-> image column ax (4-byte aligned); the valid region starts at tile column ox + 3
  const int ax = (cell.x0 - 3) & ~3, ox = (cell.x0 - 3) - ax;
  const int tdw = (ox + vw + 6 + 3) >> 2, th = vh + 6;  // dwords per tile row (<= 21)
  for (int i = tid; i < tdw * th; i += 256) {
    const int r = i / tdw, c = i - r * tdw;
    *(uint32_t *)&tile[r * kTileP + 4 * c] = *(const uint32_t *)(img + (size_t)(cell.y0 - 3 + r) * pitch + ax + 4 * c);
  }
  for (int i = tid; i < (vh + 2) * (kScoreP / 4); i += 256) ((uint32_t *)score)[i] = 0;
  if (tid < 5) s_cnt[tid] = 0;
  __syncthreads();
  // ---- phase 1: necessary test on every pixel, compaction of the passers
  const int npx = vw * vh;
  for (int i0 = 0; i0 < npx; i0 += 256) {
    const int i = i0 + tid;
    bool pass = false;
    if (i < npx) {
      const int r = i / vw, c = i - r * vw;
      pass = fast_quick(&tile[(r + 3) * kTileP + (c + 3 + ox)], floor_t);
    }
    const uint64_t m = __ballot(pass);
    if (m) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&s_cnt[4], __popcll(m));
      base = __shfl(base, 0);
      if (pass) queue[base + __popcll(m & ((1ull << lane) - 1))] = (uint16_t)i;
    }
  }
  __syncthreads();
  const int nq = s_cnt[4];
  // ---- phase 2: exact score of the queued pixels
  for (int q = tid; q < nq; q += 256) {
    const int i = queue[q];
    const int r = i / vw, c = i - r * vw;
    const int s = fast_score(&tile[(r + 3) * kTileP + (c + 3 + ox)], floor_t);
    if (s) score[(r + 1) * kScoreP + (c + 1)] = (uint8_t)s;
  }
  __syncthreads();
  // ---- phase 3: non-max suppression inside the cell
  uint32_t keep = 0;  // bit per loop iteration: queued pixel survives NMS
  int it = 0;
  for (int q = tid; q < nq; q += 256, it++) {
    const int i = queue[q];
    const int r = i / vw, c = i - r * vw;
    const uint8_t *sp = &score[(r + 1) * kScoreP + (c + 1)];
    const int s = sp[0];
    if (s == 0) continue;
    const bool is_max = s > sp[-1] && s > sp[1] && s > sp[-kScoreP - 1] && s > sp[-kScoreP] && s > sp[-kScoreP + 1] &&
                        s > sp[kScoreP - 1] && s > sp[kScoreP] && s > sp[kScoreP + 1];
    if (is_max) {
      keep |= 1u << it;
      atomicAdd(&s_cnt[1], 1);
      if (s >= iniTh) atomicAdd(&s_cnt[0], 1);
    }
  }
  __syncthreads();
  const int nHi = s_cnt[0], nLo = s_cnt[1];
  const int thr = nHi > 0 ? iniTh : minTh;  // vKeysCell.empty() -> retry with minThFAST (:848-851)
  const int nEmit = nHi > 0 ? nHi : nLo;
  if (nEmit == 0) return;
  if (tid == 0) s_cnt[3] = atomicAdd(&cand_count[frame * kMaxLevels + cell.level], nEmit);
  __syncthreads();
  const int base = s_cnt[3];
  uint32_t *out = cand + (size_t)frame * fg->cand_frame + L.cand_off;
  it = 0;
  for (int q = tid; q < nq; q += 256, it++) {
    if (!(keep & (1u << it))) continue;
    const int i = queue[q];
    const int r = i / vw, c = i - r * vw;
    const int s = score[(r + 1) * kScoreP + (c + 1)];
    if (s < thr) continue;
    const int slot = base + atomicAdd(&s_cnt[2], 1);
    if (slot < L.cand_cap) out[slot] = pack_cand(cell.x0 + c - kFastBorder, cell.y0 + r - kFastBorder, s);
  }
}

// ------------------------------------------------------------------------------------------------
// Workgroup implementation of the octree Group concept.
struct BlockGroup {
  int tid, nthreads;
  int *wtot;  // LDS, one int per wave
  __device__ void sync() { __syncthreads(); }
  __device__ int atomic_add(int *p, int v) { return atomicAdd(p, v); }
  __device__ void atomic_max(uint32_t *p, uint32_t v) { atomicMax(p, v); }
  __device__ void atomic_min(int *p, int v) { atomicMin(p, v); }
  __device__ int exclusive_scan(int *a, int n) {
    const int per = (n + nthreads - 1) / nthreads;
    const int lo = min(tid * per, n), hi = min(lo + per, n);
    int s = 0;
    for (int i = lo; i < hi; i++) s += a[i];
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
    int incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < nwaves; w++) {
      const int t = wtot[w];
      if (w < wave) base += t;
      total += t;
    }
    int run = base + incl - s;
    for (int i = lo; i < hi; i++) {
      const int v = a[i];
      a[i] = run;
      run += v;
    }
    __syncthreads();
    return total;
  }
};

__global__ __launch_bounds__(256) void k_octree(const FrameGeom *__restrict__ fg, const uint32_t *__restrict__ cand,
                                                const int *__restrict__ cand_count, uint16_t *__restrict__ node_of,
                                                uint32_t *__restrict__ sel, int *__restrict__ sel_count, int cap) {
  extern __shared__ __attribute__((aligned(16))) uint8_t oct_lds[];
  __shared__ int wtot[8];
  const int level = blockIdx.x, frame = blockIdx.y;
  const LevelGeom &L = fg->lv[level];
  octree::Params P;
  P.N = L.quota;
  P.height = L.oct_height;
  P.nIni = L.nIni;
  P.iniUL = L.iniUL;
  P.iniThresh = L.iniThresh;
  P.nCols = L.nCols;
  P.wCell = L.wCell;
  P.hCell = L.hCell;
  octree::Work W;
  octree::carve(W, oct_lds, cap);
  BlockGroup g;
  g.tid = threadIdx.x;
  g.nthreads = blockDim.x;
  g.wtot = wtot;
  int npts = cand_count[frame * kMaxLevels + level];
  if (npts > L.cand_cap) npts = L.cand_cap;
  const size_t coff = (size_t)frame * fg->cand_frame + L.cand_off;
  const int n = octree::distribute(g, P, cand + coff, npts, node_of + coff, W,
                                   sel + (size_t)frame * fg->sel_frame + L.sel_off);
  if (threadIdx.x == 0) sel_count[frame * kMaxLevels + level] = n;
}

// ------------------------------------------------------------------------------------------------
// 7x7 Gaussian, sigma 2, 8.8 fixed point, BORDER_REFLECT_101 at the level's own edges (the reference blurs a
// border-less clone, ORBextractor.cc:1129-1130).  No LDS: a thread owns 4 adjacent columns and walks down
// kBlurStrip + 6 rows with a 7-row register window.  Per row it loads 3 aligned dwords (12 px), forms the 4
// horizontal 8.8 sums with v_alignbyte + v_dot4_u32_u8, and once the window is full emits 4 output bytes
// (one 32-bit store) from 7 v_mad_u32_u24 per pixel.  Adjacent lanes own adjacent column groups, so every
// wave-level load/store is one contiguous 256-byte row segment.
__device__ __forceinline__ int reflect101(int p, int len) {
  while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * len - 2 - p;
  return p;
}

__global__ __launch_bounds__(256) void k_blur(const uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur,
                                              const FrameGeom *__restrict__ fg, Src0 s0) {
  int level = 0;
  while (level + 1 < fg->nlevels && (int)blockIdx.x >= fg->lv[level + 1].blur_block_base) level++;
  const LevelGeom &L = fg->lv[level];
  const int t = ((int)blockIdx.x - L.blur_block_base) * 256 + (int)threadIdx.x;
  if (t >= L.blur_nxg * L.blur_nys) return;
  const int gx = t % L.blur_nxg, sy = t / L.blur_nxg;
  const int x0 = gx * 4, y0 = sy * kBlurStrip;
  const int w = L.w, h = L.h;
  int spitch;
  const uint8_t *img = level_ptr(fg, s0, pyr, blockIdx.y, level, spitch);
  uint8_t *dst = blur + (size_t)blockIdx.y * fg->pyr_frame_bytes + L.img_off;
  const uint32_t T0 = fg->taps[0] | (fg->taps[1] << 8) | (fg->taps[2] << 16) | ((uint32_t)fg->taps[3] << 24);
  const uint32_t T1 = fg->taps[4] | (fg->taps[5] << 8) | (fg->taps[6] << 16);
  uint32_t k[7];
#pragma unroll
  for (int j = 0; j < 7; j++) k[j] = fg->taps[j];
  const bool interior = x0 >= 4 && x0 + 8 <= w;  // all 12 source bytes exist: aligned dword loads
  uint32_t win[7][4];
#pragma unroll
  for (int turn = 0; turn < (kBlurStrip + 6) / 7; turn++) {
#pragma unroll
    for (int s = 0; s < 7; s++) {
      const int rr = turn * 7 + s;           // 0 .. kBlurStrip+5 : source row y0 - 3 + rr
      const int ysrc = reflect101(y0 - 3 + rr, h);
      const uint8_t *row = img + (size_t)ysrc * spitch;
      uint32_t d0, d1, d2;
      if (interior) {
        d0 = *(const uint32_t *)(row + x0 - 4);
        d1 = *(const uint32_t *)(row + x0);
        d2 = *(const uint32_t *)(row + x0 + 4);
      } else {
        uint32_t b[12];
#pragma unroll
        for (int i = 0; i < 12; i++) b[i] = row[reflect101(x0 - 4 + i, w)];
        d0 = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
        d1 = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
        d2 = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
      }
      // pixel j: taps over bytes j+1 .. j+7 of {d0,d1,d2}
      win[s][0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 1), T0,
                                         __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 1), T1, 0u, false), false);
      win[s][1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 2), T0,
                                         __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 2), T1, 0u, false), false);
      win[s][2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 3), T0,
                                         __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 3), T1, 0u, false), false);
      win[s][3] = __builtin_amdgcn_udot4(d1, T0, __builtin_amdgcn_udot4(d2, T1, 0u, false), false);
      if (rr >= 6) {
        const int yo = y0 + rr - 6;
        if (yo < h) {
          uint32_t out = 0;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            uint32_t acc = 32768u;
#pragma unroll
            for (int q = 0; q < 7; q++) acc += __umul24(k[q], win[(s + 1 + q) % 7][j]);  // oldest row first
            out |= min(acc >> 16, 255u) << (8 * j);
          }
          *(uint32_t *)(dst + (size_t)yo * L.pitch + x0) = out;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Output slots: keypoints are visited level by level in octree order; those inside the lapping area
// fill the arrays from the back, the others from the front (ORBextractor.cc:1119,1152-1163).
__global__ __launch_bounds__(256) void k_slots(const FrameGeom *__restrict__ fg, const uint32_t *__restrict__ sel,
                                               const int *__restrict__ sel_count, int *__restrict__ flags,
                                               int *__restrict__ slots, FrameHeader *__restrict__ hdr, int lap0,
                                               int lap1) {
  __shared__ int wtot[8];
  __shared__ int lstart[kMaxLevels + 1];
  const int frame = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) {
    int s = 0;
    for (int l = 0; l < fg->nlevels; l++) {
      lstart[l] = s;
      s += min(sel_count[frame * kMaxLevels + l], fg->lv[l].sel_cap);
    }
    for (int l = fg->nlevels; l <= kMaxLevels; l++) lstart[l] = s;
  }
  __syncthreads();
  const int n = lstart[fg->nlevels];
  int *fl = flags + (size_t)frame * fg->out_cap;
  int *sl = slots + (size_t)frame * fg->out_cap;
  const float flap0 = (float)lap0, flap1 = (float)lap1;
  for (int i = tid; i < n; i += 256) {
    int l = 0;
    while (i >= lstart[l + 1]) l++;
    const uint32_t c = sel[(size_t)frame * fg->sel_frame + fg->lv[l].sel_off + (i - lstart[l])];
    float x = (float)(VSG_CAND_X(c) + kFastBorder);
    if (l != 0) x = fmul(x, fg->lv[l].scale);  // keypoint->pt *= scale (:1147-1150)
    fl[i] = (x >= flap0 && x <= flap1) ? 1 : 0;
  }
  __syncthreads();
  BlockGroup g;
  g.tid = tid;
  g.nthreads = 256;
  g.wtot = wtot;
  const int T = g.exclusive_scan(fl, n);  // fl[i] = lapping keypoints before i
  for (int i = tid; i < n; i += 256) {
    const int before = fl[i];
    const int after = (i + 1 < n) ? fl[i + 1] : T;
    sl[i] = (after - before) ? (n - 1 - before) : (i - before);  // stereoIndex-- / monoIndex++
  }
  if (tid == 0) {
    hdr[frame].n = n;
    hdr[frame].mono = n - T;
    for (int l = 0; l <= kMaxLevels; l++) hdr[frame].level_start[l] = lstart[l];
  }
}

// ------------------------------------------------------------------------------------------------
// One wavefront per keypoint: intensity-centroid angle on the un-blurred level, steered rBRIEF-256 on
// the blurred level, keypoint record + 32 descriptor bytes written to the keypoint's output slot.
__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

__global__ __launch_bounds__(256) void k_orient_desc(const uint8_t *__restrict__ pyr, const uint8_t *__restrict__ blur,
                                                     const FrameGeom *__restrict__ fg, Src0 s0,
                                                     const uint32_t *__restrict__ sel,
                                                     const int *__restrict__ slots, const FrameHeader *__restrict__ hdr,
                                                     const int8_t *__restrict__ pattern, KeyPointPOD *__restrict__ kps,
                                                     uint8_t *__restrict__ desc, int *__restrict__ counts, int capacity) {
  __shared__ int8_t pat[1024];
  const int frame = blockIdx.y, tid = threadIdx.x;
  ((uint32_t *)pat)[tid] = ((const uint32_t *)pattern)[tid];
  const FrameHeader &H = hdr[frame];
  const int n = H.n;
  if (blockIdx.x == 0 && tid == 0) {
    counts[frame * 2 + 0] = n;
    counts[frame * 2 + 1] = H.mono;
  }
  __syncthreads();
  const int lane = tid & 63;
  const int g = blockIdx.x * 4 + (tid >> 6);
  if (g >= n) return;
  int l = 0;
  while (g >= H.level_start[l + 1]) l++;
  const LevelGeom &L = fg->lv[l];
  const uint32_t c = sel[(size_t)frame * fg->sel_frame + L.sel_off + (g - H.level_start[l])];
  const int cx = VSG_CAND_X(c) + kFastBorder, cy = VSG_CAND_Y(c) + kFastBorder;
  const size_t foff = (size_t)frame * fg->pyr_frame_bytes + L.img_off;
  int upitch;
  const uint8_t *unblurred = level_ptr(fg, s0, pyr, frame, l, upitch);
  // ---- IC_Angle: lanes 0..61 = 31 rows x {left half, right half}
  int m10 = 0, m01 = 0;
  if (lane < 62) {
    const int v = (lane >> 1) - kHalfPatch;
    const int dmax = c_umax[v < 0 ? -v : v];
    const uint8_t *row = unblurred + (size_t)(cy + v) * upitch + cx;
    int u0, u1;
    if (lane & 1) {
      u0 = 0;
      u1 = dmax;
    } else {
      u0 = -dmax;
      u1 = -1;
    }
    int sum = 0;
    for (int u = u0; u <= u1; u++) {
      const int p = row[u];
      m10 += u * p;
      sum += p;
    }
    m01 = v * sum;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    m10 += __shfl_xor(m10, d);
    m01 += __shfl_xor(m01, d);
  }
  const float angle = fast_atan2_deg((float)m01, (float)m10);
  float a, b;
  brief_rotation(angle, &a, &b);
  // ---- descriptor: lane handles tests lane, lane+64, lane+128, lane+192
  const uint8_t *center = blur + foff + (size_t)cy * L.pitch + cx;
  const int pitch = L.pitch;
  const int slot = slots[(size_t)frame * fg->out_cap + g];
  uint64_t word = 0;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int k = r * 64 + lane;
    int dx0, dy0, dx1, dy1;
    brief_offset(pat[4 * k + 0], pat[4 * k + 1], a, b, &dx0, &dy0);
    brief_offset(pat[4 * k + 2], pat[4 * k + 3], a, b, &dx1, &dy1);
    const int t0 = center[dy0 * pitch + dx0], t1 = center[dy1 * pitch + dx1];
    const uint64_t m = __ballot(t0 < t1);
    if (lane == r) word = m;
  }
  if (slot < capacity) {
    if (lane < 4) *(uint64_t *)(desc + ((size_t)frame * capacity + slot) * 32 + lane * 8) = word;
    if (lane == 0) {
      KeyPointPOD kp;
      kp.x = (float)cx;
      kp.y = (float)cy;
      if (l != 0) {
        kp.x = fmul(kp.x, L.scale);
        kp.y = fmul(kp.y, L.scale);
      }
      kp.size = L.kp_size;
      kp.angle = angle;
      kp.response = (float)VSG_CAND_R(c);
      kp.octave = l;
      kp.class_id = -1;
      kps[(size_t)frame * capacity + slot] = kp;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// mvImagePyramid[level] with its 19 px BORDER_REFLECT_101 frame (copyMakeBorder, :1186-1192), on demand.
__global__ void k_border_copy(const uint8_t *__restrict__ img, int w, int h, int pitch, uint8_t *__restrict__ dst,
                              int dpitch, int b) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w + 2 * b) return;
  dst[(size_t)y * dpitch + x] = img[(size_t)reflect101(y - b, h) * pitch + reflect101(x - b, w)];
}

// ------------------------------------------------------------------------------------------------
// launchers (host)
void launch_resize(hipStream_t s, uint8_t *pyr, const FrameGeom *d_fg, const Short4 *d_tab, const Src0 &s0,
                   const FrameGeom &fg, int level, int nframes) {
  const LevelGeom &D = fg.lv[level];
  dim3 grid((D.w + 255) / 256, (D.h + 3) / 4, nframes), block(64, 4);
  hipLaunchKernelGGL(k_resize, grid, block, 0, s, pyr, d_fg, d_tab, s0, level);
}
void launch_fast(hipStream_t s, const uint8_t *pyr, const FrameGeom *d_fg, const CellDesc *d_cells, const Src0 &s0,
                 uint32_t *cand, int *cand_count, const FrameGeom &fg, int nframes) {
  dim3 grid(fg.total_cells, nframes), block(256);
  hipLaunchKernelGGL(k_fast_cells, grid, block, 0, s, pyr, d_fg, d_cells, s0, cand, cand_count);
}
void launch_octree(hipStream_t s, const FrameGeom *d_fg, const uint32_t *cand, const int *cand_count,
                   uint16_t *node_of, uint32_t *sel, int *sel_count, const FrameGeom &fg, int maxQuota, int nframes) {
  const int cap = octree::node_capacity(maxQuota);
  const size_t lds = octree::work_bytes(cap);
  dim3 grid(fg.nlevels, nframes), block(256);
  hipLaunchKernelGGL(k_octree, grid, block, lds, s, d_fg, cand, cand_count, node_of, sel, sel_count, cap);
}
void launch_blur(hipStream_t s, const uint8_t *pyr, uint8_t *blur, const FrameGeom *d_fg, const Src0 &s0,
                 const FrameGeom &fg, int nframes) {
  dim3 grid(fg.total_blur_blocks, nframes), block(256);
  hipLaunchKernelGGL(k_blur, grid, block, 0, s, pyr, blur, d_fg, s0);
}
void launch_slots(hipStream_t s, const FrameGeom *d_fg, const uint32_t *sel, const int *sel_count, int *flags,
                  int *slots, FrameHeader *hdr, int lap0, int lap1, int nframes) {
  hipLaunchKernelGGL(k_slots, dim3(nframes), dim3(256), 0, s, d_fg, sel, sel_count, flags, slots, hdr, lap0, lap1);
}
void launch_orient_desc(hipStream_t s, const uint8_t *pyr, const uint8_t *blur, const FrameGeom *d_fg, const Src0 &s0,
                        const uint32_t *sel, const int *slots, const FrameHeader *hdr, const int8_t *pattern,
                        KeyPointPOD *kps, uint8_t *desc, int *counts, int capacity, const FrameGeom &fg, int nframes) {
  dim3 grid((fg.out_cap + 3) / 4, nframes), block(256);
  hipLaunchKernelGGL(k_orient_desc, grid, block, 0, s, pyr, blur, d_fg, s0, sel, slots, hdr, pattern, kps, desc,
                     counts, capacity);
}
void launch_border_copy(hipStream_t s, const uint8_t *img, int w, int h, int pitch, uint8_t *dst, int dpitch, int b) {
  dim3 grid((w + 2 * b + 255) / 256, h + 2 * b), block(256);
  hipLaunchKernelGGL(k_border_copy, grid, block, 0, s, img, w, h, pitch, dst, dpitch, b);
}

}  // namespace vsg
