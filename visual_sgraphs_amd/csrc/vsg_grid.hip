// vsg_grid.hip -- the Frame feature grid on the device.
//   Frame::AssignFeaturesToGrid + PosInGrid   orb_slam3/src/Frame.cc:521-553, 870-880
//   Frame::GetFeaturesInArea                  orb_slam3/src/Frame.cc:802-868   (KeyFrame.cc:834-874 = no level filter)
// The candidate ORDER is part of the contract (ties in the Hamming argmin resolve to the earliest candidate):
// cells ix outer / iy inner, and inside a cell the insertion order = ascending keypoint index.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string.h>

#include <vector>

#include "../../include/vsg_orb.h"
#include "vsg_ctx.h"
#include "vsg_math.h"

namespace {
enum { kCols = 64, kRows = 48, kCells = kCols * kRows };  // FRAME_GRID_COLS / ROWS (Frame.h:49-50)

struct GridParams {
  float minX, minY, invW, invH;
};

// exclusive prefix sum of the per-query candidate counts (one workgroup; nq is a few thousand at most)
__global__ __launch_bounds__(256) void k_grid_offsets(const int *counts, int nq, int *cand_off) {
  __shared__ int part[256];
  const int tid = threadIdx.x, per = (nq + 255) / 256, lo = min(tid * per, nq), hi = min(lo + per, nq);
  int s = 0;
  for (int i = lo; i < hi; i++) s += counts[i];
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int t = 0; t < 256; t++) {
      const int v = part[t];
      part[t] = run;
      run += v;
    }
    cand_off[nq] = run;
  }
  __syncthreads();
  int run = part[tid];
  for (int i = lo; i < hi; i++) {
    cand_off[i] = run;
    run += counts[i];
  }
}

// GetFeaturesInArea for one query per thread; pass 0 counts, pass 1 writes at cand_off[q]
__global__ void k_grid_query(const vsg_keypoint *kps, const int *cell_start, const int *entries, GridParams P,
                             const float *qx, const float *qy, const float *qr, const int *minLevel,
                             const int *maxLevel, int nq, int *counts, const int *cand_off, int *cand_idx, int cap) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const float x = qx[q], y = qy[q], r = qr[q];
  const int lo = minLevel ? minLevel[q] : -1, hi = maxLevel ? maxLevel[q] : -1;
  int n = 0;
  const int out0 = cand_off ? cand_off[q] : 0;
  // (int)floor((x - mnMinX - factorX) * mfGridElementWidthInv) etc. (Frame.cc:810-832)
  const int nMinCellX = max(0, vsg::cvt_int_x86(floorf(vsg::fmul(vsg::fsub(vsg::fsub(x, P.minX), r), P.invW))));
  const int nMaxCellX = min(kCols - 1, vsg::cvt_int_x86(ceilf(vsg::fmul(vsg::fadd(vsg::fsub(x, P.minX), r), P.invW))));
  const int nMinCellY = max(0, vsg::cvt_int_x86(floorf(vsg::fmul(vsg::fsub(vsg::fsub(y, P.minY), r), P.invH))));
  const int nMaxCellY = min(kRows - 1, vsg::cvt_int_x86(ceilf(vsg::fmul(vsg::fadd(vsg::fsub(y, P.minY), r), P.invH))));
  if (nMinCellX < kCols && nMaxCellX >= 0 && nMinCellY < kRows && nMaxCellY >= 0) {
    const bool bCheckLevels = (lo > 0) || (hi >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
      for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
        const int c = ix * kRows + iy;
        for (int e = cell_start[c]; e < cell_start[c + 1]; e++) {
          const int i = entries[e];
          const vsg_keypoint kp = kps[i];
          if (bCheckLevels) {
            if (kp.octave < lo) continue;
            if (hi >= 0 && kp.octave > hi) continue;
          }
          const float distx = vsg::fsub(kp.x, x), disty = vsg::fsub(kp.y, y);
          if (fabsf(distx) < r && fabsf(disty) < r) {
            if (cand_idx && out0 + n < cap) cand_idx[out0 + n] = i;
            n++;
          }
        }
      }
    }
  }
  if (counts) counts[q] = n;
}
}  // namespace

struct vsg_grid {
  int device = 0, n = 0;
  GridParams P{};
  void *d_block = nullptr;  // one allocation: keypoints | cell_start | entries
  vsg_keypoint *d_kps = nullptr;
  int *d_cell_start = nullptr, *d_entries = nullptr;
};

#define G_TRY(expr)                         \
  do {                                      \
    if ((expr) != hipSuccess) return VSG_ERR_HIP; \
  } while (0)

extern "C" {

void vsg_grid_destroy(vsg_grid *g) {
  if (!g) return;
  hipSetDevice(g->device);
  hipFree(g->d_block);
  delete g;
}

int vsg_grid_build(int device, const vsg_keypoint *kps, int n, float min_x, float min_y, float max_x, float max_y,
                   vsg_grid **out) {
  if (!out || n < 0 || (n > 0 && !kps) || n > 32767) return VSG_ERR_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return VSG_ERR_NO_DEVICE;
  G_TRY(hipSetDevice(device));
  vsg_grid *g = new vsg_grid();
  g->device = device;
  g->n = n;
  // mfGridElementWidthInv = FRAME_GRID_COLS / (mnMaxX - mnMinX)   (Frame.cc:378-379)
  g->P = {min_x, min_y, (float)kCols / (max_x - min_x), (float)kRows / (max_y - min_y)};
  // AssignFeaturesToGrid (Frame.cc:521-553) is a stable bucket fill of a thousand keypoints: done here, on the
  // host that hands the keypoints over, with the reference's float operations (PosInGrid, Frame.cc:870-880; libm
  // round = half away from zero).  The device gets the finished CSR in the same copy as the keypoints; the queries
  // (the part that scales with map points x cells) run there.
  const size_t bK = sizeof(vsg_keypoint) * (size_t)(n + 1), bS = sizeof(int) * (kCells + 1), bE = sizeof(int) * (size_t)(n + 1);
  std::vector<uint8_t> stage(bK + bS + bE, 0);
  if (n) memcpy(stage.data(), kps, sizeof(vsg_keypoint) * (size_t)n);
  int *cell_start = (int *)(stage.data() + bK), *entries = (int *)(stage.data() + bK + bS);
  std::vector<int16_t> cell_of((size_t)n + 1);
  std::vector<int> cnt(kCells, 0);
  for (int i = 0; i < n; i++) {
    const int px = vsg::cvt_int_x86(roundf(vsg::fmul(vsg::fsub(kps[i].x, g->P.minX), g->P.invW)));
    const int py = vsg::cvt_int_x86(roundf(vsg::fmul(vsg::fsub(kps[i].y, g->P.minY), g->P.invH)));
    const bool in = !(px < 0 || px >= kCols || py < 0 || py >= kRows);
    cell_of[i] = in ? (int16_t)(px * kRows + py) : (int16_t)-1;
    if (in) cnt[px * kRows + py]++;
  }
  int run = 0;
  for (int c = 0; c < kCells; c++) {
    cell_start[c] = run;
    run += cnt[c];
    cnt[c] = cell_start[c];
  }
  cell_start[kCells] = run;
  for (int i = 0; i < n; i++)
    if (cell_of[i] >= 0) entries[cnt[cell_of[i]]++] = i;  // insertion order == ascending keypoint index
  // a grid is an object with a lifetime (like the Frame it indexes): one allocation at build, none per query
  int crc = VSG_OK;
  vsg::ThreadCtx *c = vsg::thread_ctx(device, &crc);
  hipError_t e = c ? hipMalloc(&g->d_block, stage.size()) : hipErrorInvalidDevice;
  if (e == hipSuccess) e = hipMemcpyAsync(g->d_block, stage.data(), stage.size(), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) {
    vsg_grid_destroy(g);
    return VSG_ERR_HIP;
  }
  g->d_kps = (vsg_keypoint *)g->d_block;
  g->d_cell_start = (int *)((uint8_t *)g->d_block + bK);
  g->d_entries = (int *)((uint8_t *)g->d_block + bK + bS);
  *out = g;
  return VSG_OK;
}

int vsg_grid_query(vsg_grid *g, const float *x, const float *y, const float *r, const int32_t *min_level,
                   const int32_t *max_level, int nq, int32_t *cand_off, int32_t *cand_idx, int cap) {
  if (!g || !x || !y || !r || !cand_off || nq < 0 || cap < 0) return VSG_ERR_INVALID;
  int rc = VSG_OK;
  vsg::ThreadCtx *c = vsg::thread_ctx(g->device, &rc);
  if (!c) return rc;
  cand_off[0] = 0;
  if (nq == 0) return 0;
  // queries up through the calling thread's pinned arena (read once by the kernels, straight over PCIe); count ->
  // offsets -> fill on the device without a host round trip in between; offsets and indices come back the same way
  const size_t Q = (size_t)nq;
  vsg::Stage st;
  const size_t oX = st.add(Q * 4), oY = st.add(Q * 4), oR = st.add(Q * 4), oLo = st.add(Q * 4), oHi = st.add(Q * 4),
               oOff = st.add((Q + 1) * 4), oIdx = st.add((size_t)cap * 4 + 4);
  rc = vsg::ctx_reserve(c, st.total, Q * 4 + 64);
  if (rc != VSG_OK) return rc;
  uint8_t *h = c->h_pin, *d = c->d_pin;
  memcpy(h + oX, x, Q * 4);
  memcpy(h + oY, y, Q * 4);
  memcpy(h + oR, r, Q * 4);
  if (min_level) memcpy(h + oLo, min_level, Q * 4);
  if (max_level) memcpy(h + oHi, max_level, Q * 4);
  const float *dx = (const float *)(d + oX), *dy = (const float *)(d + oY), *dr = (const float *)(d + oR);
  const int *dlo = min_level ? (const int *)(d + oLo) : nullptr, *dhi = max_level ? (const int *)(d + oHi) : nullptr;
  int *dcnt = (int *)c->d_buf, *doff = (int *)(d + oOff), *didx = (int *)(d + oIdx);
  hipLaunchKernelGGL(k_grid_query, dim3((nq + 63) / 64), dim3(64), 0, c->stream, g->d_kps, g->d_cell_start, g->d_entries,
                     g->P, dx, dy, dr, dlo, dhi, nq, dcnt, (const int *)nullptr, (int *)nullptr, 0);
  hipLaunchKernelGGL(k_grid_offsets, dim3(1), dim3(256), 0, c->stream, dcnt, nq, doff);
  if (cand_idx && cap > 0)
    hipLaunchKernelGGL(k_grid_query, dim3((nq + 63) / 64), dim3(64), 0, c->stream, g->d_kps, g->d_cell_start,
                       g->d_entries, g->P, dx, dy, dr, dlo, dhi, nq, (int *)nullptr, doff, didx, cap);
  G_TRY(hipGetLastError());
  G_TRY(hipStreamSynchronize(c->stream));
  memcpy(cand_off, h + oOff, (Q + 1) * 4);
  const int total = cand_off[nq];
  const int nw = total < cap ? total : cap;
  if (cand_idx && nw > 0) memcpy(cand_idx, h + oIdx, (size_t)nw * 4);
  return total;
}

}  // extern "C"
