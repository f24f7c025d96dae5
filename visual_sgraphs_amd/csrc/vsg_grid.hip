// vsg_grid.hip -- the Frame feature grid on the device.
//   Frame::AssignFeaturesToGrid + PosInGrid   orb_slam3/src/Frame.cc:521-553, 870-880
//   Frame::GetFeaturesInArea                  orb_slam3/src/Frame.cc:802-868   (KeyFrame.cc:834-874 = no level filter)
// The candidate ORDER is part of the contract (ties in the Hamming argmin resolve to the earliest candidate):
// cells ix outer / iy inner, and inside a cell the insertion order = ascending keypoint index.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/vsg_orb.h"
#include "vsg_math.h"

namespace {
enum { kCols = 64, kRows = 48, kCells = kCols * kRows };  // FRAME_GRID_COLS / ROWS (Frame.h:49-50)

struct GridParams {
  float minX, minY, invW, invH;
};

// one thread per cell walks the keypoints in index order: a stable bucket fill without any sort
__global__ void k_grid_cell_ids(const vsg_keypoint *kps, int n, GridParams P, int16_t *cell_of) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // posX = round((kp.pt.x - mnMinX) * mfGridElementWidthInv)   (Frame.cc:872-873; libm round = half away from zero)
  const int px = (int)roundf(vsg::fmul(vsg::fsub(kps[i].x, P.minX), P.invW));
  const int py = (int)roundf(vsg::fmul(vsg::fsub(kps[i].y, P.minY), P.invH));
  cell_of[i] = (px < 0 || px >= kCols || py < 0 || py >= kRows) ? (int16_t)-1 : (int16_t)(px * kRows + py);
}
__global__ void k_grid_count(const int16_t *cell_of, int n, int *cell_cnt) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= kCells) return;
  int k = 0;
  for (int i = 0; i < n; i++) k += cell_of[i] == c;
  cell_cnt[c] = k;
}
__global__ void k_grid_scan(const int *cell_cnt, int *cell_start) {  // 3072 cells: one thread is plenty
  int s = 0;
  for (int c = 0; c < kCells; c++) {
    cell_start[c] = s;
    s += cell_cnt[c];
  }
  cell_start[kCells] = s;
}
__global__ void k_grid_fill(const int16_t *cell_of, int n, const int *cell_start, int *entries) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= kCells) return;
  int o = cell_start[c];
  for (int i = 0; i < n; i++)
    if (cell_of[i] == c) entries[o++] = i;  // insertion order == ascending keypoint index
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
  const int nMinCellX = max(0, (int)floorf(vsg::fmul(vsg::fsub(vsg::fsub(x, P.minX), r), P.invW)));
  const int nMaxCellX = min(kCols - 1, (int)ceilf(vsg::fmul(vsg::fadd(vsg::fsub(x, P.minX), r), P.invW)));
  const int nMinCellY = max(0, (int)floorf(vsg::fmul(vsg::fsub(vsg::fsub(y, P.minY), r), P.invH)));
  const int nMaxCellY = min(kRows - 1, (int)ceilf(vsg::fmul(vsg::fadd(vsg::fsub(y, P.minY), r), P.invH)));
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
  hipFree(g->d_kps), hipFree(g->d_cell_start), hipFree(g->d_entries);
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
  int16_t *d_cell_of = nullptr;
  int *d_cnt = nullptr;
  hipError_t e = hipMalloc(&g->d_kps, sizeof(vsg_keypoint) * (n + 1));
  if (e == hipSuccess) e = hipMalloc(&g->d_cell_start, sizeof(int) * (kCells + 1));
  if (e == hipSuccess) e = hipMalloc(&g->d_entries, sizeof(int) * (n + 1));
  if (e == hipSuccess) e = hipMalloc(&d_cell_of, sizeof(int16_t) * (n + 1));
  if (e == hipSuccess) e = hipMalloc(&d_cnt, sizeof(int) * kCells);
  if (e == hipSuccess && n) e = hipMemcpy(g->d_kps, kps, sizeof(vsg_keypoint) * n, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    if (n) hipLaunchKernelGGL(k_grid_cell_ids, dim3((n + 255) / 256), dim3(256), 0, 0, g->d_kps, n, g->P, d_cell_of);
    hipLaunchKernelGGL(k_grid_count, dim3(kCells / 256), dim3(256), 0, 0, d_cell_of, n, d_cnt);
    hipLaunchKernelGGL(k_grid_scan, dim3(1), dim3(1), 0, 0, d_cnt, g->d_cell_start);
    hipLaunchKernelGGL(k_grid_fill, dim3(kCells / 256), dim3(256), 0, 0, d_cell_of, n, g->d_cell_start, g->d_entries);
    e = hipDeviceSynchronize();
  }
  hipFree(d_cell_of), hipFree(d_cnt);
  if (e != hipSuccess) {
    vsg_grid_destroy(g);
    return VSG_ERR_HIP;
  }
  *out = g;
  return VSG_OK;
}

int vsg_grid_query(vsg_grid *g, const float *x, const float *y, const float *r, const int32_t *min_level,
                   const int32_t *max_level, int nq, int32_t *cand_off, int32_t *cand_idx, int cap) {
  if (!g || !x || !y || !r || !cand_off || nq < 0 || cap < 0) return VSG_ERR_INVALID;
  G_TRY(hipSetDevice(g->device));
  cand_off[0] = 0;
  if (nq == 0) return 0;
  float *dx = nullptr, *dy = nullptr, *dr = nullptr;
  int *dlo = nullptr, *dhi = nullptr, *dcnt = nullptr, *doff = nullptr, *didx = nullptr;
  int total = VSG_ERR_HIP;
  std::vector<int> cnt((size_t)nq);
  hipError_t e = hipMalloc(&dx, 4 * nq);
  if (e == hipSuccess) e = hipMalloc(&dy, 4 * nq);
  if (e == hipSuccess) e = hipMalloc(&dr, 4 * nq);
  if (e == hipSuccess) e = hipMalloc(&dcnt, 4 * nq);
  if (e == hipSuccess) e = hipMalloc(&doff, 4 * (nq + 1));
  if (e == hipSuccess && min_level) e = hipMalloc(&dlo, 4 * nq);
  if (e == hipSuccess && max_level) e = hipMalloc(&dhi, 4 * nq);
  if (e == hipSuccess) e = hipMemcpy(dx, x, 4 * nq, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dy, y, 4 * nq, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dr, r, 4 * nq, hipMemcpyHostToDevice);
  if (e == hipSuccess && min_level) e = hipMemcpy(dlo, min_level, 4 * nq, hipMemcpyHostToDevice);
  if (e == hipSuccess && max_level) e = hipMemcpy(dhi, max_level, 4 * nq, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_grid_query, dim3((nq + 63) / 64), dim3(64), 0, 0, g->d_kps, g->d_cell_start, g->d_entries,
                       g->P, dx, dy, dr, dlo, dhi, nq, dcnt, (const int *)nullptr, (int *)nullptr, 0);
    e = hipMemcpy(cnt.data(), dcnt, 4 * nq, hipMemcpyDeviceToHost);
  }
  if (e == hipSuccess) {
    for (int q = 0; q < nq; q++) cand_off[q + 1] = cand_off[q] + cnt[q];
    total = cand_off[nq];
    if (total > 0 && cand_idx && cap > 0) {
      e = hipMalloc(&didx, 4 * (size_t)(total < cap ? total : cap));
      if (e == hipSuccess) e = hipMemcpy(doff, cand_off, 4 * (nq + 1), hipMemcpyHostToDevice);
      if (e == hipSuccess) {
        hipLaunchKernelGGL(k_grid_query, dim3((nq + 63) / 64), dim3(64), 0, 0, g->d_kps, g->d_cell_start,
                           g->d_entries, g->P, dx, dy, dr, dlo, dhi, nq, (int *)nullptr, doff, didx,
                           total < cap ? total : cap);
        e = hipMemcpy(cand_idx, didx, 4 * (size_t)(total < cap ? total : cap), hipMemcpyDeviceToHost);
      }
      if (e != hipSuccess) total = VSG_ERR_HIP;
    }
  }
  hipFree(dx), hipFree(dy), hipFree(dr), hipFree(dlo), hipFree(dhi), hipFree(dcnt), hipFree(doff), hipFree(didx);
  return total;
}

}  // extern "C"
