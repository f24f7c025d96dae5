// Patch-load micro-benchmark for the TILED blurred-level layout: what k_orient_desc's 37 x 37 patch costs
// a CU when the level is stored as 16 x 4-pixel tiles (one 64-byte line each) and the load's lanes are mapped to whole
// tiles -- lane = 4 * tile + tile row, 16 bytes per lane, so every group of four lanes reads one full line -- against the
// present raster layout read as 4-byte-aligned qwords, 5 lanes per row (the form the kernel uses).  Both variants also
// write the patch into LDS the way the kernel would (aligned b128 rows at pitch 80 / b64 rows at pitch 40) and read 8
// bytes per lane back, so the LDS side of the change is priced too.  Images: `nimg` levels of 752 x 480 -- 64 of them stay
// in the L2s (23 MB), 1024 (368 MB) come from HBM past the Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/ubench_tile tools/ubench_tile.hip && tools/_bin/ubench_tile
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x2 u32x2a4 __attribute__((aligned(4)));
typedef __attribute__((address_space(1))) const u32x2a4 g_u32x2;
typedef __attribute__((address_space(1))) const u32x4 g_u32x4;

__device__ __forceinline__ void next_xy(uint32_t &seed, int cols, int rows, int &x, int &y) {
  seed = seed * 1664525u + 1013904223u;
  x = 32 + (int)((seed >> 8) % (uint32_t)(cols - 128));
  y = 32 + (int)((seed >> 20) % (uint32_t)(rows - 128));
  x = __builtin_amdgcn_readfirstlane(x), y = __builtin_amdgcn_readfirstlane(y);
}

// raster: 37 rows x 5 qwords from the origin rounded down to 4 bytes, 12 rows per instruction, 4 instructions
__global__ void __launch_bounds__(256) k_raster(const uint8_t *img, int pitch, int rows, int cols, int iters,
                                                uint32_t *out, int nimg) {
  __shared__ __attribute__((aligned(16))) uint8_t s_patch[4][37 * 40];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t seed = (blockIdx.x * 4 + wv) * 2654435761u + 12345u;
  const int r0 = lane / 5, cc = lane - r0 * 5;
  const bool act = r0 < 12;
  uint32_t acc = 0;
  for (int i = 0; i < iters; i++) {
    int x, y;
    next_xy(seed, cols, rows, x, y);
    const uint8_t *base = img + (size_t)((blockIdx.x * 7 + i) % nimg) * pitch * rows + (size_t)y * pitch + (x & ~3) + 8 * cc;
    u32x2 pv[4];
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int r = it < 3 ? it * 12 + r0 : 36;
      pv[it] = act ? *(g_u32x2 *)(base + (uint32_t)(r * pitch)) : (u32x2){0u, 0u};
    }
#pragma unroll
    for (int it = 0; it < 4; it++)
      if (act && (it < 3 || r0 == 0)) *(u32x2 *)&s_patch[wv][(it * 12 + r0) * 40 + 8 * cc] = pv[it];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 8; k++) acc += s_patch[wv][((lane * 7 + k * 5) % 37) * 40 + ((lane + k * 11) % 37)];
    __builtin_amdgcn_wave_barrier();
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// tiled 16 x 4: tiles (tx0..tx1) x (ty0..ty1) that the 37 x 37 patch touches, lane = 4 * tile + row of the tile
template <int kLdsPitch>
__global__ void __launch_bounds__(256) k_tiled(const uint8_t *img, int tiles_per_row, int rows, int cols, int iters,
                                               uint32_t *out, int nimg) {
  __shared__ __attribute__((aligned(16))) uint8_t s_patch[4][40 * kLdsPitch];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t seed = (blockIdx.x * 4 + wv) * 2654435761u + 12345u;
  const int sub = lane & 3, tg = lane >> 2;
  const size_t level_bytes = (size_t)tiles_per_row * 64 * (rows / 4);
  uint32_t acc = 0;
  for (int i = 0; i < iters; i++) {
    int x, y;
    next_xy(seed, cols, rows, x, y);
    const int tx0 = x >> 4, ntx = ((x + 36) >> 4) - tx0 + 1, ty0 = y >> 2;  // 3 or 4 tile columns, always 10 tile rows
    const int nt = ntx * 10;
    const uint8_t *base = img + (size_t)((blockIdx.x * 7 + i) % nimg) * level_bytes;
    const float inv = ntx == 3 ? 1.0f / 3.0f : 0.25f;
    u32x4 pv[3];
    int lo[3];
#pragma unroll
    for (int it = 0; it < 3; it++) {
      const int t = min(it * 16 + tg, nt - 1);
      const int tyi = (int)(((float)t + 0.5f) * inv), txi = t - tyi * ntx;
      pv[it] = *(g_u32x4 *)(base + (uint32_t)(((ty0 + tyi) * tiles_per_row + tx0 + txi) * 64 + sub * 16));
      lo[it] = (4 * tyi + sub) * kLdsPitch + 16 * txi;
    }
#pragma unroll
    for (int it = 0; it < 3; it++)
      if (it * 16 + tg < nt) *(u32x4 *)&s_patch[wv][lo[it]] = pv[it];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int oy = y & 3, ox = x & 15;
#pragma unroll
    for (int k = 0; k < 8; k++) acc += s_patch[wv][(oy + (lane * 7 + k * 5) % 37) * kLdsPitch + ox + ((lane + k * 11) % 37)];
    __builtin_amdgcn_wave_barrier();
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename F>
static double time_ms(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 3;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int rows = 480, cols = 752, pitch = 768, tiles_per_row = 768 / 16;
  const int max_img = 1024;
  uint8_t *img;
  hipMalloc(&img, (size_t)pitch * rows * max_img);
  hipMemset(img, 7, (size_t)pitch * rows * max_img);
  const int blocks = p.multiProcessorCount * 8, iters = 400;
  uint32_t *out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  const double clk = p.clockRate * 1e3;
  printf("device CUs %d clock %.0f MHz; cycles are per PATCH per CU (8 blocks x 4 waves per CU resident)\n",
         p.multiProcessorCount, clk / 1e6);
  for (int nimg : {64, 1024}) {
    auto report = [&](const char *name, double ms) {
      const double patches_per_cu = (double)blocks / p.multiProcessorCount * 4 * iters;
      printf("%-52s %5d images  %7.3f ms  %7.1f cycles/patch\n", name, nimg, ms, ms * 1e-3 * clk / patches_per_cu);
    };
    report("raster, 4-aligned qwords, 5 lanes/row (4 instr)",
           time_ms([&] { hipLaunchKernelGGL(k_raster, dim3(blocks), dim3(256), 0, 0, img, pitch, rows, cols, iters, out, nimg); }));
    report("tiles 16x4, lane = tile row, 16 B (3 instr), LDS 64",
           time_ms([&] { hipLaunchKernelGGL(k_tiled<64>, dim3(blocks), dim3(256), 0, 0, img, tiles_per_row, rows, cols, iters, out, nimg); }));
    report("tiles 16x4, lane = tile row, 16 B (3 instr), LDS 80",
           time_ms([&] { hipLaunchKernelGGL(k_tiled<80>, dim3(blocks), dim3(256), 0, 0, img, tiles_per_row, rows, cols, iters, out, nimg); }));
  }
  hipFree(img);
  hipFree(out);
  return 0;
}
