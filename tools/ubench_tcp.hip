// Vector-L1 (TCP) throughput micro-benchmark for gfx950: what a wave64 PATCH load costs as a function of the
// per-lane width, the lanes-per-row layout and the byte alignment of the patch origin.  k_orient_desc reads two small
// 2-D patches per keypoint (31 x 31 un-blurred, 37 x 37 blurred) and sits on the L1's address/tag pipeline, not on
// VALU issue or HBM, so the layout that moves a patch in the fewest L1 cycles decides its speed.
// Not part of the product library.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/ubench_tcp tools/ubench_tcp.hip && tools/_bin/ubench_tcp
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u32u __attribute__((aligned(1)));
typedef __attribute__((address_space(1))) const u32u g_u32;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x2 u32x2u __attribute__((aligned(1)));
typedef u32x4 u32x4u __attribute__((aligned(1)));
typedef __attribute__((address_space(1))) const u32x2u g_u32x2;
typedef __attribute__((address_space(1))) const u32x4u g_u32x4;
typedef __attribute__((address_space(1))) const uint8_t g_u8;

// W = bytes per lane (1, 4, 8, 16); LPR = lanes per patch row; ROWS = patch rows.  A "patch" takes
// ceil(ROWS * LPR / 64) load instructions.  Patch origins are pseudo-random inside one image (L2-resident); ALIGN = -1
// leaves the x origin at any byte, 0..3 forces x % 4 == ALIGN.
template <int W, int LPR, int ROWS>
__global__ void __launch_bounds__(256) k_patch(const uint8_t *img, int pitch, int rows, int cols, int align, int iters,
                                               uint32_t *out, int nimg) {
  const int lane = threadIdx.x & 63;
  uint32_t seed = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2654435761u + 12345u;
  constexpr int kRowsPerInstr = 64 / LPR, kInstr = (ROWS + kRowsPerInstr - 1) / kRowsPerInstr;
  const int r0 = lane / LPR, c0 = lane - r0 * LPR;
  const bool act = r0 < kRowsPerInstr;
  uint32_t acc = 0;
  for (int i = 0; i < iters; i++) {
    seed = seed * 1664525u + 1013904223u;
    int x = 32 + (int)((seed >> 8) % (uint32_t)(cols - 128)), y = 32 + (int)((seed >> 20) % (uint32_t)(rows - 128));
    if (align >= 0) x = (x & ~3) + align;
    x = __builtin_amdgcn_readfirstlane(x), y = __builtin_amdgcn_readfirstlane(y);
    const uint8_t *base = img + (size_t)(blockIdx.x % nimg) * pitch * rows + (size_t)y * pitch + x;
    if (act) {
#pragma unroll
      for (int it = 0; it < kInstr; it++) {
        const int r = min(it * kRowsPerInstr + r0, ROWS - 1);
        const uint8_t *p = base + (uint32_t)(r * pitch + c0 * W);
        if (W == 1) acc += *(g_u8 *)p;
        if (W == 4) acc += *(g_u32 *)p;
        if (W == 8) { const u32x2 v = *(g_u32x2 *)p; acc += v.x ^ v.y; }
        if (W == 16) { const u32x4 v = *(g_u32x4 *)p; acc += v.x ^ v.y ^ v.z ^ v.w; }
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// The same 37-row patch read as aligned dwords from a TILED image: 128-byte lines hold TW x TH pixel tiles (x fastest),
// so a patch touches ~ (37 / TW + 1) * (37 / TH + 1) lines instead of 37 * 1.3.
template <int TW, int TH, int LPR, int ROWS>
__global__ void __launch_bounds__(256) k_patch_tiled(const uint8_t *img, int pitch, int rows, int cols, int iters,
                                                     uint32_t *out, int nimg) {
  static_assert(TW * TH == 128, "one tile per line");
  const int lane = threadIdx.x & 63;
  uint32_t seed = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2654435761u + 12345u;
  constexpr int kRowsPerInstr = 64 / LPR, kInstr = (ROWS + kRowsPerInstr - 1) / kRowsPerInstr;
  const int r0 = lane / LPR, c0 = lane - r0 * LPR;
  const bool act = r0 < kRowsPerInstr;
  const int tiles_per_row = pitch / TW;
  uint32_t acc = 0;
  for (int i = 0; i < iters; i++) {
    seed = seed * 1664525u + 1013904223u;
    int x = 32 + (int)((seed >> 8) % (uint32_t)(cols - 128)), y = 32 + (int)((seed >> 20) % (uint32_t)(rows - 128));
    x = __builtin_amdgcn_readfirstlane(x & ~3), y = __builtin_amdgcn_readfirstlane(y);
    const uint8_t *base = img + (size_t)(blockIdx.x % nimg) * pitch * rows;
    if (act) {
#pragma unroll
      for (int it = 0; it < kInstr; it++) {
        const int yy = y + min(it * kRowsPerInstr + r0, ROWS - 1), xx = x + 4 * c0;
        const uint32_t off = (uint32_t)(((yy / TH) * tiles_per_row + xx / TW) * 128 + (yy % TH) * TW + (xx % TW));
        acc += *(const __attribute__((address_space(1))) uint32_t *)(base + off);
      }
    }
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
  const int rows = 480, cols = 752, pitch = 768, nimg = 64;  // 64 images: blocks pick one by id (L2 footprint 23 MB)
  uint8_t *img;
  hipMalloc(&img, (size_t)pitch * rows * nimg);
  hipMemset(img, 7, (size_t)pitch * rows * nimg);
  const int blocks = p.multiProcessorCount * 8, iters = 400;
  uint32_t *out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  const double clk = p.clockRate * 1e3;
  printf("device CUs %d clock %.0f MHz; cycles are per PATCH per CU (4 waves x 8 blocks per CU resident)\n",
         p.multiProcessorCount, clk / 1e6);
  auto report = [&](const char *name, int align, double ms, int instr) {
    // patches per CU = blocks/CUs * 4 waves * iters
    const double patches_per_cu = (double)blocks / p.multiProcessorCount * 4 * iters;
    const double cyc = ms * 1e-3 * clk / patches_per_cu;
    printf("%-34s align %2d  %7.3f ms  %7.1f cycles/patch  %6.1f cycles/instr\n", name, align, ms, cyc, cyc / instr);
  };
#define RUN(W, LPR, ROWS, name)                                                                                      \
  for (int al : {0, 1, -1})                                                                                          \
    report(name, al, time_ms([&] {                                                                                   \
             hipLaunchKernelGGL((k_patch<W, LPR, ROWS>), dim3(blocks), dim3(256), 0, 0, img, pitch, rows, cols, al, \
                                iters, out, nimg);                                                                       \
           }),                                                                                                       \
           (ROWS + 64 / LPR - 1) / (64 / LPR));
  // the 31 x 31 moment patch (32 bytes per row)
  RUN(1, 32, 31, "ic  u8   32 lanes/row (16 instr)")
  RUN(4, 8, 31, "ic  u32   8 lanes/row ( 4 instr)")
  RUN(8, 4, 31, "ic  u64   4 lanes/row ( 2 instr)")
  RUN(16, 2, 31, "ic  u128  2 lanes/row ( 1 instr)")
  // the 37 x 37 test patch (40 / 48 bytes per row)
  RUN(1, 64, 37, "pat u8   64 lanes/row (37 instr)")
  RUN(4, 10, 37, "pat u32  10 lanes/row ( 7 instr)")
  RUN(8, 5, 37, "pat u64   5 lanes/row ( 4 instr)")
  RUN(16, 3, 37, "pat u128  3 lanes/row ( 2 instr)")
  RUN(4, 16, 37, "pat u32  16 lanes/row (10 instr)")
  RUN(8, 8, 37, "pat u64   8 lanes/row ( 5 instr)")
  RUN(16, 4, 37, "pat u128  4 lanes/row ( 3 instr)")
#define RUNT(TW, TH, name)                                                                                     \
  report(name, 0, time_ms([&] {                                                                                 \
           hipLaunchKernelGGL((k_patch_tiled<TW, TH, 10, 37>), dim3(blocks), dim3(256), 0, 0, img, pitch, rows, \
                              cols, iters, out, nimg);                                                          \
         }),                                                                                                    \
         7);
  RUNT(128, 1, "pat u32 aligned, raster          ")
  RUNT(64, 2, "pat u32 aligned, tiles 64 x 2    ")
  RUNT(32, 4, "pat u32 aligned, tiles 32 x 4    ")
  RUNT(16, 8, "pat u32 aligned, tiles 16 x 8    ")
  RUNT(8, 16, "pat u32 aligned, tiles  8 x 16   ")
  hipFree(img);
  hipFree(out);
  return 0;
}
