// ubench_lds.hip -- LDS instruction throughput on gfx950, per instruction kind and address pattern.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_lds tools/ubench_lds.hip && /tmp/ubench_lds
//
// Every wave issues blocks of 8 independent LDS instructions (immediate offsets off one address register, no VALU
// between them) and waits once per block; 8 workgroups of 256 threads per CU, so all four SIMDs of a CU feed the one
// LDS pipeline.  Reported: CU-cycles per wave64 LDS instruction (= 1 / instruction throughput of the CU's LDS pipe).
// The question behind it: k_fast_cells issues 135 LDS instructions per cell next to 800 VALU ones -- if an LDS
// instruction costs ~6 cycles of the CU's pipe whatever its width (SQ_ACTIVE_INST_LDS / SQ_INSTS_LDS = 1.45 quad-cycles
// in profiles/r03_f_pmc_sq_*), the LDS pipe is as full as the VALUs and the byte reads of the ring are worth widening.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

enum Pattern { kLinear = 0, kRing = 1, kUnaligned = 2 };

// address of the calling lane for a pattern: linear = lane * width; ring = a pixel of a 52-byte-pitch tile, lanes on
// neighbouring passers of the same few rows (what phase 2 of k_fast_cells reads); unaligned = linear + 1
__device__ __forceinline__ uint32_t lane_addr(int pattern, int width, uint32_t seed) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t a;
  if (pattern == kRing) {
    uint32_t h = (lane * 2654435761u) ^ seed;
    const uint32_t r = 3 + (lane >> 3) + ((h >> 7) & 3), c = 3 + ((h >> 11) % 40);
    a = r * 52 + c;
  } else {
    a = lane * width + (pattern == kUnaligned ? 1 : 0);
  }
  return a + wave * 4096 + 1024;  // 4 KB per wave, offsets of +-1 KB stay inside
}

#define BLOCK8(OP, ...)                                                                                         \
  asm volatile(OP " %0, %8 offset:0\n\t" OP " %1, %8 offset:52\n\t" OP " %2, %8 offset:104\n\t" OP              \
                  " %3, %8 offset:156\n\t" OP " %4, %8 offset:208\n\t" OP " %5, %8 offset:260\n\t" OP           \
                  " %6, %8 offset:312\n\t" OP " %7, %8 offset:364\n\ts_waitcnt lgkmcnt(0)"                      \
               : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7)                 \
               : "v"(a))
#define BLOCK8A(OP)                                                                                             \
  asm volatile(OP " %0, %8 offset:0\n\t" OP " %1, %8 offset:1024\n\t" OP " %2, %8 offset:2048\n\t" OP           \
                  " %3, %8 offset:3072\n\t" OP " %4, %8 offset:16\n\t" OP " %5, %8 offset:1040\n\t" OP          \
                  " %6, %8 offset:2064\n\t" OP " %7, %8 offset:3088\n\ts_waitcnt lgkmcnt(0)"                    \
               : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7)                 \
               : "v"(a))

template <int kOp>
__global__ void __launch_bounds__(256) k_lds_read(uint32_t *out, int iters, int pattern, uint32_t seed) {
  __shared__ __attribute__((aligned(16))) uint8_t s[4 * 4096 + 2048];
  for (int i = threadIdx.x; i < (int)sizeof(s) / 4; i += 256) ((uint32_t *)s)[i] = i * 2654435761u;
  __syncthreads();
  constexpr int width = kOp == 0 ? 1 : kOp == 1 ? 4 : kOp == 2 ? 8 : kOp == 3 ? 16 : 2;
  const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)s + lane_addr(pattern, width, seed);
  uint32_t acc = 0;
  if constexpr (kOp == 0 || kOp == 1 || kOp == 4) {
    uint32_t r0, r1, r2, r3, r4, r5, r6, r7;
    for (int i = 0; i < iters; ++i) {
      if constexpr (kOp == 0) BLOCK8("ds_read_u8");
      if constexpr (kOp == 1) BLOCK8("ds_read_b32");
      if constexpr (kOp == 4) BLOCK8("ds_read_u16");
      acc ^= r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
    }
  } else if constexpr (kOp == 2) {
    uint64_t r0, r1, r2, r3, r4, r5, r6, r7;
    for (int i = 0; i < iters; ++i) {
      if (pattern == kLinear)
        BLOCK8A("ds_read_b64");  // 8-byte aligned
      else
        BLOCK8("ds_read_b64");  // offsets of a 52-byte row pitch: 4-byte aligned at best
      acc ^= (uint32_t)(r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7);
    }
  } else {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 r0, r1, r2, r3, r4, r5, r6, r7;
    for (int i = 0; i < iters; ++i) {
      BLOCK8A("ds_read_b128");
      acc ^= r0.x ^ r1.y ^ r2.z ^ r3.w ^ r4.x ^ r5.y ^ r6.z ^ r7.w;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// stores: 8 per block
template <int kOp>
__global__ void __launch_bounds__(256) k_lds_write(uint32_t *out, int iters, int pattern, uint32_t seed) {
  __shared__ __attribute__((aligned(16))) uint8_t s[4 * 4096 + 2048];
  constexpr int width = kOp == 0 ? 1 : kOp == 1 ? 4 : kOp == 2 ? 8 : 16;
  const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)s + lane_addr(pattern, width, seed);
  const uint32_t v = threadIdx.x;
  const uint64_t v2 = v * 0x100000001ull;
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  const u4 v4 = {v, v, v, v};
  for (int i = 0; i < iters; ++i) {
    if constexpr (kOp == 0)
      asm volatile("ds_write_b8 %0, %1 offset:0\n\tds_write_b8 %0, %1 offset:52\n\tds_write_b8 %0, %1 offset:104\n\t"
                   "ds_write_b8 %0, %1 offset:156\n\tds_write_b8 %0, %1 offset:208\n\tds_write_b8 %0, %1 offset:260\n\t"
                   "ds_write_b8 %0, %1 offset:312\n\tds_write_b8 %0, %1 offset:364\n\ts_waitcnt lgkmcnt(0)" ::"v"(a), "v"(v));
    if constexpr (kOp == 1)
      asm volatile("ds_write_b32 %0, %1 offset:0\n\tds_write_b32 %0, %1 offset:52\n\tds_write_b32 %0, %1 offset:104\n\t"
                   "ds_write_b32 %0, %1 offset:156\n\tds_write_b32 %0, %1 offset:208\n\tds_write_b32 %0, %1 offset:260\n\t"
                   "ds_write_b32 %0, %1 offset:312\n\tds_write_b32 %0, %1 offset:364\n\ts_waitcnt lgkmcnt(0)" ::"v"(a), "v"(v));
    if constexpr (kOp == 2)
      asm volatile("ds_write_b64 %0, %1 offset:0\n\tds_write_b64 %0, %1 offset:1024\n\tds_write_b64 %0, %1 offset:2048\n\t"
                   "ds_write_b64 %0, %1 offset:3072\n\tds_write_b64 %0, %1 offset:8\n\tds_write_b64 %0, %1 offset:1032\n\t"
                   "ds_write_b64 %0, %1 offset:2056\n\tds_write_b64 %0, %1 offset:3080\n\ts_waitcnt lgkmcnt(0)" ::"v"(a), "v"(v2));
    if constexpr (kOp == 3)
      asm volatile("ds_write_b128 %0, %1 offset:0\n\tds_write_b128 %0, %1 offset:1024\n\tds_write_b128 %0, %1 offset:2048\n\t"
                   "ds_write_b128 %0, %1 offset:3072\n\tds_write_b128 %0, %1 offset:16\n\tds_write_b128 %0, %1 offset:1040\n\t"
                   "ds_write_b128 %0, %1 offset:2064\n\tds_write_b128 %0, %1 offset:3088\n\ts_waitcnt lgkmcnt(0)" ::"v"(a), "v"(v4));
    if constexpr (kOp == 4)  // ds_write2_b32: two dwords at dword offsets
      asm volatile("ds_write2_b32 %0, %1, %1 offset0:0 offset1:1\n\tds_write2_b32 %0, %1, %1 offset0:13 offset1:14\n\t"
                   "ds_write2_b32 %0, %1, %1 offset0:26 offset1:27\n\tds_write2_b32 %0, %1, %1 offset0:39 offset1:40\n\t"
                   "ds_write2_b32 %0, %1, %1 offset0:52 offset1:53\n\tds_write2_b32 %0, %1, %1 offset0:65 offset1:66\n\t"
                   "ds_write2_b32 %0, %1, %1 offset0:78 offset1:79\n\tds_write2_b32 %0, %1, %1 offset0:91 offset1:92\n\t"
                   "s_waitcnt lgkmcnt(0)" ::"v"(a), "v"(v));
  }
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = ((uint32_t *)s)[threadIdx.x];
}

// LDS reads with VALU work between them: do the two pipes overlap?  kValu independent v_pk_max_i16 per LDS instruction
template <int kValu>
__global__ void __launch_bounds__(256) k_lds_mixed(uint32_t *out, int iters, int pattern, uint32_t seed) {
  __shared__ __attribute__((aligned(16))) uint8_t s[4 * 4096 + 2048];
  for (int i = threadIdx.x; i < (int)sizeof(s) / 4; i += 256) ((uint32_t *)s)[i] = i * 2654435761u;
  __syncthreads();
  const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)s + lane_addr(pattern, 1, seed);
  uint32_t acc = 0, x0 = threadIdx.x, x1 = seed, x2 = 5, x3 = 7;
  uint32_t r0, r1, r2, r3, r4, r5, r6, r7;
  for (int i = 0; i < iters; ++i) {
    BLOCK8("ds_read_u8");
#pragma unroll
    for (int k = 0; k < kValu * 2; k++)
      asm volatile("v_pk_max_i16 %0, %0, %4\n\tv_pk_max_i16 %1, %1, %4\n\tv_pk_max_i16 %2, %2, %4\n\tv_pk_max_i16 %3, %3, %4"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)
                   : "v"(r0));
    acc ^= r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc ^ x0 ^ x1 ^ x2 ^ x3;
}

// does a mis-aligned ds_read_b32 / _b64 return the bytes at its address?
__global__ void k_lds_unaligned_check(int *bad) {
  __shared__ __attribute__((aligned(16))) uint8_t s[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) s[i] = (uint8_t)(i * 37 + 11);
  __syncthreads();
  const uint32_t off = threadIdx.x * 13 + 1;  // every residue mod 8
  const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)s + off;
  uint32_t r32;
  uint64_t r64;
  asm volatile("ds_read_b32 %0, %2\n\tds_read_b64 %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r32), "=v"(r64) : "v"(a));
  uint64_t want = 0;
  for (int b = 7; b >= 0; b--) want = (want << 8) | s[off + b];
  if (r32 != (uint32_t)want) atomicAdd(&bad[0], 1);
  if (r64 != want) atomicAdd(&bad[1], 1);
}

template <typename F>
static double time_ms(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount * 8, iters = 2000;
  const double clock_hz = p.clockRate * 1e3;
  printf("device %s CUs %d clock %.0f MHz; CU-cycles per wave64 LDS instruction, 32 waves per CU issuing\n", p.name,
         p.multiProcessorCount, clock_hz / 1e6);
  uint32_t *out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  auto report = [&](const char *name, const char *pat, double ms, double per_wave) {
    const double per_cu = (double)blocks * 4 * per_wave / p.multiProcessorCount;
    printf("%-16s %-10s %8.3f ms  %6.2f cycles\n", name, pat, ms, ms * 1e-3 * clock_hz / per_cu);
  };
  const char *pats[] = {"linear", "ring", "unaligned"};
#define RD(OP, NAME, PAT) \
  report(NAME, pats[PAT], time_ms([&] { hipLaunchKernelGGL(k_lds_read<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, PAT, 12345u); }), iters * 8.0)
#define WR(OP, NAME, PAT) \
  report(NAME, pats[PAT], time_ms([&] { hipLaunchKernelGGL(k_lds_write<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, PAT, 12345u); }), iters * 8.0)
  RD(0, "ds_read_u8", 0);
  RD(0, "ds_read_u8", 1);
  RD(4, "ds_read_u16", 0);
  RD(4, "ds_read_u16", 1);
  RD(1, "ds_read_b32", 0);
  RD(1, "ds_read_b32", 1);
  RD(1, "ds_read_b32", 2);
  RD(2, "ds_read_b64", 0);
  RD(2, "ds_read_b64", 1);
  RD(2, "ds_read_b64", 2);
  RD(3, "ds_read_b128", 0);
  WR(0, "ds_write_b8", 0);
  WR(0, "ds_write_b8", 1);
  WR(1, "ds_write_b32", 0);
  WR(1, "ds_write_b32", 2);
  WR(4, "ds_write2_b32", 0);
  WR(2, "ds_write_b64", 0);
  WR(3, "ds_write_b128", 0);
#define MX(N) \
  report("u8 + " #N " pk_max", "ring", time_ms([&] { hipLaunchKernelGGL(k_lds_mixed<N>, dim3(blocks), dim3(256), 0, 0, out, iters, 1, 12345u); }), iters * 8.0)
  MX(0);
  MX(1);
  MX(2);
  MX(4);
  MX(8);
  int *bad;
  hipMalloc(&bad, 8);
  hipMemset(bad, 0, 8);
  hipLaunchKernelGGL(k_lds_unaligned_check, dim3(1), dim3(64), 0, 0, bad);
  int hbad[2] = {-1, -1};
  hipMemcpy(hbad, bad, 8, hipMemcpyDeviceToHost);
  printf("mis-aligned ds_read_b32: %d of 64 lanes wrong; ds_read_b64: %d of 64 lanes wrong\n", hbad[0], hbad[1]);
  hipFree(bad);
  hipFree(out);
  return 0;
}
