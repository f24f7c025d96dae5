// Instruction-throughput micro-benchmark for gfx950: how many wave64 issues per cycle per SIMD each of the
// integer/byte instructions the ORB kernels lean on can sustain.  Used to price instruction mixes (DESIGN.md
// section 6); not part of the product library.
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_out/ubench tools/ubench.hip && gpurun_out/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(x) x x x x x x x x
#define BODY(INSTR)                                                                                               \
  for (int i = 0; i < iters; ++i) {                                                                               \
    REP8(asm volatile(INSTR : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)      \
                      : "v"(b), "v"(c));)                                                                         \
  }

// each asm statement holds 8 independent instructions (one per accumulator): 64 instructions per loop trip
#define I8(op, fmt) \
  op " %0, " fmt(0) "\n" op " %1, " fmt(1) "\n" op " %2, " fmt(2) "\n" op " %3, " fmt(3) "\n" \
  op " %4, " fmt(4) "\n" op " %5, " fmt(5) "\n" op " %6, " fmt(6) "\n" op " %7, " fmt(7) "\n"

#define F_AB(n) "%" #n ", %8"
#define F_ABC(n) "%" #n ", %8, %9"
#define F_A(n) "%" #n

template <int WHICH>
__global__ void __launch_bounds__(256) k(uint32_t *out, int iters, uint32_t b, uint32_t c) {
  uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  if (WHICH == 0) { BODY(I8("v_add_u32", F_AB)) }
  if (WHICH == 1) { BODY(I8("v_mul_lo_u32", F_AB)) }
  if (WHICH == 2) { BODY(I8("v_mul_u32_u24", F_AB)) }
  if (WHICH == 3) { BODY(I8("v_mad_u32_u24", F_ABC)) }
  if (WHICH == 4) { BODY(I8("v_pk_max_i16", F_AB)) }
  if (WHICH == 5) { BODY(I8("v_dot4_u32_u8", F_ABC)) }
  if (WHICH == 6) { BODY(I8("v_alignbyte_b32", F_ABC)) }
  if (WHICH == 7) { BODY(I8("v_perm_b32", F_ABC)) }
  if (WHICH == 8) { BODY(I8("v_cvt_f32_u32", F_A)) }
  if (WHICH == 9) { BODY(I8("v_mul_hi_u32", F_AB)) }
  if (WHICH == 10) { BODY(I8("v_bfe_u32", F_ABC)) }
  if (WHICH == 11) { BODY(I8("v_lshl_add_u32", F_ABC)) }
  if (WHICH == 12) { BODY(I8("v_mul_f32", F_AB)) }
  if (WHICH == 13) { BODY(I8("v_pk_mul_lo_u16", F_AB)) }
  if (WHICH == 14) { BODY(I8("v_mad_u32_u16", F_ABC)) }
  if (WHICH == 15) { BODY(I8("v_sad_u8", F_ABC)) }
  if (WHICH == 16) { BODY(I8("v_bcnt_u32_b32", F_AB)) }
  if (WHICH == 17) { BODY(I8("v_max3_i32", F_ABC)) }
  if (WHICH == 18) { BODY(I8("v_med3_i32", F_ABC)) }
  if (WHICH == 19) { BODY(I8("v_pk_add_u16", F_AB)) }
  if (WHICH == 20) { BODY(I8("v_and_b32", F_AB)) }
  if (WHICH == 21) { BODY(I8("v_or_b32", F_AB)) }
  if (WHICH == 22) { BODY(I8("v_xor_b32", F_AB)) }
  if (WHICH == 23) { BODY(I8("v_lshlrev_b32", F_AB)) }
  if (WHICH == 24) { BODY(I8("v_lshrrev_b32", F_AB)) }
  if (WHICH == 25) { BODY(I8("v_min_u32", F_AB)) }
  if (WHICH == 26) { BODY(I8("v_max_i32", F_AB)) }
  if (WHICH == 27) { BODY(I8("v_sub_u32", F_AB)) }
  if (WHICH == 28) { BODY(I8("v_add_f32", F_AB)) }
  if (WHICH == 29) { BODY(I8("v_fma_f32", F_ABC)) }
  if (WHICH == 30) { BODY(I8("v_and_or_b32", F_ABC)) }
  if (WHICH == 31) { BODY(I8("v_add3_u32", F_ABC)) }
  if (WHICH == 32) { BODY(I8("v_min_i16", F_AB)) }
  if (WHICH == 33) { BODY(I8("v_add_u16", F_AB)) }
  if (WHICH == 34) { BODY(I8("v_mov_b32", F_A)) }
  if (WHICH == 35) { BODY(I8("v_ashrrev_i32", F_AB)) }
  if (WHICH == 36) { BODY(I8("v_max_u32", F_AB)) }
  if (WHICH == 37) { BODY(I8("v_subrev_u32", F_AB)) }
#define F_ABC_BITOP(n) "%" #n ", %8, %9 bitop3:0xa8"
  if (WHICH == 38) { BODY(I8("v_bitop3_b32", F_ABC_BITOP)) }
  if (WHICH == 39) { BODY(I8("v_bfi_b32", F_ABC)) }
  if (WHICH == 40) { BODY(I8("v_ffbl_b32", F_A)) }
  if (WHICH == 41) { BODY(I8("v_pk_mad_i16", F_ABC)) }
  if (WHICH == 42) { BODY(I8("v_lshl_or_b32", F_ABC)) }
  if (WHICH == 43) { BODY(I8("v_cvt_i32_f32", F_A)) }
  if (WHICH == 44) { BODY(I8("v_mbcnt_lo_u32_b32", F_AB)) }
  if (WHICH == 45) { BODY(I8("v_or3_b32", F_ABC)) }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// 64-bit mad and sdwa forms need their own operand shapes
__global__ void __launch_bounds__(256) k_mad64(uint64_t *out, int iters, uint32_t b, uint32_t c) {
  uint64_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  for (int i = 0; i < iters; ++i) {
    REP8(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\nv_mad_u64_u32 %1, vcc, %4, %5, %1\n"
                      "v_mad_u64_u32 %2, vcc, %4, %5, %2\nv_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                      "v_mad_u64_u32 %0, vcc, %4, %5, %0\nv_mad_u64_u32 %1, vcc, %4, %5, %1\n"
                      "v_mad_u64_u32 %2, vcc, %4, %5, %2\nv_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                      : "v"(b), "v"(c)
                      : "vcc");)
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}

// FP64 vector ops (the glibc sinf / cosf kernels of k_orient_desc run in double) and the conversions around them
template <int WHICH>
__global__ void __launch_bounds__(256) k_f64(double *out, int iters, double b, double c) {
  double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  if (WHICH == 0) { BODY(I8("v_fma_f64", F_ABC)) }
  if (WHICH == 1) { BODY(I8("v_mul_f64", F_AB)) }
  if (WHICH == 2) { BODY(I8("v_add_f64", F_AB)) }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int WHICH>
__global__ void __launch_bounds__(256) k_cvt64(uint32_t *out, int iters, double b) {
  uint32_t r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double d[8];
  for (int k = 0; k < 8; k++) d[k] = b + threadIdx.x + k;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
      for (int k = 0; k < 8; k++) {
        if (WHICH == 0) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r[k]) : "v"(d[k]));
        if (WHICH == 1) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r[k]) : "v"(d[k]));
        if (WHICH == 2) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[k]) : "v"(r[k]));
      }
  }
  uint32_t acc = 0;
  for (int k = 0; k < 8; k++) acc += r[k] + (uint32_t)d[k];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ void __launch_bounds__(256) k_sdwa(uint32_t *out, int iters, uint32_t b, uint32_t c) {
  uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
#define F_SDWA(n) "%" #n ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2"
  BODY(I8("v_max_i32_sdwa", F_SDWA))
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// compare + select pair (v_cmp writes vcc, v_cndmask reads it), and a fully dependent add chain
__global__ void __launch_bounds__(256) k_cmpsel(uint32_t *out, int iters, uint32_t b, uint32_t c) {
  uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  for (int i = 0; i < iters; ++i) {
    REP8(asm volatile("v_cmp_lt_u32 vcc, %0, %4\nv_cndmask_b32 %0, %0, %5, vcc\n"
                      "v_cmp_lt_u32 vcc, %1, %4\nv_cndmask_b32 %1, %1, %5, vcc\n"
                      "v_cmp_lt_u32 vcc, %2, %4\nv_cndmask_b32 %2, %2, %5, vcc\n"
                      "v_cmp_lt_u32 vcc, %3, %4\nv_cndmask_b32 %3, %3, %5, vcc\n"
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                      : "v"(b), "v"(c)
                      : "vcc");)
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}

template <int DEP>
__global__ void __launch_bounds__(256) k_chain(uint32_t *out, int iters, uint32_t b, uint32_t c) {
  uint32_t a0 = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    if (DEP == 0) {
      REP8(asm volatile("v_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\n"
                        "v_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\n"
                        : "+v"(a0) : "v"(b));)
    } else {
      REP8(asm volatile("v_perm_b32 %0, %0, %1, %2\nv_perm_b32 %0, %0, %1, %2\nv_perm_b32 %0, %0, %1, %2\n"
                        "v_perm_b32 %0, %0, %1, %2\nv_perm_b32 %0, %0, %1, %2\nv_perm_b32 %0, %0, %1, %2\n"
                        "v_perm_b32 %0, %0, %1, %2\nv_perm_b32 %0, %0, %1, %2\n"
                        : "+v"(a0) : "v"(b), "v"(c));)
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0;
}

// MFMA i8 32x32x32: NACC independent accumulators per wave (1 = fully dependent chain)
typedef int mi32x4 __attribute__((ext_vector_type(4)));
typedef int mi32x16 __attribute__((ext_vector_type(16)));
// the block-scaled FP4 form (e2m1: 0x2 = +1, 0xA = -1; scales 127 = 2^0): 32 x 32 x 64 per instruction, exact in fp32 for +-1 data
typedef int mi32x8 __attribute__((ext_vector_type(8)));
typedef float mf32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) k_mfma_f4(float *out, int iters, int seed) {
  mi32x8 a = {seed, seed + 1, seed + 2, seed + 3, 0, 0, 0, 0}, b = {seed + 4, seed + 5, seed + 6, (int)threadIdx.x, 0, 0, 0, 0};
  mf32x16 acc[NACC];
  for (int n = 0; n < NACC; n++)
    for (int g = 0; g < 16; g++) acc[n][g] = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
      for (int n = 0; n < NACC; n++)
        acc[n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[n], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
  }
  float r = 0;
  for (int n = 0; n < NACC; n++)
    for (int g = 0; g < 16; g++) r += acc[n][g];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int NACC>
__global__ void __launch_bounds__(256) k_mfma_i8(int *out, int iters, int seed) {
  mi32x4 a = {seed, seed + 1, seed + 2, seed + 3}, b = {seed + 4, seed + 5, seed + 6, (int)threadIdx.x};
  mi32x16 acc[NACC];
  for (int n = 0; n < NACC; n++)
    for (int g = 0; g < 16; g++) acc[n][g] = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
      for (int n = 0; n < NACC; n++) acc[n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[n], 0, 0, 0);
  }
  int r = 0;
  for (int n = 0; n < NACC; n++)
    for (int g = 0; g < 16; g++) r += acc[n][g];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

__global__ void __launch_bounds__(256) k_lds(uint32_t *out, int iters, int width) {
  __shared__ uint32_t s[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) s[i] = i;
  __syncthreads();
  uint32_t acc = 0;
  int base = threadIdx.x * width;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (width == 1) acc += s[(base + u * 256 + i) & 4095];
      if (width == 2) { uint2 v = *(const uint2 *)&s[(base + u * 512 + 2 * i) & 4094]; acc += v.x + v.y; }
      if (width == 4) { uint4 v = *(const uint4 *)&s[(base + u * 1024 + 4 * i) & 4092]; acc += v.x + v.y + v.z + v.w; }
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
  printf("device %s CUs %d clock %.0f MHz\n", p.name, p.multiProcessorCount, clock_hz / 1e6);
  uint32_t *out;
  hipMalloc(&out, (size_t)blocks * 256 * 8);
  const char *names[] = {"v_add_u32", "v_mul_lo_u32", "v_mul_u32_u24", "v_mad_u32_u24", "v_pk_max_i16",
                         "v_dot4_u32_u8", "v_alignbyte_b32", "v_perm_b32", "v_cvt_f32_u32", "v_mul_hi_u32",
                         "v_bfe_u32", "v_lshl_add_u32", "v_mul_f32", "v_pk_mul_lo_u16", "v_mad_u32_u16",
                         "v_sad_u8", "v_bcnt_u32_b32", "v_max3_i32", "v_med3_i32", "v_pk_add_u16",
                         "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_min_u32",
                         "v_max_i32", "v_sub_u32", "v_add_f32", "v_fma_f32", "v_and_or_b32", "v_add3_u32",
                         "v_min_i16", "v_add_u16", "v_mov_b32", "v_ashrrev_i32", "v_max_u32", "v_subrev_u32",
                         "v_bitop3_b32", "v_bfi_b32", "v_ffbl_b32", "v_pk_mad_i16", "v_lshl_or_b32", "v_cvt_i32_f32",
                         "v_mbcnt_lo_u32_b32", "v_or3_b32"};
  auto report = [&](const char *name, double ms, double instr_per_wave) {
    const double waves = (double)blocks * 4;
    const double per_simd = waves * instr_per_wave / (p.multiProcessorCount * 4);
    const double cycles = ms * 1e-3 * clock_hz;
    printf("%-18s %8.3f ms  %.2f cycles per wave64 instruction per SIMD\n", name, ms, cycles / per_simd);
  };
#define RUN(W) report(names[W], time_ms([&] { hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u, 5u); }), iters * 64.0);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14)
  RUN(15) RUN(16) RUN(17) RUN(18) RUN(19) RUN(20) RUN(21) RUN(22) RUN(23) RUN(24) RUN(25) RUN(26) RUN(27) RUN(28)
  RUN(29) RUN(30) RUN(31) RUN(32) RUN(33) RUN(34) RUN(35) RUN(36) RUN(37) RUN(38) RUN(39) RUN(40) RUN(41) RUN(42)
  RUN(43) RUN(44) RUN(45)
  report("v_mad_u64_u32", time_ms([&] { hipLaunchKernelGGL(k_mad64, dim3(blocks), dim3(256), 0, 0, (uint64_t *)out, iters, 3u, 5u); }), iters * 64.0);
  report("v_fma_f64", time_ms([&] { hipLaunchKernelGGL(k_f64<0>, dim3(blocks), dim3(256), 0, 0, (double *)out, iters, 1.25, 0.5); }), iters * 64.0);
  report("v_mul_f64", time_ms([&] { hipLaunchKernelGGL(k_f64<1>, dim3(blocks), dim3(256), 0, 0, (double *)out, iters, 1.25, 0.5); }), iters * 64.0);
  report("v_add_f64", time_ms([&] { hipLaunchKernelGGL(k_f64<2>, dim3(blocks), dim3(256), 0, 0, (double *)out, iters, 1.25, 0.5); }), iters * 64.0);
  report("v_cvt_f32_f64", time_ms([&] { hipLaunchKernelGGL(k_cvt64<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.25); }), iters * 64.0);
  report("v_cvt_i32_f64", time_ms([&] { hipLaunchKernelGGL(k_cvt64<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.25); }), iters * 64.0);
  report("v_cvt_f64_f32", time_ms([&] { hipLaunchKernelGGL(k_cvt64<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.25); }), iters * 64.0);
  report("v_max_i32_sdwa", time_ms([&] { hipLaunchKernelGGL(k_sdwa, dim3(blocks), dim3(256), 0, 0, out, iters, 3u, 5u); }), iters * 64.0);
  report("v_cmp+v_cndmask", time_ms([&] { hipLaunchKernelGGL(k_cmpsel, dim3(blocks), dim3(256), 0, 0, out, iters, 3u, 5u); }), iters * 64.0);
  report("dep chain v_add", time_ms([&] { hipLaunchKernelGGL(k_chain<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u, 5u); }), iters * 64.0);
  report("dep chain v_perm", time_ms([&] { hipLaunchKernelGGL(k_chain<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u, 5u); }), iters * 64.0);
  report("mfma_i32_32x32x32_i8 x1", time_ms([&] { hipLaunchKernelGGL(k_mfma_i8<1>, dim3(blocks), dim3(256), 0, 0, (int *)out, iters / 8, 3); }), iters / 8 * 8.0);
  report("mfma_i32_32x32x32_i8 x2", time_ms([&] { hipLaunchKernelGGL(k_mfma_i8<2>, dim3(blocks), dim3(256), 0, 0, (int *)out, iters / 8, 3); }), iters / 8 * 16.0);
  report("mfma_i32_32x32x32_i8 x4", time_ms([&] { hipLaunchKernelGGL(k_mfma_i8<4>, dim3(blocks), dim3(256), 0, 0, (int *)out, iters / 8, 3); }), iters / 8 * 32.0);
  report("mfma_scale_f32_32x32x64_f4 x1", time_ms([&] { hipLaunchKernelGGL(k_mfma_f4<1>, dim3(blocks), dim3(256), 0, 0, (float *)out, iters / 8, 3); }), iters / 8 * 8.0);
  report("mfma_scale_f32_32x32x64_f4 x2", time_ms([&] { hipLaunchKernelGGL(k_mfma_f4<2>, dim3(blocks), dim3(256), 0, 0, (float *)out, iters / 8, 3); }), iters / 8 * 16.0);
  for (int w : {1, 2, 4})
    report(w == 1 ? "ds_read_b32" : w == 2 ? "ds_read_b64" : "ds_read_b128",
           time_ms([&] { hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), 0, 0, out, iters, w); }), iters * 8.0);
  hipFree(out);
  return 0;
}
