// vsg_math.h -- scalar arithmetic shared by the HIP kernels (device) and the host-side unit tests.
//
// Everything here must produce the same bits on gfx950 and on an x86-64 SSE2 build: no FMA
// contraction (explicit *_rn intrinsics on the device, -ffp-contract=off on the host), IEEE
// division, round-half-even conversions.  Reference arithmetic being reproduced:
//   cvRound                    [OCV] fast_math.hpp (SSE cvtss2si)
//   fastAtan2                  [OCV] mathfuncs_core.simd.hpp atan_f32, used at ORBextractor.cc:99
//   cosf/sinf of angle*factorPI  glibc >= 2.28 s_sincosf.h kernels, used at ORBextractor.cc:107-108
//   x*b + y*a, x*a - y*b       ORBextractor.cc:113-115
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VSG_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define VSG_HD inline
#endif

namespace vsg {

// (int)f as the reference's x86-64 build computes it (cvttss2si): truncation, and the "integer indefinite" value
// 0x80000000 for NaN and for anything outside [-2^31, 2^31) -- gfx950's v_cvt_i32_f32 saturates instead (and turns
// NaN into 0), which would change which grid cells a window with a wild coordinate or radius covers (Frame.cc:810-832).
VSG_HD int cvt_int_x86(float f) {
  return (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : (int)0x80000000;
}

// ---- non-contracting float / double primitives
#if defined(__HIP_DEVICE_COMPILE__)
VSG_HD float fmul(float a, float b) { return __fmul_rn(a, b); }
VSG_HD float fadd(float a, float b) { return __fadd_rn(a, b); }
VSG_HD float fsub(float a, float b) { return __fsub_rn(a, b); }
VSG_HD float fdiv(float a, float b) { return __fdiv_rn(a, b); }
VSG_HD double dmul(double a, double b) { return __dmul_rn(a, b); }
VSG_HD double dadd(double a, double b) { return __dadd_rn(a, b); }
VSG_HD double dsub(double a, double b) { return __dsub_rn(a, b); }
VSG_HD double dfma(double a, double b, double c) { return __fma_rn(a, b, c); }
VSG_HD int round_half_even(float v) { return __float2int_rn(v); }
#else
VSG_HD float fmul(float a, float b) {
  volatile float r = a * b;
  return r;
}
VSG_HD float fadd(float a, float b) {
  volatile float r = a + b;
  return r;
}
VSG_HD float fsub(float a, float b) {
  volatile float r = a - b;
  return r;
}
VSG_HD float fdiv(float a, float b) {
  volatile float r = a / b;
  return r;
}
VSG_HD double dmul(double a, double b) {
  volatile double r = a * b;
  return r;
}
VSG_HD double dadd(double a, double b) {
  volatile double r = a + b;
  return r;
}
VSG_HD double dsub(double a, double b) {
  volatile double r = a - b;
  return r;
}
VSG_HD double dfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
VSG_HD int round_half_even(float v) { return (int)__builtin_lrintf(v); }
#endif

// neg ? -v : v as ONE integer operation on the high word (the products with a run-time +-1.0 this replaces were a
// v_mul_f64 each -- FP64 issues at a fraction of the FP32 rate on gfx950; x * -1.0 == -x exactly)
VSG_HD double flip_sign(double v, bool neg) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __hiloint2double(__double2hiint(v) ^ (neg ? (int)0x80000000 : 0), __double2loint(v));
#else
  return neg ? -v : v;
#endif
}

VSG_HD uint32_t f2u(float f) {
  union {
    float f;
    uint32_t u;
  } c;
  c.f = f;
  return c.u;
}

// ---- cv::fastAtan2 scalar form, degrees in [0, 360)
VSG_HD float fast_atan2_deg(float y, float x) {
  // constants are float products folded at compile time in OpenCV: p_k * (float)(180/CV_PI)
  const float p1 = 0x1.ca44dep+5f;   // 0.9997878412794807f   * (float)(180/CV_PI)
  const float p3 = -0x1.2aaddcp+4f;  // -0.3258083974640975f  * (float)(180/CV_PI)
  const float p5 = 0x1.1d3f7ep+3f;   // 0.1555786518463281f   * (float)(180/CV_PI)
  const float p7 = -0x1.4515b2p+1f;  // -0.04432655554792128f * (float)(180/CV_PI)
  const float eps = 0x1p-52f;        // (float)DBL_EPSILON
  float ax = x < 0 ? -x : x, ay = y < 0 ? -y : y;
  float a, c, c2;
  if (ax >= ay) {
    c = fdiv(ay, fadd(ax, eps));
    c2 = fmul(c, c);
    a = fmul(fadd(fmul(fadd(fmul(fadd(fmul(p7, c2), p5), c2), p3), c2), p1), c);
  } else {
    c = fdiv(ax, fadd(ay, eps));
    c2 = fmul(c, c);
    a = fsub(90.f, fmul(fadd(fmul(fadd(fmul(fadd(fmul(p7, c2), p5), c2), p3), c2), p1), c));
  }
  if (x < 0) a = fsub(180.f, a);
  if (y < 0) a = fsub(360.f, a);
  return a;
}

// ---- glibc (>= 2.28) sinf/cosf double-precision kernels for |x| < 120.
// USE_FMA selects the contraction pattern of the x86-64 `__sinf_fma`/`__cosf_fma` ifunc variants
// (GCC -mfma contracts every a + b*c below); USE_FMA=false is the `_sse2` variant.
template <bool USE_FMA>
struct SinCosF {
  static VSG_HD double mad(double a, double b, double c) {  // a*b + c
    return USE_FMA ? dfma(a, b, c) : dadd(dmul(a, b), c);
  }
  // polynomial evaluation of s_sincosf.h sinf_poly(); n odd -> cosine polynomial
  static VSG_HD float poly(double x, double x2, bool neg_table, int n) {
    const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2c = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10,
                 c4 = 0x1.99343027bf8c3p-16;
    const double s1c = -0x1.555545995a603p-3, s2c = 0x1.1107605230bc4p-7, s3c = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
      double x3 = dmul(x, x2);
      double s1 = mad(x2, s3c, s2c);
      double x7 = dmul(x3, x2);
      double s = mad(x3, s1c, x);
      return (float)mad(x7, s1, s);
    } else {
      // __sincosf_table[1] negates the cosine coefficients
      double x4 = dmul(x2, x2);
      double cc2 = mad(x2, flip_sign(c4, neg_table), flip_sign(c3, neg_table));
      double cc1 = mad(x2, flip_sign(c1, neg_table), flip_sign(c0, neg_table));
      double x6 = dmul(x4, x2);
      double c = mad(x4, flip_sign(c2c, neg_table), cc1);
      return (float)mad(x6, cc2, c);
    }
  }
  static VSG_HD uint32_t abstop12(float x) { return (f2u(x) >> 20) & 0x7ff; }

  // is_cos=false: sinf(y); is_cos=true: cosf(y).  Valid for |y| < 120 (angles here are in [0, 2*pi]).
  static VSG_HD float eval(float y, bool is_cos) {
    double x = y;
    const float pio4 = 0x1.921FB6p-1f;
    if (abstop12(y) < abstop12(pio4)) {
      double x2 = dmul(x, x);
      if (abstop12(y) < abstop12(0x1p-12f)) return is_cos ? 1.0f : y;
      return poly(x, x2, false, is_cos ? 1 : 0);
    }
    // reduce_fast, !TOINT_INTRINSICS form
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    double r = dmul(x, hpi_inv);
    int n = ((int32_t)r + 0x800000) >> 24;
    double xr = USE_FMA ? dfma(-(double)n, hpi, x) : dsub(x, dmul((double)n, hpi));
    const int q = n & 3;
    const bool sneg = q == 1 || q == 2;  // sign[] = {1,-1,-1,1}: x * sign[q] as a sign flip
    const bool neg = (n & 2) != 0;
    return poly(flip_sign(xr, sneg), dmul(xr, xr), neg, is_cos ? (n ^ 1) : n);
  }
};

// cosf(y) AND sinf(y) of the kernels above in ONE branch-free evaluation, for 0 <= y < 120.  eval() reduces y twice and
// runs one polynomial per call; but for a given y the two calls use the SAME reduction and the same two polynomial values
// -- sinf takes the sine polynomial when n is even and the cosine one when n is odd, cosf (n ^ 1) the other -- so one
// reduction and both polynomials give both results, swapped by n's parity.  The |y| < pi/4 shortcut of eval() is the general path
// with n = 0 (x * hpi_inv < 2^23, so n = 0 and xr = fma(-0, hpi, x) = x - 0 * hpi = x exactly); the |y| < 2^-12 shortcut
// stays as a select.  Without branches the evaluation can run with a DIFFERENT y in every lane: k_orient_desc evaluates
// the rotations of a wavefront's keypoints in one go instead of one wave-uniform evaluation per keypoint (FP64 issues
// at a fraction of the FP32 rate).  tests/test_hostcore.py compares it with eval() over the whole argument range.
template <bool USE_FMA>
VSG_HD void sincos_pair(float y, float *cos_out, float *sin_out) {
  typedef SinCosF<USE_FMA> K;
  const double x = y;
  const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
  const double r = dmul(x, hpi_inv);
  const int n = ((int32_t)r + 0x800000) >> 24;
  const double xr = USE_FMA ? dfma(-(double)n, hpi, x) : dsub(x, dmul((double)n, hpi));
  const int q = n & 3;
  const bool sneg = q == 1 || q == 2, neg = (n & 2) != 0;
  const double x2 = dmul(xr, xr);
  const float ps = K::poly(flip_sign(xr, sneg), x2, neg, 0);  // the sine polynomial
  const float pc = K::poly(xr, x2, neg, 1);                   // the cosine polynomial
  float sv = (n & 1) ? pc : ps, cv = (n & 1) ? ps : pc;
  if (K::abstop12(y) < K::abstop12(0x1p-12f)) sv = y, cv = 1.0f;
  *cos_out = cv, *sin_out = sv;
}

// fast_atan2_deg without the two-sided branch (both sides are the same polynomial of min / max); bit-identical: when
// ax == ay both forms divide the same values
VSG_HD float fast_atan2_deg_sel(float y, float x) {
  const float p1 = 0x1.ca44dep+5f, p3 = -0x1.2aaddcp+4f, p5 = 0x1.1d3f7ep+3f, p7 = -0x1.4515b2p+1f, eps = 0x1p-52f;
  const float ax = x < 0 ? -x : x, ay = y < 0 ? -y : y;
  const bool swap = !(ax >= ay);
  const float num = swap ? ax : ay, den = swap ? ay : ax;
  const float c = fdiv(num, fadd(den, eps));
  const float c2 = fmul(c, c);
  float a = fmul(fadd(fmul(fadd(fmul(fadd(fmul(p7, c2), p5), c2), p3), c2), p1), c);
  if (swap) a = fsub(90.f, a);
  if (x < 0) a = fsub(180.f, a);
  if (y < 0) a = fsub(360.f, a);
  return a;
}

#ifndef VSG_SINCOS_FMA
#define VSG_SINCOS_FMA 1
#endif
typedef SinCosF<(VSG_SINCOS_FMA != 0)> SinCos;

// a = cosf(angle*factorPI), b = sinf(angle*factorPI)   (ORBextractor.cc:102,107-108)
VSG_HD void brief_rotation(float angle_deg, float *a, float *b) {
  const float factorPI = 0x1.1df46ap-6f;  // (float)(CV_PI / 180.f)
  float ang = fmul(angle_deg, factorPI);
  *a = SinCos::eval(ang, true);
  *b = SinCos::eval(ang, false);
}

// the same from the moments, in a form every lane can run on its own values: angle = fastAtan2(m01, m10), then a, b
VSG_HD void brief_rotation_of_moments(float m01, float m10, float *angle_deg, float *a, float *b) {
  const float factorPI = 0x1.1df46ap-6f;
  const float ang_deg = fast_atan2_deg_sel(m01, m10);
  sincos_pair<(VSG_SINCOS_FMA != 0)>(fmul(ang_deg, factorPI), a, b);
  *angle_deg = ang_deg;
}

// rotated, rounded sample offset of pattern point (px,py)  (ORBextractor.cc:113-115)
VSG_HD void brief_offset(int px, int py, float a, float b, int *dx, int *dy) {
  float fx = (float)px, fy = (float)py;
  *dy = round_half_even(fadd(fmul(fx, b), fmul(fy, a)));
  *dx = round_half_even(fsub(fmul(fx, a), fmul(fy, b)));
}

// Hamming distance of two 256-bit descriptors held as 8 x u32 (ORBmatcher.cc:2047-2063)
VSG_HD int popc32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __popc(v);
#else
  return __builtin_popcount(v);
#endif
}

}  // namespace vsg
