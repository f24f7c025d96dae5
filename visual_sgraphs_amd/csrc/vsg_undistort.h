// vsg_undistort.h -- Frame::UndistortKeyPoints / Frame::ComputeImageBounds (orb_slam3/src/Frame.cc:891-955) for the
// pinhole + (k1, k2, p1, p2[, k3]) cameras of the reference's settings files: the arithmetic of the one OpenCV call both
// make,  cv::undistortPoints(mat, mat, toK(), mDistCoef, cv::Mat(), mK)  -- [OCV 4.2] cvUndistortPointsInternal with
// R = I, P = K and TermCriteria(MAX_ITER, 5, 0.01): exactly five fixed-point iterations in double, no convergence test,
// float in / float out.  One source for the device (k_frame_grid_build: a frame that comes device-to-device out of the
// extractor is undistorted where it lies, FP64 on the vector units) and for the host (the four corners of
// ComputeImageBounds, once per camera).  Plain +, -, *, / on doubles: the library is built with -ffp-contract=off, the
// device's f64 division is the correctly rounded expansion, so both sides produce the bits an SSE2 build produces.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define VSG_UD_HD __host__ __device__
#else
#define VSG_UD_HD
#endif

namespace vsg {

struct CamModel {
  double fx, fy, cx, cy, ifx, ify;
  double k1, k2, p1, p2, k3;
  int distorted;  // mDistCoef.at<float>(0) != 0.0 (Frame.cc:893, 926): otherwise both routines are copies
};

// K4 = {fx, fy, cx, cy} (the floats of mK), dist = the 4 or 5 floats of mDistCoef (Tracking.cc:742-787)
inline bool make_cam_model(const float K4[4], const float *dist, int ndist, CamModel *m) {
  if (!K4 || ndist < 0 || ndist > 5 || (ndist > 0 && !dist) || !(K4[0] != 0.f) || !(K4[1] != 0.f)) return false;
  m->fx = (double)K4[0], m->fy = (double)K4[1], m->cx = (double)K4[2], m->cy = (double)K4[3];
  m->ifx = 1. / m->fx, m->ify = 1. / m->fy;
  const double d[5] = {ndist > 0 ? (double)dist[0] : 0., ndist > 1 ? (double)dist[1] : 0., ndist > 2 ? (double)dist[2] : 0.,
                       ndist > 3 ? (double)dist[3] : 0., ndist > 4 ? (double)dist[4] : 0.};
  m->k1 = d[0], m->k2 = d[1], m->p1 = d[2], m->p2 = d[3], m->k3 = d[4];
  m->distorted = ndist > 0 && dist[0] != 0.0f;
  return true;
}

// one point through cv::undistortPoints(.., K, D, Mat(), K).  The rational terms k4..k6, the thin-prism terms and the
// tilt are zero for a 4 / 5 coefficient vector: the denominator's numerator is exactly 1 and the tilt / rectification
// matrices multiply by exact ones and zeros, which leaves the values unchanged.
VSG_UD_HD inline void undistort_point(const CamModel &c, float xin, float yin, float *xout, float *yout) {
  const double u = (double)xin, v = (double)yin;
  double x = (u - c.cx) * c.ifx, y = (v - c.cy) * c.ify;
  const double x0 = x, y0 = y;
#pragma unroll 1
  for (int j = 0; j < 5; j++) {
    const double r2 = x * x + y * y;
    const double icdist = 1. / (1. + ((c.k3 * r2 + c.k2) * r2 + c.k1) * r2);
    if (icdist < 0) {
      x = (u - c.cx) * c.ifx;
      y = (v - c.cy) * c.ify;
      break;
    }
    const double deltaX = 2 * c.p1 * x * y + c.p2 * (r2 + 2 * x * x);
    const double deltaY = c.p1 * (r2 + 2 * y * y) + 2 * c.p2 * x * y;
    x = (x0 - deltaX) * icdist;
    y = (y0 - deltaY) * icdist;
  }
  *xout = (float)(c.fx * x + c.cx);
  *yout = (float)(c.fy * y + c.cy);
}

}  // namespace vsg
