// undistort_oracle.cpp -- CPU restatement of Frame::UndistortKeyPoints / Frame::ComputeImageBounds and of the
// cv::undistortPoints call they make.  TEST INFRASTRUCTURE (harness for the grid / search parity tests): only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; the product never does.
//
// **Parity unpinned** for the OpenCV piece: cv::undistortPoints lives in un-vendored OpenCV 4.2 (imgproc/src/undistort.
// dispatch.cpp, cvUndistortPointsInternal), absent from /root/reference and from this image; it is restated here from
// the published 4.2 algorithm and anchored on the reference's call sites:
//     orb_slam3/src/Frame.cc:891-921   UndistortKeyPoints:  undistortPoints(mat, mat, toK(), mDistCoef, cv::Mat(), mK)
//     orb_slam3/src/Frame.cc:924-955   ComputeImageBounds:  the four image corners through the same call
//     orb_slam3/src/Frame.cc:378-379   mfGridElementWidthInv / HeightInv from the bounds
// tools/pin_with_opencv/pin_dump.cpp dumps the real cv::undistortPoints on the same inputs when someone runs the pin
// recipe (tests/test_pin_opencv42.py then checks this file against it).
//
// [OCV 4.2] cv::undistortPoints(src, dst, K, D, R = empty, P) is the overload with TermCriteria(MAX_ITER, 5, 0.01): the
// type has no EPS bit, so the loop runs exactly 5 iterations and never evaluates the reprojection error.  All arithmetic
// is double; src / dst are CV_32FC2; K, D, P are converted from CV_32F to double element by element.  With k[12] = k[13]
// = 0 (no tilt) the tilt matrices are identities; with R empty, RR = P(3x3) * I.
#include <cstdint>
#include <cstring>

#include "orb_oracle.h"

namespace {

struct Cam {
  double fx, fy, cx, cy;
  double k[14];
};

Cam make_cam(const float K4[4], const float *dist, int ndist) {
  Cam c;
  c.fx = (double)K4[0], c.fy = (double)K4[1], c.cx = (double)K4[2], c.cy = (double)K4[3];
  for (double &v : c.k) v = 0.0;
  // distCoeffs order (k1, k2, p1, p2[, k3[, k4, k5, k6[, s1..s4[, tx, ty]]]]); the reference passes 4 or 5 floats
  // (Tracking.cc:742-787)
  for (int i = 0; i < ndist && i < 14; i++) c.k[i] = (double)dist[i];
  return c;
}

// one point of cvUndistortPointsInternal with R = I, P = K, criteria = (MAX_ITER, 5)
void undistort_one(const Cam &c, float xin, float yin, float *xout, float *yout) {
  const double *k = c.k;
  const double ifx = 1. / c.fx, ify = 1. / c.fy;
  double x = (double)xin, y = (double)yin;
  const double u = x, v = y;
  x = (x - c.cx) * ifx;
  y = (y - c.cy) * ify;
  {
    // compensate tilt distortion: invMatTilt = I
    const double ux = 1. * x + 0. * y + 0. * 1., uy = 0. * x + 1. * y + 0. * 1., uz = 0. * x + 0. * y + 1. * 1.;
    const double invProj = uz ? 1. / uz : 1;
    const double x0 = x = invProj * ux;
    const double y0 = y = invProj * uy;
    for (int j = 0; j < 5; j++) {
      const double r2 = x * x + y * y;
      const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
      if (icdist < 0) {  // "test: undistortPoints.regression_14583"
        x = (u - c.cx) * ifx;
        y = (v - c.cy) * ify;
        break;
      }
      const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
      const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
      x = (x0 - deltaX) * icdist;
      y = (y0 - deltaY) * icdist;
    }
  }
  // RR = P * I = K:  [fx 0 cx; 0 fy cy; 0 0 1]
  const double xx = c.fx * x + 0. * y + c.cx;
  const double yy = 0. * x + c.fy * y + c.cy;
  const double ww = 1. / (0. * x + 0. * y + 1.);
  x = xx * ww;
  y = yy * ww;
  *xout = (float)x;
  *yout = (float)y;
}

}  // namespace

extern "C" {

// cv::undistortPoints on n points (xy interleaved), K4 = {fx, fy, cx, cy} as the floats of mK, dist = mDistCoef.
void or_undistort_points(const float *xy_in, int n, const float K4[4], const float *dist, int ndist, float *xy_out) {
  const Cam c = make_cam(K4, dist, ndist);
  for (int i = 0; i < n; i++) undistort_one(c, xy_in[2 * i], xy_in[2 * i + 1], &xy_out[2 * i], &xy_out[2 * i + 1]);
}

// Frame::UndistortKeyPoints (Frame.cc:891-921): mvKeysUn = mvKeys with pt replaced; identity when mDistCoef(0) == 0.
void or_undistort_keypoints(const OrKeyPoint *keys, int n, const float K4[4], const float *dist, int ndist,
                            OrKeyPoint *keys_un) {
  if (keys_un != keys) std::memcpy(keys_un, keys, (size_t)n * sizeof(OrKeyPoint));
  if (ndist < 1 || dist[0] == 0.0f) return;
  const Cam c = make_cam(K4, dist, ndist);
  for (int i = 0; i < n; i++) undistort_one(c, keys[i].x, keys[i].y, &keys_un[i].x, &keys_un[i].y);
}

// Frame::ComputeImageBounds (Frame.cc:924-955) -> {mnMinX, mnMinY, mnMaxX, mnMaxY}
void or_image_bounds(int cols, int rows, const float K4[4], const float *dist, int ndist, float out[4]) {
  if (ndist >= 1 && dist[0] != 0.0f) {
    const float in[8] = {0.f, 0.f, (float)cols, 0.f, 0.f, (float)rows, (float)cols, (float)rows};
    float m[8];
    or_undistort_points(in, 4, K4, dist, ndist, m);
    out[0] = m[0] < m[4] ? m[0] : m[4];  // min(mat(0,0), mat(2,0))
    out[2] = m[2] > m[6] ? m[2] : m[6];  // max(mat(1,0), mat(3,0))
    out[1] = m[1] < m[3] ? m[1] : m[3];  // min(mat(0,1), mat(1,1))
    out[3] = m[5] > m[7] ? m[5] : m[7];  // max(mat(2,1), mat(3,1))
  } else {
    out[0] = 0.0f, out[2] = (float)cols, out[1] = 0.0f, out[3] = (float)rows;
  }
}

}  // extern "C"
