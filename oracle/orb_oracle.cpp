/*
 * orb_oracle.cpp -- CPU ORACLE (test infrastructure, NOT product code).  PARITY UNPINNED.
 * See orb_oracle.h for scope.  Every function cites the reference lines it follows
 * (paths relative to /root/reference) or the OpenCV 4.2 routine it restates ([OCV]).
 *
 * Build: see oracle/Makefile (-O3 -ffp-contract=off -fno-fast-math, no -march=native,
 * mirroring the reference's CMakeLists.txt:10-14,22 so float code is plain SSE2 scalar).
 */
#include "orb_oracle.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <list>
#include <atomic>
#include <chrono>
#include <thread>
#include <utility>
#include <vector>

namespace {

// ---------------------------------------------------------------- [OCV] fast_math.hpp
// cvRound = round-half-to-even (SSE cvtss2si / cvtsd2si under the default MXCSR mode).
inline int cvRoundF(float v) { return (int)lrintf(v); }
inline int cvRoundD(double v) { return (int)lrint(v); }
inline int cvFloorD(double v) {
  int i = (int)v;
  return i - (i > v);
}
inline int cvFloorF(float v) {
  int i = (int)v;
  return i - (i > v);
}
inline int cvCeilD(double v) {
  int i = (int)v;
  return i + (i < v);
}
// saturate_cast<short>(float) = clamp(cvRound(v)) [OCV] saturate.hpp
inline short satShortF(float v) {
  int iv = cvRoundF(v);
  return (short)(iv < SHRT_MIN ? SHRT_MIN : iv > SHRT_MAX ? SHRT_MAX : iv);
}
inline uint8_t satU8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

// [OCV] borderInterpolate(p, len, BORDER_REFLECT_101)
inline int reflect101(int p, int len) {
  if ((unsigned)p < (unsigned)len) return p;
  if (len == 1) return 0;
  do {
    if (p < 0)
      p = -p;
    else
      p = 2 * len - 2 - p;
  } while ((unsigned)p >= (unsigned)len);
  return p;
}

struct Img {  // minimal stand-in for a CV_8UC1 cv::Mat view
  int rows = 0, cols = 0, step = 0;
  uint8_t *data = nullptr;
  uint8_t *ptr(int r) const { return data + (size_t)r * step; }
};

const int PATCH_SIZE = 31;       // ORBextractor.cc:69
const int HALF_PATCH_SIZE = 15;  // ORBextractor.cc:70
const int EDGE_THRESHOLD = 19;   // ORBextractor.cc:71

const signed char kBitPattern31[256 * 4] = {
#include "brief_pattern_data.inc"
};

// ---------------------------------------------------------------- [OCV] resize.cpp, INTER_LINEAR, CV_8UC1
// resizeGeneric_<HResizeLinear<uchar,int,short,2048>, VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>>
void resizeLinearU8(const Img &src, Img &dst) {
  const int sw = src.cols, sh = src.rows, dw = dst.cols, dh = dst.rows;
  const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  const int COEF = 2048;  // INTER_RESIZE_COEF_SCALE
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> ialpha(dw * 2), ibeta(dh * 2);
  int xmax = dw;
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cvFloorF(fx);
    fx -= sx;
    if (sx < 0) {
      fx = 0;
      sx = 0;
    }
    if (sx + 1 >= sw) {
      xmax = std::min(xmax, dx);
      if (sx >= sw - 1) {
        fx = 0;
        sx = sw - 1;
      }
    }
    xofs[dx] = sx;
    float cbuf0 = 1.f - fx, cbuf1 = fx;
    ialpha[dx * 2] = satShortF(cbuf0 * COEF);
    ialpha[dx * 2 + 1] = satShortF(cbuf1 * COEF);
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cvFloorF(fy);
    fy -= sy;
    yofs[dy] = sy;
    float cbuf0 = 1.f - fy, cbuf1 = fy;
    ibeta[dy * 2] = satShortF(cbuf0 * COEF);
    ibeta[dy * 2 + 1] = satShortF(cbuf1 * COEF);
  }
  std::vector<int> row0(dw), row1(dw);
  auto hresize = [&](int sy, std::vector<int> &out) {
    const uint8_t *S = src.ptr(sy);
    int dx = 0;
    for (; dx < xmax; dx++) {
      int sx = xofs[dx];
      out[dx] = S[sx] * ialpha[dx * 2] + S[sx + 1] * ialpha[dx * 2 + 1];
    }
    for (; dx < dw; dx++) out[dx] = S[xofs[dx]] * COEF;
  };
  auto clip = [](int v, int lo, int hi) { return v < lo ? lo : v >= hi ? hi - 1 : v; };
  for (int dy = 0; dy < dh; dy++) {
    int sy0 = clip(yofs[dy], 0, sh), sy1 = clip(yofs[dy] + 1, 0, sh);
    hresize(sy0, row0);
    hresize(sy1, row1);
    const int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
    uint8_t *D = dst.ptr(dy);
    for (int x = 0; x < dw; x++)
      D[x] = (uint8_t)((((b0 * (row0[x] >> 4)) >> 16) + ((b1 * (row1[x] >> 4)) >> 16) + 2) >> 2);
  }
}

// ---------------------------------------------------------------- [OCV] copy.cpp copyMakeBorder REFLECT_101
// src (w x h) may alias the interior of dst (the reference's in-place form, ORBextractor.cc:1186):
// interior first (memmove), then the frame from the interior.
void copyMakeBorder101(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride, int b) {
  for (int y = 0; y < h; y++) {
    uint8_t *d = dst + (size_t)(y + b) * dstride + b;
    const uint8_t *s = src + (size_t)y * sstride;
    if (d != s) memmove(d, s, (size_t)w);
  }
  for (int y = 0; y < h; y++) {
    uint8_t *row = dst + (size_t)(y + b) * dstride;
    for (int x = 0; x < b; x++) {
      row[x] = row[b + reflect101(x - b, w)];
      row[b + w + x] = row[b + reflect101(w + x, w)];
    }
  }
  for (int y = 0; y < b; y++) {
    memcpy(dst + (size_t)y * dstride, dst + (size_t)(b + reflect101(y - b, h)) * dstride, (size_t)w + 2 * b);
    memcpy(dst + (size_t)(b + h + y) * dstride, dst + (size_t)(b + reflect101(h + y, h)) * dstride,
           (size_t)w + 2 * b);
  }
}

// ---------------------------------------------------------------- [OCV] smooth.dispatch.cpp / smooth.simd.hpp
// GaussianBlur(Size(7,7), 2, 2, BORDER_REFLECT_101) on a continuous CV_8U Mat = fixed-point path:
// taps are ufixedpoint16 (8.8); horizontal pass -> 8.8 in uint16, vertical -> 16.16 in uint32,
// result saturate_cast<uchar>((acc + (1<<15)) >> 16).
void gaussianBlur7(const Img &src, Img &dst, const uint16_t taps[7]) {
  const int w = src.cols, h = src.rows;
  std::vector<uint16_t> hbuf((size_t)w * h);
  for (int y = 0; y < h; y++) {
    const uint8_t *S = src.ptr(y);
    for (int x = 0; x < w; x++) {
      uint32_t acc = 0;
      for (int k = 0; k < 7; k++) acc += (uint32_t)taps[k] * S[reflect101(x + k - 3, w)];
      hbuf[(size_t)y * w + x] = (uint16_t)(acc > 0xFFFFu ? 0xFFFFu : acc);  // ufixedpoint16 '+' saturates
    }
  }
  for (int y = 0; y < h; y++) {
    uint8_t *D = dst.ptr(y);
    for (int x = 0; x < w; x++) {
      uint32_t acc = 0;
      for (int k = 0; k < 7; k++) acc += (uint32_t)taps[k] * hbuf[(size_t)reflect101(y + k - 3, h) * w + x];
      D[x] = satU8((int)((acc + 32768u) >> 16));
    }
  }
}

// ---------------------------------------------------------------- [OCV] fast_score.cpp / fast.cpp (TYPE_9_16)
void makeOffsets16(int pixel[25], int rowStride) {
  static const int offsets16[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                       {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};
  int k = 0;
  for (; k < 16; k++) pixel[k] = offsets16[k][0] + offsets16[k][1] * rowStride;
  for (; k < 25; k++) pixel[k] = pixel[k - 16];
}

int cornerScore16(const uint8_t *ptr, const int pixel[], int threshold) {
  const int K = 8, N = K * 3 + 1;
  int k, v = ptr[0];
  short d[N];
  for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
  int a0 = threshold;
  for (k = 0; k < 16; k += 2) {
    int a = std::min((int)d[k + 1], (int)d[k + 2]);
    a = std::min(a, (int)d[k + 3]);
    if (a <= a0) continue;
    a = std::min(a, (int)d[k + 4]);
    a = std::min(a, (int)d[k + 5]);
    a = std::min(a, (int)d[k + 6]);
    a = std::min(a, (int)d[k + 7]);
    a = std::min(a, (int)d[k + 8]);
    a0 = std::max(a0, std::min(a, (int)d[k]));
    a0 = std::max(a0, std::min(a, (int)d[k + 9]));
  }
  int b0 = -a0;
  for (k = 0; k < 16; k += 2) {
    int b = std::max((int)d[k + 1], (int)d[k + 2]);
    b = std::max(b, (int)d[k + 3]);
    b = std::max(b, (int)d[k + 4]);
    b = std::max(b, (int)d[k + 5]);
    if (b >= b0) continue;
    b = std::max(b, (int)d[k + 6]);
    b = std::max(b, (int)d[k + 7]);
    b = std::max(b, (int)d[k + 8]);
    b0 = std::min(b0, std::max(b, (int)d[k]));
    b0 = std::min(b0, std::max(b, (int)d[k + 9]));
  }
  threshold = -b0 - 1;
  return threshold;
}

struct FastKp {
  int x, y, score;
};

// FAST_t<16>(img, keypoints, threshold, nonmax_suppression), scalar path.
void fast9_16(const Img &img, std::vector<FastKp> &keypoints, int threshold, bool nonmax) {
  const int K = 8, N = 16 + K + 1;
  int i, j, k, pixel[25];
  makeOffsets16(pixel, img.step);
  keypoints.clear();
  threshold = std::min(std::max(threshold, 0), 255);
  uint8_t threshold_tab[512];
  for (i = -255; i <= 255; i++) threshold_tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);

  const int cols = img.cols, rows = img.rows;
  if (cols <= 0 || rows <= 0) return;
  std::vector<uint8_t> bufStore((size_t)cols * 3, 0);
  std::vector<int> cpStore((size_t)(cols + 1) * 3, 0);
  uint8_t *buf[3] = {bufStore.data(), bufStore.data() + cols, bufStore.data() + 2 * cols};
  int *cpbuf[3] = {cpStore.data() + 1, cpStore.data() + 1 + (cols + 1), cpStore.data() + 1 + 2 * (cols + 1)};

  for (i = 3; i < rows - 2; i++) {
    const uint8_t *ptr = img.ptr(i) + 3;
    uint8_t *curr = buf[(i - 3) % 3];
    int *cornerpos = cpbuf[(i - 3) % 3];
    memset(curr, 0, (size_t)cols);
    int ncorners = 0;
    if (i < rows - 3) {
      for (j = 3; j < cols - 3; j++, ptr++) {
        int v = ptr[0];
        const uint8_t *tab = &threshold_tab[0] - v + 255;
        int d = tab[ptr[pixel[0]]] | tab[ptr[pixel[8]]];
        if (d == 0) continue;
        d &= tab[ptr[pixel[2]]] | tab[ptr[pixel[10]]];
        d &= tab[ptr[pixel[4]]] | tab[ptr[pixel[12]]];
        d &= tab[ptr[pixel[6]]] | tab[ptr[pixel[14]]];
        if (d == 0) continue;
        d &= tab[ptr[pixel[1]]] | tab[ptr[pixel[9]]];
        d &= tab[ptr[pixel[3]]] | tab[ptr[pixel[11]]];
        d &= tab[ptr[pixel[5]]] | tab[ptr[pixel[13]]];
        d &= tab[ptr[pixel[7]]] | tab[ptr[pixel[15]]];
        if (d & 1) {
          int vt = v - threshold, count = 0;
          for (k = 0; k < N; k++) {
            int x = ptr[pixel[k]];
            if (x < vt) {
              if (++count > K) {
                cornerpos[ncorners++] = j;
                if (nonmax) curr[j] = (uint8_t)cornerScore16(ptr, pixel, threshold);
                break;
              }
            } else
              count = 0;
          }
        }
        if (d & 2) {
          int vt = v + threshold, count = 0;
          for (k = 0; k < N; k++) {
            int x = ptr[pixel[k]];
            if (x > vt) {
              if (++count > K) {
                cornerpos[ncorners++] = j;
                if (nonmax) curr[j] = (uint8_t)cornerScore16(ptr, pixel, threshold);
                break;
              }
            } else
              count = 0;
          }
        }
      }
    }
    cornerpos[-1] = ncorners;
    if (i == 3) continue;
    const uint8_t *prev = buf[(i - 4 + 3) % 3];
    const uint8_t *pprev = buf[(i - 5 + 3) % 3];
    cornerpos = cpbuf[(i - 4 + 3) % 3];
    ncorners = cornerpos[-1];
    for (k = 0; k < ncorners; k++) {
      j = cornerpos[k];
      int score = prev[j];
      if (!nonmax || (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
                      score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1])) {
        keypoints.push_back({j, i - 1, score});
      }
    }
  }
}

// ---------------------------------------------------------------- [OCV] mathfuncs_core.simd.hpp atan_f32
const float atan2_p1 = 0.9997878412794807f * (float)(180 / M_PI);
const float atan2_p3 = -0.3258083974640975f * (float)(180 / M_PI);
const float atan2_p5 = 0.1555786518463281f * (float)(180 / M_PI);
const float atan2_p7 = -0.04432655554792128f * (float)(180 / M_PI);

float fastAtan2(float y, float x) {
  float ax = std::abs(x), ay = std::abs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// ---------------------------------------------------------------- ORBextractor.cc:73-100
float IC_Angle(const uint8_t *center, int step, const std::vector<int> &u_max) {
  int m_01 = 0, m_10 = 0;
  for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
  for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
    int v_sum = 0;
    int d = u_max[v];
    for (int u = -d; u <= d; ++u) {
      int val_plus = center[u + v * step], val_minus = center[u - v * step];
      v_sum += (val_plus - val_minus);
      m_10 += u * (val_plus + val_minus);
    }
    m_01 += v * v_sum;
  }
  return fastAtan2((float)m_01, (float)m_10);
}

// ---------------------------------------------------------------- ORBextractor.cc:102-149
const float factorPI = (float)(M_PI / 180.f);

void computeOrbDescriptor(float kpAngle, const uint8_t *center, int step, const signed char *pattern,
                          uint8_t *desc) {
  float angle = (float)kpAngle * factorPI;
  float a = (float)cosf(angle), b = (float)sinf(angle);  // `using namespace std` float overloads (:64,:108)
  auto GET_VALUE = [&](int idx) -> int {
    const int px = pattern[2 * idx], py = pattern[2 * idx + 1];
    return center[cvRoundF(px * b + py * a) * step + cvRoundF(px * a - py * b)];
  };
  for (int i = 0; i < 32; ++i, pattern += 32) {  // 16 points (x,y) per byte
    int t0, t1, val;
    t0 = GET_VALUE(0), t1 = GET_VALUE(1);
    val = t0 < t1;
    t0 = GET_VALUE(2), t1 = GET_VALUE(3);
    val |= (t0 < t1) << 1;
    t0 = GET_VALUE(4), t1 = GET_VALUE(5);
    val |= (t0 < t1) << 2;
    t0 = GET_VALUE(6), t1 = GET_VALUE(7);
    val |= (t0 < t1) << 3;
    t0 = GET_VALUE(8), t1 = GET_VALUE(9);
    val |= (t0 < t1) << 4;
    t0 = GET_VALUE(10), t1 = GET_VALUE(11);
    val |= (t0 < t1) << 5;
    t0 = GET_VALUE(12), t1 = GET_VALUE(13);
    val |= (t0 < t1) << 6;
    t0 = GET_VALUE(14), t1 = GET_VALUE(15);
    val |= (t0 < t1) << 7;
    desc[i] = (uint8_t)val;
  }
}

// ---------------------------------------------------------------- ORBextractor.cc:482-785 octree
struct Pt2i {
  int x = 0, y = 0;
};
struct OctKey {  // the fields of cv::KeyPoint the octree touches
  float x, y, response;
  int srcIndex;
};
struct ExtractorNode {
  std::vector<OctKey> vKeys;
  Pt2i UL, UR, BL, BR;
  std::list<ExtractorNode>::iterator lit;
  bool bNoMore = false;
  void DivideNode(ExtractorNode &n1, ExtractorNode &n2, ExtractorNode &n3, ExtractorNode &n4);
};

void ExtractorNode::DivideNode(ExtractorNode &n1, ExtractorNode &n2, ExtractorNode &n3, ExtractorNode &n4) {
  const int halfX = (int)ceil(static_cast<float>(UR.x - UL.x) / 2);
  const int halfY = (int)ceil(static_cast<float>(BR.y - UL.y) / 2);
  n1.UL = UL;
  n1.UR = {UL.x + halfX, UL.y};
  n1.BL = {UL.x, UL.y + halfY};
  n1.BR = {UL.x + halfX, UL.y + halfY};
  n2.UL = n1.UR;
  n2.UR = UR;
  n2.BL = n1.BR;
  n2.BR = {UR.x, UL.y + halfY};
  n3.UL = n1.BL;
  n3.UR = n1.BR;
  n3.BL = BL;
  n3.BR = {n1.BR.x, BL.y};
  n4.UL = n3.UR;
  n4.UR = n2.BR;
  n4.BL = n3.BR;
  n4.BR = BR;
  for (size_t i = 0; i < vKeys.size(); i++) {
    const OctKey &kp = vKeys[i];
    if (kp.x < n1.UR.x) {
      if (kp.y < n1.BR.y)
        n1.vKeys.push_back(kp);
      else
        n3.vKeys.push_back(kp);
    } else if (kp.y < n1.BR.y)
      n2.vKeys.push_back(kp);
    else
      n4.vKeys.push_back(kp);
  }
  if (n1.vKeys.size() == 1) n1.bNoMore = true;
  if (n2.vKeys.size() == 1) n2.bNoMore = true;
  if (n3.vKeys.size() == 1) n3.bNoMore = true;
  if (n4.vKeys.size() == 1) n4.bNoMore = true;
}

bool compareNodes(std::pair<int, ExtractorNode *> &e1, std::pair<int, ExtractorNode *> &e2) {
  if (e1.first < e2.first) return true;
  if (e1.first > e2.first) return false;
  return e1.second->UL.x < e2.second->UL.x;
}

std::vector<OctKey> DistributeOctTree(const std::vector<OctKey> &vToDistributeKeys, int minX, int maxX, int minY,
                                      int maxY, int N) {
  std::vector<OctKey> vResultKeys;
  const int nIni = (int)round(static_cast<float>(maxX - minX) / (maxY - minY));
  if (nIni <= 0) return vResultKeys;  // reference would divide by zero / index out of range here
  const float hX = static_cast<float>(maxX - minX) / nIni;
  std::list<ExtractorNode> lNodes;
  std::vector<ExtractorNode *> vpIniNodes(nIni);
  for (int i = 0; i < nIni; i++) {
    ExtractorNode ni;
    ni.UL = {(int)(hX * static_cast<float>(i)), 0};
    ni.UR = {(int)(hX * static_cast<float>(i + 1)), 0};
    ni.BL = {ni.UL.x, maxY - minY};
    ni.BR = {ni.UR.x, maxY - minY};
    lNodes.push_back(ni);
    vpIniNodes[i] = &lNodes.back();
  }
  for (size_t i = 0; i < vToDistributeKeys.size(); i++) {
    const OctKey &kp = vToDistributeKeys[i];
    vpIniNodes[(int)(kp.x / hX)]->vKeys.push_back(kp);
  }
  auto lit = lNodes.begin();
  while (lit != lNodes.end()) {
    if (lit->vKeys.size() == 1) {
      lit->bNoMore = true;
      lit++;
    } else if (lit->vKeys.empty())
      lit = lNodes.erase(lit);
    else
      lit++;
  }
  bool bFinish = false;
  std::vector<std::pair<int, ExtractorNode *>> vSizeAndPointerToNode;
  auto addChild = [&](ExtractorNode &n, int *nToExpand) {
    if (n.vKeys.size() > 0) {
      lNodes.push_front(n);
      if (n.vKeys.size() > 1) {
        if (nToExpand) (*nToExpand)++;
        vSizeAndPointerToNode.push_back(std::make_pair((int)n.vKeys.size(), &lNodes.front()));
        lNodes.front().lit = lNodes.begin();
      }
    }
  };
  while (!bFinish) {
    int prevSize = (int)lNodes.size();
    lit = lNodes.begin();
    int nToExpand = 0;
    vSizeAndPointerToNode.clear();
    while (lit != lNodes.end()) {
      if (lit->bNoMore) {
        lit++;
        continue;
      } else {
        ExtractorNode n1, n2, n3, n4;
        lit->DivideNode(n1, n2, n3, n4);
        addChild(n1, &nToExpand);
        addChild(n2, &nToExpand);
        addChild(n3, &nToExpand);
        addChild(n4, &nToExpand);
        lit = lNodes.erase(lit);
        continue;
      }
    }
    if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) {
      bFinish = true;
    } else if (((int)lNodes.size() + nToExpand * 3) > N) {
      while (!bFinish) {
        prevSize = (int)lNodes.size();
        std::vector<std::pair<int, ExtractorNode *>> vPrev = vSizeAndPointerToNode;
        vSizeAndPointerToNode.clear();
        std::sort(vPrev.begin(), vPrev.end(), compareNodes);
        for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
          ExtractorNode n1, n2, n3, n4;
          vPrev[j].second->DivideNode(n1, n2, n3, n4);
          addChild(n1, nullptr);
          addChild(n2, nullptr);
          addChild(n3, nullptr);
          addChild(n4, nullptr);
          lNodes.erase(vPrev[j].second->lit);
          if ((int)lNodes.size() >= N) break;
        }
        if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
      }
    }
  }
  for (auto it = lNodes.begin(); it != lNodes.end(); it++) {
    std::vector<OctKey> &vNodeKeys = it->vKeys;
    OctKey *pKP = &vNodeKeys[0];
    float maxResponse = pKP->response;
    for (size_t k = 1; k < vNodeKeys.size(); k++) {
      if (vNodeKeys[k].response > maxResponse) {
        pKP = &vNodeKeys[k];
        maxResponse = vNodeKeys[k].response;
      }
    }
    vResultKeys.push_back(*pKP);
  }
  return vResultKeys;
}

}  // namespace

// ================================================================ extractor object
struct OrExtractor {
  int nfeatures;
  double scaleFactor;  // ORBextractor.h:105 (double member initialised from float)
  int nlevels, iniThFAST, minThFAST;
  std::vector<int> mnFeaturesPerLevel, umax;
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
  uint16_t taps[7] = {18, 34, 49, 55, 49, 34, 18};
  // per-call state
  std::vector<std::vector<uint8_t>> pyrStore;  // bordered buffers ("temp")
  std::vector<Img> mvImagePyramid;             // ROI views
  std::vector<std::vector<uint8_t>> blurStore;
  std::vector<std::vector<FastKp>> candidates;
  std::vector<std::vector<OrKeyPoint>> levelKps;
};

extern "C" {

OrExtractor *or_create(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST) {
  // ORBextractor.cc:411-470
  if (_nlevels < 1 || _nfeatures < 0) return nullptr;
  OrExtractor *e = new OrExtractor();
  e->nfeatures = _nfeatures;
  e->scaleFactor = _scaleFactor;
  e->nlevels = _nlevels;
  e->iniThFAST = _iniThFAST;
  e->minThFAST = _minThFAST;
  const int nlevels = _nlevels;
  e->mvScaleFactor.resize(nlevels);
  e->mvLevelSigma2.resize(nlevels);
  e->mvScaleFactor[0] = 1.0f;
  e->mvLevelSigma2[0] = 1.0f;
  for (int i = 1; i < nlevels; i++) {
    e->mvScaleFactor[i] = (float)(e->mvScaleFactor[i - 1] * e->scaleFactor);
    e->mvLevelSigma2[i] = e->mvScaleFactor[i] * e->mvScaleFactor[i];
  }
  e->mvInvScaleFactor.resize(nlevels);
  e->mvInvLevelSigma2.resize(nlevels);
  for (int i = 0; i < nlevels; i++) {
    e->mvInvScaleFactor[i] = 1.0f / e->mvScaleFactor[i];
    e->mvInvLevelSigma2[i] = 1.0f / e->mvLevelSigma2[i];
  }
  e->mnFeaturesPerLevel.resize(nlevels);
  float factor = (float)(1.0f / e->scaleFactor);
  float nDesiredFeaturesPerScale =
      (float)(e->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels)));
  int sumFeatures = 0;
  for (int level = 0; level < nlevels - 1; level++) {
    e->mnFeaturesPerLevel[level] = cvRoundF(nDesiredFeaturesPerScale);
    sumFeatures += e->mnFeaturesPerLevel[level];
    nDesiredFeaturesPerScale *= factor;
  }
  e->mnFeaturesPerLevel[nlevels - 1] = std::max(e->nfeatures - sumFeatures, 0);

  e->umax.resize(HALF_PATCH_SIZE + 1);
  int v, v0, vmax = cvFloorD(HALF_PATCH_SIZE * sqrt(2.f) / 2 + 1);
  int vmin = cvCeilD(HALF_PATCH_SIZE * sqrt(2.f) / 2);
  const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
  for (v = 0; v <= vmax; ++v) e->umax[v] = cvRoundD(sqrt(hp2 - v * v));
  for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
    while (e->umax[v0] == e->umax[v0 + 1]) ++v0;
    e->umax[v] = v0;
    ++v0;
  }
  return e;
}

void or_destroy(OrExtractor *e) { delete e; }

void or_set_blur_taps(OrExtractor *e, const uint16_t taps[7]) { memcpy(e->taps, taps, sizeof(e->taps)); }

int or_get_tables(const OrExtractor *e, float *scale, float *invScale, float *sigma2, float *invSigma2,
                  int *featuresPerLevel, int *umax16) {
  for (int i = 0; i < e->nlevels; i++) {
    if (scale) scale[i] = e->mvScaleFactor[i];
    if (invScale) invScale[i] = e->mvInvScaleFactor[i];
    if (sigma2) sigma2[i] = e->mvLevelSigma2[i];
    if (invSigma2) invSigma2[i] = e->mvInvLevelSigma2[i];
    if (featuresPerLevel) featuresPerLevel[i] = e->mnFeaturesPerLevel[i];
  }
  if (umax16)
    for (int i = 0; i <= HALF_PATCH_SIZE; i++) umax16[i] = e->umax[i];
  return e->nlevels;
}

static void ComputePyramid(OrExtractor *e, const Img &image) {
  // ORBextractor.cc:1171-1195
  e->pyrStore.assign(e->nlevels, {});
  e->mvImagePyramid.assign(e->nlevels, Img());
  for (int level = 0; level < e->nlevels; ++level) {
    float scale = e->mvInvScaleFactor[level];
    int szw = cvRoundF((float)image.cols * scale), szh = cvRoundF((float)image.rows * scale);
    int ww = szw + EDGE_THRESHOLD * 2, wh = szh + EDGE_THRESHOLD * 2;
    e->pyrStore[level].assign((size_t)ww * wh, 0);
    uint8_t *temp = e->pyrStore[level].data();
    Img roi;
    roi.rows = szh;
    roi.cols = szw;
    roi.step = ww;
    roi.data = temp + (size_t)EDGE_THRESHOLD * ww + EDGE_THRESHOLD;
    e->mvImagePyramid[level] = roi;
    if (level != 0) {
      resizeLinearU8(e->mvImagePyramid[level - 1], roi);
      copyMakeBorder101(roi.data, szw, szh, ww, temp, ww, EDGE_THRESHOLD);
    } else {
      copyMakeBorder101(image.data, image.cols, image.rows, image.step, temp, ww, EDGE_THRESHOLD);
    }
  }
}

static void ComputeKeyPointsOctTree(OrExtractor *e) {
  // ORBextractor.cc:787-900
  const int nlevels = e->nlevels;
  e->candidates.assign(nlevels, {});
  e->levelKps.assign(nlevels, {});
  const float W = 35;
  for (int level = 0; level < nlevels; ++level) {
    const Img &im = e->mvImagePyramid[level];
    const int minBorderX = EDGE_THRESHOLD - 3;
    const int minBorderY = minBorderX;
    const int maxBorderX = im.cols - EDGE_THRESHOLD + 3;
    const int maxBorderY = im.rows - EDGE_THRESHOLD + 3;
    std::vector<FastKp> &vToDistributeKeys = e->candidates[level];
    const float width = (float)(maxBorderX - minBorderX);
    const float height = (float)(maxBorderY - minBorderY);
    const int nCols = (int)(width / W);
    const int nRows = (int)(height / W);
    if (nCols <= 0 || nRows <= 0) continue;  // reference divides by zero here; callers must not get this far
    const int wCell = (int)ceil(width / nCols);
    const int hCell = (int)ceil(height / nRows);
    std::vector<FastKp> vKeysCell;
    for (int i = 0; i < nRows; i++) {
      const float iniY = (float)(minBorderY + i * hCell);
      float maxY = iniY + hCell + 6;
      if (iniY >= maxBorderY - 3) continue;
      if (maxY > maxBorderY) maxY = (float)maxBorderY;
      for (int j = 0; j < nCols; j++) {
        const float iniX = (float)(minBorderX + j * wCell);
        float maxX = iniX + wCell + 6;
        if (iniX >= maxBorderX - 6) continue;
        if (maxX > maxBorderX) maxX = (float)maxBorderX;
        Img cell;  // rowRange(iniY,maxY).colRange(iniX,maxX): float -> int truncation in cv::Range
        cell.rows = (int)maxY - (int)iniY;
        cell.cols = (int)maxX - (int)iniX;
        cell.step = im.step;
        cell.data = im.data + (size_t)((int)iniY) * im.step + (int)iniX;
        fast9_16(cell, vKeysCell, e->iniThFAST, true);
        if (vKeysCell.empty()) fast9_16(cell, vKeysCell, e->minThFAST, true);
        for (auto &kp : vKeysCell) vToDistributeKeys.push_back({kp.x + j * wCell, kp.y + i * hCell, kp.score});
      }
    }
    std::vector<OctKey> in(vToDistributeKeys.size());
    for (size_t k = 0; k < in.size(); k++)
      in[k] = {(float)vToDistributeKeys[k].x, (float)vToDistributeKeys[k].y, (float)vToDistributeKeys[k].score,
               (int)k};
    std::vector<OctKey> out = DistributeOctTree(in, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                                e->mnFeaturesPerLevel[level]);
    const int scaledPatchSize = (int)(PATCH_SIZE * e->mvScaleFactor[level]);
    std::vector<OrKeyPoint> &keypoints = e->levelKps[level];
    keypoints.resize(out.size());
    for (size_t k = 0; k < out.size(); k++) {
      OrKeyPoint kp;
      kp.x = out[k].x + minBorderX;
      kp.y = out[k].y + minBorderY;
      kp.size = (float)scaledPatchSize;
      kp.angle = -1.f;
      kp.response = out[k].response;
      kp.octave = level;
      kp.class_id = -1;
      keypoints[k] = kp;
    }
  }
  for (int level = 0; level < nlevels; ++level) {  // computeOrientation, ORBextractor.cc:472-480
    const Img &im = e->mvImagePyramid[level];
    for (auto &kp : e->levelKps[level])
      kp.angle = IC_Angle(im.ptr(cvRoundF(kp.y)) + cvRoundF(kp.x), im.step, e->umax);
  }
}

int or_extract(OrExtractor *e, const uint8_t *gray, int rows, int cols, int stride, int lap0, int lap1,
               OrKeyPoint *kpsOut, uint8_t *descOut, int capacity, int *nOut) {
  // ORBextractor.cc:1083-1169
  if (nOut) *nOut = 0;
  if (!gray || rows <= 0 || cols <= 0) return -1;
  Img image;
  image.rows = rows;
  image.cols = cols;
  image.step = stride;
  image.data = const_cast<uint8_t *>(gray);
  ComputePyramid(e, image);
  ComputeKeyPointsOctTree(e);
  int nkeypoints = 0;
  for (int level = 0; level < e->nlevels; ++level) nkeypoints += (int)e->levelKps[level].size();
  if (nkeypoints > capacity) return -2;
  if (nOut) *nOut = nkeypoints;
  e->blurStore.assign(e->nlevels, {});
  int monoIndex = 0, stereoIndex = nkeypoints - 1;
  std::vector<uint8_t> desc;
  for (int level = 0; level < e->nlevels; ++level) {
    std::vector<OrKeyPoint> &keypoints = e->levelKps[level];
    int nkeypointsLevel = (int)keypoints.size();
    if (nkeypointsLevel == 0) continue;
    const Img &roi = e->mvImagePyramid[level];
    // workingMat = mvImagePyramid[level].clone(); GaussianBlur(7x7, 2, 2, REFLECT_101)
    std::vector<uint8_t> clone((size_t)roi.cols * roi.rows);
    for (int y = 0; y < roi.rows; y++) memcpy(&clone[(size_t)y * roi.cols], roi.ptr(y), (size_t)roi.cols);
    Img src;
    src.rows = roi.rows, src.cols = roi.cols, src.step = roi.cols, src.data = clone.data();
    e->blurStore[level].assign((size_t)roi.cols * roi.rows, 0);
    Img workingMat = src;
    workingMat.data = e->blurStore[level].data();
    gaussianBlur7(src, workingMat, e->taps);
    desc.assign((size_t)nkeypointsLevel * 32, 0);
    for (int i = 0; i < nkeypointsLevel; i++) {  // computeDescriptors, ORBextractor.cc:1074-1081
      const OrKeyPoint &kp = keypoints[i];
      computeOrbDescriptor(kp.angle, workingMat.ptr(cvRoundF(kp.y)) + cvRoundF(kp.x), workingMat.step,
                           kBitPattern31, &desc[(size_t)i * 32]);
    }
    float scale = e->mvScaleFactor[level];
    for (int i = 0; i < nkeypointsLevel; i++) {
      OrKeyPoint kp = keypoints[i];
      if (level != 0) {
        kp.x *= scale;
        kp.y *= scale;
      }
      if (kp.x >= lap0 && kp.x <= lap1) {
        kpsOut[stereoIndex] = kp;
        memcpy(descOut + (size_t)stereoIndex * 32, &desc[(size_t)i * 32], 32);
        stereoIndex--;
      } else {
        kpsOut[monoIndex] = kp;
        memcpy(descOut + (size_t)monoIndex * 32, &desc[(size_t)i * 32], 32);
        monoIndex++;
      }
    }
  }
  return monoIndex;
}

int or_level_size(const OrExtractor *e, int level, int *w, int *h) {
  if (level < 0 || level >= (int)e->mvImagePyramid.size()) return -1;
  *w = e->mvImagePyramid[level].cols;
  *h = e->mvImagePyramid[level].rows;
  return 0;
}

int or_get_pyramid_level(const OrExtractor *e, int level, uint8_t *dst, int dst_stride, int with_border) {
  if (level < 0 || level >= (int)e->mvImagePyramid.size()) return -1;
  const Img &roi = e->mvImagePyramid[level];
  if (with_border) {
    const int ww = roi.cols + 2 * EDGE_THRESHOLD, wh = roi.rows + 2 * EDGE_THRESHOLD;
    for (int y = 0; y < wh; y++) memcpy(dst + (size_t)y * dst_stride, &e->pyrStore[level][(size_t)y * ww], ww);
  } else {
    for (int y = 0; y < roi.rows; y++) memcpy(dst + (size_t)y * dst_stride, roi.ptr(y), roi.cols);
  }
  return 0;
}

int or_get_blurred_level(const OrExtractor *e, int level, uint8_t *dst, int dst_stride) {
  if (level < 0 || level >= (int)e->blurStore.size() || e->blurStore[level].empty()) return -1;
  const Img &roi = e->mvImagePyramid[level];
  for (int y = 0; y < roi.rows; y++)
    memcpy(dst + (size_t)y * dst_stride, &e->blurStore[level][(size_t)y * roi.cols], roi.cols);
  return 0;
}

int or_get_candidates(const OrExtractor *e, int level, int *x, int *y, int *response, int cap) {
  if (level < 0 || level >= (int)e->candidates.size()) return -1;
  const auto &c = e->candidates[level];
  for (size_t i = 0; i < c.size() && (int)i < cap; i++) {
    x[i] = c[i].x;
    y[i] = c[i].y;
    response[i] = c[i].score;
  }
  return (int)c.size();
}

int or_get_level_keypoints(const OrExtractor *e, int level, OrKeyPoint *kps, int cap) {
  if (level < 0 || level >= (int)e->levelKps.size()) return -1;
  const auto &c = e->levelKps[level];
  for (size_t i = 0; i < c.size() && (int)i < cap; i++) kps[i] = c[i];
  return (int)c.size();
}

// ---------------------------------------------------------------- stand-alone pieces
int or_cv_round_f(float v) { return cvRoundF(v); }
int or_cv_round_d(double v) { return cvRoundD(v); }
float or_fast_atan2(float y, float x) { return fastAtan2(y, x); }

void or_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw, int dh,
                         int dstride) {
  Img s, d;
  s.rows = sh, s.cols = sw, s.step = sstride, s.data = const_cast<uint8_t *>(src);
  d.rows = dh, d.cols = dw, d.step = dstride, d.data = dst;
  resizeLinearU8(s, d);
}

void or_copy_make_border101(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride, int b) {
  copyMakeBorder101(src, w, h, sstride, dst, dstride, b);
}

void or_gaussian_blur7_u8(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride,
                          const uint16_t taps[7]) {
  Img s, d;
  s.rows = h, s.cols = w, s.step = sstride, s.data = const_cast<uint8_t *>(src);
  d.rows = h, d.cols = w, d.step = dstride, d.data = dst;
  gaussianBlur7(s, d, taps);
}

int or_fast9_16(const uint8_t *img, int w, int h, int stride, int threshold, int nonmax, int *x, int *y,
                int *score, int cap) {
  Img s;
  s.rows = h, s.cols = w, s.step = stride, s.data = const_cast<uint8_t *>(img);
  std::vector<FastKp> kps;
  fast9_16(s, kps, threshold, nonmax != 0);
  for (size_t i = 0; i < kps.size() && (int)i < cap; i++) {
    x[i] = kps[i].x;
    y[i] = kps[i].y;
    score[i] = kps[i].score;
  }
  return (int)kps.size();
}

int or_distribute_octree(const int *x, const int *y, const int *response, int n, int minX, int maxX, int minY,
                         int maxY, int N, int *outIndex, int cap) {
  std::vector<OctKey> in(n);
  for (int i = 0; i < n; i++) in[i] = {(float)x[i], (float)y[i], (float)response[i], i};
  std::vector<OctKey> out = DistributeOctTree(in, minX, maxX, minY, maxY, N);
  for (size_t i = 0; i < out.size() && (int)i < cap; i++) outIndex[i] = out[i].srcIndex;
  return (int)out.size();
}

float or_ic_angle(const uint8_t *img, int stride, int cx, int cy) {
  static const int um[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
  std::vector<int> umax(um, um + 16);
  return IC_Angle(img + (size_t)cy * stride + cx, stride, umax);
}

void or_orb_descriptor(const uint8_t *blurred, int stride, int cx, int cy, float angleDeg, uint8_t desc[32]) {
  computeOrbDescriptor(angleDeg, blurred + (size_t)cy * stride + cx, stride, kBitPattern31, desc);
}

void or_cvt_gray_u8(const uint8_t *src, int rows, int cols, int sstride, int channels, int rgb_order, uint8_t *dst,
                    int dstride, const int coeffs[3], int shift) {
  // [OCV] color_rgb.simd.hpp RGB2Gray<uchar>: CV_DESCALE(b*cb + g*cg + r*cr, shift)
  const int ri = rgb_order ? 0 : 2, bi = rgb_order ? 2 : 0;
  for (int y = 0; y < rows; y++) {
    const uint8_t *s = src + (size_t)y * sstride;
    uint8_t *d = dst + (size_t)y * dstride;
    for (int x = 0; x < cols; x++, s += channels)
      d[x] = (uint8_t)((s[ri] * coeffs[0] + s[1] * coeffs[1] + s[bi] * coeffs[2] + (1 << (shift - 1))) >> shift);
  }
}

void or_stereo_matches(const OrExtractor *left, const OrExtractor *right, const OrKeyPoint *mvKeys,
                       const uint8_t *mDescriptors, int N, const OrKeyPoint *mvKeysRight,
                       const uint8_t *mDescriptorsRight, int Nr, float mb, float mbf, float *mvuRight,
                       float *mvDepth) {
  // Frame.cc:957-1127
  for (int i = 0; i < N; i++) mvuRight[i] = -1.0f, mvDepth[i] = -1.0f;
  const int TH_HIGH = 100, TH_LOW = 50;
  const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
  const std::vector<float> &mvScaleFactors = left->mvScaleFactor, &mvInvScaleFactors = left->mvInvScaleFactor;
  const int nRows = left->mvImagePyramid[0].rows;
  std::vector<std::vector<size_t>> vRowIndices(nRows, std::vector<size_t>());
  for (int iR = 0; iR < Nr; iR++) {
    const OrKeyPoint &kp = mvKeysRight[iR];
    const float &kpY = kp.y;
    const float r = 2.0f * mvScaleFactors[mvKeysRight[iR].octave];
    const int maxr = (int)ceil(kpY + r);
    const int minr = (int)floor(kpY - r);
    for (int yi = minr; yi <= maxr; yi++)
      if (yi >= 0 && yi < nRows) vRowIndices[yi].push_back(iR);  // (the reference indexes unchecked)
  }
  const float minZ = mb;
  const float minD = 0;
  const float maxD = mbf / minZ;
  std::vector<std::pair<int, int>> vDistIdx;
  auto descDist = [](const uint8_t *a, const uint8_t *b) {
    int d = 0;
    for (int k = 0; k < 32; k++) d += __builtin_popcount((unsigned)(a[k] ^ b[k]));
    return d;
  };
  for (int iL = 0; iL < N; iL++) {
    const OrKeyPoint &kpL = mvKeys[iL];
    const int &levelL = kpL.octave;
    const float &vL = kpL.y;
    const float &uL = kpL.x;
    if ((int)vL < 0 || (int)vL >= nRows) continue;
    const std::vector<size_t> &vCandidates = vRowIndices[(size_t)vL];
    if (vCandidates.empty()) continue;
    const float minU = uL - maxD;
    const float maxU = uL - minD;
    if (maxU < 0) continue;
    int bestDist = TH_HIGH;
    size_t bestIdxR = 0;
    const uint8_t *dL = mDescriptors + (size_t)iL * 32;
    for (size_t iC = 0; iC < vCandidates.size(); iC++) {
      const size_t iR = vCandidates[iC];
      const OrKeyPoint &kpR = mvKeysRight[iR];
      if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
      const float &uR = kpR.x;
      if (uR >= minU && uR <= maxU) {
        const int dist = descDist(dL, mDescriptorsRight + iR * 32);
        if (dist < bestDist) {
          bestDist = dist;
          bestIdxR = iR;
        }
      }
    }
    if (bestDist < thOrbDist) {
      const float uR0 = mvKeysRight[bestIdxR].x;
      const float scaleFactor = mvInvScaleFactors[kpL.octave];
      const float scaleduL = roundf(kpL.x * scaleFactor);
      const float scaledvL = roundf(kpL.y * scaleFactor);
      const float scaleduR0 = roundf(uR0 * scaleFactor);
      const int w = 5;
      const Img &pl = left->mvImagePyramid[kpL.octave], &pr = right->mvImagePyramid[kpL.octave];
      // IL = pl.rowRange(scaledvL - w, scaledvL + w + 1).colRange(scaleduL - w, scaleduL + w + 1)
      const int ily = (int)(scaledvL - w), ilx = (int)(scaleduL - w);
      int bestDistS = INT_MAX;
      int bestincR = 0;
      const int L = 5;
      std::vector<float> vDists(2 * L + 1);
      const float iniu = scaleduR0 + L - w;
      const float endu = scaleduR0 + L + w + 1;
      if (iniu < 0 || endu >= pr.cols) continue;
      for (int incR = -L; incR <= +L; incR++) {
        const int irx = (int)(scaleduR0 + incR - w);
        long sum = 0;  // cv::norm(IL, IR, cv::NORM_L1)
        for (int yy = 0; yy < 2 * w + 1; yy++)
          for (int xx = 0; xx < 2 * w + 1; xx++)
            sum += std::abs((int)pl.ptr(ily + yy)[ilx + xx] - (int)pr.ptr(ily + yy)[irx + xx]);
        float dist = (float)(double)sum;
        if (dist < bestDistS) {
          bestDistS = (int)dist;
          bestincR = incR;
        }
        vDists[L + incR] = dist;
      }
      if (bestincR == -L || bestincR == L) continue;
      const float dist1 = vDists[L + bestincR - 1];
      const float dist2 = vDists[L + bestincR];
      const float dist3 = vDists[L + bestincR + 1];
      const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
      if (deltaR < -1 || deltaR > 1) continue;
      float bestuR = mvScaleFactors[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
      float disparity = (uL - bestuR);
      if (disparity >= minD && disparity < maxD) {
        if (disparity <= 0) {
          disparity = 0.01;
          bestuR = uL - 0.01;
        }
        mvDepth[iL] = mbf / disparity;
        mvuRight[iL] = bestuR;
        vDistIdx.push_back(std::pair<int, int>(bestDistS, iL));
      }
    }
  }
  if (vDistIdx.empty()) return;  // (the reference reads vDistIdx[0] of an empty vector here)
  std::sort(vDistIdx.begin(), vDistIdx.end());
  const float median = vDistIdx[vDistIdx.size() / 2].first;
  const float thDist = 1.5f * 1.4f * median;
  for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
    if (vDistIdx[i].first < thDist)
      break;
    else {
      mvuRight[vDistIdx[i].second] = -1;
      mvDepth[vDistIdx[i].second] = -1;
    }
  }
}

double or_bench_throughput(const uint8_t *frames, int nframes, int rows, int cols, int nfeatures, float scaleFactor,
                           int nlevels, int iniThFAST, int minThFAST, int nthreads, double seconds, int do_match,
                           long *frames_done) {
  if (nthreads < 1) nthreads = 1;
  std::atomic<long> total(0);
  const auto t0 = std::chrono::steady_clock::now();
  auto elapsed = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  auto worker = [&](int tid) {
    OrExtractor *e = or_create(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST);
    const int cap = nfeatures + 3 * nlevels + 64;
    std::vector<OrKeyPoint> kps(cap);
    std::vector<uint8_t> desc((size_t)cap * 32), prev((size_t)cap * 32);
    std::vector<int> best(cap), second(cap), arg(cap);
    int nprev = 0;
    long done = 0;
    for (int i = tid; elapsed() < seconds; i += nthreads) {
      int n = 0;
      or_extract(e, frames + (size_t)(i % nframes) * rows * cols, rows, cols, cols, 0, 0, kps.data(), desc.data(), cap, &n);
      if (do_match && nprev > 0 && n > 0) or_block_best2(desc.data(), n, prev.data(), nprev, best.data(), second.data(), arg.data());
      prev.swap(desc);
      nprev = n;
      done++;
    }
    total += done;
    or_destroy(e);
  };
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; t++) th.emplace_back(worker, t);
  for (auto &t : th) t.join();
  const double dt = elapsed();
  if (frames_done) *frames_done = total.load();
  return (double)total.load() / dt;
}

/* Whole-batch checker input: operator() on every frame of a batch, `nthreads` host threads (one extractor each, frames
 * dealt round-robin), outputs in fixed-capacity rows.  counts[f] = {n, monoIndex}; a frame whose n exceeds `capacity`
 * gets counts[f] = {-2, -2}. */
void or_extract_batch_mt(const uint8_t *frames, int nframes, int rows, int cols, int nfeatures, float scaleFactor,
                         int nlevels, int iniThFAST, int minThFAST, int lap0, int lap1, int nthreads, int capacity,
                         int *counts, OrKeyPoint *kps, uint8_t *desc) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > nframes) nthreads = nframes > 0 ? nframes : 1;
  auto worker = [&](int tid) {
    OrExtractor *e = or_create(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST);
    for (int f = tid; f < nframes; f += nthreads) {
      int n = 0;
      const int mono = or_extract(e, frames + (size_t)f * rows * cols, rows, cols, cols, lap0, lap1,
                                  kps + (size_t)f * capacity, desc + (size_t)f * capacity * 32, capacity, &n);
      counts[2 * f] = mono == -2 ? -2 : n;
      counts[2 * f + 1] = mono;
    }
    or_destroy(e);
  };
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; t++) th.emplace_back(worker, t);
  for (auto &t : th) t.join();
}

/* Row f of a batch of brute-force searches: descriptors a[f] (na[f] of them) against b[f] (nb[f]); rows of `capacity`
 * descriptors, `a_stride` / `b_stride` bytes between consecutive frames' descriptor blocks. */
void or_block_best2_batch_mt(const uint8_t *a, size_t a_stride, const int *na, const uint8_t *b, size_t b_stride,
                             const int *nb, int nblocks, int capacity, int nthreads, int *best, int *second, int *argbest) {
  if (nthreads < 1) nthreads = 1;
  auto worker = [&](int tid) {
    for (int f = tid; f < nblocks; f += nthreads)
      or_block_best2(a + (size_t)f * a_stride, na[f], b + (size_t)f * b_stride, nb[f], best + (size_t)f * capacity,
                     second + (size_t)f * capacity, argbest + (size_t)f * capacity);
  };
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; t++) th.emplace_back(worker, t);
  for (auto &t : th) t.join();
}

}  // extern "C"
