/*
 * routines_oracle.cpp -- CPU ORACLE (test infrastructure, NOT product code).  PARITY UNPINNED.
 *
 * Routine-level restatement of the projection / Sim3 / Fuse searches of orb_slam3/src/ORBmatcher.cc on a flattened
 * Frame / KeyFrame ("OrFrame": the members the routines read -- mvKeysUn or mvKeys || mvKeysRight, mDescriptors,
 * mvuRight, Nleft, mGrid / mGridRight with their bounds).  Each function starts where the reference has finished the
 * camera geometry of a map point (projection, distance / viewing-angle tests, PredictScale -- Eigen / Sophus code
 * that stays with the caller) and follows the reference's loop from there LITERALLY: GetFeaturesInArea per map point,
 * the candidate loop with its skips in the reference's order, strict / non-strict comparisons, thresholds, the
 * `continue`s that leave a whole map point, the rotation histogram.  MapPoint* slots are flattened like in
 * orb_oracle.h: trainBlocked[i] = (mvpMapPoints[i] && Observations() > 0), trainMatch[i] = index of the query whose
 * map point was assigned to feature i.
 */
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <set>
#include <vector>

#include "orb_oracle.h"

namespace {
const int TH_HIGH = 100;      // ORBmatcher.cc:34
const int TH_LOW = 50;        // ORBmatcher.cc:35
const int HISTO_LENGTH = 30;  // ORBmatcher.cc:36
const int FRAME_GRID_ROWS = 48, FRAME_GRID_COLS = 64;  // Frame.h:49-50

using std::max;
using std::min;
using std::vector;

// ORBmatcher.cc:2047-2063
inline int DescriptorDistance(const uint8_t *a, const uint8_t *b) {
  int32_t pa[8], pb[8];
  memcpy(pa, a, 32);
  memcpy(pb, b, 32);
  int dist = 0;
  for (int i = 0; i < 8; i++) {
    unsigned int v = pa[i] ^ pb[i];
    v = v - ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
  }
  return dist;
}

// ORBmatcher.cc:2002-2043
void ComputeThreeMaxima(vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int s = (int)histo[i].size();
    if (s > max1) {
      max3 = max2;
      max2 = max1;
      max1 = s;
      ind3 = ind2;
      ind2 = ind1;
      ind1 = i;
    } else if (s > max2) {
      max3 = max2;
      max2 = s;
      ind3 = ind2;
      ind2 = i;
    } else if (s > max3) {
      max3 = s;
      ind3 = i;
    }
  }
  if (max2 < 0.1f * (float)max1) {
    ind2 = -1;
    ind3 = -1;
  } else if (max3 < 0.1f * (float)max1) {
    ind3 = -1;
  }
}

// ORBmatcher.cc:218-224
float RadiusByViewingCos(const float &viewCos) {
  if (viewCos > 0.998)
    return 2.5;
  else
    return 4.0;
}
}  // namespace

struct OrFrame {
  int N = 0, Nleft = -1;
  float mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0, mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;
  vector<OrKeyPoint> keys;  // mvKeysUn (Nleft == -1) or mvKeys followed by mvKeysRight
  vector<uint8_t> desc;     // mDescriptors (vconcat of left and right for Nleft != -1, Frame.cc:296)
  vector<float> mvuRight;
  vector<size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS], mGridRight[FRAME_GRID_COLS][FRAME_GRID_ROWS];

  const OrKeyPoint &key(size_t i, bool bRight) const { return (Nleft == -1 || !bRight) ? keys[i] : keys[i + Nleft]; }

  // Frame.cc:870-880
  bool PosInGrid(const OrKeyPoint &kp, int &posX, int &posY) const {
    posX = (int)round((kp.x - mnMinX) * mfGridElementWidthInv);
    posY = (int)round((kp.y - mnMinY) * mfGridElementHeightInv);
    if (posX < 0 || posX >= FRAME_GRID_COLS || posY < 0 || posY >= FRAME_GRID_ROWS) return false;
    return true;
  }

  // Frame.cc:521-553
  void AssignFeaturesToGrid() {
    for (int i = 0; i < N; i++) {
      const OrKeyPoint &kp = keys[i];  // mvKeysUn[i] / mvKeys[i] / mvKeysRight[i - Nleft]
      int nGridPosX, nGridPosY;
      if (PosInGrid(kp, nGridPosX, nGridPosY)) {
        if (Nleft == -1 || i < Nleft)
          mGrid[nGridPosX][nGridPosY].push_back(i);
        else
          mGridRight[nGridPosX][nGridPosY].push_back(i - Nleft);
      }
    }
  }

  // Frame::GetFeaturesInArea (Frame.cc:802-868)
  vector<size_t> GetFeaturesInArea(const float &x, const float &y, const float &r, const int minLevel = -1,
                                   const int maxLevel = -1, const bool bRight = false) const {
    vector<size_t> vIndices;
    vIndices.reserve(N);
    float factorX = r;
    float factorY = r;
    const int nMinCellX = max(0, (int)floor((x - mnMinX - factorX) * mfGridElementWidthInv));
    if (nMinCellX >= FRAME_GRID_COLS) return vIndices;
    const int nMaxCellX = min((int)FRAME_GRID_COLS - 1, (int)ceil((x - mnMinX + factorX) * mfGridElementWidthInv));
    if (nMaxCellX < 0) return vIndices;
    const int nMinCellY = max(0, (int)floor((y - mnMinY - factorY) * mfGridElementHeightInv));
    if (nMinCellY >= FRAME_GRID_ROWS) return vIndices;
    const int nMaxCellY = min((int)FRAME_GRID_ROWS - 1, (int)ceil((y - mnMinY + factorY) * mfGridElementHeightInv));
    if (nMaxCellY < 0) return vIndices;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
      for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
        const vector<size_t> vCell = (!bRight) ? mGrid[ix][iy] : mGridRight[ix][iy];
        if (vCell.empty()) continue;
        for (size_t j = 0, jend = vCell.size(); j < jend; j++) {
          const OrKeyPoint &kpUn = key(vCell[j], bRight);
          if (bCheckLevels) {
            if (kpUn.octave < minLevel) continue;
            if (maxLevel >= 0)
              if (kpUn.octave > maxLevel) continue;
          }
          const float distx = kpUn.x - x;
          const float disty = kpUn.y - y;
          if (fabs(distx) < factorX && fabs(disty) < factorY) vIndices.push_back(vCell[j]);
        }
      }
    }
    return vIndices;
  }

  // KeyFrame::GetFeaturesInArea (KeyFrame.cc:834-874): no level filter
  vector<size_t> KFGetFeaturesInArea(const float &x, const float &y, const float &r, const bool bRight = false) const {
    vector<size_t> vIndices;
    vIndices.reserve(N);
    float factorX = r;
    float factorY = r;
    const int nMinCellX = max(0, (int)floor((x - mnMinX - factorX) * mfGridElementWidthInv));
    if (nMinCellX >= FRAME_GRID_COLS) return vIndices;
    const int nMaxCellX = min((int)FRAME_GRID_COLS - 1, (int)ceil((x - mnMinX + factorX) * mfGridElementWidthInv));
    if (nMaxCellX < 0) return vIndices;
    const int nMinCellY = max(0, (int)floor((y - mnMinY - factorY) * mfGridElementHeightInv));
    if (nMinCellY >= FRAME_GRID_ROWS) return vIndices;
    const int nMaxCellY = min((int)FRAME_GRID_ROWS - 1, (int)ceil((y - mnMinY + factorY) * mfGridElementHeightInv));
    if (nMaxCellY < 0) return vIndices;
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
      for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
        const vector<size_t> vCell = (!bRight) ? mGrid[ix][iy] : mGridRight[ix][iy];
        for (size_t j = 0, jend = vCell.size(); j < jend; j++) {
          const OrKeyPoint &kpUn = key(vCell[j], bRight);
          const float distx = kpUn.x - x;
          const float disty = kpUn.y - y;
          if (fabs(distx) < r && fabs(disty) < r) vIndices.push_back(vCell[j]);
        }
      }
    }
    return vIndices;
  }

  const uint8_t *row(size_t i) const { return desc.data() + i * 32; }
};

extern "C" {

OrFrame *or_frame_create(const OrKeyPoint *keys, const uint8_t *desc, const float *uRight, int N, int Nleft, float minX,
                         float minY, float maxX, float maxY) {
  OrFrame *f = new OrFrame();
  f->N = N, f->Nleft = Nleft;
  f->mnMinX = minX, f->mnMinY = minY, f->mnMaxX = maxX, f->mnMaxY = maxY;
  // Frame.cc:378-379: mfGridElementWidthInv = static_cast<float>(FRAME_GRID_COLS) / static_cast<float>(mnMaxX - mnMinX)
  f->mfGridElementWidthInv = static_cast<float>(FRAME_GRID_COLS) / static_cast<float>(maxX - minX);
  f->mfGridElementHeightInv = static_cast<float>(FRAME_GRID_ROWS) / static_cast<float>(maxY - minY);
  f->keys.assign(keys, keys + N);
  f->desc.assign(desc, desc + (size_t)N * 32);
  if (uRight)
    f->mvuRight.assign(uRight, uRight + N);
  else
    f->mvuRight.assign(N, -1.0f);  // Frame.cc:451-452 (monocular): mvuRight = vector<float>(N, -1)
  f->AssignFeaturesToGrid();
  return f;
}
void or_frame_destroy(OrFrame *f) { delete f; }

// grid read-back as CSR over cells ix * 48 + iy (for the device grid build test)
int or_frame_grid(const OrFrame *f, int right, int *cell_start, int *entries) {
  int run = 0;
  for (int ix = 0; ix < FRAME_GRID_COLS; ix++)
    for (int iy = 0; iy < FRAME_GRID_ROWS; iy++) {
      cell_start[ix * FRAME_GRID_ROWS + iy] = run;
      const vector<size_t> &c = right ? f->mGridRight[ix][iy] : f->mGrid[ix][iy];
      for (size_t j = 0; j < c.size(); j++) entries[run++] = (int)c[j];
    }
  cell_start[FRAME_GRID_COLS * FRAME_GRID_ROWS] = run;
  return run;
}

int or_frame_features_in_area(const OrFrame *f, float x, float y, float r, int minLevel, int maxLevel, int bRight,
                              int kfForm, int *out, int cap) {
  const vector<size_t> v = kfForm ? f->KFGetFeaturesInArea(x, y, r, bRight != 0)
                                  : f->GetFeaturesInArea(x, y, r, minLevel, maxLevel, bRight != 0);
  for (size_t i = 0; i < v.size() && (int)i < cap; i++) out[i] = (int)v[i];
  return (int)v.size();
}

/* int ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, const float th, ...)
 * (ORBmatcher.cc:42-216).  Per map point (after the :50-57 tests): the mbTrackInView / mTrackProjX... members and
 * their ...R twins; mpObserved = Observations() > 0.  mvScaleFactors = F.mvScaleFactors. */
int or_frame_search_by_projection(const OrFrame *Fp, int nMP, const uint8_t *mpDesc, const uint8_t *mpObserved,
                                  const uint8_t *mbTrackInView, const float *mTrackProjX, const float *mTrackProjY,
                                  const float *mTrackProjXR, const int *mnTrackScaleLevel, const float *mTrackViewCos,
                                  const uint8_t *mbTrackInViewR, const float *mTrackProjXR_r,
                                  const float *mTrackProjYR_r, const int *mnTrackScaleLevelR,
                                  const float *mTrackViewCosR, float th, float mfNNratio, const float *mvScaleFactors,
                                  const int *mvLeftToRightMatch, const int *mvRightToLeftMatch, uint8_t *trainBlocked,
                                  int *trainMatch) {
  const OrFrame &F = *Fp;
  int nmatches = 0, left = 0, right = 0;
  const bool bFactor = th != 1.0;
  for (int iMP = 0; iMP < nMP; iMP++) {
    const bool inViewR = mbTrackInViewR && mbTrackInViewR[iMP];
    if (!mbTrackInView[iMP] && !inViewR) continue;
    const uint8_t *MPdescriptor = mpDesc + (size_t)iMP * 32;
    if (mbTrackInView[iMP]) {
      const int &nPredictedLevel = mnTrackScaleLevel[iMP];
      float r = RadiusByViewingCos(mTrackViewCos[iMP]);
      if (bFactor) r *= th;
      const vector<size_t> vIndices = F.GetFeaturesInArea(mTrackProjX[iMP], mTrackProjY[iMP],
                                                          r * mvScaleFactors[nPredictedLevel], nPredictedLevel - 1,
                                                          nPredictedLevel);
      if (!vIndices.empty()) {
        int bestDist = 256;
        int bestLevel = -1;
        int bestDist2 = 256;
        int bestLevel2 = -1;
        int bestIdx = -1;
        for (vector<size_t>::const_iterator vit = vIndices.begin(), vend = vIndices.end(); vit != vend; vit++) {
          const size_t idx = *vit;
          if (trainBlocked[idx]) continue;  // F.mvpMapPoints[idx] && Observations() > 0
          if (F.Nleft == -1 && F.mvuRight[idx] > 0) {
            const float er = fabs(mTrackProjXR[iMP] - F.mvuRight[idx]);
            if (er > r * mvScaleFactors[nPredictedLevel]) continue;
          }
          const uint8_t *d = F.row(idx);
          const int dist = DescriptorDistance(MPdescriptor, d);
          if (dist < bestDist) {
            bestDist2 = bestDist;
            bestDist = dist;
            bestLevel2 = bestLevel;
            bestLevel = F.keys[idx].octave;  // mvKeysUn / mvKeys (idx < Nleft on the left grid)
            bestIdx = (int)idx;
          } else if (dist < bestDist2) {
            bestLevel2 = F.keys[idx].octave;
            bestDist2 = dist;
          }
        }
        if (bestDist <= TH_HIGH) {
          if (bestLevel == bestLevel2 && bestDist > mfNNratio * bestDist2) continue;
          if (bestLevel != bestLevel2 || bestDist <= mfNNratio * bestDist2) {
            trainMatch[bestIdx] = iMP;  // F.mvpMapPoints[bestIdx] = pMP
            trainBlocked[bestIdx] = mpObserved[iMP];
            if (F.Nleft != -1 && mvLeftToRightMatch[bestIdx] != -1) {
              trainMatch[mvLeftToRightMatch[bestIdx] + F.Nleft] = iMP;
              trainBlocked[mvLeftToRightMatch[bestIdx] + F.Nleft] = mpObserved[iMP];
              nmatches++;
              right++;
            }
            nmatches++;
            left++;
          }
        }
      }
    }
    if (F.Nleft != -1 && inViewR) {
      const int &nPredictedLevel = mnTrackScaleLevelR[iMP];
      if (nPredictedLevel != -1) {
        float r = RadiusByViewingCos(mTrackViewCosR[iMP]);
        const vector<size_t> vIndices = F.GetFeaturesInArea(mTrackProjXR_r[iMP], mTrackProjYR_r[iMP],
                                                            r * mvScaleFactors[nPredictedLevel], nPredictedLevel - 1,
                                                            nPredictedLevel, true);
        if (vIndices.empty()) continue;
        int bestDist = 256;
        int bestLevel = -1;
        int bestDist2 = 256;
        int bestLevel2 = -1;
        int bestIdx = -1;
        for (vector<size_t>::const_iterator vit = vIndices.begin(), vend = vIndices.end(); vit != vend; vit++) {
          const size_t idx = *vit;
          if (trainBlocked[idx + F.Nleft]) continue;
          const uint8_t *d = F.row(idx + F.Nleft);
          const int dist = DescriptorDistance(MPdescriptor, d);
          if (dist < bestDist) {
            bestDist2 = bestDist;
            bestDist = dist;
            bestLevel2 = bestLevel;
            bestLevel = F.keys[idx + F.Nleft].octave;  // mvKeysRight[idx]
            bestIdx = (int)idx;
          } else if (dist < bestDist2) {
            bestLevel2 = F.keys[idx + F.Nleft].octave;
            bestDist2 = dist;
          }
        }
        if (bestDist <= TH_HIGH) {
          if (bestLevel == bestLevel2 && bestDist > mfNNratio * bestDist2) continue;
          if (F.Nleft != -1 && mvRightToLeftMatch[bestIdx] != -1) {
            trainMatch[mvRightToLeftMatch[bestIdx]] = iMP;
            trainBlocked[mvRightToLeftMatch[bestIdx]] = mpObserved[iMP];
            nmatches++;
            left++;
          }
          trainMatch[bestIdx + F.Nleft] = iMP;
          trainBlocked[bestIdx + F.Nleft] = mpObserved[iMP];
          nmatches++;
          right++;
        }
      }
    }
  }
  (void)left, (void)right;
  return nmatches;
}

/* int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)
 * (ORBmatcher.cc:1667-1878).  Per LastFrame map point that passed :1688-1711: uv, ur = uv(0) - mbf * invzc, the
 * right-camera projection uvR (Nleft != -1), nLastOctave, kpLF.angle.  bForward / bBackward as computed at :1683-1684. */
int or_frame_search_by_projection_last(const OrFrame *Cur, int nQ, const uint8_t *mpDesc, const uint8_t *mpObserved,
                                       const float *u, const float *v, const float *ur, const float *uR,
                                       const float *vR, const int *nLastOctaveArr, const float *kpLFangle, float th,
                                       int bForward, int bBackward, const float *mvScaleFactors, int mbCheckOrientation,
                                       uint8_t *trainBlocked, int *trainMatch) {
  const OrFrame &CurrentFrame = *Cur;
  int nmatches = 0;
  vector<int> rotHist[HISTO_LENGTH];
  for (int i = 0; i < HISTO_LENGTH; i++) rotHist[i].reserve(500);
  const float factor = 1.0f / HISTO_LENGTH;
  for (int i = 0; i < nQ; i++) {
    const uint8_t *dMP = mpDesc + (size_t)i * 32;
    int nLastOctave = nLastOctaveArr[i];
    float radius = th * mvScaleFactors[nLastOctave];
    vector<size_t> vIndices2;
    if (bForward)
      vIndices2 = CurrentFrame.GetFeaturesInArea(u[i], v[i], radius, nLastOctave);
    else if (bBackward)
      vIndices2 = CurrentFrame.GetFeaturesInArea(u[i], v[i], radius, 0, nLastOctave);
    else
      vIndices2 = CurrentFrame.GetFeaturesInArea(u[i], v[i], radius, nLastOctave - 1, nLastOctave + 1);
    if (vIndices2.empty()) continue;
    int bestDist = 256;
    int bestIdx2 = -1;
    for (vector<size_t>::const_iterator vit = vIndices2.begin(), vend = vIndices2.end(); vit != vend; vit++) {
      const size_t i2 = *vit;
      if (trainBlocked[i2]) continue;
      if (CurrentFrame.Nleft == -1 && CurrentFrame.mvuRight[i2] > 0) {
        const float er = fabs(ur[i] - CurrentFrame.mvuRight[i2]);
        if (er > radius) continue;
      }
      const int dist = DescriptorDistance(dMP, CurrentFrame.row(i2));
      if (dist < bestDist) {
        bestDist = dist;
        bestIdx2 = (int)i2;
      }
    }
    if (bestDist <= TH_HIGH) {
      trainMatch[bestIdx2] = i;
      trainBlocked[bestIdx2] = mpObserved[i];
      nmatches++;
      if (mbCheckOrientation) {
        float rot = kpLFangle[i] - CurrentFrame.keys[bestIdx2].angle;
        if (rot < 0.0) rot += 360.0f;
        int bin = round(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        rotHist[bin].push_back(bestIdx2);
      }
    }
    if (CurrentFrame.Nleft != -1) {
      vector<size_t> vIndicesR;
      if (bForward)
        vIndicesR = CurrentFrame.GetFeaturesInArea(uR[i], vR[i], radius, nLastOctave, -1, true);
      else if (bBackward)
        vIndicesR = CurrentFrame.GetFeaturesInArea(uR[i], vR[i], radius, 0, nLastOctave, true);
      else
        vIndicesR = CurrentFrame.GetFeaturesInArea(uR[i], vR[i], radius, nLastOctave - 1, nLastOctave + 1, true);
      int bestDistR = 256;
      int bestIdxR = -1;
      for (vector<size_t>::const_iterator vit = vIndicesR.begin(), vend = vIndicesR.end(); vit != vend; vit++) {
        const size_t i2 = *vit;
        if (trainBlocked[i2 + CurrentFrame.Nleft]) continue;
        const int dist = DescriptorDistance(dMP, CurrentFrame.row(i2 + CurrentFrame.Nleft));
        if (dist < bestDistR) {
          bestDistR = dist;
          bestIdxR = (int)i2;
        }
      }
      if (bestDistR <= TH_HIGH) {
        trainMatch[bestIdxR + CurrentFrame.Nleft] = i;
        trainBlocked[bestIdxR + CurrentFrame.Nleft] = mpObserved[i];
        nmatches++;
        if (mbCheckOrientation) {
          float rot = kpLFangle[i] - CurrentFrame.keys[bestIdxR + CurrentFrame.Nleft].angle;  // mvKeysRight[bestIdx2]
          if (rot < 0.0) rot += 360.0f;
          int bin = round(rot * factor);
          if (bin == HISTO_LENGTH) bin = 0;
          rotHist[bin].push_back(bestIdxR + CurrentFrame.Nleft);
        }
      }
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i != ind1 && i != ind2 && i != ind3) {
        for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
          trainMatch[rotHist[i][j]] = -1;  // CurrentFrame.mvpMapPoints[...] = NULL
          trainBlocked[rotHist[i][j]] = 0;
          nmatches--;
        }
      }
    }
  }
  return nmatches;
}

/* int ORBmatcher::SearchByProjection(KeyFrame *pKF, Sim3f &Scw, vpPoints, vpMatched, int th, float ratioHamming)
 * (ORBmatcher.cc:430-528; :530-641 is the same loop plus vpMatchedKF[bestIdx] = pKFi).  Per point that passed
 * :446-480: uv, radius = th * mvScaleFactors[nPredictedLevel], nPredictedLevel.  vpMatched[i] != NULL <=> matched[i] != -1. */
int or_kf_search_by_projection_sim3(const OrFrame *pKF, int nQ, const uint8_t *mpDesc, const float *u, const float *v,
                                    const float *radiusArr, const int *nPredictedLevelArr, float ratioHamming,
                                    int *matched) {
  int nmatches = 0;
  for (int iMP = 0; iMP < nQ; iMP++) {
    const int nPredictedLevel = nPredictedLevelArr[iMP];
    const float radius = radiusArr[iMP];
    const vector<size_t> vIndices = pKF->KFGetFeaturesInArea(u[iMP], v[iMP], radius);
    if (vIndices.empty()) continue;
    const uint8_t *dMP = mpDesc + (size_t)iMP * 32;
    int bestDist = 256;
    int bestIdx = -1;
    for (vector<size_t>::const_iterator vit = vIndices.begin(), vend = vIndices.end(); vit != vend; vit++) {
      const size_t idx = *vit;
      if (matched[idx] != -1) continue;
      const int &kpLevel = pKF->keys[idx].octave;
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      const int dist = DescriptorDistance(dMP, pKF->row(idx));
      if (dist < bestDist) {
        bestDist = dist;
        bestIdx = (int)idx;
      }
    }
    if (bestDist <= TH_LOW * ratioHamming) {
      matched[bestIdx] = iMP;
      nmatches++;
    }
  }
  return nmatches;
}

/* int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound,
 * const float th, const int ORBdist) (ORBmatcher.cc:1880-2000).  Per map point that passed :1901-1930. */
int or_frame_search_by_projection_kf(const OrFrame *Cur, int nQ, const uint8_t *mpDesc, const float *u, const float *v,
                                     const float *radiusArr, const int *nPredictedLevelArr, const float *kfAngle,
                                     int ORBdist, int mbCheckOrientation, uint8_t *occupied, int *trainMatch) {
  const OrFrame &CurrentFrame = *Cur;
  int nmatches = 0;
  vector<int> rotHist[HISTO_LENGTH];
  for (int i = 0; i < HISTO_LENGTH; i++) rotHist[i].reserve(500);
  const float factor = 1.0f / HISTO_LENGTH;
  for (int i = 0; i < nQ; i++) {
    const int nPredictedLevel = nPredictedLevelArr[i];
    const float radius = radiusArr[i];
    const vector<size_t> vIndices2 =
        CurrentFrame.GetFeaturesInArea(u[i], v[i], radius, nPredictedLevel - 1, nPredictedLevel + 1);
    if (vIndices2.empty()) continue;
    const uint8_t *dMP = mpDesc + (size_t)i * 32;
    int bestDist = 256;
    int bestIdx2 = -1;
    for (vector<size_t>::const_iterator vit = vIndices2.begin(); vit != vIndices2.end(); vit++) {
      const size_t i2 = *vit;
      if (occupied[i2]) continue;  // CurrentFrame.mvpMapPoints[i2]
      const int dist = DescriptorDistance(dMP, CurrentFrame.row(i2));
      if (dist < bestDist) {
        bestDist = dist;
        bestIdx2 = (int)i2;
      }
    }
    if (bestDist <= ORBdist) {
      trainMatch[bestIdx2] = i;
      occupied[bestIdx2] = 1;
      nmatches++;
      if (mbCheckOrientation) {
        float rot = kfAngle[i] - CurrentFrame.keys[bestIdx2].angle;
        if (rot < 0.0) rot += 360.0f;
        int bin = round(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        rotHist[bin].push_back(bestIdx2);
      }
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i != ind1 && i != ind2 && i != ind3) {
        for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
          trainMatch[rotHist[i][j]] = -1;
          occupied[rotHist[i][j]] = 0;
          nmatches--;
        }
      }
    }
  }
  return nmatches;
}

/* int ORBmatcher::SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, vpMatches12, const Sim3f &S12, const float th)
 * (ORBmatcher.cc:1448-1665).  Direction 1 entries: the KF1 features i1 that passed :1490-1525 with their projection
 * into KF2; direction 2 likewise (:1569-1603). */
int or_kf_search_by_sim3(const OrFrame *pKF1, const OrFrame *pKF2, int nq1, const int *idx1, const uint8_t *desc1,
                         const float *u1, const float *v1, const float *radius1, const int *level1, int nq2,
                         const int *idx2, const uint8_t *desc2, const float *u2, const float *v2, const float *radius2,
                         const int *level2, int *matches12) {
  const int N1 = pKF1->N, N2 = pKF2->N;
  vector<int> vnMatch1(N1, -1);
  vector<int> vnMatch2(N2, -1);
  for (int k = 0; k < nq1; k++) {
    const int i1 = idx1[k];
    const int nPredictedLevel = level1[k];
    const vector<size_t> vIndices = pKF2->KFGetFeaturesInArea(u1[k], v1[k], radius1[k]);
    if (vIndices.empty()) continue;
    const uint8_t *dMP = desc1 + (size_t)k * 32;
    int bestDist = INT_MAX;
    int bestIdx = -1;
    for (vector<size_t>::const_iterator vit = vIndices.begin(), vend = vIndices.end(); vit != vend; vit++) {
      const size_t idx = *vit;
      const OrKeyPoint &kp = pKF2->keys[idx];
      if (kp.octave < nPredictedLevel - 1 || kp.octave > nPredictedLevel) continue;
      const int dist = DescriptorDistance(dMP, pKF2->row(idx));
      if (dist < bestDist) {
        bestDist = dist;
        bestIdx = (int)idx;
      }
    }
    if (bestDist <= TH_HIGH) vnMatch1[i1] = bestIdx;
  }
  for (int k = 0; k < nq2; k++) {
    const int i2 = idx2[k];
    const int nPredictedLevel = level2[k];
    const vector<size_t> vIndices = pKF1->KFGetFeaturesInArea(u2[k], v2[k], radius2[k]);
    if (vIndices.empty()) continue;
    const uint8_t *dMP = desc2 + (size_t)k * 32;
    int bestDist = INT_MAX;
    int bestIdx = -1;
    for (vector<size_t>::const_iterator vit = vIndices.begin(), vend = vIndices.end(); vit != vend; vit++) {
      const size_t idx = *vit;
      const OrKeyPoint &kp = pKF1->keys[idx];
      if (kp.octave < nPredictedLevel - 1 || kp.octave > nPredictedLevel) continue;
      const int dist = DescriptorDistance(dMP, pKF1->row(idx));
      if (dist < bestDist) {
        bestDist = dist;
        bestIdx = (int)idx;
      }
    }
    if (bestDist <= TH_HIGH) vnMatch2[i2] = bestIdx;
  }
  int nFound = 0;
  for (int i1 = 0; i1 < N1; i1++) {
    matches12[i1] = -1;
    int i2 = vnMatch1[i1];
    if (i2 >= 0) {
      int i1b = vnMatch2[i2];
      if (i1b == i1) {
        matches12[i1] = i2;  // vpMatches12[i1] = vpMapPoints2[idx2]
        nFound++;
      }
    }
  }
  return nFound;
}

/* The flattened MapPoint bookkeeping of Fuse: slotMP[i] = id of pKF->GetMapPoint(i) or -1; mpObs / mpBad per id.
 * MapPoint::Replace(pMP) (MapPoint.cc:226-283) moves the loser's observations to the survivor (for THIS keyframe: the
 * slot now holds the survivor unless the survivor already is in the keyframe), sums the counters and flags the loser bad. */
static int fuse_decide_one(int pMP, int bestIdx, bool sim3Form, int *slotMP, int *mpObs, uint8_t *mpBad, int *action,
                           int *other) {
  const int pMPinKF = slotMP[bestIdx];
  *other = pMPinKF;
  if (pMPinKF >= 0) {
    if (!mpBad[pMPinKF]) {
      if (sim3Form) {
        *action = 5;  // vpReplacePoint[iMP] = pMPinKF
      } else if (mpObs[pMPinKF] > mpObs[pMP]) {
        *action = 2;  // pMP->Replace(pMPinKF)
        mpObs[pMPinKF] += mpObs[pMP];
        mpBad[pMP] = 1;
      } else {
        *action = 3;  // pMPinKF->Replace(pMP)
        mpObs[pMP] += mpObs[pMPinKF];
        mpBad[pMPinKF] = 1;
        slotMP[bestIdx] = pMP;
      }
    } else {
      *action = 4;
    }
  } else {
    *action = 1;  // pMP->AddObservation(pKF, bestIdx); pKF->AddMapPoint(pMP, bestIdx)
    slotMP[bestIdx] = pMP;
    mpObs[pMP] += 1;
  }
  return 1;
}

/* int ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, const float th, const bool bRight)
 * (ORBmatcher.cc:1148-1329).  Per map point that passed :1176-1236: uv, ur = uv(0) - bf * invz, radius, nPredictedLevel,
 * queryMP = its id.  mvInvLevelSigma2 = pKF->mvInvLevelSigma2. */
int or_kf_fuse(const OrFrame *pKF, int nQ, const int *queryMP, const uint8_t *mpDesc, const float *u, const float *v,
               const float *urArr, const float *radiusArr, const int *nPredictedLevelArr, int bRight,
               const float *mvInvLevelSigma2, int *slotMP, int *mpObs, uint8_t *mpBad, int *bestIdxOut,
               int *bestDistOut, int *action, int *other) {
  int nFused = 0;
  for (int i = 0; i < nQ; i++) {
    action[i] = 0, other[i] = -1;
    const int nPredictedLevel = nPredictedLevelArr[i];
    const float radius = radiusArr[i];
    const float ur = urArr[i];
    bestIdxOut[i] = -1, bestDistOut[i] = 256;
    const vector<size_t> vIndices = pKF->KFGetFeaturesInArea(u[i], v[i], radius, bRight != 0);
    if (vIndices.empty()) continue;
    const uint8_t *dMP = mpDesc + (size_t)i * 32;
    int bestDist = 256;
    int bestIdx = -1;
    for (vector<size_t>::const_iterator vit = vIndices.begin(), vend = vIndices.end(); vit != vend; vit++) {
      size_t idx = *vit;
      const OrKeyPoint &kp = pKF->key(idx, bRight != 0);
      const int &kpLevel = kp.octave;
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      if (pKF->mvuRight[idx] >= 0) {
        const float &kpx = kp.x;
        const float &kpy = kp.y;
        const float &kpr = pKF->mvuRight[idx];
        const float ex = u[i] - kpx;
        const float ey = v[i] - kpy;
        const float er = ur - kpr;
        const float e2 = ex * ex + ey * ey + er * er;
        if (e2 * mvInvLevelSigma2[kpLevel] > 7.8) continue;
      } else {
        const float &kpx = kp.x;
        const float &kpy = kp.y;
        const float ex = u[i] - kpx;
        const float ey = v[i] - kpy;
        const float e2 = ex * ex + ey * ey;
        if (e2 * mvInvLevelSigma2[kpLevel] > 5.99) continue;
      }
      if (bRight) idx += pKF->Nleft;
      const int dist = DescriptorDistance(dMP, pKF->row(idx));
      if (dist < bestDist) {
        bestDist = dist;
        bestIdx = (int)idx;
      }
    }
    bestIdxOut[i] = bestIdx, bestDistOut[i] = bestDist;
    if (bestDist <= TH_LOW) nFused += fuse_decide_one(queryMP[i], bestIdx, false, slotMP, mpObs, mpBad, &action[i], &other[i]);
  }
  return nFused;
}

/* int ORBmatcher::Fuse(KeyFrame *pKF, Sim3f &Scw, const vector<MapPoint*> &vpPoints, float th, vpReplacePoint)
 * (ORBmatcher.cc:1331-1446). */
int or_kf_fuse_sim3(const OrFrame *pKF, int nQ, const int *queryMP, const uint8_t *mpDesc, const float *u,
                    const float *v, const float *radiusArr, const int *nPredictedLevelArr, int *slotMP, int *mpObs,
                    uint8_t *mpBad, int *bestIdxOut, int *bestDistOut, int *action, int *other) {
  int nFused = 0;
  for (int iMP = 0; iMP < nQ; iMP++) {
    action[iMP] = 0, other[iMP] = -1;
    bestIdxOut[iMP] = -1, bestDistOut[iMP] = INT_MAX;
    const int nPredictedLevel = nPredictedLevelArr[iMP];
    const vector<size_t> vIndices = pKF->KFGetFeaturesInArea(u[iMP], v[iMP], radiusArr[iMP]);
    if (vIndices.empty()) continue;
    const uint8_t *dMP = mpDesc + (size_t)iMP * 32;
    int bestDist = INT_MAX;
    int bestIdx = -1;
    for (vector<size_t>::const_iterator vit = vIndices.begin(); vit != vIndices.end(); vit++) {
      const size_t idx = *vit;
      const int &kpLevel = pKF->keys[idx].octave;
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      int dist = DescriptorDistance(dMP, pKF->row(idx));
      if (dist < bestDist) {
        bestDist = dist;
        bestIdx = (int)idx;
      }
    }
    bestIdxOut[iMP] = bestIdx, bestDistOut[iMP] = bestDist;
    if (bestDist <= TH_LOW) nFused += fuse_decide_one(queryMP[iMP], bestIdx, true, slotMP, mpObs, mpBad, &action[iMP], &other[iMP]);
  }
  return nFused;
}

/* int ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, vbPrevMatched, vnMatches12, int windowSize)
 * (ORBmatcher.cc:643-756) on two OrFrames (the vbPrevMatched update :751-753 is the caller's). */
int or_frame_search_for_initialization(const OrFrame *F1, const OrFrame *F2, const float *prevX, const float *prevY,
                                       int windowSize, float mfNNratio, int mbCheckOrientation, int *vnMatches12) {
  int nmatches = 0;
  for (int i = 0; i < F1->N; i++) vnMatches12[i] = -1;
  vector<int> rotHist[HISTO_LENGTH];
  for (int i = 0; i < HISTO_LENGTH; i++) rotHist[i].reserve(500);
  const float factor = 1.0f / HISTO_LENGTH;
  vector<int> vMatchedDistance(F2->N, INT_MAX);
  vector<int> vnMatches21(F2->N, -1);
  for (size_t i1 = 0, iend1 = F1->N; i1 < iend1; i1++) {
    OrKeyPoint kp1 = F1->keys[i1];
    int level1 = kp1.octave;
    if (level1 > 0) continue;
    vector<size_t> vIndices2 = F2->GetFeaturesInArea(prevX[i1], prevY[i1], windowSize, level1, level1);
    if (vIndices2.empty()) continue;
    const uint8_t *d1 = F1->row(i1);
    int bestDist = INT_MAX;
    int bestDist2 = INT_MAX;
    int bestIdx2 = -1;
    for (vector<size_t>::iterator vit = vIndices2.begin(); vit != vIndices2.end(); vit++) {
      size_t i2 = *vit;
      const uint8_t *d2 = F2->row(i2);
      int dist = DescriptorDistance(d1, d2);
      if (vMatchedDistance[i2] <= dist) continue;
      if (dist < bestDist) {
        bestDist2 = bestDist;
        bestDist = dist;
        bestIdx2 = (int)i2;
      } else if (dist < bestDist2) {
        bestDist2 = dist;
      }
    }
    if (bestDist <= TH_LOW) {
      if (bestDist < (float)bestDist2 * mfNNratio) {
        if (vnMatches21[bestIdx2] >= 0) {
          vnMatches12[vnMatches21[bestIdx2]] = -1;
          nmatches--;
        }
        vnMatches12[i1] = bestIdx2;
        vnMatches21[bestIdx2] = (int)i1;
        vMatchedDistance[bestIdx2] = bestDist;
        nmatches++;
        if (mbCheckOrientation) {
          float rot = F1->keys[i1].angle - F2->keys[bestIdx2].angle;
          if (rot < 0.0) rot += 360.0f;
          int bin = round(rot * factor);
          if (bin == HISTO_LENGTH) bin = 0;
          rotHist[bin].push_back((int)i1);
        }
      }
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
        int idx1 = rotHist[i][j];
        if (vnMatches12[idx1] >= 0) {
          vnMatches12[idx1] = -1;
          nmatches--;
        }
      }
    }
  }
  return nmatches;
}

}  // extern "C"
