/*
 * match_oracle.cpp -- CPU ORACLE (test infrastructure, NOT product code).  PARITY UNPINNED.
 * Restatement of orb_slam3/src/ORBmatcher.cc searches and the Frame grid on flattened POD
 * views (see orb_oracle.h for the flattening conventions).  Loop structure, strict/non-strict
 * comparisons, thresholds and quirks follow the cited reference lines literally.
 */
#include "orb_oracle.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <vector>

namespace {
const int TH_HIGH = 100;      // ORBmatcher.cc:34
const int TH_LOW = 50;        // ORBmatcher.cc:35
const int HISTO_LENGTH = 30;  // ORBmatcher.cc:36

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
void ComputeThreeMaxima(const std::vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3) {
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

// rotation bin, e.g. ORBmatcher.cc:351-357
inline int rotBin(float angle1, float angle2) {
  const float factor = 1.0f / HISTO_LENGTH;
  float rot = angle1 - angle2;
  if (rot < 0.0) rot += 360.0f;
  int bin = (int)round(rot * factor);
  if (bin == HISTO_LENGTH) bin = 0;
  return bin;
}

// index of node id in a sorted id array (std::map::lower_bound)
inline int lowerBound(const int *ids, int n, int key) { return (int)(std::lower_bound(ids, ids + n, key) - ids); }
}  // namespace

// ---------------------------------------------------------------- Frame grid (Frame.h:49-50, Frame.cc:521-553,802-880)
struct OrGrid {
  static const int COLS = 64, ROWS = 48;
  float mnMinX, mnMinY, mnMaxX, mnMaxY, invW, invH;
  std::vector<OrKeyPoint> kps;
  std::vector<size_t> cells[COLS][ROWS];
};

extern "C" {

int or_descriptor_distance(const uint8_t *a, const uint8_t *b) { return DescriptorDistance(a, b); }

void or_three_maxima(const int *histoSizes, int L, int *ind1, int *ind2, int *ind3) {
  std::vector<std::vector<int>> h(L);
  for (int i = 0; i < L; i++) h[i].resize(histoSizes[i]);
  int a = -1, b = -1, c = -1;
  ComputeThreeMaxima(h.data(), L, a, b, c);
  *ind1 = a, *ind2 = b, *ind3 = c;
}

OrGrid *or_grid_build(const OrKeyPoint *kps, int n, float minX, float minY, float maxX, float maxY) {
  OrGrid *g = new OrGrid();
  g->mnMinX = minX, g->mnMinY = minY, g->mnMaxX = maxX, g->mnMaxY = maxY;
  // Frame.cc:378-379 (RGB-D ctor): mfGridElementWidthInv = FRAME_GRID_COLS / (mnMaxX - mnMinX)
  g->invW = static_cast<float>(OrGrid::COLS) / static_cast<float>(maxX - minX);
  g->invH = static_cast<float>(OrGrid::ROWS) / static_cast<float>(maxY - minY);
  g->kps.assign(kps, kps + n);
  for (int i = 0; i < n; i++) {  // AssignFeaturesToGrid + PosInGrid (Frame.cc:538-552, 870-880)
    int posX = (int)round((kps[i].x - g->mnMinX) * g->invW);
    int posY = (int)round((kps[i].y - g->mnMinY) * g->invH);
    if (posX < 0 || posX >= OrGrid::COLS || posY < 0 || posY >= OrGrid::ROWS) continue;
    g->cells[posX][posY].push_back((size_t)i);
  }
  return g;
}

void or_grid_destroy(OrGrid *g) { delete g; }

int or_grid_query(const OrGrid *g, float x, float y, float r, int minLevel, int maxLevel, int *outIdx, int cap) {
  // Frame::GetFeaturesInArea, Frame.cc:802-868 (Nleft == -1, bRight == false)
  int count = 0;
  float factorX = r, factorY = r;
  const int nMinCellX = std::max(0, (int)floor((x - g->mnMinX - factorX) * g->invW));
  if (nMinCellX >= OrGrid::COLS) return 0;
  const int nMaxCellX = std::min((int)OrGrid::COLS - 1, (int)ceil((x - g->mnMinX + factorX) * g->invW));
  if (nMaxCellX < 0) return 0;
  const int nMinCellY = std::max(0, (int)floor((y - g->mnMinY - factorY) * g->invH));
  if (nMinCellY >= OrGrid::ROWS) return 0;
  const int nMaxCellY = std::min((int)OrGrid::ROWS - 1, (int)ceil((y - g->mnMinY + factorY) * g->invH));
  if (nMaxCellY < 0) return 0;
  const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
  for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
    for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
      const std::vector<size_t> &vCell = g->cells[ix][iy];
      for (size_t j = 0, jend = vCell.size(); j < jend; j++) {
        const OrKeyPoint &kpUn = g->kps[vCell[j]];
        if (bCheckLevels) {
          if (kpUn.octave < minLevel) continue;
          if (maxLevel >= 0)
            if (kpUn.octave > maxLevel) continue;
        }
        const float distx = kpUn.x - x;
        const float disty = kpUn.y - y;
        if (fabs(distx) < factorX && fabs(disty) < factorY) {
          if (count < cap) outIdx[count] = (int)vCell[j];
          count++;
        }
      }
    }
  }
  return count;
}

int or_search_by_bow_kf_f_stereo(const uint8_t *kfDesc, const float *kfAngle, const uint8_t *kfValid, int nKF,
                                 const int *kfNodeId, const int *kfOff, const int *kfIdx, int kfNodes,
                                 const uint8_t *fDesc, const float *fAngle, int nF, int nleftF, const int *fNodeId,
                                 const int *fOff, const int *fIdx, int fNodes, float mfNNratio, int checkOri,
                                 int *matchF) {
  // ORBmatcher.cc:226-428.  nleftF = F.Nleft (-1: monocular / rectified stereo; >= 0: fisheye stereo, features
  // [0, Nleft) come from the left camera and [Nleft, N) from the right, descriptors vconcat'ed, Frame.cc:296)
  (void)nKF;
  for (int i = 0; i < nF; i++) matchF[i] = -1;
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  int KFit = 0, Fit = 0;
  while (KFit != kfNodes && Fit != fNodes) {
    if (kfNodeId[KFit] == fNodeId[Fit]) {
      for (int iKF = kfOff[KFit]; iKF < kfOff[KFit + 1]; iKF++) {
        const int realIdxKF = kfIdx[iKF];
        if (!kfValid[realIdxKF]) continue;  // !pMP || pMP->isBad()
        const uint8_t *dKF = kfDesc + (size_t)realIdxKF * 32;
        int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
        int bestDist1R = 256, bestIdxFR = -1, bestDist2R = 256;
        for (int iF = fOff[Fit]; iF < fOff[Fit + 1]; iF++) {
          const int realIdxF = fIdx[iF];
          if (matchF[realIdxF] >= 0) continue;
          const int dist = DescriptorDistance(dKF, fDesc + (size_t)realIdxF * 32);
          if (nleftF == -1 || realIdxF < nleftF) {  // (:277-315)
            if (dist < bestDist1) {
              bestDist2 = bestDist1;
              bestDist1 = dist;
              bestIdxF = realIdxF;
            } else if (dist < bestDist2) {
              bestDist2 = dist;
            }
          } else {  // (:317-326)
            if (dist < bestDist1R) {
              bestDist2R = bestDist1R;
              bestDist1R = dist;
              bestIdxFR = realIdxF;
            } else if (dist < bestDist2R) {
              bestDist2R = dist;
            }
          }
        }
        if (bestDist1 <= TH_LOW) {
          if (static_cast<float>(bestDist1) < mfNNratio * static_cast<float>(bestDist2)) {
            matchF[bestIdxF] = realIdxKF;
            if (checkOri) rotHist[rotBin(kfAngle[realIdxKF], fAngle[bestIdxF])].push_back(bestIdxF);
            nmatches++;
          }
          // (:362-389) nested in the left test, ratio test short-circuited by `|| true`
          if (bestDist1R <= TH_LOW) {
            matchF[bestIdxFR] = realIdxKF;
            if (checkOri) rotHist[rotBin(kfAngle[realIdxKF], fAngle[bestIdxFR])].push_back(bestIdxFR);
            nmatches++;
          }
          (void)bestDist2R;
        }
      }
      KFit++;
      Fit++;
    } else if (kfNodeId[KFit] < fNodeId[Fit]) {
      KFit = lowerBound(kfNodeId, kfNodes, fNodeId[Fit]);
    } else {
      Fit = lowerBound(fNodeId, fNodes, kfNodeId[KFit]);
    }
  }
  if (checkOri) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
        matchF[rotHist[i][j]] = -1;
        nmatches--;
      }
    }
  }
  return nmatches;
}

int or_search_by_bow_kf_f(const uint8_t *kfDesc, const float *kfAngle, const uint8_t *kfValid, int nKF,
                          const int *kfNodeId, const int *kfOff, const int *kfIdx, int kfNodes,
                          const uint8_t *fDesc, const float *fAngle, int nF, const int *fNodeId, const int *fOff,
                          const int *fIdx, int fNodes, float mfNNratio, int checkOri, int *matchF) {
  return or_search_by_bow_kf_f_stereo(kfDesc, kfAngle, kfValid, nKF, kfNodeId, kfOff, kfIdx, kfNodes, fDesc, fAngle, nF,
                                      -1, fNodeId, fOff, fIdx, fNodes, mfNNratio, checkOri, matchF);
}

int or_search_by_bow_kf_kf(const uint8_t *desc1, const float *angle1, const uint8_t *valid1, int n1,
                           const int *nodeId1, const int *off1, const int *idx1v, int nodes1,
                           const uint8_t *desc2, const float *angle2, const uint8_t *valid2, int n2,
                           const int *nodeId2, const int *off2, const int *idx2v, int nodes2, float mfNNratio,
                           int checkOri, int *matches12) {
  // ORBmatcher.cc:758-900 with NLeft == -1
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  std::vector<bool> vbMatched2(n2, false);
  std::vector<int> rotHist[HISTO_LENGTH];
  int nmatches = 0;
  int f1it = 0, f2it = 0;
  while (f1it != nodes1 && f2it != nodes2) {
    if (nodeId1[f1it] == nodeId2[f2it]) {
      for (int i1 = off1[f1it]; i1 < off1[f1it + 1]; i1++) {
        const int idx1 = idx1v[i1];
        if (!valid1[idx1]) continue;
        const uint8_t *d1 = desc1 + (size_t)idx1 * 32;
        int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
        for (int i2 = off2[f2it]; i2 < off2[f2it + 1]; i2++) {
          const int idx2 = idx2v[i2];
          if (vbMatched2[idx2] || !valid2[idx2]) continue;
          int dist = DescriptorDistance(d1, desc2 + (size_t)idx2 * 32);
          if (dist < bestDist1) {
            bestDist2 = bestDist1;
            bestDist1 = dist;
            bestIdx2 = idx2;
          } else if (dist < bestDist2) {
            bestDist2 = dist;
          }
        }
        if (bestDist1 < TH_LOW) {
          if (static_cast<float>(bestDist1) < mfNNratio * static_cast<float>(bestDist2)) {
            matches12[idx1] = bestIdx2;
            vbMatched2[bestIdx2] = true;
            if (checkOri) rotHist[rotBin(angle1[idx1], angle2[bestIdx2])].push_back(idx1);
            nmatches++;
          }
        }
      }
      f1it++;
      f2it++;
    } else if (nodeId1[f1it] < nodeId2[f2it]) {
      f1it = lowerBound(nodeId1, nodes1, nodeId2[f2it]);
    } else {
      f2it = lowerBound(nodeId2, nodes2, nodeId1[f1it]);
    }
  }
  if (checkOri) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
        matches12[rotHist[i][j]] = -1;
        nmatches--;
      }
    }
  }
  return nmatches;
}

int or_search_for_triangulation(const uint8_t *desc1, const float *angle1, const uint8_t *eligible1, int n1,
                                const int *nodeId1, const int *off1, const int *idx1v, int nodes1,
                                const uint8_t *desc2, const float *angle2, const uint8_t *eligible2, int n2,
                                const int *nodeId2, const int *off2, const int *idx2v, int nodes2,
                                const uint32_t *pairOk, const int *pairOff, int checkOri, int *matches12) {
  // ORBmatcher.cc:902-1146 with NLeft == -1.  eligible1 = !GetMapPoint(idx1) && (!bOnlyStereo || bStereo1)
  // (:969-979), eligible2 = !GetMapPoint(idx2) && (!bOnlyStereo || bStereo2) (:998-1005; vbMatched2 is never set by
  // the reference).  The geometric predicate of (:1031-1071) -- epipole distance, epipolarConstrain or bCoarse -- is a
  // pure function of (idx1, idx2): bit pairOff[s] + i1 * n2(s) + i2 of pairOk for the s-th SHARED node (NULL: true).
  (void)n2;
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  std::vector<int> rotHist[HISTO_LENGTH];
  int nmatches = 0, shared = 0;
  int f1it = 0, f2it = 0;
  while (f1it != nodes1 && f2it != nodes2) {
    if (nodeId1[f1it] == nodeId2[f2it]) {
      const int nn2 = off2[f2it + 1] - off2[f2it];
      for (int i1 = off1[f1it]; i1 < off1[f1it + 1]; i1++) {
        const int idx1 = idx1v[i1];
        if (!eligible1[idx1]) continue;
        const uint8_t *d1 = desc1 + (size_t)idx1 * 32;
        int bestDist = TH_LOW;
        int bestIdx2 = -1;
        for (int i2 = off2[f2it]; i2 < off2[f2it + 1]; i2++) {
          const int idx2 = idx2v[i2];
          if (!eligible2[idx2]) continue;
          const int dist = DescriptorDistance(d1, desc2 + (size_t)idx2 * 32);
          if (dist > TH_LOW || dist > bestDist) continue;
          bool ok = true;
          if (pairOk) {
            const long long bit = (long long)pairOff[shared] + (long long)(i1 - off1[f1it]) * nn2 + (i2 - off2[f2it]);
            ok = (pairOk[bit >> 5] >> (bit & 31)) & 1u;
          }
          if (ok) {
            bestIdx2 = idx2;
            bestDist = dist;
          }
        }
        if (bestIdx2 >= 0) {
          matches12[idx1] = bestIdx2;
          nmatches++;
          if (checkOri) rotHist[rotBin(angle1[idx1], angle2[bestIdx2])].push_back(idx1);
        }
      }
      shared++;
      f1it++;
      f2it++;
    } else if (nodeId1[f1it] < nodeId2[f2it]) {
      f1it = lowerBound(nodeId1, nodes1, nodeId2[f2it]);
    } else {
      f2it = lowerBound(nodeId2, nodes2, nodeId1[f1it]);
    }
  }
  if (checkOri) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
        matches12[rotHist[i][j]] = -1;
        nmatches--;
      }
    }
  }
  return nmatches;
}

int or_search_by_projection_last(const uint8_t *qDesc, const float *qAngle, const uint8_t *queryBlocks, int nQ,
                                 const int *candOff, const int *candIdx, const uint8_t *tDesc, const float *tAngle,
                                 uint8_t *trainBlocked, int nT, int thHigh, int checkOri, int *trainMatch) {
  // ORBmatcher.cc:1686-1784 (left/mono block) + 1855-1875, geometry hoisted to the caller.
  (void)nT;
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  for (int q = 0; q < nQ; q++) {
    if (candOff[q] == candOff[q + 1]) continue;  // vIndices2.empty()
    const uint8_t *dMP = qDesc + (size_t)q * 32;
    int bestDist = 256, bestIdx2 = -1;
    for (int c = candOff[q]; c < candOff[q + 1]; c++) {
      const int i2 = candIdx[c];
      if (trainBlocked[i2]) continue;  // mvpMapPoints[i2] && Observations() > 0
      const int dist = DescriptorDistance(dMP, tDesc + (size_t)i2 * 32);
      if (dist < bestDist) {
        bestDist = dist;
        bestIdx2 = i2;
      }
    }
    if (bestDist <= thHigh) {
      trainMatch[bestIdx2] = q;
      trainBlocked[bestIdx2] = queryBlocks[q];
      nmatches++;
      if (checkOri) rotHist[rotBin(qAngle[q], tAngle[bestIdx2])].push_back(bestIdx2);
    }
  }
  if (checkOri) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i != ind1 && i != ind2 && i != ind3) {
        for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
          trainMatch[rotHist[i][j]] = -1;   // CurrentFrame.mvpMapPoints[...] = NULL (:1869):
          trainBlocked[rotHist[i][j]] = 0;  // no map point any more, so the feature is free again
          nmatches--;
        }
      }
    }
  }
  return nmatches;
}

int or_search_by_projection_local(const uint8_t *qDesc, const uint8_t *queryBlocks, int nQ, const int *candOff,
                                  const int *candIdx, const uint8_t *tDesc, const int *tOctave,
                                  uint8_t *trainBlocked, int nT, float mfNNratio, int *trainMatch) {
  // ORBmatcher.cc:48-144 (mbTrackInView block, Nleft == -1)
  (void)nT;
  int nmatches = 0;
  for (int q = 0; q < nQ; q++) {
    if (candOff[q] == candOff[q + 1]) continue;
    const uint8_t *MPdescriptor = qDesc + (size_t)q * 32;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int c = candOff[q]; c < candOff[q + 1]; c++) {
      const int idx = candIdx[c];
      if (trainBlocked[idx]) continue;
      const int dist = DescriptorDistance(MPdescriptor, tDesc + (size_t)idx * 32);
      if (dist < bestDist) {
        bestDist2 = bestDist;
        bestDist = dist;
        bestLevel2 = bestLevel;
        bestLevel = tOctave[idx];
        bestIdx = idx;
      } else if (dist < bestDist2) {
        bestLevel2 = tOctave[idx];
        bestDist2 = dist;
      }
    }
    if (bestDist <= TH_HIGH) {
      if (bestLevel == bestLevel2 && bestDist > mfNNratio * bestDist2) continue;
      if (bestLevel != bestLevel2 || bestDist <= mfNNratio * bestDist2) {
        trainMatch[bestIdx] = q;
        trainBlocked[bestIdx] = queryBlocks[q];
        nmatches++;
      }
    }
  }
  return nmatches;
}

void or_distinctive_descriptors(const uint8_t *desc, const int *off, int ngroups, int *best) {
  for (int g = 0; g < ngroups; g++) {
    const size_t N = (size_t)(off[g + 1] - off[g]);
    best[g] = -1;
    if (N == 0) continue;
    const uint8_t *d = desc + (size_t)off[g] * 32;
    std::vector<float> Distances(N * N);
    for (size_t i = 0; i < N; i++) {
      Distances[i * N + i] = 0;
      for (size_t j = i + 1; j < N; j++) {
        int distij = DescriptorDistance(d + i * 32, d + j * 32);
        Distances[i * N + j] = (float)distij;
        Distances[j * N + i] = (float)distij;
      }
    }
    int BestMedian = INT_MAX, BestIdx = 0;
    for (size_t i = 0; i < N; i++) {
      std::vector<int> vDists(Distances.begin() + i * N, Distances.begin() + (i + 1) * N);
      std::sort(vDists.begin(), vDists.end());
      int median = vDists[(size_t)(0.5 * (N - 1))];
      if (median < BestMedian) {
        BestMedian = median;
        BestIdx = (int)i;
      }
    }
    best[g] = BestIdx;
  }
}

int or_search_window(const uint8_t *qDesc, const uint8_t *queryBlocks, int nQ, const int *candOff, const int *candIdx,
                     const uint8_t *tDesc, uint8_t *trainBlocked, int nT, int thHigh, int *qBestIdx, int *qBestDist,
                     int *trainMatch) {
  // e.g. ORBmatcher.cc:494-524: bestDist = 256; skip taken candidates; strict '<'; accept <= threshold
  (void)nT;
  int nmatches = 0;
  for (int q = 0; q < nQ; q++) {
    int bestDist = 256, bestIdx = -1;
    for (int c = candOff[q]; c < candOff[q + 1]; c++) {
      const int idx = candIdx[c];
      if (trainBlocked && trainBlocked[idx]) continue;
      const int dist = DescriptorDistance(qDesc + (size_t)q * 32, tDesc + (size_t)idx * 32);
      if (dist < bestDist) {
        bestDist = dist;
        bestIdx = idx;
      }
    }
    qBestIdx[q] = bestIdx;
    qBestDist[q] = bestDist;
    if (bestIdx >= 0 && bestDist <= thHigh) {
      if (trainMatch) trainMatch[bestIdx] = q;
      if (trainBlocked) trainBlocked[bestIdx] = queryBlocks ? queryBlocks[q] : 0;
      nmatches++;
    }
  }
  return nmatches;
}

int or_search_for_initialization(const uint8_t *desc1, const float *angle1, const int *octave1, int n1,
                                 const int *candOff, const int *candIdx, const uint8_t *desc2, const float *angle2,
                                 int n2, float mfNNratio, int checkOri, int *vnMatches12) {
  // ORBmatcher.cc:643-748 (the vbPrevMatched update :751-753 is the caller's)
  int nmatches = 0;
  for (int i = 0; i < n1; i++) vnMatches12[i] = -1;
  std::vector<int> rotHist[HISTO_LENGTH];
  std::vector<int> vMatchedDistance(n2, INT_MAX);
  std::vector<int> vnMatches21(n2, -1);
  for (int i1 = 0; i1 < n1; i1++) {
    if (octave1[i1] > 0) continue;
    if (candOff[i1] == candOff[i1 + 1]) continue;
    const uint8_t *d1 = desc1 + (size_t)i1 * 32;
    int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
    for (int c = candOff[i1]; c < candOff[i1 + 1]; c++) {
      const int i2 = candIdx[c];
      int dist = DescriptorDistance(d1, desc2 + (size_t)i2 * 32);
      if (vMatchedDistance[i2] <= dist) continue;
      if (dist < bestDist) {
        bestDist2 = bestDist;
        bestDist = dist;
        bestIdx2 = i2;
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
        vnMatches21[bestIdx2] = i1;
        vMatchedDistance[bestIdx2] = bestDist;
        nmatches++;
        if (checkOri) rotHist[rotBin(angle1[i1], angle2[bestIdx2])].push_back(i1);
      }
    }
  }
  if (checkOri) {
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

void or_block_best2(const uint8_t *a, int na, const uint8_t *b, int nb, int *best, int *second, int *argbest) {
  for (int i = 0; i < na; i++) {
    int bestDist1 = 256, bestIdx = -1, bestDist2 = 256;
    for (int j = 0; j < nb; j++) {
      const int dist = DescriptorDistance(a + (size_t)i * 32, b + (size_t)j * 32);
      if (dist < bestDist1) {
        bestDist2 = bestDist1;
        bestDist1 = dist;
        bestIdx = j;
      } else if (dist < bestDist2) {
        bestDist2 = dist;
      }
    }
    best[i] = bestDist1;
    second[i] = bestDist2;
    argbest[i] = bestIdx;
  }
}

}  // extern "C"
