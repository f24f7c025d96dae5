// vsg_walks.h -- the ORDERED host passes of the windowed ORBmatcher searches (orb_slam3/src/ORBmatcher.cc).
//
// The device evaluates what is data-parallel (grid windows, level / stereo / chi-square gates, Hamming distances);
// what the reference does sequentially over its queries -- a claimed feature blocks later queries, the rotation
// histogram is filled in match order -- runs here, on the host, over the per-query candidate lists the device
// returns.  A dozen candidates per query: a few microseconds per call, like the reference's own loops.
// Shared by vsg_match.hip (host candidate lists in) and vsg_frame.hip (resident frames).
#pragma once
#include <stdint.h>

#include <cmath>
#include <vector>

namespace vsg {
namespace walk {

enum { TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30 };  // ORBmatcher.cc:34-36

// One candidate as the device writes it: feature index (grid-local), Hamming distance, keypoint octave.
inline int ent_idx(uint32_t e) { return (int)(e & 0x7FFFu); }
inline int ent_dist(uint32_t e) { return (int)((e >> 15) & 0x1FFu); }
inline int ent_oct(uint32_t e) { return (int)((e >> 24) & 0xFu); }

// Per-query candidate lists: CSR (off[nq + 1], cnt == nullptr) or segments (off[q] = start, cnt[q] = length -- what
// the device writes: every query reserves its segment of one compact array, in no particular order).
struct CandView {
  const uint32_t *ent = nullptr;
  const int32_t *off = nullptr;
  const int32_t *cnt = nullptr;
  const uint32_t *begin(int q) const { return ent + off[q]; }
  int size(int q) const { return cnt ? cnt[q] : off[q + 1] - off[q]; }
  // the device has just written these lists: every line is a miss; ask for the list of a later query early
  void prefetch(int q, int nq) const {
    if (q < nq) __builtin_prefetch(ent + off[q]);
  }
};

// rotation-consistency bin (e.g. ORBmatcher.cc:351-356, 1771-1777)
inline int rot_bin(float angle1, float angle2) {
  const float factor = 1.0f / HISTO_LENGTH;
  float rot = angle1 - angle2;
  if (rot < 0.0) rot += 360.0f;
  int bin = (int)std::round(rot * factor);
  if (bin == HISTO_LENGTH) bin = 0;
  return bin;
}

// ORBmatcher::ComputeThreeMaxima (ORBmatcher.cc:2002-2043)
inline void three_maxima(const std::vector<int> *histo, int L, int &ind1, int &ind2, int &ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int s = (int)histo[i].size();
    if (s > max1) {
      max3 = max2, max2 = max1, max1 = s;
      ind3 = ind2, ind2 = ind1, ind1 = i;
    } else if (s > max2) {
      max3 = max2, max2 = s;
      ind3 = ind2, ind2 = i;
    } else if (s > max3) {
      max3 = s, ind3 = i;
    }
  }
  if (max2 < 0.1f * (float)max1) {
    ind2 = -1;
    ind3 = -1;
  } else if (max3 < 0.1f * (float)max1) {
    ind3 = -1;
  }
}

// entries of the losing bins are handed to `drop`
template <class Drop>
inline void filter_rotation(std::vector<int> (&rotHist)[HISTO_LENGTH], Drop drop) {
  int ind1 = -1, ind2 = -1, ind3 = -1;
  three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
  for (int i = 0; i < HISTO_LENGTH; i++) {
    if (i == ind1 || i == ind2 || i == ind3) continue;
    for (size_t j = 0; j < rotHist[i].size(); j++) drop(rotHist[i][j]);
  }
}

// SearchByProjection(CurrentFrame, LastFrame) (ORBmatcher.cc:1686-1875).  Queries [0, nq) are the left / mono
// windows; with nleft != -1 queries [nq, 2 nq) are the right-camera windows of the same map points (:1786-1853) and
// their indices are right-grid local.  t_angle(i) = angle of CurrentFrame's keypoint i (mvKeysUn, or mvKeys ||
// mvKeysRight).
template <class AngleOf>
inline int search_last(const CandView &cv, int nq, int nleft, const float *q_angle, const uint8_t *mp_observed,
                       AngleOf t_angle, int th_high, bool check_orientation, uint8_t *train_blocked,
                       int32_t *train_match) {
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  const int nall = nleft != -1 ? 2 * nq : nq;
  for (int q = 0; q < nq; q++) {
    cv.prefetch(q + 8, nall);
    if (nleft != -1) cv.prefetch(nq + q + 8, nall);
    const int n = cv.size(q);
    if (n == 0) continue;  // vIndices2.empty() (:1727) -- skips the right block of this map point as well
    const uint32_t *e = cv.begin(q);
    int bestDist = 256, bestIdx2 = -1;
    for (int c = 0; c < n; c++) {
      const int i2 = ent_idx(e[c]);
      if (train_blocked[i2]) continue;  // mvpMapPoints[i2] with Observations() > 0 (:1739-1741)
      const int d = ent_dist(e[c]);
      if (d < bestDist) {
        bestDist = d;
        bestIdx2 = i2;
      }
    }
    if (bestDist <= th_high && bestIdx2 >= 0) {  // :1762
      train_match[bestIdx2] = q;
      train_blocked[bestIdx2] = mp_observed ? mp_observed[q] : 0;
      nmatches++;
      if (check_orientation) rotHist[rot_bin(q_angle[q], t_angle(bestIdx2))].push_back(bestIdx2);
    }
    if (nleft != -1) {  // :1786-1853
      const int nr = cv.size(nq + q);
      const uint32_t *er = cv.begin(nq + q);
      int bestDistR = 256, bestIdxR = -1;
      for (int c = 0; c < nr; c++) {
        const int i2 = ent_idx(er[c]);
        if (train_blocked[i2 + nleft]) continue;
        const int d = ent_dist(er[c]);
        if (d < bestDistR) {
          bestDistR = d;
          bestIdxR = i2;
        }
      }
      if (bestDistR <= th_high && bestIdxR >= 0) {
        train_match[bestIdxR + nleft] = q;
        train_blocked[bestIdxR + nleft] = mp_observed ? mp_observed[q] : 0;
        nmatches++;
        if (check_orientation) rotHist[rot_bin(q_angle[q], t_angle(bestIdxR + nleft))].push_back(bestIdxR + nleft);
      }
    }
  }
  if (check_orientation)  // :1855-1875: mvpMapPoints[i] = NULL -- the feature is free again
    filter_rotation(rotHist, [&](int i2) {
      train_match[i2] = -1;
      train_blocked[i2] = 0;
      nmatches--;
    });
  return nmatches;
}

// best + second best with their pyramid levels over the non-blocked candidates (ORBmatcher.cc:83-120, 172-195)
struct Best2 {
  int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
};
inline Best2 scan_best2(const uint32_t *e, int n, const uint8_t *train_blocked, int idx_off) {
  Best2 b;
  for (int c = 0; c < n; c++) {
    const int idx = ent_idx(e[c]);
    if (train_blocked[idx + idx_off]) continue;
    const int d = ent_dist(e[c]);
    if (d < b.bestDist) {
      b.bestDist2 = b.bestDist;
      b.bestDist = d;
      b.bestLevel2 = b.bestLevel;
      b.bestLevel = ent_oct(e[c]);
      b.bestIdx = idx;
    } else if (d < b.bestDist2) {
      b.bestLevel2 = ent_oct(e[c]);
      b.bestDist2 = d;
    }
  }
  return b;
}

// SearchByProjection(F, vpMapPoints) (ORBmatcher.cc:48-214).  Queries [0, n_mp) = mbTrackInView windows (inactive
// ones have empty lists), [n_mp, 2 n_mp) = the right-camera windows when nleft != -1.
inline int search_local(const CandView &cv, int n_mp, int nleft, const uint8_t *in_view, const uint8_t *in_view_r,
                        const int32_t *scale_level_r, const uint8_t *mp_observed, float nnratio,
                        const int32_t *left_to_right, const int32_t *right_to_left, uint8_t *train_blocked,
                        int32_t *train_match) {
  int nmatches = 0;
  const int nall = nleft != -1 ? 2 * n_mp : n_mp;
  for (int q = 0; q < n_mp; q++) {
    cv.prefetch(q + 8, nall);
    if (nleft != -1) cv.prefetch(n_mp + q + 8, nall);
    const bool inR = nleft != -1 && in_view_r && in_view_r[q];
    if (!in_view[q] && !inR) continue;  // :50-51
    const uint8_t blocks = mp_observed ? mp_observed[q] : 0;
    if (in_view[q]) {
      const int n = cv.size(q);
      if (n > 0) {
        const Best2 b = scan_best2(cv.begin(q), n, train_blocked, 0);
        if (b.bestDist <= TH_HIGH && b.bestIdx >= 0) {  // :123
          // the `continue` of :125-126 leaves the whole map point, right block included
          if (b.bestLevel == b.bestLevel2 && (float)b.bestDist > nnratio * (float)b.bestDist2) continue;
          if (b.bestLevel != b.bestLevel2 || (float)b.bestDist <= nnratio * (float)b.bestDist2) {  // :128
            train_match[b.bestIdx] = q;
            train_blocked[b.bestIdx] = blocks;
            if (nleft != -1 && left_to_right && left_to_right[b.bestIdx] != -1) {  // :132-137
              train_match[left_to_right[b.bestIdx] + nleft] = q;
              train_blocked[left_to_right[b.bestIdx] + nleft] = blocks;
              nmatches++;
            }
            nmatches++;
          }
        }
      }
    }
    if (inR) {  // :146-214
      if (scale_level_r[q] == -1) continue;  // :149
      const int n = cv.size(n_mp + q);
      if (n == 0) continue;  // :156-157
      const Best2 b = scan_best2(cv.begin(n_mp + q), n, train_blocked, nleft);
      if (b.bestDist <= TH_HIGH && b.bestIdx >= 0) {  // :196
        if (b.bestLevel == b.bestLevel2 && (float)b.bestDist > nnratio * (float)b.bestDist2) continue;  // :198-199
        if (right_to_left && right_to_left[b.bestIdx] != -1) {  // :201-206
          train_match[right_to_left[b.bestIdx]] = q;
          train_blocked[right_to_left[b.bestIdx]] = blocks;
          nmatches++;
        }
        train_match[b.bestIdx + nleft] = q;  // :208
        train_blocked[b.bestIdx + nleft] = blocks;
        nmatches++;
      }
    }
  }
  return nmatches;
}

// Best-only scan over the candidates that `taken` does not exclude; strict '<' from `init` (256 or INT_MAX).
template <class Taken>
inline void scan_best(const uint32_t *e, int n, Taken taken, int init, int &bestDist, int &bestIdx) {
  bestDist = init;
  bestIdx = -1;
  for (int c = 0; c < n; c++) {
    const int idx = ent_idx(e[c]);
    if (taken(idx)) continue;
    const int d = ent_dist(e[c]);
    if (d < bestDist) {
      bestDist = d;
      bestIdx = idx;
    }
  }
}

// SearchByProjection(KeyFrame*, Sim3, vpPoints, vpMatched, th, ratioHamming) (ORBmatcher.cc:494-524, :602-634)
inline int search_sim3_projection(const CandView &cv, int nq, float ratio_hamming, int32_t *matched) {
  int nmatches = 0;
  for (int q = 0; q < nq; q++) {
    cv.prefetch(q + 8, nq);
    const int n = cv.size(q);
    if (n == 0) continue;  // :487-488
    int bestDist, bestIdx;
    scan_best(cv.begin(q), n, [&](int idx) { return matched[idx] != -1; }, 256, bestDist, bestIdx);
    if ((float)bestDist <= (float)TH_LOW * ratio_hamming && bestIdx >= 0) {  // :520
      matched[bestIdx] = q;
      nmatches++;
    }
  }
  return nmatches;
}

// SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) (ORBmatcher.cc:1938-1997)
template <class AngleOf>
inline int search_kf_projection(const CandView &cv, int nq, const float *kf_angle, AngleOf t_angle, int orb_dist,
                                bool check_orientation, uint8_t *occupied, int32_t *train_match) {
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  for (int q = 0; q < nq; q++) {
    cv.prefetch(q + 8, nq);
    const int n = cv.size(q);
    if (n == 0) continue;  // :1936-1937
    int bestDist, bestIdx2;
    scan_best(cv.begin(q), n, [&](int idx) { return occupied[idx] != 0; }, 256, bestDist, bestIdx2);
    if (bestDist <= orb_dist && bestIdx2 >= 0) {  // :1957
      train_match[bestIdx2] = q;
      occupied[bestIdx2] = 1;
      nmatches++;
      if (check_orientation) rotHist[rot_bin(kf_angle[q], t_angle(bestIdx2))].push_back(bestIdx2);
    }
  }
  if (check_orientation)  // :1976-1997
    filter_rotation(rotHist, [&](int i2) {
      train_match[i2] = -1;
      occupied[i2] = 0;
      nmatches--;
    });
  return nmatches;
}

// SearchForInitialization (ORBmatcher.cc:643-748).  Queries = F1 keypoints (lists are empty for octave > 0).
template <class Angle1, class Angle2>
inline int search_initialization(const CandView &cv, int n1, int n2, const int32_t *octave1, Angle1 angle1,
                                 Angle2 angle2, float nnratio, bool check_orientation, int32_t *matches12) {
  int nmatches = 0;
  std::vector<int> vMatchedDistance((size_t)n2, 0x7FFFFFFF), vnMatches21((size_t)n2, -1);
  std::vector<int> rotHist[HISTO_LENGTH];
  for (int i1 = 0; i1 < n1; i1++) {
    cv.prefetch(i1 + 8, n1);
    if (octave1 && octave1[i1] > 0) continue;  // :659-661
    const int n = cv.size(i1);
    if (n == 0) continue;
    const uint32_t *e = cv.begin(i1);
    int bestDist = 0x7FFFFFFF, bestDist2 = 0x7FFFFFFF, bestIdx2 = -1;
    for (int c = 0; c < n; c++) {
      const int i2 = ent_idx(e[c]);
      const int d = ent_dist(e[c]);
      if (vMatchedDistance[i2] <= d) continue;  // :682
      if (d < bestDist) {
        bestDist2 = bestDist;
        bestDist = d;
        bestIdx2 = i2;
      } else if (d < bestDist2) {
        bestDist2 = d;
      }
    }
    if (bestDist <= TH_LOW) {
      if ((float)bestDist < (float)bestDist2 * nnratio) {  // :697-699
        if (vnMatches21[bestIdx2] >= 0) {
          matches12[vnMatches21[bestIdx2]] = -1;
          nmatches--;
        }
        matches12[i1] = bestIdx2;
        vnMatches21[bestIdx2] = i1;
        vMatchedDistance[bestIdx2] = bestDist;
        nmatches++;
        if (check_orientation) rotHist[rot_bin(angle1(i1), angle2(bestIdx2))].push_back(i1);
      }
    }
  }
  if (check_orientation)  // :726-748
    filter_rotation(rotHist, [&](int idx1) {
      if (matches12[idx1] >= 0) {
        matches12[idx1] = -1;
        nmatches--;
      }
    });
  return nmatches;
}

}  // namespace walk
}  // namespace vsg
