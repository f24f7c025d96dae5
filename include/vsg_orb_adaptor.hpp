// vsg_orb_adaptor.hpp -- header-only C++ mirror of the reference's ORBextractor surface over the C ABI.
//
// `vsg::ORBextractor` has the constructor, operator() and getters of VS_GRAPHS::ORBextractor
// (orb_slam3/include/ORBextractor.h:42-119) on POD types, so it compiles without OpenCV.  When OpenCV is
// available (the reference's own build), define VSG_WITH_OPENCV before including this header to get the
// exact reference signature
//     int operator()(cv::InputArray, cv::InputArray, std::vector<cv::KeyPoint>&, cv::OutputArray, std::vector<int>&)
// and a `mvImagePyramid` refresh for Frame::ComputeStereoMatches (Frame.cc:964,1054-1069).
// See INTEGRATION.md for the three-line change in Tracking.cc / Frame.cc that swaps the extractor.
#pragma once
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "vsg_orb.h"

#ifdef VSG_WITH_OPENCV
#include <opencv2/core/core.hpp>
#include <cstring>
#endif

namespace vsg {

class ORBextractor {
 public:
  enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };  // ORBextractor.h:45-49

  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device = 0)
      : nlevels_(nlevels), scaleFactor_(scaleFactor) {
    int rc = vsg_orb_create(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device, 1, &h_);
    if (rc != VSG_OK) throw std::runtime_error(std::string("vsg_orb_create: ") + vsg_last_error());
    scale_.resize(nlevels), inv_scale_.resize(nlevels), sigma2_.resize(nlevels), inv_sigma2_.resize(nlevels);
    vsg_orb_get_tables(h_, scale_.data(), inv_scale_.data(), sigma2_.data(), inv_sigma2_.data(), nullptr, nullptr);
#ifdef VSG_WITH_OPENCV
    mvImagePyramid.resize(nlevels);
#endif
  }
  ~ORBextractor() { vsg_orb_destroy(h_); }
  ORBextractor(const ORBextractor &) = delete;
  ORBextractor &operator=(const ORBextractor &) = delete;

  // POD form of operator(): gray CV_8UC1 rows x cols, `stride` bytes per row; vLappingArea = {lap0, lap1}.
  // Returns monoIndex, or -1 for an empty image like the reference (ORBextractor.cc:1087-1088).
  int operator()(const uint8_t *gray, int rows, int cols, int stride, std::vector<vsg_keypoint> &keypoints,
                 std::vector<uint8_t> &descriptors, const std::vector<int> &vLappingArea) {
    keypoints.clear();
    descriptors.clear();
    if (!gray || rows <= 0 || cols <= 0) return -1;
    const int cap = vsg_orb_capacity(h_, rows, cols);
    if (cap < 0) throw std::runtime_error(std::string("vsg_orb_capacity: ") + vsg_last_error());
    keypoints.resize(cap);
    descriptors.resize((size_t)cap * 32);
    int n = 0;
    const int mono = vsg_orb_extract(h_, gray, rows, cols, stride, vLappingArea.at(0), vLappingArea.at(1),
                                     keypoints.data(), descriptors.data(), cap, &n);
    if (mono < 0) throw std::runtime_error(std::string("vsg_orb_extract: ") + vsg_last_error());
    keypoints.resize(n);
    descriptors.resize((size_t)n * 32);
    return mono;
  }

#ifdef VSG_WITH_OPENCV
  static_assert(sizeof(cv::KeyPoint) == sizeof(vsg_keypoint), "cv::KeyPoint must be the 28-byte record");
  // The reference signature (ORBextractor.h:59-61).  The mask is ignored, as in the reference (:58).
  int operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint> &_keypoints,
                 cv::OutputArray _descriptors, std::vector<int> &vLappingArea) {
    if (_image.empty()) return -1;
    cv::Mat image = _image.getMat();
    CV_Assert(image.type() == CV_8UC1);
    const int cap = vsg_orb_capacity(h_, image.rows, image.cols);
    if (cap < 0) CV_Error(cv::Error::StsBadArg, vsg_last_error());
    std::vector<vsg_keypoint> kps(cap);
    cv::Mat desc(cap, 32, CV_8U);
    int n = 0;
    const int mono = vsg_orb_extract(h_, image.data, image.rows, image.cols, (int)image.step, vLappingArea[0],
                                     vLappingArea[1], kps.data(), desc.data, cap, &n);
    if (mono < 0) CV_Error(cv::Error::StsError, vsg_last_error());
    _keypoints.resize(n);
    if (n) std::memcpy((void *)_keypoints.data(), kps.data(), (size_t)n * sizeof(vsg_keypoint));
    if (n == 0)
      _descriptors.release();  // ORBextractor.cc:1105-1106
    else
      desc.rowRange(0, n).copyTo(_descriptors);
    return mono;
  }
  // Refresh mvImagePyramid (bordered buffers + ROI views, as ComputePyramid leaves them) on demand; only the stereo
  // matcher reads it (Frame.cc:964,1054-1069).  ONE device-to-host copy for all levels into a buffer that is reused
  // from call to call; the cv::Mat headers point into it (no allocation, no per-level transfer).
  void DownloadPyramid(int frame = 0) {
    std::vector<size_t> off(nlevels_);
    const int need = vsg_orb_copy_pyramid(h_, frame, nullptr, 0, off.data());
    if (need < 0) CV_Error(cv::Error::StsError, vsg_last_error());
    if (pyr_host_.size() < (size_t)need) pyr_host_.resize((size_t)need);
    if (vsg_orb_copy_pyramid(h_, frame, pyr_host_.data(), pyr_host_.size(), off.data()) < 0)
      CV_Error(cv::Error::StsError, vsg_last_error());
    for (int l = 0; l < nlevels_; l++) {
      int w = 0, h = 0;
      vsg_orb_level_size(h_, l, &w, &h);
      cv::Mat temp(h + 38, w + 38, CV_8UC1, pyr_host_.data() + off[l], (size_t)(w + 38));
      mvImagePyramid[l] = temp(cv::Rect(19, 19, w, h));
    }
  }
  std::vector<cv::Mat> mvImagePyramid;  // ORBextractor.h:93
#endif

  // keypoint capacity per frame for images of this size (>= nfeatures + 3 * nlevels)
  int capacity(int rows, int cols) const {
    const int cap = vsg_orb_capacity(h_, rows, cols);
    if (cap < 0) throw std::runtime_error(std::string("vsg_orb_capacity: ") + vsg_last_error());
    return cap;
  }
  int GetLevels() const { return nlevels_; }
  float GetScaleFactor() const { return scaleFactor_; }
  std::vector<float> GetScaleFactors() const { return scale_; }
  std::vector<float> GetInverseScaleFactors() const { return inv_scale_; }
  std::vector<float> GetScaleSigmaSquares() const { return sigma2_; }
  std::vector<float> GetInverseScaleSigmaSquares() const { return inv_sigma2_; }
  vsg_orb *handle() const { return h_; }

 private:
  vsg_orb *h_ = nullptr;
  int nlevels_;
  float scaleFactor_;
  std::vector<float> scale_, inv_scale_, sigma2_, inv_sigma2_;
#ifdef VSG_WITH_OPENCV
  std::vector<uint8_t> pyr_host_;  // backing store of mvImagePyramid
#endif
};

// static int ORBmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b)  (ORBmatcher.h:40, .cc:2047-2063).
// One pair is eight 32-bit popcounts: host arithmetic (a kernel launch costs four orders of magnitude more); bulk
// distances go through vsg_hamming_pairs / the search entry points.
inline int DescriptorDistance(const uint8_t *a, const uint8_t *b) {
  int dist = 0;
  for (int i = 0; i < 8; i++) {
    uint32_t x, y;
    std::memcpy(&x, a + 4 * i, 4);
    std::memcpy(&y, b + 4 * i, 4);
    dist += __builtin_popcount(x ^ y);
  }
  return dist;
}

inline void check(int rc, const char *what) {
  if (rc < 0) throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + vsg_last_error());
}

// DBoW2::FeatureVector is std::map<NodeId, std::vector<unsigned>> (FeatureVector.h:25-26); the C ABI takes it as
// CSR with ascending node ids, which is the map's iteration order.
struct FeatureVectorCSR {
  std::vector<int32_t> node, off, idx;
  FeatureVectorCSR() : off(1, 0) {}
  template <class Map>
  explicit FeatureVectorCSR(const Map &fv) : off(1, 0) {
    for (const auto &kv : fv) {
      node.push_back((int32_t)kv.first);
      for (auto i : kv.second) idx.push_back((int32_t)i);
      off.push_back((int32_t)idx.size());
    }
  }
  int nodes() const { return (int)node.size(); }
};

// One side of a match as the reference's Frame / KeyFrame members expose it: mDescriptors rows (Frame.h:280),
// mvKeysUn[i].angle / .octave, and per feature whether it carries a usable MapPoint.
struct FeatureView {
  const uint8_t *desc = nullptr;   // n x 32, contiguous
  const float *angle = nullptr;    // n (may be null when the call does not check orientation)
  const int32_t *octave = nullptr; // n (only SearchByProjection(Frame&, vector<MapPoint*>) / SearchForInitialization)
  int n = 0;
};

// CSR candidate lists = the concatenated results of Frame::GetFeaturesInArea per query (see vsg::FrameGrid).
struct Candidates {
  std::vector<int32_t> off, idx;
};

// Frame::mGrid (Frame.h:290) on the device: AssignFeaturesToGrid + GetFeaturesInArea (Frame.cc:521-553, 802-868).
class FrameGrid {
 public:
  FrameGrid(const vsg_keypoint *keysUn, int n, float mnMinX, float mnMinY, float mnMaxX, float mnMaxY, int device = 0) {
    check(vsg_grid_build(device, keysUn, n, mnMinX, mnMinY, mnMaxX, mnMaxY, &g_), "vsg_grid_build");
  }
  ~FrameGrid() { vsg_grid_destroy(g_); }
  FrameGrid(const FrameGrid &) = delete;
  FrameGrid &operator=(const FrameGrid &) = delete;
  // nq windows at once; minLevel/maxLevel may be null (= -1, -1 as KeyFrame::GetFeaturesInArea)
  Candidates GetFeaturesInArea(const float *x, const float *y, const float *r, const int32_t *minLevel,
                               const int32_t *maxLevel, int nq) const {
    Candidates c;
    c.off.resize(nq + 1);
    c.idx.resize(64 * (size_t)nq + 64);
    int total = vsg_grid_query(g_, x, y, r, minLevel, maxLevel, nq, c.off.data(), c.idx.data(), (int)c.idx.size());
    check(total, "vsg_grid_query");
    if (total > (int)c.idx.size()) {
      c.idx.resize(total);
      check(vsg_grid_query(g_, x, y, r, minLevel, maxLevel, nq, c.off.data(), c.idx.data(), total), "vsg_grid_query");
    }
    c.idx.resize(total);
    return c;
  }

 private:
  vsg_grid *g_ = nullptr;
};

// ORBVocabulary (= DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>, ORBVocabulary.h:29-30): loadFromBinFile
// image + transform() as Frame::ComputeBoW calls it (Frame.cc:882-889).
class ORBVocabulary {
 public:
  ORBVocabulary(const uint8_t *bin_image, size_t size, int device = 0) {
    check(vsg_vocab_load(device, bin_image, size, &v_), "vsg_vocab_load");
  }
  ~ORBVocabulary() { vsg_vocab_destroy(v_); }
  ORBVocabulary(const ORBVocabulary &) = delete;
  ORBVocabulary &operator=(const ORBVocabulary &) = delete;
  // mBowVec as std::map<WordId, WordValue> (BowVector.h:57-58), mFeatVec as CSR
  void transform(const uint8_t *desc, int n, std::map<unsigned, double> &bowVec, FeatureVectorCSR &featVec,
                 int levelsup = 4) const {
    std::vector<int32_t> ids(n > 0 ? n : 1);
    std::vector<double> vals(n > 0 ? n : 1);
    featVec.node.assign(n > 0 ? n : 1, 0);
    featVec.off.assign((n > 0 ? n : 1) + 1, 0);
    featVec.idx.assign(n > 0 ? n : 1, 0);
    int nb = 0, nf = 0;
    check(vsg_bow_transform(v_, desc, n, levelsup, ids.data(), vals.data(), (int)ids.size(), &nb, featVec.node.data(),
                            featVec.off.data(), featVec.idx.data(), (int)featVec.node.size(), &nf, nullptr, nullptr,
                            nullptr),
          "vsg_bow_transform");
    bowVec.clear();
    for (int i = 0; i < nb; ++i) bowVec.emplace_hint(bowVec.end(), (unsigned)ids[i], vals[i]);
    featVec.node.resize(nf);
    featVec.off.resize(nf + 1);
    featVec.idx.resize(featVec.off[nf]);
  }
  // Frame::ComputeBoW / KeyFrame::ComputeBoW (Frame.cc:882-889, KeyFrame.cc:100-110) on a resident frame's descriptors
  // (`frame` = vsg::ResidentFrame::handle()): nothing goes up, mBowVec / mFeatVec come back in final form, and the
  // FeatureVector also STAYS on the device inside the frame (Frame::mFeatVec) for the ResidentMatcher::SearchByBoW
  // overloads that take no FeatureVector.
  void ComputeBoW(vsg_frame *frame, std::map<unsigned, double> &bowVec, FeatureVectorCSR &featVec, int levelsup = 4) const {
    const int n = vsg_frame_size(frame);
    check(n, "vsg_frame_size");
    const int cap = n > 0 ? n : 1;
    std::vector<int32_t> ids(cap);
    std::vector<double> vals(cap);
    featVec.node.assign(cap, 0), featVec.off.assign(cap + 1, 0), featVec.idx.assign(cap, 0);
    int nb = 0, nf = 0;
    check(vsg_frame_bow_transform(v_, frame, levelsup, ids.data(), vals.data(), cap, &nb, featVec.node.data(), featVec.off.data(),
                                  featVec.idx.data(), cap, &nf, nullptr, nullptr, nullptr),
          "vsg_frame_bow_transform");
    bowVec.clear();
    for (int i = 0; i < nb; ++i) bowVec.emplace_hint(bowVec.end(), (unsigned)ids[i], vals[i]);
    featVec.node.resize(nf), featVec.off.resize(nf + 1), featVec.idx.resize(featVec.off[nf]);
  }
  vsg_vocab *handle() const { return v_; }

 private:
  vsg_vocab *v_ = nullptr;
};

// VS_GRAPHS::ORBmatcher (ORBmatcher.h:34-99) on flattened views.  The reference's methods take Frame& / KeyFrame* /
// MapPoint* graphs; a maintainer's glue builds the views from those (INTEGRATION.md shows it for each method) and
// writes the returned indices back as MapPoint* assignments.  Same constants, same return values (number of matches).
class ORBmatcher {
 public:
  static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;  // ORBmatcher.cc:34-36

  explicit ORBmatcher(float nnratio = 0.6f, bool checkOri = true, int device = 0)
      : mfNNratio(nnratio), mbCheckOrientation(checkOri), device_(device) {}

  static int DescriptorDistance(const uint8_t *a, const uint8_t *b) { return vsg::DescriptorDistance(a, b); }

  // SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches)  (ORBmatcher.cc:226-428)
  // kfValid[i] = (vpMapPointsKF[i] && !isBad()); matchF[iF] = KF feature index or -1.
  int SearchByBoW(const FeatureView &kf, const uint8_t *kfValid, const FeatureVectorCSR &kfFeatVec,
                  const FeatureView &f, const FeatureVectorCSR &fFeatVec, std::vector<int32_t> &matchF) const {
    matchF.assign(f.n, -1);
    int rc = vsg_search_by_bow_kf_f(device_, kf.desc, kf.angle, kfValid, kf.n, kfFeatVec.node.data(),
                                    kfFeatVec.off.data(), kfFeatVec.idx.data(), kfFeatVec.nodes(), f.desc, f.angle,
                                    f.n, fFeatVec.node.data(), fFeatVec.off.data(), fFeatVec.idx.data(),
                                    fFeatVec.nodes(), mfNNratio, mbCheckOrientation, matchF.data());
    check(rc, "vsg_search_by_bow_kf_f");
    return rc;
  }

  // SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12)  (ORBmatcher.cc:758-900)
  int SearchByBoW(const FeatureView &kf1, const uint8_t *valid1, const FeatureVectorCSR &fv1, const FeatureView &kf2,
                  const uint8_t *valid2, const FeatureVectorCSR &fv2, std::vector<int32_t> &matches12) const {
    matches12.assign(kf1.n, -1);
    int rc = vsg_search_by_bow_kf_kf(device_, kf1.desc, kf1.angle, valid1, kf1.n, fv1.node.data(), fv1.off.data(),
                                     fv1.idx.data(), fv1.nodes(), kf2.desc, kf2.angle, valid2, kf2.n,
                                     fv2.node.data(), fv2.off.data(), fv2.idx.data(), fv2.nodes(), mfNNratio,
                                     mbCheckOrientation, matches12.data());
    check(rc, "vsg_search_by_bow_kf_kf");
    return rc;
  }

  // SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, vMatchedPairs, bOnlyStereo, bCoarse)  (ORBmatcher.cc:902-1146)
  // eligible = no MapPoint yet (and stereo when bOnlyStereo).  pairOk(i1, i2, idx1, idx2) is the caller's geometric
  // predicate (:1031-1071: epipole gate, then bCoarse || epipolarConstrain); it is evaluated here for every pair of
  // every shared node and shipped as a bitmask.  Pass nullptr for "every pair passes".
  template <class Pred>
  int SearchForTriangulation(const FeatureView &kf1, const uint8_t *eligible1, const FeatureVectorCSR &fv1,
                             const FeatureView &kf2, const uint8_t *eligible2, const FeatureVectorCSR &fv2, Pred pairOk,
                             std::vector<std::pair<size_t, size_t>> &vMatchedPairs) const {
    std::vector<uint32_t> bits(1, 0u);
    std::vector<int32_t> off(1, 0);
    size_t a = 0, b = 0;
    while (a < fv1.node.size() && b < fv2.node.size()) {  // the reference's merge-join, shared nodes in id order
      if (fv1.node[a] == fv2.node[b]) {
        for (int i1 = fv1.off[a]; i1 < fv1.off[a + 1]; i1++)
          for (int i2 = fv2.off[b]; i2 < fv2.off[b + 1]; i2++) {
            const long long bit = (long long)off.back() + (long long)(i1 - fv1.off[a]) * (fv2.off[b + 1] - fv2.off[b]) +
                                  (i2 - fv2.off[b]);
            if ((size_t)(bit >> 5) >= bits.size()) bits.resize((size_t)(bit >> 5) + 64, 0u);
            if (eligible1[fv1.idx[i1]] && eligible2[fv2.idx[i2]] && pairOk(fv1.idx[i1], fv2.idx[i2]))
              bits[bit >> 5] |= 1u << (bit & 31);
          }
        off.push_back(off.back() + (fv1.off[a + 1] - fv1.off[a]) * (fv2.off[b + 1] - fv2.off[b]));
        a++, b++;
      } else if (fv1.node[a] < fv2.node[b]) {
        a++;
      } else {
        b++;
      }
    }
    bits.resize((size_t)(off.back() >> 5) + 2, 0u);
    std::vector<int32_t> m12(kf1.n > 0 ? kf1.n : 1, -1);
    int rc = vsg_search_for_triangulation(device_, kf1.desc, kf1.angle, eligible1, kf1.n, fv1.node.data(), fv1.off.data(),
                                          fv1.idx.data(), fv1.nodes(), kf2.desc, kf2.angle, eligible2, kf2.n,
                                          fv2.node.data(), fv2.off.data(), fv2.idx.data(), fv2.nodes(), bits.data(),
                                          off.data(), mbCheckOrientation, m12.data());
    check(rc, "vsg_search_for_triangulation");
    vMatchedPairs.clear();
    for (int i = 0; i < kf1.n; i++)
      if (m12[i] >= 0) vMatchedPairs.emplace_back((size_t)i, (size_t)m12[i]);  // :1136-1143
    return rc;
  }

  // SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono)  (ORBmatcher.cc:1667-1878).
  // q = LastFrame's map points that project into the image (descriptor + keypoint angle of the last frame's feature),
  // cand = CurrentFrame.GetFeaturesInArea per q; trainBlocked / trainMatch are CurrentFrame-sized.
  int SearchByProjection(const FeatureView &q, const uint8_t *queryBlocks, const Candidates &cand,
                         const FeatureView &cur, std::vector<uint8_t> &trainBlocked,
                         std::vector<int32_t> &trainMatch) const {
    trainMatch.assign(cur.n, -1);
    trainBlocked.resize(cur.n, 0);
    int rc = vsg_search_by_projection_last(device_, q.desc, q.angle, queryBlocks, q.n, cand.off.data(),
                                           cand.idx.data(), cur.desc, cur.angle, trainBlocked.data(), cur.n, TH_HIGH,
                                           mbCheckOrientation, trainMatch.data());
    check(rc, "vsg_search_by_projection_last");
    return rc;
  }

  // SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th, ...)  (ORBmatcher.cc:42-216)
  int SearchByProjection(const FeatureView &q, const uint8_t *queryBlocks, const Candidates &cand,
                         const FeatureView &frame, std::vector<uint8_t> &trainBlocked,
                         std::vector<int32_t> &trainMatch, bool localMap) const {
    (void)localMap;
    trainMatch.assign(frame.n, -1);
    trainBlocked.resize(frame.n, 0);
    int rc = vsg_search_by_projection_local(device_, q.desc, queryBlocks, q.n, cand.off.data(), cand.idx.data(),
                                            frame.desc, frame.octave, trainBlocked.data(), frame.n, mfNNratio,
                                            trainMatch.data());
    check(rc, "vsg_search_by_projection_local");
    return rc;
  }

  // Common core of SearchByProjection(KeyFrame*, Sim3, ...) x2, SearchByProjection(Frame&, KeyFrame*, ...),
  // SearchBySim3 and Fuse (see include/vsg_orb.h: vsg_search_window).
  int SearchWindow(const FeatureView &q, const uint8_t *queryBlocks, const Candidates &cand, const FeatureView &train,
                   int thHigh, std::vector<int32_t> &qBestIdx, std::vector<int32_t> &qBestDist,
                   std::vector<uint8_t> *trainBlocked = nullptr, std::vector<int32_t> *trainMatch = nullptr) const {
    qBestIdx.assign(q.n, -1);
    qBestDist.assign(q.n, 256);
    if (trainMatch) trainMatch->assign(train.n, -1);
    if (trainBlocked) trainBlocked->resize(train.n, 0);
    int rc = vsg_search_window(device_, q.desc, queryBlocks, q.n, cand.off.data(), cand.idx.data(), train.desc,
                               trainBlocked ? trainBlocked->data() : nullptr, train.n, thHigh, qBestIdx.data(),
                               qBestDist.data(), trainMatch ? trainMatch->data() : nullptr);
    check(rc, "vsg_search_window");
    return rc;
  }

  // SearchForInitialization(Frame &F1, Frame &F2, vbPrevMatched, vnMatches12, windowSize)  (ORBmatcher.cc:643-756)
  int SearchForInitialization(const FeatureView &f1, const Candidates &cand, const FeatureView &f2,
                              std::vector<int32_t> &vnMatches12) const {
    vnMatches12.assign(f1.n, -1);
    int rc = vsg_search_for_initialization(device_, f1.desc, f1.angle, f1.octave, f1.n, cand.off.data(),
                                           cand.idx.data(), f2.desc, f2.angle, f2.n, mfNNratio, mbCheckOrientation,
                                           vnMatches12.data());
    check(rc, "vsg_search_for_initialization");
    return rc;
  }

 protected:
  float mfNNratio;          // ORBmatcher.h:97
  bool mbCheckOrientation;  // ORBmatcher.h:98
  int device_;
};

// Frame::ComputeStereoMatches (Frame.cc:957-1127): mvuRight / mvDepth from the two extractors' device pyramids.
inline int ComputeStereoMatches(const ORBextractor &left, const ORBextractor &right,
                                const std::vector<vsg_keypoint> &kpsL, const uint8_t *descL,
                                const std::vector<vsg_keypoint> &kpsR, const uint8_t *descR, float mb, float mbf,
                                std::vector<float> &mvuRight, std::vector<float> &mvDepth) {
  mvuRight.assign(kpsL.size(), -1.0f);
  mvDepth.assign(kpsL.size(), -1.0f);
  int rc = vsg_stereo_matches(left.handle(), 0, right.handle(), 0, kpsL.data(), descL, (int)kpsL.size(), kpsR.data(),
                              descR, (int)kpsR.size(), mb, mbf, mvuRight.data(), mvDepth.data());
  check(rc, "vsg_stereo_matches");
  return rc;
}

// MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:380-415) for many map points in one call.
inline std::vector<int32_t> ComputeDistinctiveDescriptors(const uint8_t *desc, const std::vector<int32_t> &off,
                                                          int device = 0) {
  std::vector<int32_t> best(off.empty() ? 0 : off.size() - 1, -1);
  if (!best.empty()) check(vsg_distinctive_descriptors(device, desc, off.data(), (int)best.size(), best.data()),
                           "vsg_distinctive_descriptors");
  return best;
}


// ---------------------------------------------------------------------------------------------------------------
// Device-resident features of a Frame / KeyFrame (include/vsg_orb.h: vsg_frame): what the searches read -- mvKeysUn
// (or mvKeys || mvKeysRight with Nleft), mDescriptors, mvuRight, mGrid / mGridRight (Frame.h:280-290) -- stays on the
// GPU between calls.  A maintainer adds one `std::shared_ptr<vsg::ResidentFrame> mpResident` to Frame and KeyFrame
// (KeyFrame's constructor copies the pointer from the Frame it is built from, like it copies mGrid, KeyFrame.cc:37-60)
// and fills it at the end of the Frame constructors, after UndistortKeyPoints / ComputeStereo* / AssignFeaturesToGrid.
class ResidentFrame {
 public:
  explicit ResidentFrame(int capacity, int device = 0) {
    check(vsg_frame_create(device, capacity, &f_), "vsg_frame_create");
  }
  ~ResidentFrame() { vsg_frame_destroy(f_); }
  ResidentFrame(const ResidentFrame &) = delete;
  ResidentFrame &operator=(const ResidentFrame &) = delete;
  // keys = mvKeysUn (Nleft == -1) or mvKeys followed by mvKeysRight; uRight = mvuRight.data() or nullptr
  void Upload(const vsg_keypoint *keys, const uint8_t *desc, const float *uRight, int N, int Nleft, float mnMinX,
              float mnMinY, float mnMaxX, float mnMaxY) {
    check(vsg_frame_upload(f_, keys, desc, uRight, N, Nleft, mnMinX, mnMinY, mnMaxX, mnMaxY), "vsg_frame_upload");
  }
  // straight out of the extractor that just ran (no distortion: mvKeysUn == mvKeys)
  void FromExtractor(const ORBextractor &ex, const std::vector<vsg_keypoint> &keys, float mnMinX, float mnMinY,
                     float mnMaxX, float mnMaxY, int index = 0) {
    check(vsg_frame_from_extractor(f_, ex.handle(), index, keys.data(), (int)keys.size(), mnMinX, mnMinY, mnMaxX, mnMaxY),
          "vsg_frame_from_extractor");
  }
  // The same for a distorted pinhole camera (mDistCoef(0) != 0: TUM1.yaml, RealSense_D435i.yaml): UndistortKeyPoints
  // (Frame.cc:891-921) runs on the device inside the grid launch; mvKeysUn comes back for the host's own use.
  // K4 = {fx, fy, cx, cy} of mK, dist = mDistCoef's 4 or 5 floats, bounds = ImageBounds(...) below.
  void FromExtractorUndistort(const ORBextractor &ex, const std::vector<vsg_keypoint> &mvKeys, const float K4[4],
                              const float *dist, int ndist, float mnMinX, float mnMinY, float mnMaxX, float mnMaxY,
                              std::vector<vsg_keypoint> *mvKeysUn, int index = 0) {
    if (mvKeysUn) mvKeysUn->resize(mvKeys.size());
    check(vsg_frame_from_extractor_undistort(f_, ex.handle(), index, mvKeys.data(), (int)mvKeys.size(), K4, dist, ndist,
                                             mnMinX, mnMinY, mnMaxX, mnMaxY, mvKeysUn ? mvKeysUn->data() : nullptr),
          "vsg_frame_from_extractor_undistort");
  }
  // ExtractORB + UndistortKeyPoints + AssignFeaturesToGrid in one call and one wait (vsg_orb_extract_to_frame): the
  // front end of the monocular / RGB-D Frame constructors.  Returns monoIndex; mvKeys / descriptors sized to N.
  int ExtractInto(ORBextractor &ex, const uint8_t *gray, int rows, int cols, int stride, const int lapping[2],
                  std::vector<vsg_keypoint> &mvKeys, std::vector<uint8_t> &descriptors, const float K4[4],
                  const float *dist, int ndist, float mnMinX, float mnMinY, float mnMaxX, float mnMaxY,
                  std::vector<vsg_keypoint> *mvKeysUn) {
    const int cap = vsg_orb_capacity(ex.handle(), rows, cols);
    check(cap, "vsg_orb_capacity");
    mvKeys.resize(cap), descriptors.resize((size_t)cap * 32);
    if (mvKeysUn) mvKeysUn->resize(cap);
    int n = 0;
    const int mono = vsg_orb_extract_to_frame(ex.handle(), gray, rows, cols, stride, lapping[0], lapping[1], mvKeys.data(),
                                              descriptors.data(), cap, &n, f_, K4, dist, ndist, mnMinX, mnMinY, mnMaxX,
                                              mnMaxY, mvKeysUn ? mvKeysUn->data() : nullptr);
    check(mono, "vsg_orb_extract_to_frame");
    mvKeys.resize(n), descriptors.resize((size_t)n * 32);
    if (mvKeysUn) mvKeysUn->resize(n);
    return mono;
  }
  // Frame::ComputeImageBounds (Frame.cc:924-955): {mnMinX, mnMinY, mnMaxX, mnMaxY}
  static void ImageBounds(int cols, int rows, const float K4[4], const float *dist, int ndist, float out[4]) {
    check(vsg_camera_image_bounds(cols, rows, K4, dist, ndist, out), "vsg_camera_image_bounds");
  }
  int N() const { return vsg_frame_size(f_); }
  vsg_frame *handle() const { return f_; }

  // Frame::GetFeaturesInArea / KeyFrame::GetFeaturesInArea for many windows at once
  Candidates GetFeaturesInArea(const float *x, const float *y, const float *r, const int32_t *minLevel,
                               const int32_t *maxLevel, int nq, bool bRight = false) const {
    Candidates c;
    c.off.resize(nq + 1);
    c.idx.resize(32 * (size_t)nq + 64);
    int total = vsg_frame_features_in_area(f_, x, y, r, minLevel, maxLevel, bRight, nq, c.off.data(), c.idx.data(),
                                           (int)c.idx.size());
    check(total, "vsg_frame_features_in_area");
    if (total > (int)c.idx.size()) {
      c.idx.resize(total);
      check(vsg_frame_features_in_area(f_, x, y, r, minLevel, maxLevel, bRight, nq, c.off.data(), c.idx.data(), total),
            "vsg_frame_features_in_area");
    }
    c.idx.resize(total);
    return c;
  }

 private:
  vsg_frame *f_ = nullptr;
};

// The stereo Frame constructor's tail and the first tracking step in ONE call and one wait (vsg_frame_stereo_bow_search):
// ComputeStereoMatches (Frame.cc:957-1127) -> ComputeBoW (Frame.cc:882-889) -> SearchByBoW(pKF, F) (ORBmatcher.cc:226-428;
// Tracking::TrackReferenceKeyFrame, Tracking.cc:2766-2777).  left / right = the extractors that just processed the two eyes,
// fl / fr = their resident features; pKF = nullptr: no search (the first Frame).  Returns the number of BoW matches.
struct StereoBowResult {
  std::vector<float> mvuRight, mvDepth;
  int nStereo = 0;
  std::map<unsigned, double> mBowVec;
  FeatureVectorCSR mFeatVec;
  std::vector<int32_t> matchF;  // per feature of F: index of the KeyFrame feature whose map point it takes, or -1
};
inline int StereoBowSearch(const ORBextractor &left, const ORBextractor &right, ResidentFrame &fl, ResidentFrame &fr, float mb,
                           float mbf, const ORBVocabulary &voc, ResidentFrame *pKF, const uint8_t *kfValid, float nnratio,
                           bool checkOri, StereoBowResult &out, int levelsup = 4) {
  const int n = fl.N(), cap = n > 0 ? n : 1;
  out.mvuRight.assign(cap, -1.f), out.mvDepth.assign(cap, -1.f), out.matchF.assign(cap, -1);
  std::vector<int32_t> ids(cap);
  std::vector<double> vals(cap);
  out.mFeatVec.node.assign(cap, 0), out.mFeatVec.off.assign(cap + 1, 0), out.mFeatVec.idx.assign(cap, 0);
  int nb = 0, nf = 0, nm = 0;
  check(vsg_frame_stereo_bow_search(left.handle(), 0, right.handle(), 0, fl.handle(), fr.handle(), mb, mbf, out.mvuRight.data(),
                                    out.mvDepth.data(), &out.nStereo, voc.handle(), levelsup, ids.data(), vals.data(), cap, &nb,
                                    out.mFeatVec.node.data(), out.mFeatVec.off.data(), out.mFeatVec.idx.data(), cap, &nf,
                                    pKF ? pKF->handle() : nullptr, kfValid, nnratio, checkOri ? 1 : 0, out.matchF.data(), &nm),
        "vsg_frame_stereo_bow_search");
  out.mBowVec.clear();
  for (int i = 0; i < nb; ++i) out.mBowVec.emplace_hint(out.mBowVec.end(), (unsigned)ids[i], vals[i]);
  out.mFeatVec.node.resize(nf), out.mFeatVec.off.resize(nf + 1), out.mFeatVec.idx.resize(out.mFeatVec.off[nf]);
  out.mvuRight.resize(n), out.mvDepth.resize(n), out.matchF.resize(n);
  return nm;
}

// Projected map points as the routines' geometry code leaves them (one entry per point that passed the routine's
// visibility / distance tests); see include/vsg_orb.h for which members each search reads.
struct ProjectedPoints {
  std::vector<uint8_t> desc;       // n x 32: pMP->GetDescriptor()
  std::vector<uint8_t> observed;   // pMP->Observations() > 0
  std::vector<float> u, v, ur;     // projection (ur = u - mbf * invz where the routine uses it)
  std::vector<float> radius;       // th * mvScaleFactors[level]
  std::vector<int32_t> level;      // nPredictedLevel / nLastOctave
  std::vector<float> angle;        // keypoint angle of the source feature (rotation check)
  std::vector<float> uR, vR;       // right-camera projection (Nleft != -1)
  int n() const { return (int)level.size(); }
};

// The remaining ORBmatcher searches on resident frames (ORBmatcher.h:44-87).  Outputs are feature -> query-index maps;
// the maintainer's glue writes the MapPoint* assignments back.
class ResidentMatcher {
 public:
  static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;
  explicit ResidentMatcher(float nnratio = 0.6f, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

  // SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono)  (ORBmatcher.cc:1667-1878)
  // direction: 0 neither, 1 bForward, 2 bBackward.  trainBlocked[i] = mvpMapPoints[i] && Observations() > 0 (in/out).
  int SearchByProjection(ResidentFrame &CurrentFrame, const ProjectedPoints &last, float th, int direction,
                         const std::vector<float> &mvScaleFactors, std::vector<uint8_t> &trainBlocked,
                         std::vector<int32_t> &trainMatch) const {
    trainMatch.assign(CurrentFrame.N(), -1);
    trainBlocked.resize(CurrentFrame.N(), 0);
    int rc = vsg_frame_search_by_projection_last(
        CurrentFrame.handle(), last.n(), last.desc.data(), last.observed.data(), last.u.data(), last.v.data(),
        last.ur.empty() ? nullptr : last.ur.data(), last.uR.empty() ? nullptr : last.uR.data(),
        last.vR.empty() ? nullptr : last.vR.data(), last.level.data(), last.angle.data(), th, direction,
        mvScaleFactors.data(), (int)mvScaleFactors.size(), mbCheckOrientation, trainBlocked.data(), trainMatch.data());
    check(rc, "vsg_frame_search_by_projection_last");
    return rc;
  }

  // SearchByProjection(KeyFrame *pKF, Sim3f &Scw, vpPoints, vpMatched, th, ratioHamming)  (ORBmatcher.cc:430-528) and
  // its twin with vpPointsKFs / vpMatchedKF (:530-641): matched[i] != -1 = vpMatched[i] is set; new entries = query index
  int SearchByProjection(ResidentFrame &pKF, const ProjectedPoints &pts, float ratioHamming,
                         std::vector<int32_t> &matched) const {
    matched.resize(pKF.N(), -1);
    int rc = vsg_frame_search_by_projection_sim3(pKF.handle(), pts.n(), pts.desc.data(), pts.u.data(), pts.v.data(),
                                                 pts.radius.data(), pts.level.data(), ratioHamming, matched.data());
    check(rc, "vsg_frame_search_by_projection_sim3");
    return rc;
  }

  // SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, sAlreadyFound, th, ORBdist)  (ORBmatcher.cc:1880-2000)
  int SearchByProjection(ResidentFrame &CurrentFrame, const ProjectedPoints &kfPoints, int ORBdist,
                         std::vector<uint8_t> &occupied, std::vector<int32_t> &trainMatch) const {
    trainMatch.assign(CurrentFrame.N(), -1);
    occupied.resize(CurrentFrame.N(), 0);
    int rc = vsg_frame_search_by_projection_kf(CurrentFrame.handle(), kfPoints.n(), kfPoints.desc.data(),
                                               kfPoints.u.data(), kfPoints.v.data(), kfPoints.radius.data(),
                                               kfPoints.level.data(), kfPoints.angle.data(), ORBdist,
                                               mbCheckOrientation, occupied.data(), trainMatch.data());
    check(rc, "vsg_frame_search_by_projection_kf");
    return rc;
  }

  // SearchBySim3(pKF1, pKF2, vpMatches12, S12, th)  (ORBmatcher.cc:1448-1665): idx1 / idx2 = feature index of every
  // projected point in its own KeyFrame; matches12[i1] = i2 where both directions agree
  int SearchBySim3(ResidentFrame &pKF1, ResidentFrame &pKF2, const std::vector<int32_t> &idx1,
                   const ProjectedPoints &into2, const std::vector<int32_t> &idx2, const ProjectedPoints &into1,
                   std::vector<int32_t> &matches12) const {
    matches12.assign(pKF1.N(), -1);
    int rc = vsg_frame_search_by_sim3(pKF1.handle(), pKF2.handle(), into2.n(), idx1.data(), into2.desc.data(),
                                      into2.u.data(), into2.v.data(), into2.radius.data(), into2.level.data(), into1.n(),
                                      idx2.data(), into1.desc.data(), into1.u.data(), into1.v.data(),
                                      into1.radius.data(), into1.level.data(), matches12.data());
    check(rc, "vsg_frame_search_by_sim3");
    return rc;
  }

  // Fuse(pKF, vpMapPoints, th, bRight)  (ORBmatcher.cc:1148-1329): the search; bestIdx / bestDist per point.  The
  // caller then walks the points in order and does Replace / AddObservation on its MapPoint graph for those with
  // bestDist <= TH_LOW (:1308-1327) -- or hands flattened ids to vsg_fuse_decide.
  int Fuse(ResidentFrame &pKF, const ProjectedPoints &pts, bool bRight, const std::vector<float> &mvInvLevelSigma2,
           std::vector<int32_t> &bestIdx, std::vector<int32_t> &bestDist) const {
    bestIdx.assign(pts.n(), -1);
    bestDist.assign(pts.n(), 256);
    int rc = vsg_frame_fuse(pKF.handle(), pts.n(), pts.desc.data(), pts.u.data(), pts.v.data(), pts.ur.data(),
                            pts.radius.data(), pts.level.data(), bRight, mvInvLevelSigma2.data(),
                            (int)mvInvLevelSigma2.size(), bestIdx.data(), bestDist.data());
    check(rc, "vsg_frame_fuse");
    return rc;
  }
  // Fuse(pKF, Scw, vpPoints, th, vpReplacePoint)  (ORBmatcher.cc:1331-1446)
  int Fuse(ResidentFrame &pKF, const ProjectedPoints &pts, std::vector<int32_t> &bestIdx,
           std::vector<int32_t> &bestDist) const {
    bestIdx.assign(pts.n(), -1);
    bestDist.assign(pts.n(), 0x7FFFFFFF);
    int rc = vsg_frame_fuse_sim3(pKF.handle(), pts.n(), pts.desc.data(), pts.u.data(), pts.v.data(), pts.radius.data(),
                                 pts.level.data(), bestIdx.data(), bestDist.data());
    check(rc, "vsg_frame_fuse_sim3");
    return rc;
  }

  // SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize)  (ORBmatcher.cc:643-756)
  int SearchForInitialization(ResidentFrame &F1, ResidentFrame &F2, const std::vector<float> &prevX,
                              const std::vector<float> &prevY, int windowSize, std::vector<int32_t> &vnMatches12) const {
    vnMatches12.assign(F1.N(), -1);
    int rc = vsg_frame_search_for_initialization(F1.handle(), F2.handle(), prevX.data(), prevY.data(), windowSize,
                                                 mfNNratio, mbCheckOrientation, vnMatches12.data());
    check(rc, "vsg_frame_search_for_initialization");
    return rc;
  }

  // SearchByBoW(KeyFrame*, Frame&) / (KeyFrame*, KeyFrame*)  (ORBmatcher.cc:226-428, 758-900) on resident descriptors
  int SearchByBoW(ResidentFrame &pKF, const uint8_t *kfValid, const FeatureVectorCSR &kfFeatVec, ResidentFrame &F,
                  const FeatureVectorCSR &fFeatVec, std::vector<int32_t> &matchF) const {
    matchF.assign(F.N(), -1);
    int rc = vsg_frame_search_by_bow_kf_f(pKF.handle(), kfValid, kfFeatVec.node.data(), kfFeatVec.off.data(),
                                          kfFeatVec.idx.data(), kfFeatVec.nodes(), F.handle(), fFeatVec.node.data(),
                                          fFeatVec.off.data(), fFeatVec.idx.data(), fFeatVec.nodes(), mfNNratio,
                                          mbCheckOrientation, matchF.data());
    check(rc, "vsg_frame_search_by_bow_kf_f");
    return rc;
  }
  // The same two with the FeatureVectors both frames keep resident since their ComputeBoW (ORBVocabulary::ComputeBoW):
  // the join runs on the device, only the "has a map point" flags go up
  int SearchByBoW(ResidentFrame &pKF, const uint8_t *kfValid, ResidentFrame &F, std::vector<int32_t> &matchF) const {
    matchF.assign(F.N(), -1);
    int rc = vsg_frame_search_by_bow_kf_f(pKF.handle(), kfValid, nullptr, nullptr, nullptr, 0, F.handle(), nullptr, nullptr,
                                          nullptr, 0, mfNNratio, mbCheckOrientation, matchF.data());
    check(rc, "vsg_frame_search_by_bow_kf_f");
    return rc;
  }
  int SearchByBoW(ResidentFrame &pKF1, const uint8_t *valid1, ResidentFrame &pKF2, const uint8_t *valid2,
                  std::vector<int32_t> &vMatches12) const {
    vMatches12.assign(pKF1.N(), -1);
    int rc = vsg_frame_search_by_bow_kf_kf(pKF1.handle(), valid1, nullptr, nullptr, nullptr, 0, pKF2.handle(), valid2, nullptr,
                                           nullptr, nullptr, 0, mfNNratio, mbCheckOrientation, vMatches12.data());
    check(rc, "vsg_frame_search_by_bow_kf_kf");
    return rc;
  }
  int SearchByBoW(ResidentFrame &pKF1, const uint8_t *valid1, const FeatureVectorCSR &fv1, ResidentFrame &pKF2,
                  const uint8_t *valid2, const FeatureVectorCSR &fv2, std::vector<int32_t> &vMatches12) const {
    vMatches12.assign(pKF1.N(), -1);
    int rc = vsg_frame_search_by_bow_kf_kf(pKF1.handle(), valid1, fv1.node.data(), fv1.off.data(), fv1.idx.data(),
                                           fv1.nodes(), pKF2.handle(), valid2, fv2.node.data(), fv2.off.data(),
                                           fv2.idx.data(), fv2.nodes(), mfNNratio, mbCheckOrientation, vMatches12.data());
    check(rc, "vsg_frame_search_by_bow_kf_kf");
    return rc;
  }

  // SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, vMatchedPairs, bOnlyStereo, bCoarse)  (ORBmatcher.cc:902-1146)
  // with both KeyFrames resident (LocalMapping::CreateNewMapPoints, LocalMapping.cc:389: the current keyframe against
  // each covisible neighbour).  eligible / pairOk as in ORBmatcher::SearchForTriangulation above: the predicate of
  // :1031-1071 is evaluated here for every pair of every shared node and goes up as a bitmask; nullptr_t = all pass.
  template <class Pred>
  int SearchForTriangulation(ResidentFrame &pKF1, const uint8_t *eligible1, const FeatureVectorCSR &fv1,
                             ResidentFrame &pKF2, const uint8_t *eligible2, const FeatureVectorCSR &fv2, Pred pairOk,
                             std::vector<std::pair<size_t, size_t>> &vMatchedPairs) const {
    std::vector<uint32_t> bits(1, 0u);
    std::vector<int32_t> off(1, 0);
    size_t a = 0, b = 0;
    while (a < fv1.node.size() && b < fv2.node.size()) {  // the reference's merge-join, shared nodes in id order
      if (fv1.node[a] == fv2.node[b]) {
        const int n2 = fv2.off[b + 1] - fv2.off[b];
        for (int i1 = fv1.off[a]; i1 < fv1.off[a + 1]; i1++)
          for (int i2 = fv2.off[b]; i2 < fv2.off[b + 1]; i2++) {
            const long long bit = (long long)off.back() + (long long)(i1 - fv1.off[a]) * n2 + (i2 - fv2.off[b]);
            if ((size_t)(bit >> 5) >= bits.size()) bits.resize((size_t)(bit >> 5) + 64, 0u);
            if (eligible1[fv1.idx[i1]] && eligible2[fv2.idx[i2]] && pairOk(fv1.idx[i1], fv2.idx[i2]))
              bits[bit >> 5] |= 1u << (bit & 31);
          }
        off.push_back(off.back() + (fv1.off[a + 1] - fv1.off[a]) * n2);
        a++, b++;
      } else if (fv1.node[a] < fv2.node[b]) {
        a++;
      } else {
        b++;
      }
    }
    bits.resize((size_t)(off.back() >> 5) + 2, 0u);
    std::vector<int32_t> m12(pKF1.N() > 0 ? pKF1.N() : 1, -1);
    int rc = vsg_frame_search_for_triangulation(pKF1.handle(), eligible1, fv1.node.data(), fv1.off.data(), fv1.idx.data(),
                                                fv1.nodes(), pKF2.handle(), eligible2, fv2.node.data(), fv2.off.data(),
                                                fv2.idx.data(), fv2.nodes(), bits.data(), off.data(), mbCheckOrientation,
                                                m12.data());
    check(rc, "vsg_frame_search_for_triangulation");
    vMatchedPairs.clear();
    for (int i = 0; i < pKF1.N(); i++)
      if (m12[i] >= 0) vMatchedPairs.emplace_back((size_t)i, (size_t)m12[i]);  // :1131-1141: pairs in idx1 order
    return rc;
  }

 protected:
  float mfNNratio;
  bool mbCheckOrientation;
};

// Asynchronous operator() batches over the handle's pipeline slots (vsg_orb_submit_batch / vsg_orb_wait): H2D of
// batch k+1 beside the kernels of batch k beside the export of batch k-1.  Buffers from vsg_host_alloc (hipHostMalloc) are
// read and written by the device in place; any other memory is staged (include/vsg_orb.h, "Pinned caller memory").
inline int SubmitBatch(const ORBextractor &ex, const uint8_t *gray, int nframes, size_t frameStride, int rows, int cols,
                       int stride, const std::vector<int> &vLappingArea, vsg_keypoint *kps, uint8_t *desc,
                       int capacity) {
  int t = vsg_orb_submit_batch(ex.handle(), gray, nframes, frameStride, rows, cols, stride, vLappingArea.at(0),
                               vLappingArea.at(1), kps, desc, capacity);
  check(t, "vsg_orb_submit_batch");
  return t;
}
inline void WaitBatch(const ORBextractor &ex, int ticket, std::vector<int> &n, std::vector<int> &monoIndex) {
  check(vsg_orb_wait(ex.handle(), ticket, n.data(), monoIndex.data()), "vsg_orb_wait");
}

}  // namespace vsg
