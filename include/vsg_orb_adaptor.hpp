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
#include <stdexcept>
#include <string>
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
  // Refresh mvImagePyramid[level] (bordered buffer + ROI view, as ComputePyramid leaves it) on demand;
  // only the stereo matcher reads it.
  void DownloadPyramid() {
    for (int l = 0; l < nlevels_; l++) {
      int w = 0, h = 0;
      vsg_orb_level_size(h_, l, &w, &h);
      cv::Mat temp(h + 38, w + 38, CV_8UC1);
      vsg_orb_copy_pyramid_level(h_, 0, l, 1, temp.data, (int)temp.step);
      mvImagePyramid[l] = temp(cv::Rect(19, 19, w, h));
    }
  }
  std::vector<cv::Mat> mvImagePyramid;  // ORBextractor.h:93
#endif

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
};

// static int ORBmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b)  (ORBmatcher.h:40)
inline int DescriptorDistance(const uint8_t *a, const uint8_t *b, int device = 0) {
  const int32_t zero = 0;
  int32_t d = -1;
  int rc = vsg_hamming_pairs(device, a, 1, b, 1, &zero, &zero, 1, &d);
  if (rc != VSG_OK) throw std::runtime_error("vsg_hamming_pairs failed");
  return d;
}

}  // namespace vsg
