// DECLARATION-ONLY stand-in for the handful of OpenCV names include/vsg_orb_adaptor.hpp touches under
// VSG_WITH_OPENCV.  It exists for ONE purpose: `g++ -fsyntax-only` of that branch in an image without OpenCV
// (tests/test_abi.py::test_adaptor_opencv_branch_compiles), so that a typo there is caught before a maintainer with
// the real OpenCV 4.2 sees it.  It is NOT OpenCV, defines no behaviour, is never linked or run, and is not used to
// build the reference or the oracle.  Signatures follow opencv2/core (4.2): types.hpp (KeyPoint, Rect), mat.hpp
// (Mat, InputArray, OutputArray), base.hpp (Error codes, CV_Assert / CV_Error).
#pragma once
#include <cstddef>
#include <string>
#include <vector>
typedef unsigned char uchar;
#define CV_8U 0
#define CV_8UC1 0
namespace cv {
struct Point2f { float x, y; };
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };
struct Rect { int x, y, width, height; Rect(int, int, int, int); };
class Mat;
class _InputArray { public: bool empty() const; Mat getMat(int idx = -1) const; };
class _OutputArray : public _InputArray { public: void release() const; };
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;
struct MatStep { operator size_t() const; };
class Mat {
 public:
  Mat();
  Mat(int rows, int cols, int type);
  Mat(int rows, int cols, int type, void *data, size_t step = 0);
  int type() const;
  Mat rowRange(int startrow, int endrow) const;
  void copyTo(OutputArray m) const;
  Mat operator()(const Rect &roi) const;
  int rows, cols;
  uchar *data;
  MatStep step;
};
namespace Error { enum Code { StsError = -2, StsBadArg = -5 }; }
[[noreturn]] void error(int code, const std::string &msg, const char *func, const char *file, int line);
}  // namespace cv
#define CV_Error(code, msg) cv::error(code, msg, __func__, __FILE__, __LINE__)
#define CV_Assert(expr) do { if (!(expr)) cv::error(cv::Error::StsError, #expr, __func__, __FILE__, __LINE__); } while (0)
