// pin_dump.cpp -- runs the REFERENCE's VS_GRAPHS::ORBextractor (orb_slam3/src/ORBextractor.cc, compiled from the
// reference checkout, linked against a real OpenCV 4.2) on the seeded synthetic frames of
// tests/golden/make_golden.py and dumps what the CPU oracle restates:
//   * mvImagePyramid[l] WITH its 19 px border                                   (ORBextractor.cc:1171-1195)
//   * the blurred level the descriptors are computed on: clone + GaussianBlur   (:1129-1130)
//   * keypoints (all 7 cv::KeyPoint fields) and descriptors, monoIndex          (:1083-1169)
//   * the same for one frame of EVERY content class of synth.CONTENT_CLASSES / vsg_synth_content_frame (value noise,
//     1 / 2 px checkerboards, gratings, defocus, saturation, ramps, salt and pepper): they stress cv::resize, the blur
//     and the minThFAST retry differently from rectangles + noise; cases "content_<kind>"
//   * the same for REAL PHOTOGRAPHS (round 6): frames of synth.PHOTO_CLASSES exported by export_photo_frames.py from the
//     committed gray planes (tests/golden/photos_v1.npz) into a flat file given as the optional second argument: ~10 k
//     level-0 corners per frame, iniThFAST / minThFAST / empty cells mixed inside one frame; cases "photo_<index>_<width>"
// and the version-sensitive OpenCV pieces as observations:
//   * cv::getGaussianKernel(7, 2) and the 8-bit GaussianBlur response to a one-column line image (= the 8.8 taps)
//   * cv::cvtColor(RGB2GRAY / BGR2GRAY) on 4096 seeded colours (= the fixed-point gray coefficients)
//   * cv::resize INTER_LINEAR, cv::FAST(20, nonmax) on a seeded frame, cv::fastAtan2 on a grid of arguments
//   * cv::undistortPoints(pts, K, D, cv::Mat(), K) -- the call of Frame::UndistortKeyPoints / ComputeImageBounds
//     (Frame.cc:891-955) -- for the TUM1 and RealSense D435i cameras on seeded points and on the image corners
// Output: one binary stream of named arrays (see put()); tools/pin_with_opencv/pack_npz.py turns it into
// tests/golden/opencv42_v1.npz.  Needs OpenCV: it is NOT built or run in the authoring image.
#include <cstdint>
#include <cstdio>
#include <opencv2/opencv.hpp>
#include <string>
#include <vector>

#include "ORBextractor.h"
#include "vsg_synth.h"

static FILE *g_out = nullptr;
// record: u32 name_len, name, u8 dtype (0 = u8, 1 = i32, 2 = f32, 3 = f64), u32 ndim, u32 dims[ndim], raw data
static void put(const std::string &name, int dtype, std::vector<uint32_t> dims, const void *data) {
  static const int esz[4] = {1, 4, 4, 8};
  uint32_t nl = (uint32_t)name.size(), nd = (uint32_t)dims.size();
  size_t n = esz[dtype];
  for (uint32_t d : dims) n *= d;
  fwrite(&nl, 4, 1, g_out), fwrite(name.data(), 1, nl, g_out);
  uint8_t dt = (uint8_t)dtype;
  fwrite(&dt, 1, 1, g_out), fwrite(&nd, 4, 1, g_out), fwrite(dims.data(), 4, nd, g_out);
  if (n) fwrite(data, 1, n, g_out);
}
static void put_mat_u8(const std::string &name, const cv::Mat &m) {
  cv::Mat c = m.isContinuous() ? m : m.clone();
  put(name, 0, {(uint32_t)c.rows, (uint32_t)c.cols}, c.data);
}

struct Case {
  const char *name;
  int w, h, seed, div, nf, nl, lap0, lap1;
};
// the cases of tests/golden/make_golden.py (synth.frame(w, h, seed, amplitude_div) = sequence frame t = 0)
static const Case kCases[] = {{"qvga_rgbd", 320, 240, 42, 1, 500, 4, 0, 0},
                              {"qvga_lowcontrast", 320, 240, 43, 8, 500, 4, 0, 0},
                              {"qvga_mono_lapping", 320, 240, 44, 1, 500, 4, 0, 1000},
                              {"qvga_partial_lapping", 320, 240, 42, 1, 300, 3, 100, 220},
                              {"vga_c2", 640, 480, 7, 1, 1000, 8, 0, 0}};

int main(int argc, char **argv) {
  if (argc < 2) return fprintf(stderr, "usage: pin_dump <out.bin> [photo_frames.bin]\n"), 2;
  g_out = fopen(argv[1], "wb");
  if (!g_out) return 2;
  const std::string ver = CV_VERSION;
  put("opencv_version", 0, {(uint32_t)ver.size()}, ver.data());

  for (const Case &c : kCases) {
    cv::Mat img(c.h, c.w, CV_8UC1);
    if (vsg_synth_frame(c.w, c.h, (uint32_t)c.seed, c.div, 6, img.data, img.step)) return 3;
    VS_GRAPHS::ORBextractor ex(c.nf, 1.2f, c.nl, 20, 7);
    std::vector<cv::KeyPoint> kps;
    cv::Mat desc;
    std::vector<int> lap{c.lap0, c.lap1};
    const int mono = ex(img, cv::Mat(), kps, desc, lap);
    const std::string p = std::string(c.name) + "/";
    const int32_t params[8] = {c.w, c.h, c.seed, c.div, c.nf, c.nl, c.lap0, c.lap1};
    put(p + "params", 1, {8}, params);
    const int32_t mono32 = mono;
    put(p + "mono", 1, {1}, &mono32);
    static_assert(sizeof(cv::KeyPoint) == 28, "cv::KeyPoint layout");
    put(p + "kps", 0, {(uint32_t)kps.size(), 28}, kps.data());
    if (!desc.empty()) put_mat_u8(p + "desc", desc); else put(p + "desc", 0, {0, 32}, nullptr);
    for (int l = 0; l < c.nl; l++) {
      cv::Mat roi = ex.mvImagePyramid[l], full = roi;
      full.adjustROI(19, 19, 19, 19);  // EDGE_THRESHOLD: the bordered buffer the ROI lives in (:1177-1181)
      put_mat_u8(p + "pyr" + std::to_string(l), full);
      cv::Mat work = roi.clone();  // exactly :1129-1130
      cv::GaussianBlur(work, work, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
      put_mat_u8(p + "blur" + std::to_string(l), work);
    }
  }

  // one frame per content class (kinds in the order of synth.CONTENT_CLASSES): 320x240 / 500 / 4 levels, and the two
  // natural-image stand-ins plus the worst case for FAST once more at the headline geometry (640x480 / 1000 / 8 levels)
  {
    struct CCase {
      int kind, w, h, seq, t, nf, nl;
    };
    std::vector<CCase> cc;
    for (int kind = 0; kind < VSG_CONTENT_COUNT; kind++) cc.push_back({kind, 320, 240, 5000, 1, 500, 4});
    for (int kind : {VSG_CONTENT_VALUE_NOISE, VSG_CONTENT_DEFOCUS, VSG_CONTENT_GRATING}) cc.push_back({kind, 640, 480, 5000, 3, 1000, 8});
    for (const CCase &c : cc) {
      cv::Mat img(c.h, c.w, CV_8UC1);
      if (vsg_synth_content_frame(c.kind, c.w, c.h, (uint32_t)c.seq, c.t, img.data, img.step)) return 3;
      VS_GRAPHS::ORBextractor ex(c.nf, 1.2f, c.nl, 20, 7);
      std::vector<cv::KeyPoint> kps;
      cv::Mat desc;
      std::vector<int> lap{0, 0};
      const int mono = ex(img, cv::Mat(), kps, desc, lap);
      const std::string p = "content_" + std::to_string(c.kind) + "_" + std::to_string(c.w) + "/";
      const int32_t params[8] = {c.w, c.h, c.seq, c.t, c.nf, c.nl, c.kind, 0};
      put(p + "content_params", 1, {8}, params);
      const int32_t mono32 = mono;
      put(p + "mono", 1, {1}, &mono32);
      put(p + "kps", 0, {(uint32_t)kps.size(), 28}, kps.data());
      if (!desc.empty()) put_mat_u8(p + "desc", desc); else put(p + "desc", 0, {0, 32}, nullptr);
      for (int l = 0; l < c.nl; l++) {
        cv::Mat roi = ex.mvImagePyramid[l], full = roi;
        full.adjustROI(19, 19, 19, 19);
        put_mat_u8(p + "pyr" + std::to_string(l), full);
        cv::Mat work = roi.clone();
        cv::GaussianBlur(work, work, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
        put_mat_u8(p + "blur" + std::to_string(l), work);
      }
    }
  }

  if (argc > 2) {  // photographs: records of 8 x i32 {w, h, seq, t, nfeatures, nlevels, photo index, 0} + w * h gray bytes
    FILE *pf = fopen(argv[2], "rb");
    if (!pf) return fprintf(stderr, "cannot open %s\n", argv[2]), 2;
    int32_t params[8];
    while (fread(params, 4, 8, pf) == 8) {
      const int w = params[0], h = params[1], nf = params[4], nl = params[5];
      cv::Mat img(h, w, CV_8UC1);
      if (fread(img.data, 1, (size_t)w * h, pf) != (size_t)w * h) return 3;
      VS_GRAPHS::ORBextractor ex(nf, 1.2f, nl, 20, 7);
      std::vector<cv::KeyPoint> kps;
      cv::Mat desc;
      std::vector<int> lap{0, 0};
      const int32_t mono32 = ex(img, cv::Mat(), kps, desc, lap);
      const std::string p = "photo_" + std::to_string(params[6]) + "_" + std::to_string(w) + "/";
      put(p + "photo_params", 1, {8}, params);
      put(p + "mono", 1, {1}, &mono32);
      put(p + "kps", 0, {(uint32_t)kps.size(), 28}, kps.data());
      if (!desc.empty()) put_mat_u8(p + "desc", desc); else put(p + "desc", 0, {0, 32}, nullptr);
      for (int l = 0; l < nl; l++) {
        cv::Mat roi = ex.mvImagePyramid[l], full = roi;
        full.adjustROI(19, 19, 19, 19);
        put_mat_u8(p + "pyr" + std::to_string(l), full);
        cv::Mat work = roi.clone();
        cv::GaussianBlur(work, work, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
        put_mat_u8(p + "blur" + std::to_string(l), work);
      }
    }
    fclose(pf);
  }

  {  // Gaussian taps: the double kernel, and the 8-bit blur of a line image (every row = 255 at one column)
    cv::Mat k = cv::getGaussianKernel(7, 2, CV_64F);
    put("gauss/kernel_f64", 3, {7}, k.data);
    cv::Mat line(64, 64, CV_8UC1, cv::Scalar(0)), out;
    line.col(32).setTo(255);
    cv::GaussianBlur(line, out, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
    put_mat_u8("gauss/line_response", out.row(20).colRange(29, 36));  // (sum(taps) * tap * 255 + 32768) >> 16
    cv::Mat flat(64, 64, CV_8UC1, cv::Scalar(200)), fo;
    cv::GaussianBlur(flat, fo, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
    put_mat_u8("gauss/flat200_response", fo.row(20).colRange(29, 36));     // 200 for a sum of 256, 201 for 257
  }
  {  // gray conversion on seeded colours
    cv::Mat rgb(64, 64, CV_8UC3), rgba(64, 64, CV_8UC4), g1, g2, g3, g4;
    for (int i = 0; i < 64 * 64; i++) {
      const uint64_t r = vsg_synth_splitmix64(0xC0105EEDull, (uint64_t)i + 1);
      rgb.data[3 * i] = (uint8_t)r, rgb.data[3 * i + 1] = (uint8_t)(r >> 8), rgb.data[3 * i + 2] = (uint8_t)(r >> 16);
      for (int ch = 0; ch < 4; ch++) rgba.data[4 * i + ch] = (uint8_t)(r >> (8 * ch));
    }
    cv::cvtColor(rgb, g1, cv::COLOR_RGB2GRAY), cv::cvtColor(rgb, g2, cv::COLOR_BGR2GRAY);
    cv::cvtColor(rgba, g3, cv::COLOR_RGBA2GRAY), cv::cvtColor(rgba, g4, cv::COLOR_BGRA2GRAY);
    put_mat_u8("gray/rgb", rgb.reshape(1, 64 * 64)), put_mat_u8("gray/rgba", rgba.reshape(1, 64 * 64));
    put_mat_u8("gray/rgb2gray", g1), put_mat_u8("gray/bgr2gray", g2), put_mat_u8("gray/rgba2gray", g3);
    put_mat_u8("gray/bgra2gray", g4);
  }
  {  // stand-alone OpenCV pieces on one seeded frame
    cv::Mat img(240, 320, CV_8UC1), small;
    vsg_synth_frame(320, 240, 42, 1, 6, img.data, img.step);
    cv::resize(img, small, cv::Size(267, 200), 0, 0, cv::INTER_LINEAR);
    put_mat_u8("cv/resize_267x200", small);
    for (int th : {20, 7}) {
      std::vector<cv::KeyPoint> k;
      cv::FAST(img, k, th, true);
      std::vector<int32_t> flat;
      for (auto &kp : k) flat.push_back((int)kp.pt.x), flat.push_back((int)kp.pt.y), flat.push_back((int)kp.response);
      put("cv/fast" + std::to_string(th), 1, {(uint32_t)k.size(), 3}, flat.data());
    }
    std::vector<float> args, vals;
    for (int y = -40; y <= 40; y += 3)
      for (int x = -40; x <= 40; x += 3) {
        args.push_back((float)y * 123.5f), args.push_back((float)x * 77.25f);
        vals.push_back(cv::fastAtan2((float)y * 123.5f, (float)x * 77.25f));
      }
    put("cv/fastatan2_args", 2, {(uint32_t)vals.size(), 2}, args.data());
    put("cv/fastatan2", 2, {(uint32_t)vals.size()}, vals.data());
  }
  {  // cv::undistortPoints exactly as Frame.cc:906-909 / :938-940 call it: CV_32FC2 in place, K and D as CV_32F, R empty, P = K
    struct Cam {
      const char *name;
      float fx, fy, cx, cy;
      std::vector<float> d;
    };
    const Cam cams[] = {{"tum1", 517.306408f, 516.469215f, 318.643040f, 255.313989f,
                         {0.262383f, -0.953104f, -0.005358f, 0.002628f, 1.163314f}},   // config/RGB-D/TUM1.yaml:11-20
                        {"d435i", 6.165911254882812e+02f, 6.166796264648438e+02f, 3.242193603515625e+02f, 2.3942701721191406e+02f,
                         {1.25323e-01f, -2.51452e-01f, 7.12e-04f, 6.217e-03f}}};       // RealSense_D435i.yaml:11-19
    for (const Cam &c : cams) {
      cv::Mat K = cv::Mat::eye(3, 3, CV_32F);
      K.at<float>(0, 0) = c.fx, K.at<float>(1, 1) = c.fy, K.at<float>(0, 2) = c.cx, K.at<float>(1, 2) = c.cy;
      cv::Mat D((int)c.d.size(), 1, CV_32F);
      for (size_t i = 0; i < c.d.size(); i++) D.at<float>((int)i) = c.d[i];
      const int N = 4096 + 4;
      cv::Mat mat(N, 2, CV_32F);
      for (int i = 0; i < 4096; i++) {  // seeded points over the image and a 20 px margin around it, quarter-pixel steps
        const uint64_t r = vsg_synth_splitmix64(0xD157ull, (uint64_t)i + 1);
        mat.at<float>(i, 0) = (float)((int)(r % 2720) - 80) * 0.25f;
        mat.at<float>(i, 1) = (float)((int)((r >> 20) % 2080) - 80) * 0.25f;
      }
      const float corners[4][2] = {{0, 0}, {640, 0}, {0, 480}, {640, 480}};  // ComputeImageBounds' four points
      for (int i = 0; i < 4; i++) mat.at<float>(4096 + i, 0) = corners[i][0], mat.at<float>(4096 + i, 1) = corners[i][1];
      cv::Mat in = mat.clone();
      mat = mat.reshape(2);
      cv::undistortPoints(mat, mat, K, D, cv::Mat(), K);
      mat = mat.reshape(1);
      const float k4[4] = {c.fx, c.fy, c.cx, c.cy};
      put(std::string("undistort/") + c.name + "/K4", 2, {4}, k4);
      put(std::string("undistort/") + c.name + "/dist", 2, {(uint32_t)c.d.size()}, c.d.data());
      put(std::string("undistort/") + c.name + "/in", 2, {(uint32_t)N, 2}, in.data);
      put(std::string("undistort/") + c.name + "/out", 2, {(uint32_t)N, 2}, mat.data);
    }
  }
  fclose(g_out);
  printf("wrote %s (OpenCV %s)\n", argv[1], CV_VERSION);
  return 0;
}
