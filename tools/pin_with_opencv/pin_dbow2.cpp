// pin_dbow2.cpp -- runs the REFERENCE's vendored DBoW2 (orb_slam3/Thirdparty/DBoW2/DBoW2/{TemplatedVocabulary.h,
// FORB.cpp, BowVector.cpp, FeatureVector.cpp, ScoringObject.cpp}, compiled from the reference checkout; needs OpenCV
// core + Boost.Serialization headers only) on seeded inputs and dumps what bow_oracle.cpp / match_oracle.cpp restate:
//   * ORBVocabulary::loadFromBinFile (TemplatedVocabulary.h:1478-1552) of the synthetic k = 10, L = 6 vocabulary of
//     include/vsg_synth.h (the shape of ORBvoc.txt.bin, launch/tum1_rgbd.launch:18) and of small trees with the other
//     scoring / weighting types
//   * transform(features, BowVector, FeatureVector, levelsup) (:1139-1212) as Frame::ComputeBoW calls it (levelsup 4,
//     Frame.cc:887) for seeded descriptors incl. exact copies of node descriptors (distance-0 ties)
//   * FORB::distance (FORB.cpp:81-101) on seeded pairs -- the same arithmetic as ORBmatcher::DescriptorDistance
//     (ORBmatcher.cc:2047-2063), which cannot be compiled without the whole map model
//   * the cv::KeyPoint / cv::Mat layout facts include/vsg_orb_adaptor.hpp (VSG_WITH_OPENCV) relies on
// Output: the named-array stream of pin_dump.cpp; pack_npz.py turns it into tests/golden/dbow2_v1.npz.
// Needs OpenCV and Boost: NOT built or run in the authoring image.
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <opencv2/core/core.hpp>
#include <string>
#include <vector>

#include "FORB.h"
#include "TemplatedVocabulary.h"
#include "vsg_synth.h"

typedef DBoW2::TemplatedVocabulary<DBoW2::FORB::TDescriptor, DBoW2::FORB> ORBVocabulary;  // ORBVocabulary.h:29

static FILE *g_out = nullptr;
static void put(const std::string &name, int dtype, std::vector<uint32_t> dims, const void *data) {
  static const int esz[4] = {1, 4, 4, 8};  // 0 = u8, 1 = i32, 2 = f32, 3 = f64
  uint32_t nl = (uint32_t)name.size(), nd = (uint32_t)dims.size();
  size_t n = esz[dtype];
  for (uint32_t d : dims) n *= d;
  fwrite(&nl, 4, 1, g_out), fwrite(name.data(), 1, nl, g_out);
  uint8_t dt = (uint8_t)dtype;
  fwrite(&dt, 1, 1, g_out), fwrite(&nd, 4, 1, g_out), fwrite(dims.data(), 4, nd, g_out);
  if (n) fwrite(data, 1, n, g_out);
}

// synth.random_descriptors(n, seed): n x 32 bytes from SplitMix64(0x5EED0000 ^ seed << 8 ^ 0xD35C)
static std::vector<uint8_t> random_descriptors(int n, uint32_t seed) {
  std::vector<uint8_t> d((size_t)n * 32);
  const uint64_t s = VSG_SYNTH_SEED_BASE ^ ((uint64_t)seed << 8) ^ 0xD35Cull;
  for (int i = 0; i < n * 4; i++) {
    const uint64_t r = vsg_synth_splitmix64(s, (uint64_t)i + 1);
    memcpy(&d[(size_t)i * 8], &r, 8);
  }
  return d;
}

struct VocCase {
  const char *name;
  int k, L, seed, scoring, weighting, levelsup, ndesc;
};
static const VocCase kVocs[] = {{"k10_L6", 10, 6, 7, 0, 0, 4, 1200},   // the reference's shape and call
                                {"k10_L3_l2", 10, 3, 31, 1, 1, 2, 400},   // L2 scoring, TF
                                {"k8_L3_idf", 8, 3, 83, 2, 2, 1, 400},    // CHI_SQUARE, IDF
                                {"k6_L4_bin", 6, 4, 64, 5, 3, 5, 400}};   // DOT_PRODUCT (no normalisation), BINARY; levelsup > L

int main(int argc, char **argv) {
  if (argc < 3) return fprintf(stderr, "usage: pin_dbow2 <out.bin> <scratch dir>\n"), 2;
  g_out = fopen(argv[1], "wb");
  if (!g_out) return 2;
  const std::string ver = CV_VERSION;
  put("opencv_version", 0, {(uint32_t)ver.size()}, ver.data());

  for (const VocCase &c : kVocs) {
    std::vector<uint8_t> blob(vsg_synth_vocabulary(c.k, c.L, (uint32_t)c.seed, c.scoring, c.weighting, 0.02, nullptr, 0));
    vsg_synth_vocabulary(c.k, c.L, (uint32_t)c.seed, c.scoring, c.weighting, 0.02, blob.data(), blob.size());
    const std::string path = std::string(argv[2]) + "/" + c.name + ".bin";
    FILE *f = fopen(path.c_str(), "wb");
    if (!f || fwrite(blob.data(), 1, blob.size(), f) != blob.size()) return 3;
    fclose(f);
    ORBVocabulary voc;
    if (!voc.loadFromBinFile(path)) return 4;
    // descriptors: seeded random rows; every 7th row is an exact copy of a node's descriptor (the record of node
    // 1 + 37 i in the file), so that distance-0 minima and ties between equal children occur
    std::vector<uint8_t> d = random_descriptors(c.ndesc, 100 + (uint32_t)c.seed);
    const size_t nnodes = (blob.size() - 16) / 45;
    for (int i = 0; i < c.ndesc; i += 7) memcpy(&d[(size_t)i * 32], &blob[16 + ((size_t)(37 * i) % nnodes) * 45 + 5], 32);
    std::vector<cv::Mat> feats;
    for (int i = 0; i < c.ndesc; i++) feats.push_back(cv::Mat(1, 32, CV_8U, &d[(size_t)i * 32]).clone());
    DBoW2::BowVector bow;
    DBoW2::FeatureVector fv;
    voc.transform(feats, bow, fv, c.levelsup);
    const std::string p = std::string(c.name) + "/";
    const int32_t params[8] = {c.k, c.L, c.seed, c.scoring, c.weighting, c.levelsup, c.ndesc, (int32_t)voc.size()};
    put(p + "params", 1, {8}, params);
    put(p + "desc", 0, {(uint32_t)c.ndesc, 32}, d.data());
    std::vector<int32_t> ids, fnode, foff(1, 0), fidx;
    std::vector<double> vals;
    for (const auto &kv : bow) ids.push_back((int32_t)kv.first), vals.push_back(kv.second);
    for (const auto &kv : fv) {
      fnode.push_back((int32_t)kv.first);
      for (unsigned i : kv.second) fidx.push_back((int32_t)i);
      foff.push_back((int32_t)fidx.size());
    }
    put(p + "bow_ids", 1, {(uint32_t)ids.size()}, ids.data());
    put(p + "bow_vals", 3, {(uint32_t)vals.size()}, vals.data());
    put(p + "fv_node", 1, {(uint32_t)fnode.size()}, fnode.data());
    put(p + "fv_off", 1, {(uint32_t)foff.size()}, foff.data());
    put(p + "fv_idx", 1, {(uint32_t)fidx.size()}, fidx.data());
    // per-feature word ids through the public single-feature overload (TemplatedVocabulary.h:127)
    std::vector<int32_t> words;
    for (const cv::Mat &m : feats) words.push_back((int32_t)voc.transform(m));
    put(p + "word", 1, {(uint32_t)words.size()}, words.data());
    remove(path.c_str());
  }
  {  // FORB::distance == ORBmatcher::DescriptorDistance arithmetic
    const int n = 512;
    std::vector<uint8_t> a = random_descriptors(n, 901), b = random_descriptors(n, 902);
    memset(&a[0], 0, 32), memset(&b[0], 0xFF, 32);          // distance 256
    memcpy(&b[32], &a[32], 32);                             // distance 0
    std::vector<int32_t> dist;
    for (int i = 0; i < n; i++)
      dist.push_back(DBoW2::FORB::distance(cv::Mat(1, 32, CV_8U, &a[(size_t)i * 32]), cv::Mat(1, 32, CV_8U, &b[(size_t)i * 32])));
    put("forb/a", 0, {(uint32_t)n, 32}, a.data());
    put("forb/b", 0, {(uint32_t)n, 32}, b.data());
    put("forb/distance", 1, {(uint32_t)n}, dist.data());
  }
  {  // layout facts the VSG_WITH_OPENCV branch of include/vsg_orb_adaptor.hpp memcpy's through
    cv::Mat desc(5, 32, CV_8U);
    const int32_t facts[12] = {(int32_t)sizeof(cv::KeyPoint),
                               (int32_t)offsetof(cv::KeyPoint, pt),
                               (int32_t)offsetof(cv::KeyPoint, size),
                               (int32_t)offsetof(cv::KeyPoint, angle),
                               (int32_t)offsetof(cv::KeyPoint, response),
                               (int32_t)offsetof(cv::KeyPoint, octave),
                               (int32_t)offsetof(cv::KeyPoint, class_id),
                               desc.isContinuous() ? 1 : 0,
                               (int32_t)desc.step[0],
                               desc.type() == CV_8UC1 ? 1 : 0,
                               (int32_t)sizeof(cv::Point2f),
                               cv::KeyPoint().class_id};
    put("layout/facts", 1, {12}, facts);
  }
  fclose(g_out);
  printf("wrote %s (OpenCV %s)\n", argv[1], CV_VERSION);
  return 0;
}
