// End-to-end exercise of include/vsg_orb_adaptor.hpp from plain C++ (no Python, no OpenCV): two synthetic frames
// -> vsg::ORBextractor -> vsg::FrameGrid -> vsg::ORBmatcher::SearchWindow / SearchByBoW (with vsg::ORBVocabulary).
// Every output is dumped to a flat binary file that tests/test_gpu_adaptor.py compares with the CPU oracle.
//   usage: adaptor_check <vocab.bin> <out.bin>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>

#include "vsg_orb_adaptor.hpp"
#include "vsg_synth.h"

template <class T>
static void dump(std::ofstream &f, const std::vector<T> &v) {
  int32_t n = (int32_t)v.size();
  f.write((const char *)&n, 4);
  if (n) f.write((const char *)v.data(), sizeof(T) * v.size());
}

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  try {
    const int W = 640, H = 480;
    std::vector<uint8_t> img[2] = {std::vector<uint8_t>(W * H), std::vector<uint8_t>(W * H)};
    for (int t = 0; t < 2; ++t)
      if (vsg_synth_sequence_frame(W, H, 5, t, 1, 6, img[t].data(), W)) return 2;

    vsg::ORBextractor ex(1000, 1.2f, 8, 20, 7);
    std::vector<vsg_keypoint> kps[2];
    std::vector<uint8_t> desc[2];
    std::vector<int> lap{0, 0};
    int mono[2];
    for (int t = 0; t < 2; ++t) mono[t] = ex(img[t].data(), H, W, W, kps[t], desc[t], lap);

    // window search: frame-0 keypoints looked up in frame 1's grid at the known (3,2) px shift, r = 15 * scale
    std::vector<float> angle[2], qx, qy, qr;
    std::vector<int32_t> octave[2], lo, hi;
    for (int t = 0; t < 2; ++t)
      for (auto &k : kps[t]) angle[t].push_back(k.angle), octave[t].push_back(k.octave);
    const std::vector<float> scale = ex.GetScaleFactors();
    for (auto &k : kps[0]) {
      qx.push_back(k.x - 3.0f), qy.push_back(k.y - 2.0f), qr.push_back(15.0f * scale[k.octave]);
      lo.push_back(k.octave - 1), hi.push_back(k.octave + 1);
    }
    vsg::FrameGrid grid(kps[1].data(), (int)kps[1].size(), 0.f, 0.f, (float)W, (float)H);
    vsg::Candidates cand = grid.GetFeaturesInArea(qx.data(), qy.data(), qr.data(), lo.data(), hi.data(), (int)qx.size());

    vsg::ORBmatcher matcher(0.7f, true);
    vsg::FeatureView q{desc[0].data(), angle[0].data(), octave[0].data(), (int)kps[0].size()};
    vsg::FeatureView tr{desc[1].data(), angle[1].data(), octave[1].data(), (int)kps[1].size()};
    std::vector<int32_t> bestIdx, bestDist, trainMatch;
    std::vector<uint8_t> trainBlocked;
    std::vector<uint8_t> qBlocks(q.n, 1);
    int nwin = matcher.SearchWindow(q, qBlocks.data(), cand, tr, vsg::ORBmatcher::TH_HIGH, bestIdx, bestDist,
                                    &trainBlocked, &trainMatch);

    // BoW: vocabulary image from disk, transform both frames, SearchByBoW(KF = frame 0, F = frame 1)
    std::ifstream vf(argv[1], std::ios::binary);
    std::vector<uint8_t> blob((std::istreambuf_iterator<char>(vf)), std::istreambuf_iterator<char>());
    vsg::ORBVocabulary voc(blob.data(), blob.size());
    std::map<unsigned, double> bow[2];
    vsg::FeatureVectorCSR fv[2];
    for (int t = 0; t < 2; ++t) voc.transform(desc[t].data(), (int)kps[t].size(), bow[t], fv[t], 2);
    // round trip through the reference's container type
    std::map<unsigned, std::vector<unsigned>> featVecMap;
    for (int i = 0; i < fv[0].nodes(); ++i)
      featVecMap[(unsigned)fv[0].node[i]].assign(fv[0].idx.begin() + fv[0].off[i], fv[0].idx.begin() + fv[0].off[i + 1]);
    vsg::FeatureVectorCSR fv0(featVecMap);
    std::vector<uint8_t> kfValid(q.n, 1);
    for (int i = 0; i < q.n; i += 7) kfValid[i] = 0;
    std::vector<int32_t> matchF;
    int nbow = matcher.SearchByBoW(q, kfValid.data(), fv0, tr, fv[1], matchF);

    std::vector<int32_t> init12;
    int ninit = matcher.SearchForInitialization(q, cand, tr, init12);

    // SearchForTriangulation with an arbitrary pure pair predicate standing in for the epipolar test
    std::vector<uint8_t> elig0(q.n, 1), elig1(tr.n, 1);
    for (int i = 0; i < q.n; i += 5) elig0[i] = 0;
    for (int i = 0; i < tr.n; i += 9) elig1[i] = 0;
    std::vector<std::pair<size_t, size_t>> tri;
    int ntri = matcher.SearchForTriangulation(q, elig0.data(), fv0, tr, elig1.data(), fv[1],
                                              [](int i1, int i2) { return (i1 * 7 + i2 * 3) % 5 != 0; }, tri);
    std::vector<int32_t> tri_flat;
    for (auto &pr : tri) tri_flat.push_back((int32_t)pr.first), tri_flat.push_back((int32_t)pr.second);

    const int d01 = vsg::ORBmatcher::DescriptorDistance(desc[0].data(), desc[1].data());

    // ---- the resident path: frame 1 straight out of the extractor (it ran last), frame 0 uploaded from the host
    vsg::ResidentFrame R0(ex.capacity(H, W)), R1(ex.capacity(H, W));
    R1.FromExtractor(ex, kps[1], 0.f, 0.f, (float)W, (float)H);
    R0.Upload(kps[0].data(), desc[0].data(), nullptr, (int)kps[0].size(), -1, 0.f, 0.f, (float)W, (float)H);
    vsg::ProjectedPoints P;
    for (auto &k : kps[0]) {
      P.u.push_back(k.x - 3.0f), P.v.push_back(k.y - 2.0f), P.level.push_back(k.octave), P.angle.push_back(k.angle);
      P.radius.push_back(10.0f * scale[k.octave]), P.observed.push_back(1);
    }
    P.desc = desc[0];
    P.ur = P.u;
    vsg::ResidentMatcher rm(0.7f, true);
    std::vector<uint8_t> rBlocked;
    std::vector<int32_t> rLast, rSim3(kps[1].size(), -1), rFuseIdx, rFuseDist, rInit;
    const int nlast = rm.SearchByProjection(R1, P, 15.0f, 0, scale, rBlocked, rLast);
    const int nsim3 = rm.SearchByProjection(R1, P, 1.0f, rSim3);
    const int nfuse = rm.Fuse(R1, P, false, ex.GetInverseScaleSigmaSquares(), rFuseIdx, rFuseDist);
    std::vector<float> px, py;
    for (auto &k : kps[0]) px.push_back(k.x), py.push_back(k.y);
    const int nrinit = rm.SearchForInitialization(R0, R1, px, py, 100, rInit);
    // the same triangulation search on the two resident frames: identical pairs expected
    std::vector<std::pair<size_t, size_t>> rtri;
    const int nrtri = rm.SearchForTriangulation(R0, elig0.data(), fv0, R1, elig1.data(), fv[1],
                                                [](int i1, int i2) { return (i1 * 7 + i2 * 3) % 5 != 0; }, rtri);
    if (nrtri != ntri || rtri != tri) {
      printf("resident SearchForTriangulation differs from the host-array form (%d vs %d)\n", nrtri, ntri);
      return 4;
    }

    // ComputeBoW on the resident frames (BowVector / FeatureVector assembled on the device, the FeatureVector staying in the
    // frame) and SearchByBoW on the resident FeatureVectors: the host-array forms' results expected, to the byte
    {
      std::map<unsigned, double> rbow[2];
      vsg::FeatureVectorCSR rfv[2];
      voc.ComputeBoW(R0.handle(), rbow[0], rfv[0], 2);
      voc.ComputeBoW(R1.handle(), rbow[1], rfv[1], 2);
      std::vector<int32_t> rmatchF;
      const int nrbow = rm.SearchByBoW(R0, kfValid.data(), R1, rmatchF);
      if (rbow[1] != bow[1] || rfv[1].node != fv[1].node || rfv[1].off != fv[1].off || rfv[1].idx != fv[1].idx ||
          nrbow != nbow || rmatchF != matchF) {
        printf("resident ComputeBoW / SearchByBoW differs from the host-array form (%d vs %d)\n", nrbow, nbow);
        return 4;
      }
    }

    std::ofstream f(argv[2], std::ios::binary);
    std::vector<int32_t> head{mono[0], mono[1], nwin, nbow, ninit, d01, ntri, nlast, nsim3, nfuse, nrinit};
    dump(f, head);
    for (int t = 0; t < 2; ++t) dump(f, kps[t]), dump(f, desc[t]);
    dump(f, cand.off), dump(f, cand.idx), dump(f, bestIdx), dump(f, bestDist), dump(f, trainMatch), dump(f, matchF);
    dump(f, init12);
    std::vector<int32_t> bow_ids;
    std::vector<double> bow_vals;
    for (auto &kv : bow[1]) bow_ids.push_back((int32_t)kv.first), bow_vals.push_back(kv.second);
    dump(f, bow_ids), dump(f, bow_vals);
    dump(f, tri_flat);
    dump(f, rLast), dump(f, rSim3), dump(f, rFuseIdx), dump(f, rFuseDist), dump(f, rInit);
    printf("OK %zu %zu win=%d bow=%d init=%d\n", kps[0].size(), kps[1].size(), nwin, nbow, ninit);
    return 0;
  } catch (const std::exception &e) {
    printf("THROW %s\n", e.what());
    return 3;
  }
}
