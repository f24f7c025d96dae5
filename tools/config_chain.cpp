// config_chain.cpp -- the BASELINE.json configurations C3 and C5 AS STATED, through the C ABI from plain C++ (no Python
// in the timed path), each next to the same chain on the CPU oracle and bit-compared with it first.
//
//   C3  EuRoC-style stereo, 752x480, nFeatures 1200 (config/Stereo/EuRoC.yaml): per stereo pair
//         two ORBextractor::operator() on two host threads            Frame.cc:129-132
//         -> both eyes' features made resident (device to device)      Frame.cc:133-160 (mvKeys, mDescriptors)
//         -> Frame::ComputeStereoMatches                               Frame.cc:957-1127
//         -> Frame::ComputeBoW, transform(..., levelsup 4) on a k = 10, L = 6 vocabulary   Frame.cc:882-889
//         -> ORBmatcher::SearchByBoW(KeyFrame = previous pair, Frame)  ORBmatcher.cc:226-428 (TrackReferenceKeyFrame)
//       reported as pairs/s of ONE pipeline (the reference's call pattern) and of several concurrent pipelines.
//   C5  four concurrent 640x480 / 1250-feature camera streams (rs_d435i_rgbd_inertial.launch:13): one extractor, one
//       host thread and one resident frame pair per stream; per frame
//         operator() -> resident frame -> SearchByProjection(Cur, Last) -> SearchByProjection(F, local map points)
//       (System::TrackRGBD -> Tracking.cc:1583, 2955, 3493), streams pinned to device s mod #devices.
//
// Measurement tool: it links the oracle as the checker and as the reported CPU baseline; nothing in libvsg_orb.so does.
//   usage: config_chain [seconds per timed leg = 2] [pipelines for the C3 throughput leg = 4]
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <unistd.h>

#include "../include/vsg_orb.h"
#include "../include/vsg_synth.h"
#include "../oracle/orb_oracle.h"

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define CHECK(c)                                                                                              \
  do {                                                                                                        \
    if (!(c)) {                                                                                               \
      fprintf(stderr, "config_chain: check failed: %s (line %d): %s\n", #c, __LINE__, vsg_last_error());      \
      exit(1);                                                                                                \
    }                                                                                                         \
  } while (0)

// a host thread that stays alive between frames and runs one job at a time (the reference spawns two std::threads per
// stereo Frame, Frame.cc:129-132; a kept worker does the same work without the per-frame thread start)
struct Worker {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<void()> job;
  bool has_job = false, done = true, stop = false;
  Worker() {
    th = std::thread([this] {
      std::unique_lock<std::mutex> lk(m);
      while (true) {
        cv.wait(lk, [this] { return has_job || stop; });
        if (stop) return;
        has_job = false;
        lk.unlock();
        job();
        lk.lock();
        done = true;
        cv.notify_all();
      }
    });
  }
  void run(std::function<void()> f) {
    std::lock_guard<std::mutex> lk(m);
    job = std::move(f), has_job = true, done = false;
    cv.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [this] { return done; });
  }
  ~Worker() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
      cv.notify_all();
    }
    th.join();
  }
};

struct FeatureVec {
  std::vector<int32_t> node, off, idx;
  int n = 0;
  void size_for(int cap) { node.resize(cap + 1), off.resize(cap + 2), idx.resize(cap + 1); }
};

// ------------------------------------------------------------------------------------------------ C3
static const int W3 = 752, H3 = 480, NF3 = 1200, DISP = 17;
static const float MB = 0.11f, MBF = 47.9f;

struct StereoInputs {  // T rectified pairs of one sequence: the right eye sees the scene DISP px further left
  std::vector<std::vector<uint8_t>> left, right;
  explicit StereoInputs(int T, uint32_t seq) {
    std::vector<uint8_t> wide((size_t)(W3 + 64) * H3);
    for (int t = 0; t < T; t++) {
      CHECK(vsg_synth_sequence_frame(W3 + 64, H3, seq, t, 1, 6, wide.data(), W3 + 64) == 0);
      left.emplace_back((size_t)W3 * H3), right.emplace_back((size_t)W3 * H3);
      for (int r = 0; r < H3; r++) {
        memcpy(&left.back()[(size_t)r * W3], &wide[(size_t)r * (W3 + 64) + 16], W3);
        memcpy(&right.back()[(size_t)r * W3], &wide[(size_t)r * (W3 + 64) + 16 + DISP], W3);
      }
    }
  }
};

struct PairResult {  // everything the chain produces for one pair
  int nL = 0, nR = 0, nstereo = 0, n_bow = 0, nmatch = 0;
  std::vector<vsg_keypoint> kpL, kpR;
  std::vector<uint8_t> dsL, dsR, valid;
  std::vector<float> uR, depth;
  std::vector<int32_t> bow_id, match;
  std::vector<double> bow_val;
  FeatureVec fv;
  void size_for(int cap) {
    kpL.resize(cap), kpR.resize(cap), dsL.resize((size_t)cap * 32), dsR.resize((size_t)cap * 32), valid.resize(cap);
    uR.resize(cap), depth.resize(cap), bow_id.resize(cap + 1), bow_val.resize(cap + 1), match.resize(cap);
    fv.size_for(cap);
  }
};

struct C3Gpu {
  vsg_orb *exL = nullptr, *exR = nullptr, *ex2 = nullptr;  // ex2: ONE handle that takes both eyes as a 2-frame batch
  std::vector<uint8_t> both;                               // the two eyes back to back (the batch's input layout)
  std::vector<vsg_keypoint> kp2;
  std::vector<uint8_t> ds2;
  double batched_ms[3] = {0, 0, 0};
  vsg_frame *FL[2] = {nullptr, nullptr}, *FR = nullptr;
  vsg_vocab *voc = nullptr;  // shared, not owned
  int cap = 0, device = 0;
  PairResult res[2];  // ping-pong: res[t & 1] = current pair, res[(t + 1) & 1] = the previous one (the keyframe)
  Worker right_eye;
  double stage_ms[5] = {0, 0, 0, 0, 0}, fused_ms[3] = {0, 0, 0};
  long pairs = 0;
  void init(int dev, vsg_vocab *v) {
    device = dev, voc = v;
    CHECK(vsg_orb_create(NF3, 1.2f, 8, 20, 7, dev, 1, &exL) == VSG_OK);
    CHECK(vsg_orb_create(NF3, 1.2f, 8, 20, 7, dev, 1, &exR) == VSG_OK);
    cap = vsg_orb_capacity(exL, H3, W3);
    CHECK(cap > 0);
    for (int i = 0; i < 2; i++) {
      CHECK(vsg_frame_create(dev, cap, &FL[i]) == VSG_OK);
      res[i].size_for(cap);
    }
    CHECK(vsg_frame_create(dev, cap, &FR) == VSG_OK);
    CHECK(vsg_orb_create(NF3, 1.2f, 8, 20, 7, dev, 2, &ex2) == VSG_OK);
    CHECK(vsg_orb_capacity(ex2, H3, W3) == cap);
    both.resize((size_t)2 * W3 * H3), kp2.resize((size_t)2 * cap), ds2.resize((size_t)2 * cap * 32);
  }
  // The same first three stages with ONE handle and ONE call for both eyes (a rectified pair has the same lapping area
  // {0, 0} for both, Frame.cc:108-132): one chain of launches over two frames instead of two chains from two host threads.
  // Results go through the same checks (left / right keypoints, descriptors, mvuRight, mvDepth).
  bool pair_batched(const uint8_t *left, const uint8_t *right, const PairResult &want) {
    double t0 = now_ms();
    memcpy(both.data(), left, (size_t)W3 * H3), memcpy(both.data() + (size_t)W3 * H3, right, (size_t)W3 * H3);
    int n[2], mono[2];
    CHECK(vsg_orb_extract_batch(ex2, both.data(), 2, (size_t)W3 * H3, H3, W3, W3, 0, 0, kp2.data(), ds2.data(), cap, n, mono) == VSG_OK);
    double t1 = now_ms();
    CHECK(vsg_frame_from_extractor(FL[0], ex2, 0, kp2.data(), n[0], 0.f, 0.f, (float)W3, (float)H3) == VSG_OK);
    CHECK(vsg_frame_from_extractor(FR, ex2, 1, kp2.data() + cap, n[1], 0.f, 0.f, (float)W3, (float)H3) == VSG_OK);
    double t2 = now_ms();
    std::vector<float> uR(cap), depth(cap);
    const int ns = vsg_frame_stereo_matches(ex2, 0, ex2, 1, FL[0], FR, MB, MBF, uR.data(), depth.data());
    double t3 = now_ms();
    batched_ms[0] += t1 - t0, batched_ms[1] += t2 - t1, batched_ms[2] += t3 - t2;
    return ns == want.nstereo && n[0] == want.nL && n[1] == want.nR && !memcmp(kp2.data(), want.kpL.data(), (size_t)n[0] * 28) &&
           !memcmp(kp2.data() + cap, want.kpR.data(), (size_t)n[1] * 28) && !memcmp(ds2.data(), want.dsL.data(), (size_t)n[0] * 32) &&
           !memcmp(ds2.data() + (size_t)cap * 32, want.dsR.data(), (size_t)n[1] * 32) &&
           !memcmp(uR.data(), want.uR.data(), (size_t)n[0] * 4) && !memcmp(depth.data(), want.depth.data(), (size_t)n[0] * 4);
  }
  // one stereo pair through the chain; `have_prev`: a previous pair exists to search against
  // fused = true (round 6): ComputeStereoMatches + ComputeBoW + SearchByBoW as ONE call and one wait
  // (vsg_frame_stereo_bow_search; the KeyFrame's FeatureVector is resident from its own pair); false: the three blocking calls
  void pair(const uint8_t *left, const uint8_t *right, int t, bool have_prev, bool fused = false) {
    PairResult &R = res[t & 1], &P = res[(t + 1) & 1];
    vsg_frame *cur = FL[t & 1], *prev = FL[(t + 1) & 1];
    double t0 = now_ms();
    // each eye's thread extracts AND makes its features resident in ONE call and one wait (vsg_orb_extract_to_frame: the grid
    // launch rides behind operator()'s chain; rectified pair: no distortion model, Frame.cc:893-897): the two eyes are
    // independent until ComputeStereoMatches
    right_eye.run([&] {
      CHECK(vsg_orb_extract_to_frame(exR, right, H3, W3, W3, 0, 0, R.kpR.data(), R.dsR.data(), cap, &R.nR, FR, nullptr, nullptr, 0,
                                     0.f, 0.f, (float)W3, (float)H3, nullptr) >= 0);
    });
    CHECK(vsg_orb_extract_to_frame(exL, left, H3, W3, W3, 0, 0, R.kpL.data(), R.dsL.data(), cap, &R.nL, cur, nullptr, nullptr, 0, 0.f,
                                   0.f, (float)W3, (float)H3, nullptr) >= 0);
    double t1 = now_ms();
    right_eye.wait();
    double t2 = now_ms();
    if (fused) {
      R.nmatch = 0;
      CHECK(vsg_frame_stereo_bow_search(exL, 0, exR, 0, cur, FR, MB, MBF, R.uR.data(), R.depth.data(), &R.nstereo, voc, 4,
                                        R.bow_id.data(), R.bow_val.data(), cap, &R.n_bow, R.fv.node.data(), R.fv.off.data(),
                                        R.fv.idx.data(), cap, &R.fv.n, have_prev ? prev : nullptr, P.valid.data(), 0.7f, 1,
                                        R.match.data(), &R.nmatch) == VSG_OK);
      for (int i = 0; i < R.nL; i++) R.valid[i] = R.uR[i] >= 0.f;
      const double tf = now_ms();
      fused_ms[0] += t1 - t0, fused_ms[1] += t2 - t1, fused_ms[2] += tf - t2;
      pairs++;
      return;
    }
    R.nstereo = vsg_frame_stereo_matches(exL, 0, exR, 0, cur, FR, MB, MBF, R.uR.data(), R.depth.data());
    CHECK(R.nstereo >= 0);
    double t3 = now_ms();
    CHECK(vsg_frame_bow_transform(voc, cur, 4, R.bow_id.data(), R.bow_val.data(), cap, &R.n_bow, R.fv.node.data(),
                                  R.fv.off.data(), R.fv.idx.data(), cap, &R.fv.n, nullptr, nullptr, nullptr) == VSG_OK);
    double t4 = now_ms();
    for (int i = 0; i < R.nL; i++) R.valid[i] = R.uR[i] >= 0.f;  // "has a map point": the stereo-initialised features
    R.nmatch = 0;
    if (have_prev) {
      R.nmatch = vsg_frame_search_by_bow_kf_f(prev, P.valid.data(), P.fv.node.data(), P.fv.off.data(), P.fv.idx.data(),
                                              P.fv.n, cur, R.fv.node.data(), R.fv.off.data(), R.fv.idx.data(), R.fv.n,
                                              0.7f, 1, R.match.data());
      CHECK(R.nmatch >= 0);
    }
    double t5 = now_ms();
    stage_ms[0] += t1 - t0, stage_ms[1] += t2 - t1, stage_ms[2] += t3 - t2, stage_ms[3] += t4 - t3, stage_ms[4] += t5 - t4;
    pairs++;
  }
  ~C3Gpu() {
    for (int i = 0; i < 2; i++) vsg_frame_destroy(FL[i]);
    vsg_frame_destroy(FR);
    vsg_orb_destroy(exL), vsg_orb_destroy(exR), vsg_orb_destroy(ex2);
  }
};

struct C3Cpu {  // the same chain on the oracle: two host threads for the two eyes, everything else on the caller
  OrExtractor *exL, *exR;
  OrVocab *voc;
  int cap;
  PairResult res[2];
  Worker right_eye;
  long pairs = 0;
  C3Cpu(OrVocab *v, int cap_) : voc(v), cap(cap_) {
    exL = or_create(NF3, 1.2f, 8, 20, 7), exR = or_create(NF3, 1.2f, 8, 20, 7);
    res[0].size_for(cap), res[1].size_for(cap);
  }
  // fused = true (round 6): ComputeStereoMatches + ComputeBoW + SearchByBoW as ONE call and one wait
  // (vsg_frame_stereo_bow_search; the KeyFrame's FeatureVector is resident from its own pair); false: the three blocking calls
  void pair(const uint8_t *left, const uint8_t *right, int t, bool have_prev, bool fused = false) {
    PairResult &R = res[t & 1], &P = res[(t + 1) & 1];
    right_eye.run([&] { or_extract(exR, right, H3, W3, W3, 0, 0, (OrKeyPoint *)R.kpR.data(), R.dsR.data(), cap, &R.nR); });
    or_extract(exL, left, H3, W3, W3, 0, 0, (OrKeyPoint *)R.kpL.data(), R.dsL.data(), cap, &R.nL);
    right_eye.wait();
    or_stereo_matches(exL, exR, (const OrKeyPoint *)R.kpL.data(), R.dsL.data(), R.nL, (const OrKeyPoint *)R.kpR.data(),
                      R.dsR.data(), R.nR, MB, MBF, R.uR.data(), R.depth.data());
    R.nstereo = 0;
    for (int i = 0; i < R.nL; i++) R.valid[i] = R.uR[i] >= 0.f, R.nstereo += R.valid[i];
    or_vocab_transform(voc, R.dsL.data(), R.nL, 4, R.bow_id.data(), R.bow_val.data(), cap, &R.n_bow, R.fv.node.data(),
                       R.fv.off.data(), R.fv.idx.data(), cap, &R.fv.n, nullptr, nullptr, nullptr);
    R.nmatch = 0;
    if (have_prev) {
      std::vector<float> aP(P.nL), aR(R.nL);
      for (int i = 0; i < P.nL; i++) aP[i] = P.kpL[i].angle;
      for (int i = 0; i < R.nL; i++) aR[i] = R.kpL[i].angle;
      R.nmatch = or_search_by_bow_kf_f(P.dsL.data(), aP.data(), P.valid.data(), P.nL, P.fv.node.data(), P.fv.off.data(),
                                       P.fv.idx.data(), P.fv.n, R.dsL.data(), aR.data(), R.nL, R.fv.node.data(),
                                       R.fv.off.data(), R.fv.idx.data(), R.fv.n, 0.7f, 1, R.match.data());
    }
    pairs++;
  }
  ~C3Cpu() { or_destroy(exL), or_destroy(exR); }
};

static bool same_pair(const PairResult &a, const PairResult &b, bool with_match) {
  if (a.nL != b.nL || a.nR != b.nR || a.n_bow != b.n_bow || a.fv.n != b.fv.n || a.nstereo != b.nstereo) return false;
  if (memcmp(a.kpL.data(), b.kpL.data(), (size_t)a.nL * 28) || memcmp(a.dsL.data(), b.dsL.data(), (size_t)a.nL * 32)) return false;
  if (memcmp(a.kpR.data(), b.kpR.data(), (size_t)a.nR * 28) || memcmp(a.dsR.data(), b.dsR.data(), (size_t)a.nR * 32)) return false;
  if (memcmp(a.uR.data(), b.uR.data(), (size_t)a.nL * 4) || memcmp(a.depth.data(), b.depth.data(), (size_t)a.nL * 4)) return false;
  if (memcmp(a.bow_id.data(), b.bow_id.data(), (size_t)a.n_bow * 4) || memcmp(a.bow_val.data(), b.bow_val.data(), (size_t)a.n_bow * 8)) return false;
  if (memcmp(a.fv.node.data(), b.fv.node.data(), (size_t)a.fv.n * 4) || memcmp(a.fv.off.data(), b.fv.off.data(), (size_t)(a.fv.n + 1) * 4)) return false;
  if (memcmp(a.fv.idx.data(), b.fv.idx.data(), (size_t)a.fv.off[a.fv.n] * 4)) return false;
  if (with_match && (a.nmatch != b.nmatch || memcmp(a.match.data(), b.match.data(), (size_t)a.nL * 4))) return false;
  return true;
}

static std::string run_c3(double seconds, int npipes) {
  const int T = 8;
  StereoInputs in(T, 41);
  std::vector<uint8_t> blob(vsg_synth_vocabulary(10, 6, 7, 0, 0, 0.02, nullptr, 0));
  CHECK(vsg_synth_vocabulary(10, 6, 7, 0, 0, 0.02, blob.data(), blob.size()) == blob.size());
  vsg_vocab *voc = nullptr;
  double t0 = now_ms();
  CHECK(vsg_vocab_load(0, blob.data(), blob.size(), &voc) == VSG_OK);
  const double voc_load_ms = now_ms() - t0;
  OrVocab *ovoc = or_vocab_load(blob.data(), blob.size());
  CHECK(ovoc != nullptr);
  int vk, vL, vn, vw;
  vsg_vocab_info(voc, &vk, &vL, nullptr, nullptr, &vn, &vw);
  // ---- parity: every pair of the sequence, every output of the chain
  std::vector<C3Gpu> g(npipes > 1 ? npipes : 1);
  for (auto &p : g) p.init(0, voc);
  C3Cpu c(ovoc, g[0].cap);
  bool parity = true;
  long matches = 0, stereo = 0, nodes = 0;
  for (int t = 0; t < T; t++) {
    g[0].pair(in.left[t].data(), in.right[t].data(), t, t > 0);
    c.pair(in.left[t].data(), in.right[t].data(), t, t > 0);
    parity = parity && same_pair(g[0].res[t & 1], c.res[t & 1], t > 0);
    matches += c.res[t & 1].nmatch, stereo += c.res[t & 1].nstereo, nodes += c.res[t & 1].fv.n;
  }
  // ... and once more through the ONE-call form of the three steps behind the extraction (every pair, every output)
  bool fused_parity = true;
  for (int t = 0; t < T; t++) {
    g[0].pair(in.left[t].data(), in.right[t].data(), t, t > 0, true);
    c.pair(in.left[t].data(), in.right[t].data(), t, t > 0);
    fused_parity = fused_parity && same_pair(g[0].res[t & 1], c.res[t & 1], t > 0);
  }
  parity = parity && fused_parity;
  // ---- one pipeline, the reference's call pattern: first as three blocking calls (the per-stage figures) ...
  for (auto &p : g) memset(p.stage_ms, 0, sizeof p.stage_ms), memset(p.fused_ms, 0, sizeof p.fused_ms), p.pairs = 0;
  t0 = now_ms();
  int t = T;
  while (now_ms() - t0 < seconds * 500) g[0].pair(in.left[t % T].data(), in.right[t % T].data(), t, true), t++;
  const double sep_ms = (now_ms() - t0) / g[0].pairs;
  double st[5];
  for (int i = 0; i < 5; i++) st[i] = g[0].stage_ms[i] / g[0].pairs;
  // ... then with the one-call form (what ms_per_pair / pairs_per_s quote)
  g[0].pairs = 0;
  t0 = now_ms();
  while (now_ms() - t0 < seconds * 1e3) g[0].pair(in.left[t % T].data(), in.right[t % T].data(), t, true, true), t++;
  const double one_ms = (now_ms() - t0) / g[0].pairs;
  double fst[3];
  for (int i = 0; i < 3; i++) fst[i] = g[0].fused_ms[i] / g[0].pairs;
  // ---- several pipelines side by side (independent stereo rigs / sequences on one GPU)
  double multi = 0;
  if (npipes > 1) {
    std::atomic<long> total(0);
    std::vector<std::thread> th;
    const double tm0 = now_ms();
    for (int p = 0; p < npipes; p++)
      th.emplace_back([&, p] {
        int tt = T;
        long n = 0;
        while (now_ms() - tm0 < seconds * 1e3) g[p].pair(in.left[tt % T].data(), in.right[tt % T].data(), tt, true, true), tt++, n++;
        total += n;
      });
    for (auto &x : th) x.join();
    multi = total / ((now_ms() - tm0) * 1e-3);
  }
  // ---- both eyes as one 2-frame batch on one handle (parity against the oracle's pair, then timed)
  bool batched_parity = true;
  for (int tt = 0; tt < T; tt++) {
    c.pair(in.left[tt].data(), in.right[tt].data(), tt, false);
    batched_parity = batched_parity && g[0].pair_batched(in.left[tt].data(), in.right[tt].data(), c.res[tt & 1]);
  }
  memset(g[0].batched_ms, 0, sizeof g[0].batched_ms);
  long nb = 0;
  t0 = now_ms();
  for (t = 0; now_ms() - t0 < seconds * 500; t++, nb++) g[0].pair_batched(in.left[t % T].data(), in.right[t % T].data(), c.res[0]);
  double bst[3];
  for (int i = 0; i < 3; i++) bst[i] = g[0].batched_ms[i] / nb;
  // ---- the oracle's chain
  c.pairs = 0;
  t0 = now_ms();
  t = T;
  while (now_ms() - t0 < seconds * 1e3) c.pair(in.left[t % T].data(), in.right[t % T].data(), t, true), t++;
  const double cpu_ms = (now_ms() - t0) / c.pairs;
  char b[4096];
  snprintf(b, sizeof b,
           "{\"workload\": \"C3: stereo 752x480, nFeatures=1200; per pair 2 x (operator() -> resident frame) on two host threads "
           "(stage extract_2_eyes = the left eye's operator() + resident frame in one call, make_resident_2 = until the right eye's thread is done too) "
           "-> ComputeStereoMatches -> ComputeBoW (k=%d, L=%d vocabulary, %d nodes, levelsup 4) -> SearchByBoW(KF = "
           "previous pair, F); ms_per_pair / pairs_per_s: the three steps behind the extraction as ONE call and one wait "
           "(vsg_frame_stereo_bow_search), stage_ms: the same pairs through three blocking calls\", \"unit\": \"stereo pairs/s\", "
           "\"pairs_per_s\": %.1f, \"ms_per_pair\": %.4f, \"ms_per_pair_three_calls\": %.4f, "
           "\"fused_stage_ms\": {\"extract_2_eyes\": %.4f, \"make_resident_2\": %.4f, \"stereo_bow_search\": %.4f}, "
           "\"stage_ms\": {\"extract_2_eyes\": %.4f, \"make_resident_2\": %.4f, \"stereo_matches\": %.4f, \"compute_bow\": %.4f, "
           "\"search_by_bow\": %.4f}, \"batched_pair\": {\"what\": \"both eyes as ONE 2-frame vsg_orb_extract_batch on one handle "
           "(rectified pair: one lapping area) instead of two handles on two host threads\", \"extract_2_eyes\": %.4f, "
           "\"make_resident_2\": %.4f, \"stereo_matches\": %.4f, \"parity\": %s}, \"pipelines\": %d, \"pairs_per_s_all_pipelines\": %.1f, \"parity\": %s, "
           "\"pairs_checked\": %d, \"per_pair\": {\"stereo_matches\": %.1f, \"feature_vector_nodes\": %.1f, \"bow_matches\": %.1f}, "
           "\"cpu_oracle\": {\"pairs_per_s\": %.2f, \"ms_per_pair\": %.3f, \"threads\": 2, \"kind\": \"port\"}, "
           "\"vocabulary_load_ms\": %.1f}",
           vk, vL, vn, 1e3 / one_ms, one_ms, sep_ms, fst[0], fst[1], fst[2], st[0], st[1], st[2], st[3], st[4], bst[0], bst[1], bst[2],
           batched_parity ? "true" : "false", npipes, multi, parity ? "true" : "false", T,
           (double)stereo / T, (double)nodes / T, (double)matches / (T - 1), 1e3 / cpu_ms, cpu_ms, voc_load_ms);
  g.clear();
  vsg_vocab_destroy(voc);
  or_vocab_destroy(ovoc);
  return b;
}

// ------------------------------------------------------------------------------------------------ cameras
// The two BASELINE cameras whose Frame constructor really undistorts (mDistCoef(0) != 0): Frame::UndistortKeyPoints
// (Frame.cc:891-921) gives mvKeysUn, Frame::ComputeImageBounds (Frame.cc:924-955) fractional grid bounds.
struct Camera {
  const char *name;
  float K4[4], dist[5];
  int ndist;
  float bounds[4];  // mnMinX, mnMinY, mnMaxX, mnMaxY (vsg_camera_image_bounds, checked against the oracle's)
  void init(int cols, int rows) {
    CHECK(vsg_camera_image_bounds(cols, rows, K4, dist, ndist, bounds) == VSG_OK);
    float ob[4];
    or_image_bounds(cols, rows, K4, dist, ndist, ob);
    CHECK(!memcmp(ob, bounds, sizeof ob));
  }
};
// config/RGB-D-Inertial/RealSense_D435i.yaml:11-23 (C5) and config/RGB-D/TUM1.yaml:11-23 (C1; 640x480 / 1000 = the frame
// latency workload)
static Camera cam_d435i() {
  return {"RealSense D435i (k1 0.125, k2 -0.251, p1 0.0007, p2 0.0062)",
          {6.165911254882812e+02f, 6.166796264648438e+02f, 3.242193603515625e+02f, 2.3942701721191406e+02f},
          {1.25323e-01f, -2.51452e-01f, 7.12e-04f, 6.217e-03f, 0.f}, 4, {0, 0, 0, 0}};
}
static Camera cam_tum1() {
  return {"TUM1 (k1 0.262, k2 -0.953, p1 -0.0054, p2 0.0026, k3 1.163)",
          {517.306408f, 516.469215f, 318.643040f, 255.313989f},
          {0.262383f, -0.953104f, -0.005358f, 0.002628f, 1.163314f}, 5, {0, 0, 0, 0}};
}

// ------------------------------------------------------------------------------------------------ C5
static const int W5 = 640, H5 = 480, NF5 = 1250, NSTREAM = 5 - 1;

struct TrackResult {
  int n = 0, n_last = 0, n_local = 0;
  std::vector<vsg_keypoint> kp, kpun;  // mvKeys, mvKeysUn
  std::vector<uint8_t> ds, tb;
  std::vector<int32_t> tm_last, tm_local;
  void size_for(int cap) {
    kp.resize(cap), kpun.resize(cap), ds.resize((size_t)cap * 32), tb.resize(cap), tm_last.resize(cap), tm_local.resize(cap);
  }
};

// the projections of the previous frame's features into the current one: the scene moves by (-3, -2) px per frame
struct Queries {
  std::vector<float> u, v, ang, vc;
  std::vector<int32_t> oct;
  std::vector<uint8_t> obs;
  void from(const TrackResult &P) {
    u.resize(P.n), v.resize(P.n), ang.resize(P.n), vc.assign(P.n, 0.9f), oct.resize(P.n), obs.assign(P.n, 1);
    for (int i = 0; i < P.n; i++)  // projected from the previous frame's UNDISTORTED keypoints, like the reference's geometry
      u[i] = P.kpun[i].x - 3.f, v[i] = P.kpun[i].y - 2.f, ang[i] = P.kpun[i].angle, oct[i] = P.kpun[i].octave;
  }
};

struct C5Gpu {
  vsg_orb *ex = nullptr;
  vsg_frame *F[2] = {nullptr, nullptr};
  int cap = 0;
  float sf[8];
  TrackResult res[2];
  Queries q;
  long frames = 0;
  Camera cam = cam_d435i();
  void init(int dev) {
    cam.init(W5, H5);
    CHECK(vsg_orb_create(NF5, 1.2f, 8, 20, 7, dev, 1, &ex) == VSG_OK);
    cap = vsg_orb_capacity(ex, H5, W5);
    CHECK(cap > 0);
    vsg_orb_get_tables(ex, sf, nullptr, nullptr, nullptr, nullptr, nullptr);
    for (int i = 0; i < 2; i++) {
      CHECK(vsg_frame_create(dev, cap, &F[i]) == VSG_OK);
      res[i].size_for(cap);
    }
  }
  void frame(const uint8_t *img, int t, bool have_prev) {
    TrackResult &R = res[t & 1], &P = res[(t + 1) & 1];
    // the Frame constructor's front end in one call and one wait: operator() -> UndistortKeyPoints on the device (inside the
    // launch that builds the grid, right behind the extractor's chain) -> resident frame; mvKeysUn comes back for the host
    CHECK(vsg_orb_extract_to_frame(ex, img, H5, W5, W5, 0, 0, R.kp.data(), R.ds.data(), cap, &R.n, F[t & 1], cam.K4, cam.dist,
                                   cam.ndist, cam.bounds[0], cam.bounds[1], cam.bounds[2], cam.bounds[3], R.kpun.data()) >= 0);
    R.n_last = R.n_local = 0;
    if (have_prev) {
      q.from(P);
      std::fill(R.tb.begin(), R.tb.begin() + R.n, 0), std::fill(R.tm_last.begin(), R.tm_last.begin() + R.n, -1);
      R.n_last = vsg_frame_search_by_projection_last(F[t & 1], P.n, P.ds.data(), q.obs.data(), q.u.data(), q.v.data(), nullptr,
                                                     nullptr, nullptr, q.oct.data(), q.ang.data(), 15.f, 0, sf, 8, 1,
                                                     R.tb.data(), R.tm_last.data());
      CHECK(R.n_last >= 0);
      std::fill(R.tb.begin(), R.tb.begin() + R.n, 0), std::fill(R.tm_local.begin(), R.tm_local.begin() + R.n, -1);
      R.n_local = vsg_frame_search_by_projection(F[t & 1], P.n, P.ds.data(), q.obs.data(), q.obs.data(), q.u.data(), q.v.data(),
                                                 q.u.data(), q.oct.data(), q.vc.data(), nullptr, nullptr, nullptr, nullptr,
                                                 nullptr, 1.f, 0.8f, sf, 8, nullptr, nullptr, R.tb.data(), R.tm_local.data());
      CHECK(R.n_local >= 0);
    }
    frames++;
  }
  ~C5Gpu() {
    vsg_frame_destroy(F[0]), vsg_frame_destroy(F[1]);
    vsg_orb_destroy(ex);
  }
};

struct C5Cpu {
  OrExtractor *ex;
  int cap;
  float sf[8];
  TrackResult res[2];
  Queries q;
  long frames = 0;
  Camera cam = cam_d435i();
  explicit C5Cpu(int cap_) : cap(cap_) {
    cam.init(W5, H5);
    ex = or_create(NF5, 1.2f, 8, 20, 7);
    or_get_tables(ex, sf, nullptr, nullptr, nullptr, nullptr, nullptr);
    res[0].size_for(cap), res[1].size_for(cap);
  }
  void frame(const uint8_t *img, int t, bool have_prev) {
    TrackResult &R = res[t & 1], &P = res[(t + 1) & 1];
    or_extract(ex, img, H5, W5, W5, 0, 0, (OrKeyPoint *)R.kp.data(), R.ds.data(), cap, &R.n);
    or_undistort_keypoints((const OrKeyPoint *)R.kp.data(), R.n, cam.K4, cam.dist, cam.ndist, (OrKeyPoint *)R.kpun.data());
    OrFrame *f = or_frame_create((const OrKeyPoint *)R.kpun.data(), R.ds.data(), nullptr, R.n, -1, cam.bounds[0], cam.bounds[1],
                                 cam.bounds[2], cam.bounds[3]);
    R.n_last = R.n_local = 0;
    if (have_prev) {
      q.from(P);
      std::fill(R.tb.begin(), R.tb.begin() + R.n, 0), std::fill(R.tm_last.begin(), R.tm_last.begin() + R.n, -1);
      R.n_last = or_frame_search_by_projection_last(f, P.n, P.ds.data(), q.obs.data(), q.u.data(), q.v.data(), q.u.data(), nullptr,
                                                    nullptr, q.oct.data(), q.ang.data(), 15.f, 0, 0, sf, 1, R.tb.data(),
                                                    R.tm_last.data());
      std::fill(R.tb.begin(), R.tb.begin() + R.n, 0), std::fill(R.tm_local.begin(), R.tm_local.begin() + R.n, -1);
      R.n_local = or_frame_search_by_projection(f, P.n, P.ds.data(), q.obs.data(), q.obs.data(), q.u.data(), q.v.data(), q.u.data(),
                                                q.oct.data(), q.vc.data(), nullptr, nullptr, nullptr, nullptr, nullptr, 1.f, 0.8f,
                                                sf, nullptr, nullptr, R.tb.data(), R.tm_local.data());
    }
    or_frame_destroy(f);
    frames++;
  }
  ~C5Cpu() { or_destroy(ex); }
};

static bool same_track(const TrackResult &a, const TrackResult &b) {
  return a.n == b.n && a.n_last == b.n_last && a.n_local == b.n_local && !memcmp(a.kp.data(), b.kp.data(), (size_t)a.n * 28) &&
         !memcmp(a.kpun.data(), b.kpun.data(), (size_t)a.n * 28) &&
         !memcmp(a.ds.data(), b.ds.data(), (size_t)a.n * 32) && !memcmp(a.tm_last.data(), b.tm_last.data(), (size_t)a.n * 4) &&
         !memcmp(a.tm_local.data(), b.tm_local.data(), (size_t)a.n * 4);
}

static std::string run_c5(double seconds) {
  const int T = 6;
  int ndev = vsg_device_count();
  CHECK(ndev > 0);
  std::vector<std::vector<std::vector<uint8_t>>> img(NSTREAM);
  for (int s = 0; s < NSTREAM; s++)
    for (int t = 0; t < T; t++) {
      img[s].emplace_back((size_t)W5 * H5);
      CHECK(vsg_synth_sequence_frame(W5, H5, 500 + s, t, 1, 6, img[s].back().data(), W5) == 0);
    }
  std::vector<C5Gpu> g(NSTREAM);
  for (int s = 0; s < NSTREAM; s++) g[s].init(s % ndev);  // per-stream GPU pinning
  // ---- parity, every stream, every frame
  bool parity = true;
  long m_last = 0, m_local = 0, kps = 0;
  {
    std::vector<std::thread> th;
    std::vector<int> ok(NSTREAM, 1);
    std::vector<long> ml(NSTREAM, 0), mo(NSTREAM, 0), kk(NSTREAM, 0);
    for (int s = 0; s < NSTREAM; s++)
      th.emplace_back([&, s] {
        C5Cpu c(g[s].cap);
        for (int t = 0; t < T; t++) {
          g[s].frame(img[s][t].data(), t, t > 0);
          c.frame(img[s][t].data(), t, t > 0);
          ok[s] &= same_track(g[s].res[t & 1], c.res[t & 1]);
          ml[s] += c.res[t & 1].n_last, mo[s] += c.res[t & 1].n_local, kk[s] += c.res[t & 1].n;
        }
      });
    for (auto &x : th) x.join();
    for (int s = 0; s < NSTREAM; s++) parity = parity && ok[s], m_last += ml[s], m_local += mo[s], kps += kk[s];
  }
  // ---- the four streams, one host thread each, concurrently
  auto run_gpu = [&](int nstream) {
    std::atomic<long> total(0);
    std::vector<std::thread> th;
    const double t0 = now_ms();
    for (int s = 0; s < nstream; s++)
      th.emplace_back([&, s] {
        int t = T;
        long n = 0;
        while (now_ms() - t0 < seconds * 1e3) g[s].frame(img[s][t % T].data(), t, true), t++, n++;
        total += n;
      });
    for (auto &x : th) x.join();
    return total / ((now_ms() - t0) * 1e-3);
  };
  const double fps4 = run_gpu(NSTREAM), fps1 = run_gpu(1);
  // ---- the oracle: four host threads, one stream each
  double cpu_fps;
  {
    std::atomic<long> total(0);
    std::vector<std::thread> th;
    const double t0 = now_ms();
    for (int s = 0; s < NSTREAM; s++)
      th.emplace_back([&, s] {
        C5Cpu c(g[s].cap);
        int t = 0;
        long n = 0;
        while (now_ms() - t0 < seconds * 1e3) c.frame(img[s][t % T].data(), t, t > 0), t++, n++;
        total += n;
      });
    for (auto &x : th) x.join();
    cpu_fps = total / ((now_ms() - t0) * 1e-3);
  }
  char b[3072];
  snprintf(b, sizeof b,
           "{\"workload\": \"C5: %d concurrent 640x480 camera streams, nFeatures=1250, one extractor + one host thread per "
           "stream, camera = %s: per frame vsg_orb_extract_to_frame (operator() -> UndistortKeyPoints on the device + resident frame, one wait) on the grid bounds "
           "(%.3f, %.3f, %.3f, %.3f) of ComputeImageBounds -> SearchByProjection(Cur, Last) -> SearchByProjection(F, local "
           "map points); stream s on device s mod %d\", \"unit\": \"frames/s\", \"frames_per_s\": %.1f, "
           "\"frames_per_s_one_stream\": %.1f, \"ms_per_frame_one_stream\": %.4f, \"streams\": %d, \"devices\": %d, "
           "\"parity\": %s, \"frames_checked\": %d, \"per_frame\": {\"keypoints\": %.1f, \"matches_last_frame\": %.1f, "
           "\"matches_local_map\": %.1f}, \"cpu_oracle\": {\"frames_per_s\": %.2f, \"threads\": %d, \"kind\": \"port\"}}",
           NSTREAM, g[0].cam.name, g[0].cam.bounds[0], g[0].cam.bounds[1], g[0].cam.bounds[2], g[0].cam.bounds[3], ndev, fps4, fps1,
           1e3 / fps1, NSTREAM, ndev, parity ? "true" : "false", NSTREAM * T,
           (double)kps / (NSTREAM * T), (double)m_last / (NSTREAM * (T - 1)), (double)m_local / (NSTREAM * (T - 1)), cpu_fps,
           NSTREAM);
  return b;
}

// ------------------------------------------------------------------------------------------------ frame latency
// The call pattern the reference has: ONE frame per operator() (System::TrackRGBD -> Tracking::GrabImageRGBD ->
// Frame::ExtractORB, System.cc:359, Tracking.cc:1583, Frame.cc:344,555-563), then the two tracking searches
// (Tracking.cc:2955 TrackWithMotionModel, :3493 SearchLocalPoints).  C2 frames (640x480 / 1000), host image in, host
// records out, wall time per call through the C ABI next to the oracle's on one host thread.
static std::string run_latency(double seconds) {
  const int W = 640, H = 480, T = 6;
  std::vector<std::vector<uint8_t>> img;
  for (int t = 0; t < T; t++) {
    img.emplace_back((size_t)W * H);
    CHECK(vsg_synth_sequence_frame(W, H, 1000, t, 1, 6, img.back().data(), W) == 0);
  }
  vsg_orb *ex = nullptr;
  CHECK(vsg_orb_create(1000, 1.2f, 8, 20, 7, 0, 1, &ex) == VSG_OK);
  const int cap = vsg_orb_capacity(ex, H, W);
  OrExtractor *oe = or_create(1000, 1.2f, 8, 20, 7);
  float sf[8];
  vsg_orb_get_tables(ex, sf, nullptr, nullptr, nullptr, nullptr, nullptr);
  vsg_frame *F[2];
  TrackResult R[2], O[2];
  for (int i = 0; i < 2; i++) {
    CHECK(vsg_frame_create(0, cap, &F[i]) == VSG_OK);
    R[i].size_for(cap), O[i].size_for(cap);
  }
  Queries q;
  bool parity = true;
  Camera cam = cam_tum1();  // C1's camera (TUM1.yaml: 640x480 / 1000 features): mDistCoef(0) != 0
  cam.init(W, H);
  // one frame of each chain; level: 0 = operator() only, 1 = + UndistortKeyPoints + resident frame, 2 = + the two tracking
  // searches; `upload` = the other route to the same resident frame: mvKeysUn on the HOST (the reference's
  // cv::undistortPoints; here the oracle's restatement stands in for it), then vsg_frame_upload(mvKeysUn, bounds)
  auto gpu = [&](int t, int level, bool upload = false) {
    TrackResult &C = R[t & 1], &P = R[(t + 1) & 1];
    if (level >= 1 && !upload) {  // the fused front end (vsg_orb_extract_to_frame): one call, one wait
      CHECK(vsg_orb_extract_to_frame(ex, img[t % T].data(), H, W, W, 0, 0, C.kp.data(), C.ds.data(), cap, &C.n, F[t & 1], cam.K4,
                                     cam.dist, cam.ndist, cam.bounds[0], cam.bounds[1], cam.bounds[2], cam.bounds[3],
                                     C.kpun.data()) >= 0);
    } else {
      CHECK(vsg_orb_extract(ex, img[t % T].data(), H, W, W, 0, 0, C.kp.data(), C.ds.data(), cap, &C.n) >= 0);
    }
    if (level < 1) return;
    if (upload) {
      or_undistort_keypoints((const OrKeyPoint *)C.kp.data(), C.n, cam.K4, cam.dist, cam.ndist, (OrKeyPoint *)C.kpun.data());
      CHECK(vsg_frame_upload(F[t & 1], C.kpun.data(), C.ds.data(), nullptr, C.n, -1, cam.bounds[0], cam.bounds[1], cam.bounds[2],
                             cam.bounds[3]) == VSG_OK);
    }
    if (level < 2 || P.n == 0) return;
    q.from(P);
    std::fill(C.tb.begin(), C.tb.begin() + C.n, 0), std::fill(C.tm_last.begin(), C.tm_last.begin() + C.n, -1);
    C.n_last = vsg_frame_search_by_projection_last(F[t & 1], P.n, P.ds.data(), q.obs.data(), q.u.data(), q.v.data(), nullptr,
                                                   nullptr, nullptr, q.oct.data(), q.ang.data(), 15.f, 0, sf, 8, 1, C.tb.data(),
                                                   C.tm_last.data());
    std::fill(C.tb.begin(), C.tb.begin() + C.n, 0), std::fill(C.tm_local.begin(), C.tm_local.begin() + C.n, -1);
    C.n_local = vsg_frame_search_by_projection(F[t & 1], P.n, P.ds.data(), q.obs.data(), q.obs.data(), q.u.data(), q.v.data(),
                                               q.u.data(), q.oct.data(), q.vc.data(), nullptr, nullptr, nullptr, nullptr, nullptr,
                                               1.f, 0.8f, sf, 8, nullptr, nullptr, C.tb.data(), C.tm_local.data());
    CHECK(C.n_last >= 0 && C.n_local >= 0);
  };
  auto cpu = [&](int t, int level) {
    TrackResult &C = O[t & 1], &P = O[(t + 1) & 1];
    or_extract(oe, img[t % T].data(), H, W, W, 0, 0, (OrKeyPoint *)C.kp.data(), C.ds.data(), cap, &C.n);
    if (level < 1) return;
    or_undistort_keypoints((const OrKeyPoint *)C.kp.data(), C.n, cam.K4, cam.dist, cam.ndist, (OrKeyPoint *)C.kpun.data());
    OrFrame *f = or_frame_create((const OrKeyPoint *)C.kpun.data(), C.ds.data(), nullptr, C.n, -1, cam.bounds[0], cam.bounds[1],
                                 cam.bounds[2], cam.bounds[3]);
    if (level >= 2 && P.n > 0) {
      q.from(P);
      std::fill(C.tb.begin(), C.tb.begin() + C.n, 0), std::fill(C.tm_last.begin(), C.tm_last.begin() + C.n, -1);
      C.n_last = or_frame_search_by_projection_last(f, P.n, P.ds.data(), q.obs.data(), q.u.data(), q.v.data(), q.u.data(), nullptr,
                                                    nullptr, q.oct.data(), q.ang.data(), 15.f, 0, 0, sf, 1, C.tb.data(),
                                                    C.tm_last.data());
      std::fill(C.tb.begin(), C.tb.begin() + C.n, 0), std::fill(C.tm_local.begin(), C.tm_local.begin() + C.n, -1);
      C.n_local = or_frame_search_by_projection(f, P.n, P.ds.data(), q.obs.data(), q.obs.data(), q.u.data(), q.v.data(),
                                                q.u.data(), q.oct.data(), q.vc.data(), nullptr, nullptr, nullptr, nullptr, nullptr,
                                                1.f, 0.8f, sf, nullptr, nullptr, C.tb.data(), C.tm_local.data());
    }
    or_frame_destroy(f);
  };
  for (int t = 0; t < 3 * T; t++) {  // parity over three laps of the sequence, the last one on the upload route
    gpu(t, 2, t >= 2 * T), cpu(t, 2);
    parity = parity && same_track(R[t & 1], O[t & 1]);
  }
  double g_ms[3], c_ms[3], up_ms[2];
  for (int level = 0; level < 3; level++) {
    int t = 0, n = 0;
    for (int w = 0; w < 12; w++) gpu(t++, level);
    double t0 = now_ms();
    while (now_ms() - t0 < seconds * 400) gpu(t++, level), n++;
    g_ms[level] = (now_ms() - t0) / n;
    if (level >= 1) {  // the same level through host undistortion + vsg_frame_upload
      t = 0, n = 0;
      for (int w = 0; w < 12; w++) gpu(t++, level, true);
      t0 = now_ms();
      while (now_ms() - t0 < seconds * 400) gpu(t++, level, true), n++;
      up_ms[level - 1] = (now_ms() - t0) / n;
    }
    t = 0, n = 0;
    t0 = now_ms();
    while (now_ms() - t0 < seconds * 400) cpu(t++, level), n++;
    c_ms[level] = (now_ms() - t0) / n;
  }
  char b[3072];
  snprintf(b, sizeof b,
           "{\"workload\": \"one 640x480 / 1000-feature frame per blocking operator() call through the C ABI, host image in, "
           "host records out; camera = %s, grid bounds (%.3f, %.3f, %.3f, %.3f) from ComputeImageBounds\", "
           "\"extract_ms\": %.4f, \"extract_plus_resident_ms\": %.4f, \"track_chain_ms\": %.4f, "
           "\"upload_route\": {\"extract_plus_resident_ms\": %.4f, \"track_chain_ms\": %.4f, \"what\": \"mvKeysUn on the host "
           "(stand-in for cv::undistortPoints: the oracle's restatement) -> vsg_frame_upload(mvKeysUn, bounds) instead of the "
           "on-device UndistortKeyPoints\"}, "
           "\"cpu_oracle_1_thread\": {\"extract_ms\": %.3f, \"extract_plus_resident_ms\": %.3f, \"track_chain_ms\": %.3f}, "
           "\"parity\": %s, \"graph_launches\": %ld, \"track_chain\": \"vsg_orb_extract_to_frame = operator() -> "
           "UndistortKeyPoints (on the device, FP64, inside the grid launch behind the extractor's chain: one call, one wait) -> "
           "resident frame; then SearchByProjection(Cur, Last) -> SearchByProjection(F, local map points)\"}",
           cam.name, cam.bounds[0], cam.bounds[1], cam.bounds[2], cam.bounds[3], g_ms[0], g_ms[1], g_ms[2], up_ms[0], up_ms[1],
           c_ms[0], c_ms[1], c_ms[2], parity ? "true" : "false", vsg_orb_chain_graph_launches(ex));
  for (int i = 0; i < 2; i++) vsg_frame_destroy(F[i]);
  vsg_orb_destroy(ex);
  or_destroy(oe);
  return b;
}

// VSG_CRASH_MAPS=1: on SIGSEGV write the faulting address, the raw return addresses and /proc/self/maps to stderr (async-
// signal-safe calls only), then die by the default action -- tells WHICH libraries the frames under a crash belong to
// (profiles/r05_q_rocprofv3_kernel_trace_c5_segfault.txt: rocprofv3's own handler prints addresses without names).
static void crash_maps(int sig, siginfo_t *si, void *) {
  char buf[256];
  int n = snprintf(buf, sizeof buf, "\n=== config_chain: signal %d at address %p; return addresses:\n", sig, si->si_addr);
  (void)!write(2, buf, n);
  void *bt[48];
  const int nb = backtrace(bt, 48);
  for (int i = 0; i < nb; i++) {
    n = snprintf(buf, sizeof buf, "  #%d %p\n", i, bt[i]);
    (void)!write(2, buf, n);
  }
  (void)!write(2, "=== /proc/self/maps\n", 20);
  const int fd = open("/proc/self/maps", O_RDONLY);
  if (fd >= 0) {
    static char big[1 << 16];
    ssize_t r;
    while ((r = read(fd, big, sizeof big)) > 0) (void)!write(2, big, r);
    close(fd);
  }
  signal(sig, SIG_DFL);
  raise(sig);
}

int main(int argc, char **argv) {
  if (getenv("VSG_CRASH_MAPS")) {
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = crash_maps;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER;
    sigaction(SIGSEGV, &sa, nullptr);
    sigaction(SIGBUS, &sa, nullptr);
  }
  const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
  const int npipes = argc > 2 ? atoi(argv[2]) : 4;
  const std::string only = argc > 3 ? argv[3] : "";  // "c5": that configuration alone (kernel traces of the four streams)
  const std::string c3 = only.empty() || only == "c3" ? run_c3(seconds, npipes) : "null";
  const std::string c5 = only.empty() || only == "c5" ? run_c5(seconds) : "null";
  const std::string lat = only.empty() || only == "latency" ? run_latency(seconds) : "null";
  printf("{\"C3\": %s, \"C5\": %s, \"frame_latency\": %s}\n", c3.c_str(), c5.c_str(), lat.c_str());
  return 0;
}
