// abi_latency.cpp -- wall time per C-ABI call of the per-frame matcher entry points, from plain C++ (what the
// reference's Tracking / LocalMapping / LoopClosing threads would pay), next to the CPU oracle's routine on ONE host
// thread, on C2-sized inputs (two synthetic 640x480 frames, ~1000 features each).  Every GPU result is compared with
// the oracle's before it is timed.  Prints one JSON object.  Measurement tool: it links the oracle (the checker / the
// reported CPU baseline); nothing in libvsg_orb.so does.
//   usage: abi_latency [iterations]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../include/vsg_orb.h"
#include "../include/vsg_orb_debug.h"
#include "../include/vsg_synth.h"
#include "../oracle/orb_oracle.h"

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static double time_ms(const std::function<void()> &f, int n) {
  for (int i = 0; i < 20; i++) f();
  const double t0 = now_ms();
  for (int i = 0; i < n; i++) f();
  return (now_ms() - t0) / n;
}
#define CHECK(c)                                                \
  do {                                                          \
    if (!(c)) {                                                 \
      fprintf(stderr, "abi_latency: check failed: %s (line %d): %s\n", #c, __LINE__, vsg_last_error()); \
      return 1;                                                 \
    }                                                           \
  } while (0)

int main(int argc, char **argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 500;
  const int W = 640, H = 480;
  std::vector<uint8_t> img[2] = {std::vector<uint8_t>(W * H), std::vector<uint8_t>(W * H)};
  for (int t = 0; t < 2; t++) CHECK(vsg_synth_sequence_frame(W, H, 3, t, 1, 6, img[t].data(), W) == 0);
  vsg_orb *ex = nullptr;
  CHECK(vsg_orb_create(1000, 1.2f, 8, 20, 7, 0, 1, &ex) == VSG_OK);
  const int cap = vsg_orb_capacity(ex, H, W);
  CHECK(cap > 0);
  std::vector<vsg_keypoint> kp[2] = {std::vector<vsg_keypoint>(cap), std::vector<vsg_keypoint>(cap)};
  std::vector<uint8_t> ds[2] = {std::vector<uint8_t>(cap * 32), std::vector<uint8_t>(cap * 32)};
  int n[2];
  vsg_frame *F[2];
  OrFrame *O[2];
  for (int t = 0; t < 2; t++) {
    CHECK(vsg_orb_extract(ex, img[t].data(), H, W, W, 0, 0, kp[t].data(), ds[t].data(), cap, &n[t]) >= 0);
    CHECK(vsg_frame_create(0, cap, &F[t]) == VSG_OK);
    // straight out of the extractor: device to device, grid built by a kernel
    CHECK(vsg_frame_from_extractor(F[t], ex, 0, kp[t].data(), n[t], 0.f, 0.f, (float)W, (float)H) == VSG_OK);
    O[t] = or_frame_create((const OrKeyPoint *)kp[t].data(), ds[t].data(), nullptr, n[t], -1, 0.f, 0.f, (float)W, (float)H);
  }
  float sf[8], inv2[8];
  vsg_orb_get_tables(ex, sf, nullptr, nullptr, inv2, nullptr, nullptr);
  const int nq = n[0], nt = n[1];
  std::vector<float> u(nq), v(nq), ang(nq), r10(nq), r3(nq), vc(nq, 0.9f);
  std::vector<int32_t> oct(nq);
  std::vector<uint8_t> obs(nq, 1);
  for (int i = 0; i < nq; i++) {
    u[i] = kp[0][i].x - 3.f, v[i] = kp[0][i].y - 2.f, ang[i] = kp[0][i].angle, oct[i] = kp[0][i].octave;
    r10[i] = 10.f * sf[oct[i]], r3[i] = 3.f * sf[oct[i]];
  }
  std::string js = "{";
  auto emit = [&](const char *name, double gpu, double cpu, const char *extra = "") {
    char b[512];
    snprintf(b, sizeof b, "%s\"%s\": {\"gpu_ms\": %.4f, \"oracle_1thread_ms\": %.4f%s}", js.size() > 1 ? ", " : "", name,
             gpu, cpu, extra);
    js += b;
  };
  std::vector<uint8_t> tb(nt), tb2(nt);
  std::vector<int32_t> tm(nt), tm2(nt);
  auto reset = [&]() {
    std::fill(tb.begin(), tb.end(), 0), std::fill(tb2.begin(), tb2.end(), 0);
    std::fill(tm.begin(), tm.end(), -1), std::fill(tm2.begin(), tm2.end(), -1);
  };
  // ---- SearchByProjection(CurrentFrame, LastFrame, th = 15, bMono)   Tracking.cc:2955
  {
    auto g = [&]() {
      reset();
      return vsg_frame_search_by_projection_last(F[1], nq, ds[0].data(), obs.data(), u.data(), v.data(), nullptr, nullptr,
                                                 nullptr, oct.data(), ang.data(), 15.f, 0, sf, 8, 1, tb.data(), tm.data());
    };
    auto o = [&]() {
      return or_frame_search_by_projection_last(O[1], nq, ds[0].data(), obs.data(), u.data(), v.data(), u.data(), nullptr,
                                                nullptr, oct.data(), ang.data(), 15.f, 0, 0, sf, 1, tb2.data(), tm2.data());
    };
    const int a = g(), b = o();
    CHECK(a == b && a > 100 && tm == tm2 && tb == tb2);
    float pr[4];
    g();
    vsg_debug_call_profile(pr);
    char extra[256];
    snprintf(extra, sizeof extra, ", \"matches\": %d, \"queries\": %d, \"phases_us\": {\"fill\": %.1f, \"launch\": %.1f, \"sync\": %.1f, \"total\": %.1f}",
             a, nq, pr[0], pr[1], pr[2], pr[3]);
    emit("SearchByProjection_last_frame", time_ms([&]() { g(); }, N), time_ms([&]() { reset(); o(); }, N / 4), extra);
  }
  // ---- SearchByProjection(F, local map points, th = 1)   Tracking.cc:3493
  {
    auto g = [&]() {
      reset();
      return vsg_frame_search_by_projection(F[1], nq, ds[0].data(), obs.data(), obs.data(), u.data(), v.data(), u.data(),
                                            oct.data(), vc.data(), nullptr, nullptr, nullptr, nullptr, nullptr, 1.f, 0.8f, sf,
                                            8, nullptr, nullptr, tb.data(), tm.data());
    };
    auto o = [&]() {
      return or_frame_search_by_projection(O[1], nq, ds[0].data(), obs.data(), obs.data(), u.data(), v.data(), u.data(),
                                           oct.data(), vc.data(), nullptr, nullptr, nullptr, nullptr, nullptr, 1.f, 0.8f, sf,
                                           nullptr, nullptr, tb2.data(), tm2.data());
    };
    const int a = g(), b = o();
    CHECK(a == b && a > 100 && tm == tm2 && tb == tb2);
    emit("SearchByProjection_local_map", time_ms([&]() { g(); }, N), time_ms([&]() { reset(); o(); }, N / 4));
  }
  // ---- SearchByProjection(KeyFrame, Sim3, vpPoints, vpMatched, th = 10, ratioHamming = 1)   LoopClosing.cc:735
  {
    auto g = [&]() {
      std::fill(tm.begin(), tm.end(), -1);
      return vsg_frame_search_by_projection_sim3(F[1], nq, ds[0].data(), u.data(), v.data(), r10.data(), oct.data(), 1.f, tm.data());
    };
    auto o = [&]() {
      std::fill(tm2.begin(), tm2.end(), -1);
      return or_kf_search_by_projection_sim3(O[1], nq, ds[0].data(), u.data(), v.data(), r10.data(), oct.data(), 1.f, tm2.data());
    };
    const int a = g(), b = o();
    CHECK(a == b && a > 100 && tm == tm2);
    emit("SearchByProjection_keyframe_sim3", time_ms([&]() { g(); }, N), time_ms([&]() { o(); }, N / 4));
  }
  // ---- SearchByProjection(Frame, KeyFrame, sAlreadyFound, th = 10, ORBdist = 100)   Tracking.cc:3805
  {
    auto g = [&]() {
      reset();
      return vsg_frame_search_by_projection_kf(F[1], nq, ds[0].data(), u.data(), v.data(), r10.data(), oct.data(), ang.data(),
                                               100, 1, tb.data(), tm.data());
    };
    auto o = [&]() {
      return or_frame_search_by_projection_kf(O[1], nq, ds[0].data(), u.data(), v.data(), r10.data(), oct.data(), ang.data(),
                                              100, 1, tb2.data(), tm2.data());
    };
    const int a = g(), b = o();
    CHECK(a == b && a > 100 && tm == tm2 && tb == tb2);
    emit("SearchByProjection_frame_keyframe", time_ms([&]() { g(); }, N), time_ms([&]() { reset(); o(); }, N / 4));
  }
  // ---- SearchBySim3(KF1, KF2, th = 7.5)   LoopClosing.cc:944
  {
    std::vector<int32_t> i1(nq), i2(nt), oc2(nt), m12(nq), m12o(nq);
    std::vector<float> u2(nt), v2(nt), r2(nt), r1(nq);
    for (int i = 0; i < nq; i++) i1[i] = i, r1[i] = 7.5f * sf[oct[i]];
    for (int i = 0; i < nt; i++)
      i2[i] = i, u2[i] = kp[1][i].x + 3.f, v2[i] = kp[1][i].y + 2.f, oc2[i] = kp[1][i].octave, r2[i] = 7.5f * sf[oc2[i]];
    auto g = [&]() {
      return vsg_frame_search_by_sim3(F[0], F[1], nq, i1.data(), ds[0].data(), u.data(), v.data(), r1.data(), oct.data(), nt,
                                      i2.data(), ds[1].data(), u2.data(), v2.data(), r2.data(), oc2.data(), m12.data());
    };
    auto o = [&]() {
      return or_kf_search_by_sim3(O[0], O[1], nq, i1.data(), ds[0].data(), u.data(), v.data(), r1.data(), oct.data(), nt,
                                  i2.data(), ds[1].data(), u2.data(), v2.data(), r2.data(), oc2.data(), m12o.data());
    };
    const int a = g(), b = o();
    CHECK(a == b && a > 100 && m12 == m12o);
    emit("SearchBySim3", time_ms([&]() { g(); }, N), time_ms([&]() { o(); }, N / 4));
  }
  // ---- Fuse(KF, vpMapPoints, th = 3)   LocalMapping.cc:770
  {
    std::vector<int32_t> bi(nq), bd(nq), bi2(nq), bd2(nq), act(nq), oth(nq), act2(nq), oth2(nq), qmp(nq);
    std::vector<int32_t> slot(nt), slot2(nt), mobs(nq + nt), mobs2(nq + nt);
    std::vector<uint8_t> bad(nq + nt), bad2(nq + nt);
    for (int i = 0; i < nq; i++) qmp[i] = i;
    auto rs = [&]() {
      for (int i = 0; i < nt; i++) slot[i] = slot2[i] = (i % 2) ? nq + i : -1;
      std::fill(mobs.begin(), mobs.end(), 2), std::fill(mobs2.begin(), mobs2.end(), 2);
      std::fill(bad.begin(), bad.end(), 0), std::fill(bad2.begin(), bad2.end(), 0);
    };
    auto g = [&]() {
      rs();
      vsg_frame_fuse(F[1], nq, ds[0].data(), u.data(), v.data(), u.data(), r3.data(), oct.data(), 0, inv2, 8, bi.data(), bd.data());
      return vsg_fuse_decide(nq, qmp.data(), bi.data(), bd.data(), 0, slot.data(), nt, mobs.data(), bad.data(), nq + nt,
                             act.data(), oth.data());
    };
    auto o = [&]() {
      return or_kf_fuse(O[1], nq, qmp.data(), ds[0].data(), u.data(), v.data(), u.data(), r3.data(), oct.data(), 0, inv2,
                        slot2.data(), mobs2.data(), bad2.data(), bi2.data(), bd2.data(), act2.data(), oth2.data());
    };
    const int a = g(), b = o();
    CHECK(a == b && a > 100 && bi == bi2 && bd == bd2 && act == act2 && slot == slot2);
    emit("Fuse", time_ms([&]() { g(); }, N), time_ms([&]() { rs(); o(); }, N / 4));
  }
  // ---- SearchForInitialization(F1, F2, vbPrevMatched, windowSize = 100)   Tracking.cc:2556
  {
    std::vector<float> px(nq), py(nq);
    std::vector<int32_t> m12(nq), m12o(nq);
    for (int i = 0; i < nq; i++) px[i] = kp[0][i].x, py[i] = kp[0][i].y;
    auto g = [&]() { return vsg_frame_search_for_initialization(F[0], F[1], px.data(), py.data(), 100, 0.9f, 1, m12.data()); };
    auto o = [&]() { return or_frame_search_for_initialization(O[0], O[1], px.data(), py.data(), 100, 0.9f, 1, m12o.data()); };
    const int a = g(), b = o();
    CHECK(a == b && a > 50 && m12 == m12o);
    emit("SearchForInitialization", time_ms([&]() { g(); }, N), time_ms([&]() { o(); }, N / 4));
  }
  // ---- the per-keyframe routines on resident frames at the reference's vocabulary scale (k = 10, L = 6, levelsup 4):
  // Frame::ComputeBoW (Frame.cc:882-889), SearchByBoW(KF, F) (ORBmatcher.cc:226-428; TrackReferenceKeyFrame,
  // Tracking.cc:2838), SearchForTriangulation (ORBmatcher.cc:902-1146; LocalMapping.cc:389)
  {
    std::vector<uint8_t> blob(vsg_synth_vocabulary(10, 6, 7, 0, 0, 0.02, nullptr, 0));
    CHECK(vsg_synth_vocabulary(10, 6, 7, 0, 0, 0.02, blob.data(), blob.size()) == blob.size());
    vsg_vocab *voc = nullptr;
    CHECK(vsg_vocab_load(0, blob.data(), blob.size(), &voc) == VSG_OK);
    OrVocab *ovoc = or_vocab_load(blob.data(), blob.size());
    CHECK(ovoc != nullptr);
    struct FV {
      std::vector<int32_t> node, off, idx, bid;
      std::vector<double> bval;
      int n = 0, nb = 0;
      explicit FV(int cap) : node(cap + 1), off(cap + 2), idx(cap + 1), bid(cap + 1), bval(cap + 1) {}
    };
    FV g0(cap), g1(cap), o0(cap), o1(cap);
    auto bow_g = [&](int t, FV &f) {
      return vsg_frame_bow_transform(voc, F[t], 4, f.bid.data(), f.bval.data(), cap, &f.nb, f.node.data(), f.off.data(),
                                     f.idx.data(), cap, &f.n, nullptr, nullptr, nullptr);
    };
    auto bow_o = [&](int t, FV &f) {
      return or_vocab_transform(ovoc, ds[t].data(), n[t], 4, f.bid.data(), f.bval.data(), cap, &f.nb, f.node.data(),
                                f.off.data(), f.idx.data(), cap, &f.n, nullptr, nullptr, nullptr);
    };
    CHECK(bow_g(0, g0) == VSG_OK && bow_g(1, g1) == VSG_OK && bow_o(0, o0) == 0 && bow_o(1, o1) == 0);
    CHECK(g0.n == o0.n && g0.nb == o0.nb && g0.n > 50 && g0.node == o0.node && g0.off == o0.off && g0.idx == o0.idx &&
          g0.bid == o0.bid && !memcmp(g0.bval.data(), o0.bval.data(), (size_t)g0.nb * 8) && g1.node == o1.node && g1.idx == o1.idx);
    char extra[128];
    snprintf(extra, sizeof extra, ", \"feature_vector_nodes\": %d, \"words\": %d, \"vocabulary_nodes\": 1111111", g0.n, g0.nb);
    emit("ComputeBoW_resident_k10_L6", time_ms([&]() { bow_g(0, g0); }, N), time_ms([&]() { bow_o(0, o0); }, N / 4), extra);
    std::vector<uint8_t> valid(nq, 1);
    std::vector<float> a0(nq), a1(nt);
    for (int i = 0; i < nq; i++) a0[i] = kp[0][i].angle;
    for (int i = 0; i < nt; i++) a1[i] = kp[1][i].angle;
    std::vector<int32_t> mf(nt), mfo(nt);
    auto sg = [&]() {
      return vsg_frame_search_by_bow_kf_f(F[0], valid.data(), g0.node.data(), g0.off.data(), g0.idx.data(), g0.n, F[1],
                                          g1.node.data(), g1.off.data(), g1.idx.data(), g1.n, 0.7f, 1, mf.data());
    };
    auto so = [&]() {
      return or_search_by_bow_kf_f(ds[0].data(), a0.data(), valid.data(), nq, o0.node.data(), o0.off.data(), o0.idx.data(), o0.n,
                                   ds[1].data(), a1.data(), nt, o1.node.data(), o1.off.data(), o1.idx.data(), o1.n, 0.7f, 1,
                                   mfo.data());
    };
    {
      const int a = sg(), b = so();
      CHECK(a == b && a > 50 && mf == mfo);
      snprintf(extra, sizeof extra, ", \"matches\": %d", a);
      emit("SearchByBoW_resident_k10_L6", time_ms([&]() { sg(); }, N), time_ms([&]() { so(); }, N / 4), extra);
    }
    std::vector<int32_t> m12(nq), m12o(nq);
    std::vector<uint8_t> e2(nt, 1);
    auto tg = [&]() {
      return vsg_frame_search_for_triangulation(F[0], valid.data(), g0.node.data(), g0.off.data(), g0.idx.data(), g0.n, F[1],
                                                e2.data(), g1.node.data(), g1.off.data(), g1.idx.data(), g1.n, nullptr, nullptr,
                                                1, m12.data());
    };
    auto to = [&]() {
      return or_search_for_triangulation(ds[0].data(), a0.data(), valid.data(), nq, o0.node.data(), o0.off.data(), o0.idx.data(),
                                         o0.n, ds[1].data(), a1.data(), e2.data(), nt, o1.node.data(), o1.off.data(),
                                         o1.idx.data(), o1.n, nullptr, nullptr, 1, m12o.data());
    };
    {
      const int a = tg(), b = to();
      CHECK(a == b && a > 50 && m12 == m12o);
      snprintf(extra, sizeof extra, ", \"matches\": %d", a);
      emit("SearchForTriangulation_resident_k10_L6", time_ms([&]() { tg(); }, N), time_ms([&]() { to(); }, N / 4), extra);
    }
    vsg_vocab_destroy(voc);
    or_vocab_destroy(ovoc);
  }
  // ---- Frame::ComputeStereoMatches (Frame.cc:957-1127) on two resident eyes: a rectified pair cut out of one wider
  // synthetic frame, the right eye 17 px further along the scene
  {
    std::vector<uint8_t> wide((size_t)(W + 64) * H), imL((size_t)W * H), imR((size_t)W * H);
    CHECK(vsg_synth_sequence_frame(W + 64, H, 41, 0, 1, 6, wide.data(), W + 64) == 0);
    for (int r = 0; r < H; r++) {
      memcpy(&imL[(size_t)r * W], &wide[(size_t)r * (W + 64) + 16], W);
      memcpy(&imR[(size_t)r * W], &wide[(size_t)r * (W + 64) + 16 + 17], W);
    }
    vsg_orb *exr = nullptr;
    CHECK(vsg_orb_create(1000, 1.2f, 8, 20, 7, 0, 1, &exr) == VSG_OK);
    OrExtractor *ol = or_create(1000, 1.2f, 8, 20, 7), *orr = or_create(1000, 1.2f, 8, 20, 7);
    std::vector<vsg_keypoint> kl(cap), kr(cap);
    std::vector<uint8_t> dl(cap * 32), dr(cap * 32);
    int nl = 0, nr = 0, onl = 0, onr = 0;
    CHECK(vsg_orb_extract(ex, imL.data(), H, W, W, 0, 0, kl.data(), dl.data(), cap, &nl) >= 0);
    CHECK(vsg_orb_extract(exr, imR.data(), H, W, W, 0, 0, kr.data(), dr.data(), cap, &nr) >= 0);
    std::vector<OrKeyPoint> okl(cap), okr(cap);
    std::vector<uint8_t> odl(cap * 32), odr(cap * 32);
    or_extract(ol, imL.data(), H, W, W, 0, 0, okl.data(), odl.data(), cap, &onl);
    or_extract(orr, imR.data(), H, W, W, 0, 0, okr.data(), odr.data(), cap, &onr);
    CHECK(nl == onl && nr == onr);
    vsg_frame *FL = nullptr, *FR = nullptr;
    CHECK(vsg_frame_create(0, cap, &FL) == VSG_OK && vsg_frame_create(0, cap, &FR) == VSG_OK);
    CHECK(vsg_frame_from_extractor(FL, ex, 0, kl.data(), nl, 0.f, 0.f, (float)W, (float)H) == VSG_OK);
    CHECK(vsg_frame_from_extractor(FR, exr, 0, kr.data(), nr, 0.f, 0.f, (float)W, (float)H) == VSG_OK);
    std::vector<float> ur(cap), dp(cap), our(cap), odp(cap);
    auto g = [&]() { return vsg_frame_stereo_matches(ex, 0, exr, 0, FL, FR, 0.05f, 40.f, ur.data(), dp.data()); };
    auto o = [&]() { or_stereo_matches(ol, orr, okl.data(), odl.data(), onl, okr.data(), odr.data(), onr, 0.05f, 40.f, our.data(), odp.data()); };
    const int a = g();
    o();
    CHECK(a > 100 && !memcmp(ur.data(), our.data(), (size_t)nl * 4) && !memcmp(dp.data(), odp.data(), (size_t)nl * 4));
    char extra[64];
    snprintf(extra, sizeof extra, ", \"stereo_matches\": %d", a);
    emit("ComputeStereoMatches_resident", time_ms([&]() { g(); }, N), time_ms([&]() { o(); }, N / 4), extra);
    vsg_frame_destroy(FL), vsg_frame_destroy(FR);
    vsg_orb_destroy(exr);
    or_destroy(ol), or_destroy(orr);
  }
  // ---- making a frame resident
  {
    const double up = time_ms([&]() { vsg_frame_upload(F[0], kp[0].data(), ds[0].data(), nullptr, n[0], -1, 0.f, 0.f, (float)W, (float)H); }, N);
    const double dd = time_ms([&]() { vsg_frame_from_extractor(F[1], ex, 0, kp[1].data(), n[1], 0.f, 0.f, (float)W, (float)H); }, N);
    const double og = time_ms([&]() {
      OrFrame *f = or_frame_create((const OrKeyPoint *)kp[0].data(), ds[0].data(), nullptr, n[0], -1, 0.f, 0.f, (float)W, (float)H);
      or_frame_destroy(f);
    }, N / 4);
    char b[256];
    snprintf(b, sizeof b, ", \"frame_upload_ms\": %.4f, \"frame_from_extractor_ms\": %.4f, \"oracle_assign_features_to_grid_ms\": %.4f", up, dd, og);
    js += b;
  }
  char tail[128];
  snprintf(tail, sizeof tail, ", \"arena_growths\": %d, \"iterations\": %d}", vsg_thread_arena_growths(0), N);
  js += tail;
  puts(js.c_str());
  for (int t = 0; t < 2; t++) vsg_frame_destroy(F[t]), or_frame_destroy(O[t]);
  vsg_orb_destroy(ex);
  return 0;
}
