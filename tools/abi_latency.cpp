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
