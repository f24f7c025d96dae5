// extract_latency.cpp -- N blocking single-frame operator() calls (640x480 / 1000) through the C ABI and their wall
// time; run under `rocprofv3 --kernel-trace` (tools/latency_timeline.sh) to see the GPU timeline of a call.
//   usage: extract_latency [calls = 300]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/vsg_orb.h"
#include "../include/vsg_synth.h"

int main(int argc, char **argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 300, W = 640, H = 480;
  std::vector<uint8_t> img((size_t)W * H);
  vsg_synth_sequence_frame(W, H, 1000, 0, 1, 6, img.data(), W);
  vsg_orb *ex = nullptr;
  if (vsg_orb_create(1000, 1.2f, 8, 20, 7, 0, 1, &ex) != VSG_OK) return 1;
  const int cap = vsg_orb_capacity(ex, H, W);
  std::vector<vsg_keypoint> kp(cap);
  std::vector<uint8_t> ds((size_t)cap * 32);
  int n = 0;
  for (int i = 0; i < 30; i++) vsg_orb_extract(ex, img.data(), H, W, W, 0, 0, kp.data(), ds.data(), cap, &n);
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < N; i++) vsg_orb_extract(ex, img.data(), H, W, W, 0, 0, kp.data(), ds.data(), cap, &n);
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / N;
  printf("{\"single_frame_ms\": %.4f, \"keypoints\": %d, \"graph_launches\": %ld}\n", ms, n, vsg_orb_chain_graph_launches(ex));
  vsg_orb_destroy(ex);
  return 0;
}
