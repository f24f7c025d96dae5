// vsg_ctx.h -- per-(host thread, device) execution context of the matcher / grid / BoW / stereo entry points.
//
// the ORBmatcher surface is "re-entrant, thread-safe, stream per calling thread" -- the reference
// constructs an ORBmatcher on the stack at each call site and calls it concurrently from the Tracking, LocalMapping
// and LoopClosing threads.  Every calling thread therefore owns, per device, ONE non-blocking HIP stream and ONE
// staging arena that only ever grows:
//   * a pinned, device-mapped host block: inputs are written there by the host and read by the kernels straight over
//     PCIe (zero copy), outputs are written there by the kernels -- a call is `fill -> one launch -> sync -> read`,
//     with no hipMalloc / hipFree / NULL-stream launch in steady state (a hipFree is a device-wide sync that would
//     stall the extractor's streams of another thread);
//   * a device block for scratch that must not cross PCIe.
// Contexts are never freed implicitly (HIP calls from thread-exit / static destructors are not safe once the runtime
// unloads); vsg_thread_release() frees the calling thread's contexts explicitly.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace vsg {

struct ThreadCtx {
  int device = -1;
  hipStream_t stream = nullptr;
  uint8_t *h_pin = nullptr;  // pinned + mapped host arena (the same address is valid on the device)
  uint8_t *d_pin = nullptr;  // device alias of h_pin
  size_t pin_cap = 0;
  uint8_t *d_buf = nullptr;  // device scratch arena
  size_t dev_cap = 0;
  unsigned long n_grow = 0;  // arena (re)allocations so far: constant in steady state (tests assert it)
  // monotonically increasing device counter from which the waves of k_window_search reserve their output segments;
  // the host tracks its value (never reset: no memset per call)
  uint32_t *d_counter = nullptr;
  uint32_t counter_base = 0;
};

// The calling thread's context on `device` (created on first use).  nullptr + VSG_ERR_* in *rc on failure.
ThreadCtx *thread_ctx(int device, int *rc);
// Make the arenas at least this large (grow = sync the stream, free, allocate 1.5x).  VSG_OK or VSG_ERR_HIP.
int ctx_reserve(ThreadCtx *c, size_t pinned_bytes, size_t device_bytes);

// Bump allocator over the pinned arena for one call: lay out the blocks first (sizes only), reserve once, then take
// the pointers.  Every block is 64-byte aligned.
struct Stage {
  size_t total = 0;
  size_t add(size_t bytes) {
    const size_t off = total;
    total += (bytes + 63) & ~(size_t)63;
    return off;
  }
};

// roctx ranges around stages and entry points (the REGISTER_TIMES analogue for profilers: SURVEY 5 "tracing";
// orb_slam3/include/Settings.h:23).  Off unless VSG_ROCTX=1 (one getenv at first use); the marker library
// (librocprofiler-sdk-roctx / libroctx64) is loaded lazily, so nothing depends on it.
void range_push(const char *name);
void range_pop();
struct Range {
  explicit Range(const char *name) { range_push(name); }
  ~Range() { range_pop(); }
};

// Device attributes that are per DEVICE, not per process (hipFuncSetAttribute of a >64 KB dynamic-LDS kernel): callers
// remember the limit they raised per device ordinal under this mutex-protected table.
enum { kMaxDevices = 64 };
bool lds_limit_ensure(int slot, int device, const void *func, size_t bytes);  // slot: one per kernel (0..7)

}  // namespace vsg
