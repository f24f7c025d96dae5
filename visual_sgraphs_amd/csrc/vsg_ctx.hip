// vsg_ctx.hip -- per-(host thread, device) streams and staging arenas (see vsg_ctx.h).
#include "vsg_ctx.h"

#include <dlfcn.h>
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "../../include/vsg_orb.h"

namespace vsg {

namespace {
struct ThreadCtxSet {
  std::vector<ThreadCtx *> by_device;  // index = device ordinal
};
thread_local ThreadCtxSet t_ctx;

std::mutex g_lds_mutex;
size_t g_lds_limit[8][kMaxDevices];  // zero-initialised: "64 KB default" is applied on read
}  // namespace

ThreadCtx *thread_ctx(int device, int *rc) {
  if (rc) *rc = VSG_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    if (rc) *rc = VSG_ERR_NO_DEVICE;
    return nullptr;
  }
  if (hipSetDevice(device) != hipSuccess) {
    if (rc) *rc = VSG_ERR_NO_DEVICE;
    return nullptr;
  }
  if ((size_t)device < t_ctx.by_device.size() && t_ctx.by_device[device]) return t_ctx.by_device[device];
  ThreadCtx *c = new ThreadCtx();
  c->device = device;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc((void **)&c->d_counter, 256) != hipSuccess ||
      // on the thread's own (non-blocking) stream: the NULL stream's hipMemset is not ordered against it
      hipMemsetAsync(c->d_counter, 0, 256, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
    if (c->stream) hipStreamDestroy(c->stream);
    if (c->d_counter) hipFree(c->d_counter);
    delete c;
    if (rc) *rc = VSG_ERR_HIP;
    return nullptr;
  }
  if (t_ctx.by_device.size() <= (size_t)device) t_ctx.by_device.resize(device + 1, nullptr);
  t_ctx.by_device[device] = c;
  return c;
}

int ctx_reserve(ThreadCtx *c, size_t pinned_bytes, size_t device_bytes) {
  if (!c) return VSG_ERR_INVALID;
  if (pinned_bytes > c->pin_cap) {
    if (hipStreamSynchronize(c->stream) != hipSuccess) return VSG_ERR_HIP;
    if (c->h_pin) hipHostFree(c->h_pin);
    c->h_pin = c->d_pin = nullptr;
    c->pin_cap = 0;
    size_t cap = pinned_bytes + pinned_bytes / 2;
    cap = (cap + (1u << 16) - 1) & ~(size_t)((1u << 16) - 1);
    void *p = nullptr, *dp = nullptr;
    // coherent (fine-grained) mapped host memory: kernel writes are visible to the host after the stream sync
    if (hipHostMalloc(&p, cap, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) return VSG_ERR_HIP;
    if (hipHostGetDevicePointer(&dp, p, 0) != hipSuccess) {
      hipHostFree(p);
      return VSG_ERR_HIP;
    }
    c->h_pin = (uint8_t *)p;
    c->d_pin = (uint8_t *)dp;
    c->pin_cap = cap;
    c->n_grow++;
  }
  if (device_bytes > c->dev_cap) {
    if (hipStreamSynchronize(c->stream) != hipSuccess) return VSG_ERR_HIP;
    if (c->d_buf) hipFree(c->d_buf);
    c->d_buf = nullptr;
    c->dev_cap = 0;
    size_t cap = device_bytes + device_bytes / 2;
    cap = (cap + (1u << 16) - 1) & ~(size_t)((1u << 16) - 1);
    void *p = nullptr;
    if (hipMalloc(&p, cap) != hipSuccess) return VSG_ERR_HIP;
    c->d_buf = (uint8_t *)p;
    c->dev_cap = cap;
    c->n_grow++;
  }
  return VSG_OK;
}

namespace {
struct Roctx {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
  bool on = false;
};
Roctx g_roctx;
std::once_flag g_roctx_once;
const Roctx &roctx() {
  std::call_once(g_roctx_once, [] {
    const char *e = getenv("VSG_ROCTX");
    if (!e || !*e || *e == '0') return;
    for (const char *name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
      void *lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (!lib) continue;
      g_roctx.push = (int (*)(const char *))dlsym(lib, "roctxRangePushA");
      g_roctx.pop = (int (*)())dlsym(lib, "roctxRangePop");
      if (g_roctx.push && g_roctx.pop) {
        g_roctx.on = true;
        return;
      }
    }
  });
  return g_roctx;
}
}  // namespace

void range_push(const char *name) {
  const Roctx &r = roctx();
  if (r.on) r.push(name);
}
void range_pop() {
  const Roctx &r = roctx();
  if (r.on) r.pop();
}

bool lds_limit_ensure(int slot, int device, const void *func, size_t bytes) {
  if (bytes <= 64 * 1024) return true;
  if (slot < 0 || slot >= 8 || device < 0 || device >= kMaxDevices) return false;
  std::lock_guard<std::mutex> lk(g_lds_mutex);
  if (bytes <= g_lds_limit[slot][device]) return true;
  if (hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
  g_lds_limit[slot][device] = bytes;
  return true;
}

}  // namespace vsg

extern "C" {

int vsg_thread_release(void) {
  using namespace vsg;
  for (ThreadCtx *&c : t_ctx.by_device) {
    if (!c) continue;
    if (hipSetDevice(c->device) == hipSuccess) {
      hipStreamSynchronize(c->stream);
      if (c->h_pin) hipHostFree(c->h_pin);
      if (c->d_buf) hipFree(c->d_buf);
      if (c->d_counter) hipFree(c->d_counter);
      hipStreamDestroy(c->stream);
    }
    delete c;
    c = nullptr;
  }
  return VSG_OK;
}

// device-to-device copy of raw pointers on a caller stream: hosts that hold records as device pointers (the gathered
// records of vsg_shard_record, bench.py's boundary state) need no second HIP runtime binding for it
int vsg_copy_d2d_async(int device, void *dst, const void *src, size_t bytes, void *stream) {
  if (!dst || !src) return VSG_ERR_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return VSG_ERR_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return VSG_ERR_NO_DEVICE;
  if (bytes == 0) return VSG_OK;
  return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? VSG_OK : VSG_ERR_HIP;
}

int vsg_thread_arena_growths(int device) {
  using namespace vsg;
  if (device < 0 || (size_t)device >= t_ctx.by_device.size() || !t_ctx.by_device[device]) return 0;
  return (int)t_ctx.by_device[device]->n_grow;
}

}  // extern "C"
