// Does hipPointerGetAttributes report PAGEABLE memory as host-registered after the runtime used it in a pageable
// hipMemcpyAsync (it may pin such buffers on its own)?  Measured on ROCm 7.0 / MI355X: NO -- type 0, null pointers before,
// during and after the copies; hipHostMalloc and hipHostRegister memory both report type 1 with device pointer == host
// pointer; after hipHostUnregister type 0 again.  So vsg_orb.hip's host_pinned() (type == host) cannot mistake pageable
// memory for pinned memory this way: one suspect less for the open issue in profiles/r04_q_open_issue_gpu_fault.txt.
//   hipcc --offload-arch=gfx950 -o tools/_bin/probe_runtime_pins tools/probe_runtime_pins.cpp && tools/_bin/probe_runtime_pins
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static void show(const char *what, const void *p) {
  hipPointerAttribute_t a;
  const hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    printf("%-44s %p: not known to the runtime (%s)\n", what, p, hipGetErrorName(e));
    return;
  }
  printf("%-44s %p: type %d host %p device %p\n", what, p, (int)a.type, a.hostPointer, a.devicePointer);
}
int main() {
  const size_t n = 8u << 20;
  unsigned char *h = (unsigned char *)malloc(n), *d = nullptr, *hp = nullptr;
  memset(h, 1, n);
  hipMalloc(&d, n);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  show("malloc'd buffer, untouched", h);
  hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s);
  show("... right after a pageable hipMemcpyAsync", h + 4096);
  hipStreamSynchronize(s);
  show("... after the stream is idle", h + 4096);
  hipMemcpy2DAsync(d, 1024, h, 2048, 1024, 1024, hipMemcpyHostToDevice, s);
  hipStreamSynchronize(s);
  show("... after a pageable hipMemcpy2DAsync", h + 4096);
  hipHostMalloc(&hp, n, hipHostMallocDefault);
  show("hipHostMalloc", hp + 4096);
  hipHostRegister(h, n, hipHostRegisterMapped | hipHostRegisterPortable);
  show("the malloc'd buffer after hipHostRegister", h + 4096);
  hipHostUnregister(h);
  show("... after hipHostUnregister", h + 4096);
  // a registration that starts and ends inside pages: what do the OTHER bytes of those pages report?
  unsigned char *r0 = h + 8192 + 100;
  const size_t rn = 5000;
  hipHostRegister(r0, rn, hipHostRegisterMapped | hipHostRegisterPortable);
  show("registered [page + 100, + 5000): inside", r0 + 10);
  show("  same first page, 50 bytes BEFORE the range", r0 - 50);
  show("  same last page, 10 bytes AFTER the range", r0 + rn + 10);
  show("  the next page", r0 + rn + 4096);
  hipHostUnregister(r0);
  return 0;
}
