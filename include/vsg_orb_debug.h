/* vsg_orb_debug.h -- test and measurement hooks of libvsg_orb.so.  NOT part of the drop-in boundary: nothing a maintainer
 * of the reference binds (INTEGRATION.md does not mention this header); the tests and tools/abi_latency.cpp include it.
 * The symbols are exported by the same library so that the tests exercise the shipped code object. */
#ifndef VSG_ORB_DEBUG_H
#define VSG_ORB_DEBUG_H
#include "vsg_orb.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test hook: sorts items[0..n) (n <= 2048) by their upper 32 bits with the device code DistributeOctTree uses for
 * `std::sort(vSizeAndPointerToNode...)` (ORBextractor.cc:707): a replay of libstdc++'s introsort whose result --
 * including the order of equal keys -- must equal std::sort's.  Lets tests compare the two directly. */
int vsg_debug_device_sort(int device, uint64_t *items, int n);

/* Measurement hook: wall time in microseconds of the calling thread's last vsg_frame_* window search -- {filling the
 * pinned arena, the launch call, the stream synchronisation (kernel + PCIe), the whole entry point} */
int vsg_debug_call_profile(float us[4]);

#ifdef __cplusplus
}
#endif
#endif
