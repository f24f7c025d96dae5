// vsg_shard.hip -- frame sharding over the GPUs of one node: RCCL all-gather of keypoint / descriptor records over
// xGMI, inside the C ABI (BASELINE north_star: "multi-camera / batched-sequence extraction shards frames
// across the 8 GPUs of one node with RCCL all-gather of keypoint/descriptor buffers").
//
// The reference has no distributed layer; extraction has no cross-frame state, so frames (or camera streams) are dealt
// to ranks with NO data-path collective.  Only matching frame t against t-1 needs the neighbour's features: ranks
// exchange fixed-capacity per-frame records
//     { int32 n, int32 monoIndex, uint32 flags, 4 B pad | KeyPoint[cap] (28 B each, padded to 16) | uint8 desc[cap][32] | pad to 64 }
// (the descriptor block is 16-byte aligned, so a gathered record feeds the matcher kernels where it lies)
// with ONE ncclAllGather per batch on the caller's stream (one process per GPU; the 8-GPU node is fully connected, a
// record batch is tens of MB, so the collective is per-link bandwidth bound and is meant to run under the next
// batch's kernels).  RCCL is loaded lazily (dlopen: the copy already in the process -- e.g. PyTorch's -- or the
// system one), so the library itself does not depend on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>

#include <cstring>
#include <mutex>
#include <string>

#include "../../include/vsg_orb.h"
#include "vsg_common.h"

namespace {

struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  bool ok = false;
};
// record layout for a capacity
__host__ __device__ inline size_t off_kps() { return 16; }
__host__ __device__ inline size_t off_desc(int cap) { return 16 + (((size_t)cap * 28 + 15) & ~(size_t)15); }
__host__ __device__ inline size_t rec_bytes(int cap) { return (off_desc(cap) + (size_t)cap * 32 + 63) & ~(size_t)63; }

Rccl g_rccl;
std::once_flag g_rccl_once;
thread_local std::string t_serr;

const Rccl &rccl() {
  std::call_once(g_rccl_once, [] {
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (g_rccl.lib) break;
    }
    if (!g_rccl.lib) return;
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(g_rccl.lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(g_rccl.lib, "ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(g_rccl.lib, "ncclCommDestroy");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(g_rccl.lib, "ncclAllGather");
    g_rccl.Send = (decltype(g_rccl.Send))dlsym(g_rccl.lib, "ncclSend");
    g_rccl.Recv = (decltype(g_rccl.Recv))dlsym(g_rccl.lib, "ncclRecv");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))dlsym(g_rccl.lib, "ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))dlsym(g_rccl.lib, "ncclGroupEnd");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(g_rccl.lib, "ncclGetErrorString");
    g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(g_rccl.lib, "ncclCommCount");
    g_rccl.CommUserRank = (decltype(g_rccl.CommUserRank))dlsym(g_rccl.lib, "ncclCommUserRank");
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllGather;
  });
  return g_rccl;
}

// records of this rank's batch: {n, monoIndex, flags} + n keypoints + n descriptors per frame into the fixed-capacity
// layout (rows beyond n are never read by the receiver).  The launch covers ALL `frames` slots of the send block: slots
// at and beyond `nframes` (a partial batch, e.g. the last one of a sequence) are stamped n = 0 / monoIndex = 0 so that
// no rank ever receives the previous batch's records as valid neighbours.  A frame with more keypoints than the record
// holds is truncated to `cap`, its monoIndex is clamped to the same bound and header word 2 carries
// VSG_SHARD_FLAG_TRUNCATED.
__global__ __launch_bounds__(256) void k_pack_records(const int *__restrict__ counts, const vsg::KeyPointPOD *__restrict__ kps,
                                                      const uint8_t *__restrict__ desc, int src_cap, int cap, int nframes,
                                                      uint8_t *__restrict__ send, size_t rec) {
  const int f = blockIdx.y;
  uint32_t *r = (uint32_t *)(send + (size_t)f * rec);
  if (f >= nframes) {
    if (blockIdx.x == 0 && threadIdx.x < 4) r[threadIdx.x] = 0u;
    return;
  }
  const int n_src = counts[2 * f];
  const int n = n_src > cap ? cap : n_src;
  if (blockIdx.x == 0 && threadIdx.x < 4) {
    const int mono = counts[2 * f + 1];
    const uint32_t hdr[4] = {(uint32_t)n, (uint32_t)(mono > n ? n : mono), n_src > cap ? 1u : 0u, 0u};
    r[threadIdx.x] = hdr[threadIdx.x];
  }
  const uint32_t *sk = (const uint32_t *)(kps + (size_t)f * src_cap);
  const uint32_t *sd = (const uint32_t *)(desc + (size_t)f * src_cap * 32);
  uint32_t *dk = (uint32_t *)((uint8_t *)r + off_kps()), *dd = (uint32_t *)((uint8_t *)r + off_desc(cap));
  const int nk = n * 7, nd = n * 8, stride = gridDim.x * 256;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nd; i += stride) {
    dd[i] = sd[i];
    if (i < nk) dk[i] = sk[i];
  }
}

}  // namespace

struct vsg_shard {
  int device = 0, rank = 0, world = 1, cap = 0, frames = 0;
  size_t rec = 0;
  ncclComm_t comm = nullptr;
  uint8_t *d_send = nullptr, *d_recv = nullptr;  // [frames][rec], [world][frames][rec]
  uint8_t *d_bsend = nullptr, *d_brecv = nullptr;  // one record each: the boundary exchange (vsg_shard_send_recv_boundary)
};

#define S_HIP(expr)                                                          \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess) {                                                  \
      t_serr = std::string(#expr) + ": " + hipGetErrorString(_e);            \
      return VSG_ERR_HIP;                                                    \
    }                                                                        \
  } while (0)

extern "C" {

const char *vsg_shard_last_error(void) { return t_serr.c_str(); }

size_t vsg_shard_record_bytes(int capacity) { return capacity > 0 ? rec_bytes(capacity) : 0; }
size_t vsg_shard_record_desc_offset(int capacity) { return capacity > 0 ? off_desc(capacity) : 0; }

int vsg_shard_frame_owner(int frame, int world) { return world > 0 && frame >= 0 ? frame % world : VSG_ERR_INVALID; }

int vsg_shard_stream_owner(int stream, int n_streams, int world, int frame) {
  if (n_streams < 1 || world < 1 || stream < 0 || frame < 0) return VSG_ERR_INVALID;
  const int s = stream % n_streams;
  if (world <= n_streams) return s % world;
  const int sharers = (world - s + n_streams - 1) / n_streams;  // ranks s, s + n_streams, ... below world
  return s + n_streams * (frame % sharers);
}

int vsg_shard_unique_id(uint8_t id[128]) {
  if (!id) return VSG_ERR_INVALID;
  const Rccl &R = rccl();
  if (!R.ok) {
    t_serr = "RCCL (librccl.so.1) could not be loaded";
    return VSG_ERR_UNSUPPORTED;
  }
  ncclUniqueId u;
  const ncclResult_t r = R.GetUniqueId(&u);
  if (r != ncclSuccess) {
    t_serr = std::string("ncclGetUniqueId: ") + (R.GetErrorString ? R.GetErrorString(r) : "error");
    return VSG_ERR_HIP;
  }
  static_assert(sizeof(u) == 128, "ncclUniqueId");
  memcpy(id, &u, 128);
  return VSG_OK;
}

int vsg_shard_create(int device, int rank, int world, const uint8_t id[128], int capacity, int frames_per_rank,
                     vsg_shard **out) {
  if (!out || !id || world < 1 || rank < 0 || rank >= world || capacity < 1 || frames_per_rank < 1) return VSG_ERR_INVALID;
  *out = nullptr;
  const Rccl &R = rccl();
  if (!R.ok) {
    t_serr = "RCCL (librccl.so.1) could not be loaded";
    return VSG_ERR_UNSUPPORTED;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return VSG_ERR_NO_DEVICE;
  S_HIP(hipSetDevice(device));
  vsg_shard *s = new vsg_shard();
  s->device = device, s->rank = rank, s->world = world, s->cap = capacity, s->frames = frames_per_rank;
  s->rec = vsg_shard_record_bytes(capacity);
  ncclUniqueId u;
  memcpy(&u, id, 128);
  const ncclResult_t r = R.CommInitRank(&s->comm, world, u, rank);
  if (r != ncclSuccess) {
    t_serr = std::string("ncclCommInitRank: ") + (R.GetErrorString ? R.GetErrorString(r) : "error");
    delete s;
    return VSG_ERR_HIP;
  }
  const size_t sb = s->rec * (size_t)frames_per_rank;
  if (hipMalloc((void **)&s->d_send, sb) != hipSuccess || hipMalloc((void **)&s->d_recv, sb * (size_t)world) != hipSuccess ||
      hipMemset(s->d_send, 0, sb) != hipSuccess || hipMalloc((void **)&s->d_bsend, s->rec) != hipSuccess ||
      hipMalloc((void **)&s->d_brecv, s->rec) != hipSuccess || hipMemset(s->d_brecv, 0, s->rec) != hipSuccess) {
    t_serr = "hipMalloc of the record buffers failed";
    vsg_shard_destroy(s);
    return VSG_ERR_HIP;
  }
  (void)hipStreamSynchronize(nullptr);  // the fills ran on the NULL stream; callers' streams may be non-blocking
  *out = s;
  return VSG_OK;
}

void vsg_shard_destroy(vsg_shard *s) {
  if (!s) return;
  hipSetDevice(s->device);
  hipDeviceSynchronize();
  if (s->comm && rccl().ok) rccl().CommDestroy(s->comm);
  hipFree(s->d_send), hipFree(s->d_recv), hipFree(s->d_bsend), hipFree(s->d_brecv);
  delete s;
}

int vsg_shard_all_gather(vsg_shard *s, const int *d_counts, const vsg_keypoint *d_kps, const uint8_t *d_desc,
                         int src_capacity, int nframes, void *stream) {
  if (!s || !d_counts || !d_kps || !d_desc || nframes < 1 || nframes > s->frames || src_capacity < 1) return VSG_ERR_INVALID;
  S_HIP(hipSetDevice(s->device));
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_pack_records, dim3(4, s->frames), dim3(256), 0, st, d_counts, (const vsg::KeyPointPOD *)d_kps, d_desc,
                     src_capacity, s->cap, nframes, s->d_send, s->rec);
  S_HIP(hipGetLastError());
  // every rank always sends its full [frames][rec] block, so that rank r's records sit at a fixed offset
  const ncclResult_t r = rccl().AllGather(s->d_send, s->d_recv, s->rec * (size_t)s->frames, ncclChar, s->comm, st);
  if (r != ncclSuccess) {
    t_serr = std::string("ncclAllGather: ") + (rccl().GetErrorString ? rccl().GetErrorString(r) : "error");
    return VSG_ERR_HIP;
  }
  return VSG_OK;
}

// The neighbour-only alternative to the all-gather for CHUNK-partitioned sequences: matching needs exactly one remote
// record per rank and batch -- the last frame of the predecessor rank -- so every rank sends the record of its frame
// `frame` to rank + 1 and receives its predecessor's (rank - 1, cyclic) into a one-record buffer: one ncclSend +
// ncclRecv pair per batch (61 KB at C2) instead of world x frames records into every GPU.
int vsg_shard_send_recv_boundary(vsg_shard *s, const int *d_counts, const vsg_keypoint *d_kps, const uint8_t *d_desc,
                                 int src_capacity, int frame, void *stream) {
  if (!s || !d_counts || !d_kps || !d_desc || frame < 0 || src_capacity < 1) return VSG_ERR_INVALID;
  const Rccl &R = rccl();
  if (!R.Send || !R.Recv || !R.GroupStart || !R.GroupEnd) {
    t_serr = "ncclSend / ncclRecv not available in the loaded RCCL";
    return VSG_ERR_UNSUPPORTED;
  }
  S_HIP(hipSetDevice(s->device));
  hipStream_t st = (hipStream_t)stream;
  // pack ONE record: frame `frame` of the caller's batch into slot 0 of the boundary send buffer
  hipLaunchKernelGGL(k_pack_records, dim3(4, 1), dim3(256), 0, st, d_counts + 2 * (size_t)frame,
                     (const vsg::KeyPointPOD *)d_kps + (size_t)frame * src_capacity, d_desc + (size_t)frame * src_capacity * 32,
                     src_capacity, s->cap, 1, s->d_bsend, s->rec);
  S_HIP(hipGetLastError());
  const int next = (s->rank + 1) % s->world, prev = (s->rank + s->world - 1) % s->world;
  ncclResult_t r = R.GroupStart();
  if (r == ncclSuccess) r = R.Send(s->d_bsend, s->rec, ncclChar, next, s->comm, st);
  if (r == ncclSuccess) r = R.Recv(s->d_brecv, s->rec, ncclChar, prev, s->comm, st);
  const ncclResult_t e = R.GroupEnd();
  if (r == ncclSuccess) r = e;
  if (r != ncclSuccess) {
    t_serr = std::string("ncclSend/ncclRecv: ") + (R.GetErrorString ? R.GetErrorString(r) : "error");
    return VSG_ERR_HIP;
  }
  return VSG_OK;
}

int vsg_shard_boundary_record(vsg_shard *s, const int **d_counts, const vsg_keypoint **d_kps, const uint8_t **d_desc) {
  if (!s) return VSG_ERR_INVALID;
  const uint8_t *r = s->d_brecv;
  if (d_counts) *d_counts = (const int *)r;
  if (d_kps) *d_kps = (const vsg_keypoint *)(r + off_kps());
  if (d_desc) *d_desc = r + off_desc(s->cap);
  return VSG_OK;
}

// Number of ranks of the communicator AS RCCL REPORTS IT (ncclCommCount on the live communicator, not the `world` the
// caller passed to ncclCommInitRank): bench lines report it as rccl_ranks_seen.  VSG_ERR_UNSUPPORTED when the loaded
// RCCL does not export ncclCommCount -- never an echo of the argument.
int vsg_shard_world(const vsg_shard *s) {
  if (!s || !s->comm) return VSG_ERR_INVALID;
  const Rccl &R = rccl();
  if (!R.CommCount) {
    t_serr = "the loaded RCCL does not export ncclCommCount";
    return VSG_ERR_UNSUPPORTED;
  }
  int n = 0;
  const ncclResult_t r = R.CommCount(s->comm, &n);
  if (r != ncclSuccess) {
    t_serr = std::string("ncclCommCount: ") + (R.GetErrorString ? R.GetErrorString(r) : "error");
    return VSG_ERR_HIP;
  }
  return n;
}

// This process's rank in the communicator as RCCL reports it (ncclCommUserRank).
int vsg_shard_rank(const vsg_shard *s) {
  if (!s || !s->comm) return VSG_ERR_INVALID;
  const Rccl &R = rccl();
  if (!R.CommUserRank) {
    t_serr = "the loaded RCCL does not export ncclCommUserRank";
    return VSG_ERR_UNSUPPORTED;
  }
  int n = -1;
  const ncclResult_t r = R.CommUserRank(s->comm, &n);
  if (r != ncclSuccess) {
    t_serr = std::string("ncclCommUserRank: ") + (R.GetErrorString ? R.GetErrorString(r) : "error");
    return VSG_ERR_HIP;
  }
  return n;
}

int vsg_shard_record(vsg_shard *s, int rank, int frame, const int **d_counts, const vsg_keypoint **d_kps,
                     const uint8_t **d_desc) {
  if (!s || rank < 0 || rank >= s->world || frame < 0 || frame >= s->frames) return VSG_ERR_INVALID;
  const uint8_t *r = s->d_recv + ((size_t)rank * s->frames + frame) * s->rec;
  if (d_counts) *d_counts = (const int *)r;
  if (d_kps) *d_kps = (const vsg_keypoint *)(r + off_kps());
  if (d_desc) *d_desc = r + off_desc(s->cap);
  return VSG_OK;
}

}  // extern "C"
