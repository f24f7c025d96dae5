"""Frame sharding across ranks and the keypoint/descriptor record exchange (C ABI: vsg_shard_*).

Extraction has no cross-frame state, so frames (or cameras) are partitioned over ranks with no data-path
collective.  Matching frame t against t-1 (or a keyframe) needs the neighbour's descriptors, so the ranks
exchange fixed-capacity per-frame records with ONE all-gather per batch (RCCL over xGMI on GPUs -- through the
library's own communicator, `ShardComm`, or through torch.distributed -- and gloo in the CPU tests).  Record
layout per frame (the C ABI's, `record_bytes(cap)` bytes; the descriptor block is 16-byte aligned so that a
gathered record feeds the matcher kernels where it lies):
    int32 n, int32 monoIndex, uint32 flags (bit 0: truncated to cap), 4 B pad | KeyPoint[cap] (28 B each, padded to 16) | uint8 desc[cap][32] | pad to 64
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist


def kps_offset():
    return 16


def desc_offset(cap):
    return 16 + ((cap * 28 + 15) & ~15)


def record_bytes(cap):
    return (desc_offset(cap) + cap * 32 + 63) & ~63


def shard_frames(n_frames, rank, world):
    """Round-robin frame -> rank map (frame f goes to rank f % world); returns this rank's frame indices."""
    return list(range(rank, n_frames, world))


def global_frame_index(rank, local_index, world):
    """Inverse of shard_frames: the original frame number of a gathered record."""
    return local_index * world + rank


def chunk_frames(n_frames, rank, world):
    """Contiguous chunks of a batched sequence (SURVEY 8e "or contiguous chunks"): rank r owns frames
    [r * n / world, (r + 1) * n / world).  Inside a chunk every frame's predecessor is local; only the FIRST frame of
    a chunk needs a remote record -- the last frame of rank r - 1 (of the previous batch for rank 0)."""
    lo, hi = rank * n_frames // world, (rank + 1) * n_frames // world
    return list(range(lo, hi))


def predecessor_of_first(rank, world):
    """(rank, batch_offset) holding the predecessor of this rank's first frame of a batch: the last frame of rank
    r - 1 of the same batch, or of the last rank of the PREVIOUS batch (batch_offset = -1) for rank 0."""
    return (rank - 1, 0) if rank > 0 else (world - 1, -1)


def stream_to_rank(stream, n_streams, world, frame=0):
    """Multi-camera pinning (config C5): camera stream s of n_streams -> rank.
    world <= n_streams: stream s -> rank s % world (every frame of a stream on one rank).
    world >  n_streams: the ranks {s, s + n_streams, s + 2 n_streams, ...} < world share stream s and take its
    frames round-robin (4 streams on 8 GPUs: frame f of stream s -> rank s + 4 * (f % 2))."""
    s = stream % n_streams
    if world <= n_streams:
        return s % world
    sharers = (world - s + n_streams - 1) // n_streams  # ranks s, s + n_streams, ... below world
    return s + n_streams * (frame % sharers)


def pack_records(send, counts, kps, desc):
    """counts [B,2] int32, kps [B,cap,28] uint8, desc [B,cap,32] uint8 -> send [B, record_bytes(cap)] uint8."""
    B, cap = kps.shape[0], kps.shape[1]
    od = desc_offset(cap)
    send[:, :8].copy_(counts.contiguous().view(torch.uint8).view(B, 8))
    send[:, 8:16].zero_()  # flags: nothing is truncated here (the record capacity is the source capacity)
    send[:, 16:16 + cap * 28].copy_(kps.reshape(B, cap * 28))
    send[:, od:od + cap * 32].copy_(desc.reshape(B, cap * 32))
    return send


def unpack_records(recv, cap):
    """recv [R, record_bytes(cap)] -> (counts [R,2] int32, kps [R,cap,28] uint8, desc [R,cap,32] uint8)."""
    R = recv.shape[0]
    od = desc_offset(cap)
    counts = recv[:, :8].contiguous().view(torch.int32).view(R, 2)
    kps = recv[:, 16:16 + cap * 28].reshape(R, cap, 28)
    desc = recv[:, od:od + cap * 32].reshape(R, cap, 32)
    return counts, kps, desc


def all_gather_records(recv, send, async_op=False):
    """One collective per batch: recv [world*B, rec] <- every rank's send [B, rec] (rank-major).

    async_op=True (device tensors on the nccl = RCCL backend only) returns the collective's Work handle instead of
    making the caller's stream wait for it: the exchange of batch k then runs under the kernels of batch k+1.  The
    caller must `wait()` on the handle before it touches `recv` or refills `send`."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        if send.is_cuda and dist.get_backend() == "gloo":
            # CPU-side collective (tests / single-GPU dry runs of the multi-rank path): stage through the host
            r = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_gather_into_tensor(r, send.cpu())
            recv.copy_(r)
        elif async_op:
            return dist.all_gather_into_tensor(recv, send, async_op=True)
        else:
            dist.all_gather_into_tensor(recv, send)
    else:
        recv.copy_(send)
    return None


def send_recv_boundary(recv, send, next_rank, prev_rank):
    """The neighbour-only exchange through torch.distributed: `send` [1, rec] goes to next_rank, `recv` [1, rec] is filled
    from prev_rank (one batched isend / irecv pair; device tensors are staged through the host on gloo)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        recv.copy_(send)
        return
    via_host = send.is_cuda and dist.get_backend() == "gloo"
    s = send.cpu() if via_host else send
    r = torch.empty(recv.shape, dtype=recv.dtype) if via_host else recv
    for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, s, next_rank), dist.P2POp(dist.irecv, r, prev_rank)]):
        w.wait()
    if via_host:
        recv.copy_(r)


class ShardComm:
    """The library's own RCCL communicator (vsg_shard_*, include/vsg_orb.h): pack kernel + ncclAllGather on the
    caller's stream, gathered records addressable as device pointers.  The 128-byte ncclUniqueId is created by rank 0
    and handed to the other ranks by `exchange` (default: torch.distributed.broadcast when a process group exists)."""

    def __init__(self, device, rank, world, capacity, frames_per_rank, exchange=None):
        from . import orb
        self._L = orb.load_library()
        self._orb = orb
        uid = np.zeros(128, np.uint8)
        # every rank probes RCCL and its device BEFORE anything collective happens (ncclGetUniqueId is cheap; only rank
        # 0's id is used), and the ranks agree on the outcome: a one-sided failure would otherwise leave the others
        # blocked in the broadcast below or inside ncclCommInitRank
        probe = np.zeros(128, np.uint8)
        rc = self._L.vsg_shard_unique_id(probe.ctypes.data_as(C.POINTER(C.c_uint8)))
        if rc == 0 and not (0 <= int(device) < max(self._L.vsg_device_count(), 0)):
            rc = -4
        err = self._L.vsg_shard_last_error().decode() if rc != 0 else ""
        if world > 1 and exchange is None and dist.is_initialized():
            ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32)
            if dist.get_backend() == "nccl":
                ok = ok.cuda(device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and rc == 0:
                rc, err = -3, "another rank cannot create its communicator (RCCL or device unavailable there)"
        if rc != 0:
            raise orb.VsgError(rc, "vsg_shard_unique_id", err)
        if rank == 0:
            uid = probe
        if world > 1:
            if exchange is not None:
                uid = np.asarray(exchange(uid), np.uint8)
            else:
                t = torch.from_numpy(uid)
                if dist.get_backend() == "nccl":
                    t = t.cuda(device)
                dist.broadcast(t, src=0)
                uid = t.cpu().numpy()
        self._h = C.c_void_p()
        rc = self._L.vsg_shard_create(int(device), int(rank), int(world), uid.ctypes.data_as(C.POINTER(C.c_uint8)),
                                      int(capacity), int(frames_per_rank), C.byref(self._h))
        if rc != 0:
            raise orb.VsgError(rc, "vsg_shard_create", self._L.vsg_shard_last_error().decode())
        self.rank, self.world, self.capacity, self.frames = rank, world, capacity, frames_per_rank

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.vsg_shard_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def all_gather(self, d_counts, d_kps, d_desc, src_capacity, nframes, stream):
        """Device pointers (ints) of this rank's batch; asynchronous on `stream` (an int handle)."""
        rc = self._L.vsg_shard_all_gather(self._h, C.c_void_p(d_counts), C.c_void_p(d_kps), C.c_void_p(d_desc),
                                          int(src_capacity), int(nframes), C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise self._orb.VsgError(rc, "vsg_shard_all_gather", self._L.vsg_shard_last_error().decode())

    def world_seen(self):
        """Communicator size as RCCL reports it (ncclCommCount), not the world this object was created with."""
        n = self._L.vsg_shard_world(self._h)
        if n < 0:
            raise self._orb.VsgError(n, "vsg_shard_world", self._L.vsg_shard_last_error().decode())
        return n

    def rank_seen(self):
        """This process's rank as RCCL reports it (ncclCommUserRank)."""
        n = self._L.vsg_shard_rank(self._h)
        if n < 0:
            raise self._orb.VsgError(n, "vsg_shard_rank", self._L.vsg_shard_last_error().decode())
        return n

    def send_recv_boundary(self, d_counts, d_kps, d_desc, src_capacity, frame, stream):
        """Neighbour-only exchange: this rank's record of `frame` goes to rank + 1, the predecessor's arrives in the
        boundary slot (`boundary_record`).  Asynchronous on `stream`."""
        rc = self._L.vsg_shard_send_recv_boundary(self._h, C.c_void_p(d_counts), C.c_void_p(d_kps), C.c_void_p(d_desc),
                                                  int(src_capacity), int(frame), C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise self._orb.VsgError(rc, "vsg_shard_send_recv_boundary", self._L.vsg_shard_last_error().decode())

    def boundary_record(self):
        c, k, d = C.c_void_p(), C.c_void_p(), C.c_void_p()
        rc = self._L.vsg_shard_boundary_record(self._h, C.byref(c), C.byref(k), C.byref(d))
        if rc != 0:
            raise self._orb.VsgError(rc, "vsg_shard_boundary_record", "")
        return c.value, k.value, d.value

    def record(self, rank, frame):
        """(d_counts, d_kps, d_desc) device pointers of one gathered record."""
        c, k, d = C.c_void_p(), C.c_void_p(), C.c_void_p()
        rc = self._L.vsg_shard_record(self._h, int(rank), int(frame), C.byref(c), C.byref(k), C.byref(d))
        if rc != 0:
            raise self._orb.VsgError(rc, "vsg_shard_record", "")
        return c.value, k.value, d.value
