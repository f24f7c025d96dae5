"""Frame sharding across ranks and the keypoint/descriptor record exchange.

Extraction has no cross-frame state, so frames (or cameras) are partitioned over ranks with no data-path
collective.  Matching frame t against t-1 (or a keyframe) needs the neighbour's descriptors, so the ranks
exchange fixed-capacity per-frame records with ONE all-gather per batch (RCCL over xGMI on GPUs, gloo in
the CPU tests).  Record layout per frame, `record_bytes(cap)` bytes:
    int32 n, int32 monoIndex, KeyPoint[cap] (28 B each), uint8 desc[cap][32]
"""
import torch
import torch.distributed as dist


def record_bytes(cap):
    return 8 + cap * 60


def shard_frames(n_frames, rank, world):
    """Round-robin frame -> rank map (frame f goes to rank f % world); returns this rank's frame indices."""
    return list(range(rank, n_frames, world))


def stream_to_rank(stream, n_streams, world):
    """Multi-camera pinning (config C5): stream s -> rank s % world; with world > n_streams a stream's frames
    are further split round-robin over the world // n_streams ranks that share it."""
    return stream % world


def pack_records(send, counts, kps, desc):
    """counts [B,2] int32, kps [B,cap,28] uint8, desc [B,cap,32] uint8 -> send [B, record_bytes(cap)] uint8."""
    B, cap = kps.shape[0], kps.shape[1]
    send[:, :8].copy_(counts.contiguous().view(torch.uint8).view(B, 8))
    send[:, 8:8 + cap * 28].copy_(kps.reshape(B, cap * 28))
    send[:, 8 + cap * 28:].copy_(desc.reshape(B, cap * 32))
    return send


def unpack_records(recv, cap):
    """recv [R, record_bytes(cap)] -> (counts [R,2] int32, kps [R,cap,28] uint8, desc [R,cap,32] uint8)."""
    R = recv.shape[0]
    counts = recv[:, :8].contiguous().view(torch.int32).view(R, 2)
    kps = recv[:, 8:8 + cap * 28].reshape(R, cap, 28)
    desc = recv[:, 8 + cap * 28:].reshape(R, cap, 32)
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


def global_frame_index(rank, local_index, world):
    """Inverse of shard_frames: the original frame number of a gathered record."""
    return local_index * world + rank
