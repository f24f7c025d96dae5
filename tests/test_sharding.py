"""world_size-2 gloo test (CPU) of the N>1 path: frame sharding + record all-gather + unpack."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from visual_sgraphs_amd import sharding


def test_shard_frames_partition():
    for world in (1, 2, 4, 8):
        seen = sorted(sum((sharding.shard_frames(37, r, world) for r in range(world)), []))
        assert seen == list(range(37))
        for r in range(world):
            for li, f in enumerate(sharding.shard_frames(37, r, world)):
                assert sharding.global_frame_index(r, li, world) == f


def _fake_records(rank, B, cap):
    g = torch.Generator().manual_seed(100 + rank)
    counts = torch.randint(0, cap, (B, 2), dtype=torch.int32, generator=g)
    kps = torch.randint(0, 256, (B, cap, 28), dtype=torch.uint8, generator=g)
    desc = torch.randint(0, 256, (B, cap, 32), dtype=torch.uint8, generator=g)
    return counts, kps, desc


def _worker(rank, world, port, B, cap, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    counts, kps, desc = _fake_records(rank, B, cap)
    rec = sharding.record_bytes(cap)
    send = torch.zeros((B, rec), dtype=torch.uint8)
    recv = torch.zeros((world * B, rec), dtype=torch.uint8)
    sharding.pack_records(send, counts, kps, desc)
    sharding.all_gather_records(recv, send)
    c, k, d = sharding.unpack_records(recv, cap)
    ok = True
    # the overlapped protocol of bench.py: two buffer pairs, a Work handle per exchange, wait before reuse
    sends = [send.clone(), send.clone()]
    recvs = [torch.zeros_like(recv), torch.zeros_like(recv)]
    pending = [None, None]
    for step in range(5):
        slot = step & 1
        if pending[slot] is not None:
            pending[slot].wait()
            ok &= torch.equal(recvs[slot], recv)
        sends[slot][:, 8:12] = sends[slot][:, 8:12]  # "refill" only after the wait
        pending[slot] = sharding.all_gather_records(recvs[slot], sends[slot], async_op=True)
        ok &= pending[slot] is not None
    for w in pending:
        w.wait()
    ok &= torch.equal(recvs[0], recv) and torch.equal(recvs[1], recv)
    for r in range(world):
        rc, rk, rd = _fake_records(r, B, cap)
        ok &= torch.equal(c[r * B:(r + 1) * B], rc) and torch.equal(k[r * B:(r + 1) * B], rk)
        ok &= torch.equal(d[r * B:(r + 1) * B], rd)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_records_gloo_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 3, 40, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(timeout=60) for p in procs]
    assert res == [(0, True), (1, True)]


def test_single_process_gather_is_identity():
    counts, kps, desc = _fake_records(0, 2, 10)
    send = torch.zeros((2, sharding.record_bytes(10)), dtype=torch.uint8)
    recv = torch.zeros_like(send)
    sharding.pack_records(send, counts, kps, desc)
    sharding.all_gather_records(recv, send)
    c, k, d = sharding.unpack_records(recv, 10)
    assert torch.equal(c, counts) and torch.equal(k, kps) and torch.equal(d, desc)
