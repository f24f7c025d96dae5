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


def _boundary_worker(rank, world, port, B, cap, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    counts, kps, desc = _fake_records(rank, B, cap)
    rec = sharding.record_bytes(cap)
    send = torch.zeros((1, rec), dtype=torch.uint8)
    recv = torch.zeros((1, rec), dtype=torch.uint8)
    ok = True
    for step in range(3):  # the record of this rank's LAST frame goes to the successor, the predecessor's arrives
        sharding.pack_records(send, counts[B - 1:], kps[B - 1:], desc[B - 1:])
        sharding.send_recv_boundary(recv, send, (rank + 1) % world, (rank - 1) % world)
        c, k, d = sharding.unpack_records(recv, cap)
        pc, pk, pd = _fake_records((rank - 1) % world, B, cap)
        ok &= torch.equal(c[0], pc[B - 1]) and torch.equal(k[0], pk[B - 1]) and torch.equal(d[0], pd[B - 1])
        ok &= not recv[0, 8:16].any()  # flags / pad words of the header are zero
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_boundary_exchange_gloo_world2():
    """The neighbour-only exchange (bench.py --exchange boundary): one record to the successor rank per step."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_boundary_worker, args=(r, 2, port, 3, 40, q)) for r in range(2)]
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


def test_record_layout_matches_the_c_abi():
    from visual_sgraphs_amd import orb
    L = orb.load_library()
    for cap in (1, 40, 1024, 1274, 2024):
        assert sharding.record_bytes(cap) == L.vsg_shard_record_bytes(cap)
        assert sharding.desc_offset(cap) == L.vsg_shard_record_desc_offset(cap)
        assert sharding.desc_offset(cap) % 16 == 0 and sharding.record_bytes(cap) % 64 == 0
    for s in range(4):
        for f in range(5):
            for w in (1, 2, 4, 6, 8):
                assert sharding.stream_to_rank(s, 4, w, f) == L.vsg_shard_stream_owner(s, 4, w, f)


def test_chunk_partition_and_predecessors():
    for world in (1, 2, 3, 8):
        seen = sum((sharding.chunk_frames(64, r, world) for r in range(world)), [])
        assert seen == list(range(64))
        for r in range(world):
            pr, off = sharding.predecessor_of_first(r, world)
            first = sharding.chunk_frames(64, r, world)[0]
            want = first - 1 if first > 0 else 64 - 1  # the previous batch's last frame for the very first one
            assert sharding.chunk_frames(64, pr, world)[-1] == want and off == (0 if r > 0 else -1)


def _seq_worker(rank, world, port, n_frames, q):
    """Each rank extracts its contiguous chunk of ONE global sequence (CPU oracle standing in for the device),
    all-gathers the records and matches its first frame against the gathered last frame of its predecessor rank;
    every other frame is matched locally.  The union must equal the single-process sequence bit for bit."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    import oracle_lib as ol
    from visual_sgraphs_amd import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = sharding.chunk_frames(n_frames, rank, world)
    ex = ol.OracleExtractor(300, 1.2, 4, 20, 7)
    cap = 300 + 3 * 4 + 16
    B = len(mine)
    counts = torch.zeros((B, 2), dtype=torch.int32)
    kps = torch.zeros((B, cap, 28), dtype=torch.uint8)
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8)
    local = []
    for i, t in enumerate(mine):
        mono, k, d = ex(synth.sequence_frame(160, 120, 9, t))
        local.append(d)
        counts[i, 0], counts[i, 1] = len(k), mono
        kps[i, :len(k)] = torch.from_numpy(k.view(np.uint8).reshape(len(k), 28))
        desc[i, :len(k)] = torch.from_numpy(d)
    rec = sharding.record_bytes(cap)
    send = torch.zeros((B, rec), dtype=torch.uint8)
    recv = torch.zeros((world * B, rec), dtype=torch.uint8)
    sharding.pack_records(send, counts, kps, desc)
    sharding.all_gather_records(recv, send)
    c, _, d = sharding.unpack_records(recv, cap)
    out = {}
    for i, t in enumerate(mine):
        if i > 0:
            prev = local[i - 1]
        elif rank > 0:
            pr, _ = sharding.predecessor_of_first(rank, world)
            row = pr * B + (B - 1)
            prev = d[row, :int(c[row, 0])].numpy()
        else:
            continue  # frame 0 has no predecessor in a single batch
        out[t] = [a.tolist() for a in ol.block_best2(local[i], prev)]
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_cross_rank_boundary_match_equals_single_process_sequence():
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    import oracle_lib as ol
    from visual_sgraphs_amd import synth
    n_frames, world = 6, 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_seq_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    [p.start() for p in procs]
    got = {}
    for _ in range(world):
        got.update(q.get(timeout=180)[1])
    [p.join(timeout=60) for p in procs]
    ex = ol.OracleExtractor(300, 1.2, 4, 20, 7)
    descs = [ex(synth.sequence_frame(160, 120, 9, t))[2] for t in range(n_frames)]
    assert sorted(got) == list(range(1, n_frames))
    for t in range(1, n_frames):
        want = [a.tolist() for a in ol.block_best2(descs[t], descs[t - 1])]
        assert got[t] == want, t
