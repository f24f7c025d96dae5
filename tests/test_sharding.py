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


# ------------------------------------------------------------------------------------------------------------------
# World 4 and 8: what a first 8-GPU run exercises for the first time, rehearsed on gloo with the CPU
# oracle standing in for the device -- the all-gather at that width, the boundary WRAP (rank 0's predecessor is the last
# rank's last frame of the PREVIOUS step) over several steps, and C5's two-ranks-per-stream ownership.
def _spawn(target, world, args, timeout=300):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args) + (q,)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=timeout) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    return res


import pytest  # noqa: E402


@pytest.mark.parametrize("world", [4, 8])
def test_all_gather_records_gloo_world_4_and_8(world):
    res = sorted(_spawn(_worker, world, (2, 24)))
    assert res == [(r, True) for r in range(world)]


@pytest.mark.parametrize("world", [4, 8])
def test_boundary_exchange_gloo_world_4_and_8(world):
    res = sorted(_spawn(_boundary_worker, world, (2, 24)))
    assert res == [(r, True) for r in range(world)]


def _steps_worker(rank, world, port, B, n_steps, exchange, q):
    """bench.py's multi-step protocol on gloo: step s deals frames [(s * world + r) * B, + B) of ONE global sequence to
    rank r; frames 1 .. B-1 of a chunk are matched locally at once, the chunk's FIRST frame one step later against the
    record the exchange of its own step delivered -- the predecessor rank's last frame, for rank 0 the LAST rank's last
    frame of the PREVIOUS step, kept across the step like bench.py's d_tail buffers."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    import oracle_lib as ol
    from visual_sgraphs_amd import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ex = ol.OracleExtractor(200, 1.2, 4, 20, 7)
    cap = 200 + 3 * 4 + 16
    rec = sharding.record_bytes(cap)
    pred_rank = (rank - 1) % world
    out, tail, first_prev = {}, None, None
    for s in range(n_steps):
        t0 = (s * world + rank) * B
        counts = torch.zeros((B, 2), dtype=torch.int32)
        kps = torch.zeros((B, cap, 28), dtype=torch.uint8)
        desc = torch.zeros((B, cap, 32), dtype=torch.uint8)
        local = []
        for i in range(B):
            mono, k, d = ex(synth.sequence_frame(160, 120, 11, t0 + i))
            local.append(d)
            counts[i, 0], counts[i, 1] = len(k), mono
            kps[i, :len(k)] = torch.from_numpy(k.view(np.uint8).reshape(len(k), 28))
            desc[i, :len(k)] = torch.from_numpy(d)
        for i in range(1, B):
            out[t0 + i] = [a.tolist() for a in ol.block_best2(local[i], local[i - 1])]
        if exchange == "allgather":
            send = torch.zeros((B, rec), dtype=torch.uint8)
            recv = torch.zeros((world * B, rec), dtype=torch.uint8)
            sharding.pack_records(send, counts, kps, desc)
            sharding.all_gather_records(recv, send)
            c, _, d = sharding.unpack_records(recv, cap)
            row = pred_rank * B + (B - 1)
            got_pred = d[row, :int(c[row, 0])].numpy().copy()           # the predecessor rank's last frame, THIS step
            lrow = (world - 1) * B + (B - 1)
            got_last = d[lrow, :int(c[lrow, 0])].numpy().copy()         # the last rank's last frame, THIS step
        else:
            send = torch.zeros((1, rec), dtype=torch.uint8)
            recv = torch.zeros((1, rec), dtype=torch.uint8)
            sharding.pack_records(send, counts[B - 1:], kps[B - 1:], desc[B - 1:])
            sharding.send_recv_boundary(recv, send, (rank + 1) % world, pred_rank)
            c, _, d = sharding.unpack_records(recv, cap)
            got_pred = d[0, :int(c[0, 0])].numpy().copy()
            got_last = got_pred                                          # rank 0 receives from the last rank
        # the chunk's first frame: rank > 0 against this step's record of rank - 1; rank 0 against the LAST step's tail
        if rank > 0:
            out[t0] = [a.tolist() for a in ol.block_best2(local[0], got_pred)]
        elif tail is not None:
            out[t0] = [a.tolist() for a in ol.block_best2(local[0], tail)]
        if rank == 0:
            tail = got_last  # what the NEXT step's first frame (the next global frame) is matched against
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,exchange", [(4, "allgather"), (8, "allgather"), (8, "boundary")])
def test_multi_step_sequence_with_the_boundary_wrap(world, exchange):
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    import oracle_lib as ol
    from visual_sgraphs_amd import synth
    B, n_steps = 2, 3
    got = {}
    for _, o in _spawn(_steps_worker, world, (B, n_steps, exchange)):
        got.update(o)
    n = n_steps * world * B
    ex = ol.OracleExtractor(200, 1.2, 4, 20, 7)
    descs = [ex(synth.sequence_frame(160, 120, 11, t))[2] for t in range(n)]
    assert sorted(got) == list(range(1, n))  # every frame but the very first has been matched exactly once
    for t in range(1, n):
        assert got[t] == [a.tolist() for a in ol.block_best2(descs[t], descs[t - 1])], t


def _c5_worker(rank, world, port, n_streams, n_frames, q):
    """Config C5 on `world` ranks: camera stream s, frame f belongs to rank stream_to_rank(s, n_streams, world, f) (two
    ranks per stream at world 8, taking its frames alternately).  Every rank extracts what it owns, the records are
    all-gathered (padded to the per-rank maximum), and a frame is matched against its predecessor of the SAME stream,
    which the other sharer of the stream extracted."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    import oracle_lib as ol
    from visual_sgraphs_amd import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    owned = [(s, f) for f in range(n_frames) for s in range(n_streams)
             if sharding.stream_to_rank(s, n_streams, world, f) == rank]
    per_rank = max(sum(1 for f in range(n_frames) for s in range(n_streams)
                       if sharding.stream_to_rank(s, n_streams, world, f) == r) for r in range(world))
    ex = ol.OracleExtractor(200, 1.2, 4, 20, 7)
    cap = 200 + 3 * 4 + 16
    counts = torch.full((per_rank, 2), -1, dtype=torch.int32)
    kps = torch.zeros((per_rank, cap, 28), dtype=torch.uint8)
    desc = torch.zeros((per_rank, cap, 32), dtype=torch.uint8)
    for i, (s, f) in enumerate(owned):
        mono, k, d = ex(synth.sequence_frame(160, 120, 40 + s, f))
        counts[i, 0], counts[i, 1] = len(k), mono
        kps[i, :len(k)] = torch.from_numpy(k.view(np.uint8).reshape(len(k), 28))
        desc[i, :len(k)] = torch.from_numpy(d)
    rec = sharding.record_bytes(cap)
    send = torch.zeros((per_rank, rec), dtype=torch.uint8)
    recv = torch.zeros((world * per_rank, rec), dtype=torch.uint8)
    sharding.pack_records(send, counts, kps, desc)
    sharding.all_gather_records(recv, send)
    c, _, d = sharding.unpack_records(recv, cap)

    def record(s, f):  # where (stream, frame) sits in the gathered buffer: owner rank, position in its owned list
        r = sharding.stream_to_rank(s, n_streams, world, f)
        pos = [(ss, ff) for ff in range(n_frames) for ss in range(n_streams)
               if sharding.stream_to_rank(ss, n_streams, world, ff) == r].index((s, f))
        row = r * per_rank + pos
        return d[row, :int(c[row, 0])].numpy()

    out = {}
    for (s, f) in owned:
        if f > 0:
            out[(s, f)] = [a.tolist() for a in ol.block_best2(record(s, f), record(s, f - 1))]
    q.put((rank, owned, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_c5_stream_ownership_two_ranks_per_stream(world):
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    import oracle_lib as ol
    from visual_sgraphs_amd import synth
    n_streams, n_frames = 4, 4
    res = _spawn(_c5_worker, world, (n_streams, n_frames))
    owned_all = sorted(sum((o for _, o, _ in res), []))
    assert owned_all == sorted((s, f) for s in range(n_streams) for f in range(n_frames))  # each frame exactly once
    by_rank = {r: o for r, o, _ in res}
    if world == 8:  # two ranks per stream: s and s + 4, alternating frames
        for s in range(n_streams):
            assert [f for ss, f in by_rank[s] if ss == s] == [0, 2] and [f for ss, f in by_rank[s + 4] if ss == s] == [1, 3]
            assert all(ss == s for ss, _ in by_rank[s] + by_rank[s + 4])
    got = {}
    for _, _, o in res:
        got.update(o)
    ex = ol.OracleExtractor(200, 1.2, 4, 20, 7)
    for s in range(n_streams):
        descs = [ex(synth.sequence_frame(160, 120, 40 + s, f))[2] for f in range(n_frames)]
        for f in range(1, n_frames):
            assert got[(s, f)] == [a.tolist() for a in ol.block_best2(descs[f], descs[f - 1])], (s, f)
