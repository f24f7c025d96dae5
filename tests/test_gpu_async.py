"""GPU tests of the asynchronous host pipeline (vsg_orb_submit_batch / vsg_orb_wait), pinned caller memory, the
NULL-stream ordering of the device entry points and the one-copy pyramid read-back."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from visual_sgraphs_amd import orb, synth

pytestmark = pytest.mark.gpu

W, H, NF = 320, 240, 500


def _frames(seed, n):
    return np.stack([synth.sequence_frame(W, H, seed, t) for t in range(n)])


def _oracle(frames):
    ref = ol.OracleExtractor(NF, 1.2, 8, 20, 7)
    return [ref(f) for f in frames]


def _check(frames, kps, desc, n, mono):
    for f, (rm, rk, rd) in enumerate(_oracle(frames)):
        assert n[f] == len(rk) and mono[f] == rm
        assert kps[f, :n[f]].tobytes() == rk.tobytes() and np.array_equal(desc[f, :n[f]], rd)


@pytest.mark.parametrize("pinned", [False, True])
def test_submit_wait_three_batches_in_flight(pinned):
    B = 4
    ex = orb.ORBextractor(NF, 1.2, 8, 20, 7, max_batch=B)
    cap = ex.capacity(H, W)
    assert ex.slots() == 3
    batches = [_frames(20 + i, B) for i in range(5)]
    outs = [(np.zeros((B, cap), orb.KP_DTYPE), np.zeros((B, cap, 32), np.uint8)) for _ in batches]
    if pinned:
        batches = [orb.pin(b) for b in batches]
        outs = [(orb.pin(k), orb.pin(d)) for k, d in outs]
    tickets = [ex.submit_batch(batches[i], *outs[i]) for i in range(3)]
    with pytest.raises(orb.VsgError) as e:  # every slot holds an un-waited batch
        ex.submit_batch(batches[3], *outs[3])
    assert e.value.code == -7  # VSG_ERR_BUSY: nothing the caller could grow
    res = [ex.wait(tickets[0])]
    tickets.append(ex.submit_batch(batches[3], *outs[3]))  # the freed slot is reused while 1 and 2 are in flight
    res.append(ex.wait(tickets[1]))
    tickets.append(ex.submit_batch(batches[4], *outs[4]))
    res += [ex.wait(t) for t in tickets[2:]]
    for i in range(5):
        _check(batches[i], outs[i][0], outs[i][1], *res[i])
    with pytest.raises(orb.VsgError):
        ex._inflight[tickets[0]] = (B, None, None, None)
        ex.wait(tickets[0])  # a ticket can be waited for once
    if pinned:
        for b, (k, d) in zip(batches, outs):
            orb.unpin(b), orb.unpin(k), orb.unpin(d)


def test_host_alloc_buffers_take_the_direct_paths():
    """vsg_host_alloc (hipHostMalloc) memory behaves like registered memory: the device reads the batch from it and writes the
    n[f] records of every frame straight into it (rows beyond n[f] stay untouched: nothing was staged and copied)."""
    B = 3
    ex = orb.ORBextractor(NF, 1.2, 8, 20, 7, max_batch=B)
    cap = ex.capacity(H, W)
    fr = _frames(77, B)
    with orb.PinnedArray(fr.shape) as pin_in, orb.PinnedArray((B, cap), orb.KP_DTYPE) as pin_k, \
            orb.PinnedArray((B, cap, 32)) as pin_d:
        pin_in.a[...] = fr
        pin_d.a[...] = 0xAB
        for _ in range(3):
            n, mono = ex.wait(ex.submit_batch(pin_in.a, pin_k.a, pin_d.a))
            _check(fr, pin_k.a, pin_d.a, n, mono)
            for f in range(B):
                assert np.all(pin_d.a[f, n[f]:] == 0xAB)
        # the blocking single-frame call (latency path: the device reads the pinned image itself)
        mono, kps, desc = ex(pin_in.a[1])
        rm, rk, rd = ol.OracleExtractor(NF, 1.2, 8, 20, 7)(fr[1])
        assert mono == rm and kps.tobytes() == rk.tobytes() and np.array_equal(desc, rd)


def test_host_memory_kinds_are_told_apart_over_the_whole_range():
    """vsg_host_kind: what decides the route.  hipHostMalloc memory from vsg_host_alloc, registered heap memory, pageable
    memory; a range that starts inside an allocation and ends outside it is pageable (the whole range counts,
    not its first byte)."""
    import ctypes as C
    L = orb.load_library()
    heap = np.zeros(1 << 20, np.uint8)
    assert orb.host_kind(heap) == "pageable"
    orb.pin(heap)
    try:
        assert orb.host_kind(heap) == "registered"
        assert orb.host_kind(heap[4096:8192]) == "registered"
        # one byte beyond the registration: the range is not pinned as a whole
        assert L.vsg_host_kind(C.c_void_p(heap.ctypes.data + 4096), heap.nbytes) == 0
    finally:
        orb.unpin(heap)
    assert orb.host_kind(heap) == "pageable"
    pa = orb.PinnedArray((1 << 20,))
    assert orb.host_kind(pa.a) == "vsg_host_alloc" and orb.host_kind(pa.a[100:5000]) == "vsg_host_alloc"
    assert L.vsg_host_kind(C.c_void_p(pa.ptr + 4096), 1 << 20) == 0  # runs past the allocation's end
    import torch
    t = torch.zeros(1 << 20, dtype=torch.uint8).pin_memory()  # somebody else's pinned allocation
    # torch's caching host allocator hands out hipHostMalloc blocks: exactly kind 2 (exact kinds, so that a
    # misclassification cannot hide), unless torch was told to pin by registration
    expect = 3 if "pinned_use_cuda_host_register:True" in os.environ.get("PYTORCH_CUDA_ALLOC_CONF", "") else 2
    assert L.vsg_host_kind(C.c_void_p(t.data_ptr()), t.numel()) == expect
    # memory somebody else registered (hipHostRegister called behind the library's back, as a caller's own code would):
    # never reported as hipHostMalloc memory
    other = np.zeros(1 << 20, np.uint8)
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))  # the copy the process already runs on
    hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
    assert hip.hipHostRegister(C.c_void_p(other.ctypes.data), other.nbytes, 2 | 1) == 0  # mapped | portable
    try:
        assert L.vsg_host_kind(C.c_void_p(other.ctypes.data), other.nbytes) == 3
    finally:
        hip.hipHostUnregister.argtypes = [C.c_void_p]
        assert hip.hipHostUnregister(C.c_void_p(other.ctypes.data)) == 0


@pytest.mark.parametrize("opt_in", [False, True])
def test_registered_memory_default_route_is_staged_and_the_opt_in_route_is_direct(opt_in):
    """Registered (hipHostRegister) caller memory: by default the device never touches it -- the frames are staged through
    the slot's hipHostMalloc buffers and the records are copied out inside vsg_orb_wait, so rows beyond n[f] ARE untouched
    and the results are the oracle's; with vsg_orb_set_direct_registered(h, 1) the device reads and writes it in place.
    Both routes, throughput (submit / wait) and latency (blocking single frame) forms, against the oracle."""
    B = 3
    ex = orb.ORBextractor(NF, 1.2, 8, 20, 7, max_batch=B)
    ex.set_direct_registered(opt_in)
    cap = ex.capacity(H, W)
    fr = orb.pin(_frames(91, B))
    kps, desc = orb.pin(np.zeros((B, cap), orb.KP_DTYPE)), orb.pin(np.full((B, cap, 32), 0xCD, np.uint8))
    try:
        for _ in range(3):
            n, mono = ex.wait(ex.submit_batch(fr, kps, desc))
            _check(fr, kps, desc, n, mono)
            for f in range(B):
                assert np.all(desc[f, n[f]:] == 0xCD)
        mono, k1, d1 = ex(fr[2])
        rm, rk, rd = ol.OracleExtractor(NF, 1.2, 8, 20, 7)(fr[2])
        assert mono == rm and k1.tobytes() == rk.tobytes() and np.array_equal(d1, rd)
    finally:
        orb.unpin(fr), orb.unpin(kps), orb.unpin(desc)


def test_pinned_array_memory_outlives_its_owner_while_views_exist():
    """PinnedArray has a finalizer, and the allocation lives as long as any numpy view of it."""
    import gc
    pa = orb.PinnedArray((4, 8), np.uint8)
    v = pa.a[1:3]
    v[...] = 7
    pa.free()
    del pa
    gc.collect()
    assert orb.host_kind(v) == "vsg_host_alloc" and int(v.sum()) == 7 * 16  # still mapped, still ours
    base = v.ctypes.data
    del v
    gc.collect()
    import ctypes as C
    assert orb.load_library().vsg_host_kind(C.c_void_p(base), 16) != 1       # released with the last view
    # release(): the explicit form -- raises while a view is still referenced, frees once it is gone
    pb = orb.PinnedArray((64,), np.uint8)
    view = pb.a[8:16]
    assert pb.alive
    with pytest.raises(RuntimeError):
        pb.release()
    assert pb.alive and orb.host_kind(view) == "vsg_host_alloc"
    del view
    pb.release()
    assert not pb.alive


def test_new_image_size_is_refused_while_tickets_are_pending():
    """A different image size rebuilds the handle's buffers and frees the pipeline slots; with un-waited tickets that
    would drop their results silently, so the submit is refused and the pending batch stays intact."""
    B = 2
    ex = orb.ORBextractor(NF, 1.2, 8, 20, 7, max_batch=B)
    cap = ex.capacity(H, W)
    fr = _frames(31, B)
    kps, desc = np.zeros((B, cap), orb.KP_DTYPE), np.zeros((B, cap, 32), np.uint8)
    t = ex.submit_batch(fr, kps, desc)
    other = np.stack([synth.sequence_frame(400, 300, 32, i) for i in range(B)])
    cap2 = ex.capacity(H, W) + 100
    with pytest.raises(orb.VsgError) as e:
        ex.submit_batch(other, np.zeros((B, cap2), orb.KP_DTYPE), np.zeros((B, cap2, 32), np.uint8))
    assert e.value.code == -7 and "waited" in str(e.value)  # VSG_ERR_BUSY, not the capacity code
    n, mono = ex.wait(t)
    _check(fr, kps, desc, n, mono)
    outs = ex.extract_batch(other)  # with nothing pending the new size is taken
    ref = ol.OracleExtractor(NF, 1.2, 8, 20, 7)
    for img, (m, k, d) in zip(other, outs):
        rm, rk, rd = ref(img)
        assert m == rm and k.tobytes() == rk.tobytes() and np.array_equal(d, rd)


@pytest.mark.parametrize("strided", [False, True])
def test_large_pageable_batches_are_staged_by_the_helper_threads(strided):
    """Batches of 16 frames and more from pageable memory take the helper-thread staging path (vsg_orb.hip StagePool);
    `strided` = frames that are views into a larger buffer (row stride != width, frame stride != rows * stride)."""
    B = 24
    ex = orb.ORBextractor(NF, 1.2, 8, 20, 7, max_batch=B)
    cap = ex.capacity(H, W)
    packed = [_frames(300 + i, B) for i in range(4)]
    if strided:
        big = [np.full((B, H + 5, W + 24), 255, np.uint8) for _ in packed]
        for b, p in zip(big, packed):
            b[:, 2:2 + H, 8:8 + W] = p
        batches = [b[:, 2:2 + H, 8:8 + W] for b in big]
    else:
        batches = packed
    outs = [(np.zeros((B, cap), orb.KP_DTYPE), np.zeros((B, cap, 32), np.uint8)) for _ in batches]
    tickets = [ex.submit_batch(batches[i], *outs[i]) for i in range(3)]
    res = [ex.wait(tickets[0])]
    tickets.append(ex.submit_batch(batches[3], *outs[3]))
    res += [ex.wait(t) for t in tickets[1:]]
    want = _oracle(packed[0][:3]) + _oracle(packed[3][-3:])
    got = [(res[0], outs[0], f) for f in range(3)] + [(res[3], outs[3], f) for f in range(B - 3, B)]
    for (rm, rk, rd), ((n, mono), (k, d), f) in zip(want, got):
        assert n[f] == len(rk) and mono[f] == rm
        assert k[f, :n[f]].tobytes() == rk.tobytes() and np.array_equal(d[f, :n[f]], rd)
    # every frame of every batch against the blocking single-frame path of a second handle
    ex1 = orb.ORBextractor(NF, 1.2, 8, 20, 7)
    for i in (1, 2):
        n, mono = res[i]
        for f in range(0, B, 5):
            m1, k1, d1 = ex1(packed[i][f])
            assert n[f] == len(k1) and mono[f] == m1 and outs[i][0][f, :n[f]].tobytes() == k1.tobytes()
            assert np.array_equal(outs[i][1][f, :n[f]], d1)


def test_blocking_calls_ride_the_same_pipeline():
    """vsg_orb_extract / _batch = submit + wait; strided (non-packed) input rows and a sub-view of a larger image."""
    ex = orb.ORBextractor(NF, 1.2, 8, 20, 7, max_batch=2)
    big = np.zeros((2, H + 7, W + 13), np.uint8)
    fr = _frames(31, 2)
    big[:, 3:3 + H, 5:5 + W] = fr
    view = big[:, 3:3 + H, 5:5 + W]  # row stride W + 13, frame stride (H + 7) * (W + 13)
    cap = ex.capacity(H, W)
    kps, desc = np.zeros((2, cap), orb.KP_DTYPE), np.zeros((2, cap, 32), np.uint8)
    t = ex.submit_batch(view, kps, desc)
    n, mono = ex.wait(t)
    _check(fr, kps, desc, n, mono)
    for f, (rm, rk, rd) in zip(fr, _oracle(fr)):
        m, k, d = ex(f)
        assert m == rm and k.tobytes() == rk.tobytes() and np.array_equal(d, rd)


def test_export_writes_only_n_records():
    """D2H is sized by n: rows beyond n[f] of the caller's arrays are left untouched."""
    ex = orb.ORBextractor(NF, 1.2, 8, 20, 7, max_batch=2)
    fr = _frames(41, 2)
    fr[1] //= 8  # low contrast: far fewer keypoints than the capacity
    fr[1] += 100
    cap = ex.capacity(H, W)
    for pinned in (False, True):
        kps = np.zeros((2, cap), orb.KP_DTYPE)
        desc = np.full((2, cap, 32), 0xAB, np.uint8)
        if pinned:
            kps, desc = orb.pin(kps), orb.pin(desc)
        n, mono = ex.wait(ex.submit_batch(fr, kps, desc))
        _check(fr, kps, desc, n, mono)
        assert n[1] < cap - 8
        for f in range(2):
            assert np.all(desc[f, n[f]:] == 0xAB)
        if pinned:
            orb.unpin(kps), orb.unpin(desc)


def test_null_stream_calls_are_ordered():
    """extract_batch_device(stream = NULL) then block_best2_device(stream = NULL) behind a producer on the
    NULL stream -- all three must be ordered without any explicit synchronisation by the caller."""
    import ctypes as C

    import torch
    dev = torch.device("cuda", 0)
    B = 4
    fr = _frames(51, B)
    ex = orb.ORBextractor(NF, 1.2, 8, 20, 7, max_batch=B)
    cap = ex.capacity(H, W)
    L = orb.load_library()
    d_kps = torch.zeros((B, cap, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_counts = torch.zeros((B, 2), dtype=torch.int32, device=dev)
    best, second, arg = (torch.zeros((B - 1, cap), dtype=torch.int32, device=dev) for _ in range(3))
    src = torch.from_numpy(fr).to(dev)
    for rep in range(3):
        junk = torch.randn(4096, 4096, device=dev)
        for _ in range(4):
            junk = junk @ junk  # keeps the NULL stream busy so that the producer below finishes late
        d_gray = (src.to(torch.int16) + int(rep) - int(rep)).to(torch.uint8)  # producer on the NULL stream
        ex.extract_batch_device(d_gray.data_ptr(), B, H * W, H, W, W, d_kps.data_ptr(), d_desc.data_ptr(),
                                d_counts.data_ptr(), cap, (0, 0), None)
        rc = L.vsg_hamming_block_best2_device(0, C.c_void_p(d_desc[1].data_ptr()), C.c_void_p(d_desc[0].data_ptr()),
                                              cap * 32, C.c_void_p(d_counts[1].data_ptr()),
                                              C.c_void_p(d_counts[0].data_ptr()), 2, B - 1, cap,
                                              C.c_void_p(best.data_ptr()), C.c_void_p(second.data_ptr()),
                                              C.c_void_p(arg.data_ptr()), None)
        assert rc == 0
        counts = d_counts.cpu().numpy()  # a NULL-stream copy: ordered behind both
        kps, desc = d_kps.cpu().numpy(), d_desc.cpu().numpy()
        ref = _oracle(fr)
        for f, (rm, rk, rd) in enumerate(ref):
            n = counts[f, 0]
            assert n == len(rk) and kps[f, :n].tobytes() == rk.tobytes() and np.array_equal(desc[f, :n], rd)
        for f in range(B - 1):
            rb, rs, ra = ol.block_best2(ref[f + 1][2], ref[f][2])
            n = len(rb)
            assert np.array_equal(best[f, :n].cpu().numpy(), rb) and np.array_equal(arg[f, :n].cpu().numpy(), ra)
        d_desc.zero_(), d_counts.zero_()


def test_copy_pyramid_is_one_transfer_and_equals_per_level_copies():
    ex = orb.ORBextractor(NF, 1.2, 8, 20, 7, max_batch=2)
    fr = _frames(61, 2)
    ex.extract_batch(fr)
    ref = ol.OracleExtractor(NF, 1.2, 8, 20, 7)
    for f in range(2):
        ref(fr[f])
        levels = ex.copy_pyramid(f)
        for l, lv in enumerate(levels):
            assert np.array_equal(lv, ref.pyramid_level(l, with_border=True))
            assert np.array_equal(lv, ex.image_pyramid(l, frame=f, with_border=True))


def test_two_handles_two_host_threads_large_lds_and_time_stats():
    """Frame.cc:129-132: left / right extractors run concurrently on two host threads.  A level quota above ~1000
    needs more than 64 KB of dynamic LDS in the octree: the raised limit is tracked per device under a mutex
    (vsg_ctx.h lds_limit_ensure), so two handles racing through their first launch both get it."""
    import threading
    img = synth.frame(640, 480, 5)
    want = ol.OracleExtractor(3000, 1.2, 2, 20, 7)(img)
    outs, errs = [None, None], []

    def run(i):
        try:
            ex = orb.ORBextractor(3000, 1.2, 2, 20, 7)  # ~1640 features on level 0
            for _ in range(3):
                outs[i] = ex(img)
            n, mean, sd = ex.time_stats()
            assert n == 3 and mean > 0 and sd >= 0
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for o in outs:
        assert o[0] == want[0] and o[1].tobytes() == want[1].tobytes() and np.array_equal(o[2], want[2])


def test_two_handles_on_two_devices():
    """C5 pinning: one process driving two GPUs.  hipFuncSetAttribute is per device, so the large-LDS launches must
    work on the second device too."""
    if orb.device_count() < 2:
        pytest.skip("needs two HIP devices")
    img = synth.frame(640, 480, 5)
    want = ol.OracleExtractor(3000, 1.2, 2, 20, 7)(img)
    for dev in (0, 1, 0):
        ex = orb.ORBextractor(3000, 1.2, 2, 20, 7, device=dev)
        o = ex(img)
        assert o[0] == want[0] and o[1].tobytes() == want[1].tobytes() and np.array_equal(o[2], want[2])
    s = __import__("scenarios")
    k = s.kf_projection_scenario(1)
    f1 = orb.Frame(len(k["keys"]), device=1).upload(k["keys"], k["desc"], s.BOUNDS)
    o = ol.OracleFrame(k["keys"], k["desc"], s.BOUNDS)
    m0 = np.full(len(k["keys"]), -1, np.int32)
    got = f1.SearchByProjection_Sim3(k["q_desc"], k["u"], k["v"], k["radius"], k["level"], 1.0, m0)
    ref = o.search_by_projection_sim3(k["q_desc"], k["u"], k["v"], k["radius"], k["level"], 1.0, m0)
    assert got[0] == ref[0] and np.array_equal(got[1], ref[1])


def test_c5_four_camera_streams_pinned_to_devices():
    """Config C5 (rs_d435i_rgbd_inertial: 4 concurrent 640x480 streams, nFeatures 1250): one extractor handle + one
    host thread per camera stream, each pinned to the device sharding.stream_to_rank gives it (all four land on the
    devices that exist: with one GPU on device 0, with 4-8 GPUs on distinct ones), frames of every stream in order,
    consecutive frames of a stream matched on the device that holds both."""
    import threading
    from visual_sgraphs_amd import sharding
    ndev = orb.device_count()
    n_streams, n_frames = 4, 3
    seqs = [[synth.sequence_frame(640, 480, 900 + s, t) for t in range(n_frames)] for s in range(n_streams)]
    ref = ol.OracleExtractor(1250, 1.2, 8, 20, 7)
    want = [[ref(f) for f in seq] for seq in seqs]
    got, errs = [[None] * n_frames for _ in range(n_streams)], []

    def camera(s):
        try:
            dev = sharding.stream_to_rank(s, n_streams, min(ndev, n_streams))
            ex = orb.ORBextractor(1250, 1.2, 8, 20, 7, device=dev)
            m = orb.ORBmatcher(0.7, True, device=dev)
            prev = None
            for t, img in enumerate(seqs[s]):
                mono, k, d = ex(img)
                best = m.block_best2(d, prev)[0] if prev is not None else None
                got[s][t] = (mono, k, d, best)
                prev = d
        except Exception as e:  # noqa: BLE001
            errs.append((s, e))
    ts = [threading.Thread(target=camera, args=(s,)) for s in range(n_streams)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for s in range(n_streams):
        for t in range(n_frames):
            mono, k, d, best = got[s][t]
            rm, rk, rd = want[s][t]
            assert mono == rm and k.tobytes() == rk.tobytes() and np.array_equal(d, rd)
            if t > 0:
                assert np.array_equal(best, ol.block_best2(rd, want[s][t - 1][2])[0])
