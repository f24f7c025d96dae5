"""GPU test of the C ABI's RCCL path (vsg_shard_*): communicator of one rank on the one GPU of the test box -- pack
kernel, ncclAllGather, record views -- and a brute-force match that reads the gathered record where it lies.  The
multi-rank protocol itself is covered on CPU (tests/test_sharding.py, gloo, world 2)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
from visual_sgraphs_amd import orb, sharding, synth

pytestmark = pytest.mark.gpu


def _hip():
    """the HIP runtime already in the process (raw device pointers of the gathered records are read back with it)"""
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        try:
            return C.CDLL(name)
        except OSError:
            continue
    pytest.skip("libamdhip64 not found by name")


def test_rccl_all_gather_of_records_world1_and_match_from_gathered_record():
    import torch
    dev = torch.device("cuda", 0)
    B, W, H, NF = 4, 320, 240, 500
    fr = np.stack([synth.sequence_frame(W, H, 71, t) for t in range(B)])
    ex = orb.ORBextractor(NF, 1.2, 8, 20, 7, max_batch=B)
    cap = ex.capacity(H, W)
    d_gray = torch.from_numpy(fr).to(dev)
    d_kps = torch.zeros((B, cap, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_counts = torch.zeros((B, 2), dtype=torch.int32, device=dev)
    st = torch.cuda.Stream(device=dev)
    try:
        comm = sharding.ShardComm(0, 0, 1, cap, B)
    except orb.VsgError as e:
        if e.code == -3:
            pytest.skip("RCCL not loadable on this box")
        raise
    ex.extract_batch_device(d_gray.data_ptr(), B, H * W, H, W, W, d_kps.data_ptr(), d_desc.data_ptr(),
                            d_counts.data_ptr(), cap, (0, 0), st.cuda_stream)
    comm.all_gather(d_counts.data_ptr(), d_kps.data_ptr(), d_desc.data_ptr(), cap, B, st.cuda_stream)
    # match frame 1 (local buffers) against the GATHERED record of frame 0, straight from the receive buffer
    c0, k0, dd0 = comm.record(0, 0)
    best, second, arg = (torch.zeros((1, cap), dtype=torch.int32, device=dev) for _ in range(3))
    L = orb.load_library()
    rc = L.vsg_hamming_block_best2_device(0, C.c_void_p(d_desc[1].data_ptr()), C.c_void_p(dd0), 0,
                                          C.c_void_p(d_counts[1].data_ptr()), C.c_void_p(c0), 2, 1, cap,
                                          C.c_void_p(best.data_ptr()), C.c_void_p(second.data_ptr()),
                                          C.c_void_p(arg.data_ptr()), C.c_void_p(st.cuda_stream))
    assert rc == 0
    st.synchronize()
    ref = ol.OracleExtractor(NF, 1.2, 8, 20, 7)
    want = [ref(f) for f in fr]
    rec = sharding.record_bytes(cap)
    for f in range(B):
        cp, kp, dp = comm.record(0, f)
        raw = np.zeros(rec, np.uint8)
        assert _hip().hipMemcpy(C.c_void_p(raw.ctypes.data), C.c_void_p(cp), C.c_size_t(rec), 2) == 0  # DeviceToHost
        n, mono = raw[:8].view(np.int32)
        rm, rk, rd = want[f]
        assert n == len(rk) and mono == rm
        assert raw[16:16 + n * 28].tobytes() == rk.tobytes()
        od = sharding.desc_offset(cap)
        assert np.array_equal(raw[od:od + n * 32].reshape(n, 32), rd)
        assert kp == cp + 16 and dp == cp + od
    rb, rs, ra = ol.block_best2(want[1][2], want[0][2])
    n1 = len(rb)
    assert np.array_equal(best[0, :n1].cpu().numpy(), rb) and np.array_equal(second[0, :n1].cpu().numpy(), rs)
    assert np.array_equal(arg[0, :n1].cpu().numpy(), ra)
    comm.close()


def _read(ptr, nbytes):
    raw = np.zeros(nbytes, np.uint8)
    assert _hip().hipMemcpy(C.c_void_p(raw.ctypes.data), C.c_void_p(ptr), C.c_size_t(nbytes), 2) == 0
    return raw


def test_partial_batch_and_truncated_records():
    """A partial batch (the last one of a sequence) must not leave the previous batch's records in the slots it does
    not fill: they are sent with n = 0.  A frame with more keypoints than the record capacity is truncated, its
    monoIndex clamped, and flagged."""
    import torch
    dev = torch.device("cuda", 0)
    B, W, H, NF = 4, 320, 240, 500
    fr = np.stack([synth.sequence_frame(W, H, 72, t) for t in range(B)])
    ex = orb.ORBextractor(NF, 1.2, 8, 20, 7, max_batch=B)
    cap = ex.capacity(H, W)
    d_gray = torch.from_numpy(fr).to(dev)
    d_kps = torch.zeros((B, cap, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_counts = torch.zeros((B, 2), dtype=torch.int32, device=dev)
    st = torch.cuda.Stream(device=dev)
    small = 100  # record capacity below the ~500 keypoints per frame
    try:
        comm, comm_small = sharding.ShardComm(0, 0, 1, cap, B), sharding.ShardComm(0, 0, 1, small, B)
    except orb.VsgError as e:
        if e.code == -3:
            pytest.skip("RCCL not loadable on this box")
        raise
    ex.extract_batch_device(d_gray.data_ptr(), B, H * W, H, W, W, d_kps.data_ptr(), d_desc.data_ptr(),
                            d_counts.data_ptr(), cap, (0, 0), st.cuda_stream)
    args = (d_counts.data_ptr(), d_kps.data_ptr(), d_desc.data_ptr(), cap)
    comm.all_gather(*args, B, st.cuda_stream)   # a full batch first ...
    comm.all_gather(*args, 2, st.cuda_stream)   # ... then a partial one: slots 2, 3 must read as empty
    comm_small.all_gather(*args, B, st.cuda_stream)
    st.synchronize()
    counts = d_counts.cpu().numpy()
    for f in range(B):
        hdr = _read(comm.record(0, f)[0], 16).view(np.int32)
        assert hdr.tolist() == ([int(counts[f, 0]), int(counts[f, 1]), 0, 0] if f < 2 else [0, 0, 0, 0])
    desc_h = d_desc.cpu().numpy()
    for f in range(B):
        cp, _, dp = comm_small.record(0, f)
        hdr = _read(cp, 16).view(np.int32)
        assert counts[f, 0] > small and hdr.tolist() == [small, min(int(counts[f, 1]), small), 1, 0]
        assert np.array_equal(_read(dp, small * 32).reshape(small, 32), desc_h[f, :small])
    comm.close(), comm_small.close()


def test_boundary_send_recv_world1():
    """vsg_shard_send_recv_boundary with one rank: the record of the chosen frame goes to rank (r + 1) mod 1 = itself
    through ncclSend / ncclRecv and must arrive bit for bit."""
    import torch
    dev = torch.device("cuda", 0)
    B, W, H, NF = 3, 320, 240, 500
    fr = np.stack([synth.sequence_frame(W, H, 73, t) for t in range(B)])
    ex = orb.ORBextractor(NF, 1.2, 8, 20, 7, max_batch=B)
    cap = ex.capacity(H, W)
    d_gray = torch.from_numpy(fr).to(dev)
    d_kps = torch.zeros((B, cap, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_counts = torch.zeros((B, 2), dtype=torch.int32, device=dev)
    st = torch.cuda.Stream(device=dev)
    try:
        comm = sharding.ShardComm(0, 0, 1, cap, B)
    except orb.VsgError as e:
        if e.code == -3:
            pytest.skip("RCCL not loadable on this box")
        raise
    assert comm.world_seen() == 1 and comm.rank_seen() == 0  # ncclCommCount / ncclCommUserRank on the live communicator
    ex.extract_batch_device(d_gray.data_ptr(), B, H * W, H, W, W, d_kps.data_ptr(), d_desc.data_ptr(),
                            d_counts.data_ptr(), cap, (0, 0), st.cuda_stream)
    comm.send_recv_boundary(d_counts.data_ptr(), d_kps.data_ptr(), d_desc.data_ptr(), cap, B - 1, st.cuda_stream)
    st.synchronize()
    cp, kp, dp = comm.boundary_record()
    rm, rk, rd = ol.OracleExtractor(NF, 1.2, 8, 20, 7)(fr[B - 1])
    hdr = _read(cp, 16).view(np.int32)
    assert hdr.tolist() == [len(rk), rm, 0, 0]
    assert _read(kp, len(rk) * 28).tobytes() == rk.tobytes()
    assert np.array_equal(_read(dp, len(rk) * 32).reshape(-1, 32), rd)
    # d2d copy through the C ABI (what bench.py uses for its boundary state)
    dst = torch.zeros(len(rk) * 32, dtype=torch.uint8, device=dev)
    orb.copy_d2d_async(dst.data_ptr(), dp, len(rk) * 32, st.cuda_stream)
    st.synchronize()
    assert np.array_equal(dst.cpu().numpy().reshape(-1, 32), rd)
    comm.close()
