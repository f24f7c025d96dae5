"""The batches that are TIMED, compared frame by frame: every frame and every match row of a 512-frame
C2 batch and a 128-frame C4 batch -- the shapes bench.py measures -- through vsg_orb_extract_batch_device +
vsg_hamming_block_best2_device against the CPU oracle run on all host cores (or_extract_batch_mt /
or_block_best2_batch_mt), as one batch and cut into sub-batches, on the default content and on value noise (every cell
empty at iniThFAST).  The reference's contract is per frame (Frame.cc:555-563 -> ORBextractor.cc:1083-1169); what only
exists at this scale -- the XCD block remap, several cells per workgroup with a tile in flight, the octree's launch
carrying the blur, sub-batch streams -- is certified here by all of the frames it produces, not by two of them.

Plus a race hunter: the same batch many times over with the matcher on a second stream and a second handle busy on
another host thread; every repetition's outputs are compared ON THE DEVICE with the first (no read-back in the loop)."""
import ctypes as C
import threading

import numpy as np
import pytest

import oracle_lib as ol
from visual_sgraphs_amd import orb, synth

pytestmark = pytest.mark.gpu

GEOM = {"C2": (640, 480, 1000), "C4": (1280, 720, 2000)}


def _frames(kind, W, H, B, nuniq):
    if kind == "rectangles":
        uniq = [synth.sequence_frame(W, H, 1000, t) for t in range(nuniq)]
    else:
        uniq = [synth.content_frame(kind, W, H, 5000, t) for t in range(nuniq)]
    uniq = np.stack(uniq)
    return uniq, np.concatenate([uniq] * ((B + nuniq - 1) // nuniq))[:B]


def _run_device(ex, frames, cap, device=0):
    """extract + match of every frame against its predecessor (frame 0 against the batch's last frame of a previous,
    identical step) with everything resident; returns host copies."""
    import torch
    B, H, W = frames.shape
    dev = torch.device("cuda", device)
    d_gray = torch.from_numpy(frames).to(dev)
    d_kps = torch.zeros((B + 1, cap, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B + 1, cap, 32), dtype=torch.uint8, device=dev)
    d_counts = torch.zeros((B + 1, 2), dtype=torch.int32, device=dev)
    d_best, d_second, d_arg = (torch.zeros((B, cap), dtype=torch.int32, device=dev) for _ in range(3))
    st = torch.cuda.Stream(device=dev)
    L = orb.load_library()
    vp = C.c_void_p
    with torch.cuda.stream(st):
        for _ in range(2):  # the second step matches frame 0 against the first step's last frame
            d_desc[0].copy_(d_desc[B])
            d_counts[0].copy_(d_counts[B])
            ex.extract_batch_device(d_gray.data_ptr(), B, H * W, H, W, W, d_kps[1].data_ptr(), d_desc[1].data_ptr(),
                                    d_counts[1].data_ptr(), cap, (0, 0), st.cuda_stream)
            rc = L.vsg_hamming_block_best2_device(device, vp(d_desc[1].data_ptr()), vp(d_desc[0].data_ptr()), cap * 32,
                                                  vp(d_counts[1].data_ptr()), vp(d_counts[0].data_ptr()), 2, B, cap,
                                                  vp(d_best.data_ptr()), vp(d_second.data_ptr()), vp(d_arg.data_ptr()),
                                                  vp(st.cuda_stream))
            assert rc == 0, rc
    torch.cuda.synchronize()
    return (d_counts.cpu().numpy(), d_kps.cpu().numpy(), d_desc.cpu().numpy(), d_best.cpu().numpy(),
            d_second.cpu().numpy(), d_arg.cpu().numpy())


def check_every_frame(counts, kps, desc, best, second, arg, uniq, nfeat, cap, what):
    """counts / kps / desc: [B + 1, ...] with slot 0 = the predecessor of frame 0 (= the last frame); best / second /
    arg: [B, cap].  The oracle extracts the distinct frames once; every device frame is compared with its own."""
    B, nuniq = len(counts) - 1, len(uniq)
    rc, rk, rd = ol.extract_batch(uniq, nfeat, cap)
    idx = np.arange(B) % nuniq
    bad = ol.compare_batch(counts[1:], kps[1:], desc[1:], rc[idx], rk[idx], rd[idx])
    assert bad == [], f"{what}: {len(bad)} of {B} frames differ from the oracle, first {bad[:8]}"
    assert counts[0].tolist() == counts[B].tolist() and np.array_equal(desc[0], desc[B])
    # every match row: frame f against f - 1 (frame 0 against the last one), on the oracle's own descriptors
    pidx = np.concatenate([[(B - 1) % nuniq], idx[:-1]])
    pairs = sorted(set(zip(idx.tolist(), pidx.tolist())))
    a = np.stack([rd[i] for i, _ in pairs])
    b = np.stack([rd[j] for _, j in pairs])
    rb, rs, ra = ol.block_best2_batch(a, [rc[i, 0] for i, _ in pairs], b, [rc[j, 0] for _, j in pairs])
    row = {p: k for k, p in enumerate(pairs)}
    badm = []
    for f in range(B):
        k, n = row[(int(idx[f]), int(pidx[f]))], int(rc[idx[f], 0])
        if not (np.array_equal(best[f, :n], rb[k, :n]) and np.array_equal(second[f, :n], rs[k, :n])
                and np.array_equal(arg[f, :n], ra[k, :n])):
            badm.append(f)
    assert badm == [], f"{what}: {len(badm)} of {B} match rows differ from the oracle, first {badm[:8]}"
    return int(rc[:, 0].min()), int(rc[:, 0].max())


@pytest.mark.parametrize("nsub", [None, "2"])
@pytest.mark.parametrize("kind", ["rectangles", "value_noise"])
def test_c2_batch_512_every_frame(kind, nsub, monkeypatch):
    if nsub is None:
        monkeypatch.delenv("VSG_SUBBATCH", raising=False)
    else:
        monkeypatch.setenv("VSG_SUBBATCH", nsub)
    W, H, nfeat = GEOM["C2"]
    B = 512
    uniq, frames = _frames(kind, W, H, B, 64)
    ex = orb.ORBextractor(nfeat, 1.2, 8, 20, 7, max_batch=B)
    cap = ex.capacity(H, W)
    lo, hi = check_every_frame(*_run_device(ex, frames, cap), uniq, nfeat, cap, f"C2/512 {kind} nsub={nsub}")
    assert lo > 900


@pytest.mark.parametrize("nsub", [None, "2"])
@pytest.mark.parametrize("kind", ["rectangles", "value_noise"])
def test_c4_batch_128_every_frame(kind, nsub, monkeypatch):
    if nsub is None:
        monkeypatch.delenv("VSG_SUBBATCH", raising=False)
    else:
        monkeypatch.setenv("VSG_SUBBATCH", nsub)
    W, H, nfeat = GEOM["C4"]
    B = 128
    uniq, frames = _frames(kind, W, H, B, 32)
    ex = orb.ORBextractor(nfeat, 1.2, 8, 20, 7, max_batch=B)
    cap = ex.capacity(H, W)
    lo, hi = check_every_frame(*_run_device(ex, frames, cap), uniq, nfeat, cap, f"C4/128 {kind} nsub={nsub}")
    assert lo > 1800


@pytest.mark.parametrize("cfg,B,nuniq,floor", [("C2", 1024, 64, 900), ("C4", 256, 32, 1800)])
@pytest.mark.parametrize("kind", ["rectangles", "value_noise", "photo_china"])
def test_bench_default_batches_every_frame(cfg, B, nuniq, floor, kind, monkeypatch):
    """The batch sizes bench.py runs by default since the end of round 5 (C2 / 1024 for `value`, C4 / 256 in other_configs):
    every frame and every match row of one such batch against the oracle.  `photo_china` (round 6): a real photograph --
    5.7 k (C2) / 19.6 k (C4) level-0 candidates per frame, 3.5 - 12 x the rectangles' -- so the octree's memory-resident
    form (more than 2048 candidates of a level) and the long candidate segments run at the timed batch sizes."""
    monkeypatch.delenv("VSG_SUBBATCH", raising=False)
    W, H, nfeat = GEOM[cfg]
    uniq, frames = _frames(kind, W, H, B, nuniq)
    ex = orb.ORBextractor(nfeat, 1.2, 8, 20, 7, max_batch=B)
    cap = ex.capacity(H, W)
    lo, hi = check_every_frame(*_run_device(ex, frames, cap), uniq, nfeat, cap, f"{cfg}/{B} {kind}")
    assert lo > floor


@pytest.mark.parametrize("cfg,B,nuniq", [("C2", 510, 48), ("C4", 96, 24)])
def test_every_frame_of_a_batch_of_photographs(cfg, B, nuniq):
    """All three photographs interleaved frame by frame (a building: dense corners everywhere but the sky; a portrait:
    a quarter of the keypoints from the minThFAST retry; a flower: large defocused areas), so neighbouring workgroups of the
    batch-wide launches see 10 x different candidate counts."""
    W, H, nfeat = GEOM[cfg]
    kinds = list(synth.PHOTO_CLASSES)
    uniq = np.stack([synth.content_frame(kinds[t % 3], W, H, 7000 + t // 3, t) for t in range(nuniq)])
    frames = np.concatenate([uniq] * ((B + nuniq - 1) // nuniq))[:B]
    ex = orb.ORBextractor(nfeat, 1.2, 8, 20, 7, max_batch=B)
    cap = ex.capacity(H, W)
    check_every_frame(*_run_device(ex, frames, cap), uniq, nfeat, cap, f"{cfg}/{B} photographs")


def test_every_frame_of_a_ragged_tail_batch():
    """509 frames (not a multiple of the 8 XCDs, of 32, or of the cells-per-workgroup factor) of mixed content classes,
    one class per frame."""
    W, H, nfeat = GEOM["C2"]
    B = 509
    kinds = list(synth.CONTENT_CLASSES)
    uniq = np.stack([synth.content_frame(kinds[t % len(kinds)], W, H, 6000, t) for t in range(50)])
    frames = np.concatenate([uniq] * 11)[:B]
    ex = orb.ORBextractor(nfeat, 1.2, 8, 20, 7, max_batch=512)
    cap = ex.capacity(H, W)
    check_every_frame(*_run_device(ex, frames, cap), uniq, nfeat, cap, "C2/509 mixed classes")


def test_race_hunter_repeat_and_compare_on_device():
    """The same 512-frame batch 200 times with the match of step k on a second stream under the extraction of step k + 1
    (bench.py --match-stream 1: two alternating output sets) while a second handle extracts other frames from another
    host thread on the same device.  After every step the outputs are compared with the first step's ON THE DEVICE
    (torch.equal per array, accumulated into one flag tensor: no read-back inside the loop); the first step itself is
    checked against the oracle frame by frame."""
    import torch
    W, H, nfeat = GEOM["C2"]
    B, REPS = 512, 200
    uniq, frames = _frames("rectangles", W, H, B, 64)
    dev = torch.device("cuda", 0)
    ex = orb.ORBextractor(nfeat, 1.2, 8, 20, 7, max_batch=B)
    cap = ex.capacity(H, W)
    L = orb.load_library()
    vp = C.c_void_p
    d_gray = torch.from_numpy(frames).to(dev)
    sets = [tuple(torch.zeros(s, dtype=t, device=dev) for s, t in (((B + 1, cap, 28), torch.uint8),
                                                                    ((B + 1, cap, 32), torch.uint8),
                                                                    ((B + 1, 2), torch.int32))) for _ in range(2)]
    d_match = [tuple(torch.zeros((B, cap), dtype=torch.int32, device=dev) for _ in range(3)) for _ in range(2)]
    gold = None
    n_diff = torch.zeros((), dtype=torch.int32, device=dev)
    tstream, mstream = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    ev_ext = torch.cuda.Event()
    ev_matched = [None, None]

    # the second handle: another thread, another stream set, other frames, until told to stop
    stop = threading.Event()
    other_err = []

    def other():
        try:
            ex2 = orb.ORBextractor(500, 1.2, 8, 20, 7, max_batch=8)
            imgs = np.stack([synth.sequence_frame(320, 240, 31, t) for t in range(8)])
            want = None
            while not stop.is_set():
                outs = ex2.extract_batch(imgs)
                sig = [(m, k.tobytes(), d.tobytes()) for m, k, d in outs]
                if want is None:
                    want = sig
                elif sig != want:
                    other_err.append("second handle: outputs changed between repetitions")
                    return
        except Exception as e:  # noqa: BLE001
            other_err.append(repr(e))

    th = threading.Thread(target=other)
    th.start()
    try:
        for k in range(REPS):
            i = k & 1
            o_kps, o_desc, o_counts = sets[i]
            p_desc, p_counts = sets[i ^ 1][1], sets[i ^ 1][2]
            best, second, arg = d_match[i]
            with torch.cuda.stream(tstream):
                if ev_matched[i] is not None:
                    tstream.wait_event(ev_matched[i])
                ex.extract_batch_device(d_gray.data_ptr(), B, H * W, H, W, W, o_kps[1].data_ptr(), o_desc[1].data_ptr(),
                                        o_counts[1].data_ptr(), cap, (0, 0), tstream.cuda_stream)
                o_desc[0].copy_(p_desc[B])
                o_counts[0].copy_(p_counts[B])
                ev_ext.record(tstream)
            mstream.wait_event(ev_ext)
            with torch.cuda.stream(mstream):
                rc = L.vsg_hamming_block_best2_device(0, vp(o_desc[1].data_ptr()), vp(o_desc[0].data_ptr()), cap * 32,
                                                      vp(o_counts[1].data_ptr()), vp(o_counts[0].data_ptr()), 2, B, cap,
                                                      vp(best.data_ptr()), vp(second.data_ptr()), vp(arg.data_ptr()),
                                                      vp(mstream.cuda_stream))
                assert rc == 0, rc
                if k == 1:
                    # step 1 is the first whose frame 0 has a real predecessor: it becomes the comparison copy
                    gold = [t.clone() for t in (o_kps[1:], o_desc[1:], o_counts[1:], best, second, arg)]
                elif k > 1:
                    for g, t in zip(gold, (o_kps[1:], o_desc[1:], o_counts[1:], best, second, arg)):
                        n_diff += (g != t).any().to(torch.int32)
                ev = torch.cuda.Event()
                ev.record(mstream)
                ev_matched[i] = ev
        torch.cuda.synchronize()
    finally:
        stop.set()
        th.join()
    assert other_err == [], other_err
    assert int(n_diff.item()) == 0, f"{int(n_diff.item())} output arrays differed from the first repetition"
    counts = torch.cat([gold[2][-1:], gold[2]]).cpu().numpy()
    kps = torch.cat([gold[0][-1:], gold[0]]).cpu().numpy()
    desc = torch.cat([gold[1][-1:], gold[1]]).cpu().numpy()
    check_every_frame(counts, kps, desc, gold[3].cpu().numpy(), gold[4].cpu().numpy(), gold[5].cpu().numpy(), uniq, nfeat,
                      cap, "race hunter, repetition 1")
