#!/usr/bin/env python3
"""bench.py -- frames/s of ORB extract+match on MI355X (BASELINE.json metric, config C2).

One STEP = one pass of the hot path over one batch of `--batch` (default 1024) synthetic 640x480 gray frames that are
already resident in HBM: ORBextractor::operator() for every frame (pyramid, per-cell FAST, octree,
orientation, 7x7 blur, rBRIEF-256; nFeatures=1000, 8 levels) + the brute-force Hamming best/second-best
match of every frame against its predecessor (the inner search of ORBmatcher::SearchByBoW with one node).
With N > 1 ranks (torchrun) every rank processes its own `--batch` frames per step (frame sharding, weak
scaling) and the keypoint/descriptor records are exchanged with one RCCL all-gather per step.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel by ALGORITHMIC bytes over
its HIP-event duration; `cpu_baseline` is the CPU oracle (a port of the reference algorithm, not the
reference binary) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import numpy as np  # noqa: E402

sys.path.insert(0, str(ROOT / "tools"))
import bench_extras  # noqa: E402  (the legs outside the timed region)
from bench_common import (HBM_PEAK_GBS, ISA_MIX_FILE, TRAFFIC_FILE, WORKLOADS, BatchOracle, algorithmic_bytes,  # noqa: E402,F401
                          cpu_baseline, effective_cores, gate_report, pmc_traffic)
from bench_extras import config_chain_leg, content_sweep_leg, device_rate, host_api_leg, matcher_latency_leg  # noqa: E402,F401
from bench_launch import (check_rccl_world, launch_ranks, preflight, refuse_without_library_exchange,  # noqa: E402,F401
                          run_teeing_stderr, visible_gpu_count, wants_library_exchange)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024,
                    help="frames per step per GPU (1024: 383-385 k frames/s where 512 gives 374-375 k in the same run -- six "
                         "launch tails per step, whatever its size; profiles/r05_v_*)")
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--content", default="rectangles",
                    help="content class of the timed frames (visual_sgraphs_amd.synth.CONTENT_CLASSES; the default is what "
                         "`value` is quoted on; photo_china / photo_hopper / photo_flower are real photographs)")
    ap.add_argument("--no-match", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--ramp-steps", type=int, default=150,
                    help="untimed steps BEFORE the W warm-up steps: the GPU needs ~0.1-0.2 s of load to reach its steady "
                         "clocks (K = 20 after W = 5: 317.7 k frames/s, after 100+ steps of load: 325.1 k; DESIGN.md section 5)")
    ap.add_argument("--no-rotate-inputs", action="store_true",
                    help="N = 1: extract the same resident batch every step instead of alternating between two")
    ap.add_argument("--match-stream", type=int, default=0, choices=[0, 1],
                    help="N = 1: 1 = the match of step k runs on a second stream under the extraction of step k + 1 "
                         "(two output sets, alternating); 0 = everything on one stream")
    ap.add_argument("--sync-gather", action="store_true",
                    help="make every step wait for its own all-gather (default: the exchange of batch k overlaps the "
                         "kernels of batch k+1, two record buffers in flight)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="budget per CPU baseline leg (0 = skip)")
    ap.add_argument("--no-stage-timing", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the legs outside the timed region (host API, matcher latency, C4 rate)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) for real runs; gloo for dry runs")
    ap.add_argument("--torch-gather", action="store_true",
                    help="exchange the records through torch.distributed instead of the library's own RCCL communicator")
    ap.add_argument("--exchange", default="allgather", choices=["allgather", "boundary"],
                    help="N > 1: allgather = every rank's whole batch of records into every GPU (what north_star words); "
                         "boundary = only what the chunk partition needs, one ncclSend/ncclRecv of the last frame's record "
                         "to the successor rank")
    ap.add_argument("--preflight", action="store_true",
                    help="check what an N-GPU run depends on (devices, one HIP runtime for torch and libvsg_orb.so, RCCL's "
                         "exports, memory for the exchange buffers) in a child process, print the findings and exit; "
                         "non-zero exit with the reason when a check fails")
    ap.add_argument("--no-preflight", action="store_true", help="--gpus N > 1: start the ranks without the preflight checks")
    ap.add_argument("--one-device", action="store_true",
                    help="dry run of the multi-rank path on a single GPU: every rank uses device 0 (needs gloo)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.preflight:
        sys.exit(preflight(args.gpus, args.batch, args.workload, one_device=args.one_device))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, sys.argv[1:]))  # before torch / HIP are touched in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:  # a launcher / flag mismatch is an error, never a line with the wrong n_gpus
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with --nproc-per-node {args.gpus} "
                         "(or plain `python bench.py --gpus N`, which starts the ranks itself)")

    import torch
    import torch.distributed as dist
    from visual_sgraphs_amd import orb, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the ORB front-end has no CPU fallback)")
    if args.one_device:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants device {local_rank} but only {torch.cuda.device_count()} visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)

    W, H, nfeat = WORKLOADS[args.workload]
    B, K, Wu = args.batch, args.steps, args.warmup
    ex = orb.ORBextractor(nfeat, 1.2, 8, 20, 7, device=local_rank, max_batch=B)
    cap = ex.capacity(H, W)

    # synthetic frames: ONE global sequence per step, dealt to the ranks in contiguous chunks (sharding.chunk_frames):
    # rank r owns frames [r * B, (r + 1) * B); consecutive frames are translated copies (matchable), also across the
    # chunk boundary -- the predecessor of a rank's first frame is the LAST frame of rank r - 1, which only the record
    # exchange can deliver (for rank 0: the last rank's last frame of the previous step)
    t_first = rank * B
    def frame_of(t):  # frame t of the global sequence (content_frame("rectangles", ...) is sequence_frame)
        return synth.content_frame(args.content, W, H, 1000, t)
    frames = np.stack([frame_of(t_first + t) for t in range(B)])
    d_gray = torch.from_numpy(frames).to(dev)
    # N = 1: a second resident batch (the same frames in reverse order: other addresses, other neighbours) alternates
    # with the first from step to step, so no step re-reads what the previous one left in the caches (the 157 MB of a
    # 512-frame batch are below the 256 MB of Infinity Cache)
    rotate = not distributed and not args.no_rotate_inputs
    d_gray_in = [d_gray] + ([torch.flip(d_gray, dims=[0]).contiguous()] if rotate else [])
    in_state = {"k": 0, "last": 0}

    def next_input():
        i = in_state["k"] % len(d_gray_in)
        in_state["k"] += 1
        in_state["last"] = i
        return d_gray_in[i].data_ptr()
    # output records: slot 0 keeps the previous batch's last frame so every frame has a predecessor to match (N = 1)
    d_kps = torch.zeros((B + 1, cap, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B + 1, cap, 32), dtype=torch.uint8, device=dev)
    d_counts = torch.zeros((B + 1, 2), dtype=torch.int32, device=dev)
    d_best = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    d_second = torch.zeros_like(d_best)
    d_arg = torch.zeros_like(d_best)
    from visual_sgraphs_amd import sharding
    L = orb.load_library()
    import ctypes as C
    # All kernels of a step are ordered on ONE explicit (non-default) HIP stream: torch copies, the extractor's stage
    # chain, the match kernels.  The record exchange runs on a second stream, under the NEXT step's kernels.
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0
    exchange = distributed and not args.no_gather
    comm, comm_kind = None, None
    if exchange:
        cstream = torch.cuda.Stream(device=dev)
        rec_bytes = sharding.record_bytes(cap)
        od = sharding.desc_offset(cap)
        if wants_library_exchange(args.dist_backend, args.one_device, args.torch_gather):
            why = ""
            try:  # the library's own RCCL communicator (C ABI: vsg_shard_*)
                comm = sharding.ShardComm(local_rank, rank, world, cap, B)
                comm_kind = "C ABI vsg_shard_* (ncclAllGather)"
            except Exception as e:  # noqa: BLE001  (the message carries vsg_shard_last_error())
                why = str(e)
            # every rank learns whether ALL ranks have their communicator; a multi-GPU line is this library's exchange or it
            # is not printed at all (the run used to go on over torch.distributed with only `comm_kind` saying so)
            ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if comm is not None:
                    comm.close()
                dist.destroy_process_group()
                refuse_without_library_exchange(rank, world, why)
            check_rccl_world(comm.world_seen(), world, rank)
        if comm is None:
            comm_kind = f"torch.distributed {args.dist_backend} " + ("all_gather_into_tensor" if args.exchange == "allgather"
                                                                     else "batch_isend_irecv")
            send = torch.zeros((B, rec_bytes), dtype=torch.uint8, device=dev)
            recv = torch.zeros((world * B, rec_bytes), dtype=torch.uint8, device=dev)
            brecv = torch.zeros((1, rec_bytes), dtype=torch.uint8, device=dev)
        elif args.exchange == "boundary":
            comm_kind = "C ABI vsg_shard_send_recv_boundary (ncclSend + ncclRecv)"
        # boundary state: this rank's first frame of the previous step, and (rank 0) the last rank's last frame
        d_first_desc = torch.zeros((cap, 32), dtype=torch.uint8, device=dev)
        d_first_cnt = torch.zeros((2,), dtype=torch.int32, device=dev)
        d_tail_desc = torch.zeros((cap, 32), dtype=torch.uint8, device=dev)
        d_tail_cnt = torch.zeros((2,), dtype=torch.int32, device=dev)
        d_bbest, d_bsecond, d_barg = (torch.zeros((1, cap), dtype=torch.int32, device=dev) for _ in range(3))
        ev_extracted, ev_consumed, ev_gathered = (torch.cuda.Event() for _ in range(3))
        state = {"steps": 0}
        pred_rank = rank - 1 if rank > 0 else world - 1

    match_events = []
    # N = 1: the matcher of step k on its own stream, under the pyramid of step k + 1 (the MFMA matcher and the
    # LDS / barrier-bound pyramid leave each other room).  Step k + 1 then writes a SECOND set of output arrays while
    # the matcher still reads the first; a set is written again only after the match that read it (event).
    overlap_match = {"on": bool(args.match_stream) and not exchange and not args.no_match}
    out_sets = [(d_kps, d_desc, d_counts)]
    if overlap_match["on"]:
        out_sets.append((torch.zeros_like(d_kps), torch.zeros_like(d_desc), torch.zeros_like(d_counts)))
        mstream = torch.cuda.Stream(device=dev)
        ev_ext = torch.cuda.Event()
        ev_matched = [None, None]
    cur = {"k": 0, "last": 0}

    def best2(a_desc, b_desc, a_cnt, b_cnt, nblocks, stride_bytes, best, second, arg, on_stream=None):
        rc = L.vsg_hamming_block_best2_device(local_rank, C.c_void_p(a_desc), C.c_void_p(b_desc), stride_bytes,
                                              C.c_void_p(a_cnt), C.c_void_p(b_cnt), 2, nblocks, cap,
                                              C.c_void_p(best), C.c_void_p(second), C.c_void_p(arg),
                                              C.c_void_p(stream if on_stream is None else on_stream))
        assert rc == 0, rc

    def gathered_record(r, f):
        """(counts, desc) device pointers of frame f of rank r in the last completed exchange (boundary mode: the one
        record that exchange delivers, the predecessor rank's last frame)"""
        if args.exchange == "boundary":
            assert r == pred_rank and f == B - 1
            if comm is not None:
                c, _, d = comm.boundary_record()
                return c, d
            return brecv[0].data_ptr(), brecv[0].data_ptr() + od
        if comm is not None:
            c, _, d = comm.record(r, f)
            return c, d
        row = recv[r * B + f]
        return row.data_ptr(), row.data_ptr() + od

    def timed_match(fn, strm, time_match):
        """run fn() on `strm`, bracketed by HIP events when the serialized pass wants the match kernel's duration"""
        if not time_match:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(strm)
        fn()
        e1.record(strm)
        match_events.append((e0, e1))

    def step(time_match=False):
        """ONE step = one pass of the hot path over one resident batch: extract B frames, match each against its predecessor.
        N = 1 (what `value` is quoted on) is the first half; frame-sharded ranks take step_sharded."""
        if exchange:
            return step_sharded(time_match)
        if len(out_sets) == 2:
            # --match-stream 1: two output sets, extract into one while the previous step's match reads the other
            i = cur["k"] & 1
            cur["k"] += 1
            cur["last"] = i
            o_kps, o_desc, o_counts = out_sets[i]
            p_desc, p_counts = out_sets[i ^ 1][1], out_sets[i ^ 1][2]
            side = overlap_match["on"] and not time_match
            if ev_matched[i] is not None:
                tstream.wait_event(ev_matched[i])  # the match that read this set two steps ago
                ev_matched[i] = None
            ex.extract_batch_device(next_input(), B, H * W, H, W, W, o_kps[1].data_ptr(), o_desc[1].data_ptr(),
                                    o_counts[1].data_ptr(), cap, (0, 0), stream)
            ms = mstream if side else tstream
            if side:
                # the last frame of the previous batch into slot 0 ON THE EXTRACTION STREAM: the next step extracts into
                # the set these two copies read, and it only waits for the match recorded two steps ago -- on the match
                # stream they could still be reading while that extraction writes
                o_desc[0].copy_(p_desc[B])
                o_counts[0].copy_(p_counts[B])
                ev_ext.record(tstream)
                mstream.wait_event(ev_ext)
            with torch.cuda.stream(ms):
                def run():
                    if not side:
                        # the last frame of the previous batch into slot 0 (its extraction is earlier work of this stream)
                        o_desc[0].copy_(p_desc[B])
                        o_counts[0].copy_(p_counts[B])
                    best2(o_desc[1].data_ptr(), o_desc[0].data_ptr(), o_counts[1].data_ptr(), o_counts[0].data_ptr(), B,
                          cap * 32, d_best.data_ptr(), d_second.data_ptr(), d_arg.data_ptr(), on_stream=ms.cuda_stream)
                timed_match(run, ms, time_match)
                if side:
                    ev = torch.cuda.Event()
                    ev.record(ms)
                    ev_matched[i] = ev
            return
        # the default: everything on one stream.  Carry the last frame of the previous batch into slot 0, extract the batch
        # (pyramid, FAST, octree + blur, orientation + rBRIEF: one stage chain per call), match every frame against its
        # predecessor (MFMA brute-force best / second best).
        d_desc[0].copy_(d_desc[B])
        d_counts[0].copy_(d_counts[B])
        ex.extract_batch_device(next_input(), B, H * W, H, W, W, d_kps[1].data_ptr(), d_desc[1].data_ptr(),
                                d_counts[1].data_ptr(), cap, (0, 0), stream)
        if not args.no_match:
            timed_match(lambda: best2(d_desc[1].data_ptr(), d_desc[0].data_ptr(), d_counts[1].data_ptr(), d_counts[0].data_ptr(),
                                      B, cap * 32, d_best.data_ptr(), d_second.data_ptr(), d_arg.data_ptr()), tstream, time_match)

    def step_sharded(time_match=False):
        """N > 1: this rank's chunk of the sequence; the record exchange of step k runs on a second stream under the kernels of
        step k + 1, and the chunk's first frame is matched against the predecessor rank's last frame one step later."""
        ex.extract_batch_device(next_input(), B, H * W, H, W, W, d_kps[1].data_ptr(), d_desc[1].data_ptr(),
                                d_counts[1].data_ptr(), cap, (0, 0), stream)
        ev_extracted.record(tstream)
        if not args.no_match:
            def run():
                # frames 1 .. B-1 against their local predecessors
                if B > 1:
                    best2(d_desc[2].data_ptr(), d_desc[1].data_ptr(), d_counts[2].data_ptr(), d_counts[1].data_ptr(),
                          B - 1, cap * 32, d_best[1].data_ptr(), d_second[1].data_ptr(), d_arg[1].data_ptr())
                # the previous step's first frame against its predecessor on the neighbour rank, from the records
                # that step's all-gather delivered (the exchange of step k runs under the kernels of step k + 1)
                if state["steps"] > 0:
                    tstream.wait_event(ev_gathered)
                    if rank > 0:
                        pc, pd = gathered_record(pred_rank, B - 1)
                    else:
                        pc, pd = d_tail_cnt.data_ptr(), d_tail_desc.data_ptr()
                    best2(d_first_desc.data_ptr(), pd, d_first_cnt.data_ptr(), pc, 1, 0, d_bbest.data_ptr(),
                          d_bsecond.data_ptr(), d_barg.data_ptr())
                    if rank == 0:  # keep the last rank's last frame for the NEXT boundary match
                        tc, td = gathered_record(world - 1, B - 1)
                        L_memcpy(d_tail_cnt.data_ptr(), tc, 8)
                        L_memcpy(d_tail_desc.data_ptr(), td, cap * 32)
                d_first_desc.copy_(d_desc[1])
                d_first_cnt.copy_(d_counts[1])
                ev_consumed.record(tstream)
            timed_match(run, tstream, time_match)
        else:
            ev_consumed.record(tstream)
        with torch.cuda.stream(cstream):
            cstream.wait_event(ev_extracted)  # this step's records exist
            cstream.wait_event(ev_consumed)   # the previous exchange's records have been read
            if comm is not None and args.exchange == "boundary":
                comm.send_recv_boundary(d_counts[1].data_ptr(), d_kps[1].data_ptr(), d_desc[1].data_ptr(), cap, B - 1,
                                        cstream.cuda_stream)
            elif comm is not None:
                comm.all_gather(d_counts[1].data_ptr(), d_kps[1].data_ptr(), d_desc[1].data_ptr(), cap, B,
                                cstream.cuda_stream)
            elif args.exchange == "boundary":
                sharding.pack_records(send[:1], d_counts[B:B + 1], d_kps[B:B + 1], d_desc[B:B + 1])
                sharding.send_recv_boundary(brecv, send[:1], (rank + 1) % world, pred_rank)
            else:
                sharding.pack_records(send, d_counts[1:], d_kps[1:], d_desc[1:])
                w = sharding.all_gather_records(recv, send, async_op=args.dist_backend == "nccl")
                if w is not None:
                    w.wait()  # stream-level: cstream waits for the collective, the host does not block
            ev_gathered.record(cstream)
        state["steps"] += 1
        if args.sync_gather:
            tstream.wait_event(ev_gathered)

    def L_memcpy(dst, src, nbytes):
        """device-to-device copy of raw pointers on the compute stream (C ABI: the library's own HIP runtime)"""
        orb.copy_d2d_async(dst, src, nbytes, stream, local_rank)

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.ramp_steps, 0)):  # clock ramp: the same count on every rank (steps may hold collectives)
        step()
    for _ in range(Wu):
        step()
    barrier()
    if not args.no_stage_timing:
        # events around the dominant kernel's launch only (on the stream it is launched on): the chain keeps the shape
        # of an untimed call -- with events around every stage the blur leaves the octree's launch and the step is
        # a few % slower (the per-stage figures come from the serialized pass below)
        ex.enable_timing(2)
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    stage_ms_timed = ex.timing_ms() if not args.no_stage_timing else {}
    # The parity gate's subject: the outputs of the LAST TIMED STEP (device-side copies, taken after the clock stopped):
    # the serialized pass below runs other launch forms (k_octree / k_blur apart, k_slots) and overwrites the arrays.
    steps_done = max(args.ramp_steps, 0) + Wu + K
    timed_out = {"set": [t.clone() for t in out_sets[cur["last"]]], "match": [t.clone() for t in (d_best, d_second, d_arg)],
                 "input": in_state["last"], "steps_done": steps_done}
    # Per-kernel durations for the roofline: the same K steps once more with every kernel on ONE stream, so a
    # kernel's HIP-event span is its own duration (in the timed region the blur runs beside FAST/octree and the
    # spans stretch each other).  Not part of `value`.
    stage_ms = {}
    if not args.no_stage_timing:
        ex.set_serialize(True)
        ex.enable_timing(True)
        for _ in range(K):
            step(time_match=True)
        barrier()
        stage_ms = ex.timing_ms()
        if match_events:
            stage_ms["match"] = sum(a.elapsed_time(b) for a, b in match_events) / len(match_events)
        ex.set_serialize(False)

    # Parity gate on rank 0: EVERY frame and EVERY match row of the last timed step's batch against the CPU oracle, bit for
    # bit (the oracle runs on all host threads; the reference's contract is per frame, Frame.cc:555-563); then the same for
    # the last step of the serialized pass (the other launch forms).
    n_kp = float(timed_out["set"][2][1:, 0].float().mean().item())
    parity = None
    if rank == 0:
        import oracle_lib as ol
        chk = BatchOracle(frames, nfeat, cap)

        def gate(kps_t, desc_t, counts_t, match_t, input_index, nsteps):
            idx = np.arange(B)[::-1].copy() if input_index == 1 else np.arange(B)
            if exchange:
                pidx = np.concatenate([[-1], idx[:-1]])  # row 0 is the boundary match below
            elif nsteps < 2:
                pidx = np.concatenate([[-1], idx[:-1]])  # no previous step: slot 0 is empty
            else:
                # slot 0 = the previous step's last frame: the other ordering's last frame when the inputs rotate
                pidx = np.concatenate([[idx[0] if len(d_gray_in) == 2 else idx[-1]], idx[:-1]])
            counts, kps_h, desc_h = counts_t.cpu().numpy(), kps_t.cpu().numpy(), desc_t.cpu().numpy()
            bad_f = chk.frames(idx, counts[1:], kps_h[1:], desc_h[1:])
            bad_r, nrows = [], 0
            if not args.no_match:
                bad_r, nrows = chk.match_rows(idx, pidx, *(t.cpu().numpy() for t in match_t))
            return gate_report(bad_f, B, bad_r, nrows, chk.threads)

        parity = gate(*timed_out["set"], timed_out["match"], timed_out["input"], timed_out["steps_done"])
        parity["subject"] = "the last step of the timed region (outputs copied on the device after the clock stopped)"
        if stage_ms:
            ser = gate(*out_sets[cur["last"]], (d_best, d_second, d_arg), in_state["last"], steps_done + K)
            parity["serialized_pass"] = {k: ser[k] for k in ("bit_exact_vs_oracle", "frames_checked", "match_rows_checked")}
            parity["bit_exact_vs_oracle"] = bool(parity["bit_exact_vs_oracle"] and ser["bit_exact_vs_oracle"])
        if exchange and not args.no_match and K + Wu >= 3:
            # the boundary match: rank 0's first frame against the LAST frame of the last rank, which only the
            # record exchange delivered -- the oracle extracts that remote frame itself
            ref = ol.OracleExtractor(nfeat, 1.2, 8, 20, 7)
            _, _, rd_first = ref(frames[0])
            _, _, rd_pred = ref(frame_of(world * B - 1))
            rb, rs, ra = ol.block_best2(rd_first, rd_pred)
            n = len(rb)
            okb = np.array_equal(d_bbest[0, :n].cpu().numpy(), rb) and np.array_equal(d_barg[0, :n].cpu().numpy(), ra)
            okb &= np.array_equal(d_bsecond[0, :n].cpu().numpy(), rs)
            parity["boundary_match_row"] = bool(okb)
            parity["bit_exact_vs_oracle"] = bool(parity["bit_exact_vs_oracle"] and okb)
    del timed_out

    if rank != 0:
        if distributed:
            dist.destroy_process_group()
        return

    total_frames = world * B * K
    fps = total_frames / dt
    stages, bytes_per_frame = algorithmic_bytes(ex, n_kp)
    roofline = None
    # the octree moves almost no bytes (latency-bound list surgery): it has no HBM roofline, so it is not a candidate
    timed = {k: stage_ms[k] for k in ("pyramid", "fast", "blur", "orient_desc", "match") if stage_ms.get(k, 0) > 0}
    if timed:
        dom = max(timed, key=timed.get)
        # the dominant kernel's launches were bracketed by HIP events inside the timed region itself (on its own stream,
        # nothing beside it): that figure prices the roofline; the serialized pass supplies the other stages
        if dom == "fast" and stage_ms_timed.get("fast", 0) > 0:
            timed[dom] = stage_ms_timed["fast"]
        dom_bytes = stages.get(dom, 0) * B  # algorithmic bytes one launch (batch of B frames) moves
        ach = dom_bytes / (timed[dom] * 1e-3) / 1e9 if timed[dom] > 0 else 0.0
        traffic, valu = None, None
        stage_traffic, issue, traffic_note = {}, {}, None
        try:  # PMC-derived HBM bytes / SQ counters per launch of every kernel, collected offline (profiles/)
            sys.path.insert(0, str(ROOT / "tools"))
            from source_hash import source_hash
            doc, traffic_source = {}, None
            for name in (TRAFFIC_FILE,):
                if (ROOT / "profiles" / name).exists():
                    doc = json.load(open(ROOT / "profiles" / name))
                    traffic_source = f"profiles/{name}"
                    break
            if doc and doc.get("source_hash") != source_hash():
                # counters of other sources say nothing about this build: refuse them instead of quoting them
                traffic_note = (f"{traffic_source} was measured on sources {doc.get('source_hash')}, this build is "
                                f"{source_hash()}: PMC figures withheld (re-run tools/profile_round.sh)")
                doc = {}
            isa = {}
            if (ROOT / "profiles" / ISA_MIX_FILE).exists():
                isa = json.load(open(ROOT / "profiles" / ISA_MIX_FILE))
                if isa.get("source_hash") != source_hash():
                    isa = {}
            per_stage = doc.get(f"{args.workload}/{B}", {})
            tr = per_stage.get(dom)
            # gfx950 correction of the guide (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies the 128-B requests of wide
            # (16 B per lane) streaming reads at 64 B, i.e. reports half their bytes -> doubled for the kernels whose
            # global reads are 16 B per lane (FAST's tile chunks, the pyramid's level-0 region); the others read 4-8 B
            # per lane, for which the counter is uncalibrated: raw value, a lower bound.  WRITE_SIZE is exact.
            wide = {"fast": 2, "pyramid": 2}
            if tr:
                traffic = wide.get(dom, 1) * tr["fetch_bytes"] + tr["write_bytes"]
                if not traffic_note:
                    traffic_note = (f"FETCH_SIZE x {wide.get(dom, 1)} (gfx950: 16-B-per-lane reads are tallied at half their "
                                    f"bytes) + WRITE_SIZE, per launch; raw counters in {traffic_source}")
            # Issue roofline per stage.  MI355X: 1024 SIMDs; a wave64 VALU instruction holds its SIMD's vector issue
            # for 2.3 cycles (plain 32-bit add / logic / fp32) or 4.2 (everything else these kernels use; measured:
            # profiles/r01_c_ubench_valu_rates.txt).  frac_at_4_cycles prices every instruction at the classic 4
            # cycles; frac_priced uses the kernel's STATIC full-rate / half-rate mix (profiles/r04_isa_mix.json).
            # wait_any = share of resident wave-cycles parked at s_waitcnt / barriers (SQ_WAIT_ANY / SQ_WAVE_CYCLES),
            # issue_stall = SQ_WAIT_INST_ANY share (the instruction buffer has work, the pipe is not free).
            for k, v in per_stage.items():
                if stage_ms.get(k, 0) > 0 and v.get("valu_insts"):
                    simd_cycles = 1024 * stage_ms[k] * 1e-3 * 2.4e9
                    e = {"valu_insts_per_launch": v["valu_insts"],
                         "frac_at_4_cycles": round(v["valu_insts"] * 4.0 / simd_cycles, 3)}
                    if k in isa:
                        e["frac_priced"] = round(v["valu_insts"] * isa[k]["priced_cycles_per_inst"] / simd_cycles, 3)
                        e["static_mix_cycles_per_inst"] = isa[k]["priced_cycles_per_inst"]
                    if v.get("wave_cycles"):
                        e["wait_any_frac"] = round(v.get("wait_any", 0) / v["wave_cycles"], 3)
                        e["issue_stall_frac"] = round(v.get("wait_inst_any", 0) / v["wave_cycles"], 3)
                        # SQ_WAVE_CYCLES counts quad-cycles summed over waves: resident waves per SIMD on average
                        e["waves_per_simd"] = round(v["wave_cycles"] * 4.0 / simd_cycles, 2)
                    issue[k] = e
            if issue:
                issue["_source"] = (f"{traffic_source} (rocprofv3 --pmc SQ_* passes of tools/profile_round.sh, source hash "
                                    f"{doc.get('source_hash')}); durations = stage_ms of this run; both describe the "
                                    "SERIALISED step (VSG_NO_OVERLAP: k_octree and k_blur as separate launches) -- in the "
                                    "timed region the blur's workgroups ride in k_octree_blur")
            # PMC traffic over algorithmic bytes per stage: well above 1 = wasted re-reads (the first thing to fix)
            for k, v in per_stage.items():
                if k in stages and stages[k] > 0 and "fetch_bytes" in v:
                    pmc_b = wide.get(k, 1) * v["fetch_bytes"] + v.get("write_bytes", 0)
                    stage_traffic[k] = {"pmc_bytes": pmc_b, "fetch_correction": wide.get(k, 1),
                                        "algorithmic_bytes": int(stages[k] * B),
                                        "ratio": round(pmc_b / (stages[k] * B), 2)}
            if stage_traffic:
                stage_traffic["_source"] = traffic_source + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
        except (OSError, ValueError, KeyError, ImportError):
            pass
        # what the counters say limits the dominant kernel (the HBM roof below is the contract's pricing, not a claim
        # that the kernel is bandwidth-bound)
        limited_by = None
        if issue.get(dom):
            d_ = issue[dom]
            limited_by = ("VALU issue: %.0f %% of the vector issue slots at 4 cycles per instruction" % (100 * d_["frac_at_4_cycles"])
                          + (", %.0f %% priced by the static instruction mix" % (100 * d_["frac_priced"]) if "frac_priced" in d_ else "")
                          + ("; waves parked %.0f %% of their cycles" % (100 * d_["wait_any_frac"]) if "wait_any_frac" in d_ else ""))
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                    **({"limited_by": limited_by} if limited_by else {}),
                    **({"issue": issue} if issue else {}),
                    **({"traffic_note": traffic_note} if traffic_note else {}),
                    **({"stage_traffic_vs_algorithmic": stage_traffic} if stage_traffic else {}),
                    "launch_ms": round(timed[dom], 4), "bytes_per_launch": int(dom_bytes),
                    "timing": "launch_ms: HIP events around the dominant kernel's launches inside the timed region, on the "
                              "stream it is launched on; stage_ms: HIP events per kernel in a serialized pass of the same "
                              "steps (in the timed region the blur's workgroups ride in the octree's launch)",
                    "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
                    # every stage against the same HBM roof: algorithmic bytes per launch / its duration
                    "stage_algorithmic_GBs": {k: round(stages[k] * B / (v * 1e-3) / 1e9, 1)
                                              for k, v in stage_ms.items() if k in stages and v > 0},
                    **({"stage_ms_timed_region": {k: round(v, 4) for k, v in stage_ms_timed.items() if v > 0}}
                       if any(v > 0 for v in stage_ms_timed.values()) else {}),
                    "pipeline_achieved_GBs": round(bytes_per_frame * fps / world / 1e9, 2),
                    "bytes_per_frame": int(bytes_per_frame)}

    cpu = None
    extra = {}
    if args.cpu_seconds > 0 and world > 1:
        # N > 1 lines carry the same baseline on a short budget (rank 0's host, one thread; the other ranks are done)
        sample = [frames[i] for i in range(min(B, 16))]
        budget = min(3.0, args.cpu_seconds)
        v1, n1 = cpu_baseline(W, H, nfeat, sample, budget, 1)
        cpu = {"value": round(v1, 2), "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": f"{n1} frames of the same {W}x{H}/{nfeat} workload, extract + brute-force match, "
                         f"{budget:.0f} s on rank 0's host, CPU oracle (port of the reference algorithm)"}
    if args.cpu_seconds > 0 and world == 1:
        sample = [frames[i] for i in range(min(B, 16))]
        v1, n1 = cpu_baseline(W, H, nfeat, sample, args.cpu_seconds, 1)
        cpu = {"value": round(v1, 2), "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": f"{n1} frames of the same {W}x{H}/{nfeat} workload, extract + brute-force match, "
                         f"{args.cpu_seconds:.0f} s, CPU oracle (port of the reference algorithm)"}
        v2, n2 = cpu_baseline(W, H, nfeat, sample, max(2.0, args.cpu_seconds / 2), 2)
        extra["cpu_baseline_2_threads"] = {"value": round(v2, 2), "unit": "frames/s", "cores": 2, "kind": "port",
                                           "sample": f"{n2} frames; the reference's stereo analogue (Frame.cc:129-132: "
                                                     "one extractor per eye on two host threads)"}
        ncores = effective_cores()  # the box advertises 256 hardware threads but the cgroup quota is what we get
        va, na = cpu_baseline(W, H, nfeat, sample, args.cpu_seconds, ncores)
        extra["cpu_baseline_all_cores"] = {"value": round(va, 2), "unit": "frames/s", "cores": ncores, "kind": "port",
                                           "sample": f"{na} frames in {args.cpu_seconds:.0f} s, one independent "
                                                     "extractor per host thread (native std::thread); cores = "
                                                     f"min(affinity, cgroup cpu quota) of {os.cpu_count()} hw threads"}

    out = {
        "metric": "frames/sec ORB extract+match @640x480, 1000 feats; bit-exact vs CPU",
        "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wu,
        "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {W}x{H} gray, nFeatures={nfeat}, 8 levels, scale 1.2, FAST 20/7, "
                               f"extract{'' if args.no_match else ' + brute-force Hamming best2 match vs previous frame'}",
                   "frames_per_step_per_gpu": B, "parallelism": f"frame-sharded x{world}"
                   + (f", contiguous chunks of one sequence per step; "
                      + ("all-gather of every rank's keypoint/descriptor records" if args.exchange == "allgather" else
                         "neighbour-only exchange of the chunk boundary record (last frame -> successor rank)")
                      + f" per step via {comm_kind} on a second stream under the next step's kernels; every rank's "
                      "first frame is matched against the received last frame of the previous rank"
                      if exchange else ""),
                   **({"exchange": args.exchange,
                       "exchange_bytes_in_per_gpu_per_step": int((world - 1) * B * rec_bytes if args.exchange == "allgather"
                                                                 else rec_bytes),
                       # ncclCommCount / ncclCommUserRank on the live communicator (not an echo of the arguments)
                       "rccl_ranks_seen": comm.world_seen() if comm is not None else None,
                       "rccl_rank_seen": comm.rank_seen() if comm is not None else None,
                       "dist_world_size": dist.get_world_size()} if exchange else {}),
                   **({"match_overlap": "the match of step k runs on a second HIP stream under the extraction of step "
                                        "k + 1 (two alternating output sets); all K matches end inside the timed region"}
                      if overlap_match["on"] else {}),
                   "keypoints_per_frame": round(n_kp, 1),
                   "content": ("rectangles (synth.sequence_frame: rectangles + uniform noise, translated per frame); the other "
                               "nine generated classes and three real photographs: `content_sweep`, the natural-image stand-ins "
                               "also as `value_value_noise` / `value_defocus`, the slowest photograph as `value_photo`")
                   if args.content == "rectangles" else
                   f"{args.content} (synth.content_frame, --content: NOT the content `value` is quoted on)",
                   "clock_ramp": f"{max(args.ramp_steps, 0)} untimed steps before the {Wu} warm-up steps (the GPU reaches its "
                                 "steady clocks after ~0.1-0.2 s of load)",
                   "inputs": "resident in HBM" + (": two batches (the frames in forward / reverse order) alternate from step to "
                                                  "step" if rotate else "")},
        "parity": parity,
        "roofline": roofline, "cpu_baseline": cpu,
    }
    out.update(extra)
    if world == 1 and not args.no_extras:
        # claims the driver cannot otherwise see (host API, matcher latency, the other BASELINE configurations, the content
        # sweep incl. real photographs, single-frame latency), each behind a short budget, all OUTSIDE the timed region:
        # tools/bench_extras.py
        del d_gray, d_gray_in, d_kps, d_desc, d_best, d_second, d_arg, out_sets
        torch.cuda.empty_cache()
        bench_extras.add_extras(out, W, H, nfeat, B, local_rank, args.cpu_seconds)
    print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
