"""bench.py's legs OUTSIDE the timed region (VERDICT r5 #8d: the contract line's timed region, gate and roofline stay in
bench.py): throughput through the host API, the matcher's call latency, the other BASELINE.json configurations (C3 pair chain, C4,
C5, C3's eyes as a batch), the content sweep over the ten generated classes and three real photographs, single-frame latency.
Each leg checks its outputs against the oracle before it reports a rate."""
import json
import subprocess
import sys
import time

import numpy as np

from bench_common import (HBM_PEAK_GBS, ROOT, WORKLOADS, BatchOracle, algorithmic_bytes, cpu_baseline, effective_cores,  # noqa: F401
                          gate_report, pmc_traffic)


def host_api_leg(W, H, nfeat, device, batch=64, seconds=1.5):
    """Throughput THROUGH the drop-in boundary: host uint8 frames in, keypoint / descriptor records out
    (vsg_orb_submit_batch / vsg_orb_wait, three batches in flight), with pinned (vsg_host_alloc = hipHostMalloc) and with
    pageable caller memory, every frame of every slot's last batch bit-compared with the oracle, and the latency of one
    blocking single-frame operator().  PCIe-inclusive: never `value`."""
    from visual_sgraphs_amd import orb, synth
    ex = orb.ORBextractor(nfeat, 1.2, 8, 20, 7, device=device, max_batch=batch)
    cap = ex.capacity(H, W)
    nslot = ex.slots()
    ring_in = [np.stack([synth.sequence_frame(W, H, 2000 + r, t) for t in range(batch)]) for r in range(nslot)]
    out = {"batch": batch, "slots": nslot}
    chk = BatchOracle(np.concatenate(ring_in), nfeat, cap)
    for mode in ("pinned", "pageable"):
        # pinned = memory from vsg_host_alloc (hipHostMalloc): the device reads the frames and writes the records in place
        owners = []
        if mode == "pinned":
            ins, outs = [], []
            for a in ring_in:
                pa = orb.PinnedArray(a.shape)
                pa.a[...] = a
                pk, pd = orb.PinnedArray((batch, cap), orb.KP_DTYPE), orb.PinnedArray((batch, cap, 32))
                owners += [pa, pk, pd]
                ins.append(pa.a), outs.append((pk.a, pd.a))
        else:
            ins = ring_in
            outs = [(np.zeros((batch, cap), orb.KP_DTYPE), np.zeros((batch, cap, 32), np.uint8)) for _ in range(nslot)]
        tickets, done, k = [], 0, 0
        last = {}
        t_end = None
        t0 = time.perf_counter()
        warm = 2 * nslot
        while True:
            if k == warm:
                t0 = time.perf_counter()
                t_end = t0 + seconds
            tickets.append((k % nslot, ex.submit_batch(ins[k % nslot], *outs[k % nslot])))
            k += 1
            if len(tickets) == nslot:
                r, t = tickets.pop(0)
                last[r] = ex.wait(t)
                done += 1
            if t_end is not None and time.perf_counter() >= t_end:
                break
        while tickets:
            r, t = tickets.pop(0)
            last[r] = ex.wait(t)
            done += 1
        dt = time.perf_counter() - t0
        out[f"{mode}_frames_per_s"] = round((k - warm) * batch / dt, 1)
        out[f"{mode}_keypoints_last_frame"] = int(last[(k - 1) % nslot][0][-1])
        # every frame of the last batch of every slot against the oracle
        bad = []
        for r, (n, mono) in last.items():
            counts = np.stack([n, mono], axis=1)
            bad += [r * batch + f for f in chk.frames(np.arange(batch) + r * batch, counts, outs[r][0], outs[r][1])]
        out[f"{mode}_parity"] = {"bit_exact_vs_oracle": not bad, "frames_checked": len(last) * batch,
                                 **({"frames_differing": bad[:16]} if bad else {})}
        del ins, outs
        for o in owners:
            o.free()
    img = ring_in[0][0]
    for _ in range(20):
        ex(img)
    t0 = time.perf_counter()
    reps = 200
    for _ in range(reps):
        ex(img)
    out["single_frame_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 4)
    out["note"] = ("host memory in, host memory out through vsg_orb_submit_batch / vsg_orb_wait (H2D, kernels and the "
                   "n-sized export of three batches overlap); PCIe-inclusive, not `value`")
    return out


def matcher_latency_leg():
    """Per-call latency of the per-frame ORBmatcher entry points on device-resident frames, from plain C++
    (tools/abi_latency.cpp), next to the CPU oracle's routine on one host thread."""
    import subprocess
    exe = ROOT / "tools" / "_bin" / "abi_latency"
    if not exe.exists():
        return {"error": "tools/_bin/abi_latency not built (make -C tools)"}
    try:
        r = subprocess.run([str(exe), "300"], capture_output=True, text=True, timeout=120)
        if r.returncode != 0:
            return {"error": r.stderr.strip()[-300:]}
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}


def config_chain_leg(seconds=2.0, pipelines=4):
    """BASELINE configs C3 (stereo pair -> ComputeStereoMatches -> ComputeBoW on a k=10, L=6 vocabulary -> SearchByBoW,
    all on device-resident frames) and C5 (four concurrent 1250-feature camera streams) as BASELINE.json states them,
    from plain C++ through the C ABI (tools/config_chain.cpp); every output is bit-compared with the same chain on the
    CPU oracle first, whose rate is reported beside the GPU's."""
    import subprocess
    exe = ROOT / "tools" / "_bin" / "config_chain"
    if not exe.exists():
        return {"error": "tools/_bin/config_chain not built (make -C tools)"}
    try:
        r = subprocess.run([str(exe), str(seconds), str(pipelines)], capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            return {"error": r.stderr.strip()[-300:]}
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}


def device_rate(workload, batch, steps, device, cpu_seconds=3.0, uniq=None, label=None, ex=None):
    """Another BASELINE config -- or another CONTENT CLASS of the headline config (`uniq`: the distinct frames the batch
    cycles through) -- on the main bench's terms (frames resident in HBM, one batch per step): extract + brute-force
    best2 match of every frame against its predecessor, EVERY frame and EVERY match row of the last step bit-compared with
    the CPU oracle (all host threads), the oracle's own rate on one host thread beside it, the FAST kernel's launch
    duration by HIP events around its launches inside the timed steps."""
    import ctypes as C
    import torch
    import oracle_lib as ol
    from visual_sgraphs_amd import orb, synth
    W, H, nfeat = WORKLOADS[workload]
    dev = torch.device("cuda", device)
    if ex is None:
        ex = orb.ORBextractor(nfeat, 1.2, 8, 20, 7, device=device, max_batch=batch)
    cap = ex.capacity(H, W)
    if uniq is None:
        uniq = np.stack([synth.sequence_frame(W, H, 3000, t) for t in range(min(batch, 16))])
    nuniq = len(uniq)
    frames = np.concatenate([uniq] * ((batch + nuniq - 1) // nuniq))[:batch]
    d_gray = torch.from_numpy(frames).to(dev)
    d_kps = torch.zeros((batch + 1, cap, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((batch + 1, cap, 32), dtype=torch.uint8, device=dev)
    d_counts = torch.zeros((batch + 1, 2), dtype=torch.int32, device=dev)
    d_best, d_second, d_arg = (torch.zeros((batch, cap), dtype=torch.int32, device=dev) for _ in range(3))
    st = torch.cuda.Stream(device=dev)
    L = orb.load_library()
    vp = C.c_void_p
    with torch.cuda.stream(st):
        warm = 150  # untimed: the GPU's clocks have dropped during the CPU legs before this one (see --ramp-steps)
        for i in range(steps + warm):
            if i == warm:
                torch.cuda.synchronize()
                ex.enable_timing(2)  # events around the FAST launches only
                t0 = time.perf_counter()
            d_desc[0].copy_(d_desc[batch])
            d_counts[0].copy_(d_counts[batch])
            ex.extract_batch_device(d_gray.data_ptr(), batch, H * W, H, W, W, d_kps[1].data_ptr(), d_desc[1].data_ptr(),
                                    d_counts[1].data_ptr(), cap, (0, 0), st.cuda_stream)
            rc = L.vsg_hamming_block_best2_device(device, vp(d_desc[1].data_ptr()), vp(d_desc[0].data_ptr()), cap * 32,
                                                  vp(d_counts[1].data_ptr()), vp(d_counts[0].data_ptr()), 2, batch, cap,
                                                  vp(d_best.data_ptr()), vp(d_second.data_ptr()), vp(d_arg.data_ptr()),
                                                  vp(st.cuda_stream))
            assert rc == 0, rc
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fast_ms = ex.timing_ms().get("fast")
    ex.enable_timing(0)
    counts, kps_h, desc_h = d_counts.cpu().numpy(), d_kps.cpu().numpy(), d_desc.cpu().numpy()
    # the gate: EVERY frame and EVERY match row of the last step's batch against the oracle (all host threads)
    chk = BatchOracle(uniq, nfeat, cap)
    idx = np.arange(batch) % nuniq
    pidx = np.concatenate([[idx[-1]], idx[:-1]])  # row 0: against the previous (identical) step's last frame
    bad_f = chk.frames(idx, counts[1:], kps_h[1:], desc_h[1:])
    bad_r, nrows = chk.match_rows(idx, pidx, d_best.cpu().numpy(), d_second.cpu().numpy(), d_arg.cpu().numpy())
    gate = gate_report(bad_f, batch, bad_r, nrows, chk.threads)
    ok = gate["bit_exact_vs_oracle"]
    v1, n1 = ol.bench_throughput(uniq, nfeat, 1, cpu_seconds, do_match=True)
    # the dominant kernel (FAST) against the HBM roof, as the headline prices it: P bytes per frame x the frames of one
    # launch / the launch's HIP-event duration inside the timed steps
    roof = None
    if fast_ms:
        stages, bytes_per_frame = algorithmic_bytes(ex, float(counts[1:, 0].mean()))
        ach = stages["fast"] * batch / (fast_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": "fast", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(f"{workload}/{batch}", "fast"),
                "launch_ms": round(fast_ms, 4), "bytes_per_launch": int(stages["fast"] * batch),
                "pipeline_achieved_GBs": round(bytes_per_frame * batch * steps / dt / 1e9, 2),
                "bytes_per_frame": int(bytes_per_frame)}
    return {"workload": label or f"{workload}: {W}x{H}, nFeatures={nfeat}, extract + brute-force best2 match vs previous "
                                 f"frame, {batch}-frame batches resident in HBM", "unit": "frames/s", "frames_per_step": batch,
            "frames_per_s": round(batch * steps / dt, 1), "keypoints_per_frame": round(float(counts[1:, 0].mean()), 1),
            "fast_ms": round(fast_ms, 4) if fast_ms else None,
            "parity": bool(ok), "parity_gate": gate, "roofline": roof,
            "cpu_oracle": {"frames_per_s": round(v1, 2), "threads": 1, "kind": "port", "frames": n1}}


def content_sweep_leg(device, batch=512, steps=12, nuniq=32, cpu_seconds=1.0):
    """The headline workload (C2: 640x480 / 1000, extract + match, `batch`-frame batches resident in HBM) on every
    content class of synth.CONTENT_CLASSES: frames/s, the FAST kernel's launch time, the parity flag and the CPU oracle's
    rate per class -- the spread of `value` over image statistics (a headline measured on rectangles +
    noise alone has no error bar).  The batch cycles through `nuniq` distinct frames of the class (consecutive frames of
    one translated sequence)."""
    from visual_sgraphs_amd import orb, synth
    W, H, nfeat = WORKLOADS["C2"]
    ex = orb.ORBextractor(nfeat, 1.2, 8, 20, 7, device=device, max_batch=batch)
    out = {}
    for kind in synth.CONTENT_CLASSES:
        uniq = np.stack([synth.content_frame(kind, W, H, 5000, t) for t in range(nuniq)])
        try:
            r = device_rate("C2", batch, steps, device, cpu_seconds, uniq=uniq, label=kind, ex=ex)
            out[kind] = {k: r[k] for k in ("frames_per_s", "fast_ms", "keypoints_per_frame", "parity")}
            out[kind]["frames_checked"] = r["parity_gate"]["frames_checked"]
            out[kind]["match_rows_checked"] = r["parity_gate"]["match_rows_checked"]
            out[kind]["cpu_oracle_frames_per_s"] = r["cpu_oracle"]["frames_per_s"]
        except Exception as e:  # noqa: BLE001
            out[kind] = {"error": str(e)}
    good = [v["frames_per_s"] for v in out.values() if "frames_per_s" in v]
    fast = [v["fast_ms"] for v in out.values() if v.get("fast_ms")]
    return {"workload": f"C2 geometry, extract + match, {batch}-frame batches resident in HBM ({nuniq} distinct frames per "
                        "class, cycled), one entry per content class of synth.CONTENT_CLASSES",
            "classes": out, "frames_per_s_min": min(good) if good else None, "frames_per_s_max": max(good) if good else None,
            "fast_ms_min": min(fast) if fast else None, "fast_ms_max": max(fast) if fast else None,
            "all_parity": all(v.get("parity") is True for v in out.values())}


def add_extras(out, W, H, nfeat, B, local_rank, cpu_seconds):
    """Fill `out` (the JSON line) with every leg above.  The caller has released the timed region's device arrays."""
    args_cpu_seconds = cpu_seconds
    try:
        out["host_api"] = host_api_leg(W, H, nfeat, local_rank)
    except Exception as e:  # noqa: BLE001
        out["host_api"] = {"error": str(e)}
    out["matcher_latency"] = matcher_latency_leg()
    # the other BASELINE.json configurations as stated there, each with its parity flag and the CPU oracle's rate
    other = []
    chain = config_chain_leg()
    c3 = chain.get("C3", {"workload": "C3", **chain})
    if isinstance(c3.get("stage_ms"), dict) and c3["stage_ms"].get("extract_2_eyes"):
        # the pair chain is ONE stereo pair per blocking call: dependency-bound, priced against the same HBM roof for
        # the record (5 630 695 algorithmic bytes per 752x480 / 1200 eye; both eyes extract side by side)
        ach = 2 * 5630695 / (c3["stage_ms"]["extract_2_eyes"] * 1e-3) / 1e9
        c3["roofline"] = {"bound": "hbm", "kernel": "operator() of both eyes (one pair per call: latency-bound)",
                          "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                          "traffic": None, "launch_ms": c3["stage_ms"]["extract_2_eyes"], "bytes_per_launch": 2 * 5630695,
                          "counters": "profiles/r05_*_c3_chain_pmc.txt (k_stereo, k_bow_descend, k_search_by_bow per call)"}
    other.append(c3)
    try:
        r4 = device_rate("C4", 256, 10, local_rank)
        try:  # the same configuration on a real photograph (~20 k level-0 candidates per frame: the octree's memory-resident form)
            from visual_sgraphs_amd import synth as _s
            W4, H4, _ = WORKLOADS["C4"]
            uniq = np.stack([_s.content_frame("photo_china", W4, H4, 5000, t) for t in range(16)])
            rp = device_rate("C4", 256, 10, local_rank, cpu_seconds=1.0, uniq=uniq, label="photo_china")
            r4["photo_china"] = {k: rp[k] for k in ("frames_per_s", "fast_ms", "keypoints_per_frame", "parity")}
            r4["photo_china"]["frames_checked"] = rp["parity_gate"]["frames_checked"]
        except Exception as e:  # noqa: BLE001
            r4["photo_china"] = {"error": str(e)}
        other.append(r4)
    except Exception as e:  # noqa: BLE001
        other.append({"workload": "C4", "error": str(e)})
    other.append(chain.get("C5", {"workload": "C5", **chain}))
    try:  # C3's eyes as a throughput batch (the chain above is the per-pair latency form): its FAST roofline
        r3 = device_rate("C3", 256, 10, local_rank)
        r3["workload"] = "C3 eyes as a batch: " + r3["workload"]
        other.append(r3)
    except Exception as e:  # noqa: BLE001
        other.append({"workload": "C3 eyes as a batch", "error": str(e)})
    out["other_configs"] = other
    try:
        out["content_sweep"] = content_sweep_leg(local_rank, batch=B, cpu_seconds=min(1.0, max(0.3, args_cpu_seconds / 8)))
    except Exception as e:  # noqa: BLE001
        out["content_sweep"] = {"error": str(e)}
    # `value` is measured on rectangles + noise (config.content), the most favourable but one of the ten classes; the same
    # workload on the natural-image stand-ins -- every FAST cell empty at iniThFAST, the reference's second pass at
    # minThFAST on all of them -- beside it, from the sweep above (every frame checked there too)
    cls = out["content_sweep"].get("classes", {}) if isinstance(out["content_sweep"], dict) else {}
    for kind in ("value_noise", "defocus"):
        if isinstance(cls.get(kind), dict) and "frames_per_s" in cls[kind]:
            out[f"value_{kind}"] = cls[kind]["frames_per_s"]
    # ... and on REAL photographs (round 6: tests/golden/photos_v1.npz, gray planes committed as data): the slowest of the
    # three, its class, and the parity flag over every frame and match row of its batch
    from visual_sgraphs_amd import synth as _synth
    photos = {k: cls[k] for k in _synth.PHOTO_CLASSES if isinstance(cls.get(k), dict) and "frames_per_s" in cls[k]}
    if photos:
        worst = min(photos, key=lambda k: photos[k]["frames_per_s"])
        out["value_photo"] = photos[worst]["frames_per_s"]
        out["value_photo_detail"] = {"class": worst, "parity": photos[worst]["parity"],
                                     "frames_checked": photos[worst]["frames_checked"],
                                     "match_rows_checked": photos[worst]["match_rows_checked"],
                                     "fast_ms": photos[worst]["fast_ms"],
                                     "all_photos": {k: v["frames_per_s"] for k, v in photos.items()},
                                     "all_photos_parity": all(v["parity"] is True for v in photos.values())}
    # the call pattern the reference has: ONE frame per blocking operator() (System.cc:359, Tracking.cc:1583,
    # Frame.cc:344,555-563), from plain C++ through the C ABI, with the CPU oracle's chain beside each figure
    fl = dict(chain.get("frame_latency") or {"error": chain.get("error", "config_chain gave no frame_latency")})
    c3 = chain.get("C3") or {}
    if "stage_ms" in c3:
        st = c3["stage_ms"]
        fl["stereo_pair_ms"] = round(st["extract_2_eyes"] + st["make_resident_2"] + st["stereo_matches"], 4)
        fl["stereo_pair"] = ("752x480 / 1200: two handles on two host threads (Frame.cc:129-132) -> both eyes resident "
                             "-> ComputeStereoMatches (Frame.cc:957)")
        bp = c3.get("batched_pair") or {}
        if "extract_2_eyes" in bp:
            fl["stereo_pair_batched_ms"] = round(bp["extract_2_eyes"] + bp["make_resident_2"] + bp["stereo_matches"], 4)
            fl["stereo_pair_batched"] = bp.get("what", "") + f"; parity {bp.get('parity')}"
    out["frame_latency"] = fl
