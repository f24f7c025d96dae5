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

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
TRAFFIC_FILE = "traffic_r06.json"   # profiles/: PMC counters per launch (tools/profile_round.sh), tied to a source hash
ISA_MIX_FILE = "r06_isa_mix.json"


def pmc_traffic(key, stage):
    """HBM bytes per launch of `stage` for workload key (e.g. "C4/128") from the committed counter file -- FETCH_SIZE
    (doubled for the 16-B-per-lane readers, MI355X_MICROARCH.md HBM section) + WRITE_SIZE -- or None when the file is
    absent or was measured on other sources."""
    try:
        sys.path.insert(0, str(ROOT / "tools"))
        from source_hash import source_hash
        doc = json.load(open(ROOT / "profiles" / TRAFFIC_FILE))
        if doc.get("source_hash") != source_hash():
            return None
        tr = doc.get(key, {}).get(stage)
        if not tr or "fetch_bytes" not in tr:
            return None
        return {"fast": 2, "pyramid": 2}.get(stage, 1) * tr["fetch_bytes"] + tr.get("write_bytes", 0)
    except (OSError, ValueError, KeyError, ImportError):
        return None

WORKLOADS = {
    # name: (W, H, nfeatures)
    "C2": (640, 480, 1000),
    "C3": (752, 480, 1200),
    "C4": (1280, 720, 2000),
    "C5": (640, 480, 1250),
}


def algorithmic_bytes(ex, n_kp):
    """Per-frame algorithmic bytes per stage: one read/write per unavoidable stage boundary."""
    sizes = [ex.level_size(l) for l in range(ex.nlevels)]
    P = sum(w * h for w, h in sizes)
    wh0 = sizes[0][0] * sizes[0][1]
    whl = sizes[-1][0] * sizes[-1][1]
    stages = {
        "ingest": wh0,
        "pyramid": (P - whl) + (P - wh0),
        "fast": P,
        "blur": 2 * P,
        "orient_desc": 60 * n_kp,
        "match": 64 * n_kp,  # both descriptor sets read once (32*(n_q + n_t) bytes per block)
    }
    return stages, sum(v for k, v in stages.items() if k != "match")


class BatchOracle:
    """The parity gate's checker: the CPU oracle's operator() output for the DISTINCT frames of a workload, computed once
    on all host threads (or_extract_batch_mt), against which EVERY frame and EVERY match row of a device batch is
    compared bit for bit (a batch position maps to its distinct frame through `idx`)."""

    def __init__(self, uniq, nfeat, cap):
        import oracle_lib as ol
        self.ol, self.cap, self.uniq = ol, cap, uniq
        self.threads = ol.host_threads()
        self.counts, self.kps, self.desc = ol.extract_batch(uniq, nfeat, cap, nthreads=self.threads)
        self._rows = {}

    def frames(self, idx, counts, kps, desc):
        """device outputs [B, ...] of the frames uniq[idx[f]]: list of differing batch positions"""
        idx = np.asarray(idx)
        return self.ol.compare_batch(counts, kps, desc, self.counts[idx], self.kps[idx], self.desc[idx])

    def match_rows(self, idx, pidx, best, second, arg):
        """row f = brute-force best2 of frame uniq[idx[f]] against uniq[pidx[f]] (pidx[f] < 0: row not checked)"""
        need = sorted({(int(i), int(j)) for i, j in zip(idx, pidx) if j >= 0} - set(self._rows))
        if need:
            a = np.stack([self.desc[i] for i, _ in need])
            b = np.stack([self.desc[j] for _, j in need])
            rb, rs, ra = self.ol.block_best2_batch(a, [self.counts[i, 0] for i, _ in need], b,
                                                   [self.counts[j, 0] for _, j in need], nthreads=self.threads)
            for k, p in enumerate(need):
                self._rows[p] = (rb[k], rs[k], ra[k])
        bad, checked = [], 0
        for f, (i, j) in enumerate(zip(idx, pidx)):
            if j < 0:
                continue
            checked += 1
            rb, rs, ra = self._rows[(int(i), int(j))]
            n = int(self.counts[i, 0])
            if not (np.array_equal(best[f, :n], rb[:n]) and np.array_equal(second[f, :n], rs[:n])
                    and np.array_equal(arg[f, :n], ra[:n])):
                bad.append(f)
        return bad, checked


def gate_report(bad_frames, nframes, bad_rows, nrows, threads):
    ok = not bad_frames and not bad_rows
    rep = {"bit_exact_vs_oracle": bool(ok), "checked_frames": "all", "frames_checked": int(nframes),
           "match_rows_checked": int(nrows), "oracle_threads": int(threads)}
    if not ok:
        rep["frames_differing"] = [int(f) for f in bad_frames[:16]]
        rep["match_rows_differing"] = [int(f) for f in bad_rows[:16]]
    return rep


def effective_cores():
    """Host threads this process may really use: min(affinity, cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(w, h, nfeatures, frames, budget_s, threads):
    """Time the CPU oracle (kind 'port') on `threads` native host threads for about budget_s seconds."""
    import oracle_lib as ol
    return ol.bench_throughput(np.stack(frames), nfeatures, threads, budget_s, do_match=True)


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


def wants_library_exchange(backend, one_device, torch_gather):
    """The record exchange of a real multi-GPU run goes through the library's own RCCL communicator (vsg_shard_*);
    torch.distributed carries it only in the rehearsals: gloo, --one-device, or an explicit --torch-gather."""
    return backend == "nccl" and not one_device and not torch_gather


def refuse_without_library_exchange(rank, world, why):
    """--gpus N > 1 on the nccl backend without --torch-gather and vsg_shard_create failed on some rank: exit non-zero with
    the reason, on every rank, instead of measuring torch's all-gather under this library's name."""
    msg = (f"[bench] rank {rank}/{world}: vsg_shard_create failed on at least one rank"
           + (f" (here: {why})" if why else " (not on this one)")
           + "; refusing to fall back to torch.distributed -- pass --torch-gather to measure that on purpose")
    print(msg, file=sys.stderr)
    raise SystemExit(3)


def check_rccl_world(seen, world, rank=0):
    """The multi-GPU line is printed only if the live communicator (ncclCommCount) spans exactly the launched ranks."""
    if seen != world:
        print(f"[bench] rank {rank}: RCCL communicator spans {seen} ranks, launched {world}: no line", file=sys.stderr)
        raise SystemExit(4)
    return True


def visible_gpu_count():
    """GPUs this process may use, WITHOUT importing torch or initialising HIP (the launcher below must not touch the GPU
    before it starts its ranks): the visibility variables if set, else the KFD topology (nodes with SIMDs); None when
    neither answers -- the ranks then report a missing device themselves."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    n = 0
    try:
        for node in Path("/sys/class/kfd/kfd/topology/nodes").iterdir():
            props = dict(line.split()[:2] for line in (node / "properties").read_text().splitlines() if " " in line)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        if n:
            return n
    except (OSError, ValueError):
        pass
    return None


def preflight(n, batch, workload, one_device=False, quiet=False):
    """`bench.py --gpus N --preflight`: what a first N-GPU run can trip over, checked in a CHILD process (it touches
    the GPU) before any rank is started -- visible devices; that libvsg_orb.so and torch resolve the SAME libamdhip64
    (bench.py hands torch streams and device pointers to a library that links the runtime by soname: INTEGRATION.md
    section 5); that every pair of the N devices is peer-accessible (hipDeviceCanAccessPeer) and over which link
    (hipExtGetLinkTypeAndHopCount: xgmi / pcie); that the RCCL the library will dlopen exports what vsg_shard_* binds,
    ncclCommCount included; the record exchange's receive buffer (world x batch records per rank) against the free memory of every device.  Prints one JSON
    object; exit code 0 only when every check passed."""
    import subprocess
    code = r"""
import ctypes as C, json, os, sys
sys.path.insert(0, %r)
n, batch, workload, one_device = %d, %d, %r, %r
out = {"requested_gpus": n, "checks": {}}
def check(name, ok, **info):
    out["checks"][name] = dict(ok=bool(ok), **info)
class DlInfo(C.Structure):
    _fields_ = [("fname", C.c_char_p), ("fbase", C.c_void_p), ("sname", C.c_char_p), ("saddr", C.c_void_p)]
libdl = C.CDLL(None)
libdl.dladdr.argtypes = [C.c_void_p, C.POINTER(DlInfo)]
def owner(lib, sym):
    addr = C.cast(getattr(lib, sym), C.c_void_p).value
    info = DlInfo()
    libdl.dladdr(addr, C.byref(info))
    return addr, (info.fname or b"?").decode()
import torch
from visual_sgraphs_amd import orb, sharding
import bench
L = orb.load_library()
tlib = None
tdir = os.path.join(os.path.dirname(torch.__file__), "lib")
for cand in ("libtorch_hip.so", "libc10_hip.so"):
    try:
        tlib = C.CDLL(os.path.join(tdir, cand)); break
    except OSError:
        pass
try:
    a_vsg, f_vsg = owner(L, "hipMalloc")
    a_t, f_t = owner(tlib, "hipMalloc") if tlib is not None else (None, "torch's HIP library not found")
    check("one_hip_runtime", a_vsg == a_t and os.path.realpath(f_vsg) == os.path.realpath(f_t), libvsg_orb=f_vsg, torch=f_t)
except Exception as e:
    check("one_hip_runtime", False, error=repr(e))
ndev_t = torch.cuda.device_count()
ndev_v = L.vsg_device_count()
need_dev = 1 if one_device else n  # --one-device: every rank on device 0 (dry runs of the multi-rank path)
check("devices", ndev_t >= need_dev and ndev_v >= need_dev, torch_device_count=ndev_t, vsg_device_count=ndev_v,
      visible_without_runtime=bench.visible_gpu_count(), needed=need_dev)
# every pair of the N devices peer-accessible, and over which link (round 6, VERDICT r5 #5c): RCCL's all-gather and the
# neighbour send / recv go device to device; a pair without peer access would fall back to staging through the host
try:
    hip = C.CDLL(f_vsg)  # the libamdhip64 the process already runs on
    hip.hipDeviceCanAccessPeer.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int]
    hip.hipExtGetLinkTypeAndHopCount.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    LINK = {0: "hypertransport", 1: "qpi", 2: "pcie", 3: "infiniband", 4: "xgmi"}
    pairs, bad = [], []
    ndev_p = 0 if one_device else min(n, ndev_t)
    for a in range(ndev_p):
        for b in range(ndev_p):
            if a == b:
                continue
            can, lt, hops = C.c_int(0), C.c_uint32(99), C.c_uint32(0)
            rc1 = hip.hipDeviceCanAccessPeer(C.byref(can), a, b)
            rc2 = hip.hipExtGetLinkTypeAndHopCount(a, b, C.byref(lt), C.byref(hops))
            pairs.append({"from": a, "to": b, "peer": bool(can.value) and rc1 == 0,
                          "link": LINK.get(lt.value, str(lt.value)) if rc2 == 0 else None, "hops": hops.value if rc2 == 0 else None})
            if rc1 != 0 or not can.value:
                bad.append((a, b))
    links = sorted({p_["link"] for p_ in pairs if p_["link"]})
    check("peer_access", not bad, device_pairs_checked=len(pairs), pairs_without_peer_access=bad, link_types=links,
          all_xgmi_one_hop=bool(pairs) and all(p_["link"] == "xgmi" and p_["hops"] == 1 for p_ in pairs),
          note="one device per rank; --one-device and N = 1 have no pairs to check")
except Exception as e:
    check("peer_access", False, error=repr(e))
rccl = None
for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
    try:
        rccl = C.CDLL(name, mode=C.RTLD_GLOBAL); break
    except OSError:
        pass
need = ["ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllGather", "ncclSend", "ncclRecv", "ncclGroupStart",
        "ncclGroupEnd", "ncclGetErrorString", "ncclCommCount", "ncclCommUserRank"]
if rccl is None:
    check("rccl", False, error="librccl.so.1 not loadable")
else:
    missing = [x for x in need if not hasattr(rccl, x)]
    _, f_r = owner(rccl, "ncclCommCount") if not missing else (None, "?")
    uid = (C.c_uint8 * 128)()
    rc = L.vsg_shard_unique_id(uid)
    check("rccl", not missing and rc == 0, library=f_r, missing=missing, vsg_shard_unique_id=rc)
W, H, nfeat = bench.WORKLOADS[workload]
cap = nfeat + 3 * 8  # >= vsg_orb_capacity for 8 levels; the handle's own figure needs a device allocation
rec = sharding.record_bytes(cap + 64)
recv = n * batch * rec
resident = batch * (W * H * 2 + int(2.6 * 1.38 * W * H) + 2 * (cap + 64) * 60 + 3 * (cap + 64) * 4)
mem = []
for d in range(min(need_dev, ndev_t)):
    free, total = torch.cuda.mem_get_info(d)
    mem.append({"device": d, "free_bytes": free, "total_bytes": total})
ranks_per_dev = n if one_device else 1
check("memory", bool(mem) and all(m["free_bytes"] > 2 * ranks_per_dev * (recv + resident) for m in mem), exchange_recv_bytes_per_rank=recv,
      resident_estimate_bytes_per_rank=resident, devices=mem)
out["ok"] = all(c["ok"] for c in out["checks"].values())
print(json.dumps(out))
sys.exit(0 if out["ok"] else 4)
""" % (str(ROOT), n, batch, workload, bool(one_device))
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=str(ROOT))
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    if lines and not quiet:
        print(lines[-1], flush=True)
    elif lines and r.returncode != 0:
        sys.stderr.write(lines[-1] + "\n")
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-2000:])
        if lines:
            bad = [k for k, v in json.loads(lines[-1])["checks"].items() if not v["ok"]]
            sys.stderr.write(f"bench.py --preflight: FAILED checks: {bad}\n")
        else:
            sys.stderr.write(f"bench.py --preflight: the probe died (exit code {r.returncode})\n")
    return r.returncode


def run_teeing_stderr(cmd, env):
    """Run `cmd`, relaying its stderr LIVE (a multi-GPU run that hangs in the RCCL rendezvous or in IPC shows its
    diagnostics while it hangs) and keeping a copy for the caller; stdout is captured (the one JSON line)."""
    import subprocess
    import threading
    import types
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    kept = []

    def pump():
        for line in p.stderr:
            sys.stderr.write(line)
            sys.stderr.flush()
            kept.append(line)
    t = threading.Thread(target=pump, daemon=True)
    t.start()
    out = p.stdout.read()
    p.wait()
    t.join()
    return types.SimpleNamespace(returncode=p.returncode, stdout=out, stderr="".join(kept[-400:]))


def launch_ranks(args, argv):
    """`python bench.py --gpus N` (N > 1, no WORLD_SIZE in the environment): start the N ranks -- one process per GPU,
    `python -m torch.distributed.run` -- as a CHILD process before this one imports torch or touches a GPU, relay rank
    0's JSON line, and fail loudly (non-zero exit, no line) if fewer than N devices are visible, a rank fails, or the line
    does not say n_gpus == N.  Never a silent one-rank line for an N-GPU request."""
    import socket
    import subprocess
    n = args.gpus
    if not args.one_device:
        have = visible_gpu_count()
        if have is not None and have < n:
            sys.stderr.write(f"bench.py: --gpus {n} but only {have} GPU(s) visible on this node\n")
            return 2
    if not args.no_preflight:
        # what an N-GPU run depends on, checked in a child BEFORE any rank starts (stdout stays the one result line): a
        # torch wheel with another libamdhip64 than libvsg_orb.so resolves, an RCCL without ncclCommCount, too little memory
        # for the exchange buffers -- each ends here with its reason instead of inside a hung or crashed rank
        rc = preflight(n, args.batch, args.workload, one_device=args.one_device, quiet=True)
        if rc != 0:
            sys.stderr.write(f"bench.py: --gpus {n}: preflight failed (exit code {rc}); no ranks started (--no-preflight skips it)\n")
            return rc
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = None
    for attempt in range(3):
        # a free port is found by bind-then-close, so another process may take it before torchrun binds it: a run
        # that dies on the rendezvous address is started again on another port
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
               "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + list(argv)
        r = run_teeing_stderr(cmd, env)
        # started again only when the RENDEZVOUS lost its port: the address error is there and no rank got as far as
        # printing anything of its own (a rank-side socket error of gloo / RCCL carries the same words and is a real failure)
        addr_in_use = any(m in r.stderr for m in ("EADDRINUSE", "Address already in use", "address already in use"))
        if r.returncode == 0 or not addr_in_use or "[bench]" in r.stderr or r.stdout.strip():
            break
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    for x in r.stdout.splitlines():
        if not x.startswith("{"):
            sys.stderr.write(x + "\n")
    if r.returncode != 0:
        sys.stderr.write(f"bench.py: the {n}-rank run failed (exit code {r.returncode}); no result line\n")
        return r.returncode
    try:
        d = json.loads(lines[-1])
    except (IndexError, ValueError):
        sys.stderr.write("bench.py: the ranks printed no JSON line\n")
        return 3
    if d.get("n_gpus") != n:
        sys.stderr.write(f"bench.py: asked for {n} GPUs, the line says n_gpus = {d.get('n_gpus')}\n")
        return 3
    print(lines[-1], flush=True)
    return 0


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

    def step(time_match=False):
        if not exchange and len(out_sets) == 2:
            # N = 1, two output sets: extract into one while the previous step's match reads the other
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
                if time_match:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(ms)
                if not side:
                    # the last frame of the previous batch into slot 0 (its extraction is earlier work of this stream)
                    o_desc[0].copy_(p_desc[B])
                    o_counts[0].copy_(p_counts[B])
                best2(o_desc[1].data_ptr(), o_desc[0].data_ptr(), o_counts[1].data_ptr(), o_counts[0].data_ptr(), B,
                      cap * 32, d_best.data_ptr(), d_second.data_ptr(), d_arg.data_ptr(), on_stream=ms.cuda_stream)
                if time_match:
                    e1.record(ms)
                    match_events.append((e0, e1))
                if side:
                    ev = torch.cuda.Event()
                    ev.record(ms)
                    ev_matched[i] = ev
            return
        if not exchange:
            # N = 1: carry the last frame of the previous batch into slot 0
            d_desc[0].copy_(d_desc[B])
            d_counts[0].copy_(d_counts[B])
        ex.extract_batch_device(next_input(), B, H * W, H, W, W, d_kps[1].data_ptr(), d_desc[1].data_ptr(),
                                d_counts[1].data_ptr(), cap, (0, 0), stream)
        if exchange:
            ev_extracted.record(tstream)
        if not args.no_match:
            if time_match:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(tstream)
            if not exchange:
                best2(d_desc[1].data_ptr(), d_desc[0].data_ptr(), d_counts[1].data_ptr(), d_counts[0].data_ptr(), B,
                      cap * 32, d_best.data_ptr(), d_second.data_ptr(), d_arg.data_ptr())
            else:
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
            if time_match:
                e1.record(tstream)
                match_events.append((e0, e1))
        if exchange:
            if args.no_match:
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
        # claims the driver cannot otherwise see, each behind a short budget, all OUTSIDE the timed region
        del d_gray, d_gray_in, d_kps, d_desc, d_best, d_second, d_arg, out_sets
        torch.cuda.empty_cache()
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
            other.append(device_rate("C4", 256, 10, local_rank))
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
            out["content_sweep"] = content_sweep_leg(local_rank, batch=B, cpu_seconds=min(1.0, max(0.3, args.cpu_seconds / 8)))
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
    print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
