"""What bench.py and its outside-the-timed-region legs (tools/bench_extras.py) share: the workloads, the algorithmic byte
counts of SURVEY.md 8d, the committed PMC traffic figures, the every-frame parity checker and the CPU baseline leg."""
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests"), str(ROOT / "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

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
