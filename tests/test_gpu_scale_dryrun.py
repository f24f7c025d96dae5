"""The multi-rank path of bench.py on the one GPU of the test box (tools/scale_dryrun.sh): two ranks on device 0, the
record exchange over gloo, both exchange forms (all-gather of every record / neighbour-only boundary record), the
cross-rank boundary match checked against the oracle by bench.py's own parity gate.  No N > 1 hardware run exists
yet; this keeps the code path that run will take exercised every round."""
import json
import os
import random
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("exchange", ["allgather", "boundary"])
def test_two_ranks_on_one_device(exchange):
    port = 29500 + random.randrange(2000)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--one-device",
           "--dist-backend", "gloo", "--exchange", exchange, "--batch", "16", "--steps", "3", "--warmup", "3",
           "--cpu-seconds", "1", "--no-stage-timing"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-800:]
    line = [x for x in r.stdout.strip().splitlines() if x.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["parity"]["bit_exact_vs_oracle"] is True          # incl. the boundary match against the remote frame
    assert d["config"]["exchange"] == exchange and d["config"]["dist_world_size"] == 2
    assert d["cpu_baseline"] and d["cpu_baseline"]["value"] > 0   # N > 1 lines carry the CPU baseline too


def test_four_ranks_on_one_device():
    """World 4 on the one GPU (VERDICT r4 #6b): `python bench.py --gpus 4 --one-device` starts four ranks itself; the
    all-gather is 4 wide, rank 0's boundary predecessor is rank 3 of the previous step."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "4", "--one-device", "--dist-backend", "gloo", "--batch", "8",
           "--steps", "3", "--warmup", "2", "--ramp-steps", "2", "--cpu-seconds", "1", "--no-stage-timing"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-1500:]
    d = json.loads([x for x in r.stdout.strip().splitlines() if x.startswith("{")][-1])
    assert d["n_gpus"] == 4 and d["config"]["dist_world_size"] == 4
    assert d["parity"]["bit_exact_vs_oracle"] is True and d["parity"]["boundary_match_row"] is True
    assert d["parity"]["frames_checked"] == 8 and d["parity"]["match_rows_checked"] == 7


def test_preflight_passes_on_this_box_for_one_gpu():
    """`bench.py --gpus 1 --preflight`: every check answered and passed on the test box (one HIP runtime shared by torch
    and libvsg_orb.so, RCCL exports incl. ncclCommCount, memory for the exchange buffers)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--preflight"], capture_output=True, text=True,
                       timeout=300, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-1500:] + r.stdout[-1500:]
    d = json.loads([x for x in r.stdout.strip().splitlines() if x.startswith("{")][-1])
    assert d["ok"] is True and all(c["ok"] for c in d["checks"].values())
    r8 = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--preflight"], capture_output=True,
                        text=True, timeout=300, env=env, cwd=str(ROOT))
    assert r8.returncode != 0 and "devices" in r8.stderr  # one GPU here: the 8-GPU preflight says so


def test_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` WITHOUT an outer torchrun: bench.py starts the two ranks as a child process before it
    touches the GPU and relays rank 0's line (--gpus used to be parsed and never read)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--one-device", "--dist-backend", "gloo", "--exchange",
           "boundary", "--batch", "16", "--steps", "3", "--warmup", "2", "--ramp-steps", "2", "--cpu-seconds", "1",
           "--no-stage-timing"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-1500:]
    lines = [x for x in r.stdout.strip().splitlines() if x.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["dist_world_size"] == 2 and d["parity"]["bit_exact_vs_oracle"] is True


def test_gpus_flag_refuses_more_ranks_than_devices():
    """One GPU on the test box: --gpus 2 (without --one-device) must fail loudly, not print a one-rank line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = "0"
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=str(ROOT))
    assert r.returncode != 0 and not [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert "only 1 GPU" in r.stderr
