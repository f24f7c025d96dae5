"""bench.py's --gpus handling, the parts that need no GPU: a request for more ranks than visible devices and a
launcher / flag mismatch both end non-zero without a result line."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ROOT = Path(__file__).resolve().parent.parent


def _run(args, **env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, timeout=300,
                          env=env, cwd=str(ROOT))


def test_more_ranks_than_devices_is_refused_before_anything_starts():
    r = _run(["--gpus", "4"], HIP_VISIBLE_DEVICES="0,1")
    assert r.returncode == 2 and "only 2 GPU" in r.stderr and "{" not in r.stdout


def test_world_size_and_gpus_flag_must_agree():
    r = _run(["--gpus", "4"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and "{" not in r.stdout
    r = _run(["--gpus", "1"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "{" not in r.stdout


def test_visible_gpu_count_reads_the_visibility_variables(monkeypatch):
    sys.path.insert(0, str(ROOT))
    import bench
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    assert bench.visible_gpu_count() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpu_count() == 0


def test_preflight_reports_every_check_and_fails_without_devices():
    """`bench.py --gpus N --preflight` on a box without a GPU: the probe runs in a child, prints its findings, names the
    failed checks on stderr and exits non-zero; the checks that need no device (one HIP runtime for torch and
    libvsg_orb.so, RCCL's exports) are answered all the same."""
    import json
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU")
    r = _run(["--gpus", "8", "--preflight"])
    assert r.returncode != 0 and "FAILED checks" in r.stderr and "devices" in r.stderr
    d = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert d["ok"] is False and d["requested_gpus"] == 8
    assert set(d["checks"]) == {"one_hip_runtime", "devices", "peer_access", "rccl", "memory"}
    # no device here: no pair to check, the check itself must have run (hipDeviceCanAccessPeer / link type bound)
    assert d["checks"]["peer_access"].get("device_pairs_checked") == 0 and "error" not in d["checks"]["peer_access"]
    assert d["checks"]["one_hip_runtime"]["ok"] is True
    assert d["checks"]["one_hip_runtime"]["libvsg_orb"] == d["checks"]["one_hip_runtime"]["torch"]
    assert d["checks"]["rccl"]["ok"] is True and d["checks"]["rccl"]["missing"] == []
    assert d["checks"]["memory"]["exchange_recv_bytes_per_rank"] > 200e6  # 8 x 512 records of ~64 KB


def test_a_failed_library_communicator_is_a_refusal_not_a_fallback(capsys):
    """with --gpus N > 1, the nccl backend and no --torch-gather, a failed vsg_shard_create on any rank ends
    the run non-zero with the reason; torch.distributed carries the exchange only in the rehearsals."""
    import bench
    assert bench.wants_library_exchange("nccl", False, False)
    assert not bench.wants_library_exchange("gloo", False, False)       # CPU rehearsal
    assert not bench.wants_library_exchange("nccl", True, False)        # --one-device dry run
    assert not bench.wants_library_exchange("nccl", False, True)        # --torch-gather: measured on purpose
    with pytest.raises(SystemExit) as e:
        bench.refuse_without_library_exchange(3, 8, "vsg_shard_create: ncclCommInitRank: unhandled system error")
    assert e.value.code == 3
    err = capsys.readouterr().err
    assert "rank 3/8" in err and "ncclCommInitRank" in err and "--torch-gather" in err
    with pytest.raises(SystemExit) as e:
        bench.refuse_without_library_exchange(0, 8, "")                 # a rank whose own create succeeded leaves too
    assert e.value.code == 3


def test_the_line_needs_a_communicator_that_spans_the_launched_ranks(capsys):
    import bench
    assert bench.check_rccl_world(8, 8)
    with pytest.raises(SystemExit) as e:
        bench.check_rccl_world(4, 8, rank=2)
    assert e.value.code == 4 and "spans 4 ranks, launched 8" in capsys.readouterr().err


def test_the_fallback_is_gone_from_the_source():
    """No path from a failed ShardComm to torch.distributed on the nccl backend is left in bench.py."""
    src = (ROOT / "bench.py").read_text()
    assert "using torch.distributed for the exchange" not in src
    i = src.index("comm = sharding.ShardComm(")
    assert "refuse_without_library_exchange(rank, world, why)" in src[i:i + 1500]
