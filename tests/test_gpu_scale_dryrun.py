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
