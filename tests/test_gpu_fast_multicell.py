"""k_fast_cells with several cells per workgroup (the next cell's tile in flight while the current one is worked on):
the launcher only takes that form for launches of tens of thousands of cells, which no other parity test reaches, so it
is forced here through VSG_FAST_K -- one child process per value, because the library reads the variable once."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
CHILD = Path(__file__).parent / "_fast_cells_per_wg_check.py"


@pytest.mark.parametrize("k", ["1", "2", "3", "5", "16"])
def test_every_cells_per_workgroup_count_gives_the_oracles_output(k):
    env = dict(os.environ, VSG_FAST_K=k)
    r = subprocess.run([sys.executable, str(CHILD)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.strip().splitlines()[-1].startswith("OK 91"), r.stdout[-500:]
