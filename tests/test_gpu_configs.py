"""GPU tests of the BASELINE.json configurations as stated (C3: stereo + ComputeStereoMatches + ComputeBoW on a
k = 10, L = 6 vocabulary + SearchByBoW; C5: four concurrent 1250-feature camera streams with the per-frame tracking
searches), run from plain C++ through the C ABI by tools/config_chain.cpp, which bit-compares every output of both
chains with the CPU oracle (pairs / frames of whole short sequences) before it times anything."""
import json
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_c3_and_c5_chains_equal_the_oracle():
    exe = ROOT / "tools" / "_bin" / "config_chain"
    if not exe.exists():
        subprocess.check_call(["make", "-C", str(ROOT / "tools")])
    r = subprocess.run([str(exe), "0.3", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-500:]
    doc = json.loads(r.stdout.strip().splitlines()[-1])
    c3, c5 = doc["C3"], doc["C5"]
    assert c3["parity"] is True and c3["pairs_checked"] == 8
    # the shape the reference's SearchByBoW sees: ~100 level-2 nodes, hundreds of matches per pair
    assert 60 <= c3["per_pair"]["feature_vector_nodes"] <= 100 and c3["per_pair"]["bow_matches"] > 100
    assert c3["per_pair"]["stereo_matches"] > 300
    assert c5["parity"] is True and c5["frames_checked"] == 24 and c5["streams"] == 4
    assert c5["per_frame"]["matches_last_frame"] > 200 and c5["per_frame"]["matches_local_map"] > 100
