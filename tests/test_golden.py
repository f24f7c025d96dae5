"""The oracle must reproduce the committed golden fixtures byte for byte (tests/golden/make_golden.py)."""
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as ol
from visual_sgraphs_amd import synth

GOLDEN = np.load(Path(__file__).parent / "golden" / "orb_golden_v1.npz")
CASES = sorted({k.split("/")[0] for k in GOLDEN.files if k.endswith("/params")})


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(name):
    w, h, seed, div, nf, nl, lap0, lap1 = GOLDEN[name + "/params"].tolist()
    e = ol.OracleExtractor(nf, 1.2, nl, 20, 7)
    mono, kps, desc = e(synth.frame(w, h, seed, amplitude_div=div), (lap0, lap1))
    assert mono == int(GOLDEN[name + "/mono"][0])
    assert kps.tobytes() == GOLDEN[name + "/kps"].tobytes()
    assert np.array_equal(desc, GOLDEN[name + "/desc"])


def test_oracle_reproduces_golden_match():
    best, second, arg = ol.block_best2(GOLDEN["match/desc0"], GOLDEN["match/desc1"])
    assert np.array_equal(best, GOLDEN["match/best"])
    assert np.array_equal(second, GOLDEN["match/second"])
    assert np.array_equal(arg, GOLDEN["match/arg"])
    assert (best <= 50).sum() > 100  # translated scene: many true matches
