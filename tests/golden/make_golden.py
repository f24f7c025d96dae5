#!/usr/bin/env python3
"""Generate tests/golden/orb_golden_v1.npz from the CPU oracle.

The reference has no golden vectors for this path and cannot be built here
(SURVEY.md 8c), so these fixtures are produced by the build's own CPU oracle
on seeded synthetic frames.  They pin the oracle against regressions and
against platform drift (libm cosf/sinf, libstdc++ std::sort), and give the GPU
tests byte-exact expected outputs that do not need the oracle at run time.

Usage: python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib as ol  # noqa: E402
from visual_sgraphs_amd import synth  # noqa: E402

CASES = [
    # name, w, h, seed, amplitude_div, nfeatures, nlevels, lapping
    ("qvga_rgbd", 320, 240, 42, 1, 500, 4, (0, 0)),
    ("qvga_lowcontrast", 320, 240, 43, 8, 500, 4, (0, 0)),
    ("qvga_mono_lapping", 320, 240, 44, 1, 500, 4, (0, 1000)),
    ("qvga_partial_lapping", 320, 240, 42, 1, 300, 3, (100, 220)),
    ("vga_c2", 640, 480, 7, 1, 1000, 8, (0, 0)),
]


def main():
    out = {}
    for name, w, h, seed, div, nf, nl, lap in CASES:
        img = synth.frame(w, h, seed, amplitude_div=div)
        e = ol.OracleExtractor(nf, 1.2, nl, 20, 7)
        mono, kps, desc = e(img, lap)
        out[name + "/params"] = np.array([w, h, seed, div, nf, nl, lap[0], lap[1]], np.int64)
        out[name + "/mono"] = np.array([mono], np.int64)
        out[name + "/kps"] = kps
        out[name + "/desc"] = desc
        print(name, "n =", len(kps), "mono =", mono)
    # matcher fixture: consecutive frames of one sequence
    e = ol.OracleExtractor(500, 1.2, 4, 20, 7)
    _, k0, d0 = e(synth.sequence_frame(320, 240, 9, 0))
    _, k1, d1 = e(synth.sequence_frame(320, 240, 9, 1))
    best, second, arg = ol.block_best2(d0, d1)
    out["match/desc0"], out["match/desc1"] = d0, d1
    out["match/best"], out["match/second"], out["match/arg"] = best, second, arg
    np.savez_compressed(Path(__file__).parent / "orb_golden_v1.npz", **out)


if __name__ == "__main__":
    main()
