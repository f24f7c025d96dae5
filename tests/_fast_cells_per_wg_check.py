"""Child process of tests/test_gpu_fast_multicell.py: VSG_FAST_K (cells per FAST workgroup) is read once per process, so
every value gets a process of its own.  Extracts a mixed batch -- corner-rich frames, the contrast ladder, a
low-contrast frame (cells that retry at minThFAST), a constant frame (empty cells), noise (pixels that pass the
necessary test on both sides: the queue's overflow path at low thresholds) -- and compares every frame, and the FAST
candidate multiset of a few of them, with the CPU oracle.  Prints OK and the number of frames checked."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import oracle_lib as ol  # noqa: E402
from test_gpu_extract import _contrast_ladder, assert_same_output  # noqa: E402
from visual_sgraphs_amd import orb, synth  # noqa: E402


def frames(w, h):
    rng = np.random.default_rng(11)
    out = [synth.sequence_frame(w, h, 5, t) for t in range(3)]
    out += [_contrast_ladder(w, h, s) for s in (3, 4)]
    out.append(synth.frame(w, h, 7, amplitude_div=8))
    out.append(np.full((h, w), 77, np.uint8))
    out.append(rng.integers(0, 256, (h, w), dtype=np.uint8))
    out.append(np.clip(128 + rng.integers(-12, 13, (h, w)), 0, 255).astype(np.uint8))
    # content classes that fill the queue (one-sided: checkerboard; two-sided: the steep ramp), that leave whole levels to
    # the minThFAST pass (value noise) and that scatter isolated corners (salt and pepper)
    out += [synth.content_frame(kind, w, h, 13, 2) for kind in ("checker1", "sawtooth", "value_noise", "salt_pepper")]
    return np.stack(out)


def main():
    checked = 0
    for (w, h, nf, nl, ini, mn) in ((640, 480, 1000, 8, 20, 7), (640, 480, 1500, 4, 2, 1), (752, 480, 1200, 8, 9, 8),
                                    (416, 300, 700, 5, 20, 7),
                                    # 40 x 69-pixel cells: their tiles (75 rows of 4 chunks) exceed the two prefetch chunks
                                    # per thread of the narrow tile class, so these cells are staged when their turn comes
                                    (151, 101, 200, 2, 20, 7),
                                    # one column of 69-pixel-wide cells: the widest tile class (84-byte rows, 4 chunks per
                                    # thread); and 45 x 51 cells: the middle one (68-byte rows)
                                    (101, 151, 200, 2, 20, 7), (165, 133, 300, 2, 20, 7)):
        imgs = frames(w, h)
        ex = orb.ORBextractor(nf, 1.2, nl, ini, mn, max_batch=len(imgs))
        ref = ol.OracleExtractor(nf, 1.2, nl, ini, mn)
        outs = ex.extract_batch(imgs)
        for t in range(len(imgs)):
            assert_same_output(outs[t], ref(imgs[t]), f"{w}x{h} thresholds {ini}/{mn} frame {t}")
            checked += 1
        # the candidate multiset of every level of the LAST frame of the batch (what the handle's stage read-back sees)
        ref(imgs[-1])
        for l in range(nl):
            gx, gy, gr = ex.candidates(l, frame=len(imgs) - 1)
            ox, oy, orr = ref.candidates(l)
            assert sorted(zip(gx.tolist(), gy.tolist(), gr.tolist())) == sorted(zip(ox.tolist(), oy.tolist(), orr.tolist())), (w, h, l)
    print("OK", checked)


if __name__ == "__main__":
    main()
