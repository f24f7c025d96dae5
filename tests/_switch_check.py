"""Child process of tests/test_gpu_switches.py: the library reads its remaining environment switches once per process
(VSG_NO_OVERLAP, VSG_GRAPH, VSG_ROCTX -- VSG_FAST_K and VSG_SUBBATCH have tests of their own), so every switch gets a
process of its own.  Runs the blocking one-frame operator() twelve times (the hipGraph of VSG_GRAPH is recorded on a
pipeline slot's second call and replayed from its third) and a 70-frame throughput batch with stage read-back,
everything against the CPU oracle.  Prints OK, the frames checked and the graph launches seen."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import oracle_lib as ol  # noqa: E402
from test_gpu_extract import assert_same_output  # noqa: E402
from visual_sgraphs_amd import orb, synth  # noqa: E402


def main():
    checked = 0
    W, H, NF = 640, 480, 1000
    ref = ol.OracleExtractor(NF, 1.2, 8, 20, 7)
    ex = orb.ORBextractor(NF, 1.2, 8, 20, 7, max_batch=70)
    # one frame per blocking call (the latency path): 12 calls = 4 per pipeline slot; under VSG_GRAPH a slot's chain is
    # recorded on its second call and replayed from its third on
    for t in range(12):
        img = synth.sequence_frame(W, H, 3, t)
        assert_same_output(ex(img), ref(img), f"single frame call {t}")
        checked += 1
    launches = ex.chain_graph_launches()
    # a throughput batch
    imgs = np.stack([synth.sequence_frame(W, H, 5, t % 35) for t in range(70)])
    outs = ex.extract_batch(imgs)
    for t in (0, 1, 34, 35, 68, 69):
        assert_same_output(outs[t], ref(imgs[t]), f"batch frame {t}")
        checked += 1
    for l in (0, 3, 7):
        ref(imgs[69])
        assert np.array_equal(ex.image_pyramid(l, frame=69), ref.pyramid_level(l)), l
        assert np.array_equal(ex.blurred_level(l, frame=69), ref.blurred_level(l)), l
    print("OK", checked, launches)


if __name__ == "__main__":
    main()
