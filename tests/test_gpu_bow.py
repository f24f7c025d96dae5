"""GPU parity of the DBoW2 vocabulary transform and the C3 chain
extract -> ComputeBoW -> SearchByBoW with a synthetic vocabulary (the real ORBvoc.txt.bin is a missing blob)."""
import numpy as np
import pytest

import oracle_lib as ol
from visual_sgraphs_amd import orb, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def frames():
    e = ol.OracleExtractor(1200, 1.2, 8, 20, 7)
    _, k0, d0 = e(synth.sequence_frame(752, 480, 13, 0))
    _, k1, d1 = e(synth.sequence_frame(752, 480, 13, 1))
    return (k0, d0), (k1, d1)


@pytest.mark.parametrize("k,L,levelsup,scoring,weighting", [(10, 3, 2, 0, 0), (10, 4, 4, 0, 0), (8, 3, 1, 1, 1),
                                                            (10, 3, 5, 5, 0), (6, 4, 2, 2, 2), (10, 3, 2, 0, 3)])
def test_bow_transform_matches_dbow2_restatement(frames, k, L, levelsup, scoring, weighting):
    (k0, d0), _ = frames
    blob = synth.synthetic_vocabulary(k, L, seed=k * 10 + L, scoring=scoring, weighting=weighting)
    ref, voc = ol.OracleVocabulary(blob), orb.ORBVocabulary(blob)
    assert (voc.k, voc.L, voc.scoring, voc.weighting, voc.nnodes, voc.nwords) == \
        (ref.k, ref.L, ref.scoring, ref.weighting, ref.nnodes, ref.nwords)
    want, got = ref.transform(d0, levelsup), voc.transform(d0, levelsup)
    for key in ("word", "node", "bow_ids"):
        assert np.array_equal(got[key], want[key]), key
    assert np.array_equal(got["weight"].view(np.uint64), want["weight"].view(np.uint64))
    assert np.array_equal(got["bow_vals"].view(np.uint64), want["bow_vals"].view(np.uint64))  # bit-identical doubles
    for a, b in zip(got["fv"], want["fv"]):
        assert np.array_equal(a, b)
    assert len(want["bow_ids"]) > 20


def test_bow_transform_ties_take_first_child():
    """All-equal node descriptors: every distance ties, the first child must win at every level."""
    import struct
    k, L = 4, 3
    n_nodes = (k ** (L + 1) - 1) // (k - 1)
    first_leaf = (k ** L - 1) // (k - 1)
    blob = bytearray(struct.pack("<iiii", k, L, 0, 0))
    for i in range(1, n_nodes):
        blob += struct.pack("<iB", (i - 1) // k, 1 if i >= first_leaf else 0) + bytes(32) + struct.pack("<d", 1.0)
    desc = synth.random_descriptors(50, 3)
    ref, voc = ol.OracleVocabulary(bytes(blob)), orb.ORBVocabulary(bytes(blob))
    want, got = ref.transform(desc, 1), voc.transform(desc, 1)
    assert np.array_equal(got["word"], want["word"]) and np.all(want["word"] == 0)
    assert np.array_equal(got["node"], want["node"])


def test_c3_chain_extract_bow_search(frames):
    """Config C3: extraction (GPU) -> ComputeBoW (GPU) -> SearchByBoW(KF, F) (GPU) == the same chain on the oracle."""
    (k0, d0), (k1, d1) = frames
    ex = orb.ORBextractor(1200, 1.2, 8, 20, 7, max_batch=2)
    outs = ex.extract_batch(np.stack([synth.sequence_frame(752, 480, 13, 0), synth.sequence_frame(752, 480, 13, 1)]))
    assert outs[0][1].tobytes() == k0.tobytes() and np.array_equal(outs[1][2], d1)
    blob = synth.synthetic_vocabulary(10, 4, seed=5)
    ref, voc = ol.OracleVocabulary(blob), orb.ORBVocabulary(blob)
    fv_kf, fv_f = voc.transform(outs[0][2], 2)["fv"], voc.transform(outs[1][2], 2)["fv"]
    rfv_kf, rfv_f = ref.transform(d0, 2)["fv"], ref.transform(d1, 2)["fv"]
    for a, b in zip(fv_kf + fv_f, rfv_kf + rfv_f):
        assert np.array_equal(a, b)
    valid = np.ones(len(d0), np.uint8)
    m = orb.ORBmatcher(0.7, True)
    n_got, got = m.SearchByBoW_KF_F(outs[0][2], outs[0][1]["angle"], valid, fv_kf, outs[1][2], outs[1][1]["angle"], fv_f)
    n_want, want = ol.search_by_bow_kf_f(d0, k0["angle"], valid, rfv_kf, d1, k1["angle"], rfv_f, 0.7, True)
    assert n_got == n_want and np.array_equal(got, want)
    assert n_want > 100
