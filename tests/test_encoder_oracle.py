"""CPU: the encoder oracle (float64 numpy restatement) against the committed outputs of
transformers' own model (tests/golden/encoder_golden.npz, made by make_encoder_golden.py)."""
import numpy as np
import pytest

from encoder_cases import load_cases
from oracle import encoder as E

CASES = load_cases()


@pytest.mark.parametrize("i", range(len(CASES)))
def test_restatement_matches_transformers_golden(i):
    c = CASES[i]
    if c["name"] == "e5-small-dims" and (c["S"] > 64 or c["B"] > 16):
        pytest.skip("kept for the GPU suite (slow in float64 numpy)")
    cfg = E.make_config(c["name"])
    w = E.make_weights(cfg, c["wseed"])
    ids, mask = E.make_inputs(cfg, c["B"], c["S"], c["iseed"])
    assert np.array_equal(ids, c["ids"]) and np.array_equal(mask, c["mask"])  # generators are stable
    hidden, emb = E.numpy_forward(cfg, w, ids, mask)
    np.testing.assert_allclose(emb, c["emb"], atol=2e-6, rtol=0)
    if c["hidden_valid"] is not None:
        np.testing.assert_allclose(hidden[mask.astype(bool)], c["hidden_valid"], atol=5e-5, rtol=0)
    if c["cls_emb"] is not None:  # bge-m3 dense vector: normalised CLS state
        cls = hidden[:, 0] / np.linalg.norm(hidden[:, 0], axis=1, keepdims=True)
        np.testing.assert_allclose(cls, c["cls_emb"], atol=2e-6, rtol=0)


def test_batched_row_equals_single_sentence():
    """BASELINE config 5 batches 256 sentences; the reference only ever runs B = 1.  A row of a
    right-padded batch must equal the B = 1 forward of that sentence."""
    cfg = E.make_config("tiny")
    w = E.make_weights(cfg, 7)
    ids, mask = E.make_inputs(cfg, 4, 12, 11)
    _, emb = E.numpy_forward(cfg, w, ids, mask)
    for b in range(4):
        n = int(mask[b].sum())
        _, e1 = E.numpy_forward(cfg, w, ids[b:b + 1, :n], mask[b:b + 1, :n])
        np.testing.assert_allclose(emb[b], e1[0], atol=1e-12)
