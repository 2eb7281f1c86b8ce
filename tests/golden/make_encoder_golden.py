#!/usr/bin/env python3
"""Generate tests/golden/encoder_golden.npz: outputs of transformers' own BertModel /
XLMRobertaModel (CPU, fp32) + the reference's average_pool / F.normalize on seeded weights and
token ids (oracle/encoder.py).  Weights are NOT stored: both sides regenerate them from the seed.
Run in the build container; the test-suite only reads the committed .npz."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import encoder as E  # noqa: E402

CASES = [  # (config, weight seed, B, S, input seed)
    ("tiny", 7, 3, 9, 3),
    ("tiny", 7, 1, 5, 4),          # the reference's own shape: one un-padded sentence
    ("hd64", 8, 2, 17, 5),
    ("xlmr-tiny", 9, 3, 11, 6),
    ("e5-small-dims", 10, 4, 32, 7),
    ("e5-small-dims", 10, 2, 130, 8),  # more than one attention key tile / query tile
    ("xlmr-large-dims", 12, 3, 70, 9),  # H = 1024 row kernels, head_dim 64 attention, XLM-R positions
    ("e5-small-dims", 10, 2, 512, 13),  # the reference's truncation cap (max_length = 512)
    ("e5-small-dims", 10, 256, 32, 31),  # BASELINE config 5's batch: 256 ragged sentences (embeddings only)
    ("xlmr-large-dims", 12, 4, 33, 41),  # bge-m3: CLS state of XLMRobertaModel, right-padded batch
    ("h96", 14, 5, 23, 51),   # H = 96, FFN 160: multiples of 32 but not of 64 (partial lane slots, partial column tiles)
    ("h96", 14, 70, 13, 52),  # the same widths over several 64-row bands
]
EMB_ONLY = {8}  # cases whose hidden states are not stored (size)


def main():
    out = {}
    for i, (name, wseed, B, S, iseed) in enumerate(CASES):
        cfg = E.make_config(name)
        w = E.make_weights(cfg, wseed)
        ids, mask = E.make_inputs(cfg, B, S, iseed)
        hidden, emb = E.hf_forward(cfg, w, ids, mask)
        h64, e64 = E.numpy_forward(cfg, w, ids, mask)
        m = mask.astype(bool)
        print(name, B, S, "hf vs float64 restatement:", np.abs(hidden[m] - h64[m]).max(), np.abs(emb - e64).max())
        out[f"case{i}_meta"] = np.array([name, str(wseed), str(B), str(S), str(iseed)])
        out[f"case{i}_ids"] = ids
        out[f"case{i}_mask"] = mask
        out[f"case{i}_emb"] = emb.astype(np.float32)
        if i not in EMB_ONLY:
            out[f"case{i}_hidden_valid"] = hidden[m].astype(np.float32)
        if cfg["model_type"] != "bert":
            # FlagEmbedding's BGEM3FlagModel.encode(...)['dense_vecs'] (minivectordb/embedding_model.py:74-78):
            # last_hidden_state[:, 0] of the XLM-R encoder, L2-normalised
            cls = hidden[:, 0]
            out[f"case{i}_cls_emb"] = (cls / np.linalg.norm(cls, axis=1, keepdims=True)).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "encoder_golden.npz"), **out)
    print("wrote encoder_golden.npz", os.path.getsize(os.path.join(HERE, "encoder_golden.npz")))


if __name__ == "__main__":
    main()
