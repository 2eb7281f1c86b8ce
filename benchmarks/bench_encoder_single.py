#!/usr/bin/env python3
"""ONE sentence per call — the shape of the reference's own API (`extract_embeddings(text)` tokenises `[text]`:
minivectordb/embedding_model.py:62-71): host ids in, host embedding out, through GpuEncoder.forward (H2D copy of the ids,
the forward as one replayed hipGraph, D2H copy of 384 floats, one stream wait).  Per sequence length: the FIRST call (graph
capture + instantiation for that shape) and the p50 / p99 of the calls after it; then a stream of sentences of random
lengths.  One JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from minivectordb_amd.embedding_model import GpuEncoder  # noqa: E402
from oracle.encoder import make_weights  # noqa: E402

cfg = {"model_type": "bert", "vocab_size": 30000, "hidden_size": 384, "num_hidden_layers": 12,
       "num_attention_heads": 12, "intermediate_size": 1536, "max_position_embeddings": 512,
       "type_vocab_size": 2, "layer_norm_eps": 1e-12, "hidden_act": "gelu", "pad_token_id": 0}
LARGE = "--large" in sys.argv   # the e5-large / bge-m3 shape (XLM-R large widths): H 1024, 24 layers, 16 heads, FFN 4096
if LARGE:
    cfg.update(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)
w = make_weights(cfg, 1)
enc = GpuEncoder(cfg, {k: torch.from_numpy(v) for k, v in w.items()}, device=0)
rs = np.random.RandomState(0)
enc.forward(rs.randint(5, 30000, size=(1, 7)).astype(np.int32), np.ones((1, 7), np.int32))   # weights' fp16 images, workspace
out = {"what": ("e5-large / bge-m3-shaped" if LARGE else "e5-small-shaped") + " encoder, ONE sentence per call, host in / host out "
       "(GpuEncoder.forward), default mode", "walks": {str(S): bool(enc.walks(1, S)) for S in (8, 64, 128)}, "per_length": []}
for S in (8, 16, 24, 32, 64, 128, 256):
    ids = rs.randint(5, 30000, size=(1, S)).astype(np.int32)
    mask = np.ones((1, S), np.int32)
    t0 = time.perf_counter()
    enc.forward(ids, mask)
    first = time.perf_counter() - t0
    lat = []
    for _ in range(300):
        t0 = time.perf_counter()
        enc.forward(ids, mask)
        lat.append(time.perf_counter() - t0)
    out["per_length"].append({"S": S, "first_call_ms": round(first * 1e3, 3), "p50_ms": round(float(np.median(lat)) * 1e3, 4),
                              "p99_ms": round(float(np.percentile(lat, 99)) * 1e3, 4)})
# a stream of sentences of random lengths 4..60 tokens (every new length captures once)
lens = rs.randint(4, 61, size=2000)
lat = []
for n in lens:
    ids = rs.randint(5, 30000, size=(1, int(n))).astype(np.int32)
    mask = np.ones((1, int(n)), np.int32)
    t0 = time.perf_counter()
    enc.forward(ids, mask)
    lat.append(time.perf_counter() - t0)
lat = np.array(lat)
out["random_lengths_4_60"] = {"calls": len(lat), "mean_ms": round(float(lat.mean()) * 1e3, 4), "p50_ms": round(float(np.median(lat)) * 1e3, 4),
                              "p99_ms": round(float(np.percentile(lat, 99)) * 1e3, 4), "max_ms": round(float(lat.max()) * 1e3, 3),
                              "first_200_mean_ms": round(float(lat[:200].mean()) * 1e3, 4),
                              "last_1000_mean_ms": round(float(lat[1000:].mean()) * 1e3, 4)}
print(json.dumps(out))
