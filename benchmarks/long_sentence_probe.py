#!/usr/bin/env python3
"""ONE long sentence per call (129 .. 512 tokens — extract_embeddings truncates at 512, minivectordb/embedding_model.py:64,77):
host ids in, host embedding out, p50 / p99 per length for the forms selected by the environment switches read at encoder
creation.  usage: long_sentence_probe.py [--large] [--lengths 128,256,512] [--variants default,nowalk,lnfused] [--calls N]
One JSON line per (variant, length)."""
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


def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


LARGE = "--large" in sys.argv
lengths = [int(v) for v in arg("--lengths", "64,128,192,256,384,512").split(",")]
variants = arg("--variants", "default,nowalk,lnfused").split(",")
calls = int(arg("--calls", "300"))
ENV = {"default": {}, "nowalk": {"MVDB_ENCODER_WALK": "0"}, "lnfused": {"MVDB_ENCODER_WALK": "0", "MVDB_GEMM_LN_FUSED": "2"}}
for v in variants:
    if v not in ENV:   # KEY=VAL+KEY=VAL spelled on the command line
        ENV[v] = dict(kv.split("=") for kv in v.split("+"))

cfg = {"model_type": "bert", "vocab_size": 30000, "hidden_size": 384, "num_hidden_layers": 12,
       "num_attention_heads": 12, "intermediate_size": 1536, "max_position_embeddings": 512,
       "type_vocab_size": 2, "layer_norm_eps": 1e-12, "hidden_act": "gelu", "pad_token_id": 0}
if LARGE:
    cfg.update(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)
w = {k: torch.from_numpy(v) for k, v in make_weights(cfg, 1).items()}
rs = np.random.RandomState(0)
inputs = {S: rs.randint(5, 30000, size=(1, S)).astype(np.int32) for S in lengths}
ref = {}
for name in variants:
    keep = {k: os.environ.get(k) for k in ENV[name]}
    os.environ.update(ENV[name])
    enc = GpuEncoder(cfg, w, device=0)
    for k, v in keep.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    for S in lengths:
        ids, mask = inputs[S], np.ones((1, S), np.int32)
        t0 = time.perf_counter()
        got = enc.forward(ids, mask)
        first = time.perf_counter() - t0
        lat = []
        for _ in range(calls):
            t0 = time.perf_counter()
            enc.forward(ids, mask)
            lat.append(time.perf_counter() - t0)
        d = None
        if S in ref:
            d = float(np.abs(got - ref[S]).max())
        else:
            ref[S] = got
        print(json.dumps({"shape": "large" if LARGE else "e5-small", "variant": name, "S": S, "walks": bool(enc.walks(1, S)),
                          "first_call_ms": round(first * 1e3, 3), "p50_ms": round(float(np.median(lat)) * 1e3, 4),
                          "p99_ms": round(float(np.percentile(lat, 99)) * 1e3, 4), "max_abs_diff_vs_first_variant": d}), flush=True)
    enc.close()
