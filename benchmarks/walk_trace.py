#!/usr/bin/env python3
"""Phase timeline of the layer-walking encoder launch (ablation build: make -C minivectordb_amd/csrc ABLATE=1, run with
MVDB_LIBMVDB=minivectordb_amd/lib/libmvdb_ablate.so).  Per phase: the longest body (previous release -> arrival) and the
barrier wait of the LAST arriver (its arrival -> its release: the barrier's own latency), in microseconds, medians over layers."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from minivectordb_amd import _native  # noqa: E402
from minivectordb_amd.embedding_model import GpuEncoder  # noqa: E402
from oracle.encoder import make_weights  # noqa: E402

large = len(sys.argv) > 2 and sys.argv[2] == "large"
cfg = {"model_type": "bert", "vocab_size": 30000, "hidden_size": 384, "num_hidden_layers": 12,
       "num_attention_heads": 12, "intermediate_size": 1536, "max_position_embeddings": 512,
       "type_vocab_size": 2, "layer_norm_eps": 1e-12, "hidden_act": "gelu", "pad_token_id": 0}
if large:
    cfg.update(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
w = make_weights(cfg, 1)
enc = GpuEncoder(cfg, {k: torch.from_numpy(v) for k, v in w.items()}, device=0)
rs = np.random.RandomState(0)
ids = rs.randint(5, 30000, size=(1, S)).astype(np.int32)
mask = np.ones((1, S), np.int32)
for _ in range(20):
    enc.forward(ids, mask)
lib = _native.lib()
lib.mvdb_debug_walk_trace.restype = ctypes.c_int
lib.mvdb_debug_walk_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
NS = 512
buf = np.zeros((256, NS), np.uint64)
n = lib.mvdb_debug_walk_trace(enc._h, buf.ctypes.data_as(ctypes.c_void_p), 256)
assert n > 0, n
t = buf[:n].astype(np.int64)
L = cfg["num_hidden_layers"]
t0 = t[:, NS - 2].min()
us = np.where(t > 0, (t - t0) / 100.0, np.nan)
names = ["qkv", "attn+wo", "sum+ln1", "ffn", "sum+ln2", "ffn1"]   # (ffn1: wide shapes only — there "ffn" is FFN2)
ORDER = [0, 1, 2, 5, 3, 4]
out = {"S": S, "large": large, "workgroups": int(n), "total_us": round(float(np.nanmax(us[:, NS - 1])), 2), "phases": {}}
prev_done = np.zeros(L * 5 + 1)
for k, nm in enumerate(names):
    rows = []
    for layer in range(L):
        base = (layer * 6 + k) * 3
        w0, rel, arr = us[:, base], us[:, base + 1], us[:, base + 2]
        if np.all(np.isnan(arr)):
            continue
        # body: released -> arrived (per workgroup); the phase is complete when its last producer has arrived
        body = arr - rel
        rows.append((np.nanmax(body), np.nanmedian(body), np.nanmax(arr), np.nanmin(rel), int(np.sum(~np.isnan(arr)))))
    if rows:
        v = np.array(rows)
        out["phases"][nm] = {"producers": int(v[0, 4]), "body_max_us": round(float(np.median(v[:, 0])), 2),
                             "body_median_wg_us": round(float(np.median(v[:, 1])), 2)}
# hand-off latency: last producer of a phase arrived -> first consumer of the next phase released
order = []
for layer in range(L):
    for k in ORDER:
        base = (layer * 6 + k) * 3
        if not np.all(np.isnan(us[:, base + 2])):
            order.append((layer, k))
gaps = {}
for (l0, k0), (l1, k1) in zip(order[:-1], order[1:]):
    last_arr = np.nanmax(us[:, (l0 * 6 + k0) * 3 + 2])
    first_rel = np.nanmin(us[:, (l1 * 6 + k1) * 3 + 1])
    last_rel = np.nanmax(us[:, (l1 * 6 + k1) * 3 + 1])
    gaps.setdefault(names[k0] + "->" + names[k1], []).append((first_rel - last_arr, last_rel - last_arr))
out["handoff_us"] = {k: {"first_release": round(float(np.median([g[0] for g in v])), 2),
                         "last_release": round(float(np.median([g[1] for g in v])), 2)} for k, v in gaps.items()}
clk = (t[:, NS - 3] - t[:, NS - 4]) / np.maximum(1, (t[:, NS - 1] - t[:, NS - 2])) * 100.0
out["shader_clock_mhz"] = round(float(np.median(clk)), 0)
per_layer = [np.nanmax(us[:, (l * 6 + 4) * 3 + 2]) for l in range(L)]
out["layer_us"] = round(float(np.median(np.diff(per_layer))), 2) if L > 1 else None
if os.environ.get("WALK_TRACE_PER_WG"):
    # per workgroup, layer 5: when each phase released it, relative to the last arrival of the phase before
    layer = 5
    per = {}
    seq = [(l, k) for (l, k) in order if l == layer]
    for (l1, k1) in seq:
        idx = order.index((l1, k1))
        if idx == 0:
            continue
        l0, k0 = order[idx - 1]
        last_arr = np.nanmax(us[:, (l0 * 6 + k0) * 3 + 2])
        rel = us[:, (l1 * 6 + k1) * 3 + 1] - last_arr
        wait0 = us[:, (l1 * 6 + k1) * 3 + 0] - last_arr      # when the workgroup began to wait (after issuing its weight loads)
        per[names[k1]] = [[int(w), round(float(wait0[w]), 2), round(float(rel[w]), 2)] for w in range(n) if not np.isnan(rel[w])]
    out["per_wg_layer5 [wg, began waiting, released] us after the previous phase's last arrival"] = per
print(json.dumps(out))
