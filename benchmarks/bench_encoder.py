#!/usr/bin/env python3
"""BASELINE config 5, encoder half: multilingual-e5-small-shaped encoder forward (random-init
weights, synthetic token ids), batch 256, S in {32, 512}.  Reports sentences/s and, per arithmetic mode, the
matrix-core rate against THAT mode's peak (/opt/skills/guides/MI355X_MICROARCH.md): the exact mode issues fp32 MFMAs
(157.3 TFLOP/s); the split-precision mode issues THREE fp16 products per fp32 product on the 16-bit cores (2,500 TFLOP/s
dense) — its fraction is 3 x the algorithmic FLOPs / time / 2.5e15, never a fraction of the fp32 peak.
MVDB_BENCH_S="32,512", MVDB_BENCH_COMPUTE="0,2", MVDB_BENCH_REPS=10, MVDB_BENCH_MODEL=e5-small|e5-large."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from minivectordb_amd.embedding_model import GpuEncoder  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(dev))  # a real stream: the encoder replays hipGraphs on it
    cfg = {"model_type": "bert", "vocab_size": 250037, "hidden_size": 384, "num_hidden_layers": 12,
           "num_attention_heads": 12, "intermediate_size": 1536, "max_position_embeddings": 512,
           "type_vocab_size": 2, "layer_norm_eps": 1e-12, "hidden_act": "gelu", "pad_token_id": 0}
    model = os.environ.get("MVDB_BENCH_MODEL", "e5-small")
    if model == "e5-large":  # multilingual-e5-large / bge-m3 (XLM-R large): the reference's AlternativeModel.large / .bgem3
        cfg.update({"model_type": "xlm-roberta", "vocab_size": 250002, "hidden_size": 1024, "num_hidden_layers": 24,
                    "num_attention_heads": 16, "intermediate_size": 4096, "max_position_embeddings": 514,
                    "type_vocab_size": 1, "layer_norm_eps": 1e-5, "pad_token_id": 1})
    from oracle.encoder import weight_names
    g = torch.Generator(device="cpu").manual_seed(0)
    H, F, L = cfg["hidden_size"], cfg["intermediate_size"], cfg["num_hidden_layers"]
    sd = {}
    for name in weight_names(cfg):
        if name == "embeddings.word_embeddings.weight":
            shape = (cfg["vocab_size"], H)
        elif name == "embeddings.position_embeddings.weight":
            shape = (cfg["max_position_embeddings"], H)
        elif name == "embeddings.token_type_embeddings.weight":
            shape = (cfg["type_vocab_size"], H)
        elif name.endswith("intermediate.dense.weight"):
            shape = (F, H)
        elif name.endswith("intermediate.dense.bias"):
            shape = (F,)
        elif name.endswith("output.dense.weight") and "attention" not in name:
            shape = (H, F)
        elif name.endswith(".weight") and "LayerNorm" not in name:
            shape = (H, H)
        else:
            shape = (H,)
        t = torch.randn(shape, generator=g) * 0.05
        if "LayerNorm.weight" in name:
            t = t + 1.0
        sd[name] = t
    enc = GpuEncoder(cfg, sd, device=0)
    B = int(os.environ.get("MVDB_BENCH_B", "256"))
    for S in [int(v) for v in os.environ.get("MVDB_BENCH_S", "32,512").split(",")]:
        rs = np.random.RandomState(S)
        ids = torch.from_numpy(rs.randint(5, 250000, size=(B, S)).astype(np.int32)).to(dev)
        for ragged in (False, True):
            lens = rs.randint(S // 4, S + 1, size=B) if ragged else np.full(B, S)
            mask = torch.from_numpy((np.arange(S)[None, :] < lens[:, None]).astype(np.int32)).to(dev)
            T = int(lens.sum())
            modes = os.environ.get("MVDB_BENCH_COMPUTE")
            modes = [int(v) for v in modes.split(",")] if modes else [0, 2]
            for compute in modes:
                for _ in range(3):
                    enc.forward_device(ids, mask, compute=compute)
                torch.cuda.synchronize()
                n = int(os.environ.get("MVDB_BENCH_REPS", "10"))
                t0 = time.perf_counter()
                for _ in range(n):
                    enc.forward_device(ids, mask, compute=compute)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n
                gemm = T * L * (4 * 2 * H * H + 2 * 2 * H * F)
                attn = float(sum(L * 4 * int(l) * int(l) * H for l in lens))
                print(json.dumps({"model": model, "B": B, "S": S, "ragged": ragged, "compute": {0: "fp32", 2: "fp16x3"}[compute],
                                  "tokens": T, "ms": round(dt * 1e3, 3), "sentences_per_s": round(B / dt, 1),
                                  "tflops": round((gemm + attn) / dt / 1e12, 2), "gemm_tflop": round(gemm / 1e12, 3),
                                  "attn_tflop": round(attn / 1e12, 3),
                                  # matrix-core products issued per algorithmic product, and the peak they run against
                                  "mfma_products_per_flop": {0: 1, 2: 3}[compute],
                                  "mfma_peak_tflops": {0: 157.3, 2: 2500.0}[compute],
                                  "frac_of_mfma_peak": round({0: 1, 2: 3}[compute] * (gemm + attn) / dt /
                                                             ({0: 157.3e12, 2: 2.5e15}[compute]), 4)}), flush=True)
    enc.close()


if __name__ == "__main__":
    main()
