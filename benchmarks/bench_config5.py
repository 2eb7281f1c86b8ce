#!/usr/bin/env python3
"""BASELINE config 5 end to end on one MI355X: e5-small-shaped encoder forward (batch 256
sentences, random-init weights, synthetic token ids) -> pooled, normalised 384-d embeddings (never
leave the device) -> kNN (k = 10) over a resident 10M x 384 fp32 corpus, 32 queries per MFMA pass.
Prints one JSON line per sequence length."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from minivectordb_amd import _native as native  # noqa: E402
from minivectordb_amd.embedding_model import GpuEncoder  # noqa: E402
from oracle.encoder import weight_names  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(dev))  # a real stream: the encoder replays hipGraphs on it
    H, F, V = 384, 1536, 250037
    cfg = {"model_type": "bert", "vocab_size": V, "hidden_size": H, "num_hidden_layers": 12,
           "num_attention_heads": 12, "intermediate_size": F, "max_position_embeddings": 512,
           "type_vocab_size": 2, "layer_norm_eps": 1e-12, "hidden_act": "gelu", "pad_token_id": 0}
    g = torch.Generator(device="cpu").manual_seed(0)
    shapes = {"embeddings.word_embeddings.weight": (V, H), "embeddings.position_embeddings.weight": (512, H),
              "embeddings.token_type_embeddings.weight": (2, H)}
    sd = {}
    for name in weight_names(cfg):
        if name in shapes:
            shape = shapes[name]
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
        sd[name] = t + 1.0 if "LayerNorm.weight" in name else t
    enc = GpuEncoder(cfg, sd, device=0)
    n, k, B = 10_000_000, 10, 256
    idx = native.FlatIndex(H, device=0)
    idx.reserve(n)
    idx.add_synthetic(n, 1234, normalize=True)
    D = torch.empty((B, k), dtype=torch.float32, device=dev)
    I = torch.empty((B, k), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    for S in (32, 512):
        rs = np.random.RandomState(S)
        lens = rs.randint(max(1, S // 4), S + 1, size=B)
        ids = torch.from_numpy(rs.randint(5, 250000, size=(B, S)).astype(np.int32)).to(dev)
        mask = torch.from_numpy((np.arange(S)[None, :] < lens[:, None]).astype(np.int32)).to(dev)

        def step():
            emb, _ = enc.forward_device(ids, mask)
            idx.search_device(emb.data_ptr(), B, k, D.data_ptr(), I.data_ptr(), stream=stream)
            return emb

        for _ in range(2):
            step()
        torch.cuda.synchronize()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            enc.forward_device(ids, mask)
        torch.cuda.synchronize()
        t_enc = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / reps
        print(json.dumps({"config": "e5-small forward (B=256) + 10M x 384 kNN, k=10", "S": S,
                          "tokens": int(lens.sum()), "encoder_ms": round(t_enc * 1e3, 3),
                          "knn_ms": round((t_all - t_enc) * 1e3, 3), "end_to_end_ms": round(t_all * 1e3, 3),
                          "sentences_per_s": round(B / t_all, 1),
                          # up to 256 queries per corpus pass on the certified fp16 pass (k <= 12)
                          "encoder_compute": {0: "fp32", 2: "fp16x3"}[enc.default_compute],
                          "knn_corpus_passes": -(-B // max(native.half_max_queries(H), 128)),
                          "knn_GBps": round(-(-B // max(native.half_max_queries(H), 128)) * n * H * 4 / (t_all - t_enc) / 1e9, 1)}), flush=True)
    idx.close()
    enc.close()


if __name__ == "__main__":
    main()
