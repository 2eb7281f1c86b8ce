#!/usr/bin/env python3
"""Encoder forward at the B = 256, S = 32 shape only (for rocprofv3 kernel traces).
MVDB_S32_MODEL=e5-large: the XLM-R-large shape (H 1024, 24 layers, FFN 4096); MVDB_S32_S: another sequence length."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "benchmarks"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from minivectordb_amd.embedding_model import GpuEncoder  # noqa: E402
from oracle.encoder import make_weights  # noqa: E402

cfg = {"model_type": "bert", "vocab_size": 30000, "hidden_size": 384, "num_hidden_layers": 12,
       "num_attention_heads": 12, "intermediate_size": 1536, "max_position_embeddings": 512,
       "type_vocab_size": 2, "layer_norm_eps": 1e-12, "hidden_act": "gelu", "pad_token_id": 0}
if os.environ.get("MVDB_S32_MODEL") == "e5-large":
    cfg.update({"hidden_size": 1024, "num_hidden_layers": 24, "num_attention_heads": 16, "intermediate_size": 4096})
dev = torch.device("cuda", 0)
w = make_weights(cfg, 1)
enc = GpuEncoder(cfg, {k: torch.from_numpy(v) for k, v in w.items()}, device=0)
B, S = 256, int(os.environ.get("MVDB_S32_S", "32"))
rs = np.random.RandomState(0)
ids = torch.from_numpy(rs.randint(5, 30000, size=(B, S)).astype(np.int32)).to(dev)
mask = torch.ones((B, S), dtype=torch.int32, device=dev)
if len(sys.argv) > 2 and sys.argv[2] == "ragged":
    lens = rs.randint(S // 4, S + 1, size=B)
    mask = torch.from_numpy((np.arange(S)[None, :] < lens[:, None]).astype(np.int32)).to(dev)
os.environ["MVDB_ENCODER_GRAPH"] = "0"
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    enc.forward_device(ids, mask)
torch.cuda.synchronize()
