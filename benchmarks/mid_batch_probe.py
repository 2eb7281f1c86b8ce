"""Encoder forward of mid-size batches (4 .. 256 sentences), ms per forward through the device entry, best of 5 x 30: where the
split-K rule of the N = H GEMMs applies (x3_splitk_parts).  usage: mid_batch_probe.py [--large]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from minivectordb_amd.embedding_model import GpuEncoder
from oracle.encoder import make_weights
large = "--large" in sys.argv
cfg = {"model_type": "bert", "vocab_size": 30000, "hidden_size": 384, "num_hidden_layers": 12, "num_attention_heads": 12, "intermediate_size": 1536, "max_position_embeddings": 512, "type_vocab_size": 2, "layer_norm_eps": 1e-12, "hidden_act": "gelu", "pad_token_id": 0}
if large: cfg.update(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)
dev = torch.device("cuda", 0)
w = make_weights(cfg, 1)
enc = GpuEncoder(cfg, {k: torch.from_numpy(v) for k, v in w.items()}, device=0)
rs = np.random.RandomState(0)
out = []
for B, S in ((4, 32), (8, 32), (16, 32), (32, 32), (48, 32), (64, 32), (96, 32), (128, 32), (192, 32), (256, 32), (16, 128), (32, 128), (8, 512)):
    ids = torch.from_numpy(rs.randint(5, 30000, size=(B, S)).astype(np.int32)).to(dev)
    mask = torch.ones((B, S), dtype=torch.int32, device=dev)
    for _ in range(5): enc.forward_device(ids, mask)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(30): enc.forward_device(ids, mask)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 30 * 1e3)
    out.append(f"{B}x{S}:{min(ts):.3f}")
print(("large " if large else "small ") + "MVDB_GEMM_X3_SPLITK=" + os.environ.get("MVDB_GEMM_X3_SPLITK", "1") + " PARTS=" + os.environ.get("MVDB_GEMM_X3_SPLITK_PARTS", "-") + "  " + "  ".join(out))
