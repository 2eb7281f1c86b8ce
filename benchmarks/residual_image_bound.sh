cd $GRAFT_REPO_ROOT
ABL=minivectordb_amd/lib/libmvdb_ablate.so
for rep in 1 2; do
for sk in 0 1; do
for S in 32 512; do
MVDB_LIBMVDB=$ABL MVDB_LN_SKIP_X=$sk MVDB_S32_S=$S python3 - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "benchmarks")
import numpy as np, torch
from minivectordb_amd.embedding_model import GpuEncoder
from oracle.encoder import make_weights
cfg = {"model_type": "bert", "vocab_size": 30000, "hidden_size": 384, "num_hidden_layers": 12, "num_attention_heads": 12, "intermediate_size": 1536, "max_position_embeddings": 512, "type_vocab_size": 2, "layer_norm_eps": 1e-12, "hidden_act": "gelu", "pad_token_id": 0}
dev = torch.device("cuda", 0)
w = make_weights(cfg, 1)
enc = GpuEncoder(cfg, {k: torch.from_numpy(v) for k, v in w.items()}, device=0)
B, S = 256, int(os.environ["MVDB_S32_S"])
rs = np.random.RandomState(0)
ids = torch.from_numpy(rs.randint(5, 30000, size=(B, S)).astype(np.int32)).to(dev)
mask = torch.ones((B, S), dtype=torch.int32, device=dev)
for _ in range(5): enc.forward_device(ids, mask)
torch.cuda.synchronize()
n = 40 if S == 32 else 8
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(n): enc.forward_device(ids, mask)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / n * 1e3)
print(f"S={S} MVDB_LN_SKIP_X={os.environ['MVDB_LN_SKIP_X']}: ms per forward, 5 runs: " + " ".join(f"{t:.3f}" for t in ts) + f"  (best {min(ts):.3f})")
PY
done; done; done
