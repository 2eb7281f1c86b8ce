# A/B of the attention kernel's MFMA chains: product build vs MVDB_ATTN_ILP build (make EXTRA=-DMVDB_ATTN_ILP SUFFIX=_ilp)
cd $GRAFT_REPO_ROOT
for lib in minivectordb_amd/lib/libmvdb.so minivectordb_amd/lib/libmvdb_ilp.so; do
echo "== $lib"
MVDB_LIBMVDB=$lib python3 benchmarks/long_sentence_probe.py --variants default --lengths 192,256,384,512 | python3 -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['shape'], r['S'], r['p50_ms'])"
MVDB_LIBMVDB=$lib python3 benchmarks/long_sentence_probe.py --variants default --large --lengths 256,512 | python3 -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['shape'], r['S'], r['p50_ms'])"
for S in 32 128 512; do
MVDB_LIBMVDB=$lib MVDB_S32_S=$S python3 - <<'PY'
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
n = 40 if S <= 32 else 8
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(n): enc.forward_device(ids, mask)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / n * 1e3)
print(f"batch 256 x {S}: best of 5 = {min(ts):.3f} ms per forward")
PY
done
done
echo "== parity of the ILP build"
MVDB_LIBMVDB=minivectordb_amd/lib/libmvdb_ilp.so timeout 900 python3 -m pytest tests/test_encoder_gpu.py -x -q -m gpu 2>&1 | tail -3
