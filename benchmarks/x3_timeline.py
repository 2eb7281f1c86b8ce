#!/usr/bin/env python3
"""Per-workgroup timeline of the split-precision GEMM kernel (ablation build only):
    make -C minivectordb_amd/csrc ABLATE=1
    MVDB_LIBMVDB=minivectordb_amd/lib/libmvdb_ablate.so MVDB_GEMM_X3_DBG=5 python benchmarks/x3_timeline.py [S]
(MVDB_GEMM_X3_DBG=6: the same with 16x16x32 stand-in MFMAs in the K loop — timing only, results wrong.)
Runs the e5-small-shaped forward of 256 x S tokens and reads the trace of the LAST traced GEMM launch (the last
layer's FFN1 when the N = H GEMMs run LayerNorm-fused): for every workgroup [start, first stage landed, K loop done,
stores issued, stores acknowledged] in 10-ns ticks of one chip-wide clock, plus the CU it ran on.  Prints the phase
lengths (median / mean) and, per CU, the gap between one workgroup's end and the next one's start."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from minivectordb_amd import _native  # noqa: E402
from minivectordb_amd.embedding_model import GpuEncoder  # noqa: E402
from oracle.encoder import make_weights  # noqa: E402

cfg = {"model_type": "bert", "vocab_size": 30000, "hidden_size": 384, "num_hidden_layers": 12,
       "num_attention_heads": 12, "intermediate_size": 1536, "max_position_embeddings": 512,
       "type_vocab_size": 2, "layer_norm_eps": 1e-12, "hidden_act": "gelu", "pad_token_id": 0}
dev = torch.device("cuda", 0)
w = make_weights(cfg, 1)
enc = GpuEncoder(cfg, {k: torch.from_numpy(v) for k, v in w.items()}, device=0)
B, S = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 512
rs = np.random.RandomState(0)
ids = torch.from_numpy(rs.randint(5, 30000, size=(B, S)).astype(np.int32)).to(dev)
mask = torch.ones((B, S), dtype=torch.int32, device=dev)
os.environ["MVDB_ENCODER_GRAPH"] = "0"
for _ in range(3):
    enc.forward_device(ids, mask)
torch.cuda.synchronize()
lib = _native.lib()
fn = lib.mvdb_debug_x3_trace
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
fn.restype = ctypes.c_int
NB = 16384
buf = np.zeros((NB, 16), np.uint64)
assert fn(buf.ctypes.data, NB) == 0
enc.forward_device(ids, mask)  # one traced forward from a clean buffer
torch.cuda.synchronize()
assert fn(buf.ctypes.data, NB) == 0
t = buf[buf[:, 3] != 0]
print(f"workgroups traced: {len(t)}")
t0 = t[:, 3].min()
tt = (t[:, 3:8].astype(np.int64) - int(t0)) * 0.01  # us
names = ["launch->first stage", "K loop", "epilogue issue", "store drain"]
for i, nme in enumerate(names):
    d = tt[:, i + 1] - tt[:, i]
    print(f"{nme:22s} median {np.median(d):7.2f} us  mean {d.mean():7.2f}  p10 {np.percentile(d, 10):7.2f}  p90 {np.percentile(d, 90):7.2f}")
kl_t = (t[:, 5].astype(np.int64) - t[:, 4].astype(np.int64)) * 10e-9  # s
kl_c = t[:, 9].astype(np.int64) - t[:, 8].astype(np.int64)
ok = (kl_t > 0) & (kl_c > 0)
if ok.any():
    ghz = kl_c[ok] / kl_t[ok] / 1e9
    print(f"shader clock over the K loop (s_memtime / s_memrealtime): median {np.median(ghz):.3f} GHz  p10 {np.percentile(ghz, 10):.3f}  p90 {np.percentile(ghz, 90):.3f}; "
          f"K loop median {np.median(kl_c[ok]):.0f} cycles")
print(f"whole kernel: {tt[:, 4].max():.1f} us; per workgroup start->acknowledged median {(np.median(tt[:, 4] - tt[:, 0])):.2f} us")
# per CU: (xcc, se, sh?, cu) from HW_ID: cu_id bits 11:8, sh_id 12, se_id 15:13 (gfx9 layout)
hw = t[:, 1].astype(np.int64)
cu = ((t[:, 2].astype(np.int64) & 0xF) << 16) | (hw & 0xFF00)
gaps, spans = [], []
for c in np.unique(cu):
    rows = tt[cu == c]
    rows = rows[np.argsort(rows[:, 0])]
    if len(rows) > 1:
        gaps.extend(rows[1:, 0] - rows[:-1, 4])
    spans.append(len(rows))
gaps = np.array(gaps)
print(f"CUs seen: {len(np.unique(cu))}; workgroups per CU: min {min(spans)} max {max(spans)}")
print(f"gap between a workgroup's last acknowledged store and the next start on its CU: median {np.median(gaps):.2f} us "
      f"mean {gaps.mean():.2f} p10 {np.percentile(gaps, 10):.2f} p90 {np.percentile(gaps, 90):.2f} (negative = overlap)")
# when do epilogues happen relative to one another (lock-step?): histogram of 'K loop done' times
edges = np.arange(0, tt[:, 4].max() + 5, 5.0)
hk, _ = np.histogram(tt[:, 2], edges)
hs, _ = np.histogram(tt[:, 0], edges)
print("5-us bins: workgroups starting | workgroups finishing their K loop")
for i in range(len(edges) - 1):
    print(f"  {edges[i]:6.0f}  {hs[i]:5d} {hk[i]:5d}")
