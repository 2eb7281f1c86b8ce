#!/usr/bin/env python3
"""Randomised shapes through both parity modes of the encoder (compute = 0 exact fp32 MFMA, 2 = split fp16 x 3) and, for
the small cases, the float64 restatement: max |embedding difference| per case.  usage: fuzz_encoder_modes.py SEED SECONDS"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from oracle import encoder as E  # noqa: E402
from minivectordb_amd.embedding_model import GpuEncoder  # noqa: E402

rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 60)
encs = {}
for name in ("tiny", "hd64", "e5-small-dims", "xlmr-tiny", "xlmr-large-dims"):
    cfg = E.make_config(name)
    w = E.make_weights(cfg, 7)
    encs[name] = (cfg, w, GpuEncoder(cfg, {k: torch.from_numpy(v) for k, v in w.items()}, device=0))
cases = 0
worst = {"mode": 0.0, "f64_exact": 0.0, "f64_split": 0.0}
while time.time() < t_end:
    name = str(rs.choice(list(encs)))
    cfg, w, enc = encs[name]
    maxs = min(int(cfg["max_position_embeddings"]) - 4, 512)
    B = int(rs.choice([1, 2, 3, 7, 16, 63, 64, 65, 200]))
    S = int(rs.choice([1, 2, 5, 31, 32, 33, 64, 100, 127, 128, 129, 300, maxs]))
    S = min(S, maxs)
    if B * S > 40000:
        B = max(1, 40000 // S)
    ids, mask = E.make_inputs(cfg, B, S, int(rs.randint(1 << 30)), ragged=bool(rs.rand() < 0.7))
    if rs.rand() < 0.2:
        mask[:, rs.randint(0, S)] = 0            # a hole: non-prefix masks are honoured too
        mask[:, 0] = 1
    e0 = enc.forward(ids, mask, compute=0)
    e2 = enc.forward(ids, mask, compute=2)
    ok = np.isfinite(e0).all() and np.isfinite(e2).all()
    d = float(np.abs(e0 - e2).max())
    worst["mode"] = max(worst["mode"], d)
    msg = ""
    if B * S <= 600:
        _, e64 = E.numpy_forward(cfg, w, ids, mask)
        worst["f64_exact"] = max(worst["f64_exact"], float(np.abs(e0 - e64).max()))
        worst["f64_split"] = max(worst["f64_split"], float(np.abs(e2 - e64).max()))
    cases += 1
    if not ok or d > 2e-6:
        print("FAIL", name, B, S, d, ok, flush=True)
print("cases", cases, "worst |split - exact|", worst["mode"], "worst |exact - float64|", worst["f64_exact"],
      "worst |split - float64|", worst["f64_split"])
