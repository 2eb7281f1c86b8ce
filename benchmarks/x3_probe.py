#!/usr/bin/env python3
"""Parity of the encoder's compute modes against transformers' golden outputs (tests/golden/encoder_golden.npz):
max |embedding error| and max |hidden error| per case for compute = 0 (exact fp32 MFMA) and 2 (fp16 x 3 split)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from encoder_cases import load_cases  # noqa: E402
from oracle import encoder as E  # noqa: E402
from minivectordb_amd.embedding_model import GpuEncoder  # noqa: E402

dev = torch.device("cuda", 0)
for c in load_cases():
    cfg = E.make_config(c["name"])
    w = E.make_weights(cfg, c["wseed"])
    enc = GpuEncoder(cfg, {k: torch.from_numpy(v) for k, v in w.items()}, device=0)
    row = {"case": c["name"], "shape": list(c["ids"].shape)}
    for compute in (0, 2):
        out, hidden = enc.forward_device(torch.from_numpy(c["ids"]).to(dev), torch.from_numpy(c["mask"]).to(dev),
                                         compute=compute, want_hidden=True)
        torch.cuda.synchronize()
        row[f"emb_err_c{compute}"] = float(np.abs(out.cpu().numpy() - c["emb"]).max())
        if c["hidden_valid"] is not None:
            m = c["mask"].astype(bool)
            row[f"hid_err_c{compute}"] = float(np.abs(hidden.cpu().numpy()[m] - c["hidden_valid"]).max())
    print(json.dumps(row), flush=True)
    enc.close()
