"""Determinism / batch-independence probe of the encoder (bitwise)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import encoder as E
from minivectordb_amd.embedding_model import GpuEncoder
cfg = E.make_config("e5-small-dims")
w = E.make_weights(cfg, 21)
enc = GpuEncoder(cfg, {k: torch.from_numpy(v) for k, v in w.items()}, device=0)
ids, mask = E.make_inputs(cfg, 8, 40, 22)
print("lens", mask.sum(1))
ref = enc.forward(ids, mask)
nd = sum(not np.array_equal(enc.forward(ids, mask), ref) for _ in range(20))
print("same batch, 20 repeats: differing runs =", nd)
for nb in (1, 2, 3, 5, 7):
    sub = enc.forward(ids[:nb], mask[:nb])
    d = np.abs(sub - ref[:nb]).max(axis=1)
    print("first", nb, "sentences vs batch of 8: max abs diff per row", d)
dev = torch.device("cuda", 0)
out, hid = enc.forward_device(torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev), want_hidden=True)
out3, hid3 = enc.forward_device(torch.from_numpy(ids[:3]).to(dev), torch.from_numpy(mask[:3]).to(dev), want_hidden=True)
torch.cuda.synchronize()
dh = (hid[:3] - hid3).abs().amax(dim=2).cpu().numpy()
print("hidden diff per (sentence, token) max:", dh.max(), "positions:", np.argwhere(dh > 0)[:12].tolist())
