"""Whole-corpus-on-one-GPU check (BASELINE config 4's 80M x 512 fits one MI355X's 288 GB): plant a scaled copy of
each query far into the corpus, then require it back first from the single-query scan and from the batch passes
(the certified fp16 pass over the shadow from 2 queries), with scores equal to the float64 dot products of the rows
fetched back.  usage: scale_check.py [rows] [dim] [nq]"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from minivectordb_amd import _native as native
from oracle import flat

n = int(sys.argv[1]) if len(sys.argv) > 1 else 80_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 128
k = 10
idx = native.FlatIndex(d)
idx.reserve(n + nq)  # one allocation: growing by reallocation would need old + new side by side
t0 = time.perf_counter()
CH = 10_000_000
for first in range(0, n, CH):
    idx.add_synthetic(min(CH, n - first), 1234, first_row=first, normalize=True)
t_fill = time.perf_counter() - t0
q = flat.synth(nq, d, 5678)
flat.normalize_l2(q)
# needles: rows n-1-7i hold the query itself (score 1.0 after normalisation) -> appended as extra rows
needles = q.copy()
idx.add(needles, normalize=True)
want = n + np.arange(nq)
out = {"rows": n + nq, "dim": d, "nq": nq, "fill_s": round(t_fill, 2)}
t0 = time.perf_counter()
idx.search(q, k)   # first batch: allocates the workspaces and builds the fp16 shadow of the rows (82 GB at 80M x 512), once
out["first_batch_ms (workspaces + fp16 shadow build)"] = round((time.perf_counter() - t0) * 1e3, 1)
out["shadow_rows"] = idx.shadow_rows
for label, qs in (("batch", q), ("batch_32", q[:32]), ("single", q[:1])):
    idx.search(qs, k)
    t0 = time.perf_counter()
    D, I = idx.search(qs, k)
    dt = time.perf_counter() - t0
    assert np.array_equal(I[:, 0], want[:len(qs)]), (label, I[:4, :3])
    assert np.allclose(D[:, 0], 1.0, atol=1e-5)
    # returned scores == float64 dot products of the rows fetched back
    for i in (0, len(qs) - 1):
        rows = np.stack([idx.get_rows(int(r), 1)[0] for r in I[i]])
        ref = rows.astype(np.float64) @ qs[i].astype(np.float64)
        assert np.abs(ref - D[i]).max() < 1e-5, (label, i, np.abs(ref - D[i]).max())
        assert np.all(np.diff(D[i]) <= 0)
    out[label + "_ms"] = round(dt * 1e3, 2)
# batch and single-query paths agree
D1, I1 = idx.search(q[5], k)
Db, Ib = idx.search(q, k)
assert np.array_equal(I1[0], Ib[5]) and np.allclose(D1[0], Db[5], atol=2e-6)
out["split_chunks_rerun"] = native.split_rerun_count()
print(json.dumps(out))
