import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from minivectordb_amd import _native as native
from oracle import flat
dev = torch.device("cuda", 0)
n, d, k = 1_000_000, 512, 10
base = flat.synth(n, d, 1234); flat.normalize_l2(base)
q = flat.synth(64, d, 5678); flat.normalize_l2(q)
def timeit(idx, nq, reps=20):
    qq = q[:nq]
    idx.search(qq, k)
    t0 = time.perf_counter()
    for _ in range(reps): idx.search(qq, k)
    return round((time.perf_counter() - t0) / reps * 1e3, 3)
for name, frac in (("no duplicates", 0.0), ("10% of the rows are one vector", 0.1), ("every row is the same vector", 1.0)):
    x = base.copy()
    if frac > 0:
        m = int(n * frac)
        sel = np.random.RandomState(1).choice(n, m, replace=False) if frac < 1 else np.arange(n)
        x[sel] = q[0]    # exact duplicates of the first query: the best match, tied m times
    idx = native.FlatIndex(d); idx.add(x)
    D, I = idx.search(q[:1], k)
    print(name, "| ms per call: 1 query", timeit(idx, 1), "8 queries", timeit(idx, 8), "64 queries", timeit(idx, 64), "| ids of query 0", I[0].tolist()[:5])
    if frac > 0:
        want = np.sort(sel)[:k]
        assert I[0].tolist() == want.tolist(), (I[0], want)
    idx.close()
