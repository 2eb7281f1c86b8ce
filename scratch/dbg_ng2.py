import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import flat
from minivectordb_amd import _native as native
n, d, k = 20000, 512, 10
x = flat.synth(n, d, 1234); flat.normalize_l2(x)
idx = native.FlatIndex(d); idx.add(x)
for nq in (17, 32, 20):
    q = flat.synth(nq, d, 5678); flat.normalize_l2(q)
    D1 = np.concatenate([idx.search(q[i], k)[0] for i in range(nq)])
    for rep in range(3):
        D, I = idx.search(q, k)
        err = np.abs(D - D1).max(axis=1)
        bad = np.nonzero(err > 1e-5)[0]
        print(nq, rep, "bad queries:", bad.tolist(), err[bad].round(5).tolist())
