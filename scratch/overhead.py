import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import flat
from minivectordb_amd import _native as native
d=512
for n in (1000, 100000, 1000000):
    idx = native.FlatIndex(d); idx.add_synthetic(n, 1, normalize=True)
    q = flat.synth(1, d, 2)
    for _ in range(50): idx.search(q, 10)
    t0=time.perf_counter()
    for _ in range(1000): idx.search(q, 10)
    print("host API search on %d rows: %.1f us/call" % (n, (time.perf_counter()-t0)/1000*1e6))
    idx.close()
