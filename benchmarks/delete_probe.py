#!/usr/bin/env python3
"""What one delete costs on a resident corpus (mvdb_index_remove_rows alone, device-filled rows): one early row, one late row, a
run of 16 rows, 8 / 64 / 1,000 scattered rows — with the one-pass in-place shift (default) and with the staging path
(MVDB_COMPACT_INPLACE=0).  usage: delete_probe.py [rows] [dim]; one JSON line per mode."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from minivectordb_amd import _native as native  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
rng = np.random.RandomState(5)
for inplace in (1, 0):
    os.environ["MVDB_COMPACT_INPLACE"] = str(inplace)
    idx = native.FlatIndex(d)
    idx.add_synthetic(n, seed=1234, normalize=True)
    idx.remove_rows(np.array([n - 1], np.int64))           # allocations of the first delete
    out = {"rows": n, "d": d, "inplace": inplace}
    for name, dels in (("early_row", [5]), ("middle_row", [n // 2]), ("late_row", [n - 1000]), ("run_of_16", list(range(100, 116))),
                       ("scattered_8", sorted(rng.choice(n // 2, 8, replace=False).tolist())),
                       ("scattered_64", sorted(rng.choice(n // 2, 64, replace=False).tolist())),
                       ("scattered_1000", sorted(rng.choice(n // 2, 1000, replace=False).tolist()))):
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            idx.remove_rows(np.array(dels, np.int64))
            ts.append((time.perf_counter() - t0) * 1e3)
        out[name + "_ms"] = round(sorted(ts)[1], 3)
    idx.close()
    print(json.dumps(out), flush=True)
