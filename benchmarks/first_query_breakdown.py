#!/usr/bin/env python3
"""Where the first query's time goes (drop-in at 1M x 512): library load, first kernel launch of each code object,
upload, first search, steady state."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t = {}
t0 = time.perf_counter()
import numpy as np
t["import_numpy"] = time.perf_counter() - t0
t0 = time.perf_counter()
from minivectordb_amd import _native as native
native.lib()
t["load_libmvdb (torch's HIP runtime pre-loaded by path, torch NOT imported: %s)" % ("torch" not in sys.modules)] = time.perf_counter() - t0
t0 = time.perf_counter()
native.device_count()
t["hip_runtime_init (first HIP call)"] = time.perf_counter() - t0
from oracle import flat
d = 512
x = flat.synth(1_000_000, d, 1)
q = flat.synth(4, d, 2)
t0 = time.perf_counter()
idx = native.FlatIndex(d)
t["index_create"] = time.perf_counter() - t0
t0 = time.perf_counter()
idx.add(x[:1000], normalize=True)
t["first_add_1000_rows (first kernel of the search code object)"] = time.perf_counter() - t0
t0 = time.perf_counter()
idx.add(x[1000:], normalize=True)
t["add_999k_rows (2 GB over PCIe + normalise)"] = time.perf_counter() - t0
for name in ("first_search", "second_search", "third_search"):
    t0 = time.perf_counter()
    idx.search(q[0], 10, normalize_q=True)
    t[name] = time.perf_counter() - t0
t0 = time.perf_counter()
idx.search(q, 10)
t["first_4_query_search (MFMA pass kernels)"] = time.perf_counter() - t0
t0 = time.perf_counter()
idx.search_subset(q[0], 10, np.arange(0, 1_000_000, 100, dtype=np.int64))
t["first_subset_search"] = time.perf_counter() - t0
t0 = time.perf_counter()
idx.search(flat.synth(128, d, 3), 10)
t["first_128_query_search (half_scan code object)"] = time.perf_counter() - t0
print(json.dumps({k: round(v * 1e3, 2) for k, v in t.items()}, indent=1))
