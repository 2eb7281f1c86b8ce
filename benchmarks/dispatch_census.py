#!/usr/bin/env python3
"""Which kernel generation answers which call: a grid of (rows, width, queries per call, k, metric, filter) through the
product's own dispatcher with the library's profiling hooks on, launches counted per kernel family (mvdb_prof_read).
Round 4's review asked what still reached the bf16-split kernels and the fp16 nomination over fp32 rows once batches of
2+ queries streamed the fp16 shadow; both were retired in round 6 on this census (profiles/r06_dispatch_census.json).  One JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from minivectordb_amd import _native as native  # noqa: E402

FAMILIES = ("ip_scan", "ip_scan_scores", "ip_scan_mfma", "ip_scan_gemm", "ip_scan_half", "ip_scan_half_seed", "ip_scan_rescue",
            "ip_scan_rerun")
out = {"grid": [], "families": list(FAMILIES)}
native.prof_enable(True)
for d in (64, 128, 192, 256, 384, 512, 768, 1024, 1536):
    for n in (20_000, 200_000, 1_000_000):
        if n * d * 4 > 3e9:
            continue
        for metric in (native.METRIC_IP, native.METRIC_L2):
            idx = native.FlatIndex(d, metric=metric)
            idx.add_synthetic(n, 1234, normalize=True)
            rs = np.random.RandomState(1)
            for nq in (1, 2, 8, 16, 33, 130):
                for k in (10, 40):
                    q = rs.standard_normal((nq, d)).astype(np.float32)
                    q /= np.linalg.norm(q, axis=1, keepdims=True)
                    for name in FAMILIES:
                        native.prof_read(name)
                    idx.search(q, k)
                    used = {name: native.prof_read(name)[0] for name in FAMILIES}
                    sym = {name: native.prof_symbol(name).split("<")[0] for name, c in used.items() if c}
                    out["grid"].append({"d": d, "n": n, "metric": "ip" if metric == native.METRIC_IP else "l2", "nq": nq, "k": k,
                                        "launches": {kk: v for kk, v in used.items() if v}, "kernels": sym})
            idx.close()
native.prof_enable(False)
tot = {}
for g in out["grid"]:
    for name in g["kernels"].values():
        tot[name] = tot.get(name, 0) + 1
out["grid_points_reaching_kernel"] = tot
print(json.dumps(out))
