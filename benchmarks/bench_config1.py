#!/usr/bin/env python3
"""BASELINE config 1: 1k vectors x dim 512, k=5, through VectorDatabase.find_most_similar (the reference's own
CPU-runnable case).  Times the drop-in (GPU) and, beside it, the CPU restatement of the reference's faiss calls
behind the same Python plumbing (tests/oracle_backend.OracleIndex standing in for faiss.IndexFlatIP).  One JSON line."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from minivectordb_amd import VectorDatabase  # noqa: E402
from oracle import flat  # noqa: E402


def main():
    n, d, k = 1000, 512, 5
    x = flat.synth(n, d, 1234)
    q = flat.synth(256, d, 5678)
    db = VectorDatabase(storage_file=os.path.join(tempfile.mkdtemp(), "db.pkl"))
    db.store_embeddings_batch(list(range(n)), x, [{"i": i} for i in range(n)])
    db.find_most_similar(q[0], k=k)
    lat = []
    for i in range(256):
        t0 = time.perf_counter()
        ids, dist, meta = db.find_most_similar(q[i], k=k)
        lat.append(time.perf_counter() - t0)
    # CPU port of the faiss calls on the same data (1 thread, as faiss at nq = 1)
    xs = x.copy()
    flat.normalize_l2(xs)
    qn = q.copy()
    flat.normalize_l2(qn)  # outside the timed loop: alternating thread counts makes libgomp rebuild its pool per call
    cl = []
    for i in range(256):
        t0 = time.perf_counter()
        flat.flat_search(xs, qn[i:i + 1], k, nthreads=1)
        cl.append(time.perf_counter() - t0)
    # parity on this config: same ids as the oracle for every query
    same = 0
    for i in range(256):
        Do, Io = flat.flat_search(xs, qn[i:i + 1], k, nthreads=1)
        ids, dist, meta = db.find_most_similar(q[i], k=k)
        same += int(list(ids) == Io[0].tolist())
    print(json.dumps({"config": "VectorDatabase.find_most_similar, 1k x 512, k=5 (BASELINE config 1)",
                      "dropin_p50_ms": round(float(np.median(lat)) * 1e3, 4),
                      "dropin_qps": round(len(lat) / sum(lat), 1),
                      "cpu_port_scan_p50_ms": round(float(np.median(cl)) * 1e3, 4),
                      "queries_with_identical_ids": f"{same}/256"}))


if __name__ == "__main__":
    main()
