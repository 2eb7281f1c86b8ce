#!/usr/bin/env python3
"""The drop-in classes end to end at 1M x 512 (BASELINE config 2 through the reference's own Python
API): ingest, first query (uploads + normalises on the device), steady-state query latency with and
without a metadata filter.  One JSON line."""
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
    n, d = 1_000_000, 512
    x = flat.synth(n, d, 1234)
    q = flat.synth(64, d, 5678)
    db = VectorDatabase(storage_file=os.path.join(tempfile.mkdtemp(), "db.pkl"))
    t0 = time.perf_counter()
    step = 100_000
    for s in range(0, n, step):
        db.store_embeddings_batch(list(range(s, s + step)), x[s:s + step],
                                  [{"bucket": i % 100, "i": i} for i in range(s, s + step)])
    t_ingest = time.perf_counter() - t0
    t0 = time.perf_counter()
    ids, dist, meta = db.find_most_similar(q[0], k=10)
    t_first = time.perf_counter() - t0
    lat = []
    for i in range(64):
        t0 = time.perf_counter()
        ids, dist, meta = db.find_most_similar(q[i], k=10)
        lat.append(time.perf_counter() - t0)
    latf = []
    for i in range(32):
        t0 = time.perf_counter()
        idsf, distf, metaf = db.find_most_similar(q[i], k=10, metadata_filter={"bucket": i % 100})
        latf.append(time.perf_counter() - t0)
    assert all(m["bucket"] == 31 for m in metaf)
    # check one result against the oracle on the normalised host matrix the class exposes
    qn = q[63:64].copy()
    flat.normalize_l2(qn)
    Do, Io = flat.flat_search(db.embeddings, qn, 10, nthreads=flat.max_threads())
    assert list(ids) == Io[0].tolist(), (ids, Io[0])
    t0 = time.perf_counter()
    db.store_embedding("new", x[0] * 0.5 + x[1], {"bucket": 7})
    ids2, _, _ = db.find_most_similar(q[0], k=10)
    t_append_query = time.perf_counter() - t0
    t0 = time.perf_counter()
    db.delete_embedding(123456)
    ids3, _, _ = db.find_most_similar(q[0], k=10)
    t_delete_query = time.perf_counter() - t0
    print(json.dumps({
        "config": "VectorDatabase drop-in, 1M x 512, k=10",
        "ingest_s": round(t_ingest, 2), "first_query_ms": round(t_first * 1e3, 1),
        "query_p50_ms": round(float(np.median(lat)) * 1e3, 3), "query_qps": round(1.0 / float(np.mean(lat)), 1),
        "filtered_query_p50_ms (1% of rows, Python filter + device subset search)": round(float(np.median(latf)) * 1e3, 3),
        "append_one_then_query_ms": round(t_append_query * 1e3, 2),
        "delete_one_then_query_ms": round(t_delete_query * 1e3, 2)}), flush=True)


if __name__ == "__main__":
    main()
