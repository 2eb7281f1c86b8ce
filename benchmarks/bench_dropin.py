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
    # writes followed by a query: the next query uploads only the new row / the delete compacts the device tail
    app, dele, dele_early = [], [], []
    for j in range(5):
        t0 = time.perf_counter()
        db.store_embedding(f"new{j}", x[j] * 0.5 + x[j + 1], {"bucket": 7})
        ids2, _, _ = db.find_most_similar(q[j], k=10)
        app.append(time.perf_counter() - t0)
    assert ids2 is not None and len(db.inverse_id_map) == n + 5
    for j, uid in enumerate((123456, 700001, 999999, 500000, 250000)):
        t0 = time.perf_counter()
        db.delete_embedding(uid)
        ids3, _, _ = db.find_most_similar(q[j], k=10)
        dele.append(time.perf_counter() - t0)
        assert uid not in ids3
    for j, uid in enumerate((5, 6, 7)):   # the whole matrix moves up
        t0 = time.perf_counter()
        db.delete_embedding(uid)
        ids3, _, _ = db.find_most_similar(q[j], k=10)
        dele_early.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    idsf, _, metaf = db.find_most_similar(q[0], k=10, metadata_filter={"bucket": 31})  # first filter after deletes
    t_filter_after_delete = time.perf_counter() - t0
    assert all(m["bucket"] == 31 for m in metaf)
    # round 4: the value index and the id index are maintained through writes, row sets are cached per filter + generation
    inc = {}
    db.find_most_similar(q[0], k=10, metadata_filter={"bucket": 31})
    t0 = time.perf_counter(); db.find_most_similar(q[1], k=10, metadata_filter={"bucket": 31})
    inc["repeated_filter_query_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
    t = []
    for j in range(5):
        db.store_embedding(f"inc{j}", x[10 + j] * 0.25 + x[j], {"bucket": 31, "i": -j})
        t0 = time.perf_counter()
        idsf, _, metaf = db.find_most_similar(x[10 + j] * 0.25 + x[j], k=10, metadata_filter={"bucket": 31})
        t.append(time.perf_counter() - t0)
        assert idsf[0] == f"inc{j}" and all(m["bucket"] == 31 for m in metaf)
    inc["first_filtered_query_after_a_store_ms"] = round(float(np.median(t)) * 1e3, 3)
    t = []
    for j, uid in enumerate((31, 100031, 200031, 300031, 400031)):
        db.delete_embedding(uid)
        t0 = time.perf_counter()
        idsf, _, metaf = db.find_most_similar(q[j], k=10, metadata_filter={"bucket": 31})
        t.append(time.perf_counter() - t0)
        assert uid not in idsf and all(m["bucket"] == 31 for m in metaf)
    inc["first_filtered_query_after_a_delete_ms"] = round(float(np.median(t)) * 1e3, 3)
    t0 = time.perf_counter()
    for j in range(1000):
        db.store_embedding(f"burst{j}", x[1000 + j], {"bucket": j % 100})
    t_store = time.perf_counter() - t0
    t0 = time.perf_counter()
    idsb, _, _ = db.find_most_similar(x[1500], k=10)
    inc["store_1000_singles_s"] = round(t_store, 4)
    inc["query_after_1000_single_stores_ms (one coalesced add)"] = round((time.perf_counter() - t0) * 1e3, 3)
    assert "burst500" in idsb[:2]
    # the batch API (an extension): 64 queries under no filter / under one filter in ONE call
    qb = q[:64]
    db.find_most_similar_batch(qb, k=10)
    t0 = time.perf_counter(); many = db.find_most_similar_batch(qb, k=10)
    inc["find_most_similar_batch_64_queries_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
    db.find_most_similar_batch(qb, k=10, metadata_filter={"bucket": 31})
    t0 = time.perf_counter(); db.find_most_similar_batch(qb, k=10, metadata_filter={"bucket": 31})
    inc["find_most_similar_batch_64_queries_filtered_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
    assert list(many[5][0]) == list(db.find_most_similar(qb[5], k=10)[0])
    v = db.get_vector("new0")
    assert abs(float(np.linalg.norm(v)) - 1.0) < 1e-5   # read back from the device, normalised there
    t_append_query, t_delete_query = float(np.median(app)), float(np.median(dele))
    # the device half alone at 10M x 512 (C-ABI): append one row, delete row 5, one query after each
    from minivectordb_amd import _native
    big = _native.FlatIndex(d)
    big.reserve(10_000_100)
    big.add_synthetic(10_000_000, 1234, normalize=True)
    big.search(q[0], 10, normalize_q=True)
    dev = {}
    t0 = time.perf_counter(); big.add(x[:1], normalize=True); big.search(q[0], 10, normalize_q=True)
    dev["append_one_then_query_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    t0 = time.perf_counter(); big.remove_rows([5]); big.search(q[0], 10, normalize_q=True)
    dev["delete_row5_then_query_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    t0 = time.perf_counter(); big.remove_rows([9_999_000]); big.search(q[0], 10, normalize_q=True)
    dev["delete_late_row_then_query_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    big.close()
    # the sharded class at the same size: its own bookkeeping (ids, metadata, inverted index, device rows) against the
    # shard FILE rewrite every delete performs by contract
    from minivectordb_amd import ShardedVectorDatabase
    sdir = tempfile.mkdtemp()
    sdb = ShardedVectorDatabase(storage_dir=sdir, shard_size=5000)
    t0 = time.perf_counter()
    for s0 in range(0, n, step):
        sdb.store_embeddings_batch(list(range(s0, s0 + step)), x[s0:s0 + step],
                                   [{"bucket": i % 100, "i": i} for i in range(s0, s0 + step)])
    sh = {"ingest_s": round(time.perf_counter() - t0, 2)}
    sdb.find_most_similar(q[0], k=10)
    file_s = [0.0]
    rewrite = sdb._remove_embeddings_from_shard

    def timed_rewrite(shard_id, uids):
        a = time.perf_counter()
        rewrite(shard_id, uids)
        file_s[0] += time.perf_counter() - a
    sdb._remove_embeddings_from_shard = timed_rewrite
    tot, book = [], []
    for j, uid in enumerate((123456, 700001, 999999, 500000, 250000)):
        file_s[0] = 0.0
        t0 = time.perf_counter()
        sdb.delete_embeddings_batch([uid])
        ids4, _, _ = sdb.find_most_similar(q[j], k=10)
        el = time.perf_counter() - t0
        tot.append(el)
        book.append(el - file_s[0])
        assert uid not in ids4
    sh["delete_one_then_query_ms"] = round(float(np.median(tot)) * 1e3, 2)
    sh["of_which_bookkeeping_device_and_query_ms"] = round(float(np.median(book)) * 1e3, 2)
    t0 = time.perf_counter()
    idsf, _, metaf = sdb.find_most_similar(q[0], k=10, metadata_filter={"bucket": 31})
    sh["first_filtered_query_ms (builds the value index of one key)"] = round((time.perf_counter() - t0) * 1e3, 2)
    sdb.delete_embeddings_batch([31])
    t0 = time.perf_counter()
    idsf, _, metaf = sdb.find_most_similar(q[0], k=10, metadata_filter={"bucket": 31})
    sh["first_filtered_query_after_a_delete_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
    assert 31 not in idsf and all(m["bucket"] == 31 for m in metaf)
    import shutil
    shutil.rmtree(sdir, ignore_errors=True)
    print(json.dumps({
        "config": "VectorDatabase drop-in, 1M x 512, k=10",
        "incremental_filter_state": inc,
        "sharded_class_1M_x_512_shard_size_5000": sh,
        "ingest_s": round(t_ingest, 2), "first_query_ms": round(t_first * 1e3, 1),
        "query_p50_ms": round(float(np.median(lat)) * 1e3, 3), "query_qps": round(1.0 / float(np.mean(lat)), 1),
        "filtered_query_p50_ms (1% of rows, Python filter + device subset search)": round(float(np.median(latf)) * 1e3, 3),
        "append_one_then_query_ms": round(t_append_query * 1e3, 2),
        "delete_one_then_query_ms": round(t_delete_query * 1e3, 2),
        "delete_early_row_then_query_ms": round(float(np.median(dele_early)) * 1e3, 2),
        "first_filtered_query_after_deletes_ms": round(t_filter_after_delete * 1e3, 2),
        "device_only_10M_x_512": dev}), flush=True)


if __name__ == "__main__":
    main()
