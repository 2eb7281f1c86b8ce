#!/usr/bin/env python3
"""Secondary measurements on one MI355X (10M x 512 fp32 resident unless noted), host API (numpy in /
numpy out, PCIe-inclusive): filtered (subset) search at several selectivities,
large-k select path, L2 metric, add+normalise, remove_rows.  One JSON line each."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from minivectordb_amd import _native as native  # noqa: E402
from oracle import flat  # noqa: E402


def timeit(fn, reps):
    """Median of `reps` calls after three warm-up calls: the first calls of a kernel variant in a process pay one-off costs
    (code-object load, LDS attribute, workspace growth — up to ~100 ms, and not always on the very first call)."""
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def main():
    n, d = 10_000_000, 512
    idx = native.FlatIndex(d)
    idx.reserve(n)
    t0 = time.perf_counter()
    idx.add_synthetic(n, 1234, normalize=True)
    print(json.dumps({"what": "generate + normalise 10M x 512 on device", "ms": round((time.perf_counter() - t0) * 1e3, 2)}),
          flush=True)
    q = flat.synth(8, d, 5678)
    flat.normalize_l2(q)
    row_bytes = d * 4

    dt = timeit(lambda: idx.search(q[0], 10), 30)
    print(json.dumps({"what": "full search k=10 (host API)", "ms": round(dt * 1e3, 3), "GBps": round(n * row_bytes / dt / 1e9, 1)}),
          flush=True)
    rs = np.random.RandomState(0)
    for frac in (0.5, 0.1, 0.01, 0.001):
        m = int(n * frac)
        rows = np.sort(rs.choice(n, m, replace=False)).astype(np.int64)
        dt = timeit(lambda: idx.search_subset(q[0], 10, rows), 10)
        # includes the H2D copy of the row list (8 B per row) — what find_most_similar's filtered branch pays
        print(json.dumps({"what": f"subset search, {frac:g} of the rows (sorted ids), k=10", "rows": m,
                          "ms": round(dt * 1e3, 3), "GBps_rows_touched": round(m * row_bytes / dt / 1e9, 1)}), flush=True)
    # a filter's rows four ways, host API: the row list per query (what round 2 had), the bitmap per query, and the
    # RESIDENT row set (uploaded once: list, bitmap, or "every row but these") — what the drop-in keeps per filter
    for frac in (0.99, 0.9, 0.5, 0.1):
        m = int(n * frac)
        rows = np.sort(rs.choice(n, m, replace=False)).astype(np.int64)
        out = {"what": f"filter keeping {frac:g} of the rows, k=10 (host API incl. PCIe)", "rows": m}
        out["row_list_per_query_ms"] = round(timeit(lambda: idx.search_subset(q[0], 10, rows), 10) * 1e3, 3)
        t0 = time.perf_counter()
        mask = native.pack_row_mask(n, rows=rows)
        out["pack_bitmap_on_host_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
        out["bitmap_per_query_ms"] = round(timeit(lambda: idx.search_masked(q[0], 10, mask), 10) * 1e3, 3)
        t0 = time.perf_counter()
        rsid = idx.rowset(rows)
        out["rowset_create_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
        out["rowset_is_bitmap"] = rsid.is_bitmap
        out["resident_rowset_ms"] = round(timeit(lambda: idx.search_rowset(q[0], 10, rsid), 20) * 1e3, 3)
        rsid.close()
        if frac >= 0.9:
            excl = np.setdiff1d(np.arange(n, dtype=np.int64), rows)
            t0 = time.perf_counter()
            rse = idx.rowset(excl, excluded=True)
            out["excluded_rowset_create_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
            out["excluded_rowset_ms"] = round(timeit(lambda: idx.search_rowset(q[0], 10, rse), 20) * 1e3, 3)
            rse.close()
        print(json.dumps(out), flush=True)
    # BATCHES under a filter (bitmap / resident bitmap row set): the certified fp16 pass and the fp32-MFMA pass take the bitmap
    qb = flat.synth(256, d, 91)
    flat.normalize_l2(qb)
    for frac in (0.5, 0.99):
        excl = np.sort(rs.choice(n, n - int(n * frac), replace=False)).astype(np.int64)
        rse = idx.rowset(excl, excluded=True)
        out = {"what": f"batches under a filter keeping {frac:g} of the rows (resident bitmap), k=10, host API"}
        for nb in (8, 32, 128, 256):
            dtb = timeit(lambda: idx.search_rowset(qb[:nb], 10, rse), 5)
            out[f"nq{nb}_ms"] = round(dtb * 1e3, 3)
            out[f"nq{nb}_qps"] = round(nb / dtb, 1)
        rse.close()
        print(json.dumps(out), flush=True)
    # BATCHES under a resident row LIST (a filter keeping 30 % / 5 % of the rows stays a list): round 4 — the list's bitmap twin lets
    # a batch share corpus passes where that beats one gathered scan per query
    for frac in (0.3, 0.05):
        keep = np.sort(rs.choice(n, int(n * frac), replace=False)).astype(np.int64)
        rsl = idx.rowset(keep)
        out = {"what": f"batches under a resident row LIST keeping {frac:g} of the rows, k=10, host API", "is_bitmap": bool(rsl.is_bitmap)}
        for nb in (1, 8, 64, 256):
            out[f"nq{nb}_ms"] = round(timeit(lambda: idx.search_rowset(qb[:nb], 10, rsl), 5) * 1e3, 3)
        os.environ["MVDB_DISABLE_MASKED_BATCH"] = "1"
        idx.reload_env()
        for nb in (8, 64):
            out[f"nq{nb}_ms_one_gathered_scan_per_query (MVDB_DISABLE_MASKED_BATCH=1)"] = round(
                timeit(lambda: idx.search_rowset(qb[:nb], 10, rsl), 3) * 1e3, 3)
        del os.environ["MVDB_DISABLE_MASKED_BATCH"]
        idx.reload_env()
        rsl.close()
        print(json.dumps(out), flush=True)
    rows = rs.permutation(n)[:n // 10].astype(np.int64)
    dt = timeit(lambda: idx.search_subset(q[0], 10, rows), 10)
    print(json.dumps({"what": "subset search, 0.1 of the rows (random order), k=10", "rows": len(rows),
                      "ms": round(dt * 1e3, 3), "GBps_rows_touched": round(len(rows) * row_bytes / dt / 1e9, 1)}), flush=True)
    for k in (64, 100, 1000, 10000):
        dt = timeit(lambda: idx.search(q[0], k), 10)
        print(json.dumps({"what": f"full search k={k}" + (" (fused select)" if k <= 64 else " (scores + radix select)"),
                          "ms": round(dt * 1e3, 3)}), flush=True)
    dt = timeit(lambda: idx.search(q, 10), 10)
    print(json.dumps({"what": "8 queries in one call, k=10 (MFMA pass)", "ms": round(dt * 1e3, 3)}), flush=True)
    t0 = time.perf_counter()
    idx.remove_rows(np.array([5, 5_000_000, 9_999_999], np.int64))
    print(json.dumps({"what": "remove 3 rows (compaction of the tail after row 5)", "ms": round((time.perf_counter() - t0) * 1e3, 2)}),
          flush=True)
    idx.close()

    n2 = 2_000_000
    x = flat.synth(n2, d, 7)
    idx = native.FlatIndex(d)
    t0 = time.perf_counter()
    idx.add(x, normalize=True)
    dt = time.perf_counter() - t0
    print(json.dumps({"what": "add 2M x 512 from host memory + normalise (PCIe)", "ms": round(dt * 1e3, 1),
                      "GBps": round(n2 * row_bytes / dt / 1e9, 2)}), flush=True)
    idx.close()
    idxl = native.FlatIndex(d, metric=native.METRIC_L2)
    idxl.reserve(n)
    idxl.add_synthetic(n, 1234, normalize=True)
    dt = timeit(lambda: idxl.search(q[0], 10), 20)
    print(json.dumps({"what": "L2 metric full search k=10", "ms": round(dt * 1e3, 3), "GBps": round(n * row_bytes / dt / 1e9, 1)}),
          flush=True)
    ql = flat.synth(256, d, 77)
    flat.normalize_l2(ql)
    out = {"what": "L2 metric, several queries per call, k=10 (up to 13: fp32-MFMA pass, |q|^2 + |x|^2 - 2 q.x; 14+ over rows "
                   "of one norm: the inner product's certified passes with an L2 re-score and a norm-range certificate)"}
    for nb in (8, 32, 128, 256):
        out[f"nq{nb}_ms"] = round(timeit(lambda: idxl.search(ql[:nb], 10), 5) * 1e3, 3)
    idxl.reload_env()
    os.environ["MVDB_DISABLE_L2_CERT"] = "1"
    idxl.reload_env()
    for nb in (32, 128):
        out[f"nq{nb}_ms_exact_fp32_kernels (MVDB_DISABLE_L2_CERT=1)"] = round(timeit(lambda: idxl.search(ql[:nb], 10), 3) * 1e3, 3)
    del os.environ["MVDB_DISABLE_L2_CERT"]
    print(json.dumps(out), flush=True)
    idxl.close()
    # rows of MIXED norms (un-normalised synthetic rows: |x|^2 spread of a few per cent, far beyond the 2^-10 of the norm-range
    # certificate): the shadow pass nominates by q.x - |x|^2 / 2 with per-row offsets (round 4)
    idxm = native.FlatIndex(d, metric=native.METRIC_L2)
    idxm.reserve(n)
    idxm.add_synthetic(n, 1234, normalize=False)
    qm = flat.synth(256, d, 78)
    out = {"what": "L2 metric over rows of mixed norms, several queries per call, k=10 (2+ queries: nomination over the fp16 shadow by "
                   "q.x - |x|^2 / 2, per-row offsets through the scalar cache, L2 re-score, certificate)"}
    reruns = native.split_rerun_count()
    for nb in (8, 32, 128, 256):
        out[f"nq{nb}_ms"] = round(timeit(lambda: idxm.search(qm[:nb], 10), 5) * 1e3, 3)
    out["chunks_rerun_on_exact_kernels"] = native.split_rerun_count() - reruns
    os.environ["MVDB_DISABLE_L2_CERT"] = "1"
    idxm.reload_env()
    for nb in (32, 128):
        out[f"nq{nb}_ms_exact_fp32_kernels (MVDB_DISABLE_L2_CERT=1)"] = round(timeit(lambda: idxm.search(qm[:nb], 10), 3) * 1e3, 3)
    del os.environ["MVDB_DISABLE_L2_CERT"]
    print(json.dumps(out), flush=True)
    idxm.close()


if __name__ == "__main__":
    main()
