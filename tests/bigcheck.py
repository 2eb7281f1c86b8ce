"""Full-size oracle comparison for the -m gpu suite ("1,000 queries per config, 100 at 80M").

A 10M x 512 corpus is 20 GB and lives only in HBM; the CPU oracle sees it in 1M-row blocks fetched with
``mvdb_index_get_rows`` (2 GB of host memory at a time).  Every block is scored against ALL queries by
``oracle_flat_search_block`` — the arithmetic and (score desc, id asc) order of ``oracle_flat_search``
(reference call site minivectordb/vector_database.py:497), labels = global row numbers — and the per-block lists are
merged by the same total order, so the result is exactly what one sequential scan of the whole corpus returns.

``compare`` holds a device result against it: distances within 1e-4 position by position, ids identical; where a
query's id list differs, the rows of BOTH lists are fetched back and scored in float64 — every id the device returned
that the float64 ranking of that union does not put in the top-k must sit within 2e-6 of the float64 k-th score (a
near-tie: two fp32 summation orders may rank such rows either way).  The number of such queries is reported.
"""
import json
import os
import time

import numpy as np

from oracle import flat

TOL = 1e-4
TIE_EPS = 2e-6
BLOCK = 1_000_000


def oracle_topk_streamed(idx, n, q, k, keeps=(None,), metric=flat.METRIC_IP, block=BLOCK):
    """Top-k of rows [0, n) of the device index `idx` for every query, by the CPU oracle, streaming the corpus through one
    host buffer.  keeps: one entry per wanted result — None (all rows) or a uint8[n] row selection.  Returns a list of
    (D, I) in the order of `keeps`, plus the seconds spent fetching / scoring."""
    d = idx.d
    buf = np.empty((min(block, n), d), dtype=np.float32)
    parts = [[] for _ in keeps]
    t_fetch = t_score = 0.0
    for b in range(0, n, block):
        m = min(block, n - b)
        t0 = time.time()
        idx.get_rows(b, m, out=buf[:m])
        t1 = time.time()
        for j, keep in enumerate(keeps):
            parts[j].append(flat.flat_search_block(buf[:m], q, k, id_base=b, metric=metric,
                                                   keep=None if keep is None else keep[b:b + m]))
        t_fetch += t1 - t0
        t_score += time.time() - t1
    return [flat.merge_topk(p, metric=metric) for p in parts], {"fetch_s": round(t_fetch, 2), "score_s": round(t_score, 2)}


def compare(idx, q, D, I, Do, Io, what, metric=flat.METRIC_IP, tol=TOL, tie_eps=TIE_EPS):
    """Assert the device result (D, I) against the oracle's (Do, Io) for the same queries; returns the statistics."""
    assert D.shape == Do.shape and I.shape == Io.shape, (what, D.shape, Do.shape)
    assert np.array_equal(I >= 0, Io >= 0), f"{what}: result counts differ"
    valid = Io >= 0
    err = float(np.abs(D[valid].astype(np.float64) - Do[valid].astype(np.float64)).max()) if valid.any() else 0.0
    assert err <= tol, f"{what}: distances differ from the oracle's by {err:.3e} > {tol}"
    sgn = 1.0 if metric == flat.METRIC_IP else -1.0
    differ = [i for i in range(q.shape[0]) if not np.array_equal(I[i], Io[i])]
    near_tie_queries = 0
    for i in differ:
        got, want = I[i][I[i] >= 0], Io[i][Io[i] >= 0]
        assert len(set(got.tolist())) == len(got), f"{what}: query {i} returns a row twice"
        union = np.array(sorted(set(got.tolist()) | set(want.tolist())), dtype=np.int64)
        rows = np.stack([idx.get_rows(int(r), 1)[0] for r in union]).astype(np.float64)
        q64 = q[i].astype(np.float64)
        s64 = rows @ q64 if metric == flat.METRIC_IP else -((rows - q64) ** 2).sum(axis=1)
        score_of = dict(zip(union.tolist(), s64.tolist()))
        order = sorted(union.tolist(), key=lambda r: (-score_of[r], r))
        top = set(order[:len(got)])
        kth = score_of[order[len(got) - 1]]
        for r in got.tolist():
            if r not in top:
                assert abs(score_of[r] - kth) <= tie_eps, (
                    f"{what}: query {i} returns row {r} (float64 score {sgn * score_of[r]:.9f}) which is not a near-tie of "
                    f"the k-th best ({sgn * kth:.9f}); oracle ids {want.tolist()}, device ids {got.tolist()}")
        # the device's own order must also be an order of ITS scores (checked above through D vs Do) and of the
        # float64 scores up to near-ties
        g64 = np.array([score_of[r] for r in got.tolist()])
        assert np.all(np.diff(g64) <= tie_eps), f"{what}: query {i} is not sorted by the float64 scores"
        near_tie_queries += 1
    return {"what": what, "queries": int(q.shape[0]), "k": int(D.shape[1]), "queries_with_id_differences": len(differ),
            "adjudicated_near_ties": near_tie_queries, "max_distance_error_vs_oracle": err}


def report(record):
    """One JSON line per comparison: printed (pytest -s / -rA) and appended to gpurun_out/ when that directory exists."""
    line = json.dumps(record)
    print("[fullsize parity]", line)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "fullsize_parity.jsonl"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
