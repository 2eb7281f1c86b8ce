"""Randomised GPU fuzz of the WRITE side of an index (round 4): random adds, deletes (single early rows, scattered sets,
whole prefixes / suffixes) and searches interleaved on one index, against a numpy mirror of the rows (np.delete /
np.concatenate) and the CPU oracle on that mirror.  Exercises the in-place chunked compaction (MVDB_COMPACT_BYTES drawn from one
row .. the default), the fp16 shadow that follows adds and is dropped by deletes, resident row sets across appends, the
stale-row-set refusal after a delete, and get_rows byte equality after every mutation.
usage: fuzz_mutations.py SEED SECONDS   (on a GPU box)"""
import os, sys, time
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import flat
from minivectordb_amd import _native as native

rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 120)
indexes = searches = mutations = fails = 0
threads = min(64, os.cpu_count() or 8)


def check_search(idx, mirror, metric, label):
    global searches, fails
    n, d = mirror.shape
    nq = int(rs.choice([1, 1, 2, 8, 14, 40, 130]))
    k = int(rs.choice([1, 5, 10, 12]))
    q = flat.synth(nq, d, rs.randint(1 << 30))
    flat.normalize_l2(q)
    rows = None
    if n > 4 and rs.rand() < 0.3:     # through a resident row set built NOW (valid until the next delete)
        sel = rs.rand(n) < float(rs.choice([0.05, 0.5, 0.95]))
        sel[rs.randint(n)] = True
        rows = np.flatnonzero(sel).astype(np.int64)
        rset = idx.rowset(rows)
        D, I = idx.search_rowset(q, k, rset)
        rset.close()
        I = np.where(I >= 0, np.searchsorted(rows, np.maximum(I, 0)), -1)
    else:
        D, I = idx.search(q, k)
    Do, Io = flat.flat_search(mirror, q, k, metric=metric, rows=rows, nthreads=threads)
    searches += 1
    for i in range(nq):
        if np.array_equal(I[i], Io[i]) and np.abs(D[i] - Do[i]).max() <= 2e-6:
            continue
        ok, msg = flat.adjudicate(mirror, q[i], k, D[i], I[i], metric=metric, rows=rows, tol=1e-4, tie_eps=2e-6)
        if not ok:
            fails += 1
            print("FAIL search", label, dict(n=n, d=d, nq=nq, k=k, metric=metric, subset=None if rows is None else len(rows)), i, msg,
                  flush=True)
            return


while time.time() < t_end:
    d = int(rs.choice([32, 64, 100, 256, 256, 512]))
    big = d in (256, 512) and rs.rand() < 0.5      # large enough for the fp16 shadow (>= 100k rows)
    n0 = int(rs.randint(100_000, 160_000)) if big else int(rs.choice([1, 2, 50, 1000, rs.randint(1, 30000)]))
    metric = int(rs.choice([0, 0, 1]))
    row_bytes = 4 * ((d + 3) // 4 * 4)
    staging = rs.choice([None, row_bytes, 7 * row_bytes, 1 << 16, 1 << 20])
    if staging is None:
        os.environ.pop("MVDB_COMPACT_BYTES", None)
    else:
        os.environ["MVDB_COMPACT_BYTES"] = str(int(staging))
    idx = native.FlatIndex(d, metric=metric)
    if rs.rand() < 0.3:
        idx.set_option("shadow_single_query", 1)
    mirror = flat.synth(n0, d, rs.randint(1 << 30))
    flat.normalize_l2(mirror)
    idx.add(mirror)
    indexes += 1
    label = dict(d=d, n0=n0, metric=metric, staging=None if staging is None else int(staging))
    stale = None
    for _ in range(int(rs.randint(4, 10))):
        op = rs.choice(["add", "del_one_early", "del_scatter", "del_range", "search", "search"])
        n = mirror.shape[0]
        if op == "add":
            m = int(rs.choice([1, 3, 100, rs.randint(1, 5000)]))
            x = flat.synth(m, d, rs.randint(1 << 30))
            flat.normalize_l2(x)
            if stale is None and n > 8 and rs.rand() < 0.5:   # a row set built before an append stays valid
                keep = np.flatnonzero(rs.rand(n) < 0.5).astype(np.int64)
                if len(keep):
                    stale = (idx.rowset(keep), keep)
            idx.add(x)
            mirror = np.concatenate([mirror, x])
            mutations += 1
            if stale is not None:
                rset, keep = stale
                q = flat.synth(3, d, rs.randint(1 << 30))
                flat.normalize_l2(q)
                D, I = idx.search_rowset(q, 5, rset)
                Do, Io = flat.flat_search(mirror, q, 5, metric=metric, rows=keep, nthreads=threads)
                got = np.where(I >= 0, np.searchsorted(keep, np.maximum(I, 0)), -1)
                if not np.array_equal(got, Io):
                    okall = all(flat.adjudicate(mirror, q[i], 5, D[i], got[i], metric=metric, rows=keep, tol=1e-4, tie_eps=2e-6)[0]
                                for i in range(3))
                    if not okall:
                        fails += 1
                        print("FAIL row set across an append", label, flush=True)
        elif op.startswith("del") and n > 1:
            if op == "del_one_early":
                dels = np.array([rs.randint(0, min(n, 10))])
            elif op == "del_scatter":
                dels = rs.permutation(n)[:int(rs.randint(1, max(2, min(n - 1, 3000))))]
            else:
                a = int(rs.choice([0, rs.randint(0, n)]))
                b = min(n, a + int(rs.randint(1, max(2, n // 3))))
                if b - a >= n:
                    b = n - 1
                dels = np.arange(a, b)
            if len(dels) == 0 or len(dels) >= n:
                continue
            idx.remove_rows(dels.astype(np.int64))
            mirror = np.delete(mirror, dels, 0)
            mutations += 1
            if stale is not None:     # rows were renumbered: the old row set must be refused, not silently used
                rset, keep = stale
                try:
                    idx.search_rowset(mirror[:1].copy(), 1, rset)
                    fails += 1
                    print("FAIL stale row set accepted after a delete", label, flush=True)
                except (ValueError, RuntimeError):
                    pass
                rset.close()
                stale = None
            if idx.shadow_rows not in (-1, 0):
                fails += 1
                print("FAIL shadow kept across a delete", label, idx.shadow_rows, flush=True)
        else:
            check_search(idx, mirror, metric, label)
            continue
        n = mirror.shape[0]
        if idx.ntotal != n:
            fails += 1
            print("FAIL ntotal", label, idx.ntotal, n, flush=True)
            break
        # bytes of a few windows of rows after every mutation (whole matrix when small)
        if n <= 20000:
            same = idx.get_rows(0, n).tobytes() == mirror.tobytes()
        else:
            same = True
            for r0 in [0, n - 64] + [int(v) for v in rs.randint(0, n - 64, size=6)]:
                same = same and idx.get_rows(r0, 64).tobytes() == mirror[r0:r0 + 64].tobytes()
        if not same:
            fails += 1
            print("FAIL rows differ from the numpy mirror after", op, label, flush=True)
            break
        if rs.rand() < 0.6:
            check_search(idx, mirror, metric, label)
    if stale is not None:
        stale[0].close()
    idx.close()
print("indexes", indexes, "mutations", mutations, "searches", searches, "fails", fails)
