#!/usr/bin/env python3
"""Randomised check of the rescue launches' tile lists at their real sizes (1M rows and more; the certified pass keeps tile
flags by default): random (rows, width, family, queries per call, k, bitmap) cases, each searched with the tile lists and with
every tile scanned (MVDB_TILE_FLAGS=1 / 0) — results must be the same bits — and a few queries per case adjudicated against the
float64 oracle.  usage: fuzz_tile_flags.py SEED SECONDS"""
import os, sys, time
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import flat
from minivectordb_amd import _native as native

rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 120)
cases = fails = refused_cases = 0
listed0, total0 = native.rescue_tile_stats()
while time.time() < t_end:
    d = int(rs.choice([128, 256, 384, 512, 640, 1024]))
    n = int(rs.randint(1_050_000, 2_400_000 if d <= 512 else 1_300_000))
    fam = int(rs.choice([0, 1, 2, 2, 2])) << 56
    nq = int(rs.choice([2, 8, 31, 32, 33, 100, 128, 129, 256, 300]))
    k = int(rs.choice([1, 3, 10, 10, 12, 16]))
    masked = rs.rand() < 0.3
    idx = native.FlatIndex(d)
    idx.reserve(n)
    idx.add_synthetic(n, 1234 | fam, normalize=True)
    q = flat.synth(nq, d, int(rs.randint(1, 1 << 30)) | fam)
    flat.normalize_l2(q)
    mask = keep = None
    if masked:
        keep = rs.rand(n) < rs.uniform(0.2, 0.95)
        mask = native.pack_row_mask(n, rows=np.flatnonzero(keep))
    def run():
        return idx.search_masked(q, k, mask, labels="rows") if masked else idx.search(q, k)
    before = native.split_rerun_count()
    os.environ["MVDB_TILE_FLAGS"] = "1"          # always (the default keeps them only while the index has been refusing certificates)
    idx.reload_env()
    D1, I1 = run()
    refused = native.split_rerun_count() > before
    os.environ["MVDB_TILE_FLAGS"] = "0"
    idx.reload_env()
    D0, I0 = run()
    ok = D1.tobytes() == D0.tobytes() and I1.tobytes() == I0.tobytes()
    msg = "" if ok else "tile lists changed the result"
    if ok:
        stored = idx.get_rows(0, n)
        for i in rs.choice(nq, min(nq, 3), replace=False):
            if masked:
                sub = np.flatnonzero(keep)
                pos = np.searchsorted(sub, I1[i])          # row labels -> positions in the kept list
                good, m = flat.adjudicate(stored[keep], q[i], k, D1[i], np.where(I1[i] >= 0, pos, -1), tol=1e-4, tie_eps=4e-6)
            else:
                good, m = flat.adjudicate(stored, q[i], k, D1[i], I1[i], tol=1e-4, tie_eps=4e-6)
            if not good:
                ok, msg = False, f"query {i}: {m}"
                break
        del stored
    cases += 1
    refused_cases += int(refused)
    if not ok:
        fails += 1
        print("FAIL", dict(n=n, d=d, fam=fam >> 56, nq=nq, k=k, masked=masked), msg, flush=True)
    idx.close()
listed, total = native.rescue_tile_stats()
print("cases", cases, "fails", fails, "cases with refused certificates", refused_cases,
      "rescue tiles listed / total (both modes together)", listed - listed0, total - total0)
