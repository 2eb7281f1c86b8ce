"""Randomised GPU parity fuzz: random (n, d, nq, k, metric, subset, normalisation) cases through the C-ABI,
each query adjudicated against the float64 oracle.  usage: fuzz_parity.py SEED SECONDS [split | masked] (on a GPU box);
`split` biases the cases towards the certified batch passes (nq >= 40, k <= 32, d % 32 == 0, up to 400k rows so
that the seed launch runs too) and reports how many chunks fell back to the exact kernels; `masked` is `split` under a
random BITMAP (density 0.02 .. 0.99, any nq >= 2: the fp16 pass, the fp32-MFMA pass and the one-query scan all take it),
half of the cases through a resident row set; `shadow` (round 4) is `split` over LARGER corpora (100k - 620k rows, d in
{256, 384, 512, 768, 1024}) with small and large batches (2 .. 300 queries) and both metrics over normalised rows: the
fp16-shadow nomination pass from 2 / 8 queries on, the L2 certificate, the device-gated L2 re-run; only queries whose ids or
distances differ from the multi-threaded fp32 oracle are adjudicated in float64."""
import os, sys, time
# the float64 adjudication is thousands of SMALL numpy / OpenMP products: with one thread per core of a 256-thread host each
# costs ~0.1 s (tests/conftest.py has the same bound); the big fp32 oracle scans below name their thread count explicitly
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import flat
from minivectordb_amd import _native as native
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
dims = [1,2,3,4,5,7,8,16,31,32,33,48,64,96,100,128,160,192,224,256,300,384,512,640,768,1000,1024,1100,2048,4096]
t_end = time.time() + float(sys.argv[2]) if len(sys.argv) > 2 else time.time() + 120
masked_mode = len(sys.argv) > 3 and sys.argv[3] == "masked"
shadow_mode = len(sys.argv) > 3 and sys.argv[3] == "shadow"
split_mode = masked_mode or shadow_mode or (len(sys.argv) > 3 and sys.argv[3] == "split")
cases = fails = 0
reruns0 = native.split_rerun_count()
while time.time() < t_end:
    d = int(rs.choice(dims))
    n = int(rs.choice([1,2,15,16,17,63,64,65,255,256,257,1000,4097, rs.randint(1, 5000)]))
    nq = int(rs.choice([1,1,1,2,3,5,15,16,17,31,32,33,40,104,128,150,260]))
    k = int(rs.choice([1,2,5,10,63,64,65,100,300, rs.randint(1, 200)]))
    metric = int(rs.choice([0,0,0,1]))
    if split_mode:
        d = int(rs.choice([32, 64, 96, 128, 256, 384, 512, 640, 768, 896, 1024]))
        n = int(rs.choice([15, 16, 17, 127, 128, 129, 1000, rs.randint(1, 20000), rs.randint(262144, 400000)]))
        nq = int(rs.choice([40, 41, 64, 127, 128, 129, 200, 256, rs.randint(40, 300)]))
        k = int(rs.randint(1, 33))
        metric = 0
        if masked_mode:
            nq = int(rs.choice([2, 8, 31, 32, 33, nq, nq]))
            metric = int(rs.choice([0, 0, 0, 1]))   # round 4: L2 batches under a bitmap go through the fp16 nomination pass too
        if shadow_mode:
            d = int(rs.choice([256, 384, 512, 512, 768, 1024]))
            n = int(rs.choice([rs.randint(100_000, 130_000), rs.randint(500_000, 620_000), rs.randint(20_000, 60_000)]))
            nq = int(rs.choice([1, 1, 2, 3, 8, 9, 13, 14, 24, 32, 33, 100, 128, 129, 256, rs.randint(2, 300)]))
            metric = int(rs.choice([0, 0, 1]))
    if n * d > (330_000_000 if shadow_mode else 30_000_000): n = (330_000_000 if shadow_mode else 30_000_000) // d
    x = flat.synth(n, d, rs.randint(1<<30)); 
    if shadow_mode or rs.rand() < 0.7: flat.normalize_l2(x)
    if shadow_mode and rs.rand() < 0.5:   # rows of mixed norms (a factor of up to 100 apart): the L2 pass over the shadow nominates with per-row offsets
        x *= np.exp(rs.uniform(np.log(0.1), np.log(10.0), size=(n, 1))).astype(np.float32)
    if rs.rand() < 0.2 and n > 4: x[rs.randint(n)] = x[rs.randint(n)]   # duplicate row -> exact tie
    q = flat.synth(nq, d, rs.randint(1<<30))
    if shadow_mode and rs.rand() < 0.3: q *= np.exp(rs.uniform(np.log(0.02), np.log(5.0), size=(nq, 1))).astype(np.float32)
    normq = bool(rs.rand() < 0.5)
    idx = native.FlatIndex(d, metric=metric)
    if shadow_mode and rs.rand() < 0.5: idx.set_option("shadow_single_query", 1)   # single queries through the certified pass too
    idx.add(x)
    subset = None
    if masked_mode and n > 1:
        dens = float(rs.choice([0.02, 0.3, 0.5, 0.9, 0.99]))
        sel = rs.rand(n) < dens
        sel[rs.randint(n)] = True
        subset = np.flatnonzero(sel).astype(np.int64)
        if rs.rand() < 0.5:
            D, I = idx.search_masked(q, k, native.pack_row_mask(n, rows=subset), normalize_q=normq, labels="positions")
        else:
            rset = idx.rowset(np.flatnonzero(~sel).astype(np.int64), excluded=True)
            D, I = idx.search_rowset(q, k, rset, normalize_q=normq)
            rset.close()
            I = np.where(I >= 0, np.searchsorted(subset, np.maximum(I, 0)), -1)   # row numbers -> positions in the selection
    elif rs.rand() < 0.3 and n > 1:
        m = rs.randint(1, n+1); subset = rs.permutation(n)[:m].astype(np.int64)
        D, I = idx.search_subset(q, k, subset, normalize_q=normq)
    else:
        D, I = idx.search(q, k, normalize_q=normq)
    qq = q.copy()
    if normq: flat.normalize_l2(qq)
    Do, Io = flat.flat_search(x, qq, k, metric=metric, rows=subset, nthreads=min(64, os.cpu_count() or 8) if shadow_mode else 1)
    cases += 1
    bad = None
    for i in range(nq):
        if shadow_mode and np.array_equal(I[i], Io[i]) and np.abs(D[i] - Do[i]).max() <= 2e-6:
            continue   # id for id the fp32 oracle's answer: nothing for float64 to adjudicate
        # un-normalised data: tolerances scale with the magnitude of the scores (an fp32 ulp at 90 is 7.6e-6)
        mag = max(1.0, float(np.abs(Do[i][Io[i]>=0]).max()) if (Io[i]>=0).any() else 1.0)
        ok, msg = flat.adjudicate(x, qq[i], k, D[i], I[i], metric=metric, rows=subset, tol=1e-4 * mag, tie_eps=2e-6 * mag)
        if not ok: bad = (i, msg); break
    if bad:
        fails += 1
        print("FAIL", dict(n=n, d=d, nq=nq, k=k, metric=metric, normq=normq, subset=None if subset is None else len(subset)), bad, flush=True)
    idx.close()
print("cases", cases, "fails", fails, "split chunks re-run on the exact kernels", native.split_rerun_count() - reruns0)
