"""ctypes binding of oracle/liboracle.so (the C restatement) plus numpy float64 helpers.

TEST INFRASTRUCTURE ONLY: the product (minivectordb_amd/) never imports this module.
PARITY UNPINNED at the faiss boundary — see the header of oracle/flat_oracle.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

METRIC_IP = 0
METRIC_L2 = 1


def build(force=False):
    """Compile liboracle.so with gcc (a few seconds)."""
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(
            os.path.join(_HERE, "flat_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        f32p = ctypes.POINTER(ctypes.c_float)
        f64p = ctypes.POINTER(ctypes.c_double)
        i64p = ctypes.POINTER(ctypes.c_int64)
        L.oracle_synth_fill.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, ctypes.c_uint64, ctypes.c_int64]
        L.oracle_synth_fill.restype = None
        L.oracle_normalize_l2.argtypes = [f32p, ctypes.c_int64, ctypes.c_int]
        L.oracle_normalize_l2.restype = None
        L.oracle_flat_search.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, f32p, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, i64p, ctypes.c_int, f32p, i64p]
        L.oracle_flat_search.restype = None
        L.oracle_flat_search_f64.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, f32p, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, i64p, ctypes.c_int, f64p, i64p]
        L.oracle_flat_search_f64.restype = None
        u8p = ctypes.POINTER(ctypes.c_uint8)
        L.oracle_flat_search_block.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, f32p, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_int64, u8p, ctypes.c_int, f32p, i64p]
        L.oracle_flat_search_block.restype = None
        L.oracle_merge_topk.argtypes = [f32p, i64p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, f32p, i64p]
        L.oracle_merge_topk.restype = None
        L.oracle_scores_f64.argtypes = [f32p, ctypes.c_int, f32p, ctypes.c_int, i64p, ctypes.c_int64, f64p]
        L.oracle_scores_f64.restype = None
        L.oracle_max_threads.restype = ctypes.c_int
        L.oracle_first_touch_copy.argtypes = [f32p, f32p, ctypes.c_int64, ctypes.c_int, ctypes.c_int]
        L.oracle_first_touch_copy.restype = None
        _LIB = L
    return _LIB


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i64(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))


# families of the synthetic stream (the seed's top byte; flat_oracle.c synth_value): OR one into a seed
SYNTH_POSITIVE = 1 << 56    # uniform [0, 1) rows — the reference's own test vectors (numpy.random.rand), a narrow cone once normalised
SYNTH_CLUSTERED = 2 << 56   # 4,096 shared centres + 6 % noise, exact and near duplicates: the certified passes' unfriendly case


def synth(n, d, seed, first_row=0):
    """Rows [first_row, first_row+n) of synthetic stream `seed` (un-normalised), float32 [n,d]."""
    out = np.empty((n, d), dtype=np.float32)
    lib().oracle_synth_fill(out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), n, d, seed, first_row)
    return out


def normalize_l2(x):
    """faiss.normalize_L2 restated: in place on a C-contiguous float32 [n,d] array."""
    assert x.dtype == np.float32 and x.flags["C_CONTIGUOUS"] and x.ndim == 2
    lib().oracle_normalize_l2(x.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), x.shape[0], x.shape[1])
    return x


def flat_search(x, q, k, metric=METRIC_IP, normalize_q=False, rows=None, nthreads=1, f64=False):
    """IndexFlat{IP,L2}.search restated.  Returns (D [nq,k], I [nq,k])."""
    x, xp = _f32(x)
    q = np.atleast_2d(np.asarray(q, dtype=np.float32))
    q, qp = _f32(q)
    nq, d = q.shape
    assert x.ndim == 2 and x.shape[1] == d
    if rows is not None:
        rows, rp = _i64(rows)
        n = rows.shape[0]
    else:
        rp = None
        n = x.shape[0]
    I = np.empty((nq, k), dtype=np.int64)
    Ip = I.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
    if f64:
        D = np.empty((nq, k), dtype=np.float64)
        lib().oracle_flat_search_f64(xp, n, d, qp, nq, k, metric, int(normalize_q), rp, nthreads,
                                     D.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), Ip)
    else:
        D = np.empty((nq, k), dtype=np.float32)
        lib().oracle_flat_search(xp, n, d, qp, nq, k, metric, int(normalize_q), rp, nthreads,
                                 D.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), Ip)
    return D, I


def flat_search_block(x, q, k, id_base=0, metric=METRIC_IP, keep=None, nthreads=None):
    """All queries against ONE block of rows (labels id_base + row): the arithmetic and (score, id) order of
    flat_search, looped tile-major so that many queries share each pass over the block.  keep: optional uint8[n], rows
    with 0 are outside the searched set.  Returns (D [nq,k], I [nq,k])."""
    x, xp = _f32(x)
    q, qp = _f32(np.atleast_2d(np.asarray(q, dtype=np.float32)))
    nq, d = q.shape
    assert x.ndim == 2 and x.shape[1] == d
    kp = None
    if keep is not None:
        keep = np.ascontiguousarray(keep, dtype=np.uint8)
        assert keep.shape == (x.shape[0],)
        kp = keep.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    lib().oracle_flat_search_block(xp, x.shape[0], d, qp, nq, k, metric, int(id_base), kp,
                                   int(nthreads or max_threads()), D.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                   I.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    return D, I


def merge_topk(parts, metric=METRIC_IP):
    """Top-k of the union of per-block results [(D [nq,k], I [nq,k]), ...] by the oracle's total order."""
    Dp = np.ascontiguousarray(np.stack([p[0] for p in parts]), dtype=np.float32)
    Ip = np.ascontiguousarray(np.stack([p[1] for p in parts]), dtype=np.int64)
    _, nq, k = Dp.shape
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    lib().oracle_merge_topk(Dp.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                            Ip.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), Dp.shape[0], nq, k, metric,
                            D.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                            I.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    return D, I


def scores_f64(x, q, rows, metric=METRIC_IP):
    """float64 scores (IP) / squared distances (L2) of the listed rows against one query."""
    x, xp = _f32(x)
    q, qp = _f32(np.asarray(q, dtype=np.float32).reshape(-1))
    rows, rp = _i64(rows)
    out = np.empty(rows.shape[0], dtype=np.float64)
    lib().oracle_scores_f64(xp, x.shape[1], qp, metric, rp, rows.shape[0],
                            out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return out


def first_touch_copy(x, nthreads):
    """Copy of x whose pages are first touched by the thread that scans them in flat_search(..., nthreads=nthreads)
    (NUMA placement for the multi-threaded CPU baseline)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)   # untouched pages
    lib().oracle_first_touch_copy(out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                  x.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), x.shape[0], x.shape[1], int(nthreads))
    return out


def first_touch_copy_block(dst, src, nthreads, row0):
    """src [m,d] -> rows [row0, row0+m) of dst [n,d] (np.empty: untouched pages), each row first touched by the thread that
    scans it in flat_search(dst, ..., nthreads=nthreads): a corpus streamed in blocks lands NUMA-placed."""
    assert dst.dtype == np.float32 and dst.flags["C_CONTIGUOUS"] and src.dtype == np.float32 and src.flags["C_CONTIGUOUS"]
    assert src.shape[1] == dst.shape[1] and row0 + src.shape[0] <= dst.shape[0]
    fp = ctypes.POINTER(ctypes.c_float)
    f = lib().oracle_first_touch_copy_block
    f.argtypes = [fp, fp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64]
    f.restype = None
    f(dst.ctypes.data_as(fp), src.ctypes.data_as(fp), dst.shape[0], dst.shape[1], int(nthreads), int(row0), src.shape[0])


def max_threads():
    return lib().oracle_max_threads()


def adjudicate(x, q, k, D_got, I_got, metric=METRIC_IP, rows=None, tol=1e-4, tie_eps=2e-6):
    """Compare one query's result with the float64 ground truth.

    Returns (ok, message).  ok requires: distances within `tol` of the float64 score of the
    returned row; every returned id either in the float64 top-k or a near-tie (its float64 score
    within `tie_eps` of the float64 k-th best); list sorted (non-increasing for IP).
    """
    x = np.ascontiguousarray(x, dtype=np.float32)
    q = np.asarray(q, dtype=np.float32).reshape(-1)
    D64, I64 = flat_search(x, q, k, metric=metric, rows=rows, f64=True)
    D64, I64 = D64[0], I64[0]
    I_got = np.asarray(I_got).reshape(-1)
    D_got = np.asarray(D_got).reshape(-1)
    valid = I_got >= 0
    if valid.sum() != (I64 >= 0).sum():
        return False, f"result count {valid.sum()} != {(I64 >= 0).sum()}"
    ids = I_got[valid]
    if len(set(ids.tolist())) != len(ids):
        return False, "duplicate ids in result"
    phys = ids if rows is None else np.asarray(rows, dtype=np.int64)[ids]
    true = scores_f64(x, q, phys, metric)
    err = np.abs(true - D_got[valid].astype(np.float64))
    if err.size and err.max() > tol:
        return False, f"distance error {err.max():.3e} > {tol}"
    sgn = 1.0 if metric == METRIC_IP else -1.0
    dd = sgn * D_got[valid].astype(np.float64)
    if np.any(np.diff(dd) > 0):
        return False, "result not sorted"
    if valid.sum():
        kth = D64[valid.sum() - 1]
        extra = [i for i in ids.tolist() if i not in set(I64.tolist())]
        for i in extra:
            p = i if rows is None else int(np.asarray(rows)[i])
            s = scores_f64(x, q, np.array([p]), metric)[0]
            if abs(s - kth) > tie_eps:
                return False, f"id {i} (score {s:.9f}) is not in the float64 top-{k} (k-th {kth:.9f})"
    return True, "ok"
