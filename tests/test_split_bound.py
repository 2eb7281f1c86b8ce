"""The error bound the certified batch pass certifies with (minivectordb_amd/csrc/half_scan.hip: half_eps, exported as
mvdb_half_eps): ONE product of the fp16 images of the scaled query and row nominates, so

    |a(x) - q.x| <= half_eps(d) |q| max|x|     (1.04e-3 at d = 512)

covers the rounding of both operands to fp16, elements below fp16's normal range, the WORST-CASE fp32 accumulation of the d
products in any order, the fp32 re-score, |q| and the comparison's rounding.  This file restates the formula, pins the
library's value to it, and checks it against emulations of the matrix cores' accumulation (sequential fp32 chains, rounded to
nearest and truncated; fp16 subnormals kept and flushed).  (Until round 6 it also held the bound of the retired bf16-split
generation, scan_split_kernels.hpp.)
"""
import numpy as np
import pytest


def _fp32_chain(terms, mode):
    """Sequential fp32 accumulation of float64 `terms` [..., n] along the last axis.  mode 'rne': every addition
    rounded to nearest (IEEE); 'trunc': every addition truncated toward zero at fp32 width (the pessimistic model of
    a matrix-core adder)."""
    acc = np.zeros(terms.shape[:-1], dtype=np.float32)
    for i in range(terms.shape[-1]):
        exact = acc.astype(np.float64) + terms[..., i]
        r = exact.astype(np.float32)
        if mode == "trunc":
            over = np.abs(r.astype(np.float64)) > np.abs(exact)
            r = np.where(over, np.nextafter(r, np.float32(0)), r).astype(np.float32)
        acc = r
    return acc



# ---- the fp16 single-product nomination pass (half_scan.hip) ----------------------------------------------------------
def half_eps(d):
    """Restatement of half_scan.hip: half_eps (per unit |q| * max|x|)."""
    u11, u23, u24 = 2.0 ** -11, 2.0 ** -23, 2.0 ** -24
    e_op = 2 * u11 + u11 * u11
    e_uf = np.sqrt(float(d)) * 2.0 ** -27 * (1 + u11) + d * 2.0 ** -56
    n = d + 4.0
    e_acc = n * u23 / (1 - n * u23) * (1 + u11) ** 2
    depth = ((d + 3) // 4 + 63) // 64 * 4 + 6
    e_re = depth * u24 / (1 - depth * u24)
    return (e_op + e_uf + e_acc + e_re) * (1 + 4e-6) + 4 * u24


@pytest.mark.parametrize("d", [128, 256, 384, 512, 1024])
def test_library_half_eps_is_the_documented_formula(d):
    from minivectordb_amd import _native
    got = _native.half_eps(d)
    assert got == pytest.approx(half_eps(d), rel=1e-12)
    assert got > 2.0 ** -10 + d * 2.0 ** -23          # operand rounding + accumulation, nothing fitted away


def _pow2_scale(bound):
    """s = 2^(15 - e), bound = m 2^e with m in [0.5, 1): half_scan.hip's half_xscale / half_queries_kernel."""
    _, e = np.frexp(bound)
    return np.ldexp(1.0, 15 - e)


def _fp16_image(v, scale, flush):
    """fp32 -> fp16 (RNE) of scale * v, as float64; flush = subnormal results become zero (the pessimistic model of a
    matrix core that does not read fp16 subnormals)."""
    s = (v.astype(np.float32) * np.float32(scale)).astype(np.float32)      # exact: power of two
    h = s.astype(np.float16)
    assert np.all(np.isfinite(h))
    h64 = h.astype(np.float64)
    if flush:
        h64 = np.where(np.abs(h64) < 2.0 ** -14, 0.0, h64)
    return h64


def _half_cases(d):
    rs = np.random.RandomState(500 + d)
    unit = lambda a: (a / np.linalg.norm(a.astype(np.float64), axis=-1, keepdims=True)).astype(np.float32)
    g = unit(rs.randn(24, d))
    yield "gaussian unit", unit(rs.randn(24, d)), g
    yield "all positive", unit(rs.rand(24, d) + 0.5), unit(rs.rand(24, d) + 0.5)
    yield "parallel", g, g
    m = np.float32(1.0) + np.float32(2.0 ** -11) * (1 - 2.0 ** -10)     # just below an fp16 rounding midpoint
    yield "fp16 midpoints", unit(np.full((2, d), m, np.float32)), unit(np.full((2, d), m, np.float32))
    wide = rs.randn(24, d) * 10.0 ** rs.randint(-9, 1, (24, d))           # many elements below 2^-29 of the largest
    yield "wide dynamic range (underflow)", unit(wide), unit(rs.randn(24, d) * 10.0 ** rs.randint(-9, 1, (24, d)))
    one_hot = np.zeros((4, d), np.float32)
    one_hot[np.arange(4), rs.randint(0, d, 4)] = 1.0
    yield "one-hot query against tiny elements", one_hot, unit(rs.randn(4, d) * 10.0 ** rs.randint(-9, 1, (4, d)))
    yield "raw rows and queries (norms 30 and 7)", (g * np.float32(7.0)).astype(np.float32), \
        (unit(rs.randn(24, d)) * np.float32(30.0)).astype(np.float32)


@pytest.mark.parametrize("d", [256, 512])
@pytest.mark.parametrize("mode", ["rne", "trunc"])
@pytest.mark.parametrize("flush", [False, True])
def test_half_eps_covers_emulated_nomination(d, mode, flush):
    """a(x) = fp32 chain over the d products of the two fp16 images (scaled by powers of two as the kernels do), plus
    the fp32 re-score, stay within half_eps(d) |q| max|x| of the real-number score — nearest and truncating adders,
    fp16 subnormals kept and flushed."""
    eps = half_eps(d)
    worst = 0.0
    for name, q, x in _half_cases(d):
        xnorm = np.linalg.norm(x.astype(np.float64), axis=-1)
        bound = np.float32(xnorm.max() * (1 + 4e-6))                       # the index's row-norm bound
        sx = _pow2_scale(bound)
        sq = np.array([_pow2_scale(np.abs(r).max()) if np.abs(r).max() > 0 else 1.0 for r in q])[:, None]
        qh = np.stack([_fp16_image(q[i], sq[i, 0], flush) for i in range(q.shape[0])])
        xh = _fp16_image(x, sx, flush)
        assert np.abs(qh).max() < 2.0 ** 15 + 1 and np.abs(xh).max() < 2.0 ** 15 + 1
        t = np.sum(q.astype(np.float64) * x.astype(np.float64), axis=-1)
        approx = _fp32_chain(qh * xh, mode).astype(np.float64) / (sq[:, 0] * sx)
        scale = np.linalg.norm(q.astype(np.float64), axis=-1) * float(bound)
        e_nom = np.abs(approx - t) / scale
        resc = _fp32_chain(q.astype(np.float64) * x.astype(np.float64), "rne").astype(np.float64)
        e_re = np.abs(resc - t) / scale
        assert np.all(e_nom + e_re <= eps), (name, mode, flush, float((e_nom + e_re).max()), eps)
        worst = max(worst, float((e_nom + e_re).max()))
    assert 0 < worst < eps


def test_half_pass_widths_by_dimension():
    """Host-only query of the pass table: 256 queries per corpus pass up to d = 512, 128 at the wider
    dimensions (one wave per SIMD holds the query fragments), 0 where the exact fp32 passes serve."""
    from minivectordb_amd import _native
    for d in (128, 256, 384, 512):   # d = 128: round 6 (the shadow kernel's 16 slots per row fit KT = 8)
        assert _native.half_max_queries(d) == 256
    for d in (640, 768, 896, 1024):   # round 3: every even multiple of 64 from 256 to 1024
        assert _native.half_max_queries(d) == 128
    for d in (32, 64, 100, 192, 320, 576, 1536, 2048):
        assert _native.half_max_queries(d) == 0
