"""The operand-error bound the split-precision batch pass certifies with (minivectordb_amd/csrc/scan_split_kernels.hpp):

    |q.x - (qh.xh + qh.xl + ql.xh)| <= 3 * 2^-16 * sum|q_i x_i| <= 4.6e-5 * |q| * |x|

with (h, l) the bf16 round-to-nearest-even split of an fp32 value.  Checked here in numpy (bf16 emulated bit-exactly,
sums in float64 so that only the operand error is measured) on random and adversarial inputs.

The certificate itself uses eps(d) (mvdb.hip: split_eps, exported as mvdb_split_eps): operand term + the WORST-CASE
fp32 accumulation of the 3 d products in any order + the fp32 re-score + |q| and comparison rounding.  The second
half of this file restates that formula, pins the library's value to it, and checks it against emulations of the
matrix cores' accumulation (sequential fp32 chains in the kernel's product order, rounded to nearest and truncated).
"""
import numpy as np
import pytest


def bf16_rne(x):
    """fp32 -> nearest bf16 (ties to even), returned as fp32."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


def split(x):
    h = bf16_rne(x)
    l = bf16_rne((x.astype(np.float32) - h).astype(np.float32))
    return h, l


def cases():
    rs = np.random.RandomState(7)
    d = 512
    yield "gaussian", rs.randn(64, d).astype(np.float32), rs.randn(64, d).astype(np.float32)
    yield "all positive", rs.rand(64, d).astype(np.float32), rs.rand(64, d).astype(np.float32)
    # worst rounding: mantissas just below / above the bf16 midpoints, same sign everywhere
    m = np.float32(1.0) + np.float32(2.0 ** -8) * (1 - 2.0 ** -10)
    yield "midpoints", np.full((4, d), m, np.float32), np.full((4, d), m, np.float32)
    yield "wide dynamic range", (rs.randn(64, d) * 10.0 ** rs.randint(-6, 6, (64, d))).astype(np.float32), \
        (rs.randn(64, d) * 10.0 ** rs.randint(-6, 6, (64, d))).astype(np.float32)
    yield "sparse", (rs.randn(64, d) * (rs.rand(64, d) < 0.02)).astype(np.float32), rs.randn(64, d).astype(np.float32)


@pytest.mark.parametrize("name,q,x", list(cases()), ids=[c[0] for c in cases()])
def test_split_operand_error_bound(name, q, x):
    qh, ql = split(q)
    xh, xl = split(x)
    # the representation residuals themselves: 16 significant bits
    for v, h, l in ((q, qh, ql), (x, xh, xl)):
        res = np.abs(v.astype(np.float64) - h.astype(np.float64) - l.astype(np.float64))
        assert np.all(res <= 2.0 ** -16 * np.abs(v.astype(np.float64)) + 1e-300)
    f = lambda a: a.astype(np.float64)
    exact = np.einsum("id,jd->ij", f(q), f(x))
    approx = np.einsum("id,jd->ij", f(qh), f(xh)) + np.einsum("id,jd->ij", f(qh), f(xl)) + \
        np.einsum("id,jd->ij", f(ql), f(xh))
    l1 = np.einsum("id,jd->ij", np.abs(f(q)), np.abs(f(x)))
    err = np.abs(exact - approx)
    assert np.all(err <= 3 * 2.0 ** -16 * l1 + 1e-300), (name, float((err / np.maximum(l1, 1e-300)).max()))
    # Cauchy-Schwarz: what the certificate uses
    norms = np.linalg.norm(f(q), axis=1)[:, None] * np.linalg.norm(f(x), axis=1)[None, :]
    assert np.all(err <= 4.6e-5 * norms + 1e-300)


def test_bf16_rne_matches_torch():
    torch = pytest.importorskip("torch")
    rs = np.random.RandomState(3)
    x = (rs.randn(10000) * 10.0 ** rs.randint(-10, 10, 10000)).astype(np.float32)
    want = torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()
    assert np.array_equal(bf16_rne(x), want)


# ---- eps(d): the whole certificate bound -----------------------------------------------------------------------------
def split_eps(d):
    """Restatement of mvdb.hip: split_eps (per unit |q| * max|x|)."""
    u8, u16, u23, u24 = 2.0 ** -8, 2.0 ** -16, 2.0 ** -23, 2.0 ** -24
    e_op = u16 * ((1 + u16) + (1 + u8) ** 2 + 1)
    n = 3.0 * d
    e_acc = n * u23 / (1 - n * u23) * (1 + u8) ** 2 * (1 + 2 * u8)
    depth = ((d + 3) // 4 + 63) // 64 * 4 + 6
    e_re = depth * u24 / (1 - depth * u24)
    return (e_op + e_acc + e_re) * (1 + 4e-6) + 4 * u24


@pytest.mark.parametrize("d", [32, 64, 128, 256, 384, 512, 768, 1024, 4096])
def test_library_eps_is_the_documented_formula(d):
    from minivectordb_amd import _native
    got = _native.split_eps(d)          # host-only entry point: no device needed
    assert got == pytest.approx(split_eps(d), rel=1e-12)
    # the operand term alone (4.6e-5) is not enough: the bound must grow with d
    assert got > 3 * 2.0 ** -16 + 3 * d * 2.0 ** -24
    if d >= 64:
        assert _native.split_eps(d) > _native.split_eps(d // 2)


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


def _kernel_order_terms(q, x):
    """The 3 d products in the order flat_scan_split*_kernel issues them: per 16-element block, xl.qh, xh.ql,
    xh.qh (scan_split_kernels.hpp: mfma_step); within an instruction the elements are taken in k order."""
    qh, ql = split(q)
    xh, xl = split(x)
    f = lambda a: a.astype(np.float64)
    d = q.shape[-1]
    out = []
    for b in range(0, d, 16):
        sl = slice(b, b + 16)
        out += [f(xl[..., sl]) * f(qh[..., sl]), f(xh[..., sl]) * f(ql[..., sl]), f(xh[..., sl]) * f(qh[..., sl])]
    return np.concatenate(out, axis=-1)


def _bound_cases(d):
    rs = np.random.RandomState(100 + d)
    unit = lambda a: (a / np.linalg.norm(a.astype(np.float64), axis=-1, keepdims=True)).astype(np.float32)
    g = unit(rs.randn(24, d))
    yield "gaussian unit", unit(rs.randn(24, d)), g
    p = unit(rs.rand(24, d) + 0.5)
    yield "all positive (every addition rounds the same way)", unit(rs.rand(24, d) + 0.5), p
    yield "parallel (|q.x| = |q||x|)", g, g
    m = np.float32(1.0) + np.float32(2.0 ** -8) * (1 - 2.0 ** -10)
    yield "bf16 midpoints", unit(np.full((2, d), m, np.float32)), unit(np.full((2, d), m, np.float32))
    big_first = unit(np.sort(rs.rand(8, d).astype(np.float32) ** 4, axis=-1)[:, ::-1].copy())
    yield "decaying magnitudes", big_first, big_first


@pytest.mark.parametrize("d", [512, 1024])
@pytest.mark.parametrize("mode", ["rne", "trunc"])
def test_eps_covers_emulated_accumulation(d, mode):
    """approx (fp32 chain over the kernel's 3 d products) and re-score (fp32 chain over the d exact products) both
    stay within their share of eps(d) of the real-number score — for nearest and for truncating adders."""
    eps = split_eps(d)
    worst = 0.0
    for name, q, x in _bound_cases(d):
        t = np.sum(q.astype(np.float64) * x.astype(np.float64), axis=-1)
        approx = _fp32_chain(_kernel_order_terms(q, x), mode).astype(np.float64)
        scale = np.linalg.norm(q.astype(np.float64), axis=-1) * np.linalg.norm(x.astype(np.float64), axis=-1)
        e_nom = np.abs(approx - t) / scale
        # the re-score: fp32 fused multiply-adds = one rounding per exact product added (a d-deep chain here, deeper
        # than the kernel's per-lane chains + butterfly)
        resc = _fp32_chain(q.astype(np.float64) * x.astype(np.float64), "rne").astype(np.float64)
        e_re = np.abs(resc - t) / scale
        assert np.all(e_nom + e_re <= eps), (name, mode, float((e_nom + e_re).max()), eps)
        worst = max(worst, float((e_nom + e_re).max()))
    # the bound is a worst case, not a fit: it must hold with room, and the operand term alone must NOT explain it
    assert worst < eps


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
    assert got > _native.split_eps(d)                  # one coarse product: a wider margin than the bf16 split's


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
    """Host-only query of the pass table: 256 queries per corpus pass where the query-split kernel
    exists, 128 where only the K-split kernel does, 0 where the bf16-split kernels still serve."""
    from minivectordb_amd import _native
    for d in (256, 384, 512):
        assert _native.half_max_queries(d) == 256
    for d in (640, 768, 896, 1024):   # round 3: every even multiple of 64 from 256 to 1024
        assert _native.half_max_queries(d) == 128
    for d in (32, 64, 100, 128, 192, 320, 576, 2048):   # d = 128: tried and dropped (half_scan.hip: half_kq)
        assert _native.half_max_queries(d) == 0
