"""The operand-error bound the split-precision batch pass certifies with (minivectordb_amd/csrc/scan_split_kernels.hpp):

    |q.x - (qh.xh + qh.xl + ql.xh)| <= 3 * 2^-16 * sum|q_i x_i| <= 4.6e-5 * |q| * |x|

with (h, l) the bf16 round-to-nearest-even split of an fp32 value.  Checked here in numpy (bf16 emulated bit-exactly,
sums in float64 so that only the operand error is measured) on random and adversarial inputs; the GPU kernels add the
fp32 accumulation error, which kSplitEps = 1e-4 leaves room for.
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
