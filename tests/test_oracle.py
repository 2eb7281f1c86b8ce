"""The CPU oracle itself: checked against float64 numpy brute force and known answers.

The reference's tests hold NO numeric golden vectors for this path (no test asserts a
score), and faiss is absent, so the oracle is pinned by (i) float64 ground truth, (ii) the golden
scenarios recorded through the reference's own plumbing (test_golden_cpu.py) and (iii) the
known-answer vectors below for the synthetic stream."""
import numpy as np
import pytest

from oracle import flat


def brute(x, q, k):
    s = x.astype(np.float64) @ q.astype(np.float64)
    order = np.lexsort((np.arange(len(s)), -s))[:k]
    return s[order], order


@pytest.mark.parametrize("n,d,k", [(1000, 512, 5), (5000, 384, 10), (300, 7, 300), (64, 1, 3), (2000, 100, 64)])
def test_flat_search_matches_float64(n, d, k):
    x = flat.synth(n, d, 1234)
    flat.normalize_l2(x)
    q = flat.synth(5, d, 5678)
    flat.normalize_l2(q)
    D, I = flat.flat_search(x, q, k)
    D64, I64 = flat.flat_search(x, q, k, f64=True)
    for i in range(5):
        s, order = brute(x, q[i], k)
        assert np.array_equal(I64[i], order)
        np.testing.assert_allclose(D64[i], s, rtol=0, atol=1e-12)
        np.testing.assert_allclose(D[i], s, rtol=0, atol=2e-6)
        ok, msg = flat.adjudicate(x, q[i], k, D[i], I[i])
        assert ok, msg


def test_threads_do_not_change_results():
    x = flat.synth(20000, 64, 3)
    q = flat.synth(3, 64, 4)
    a = flat.flat_search(x, q, 10, nthreads=1)
    b = flat.flat_search(x, q, 10, nthreads=4)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_ties_lower_row_first_and_padding():
    x = np.ones((10, 4), np.float32)
    D, I = flat.flat_search(x, np.ones((1, 4), np.float32), 4)
    assert I[0].tolist() == [0, 1, 2, 3]
    D, I = flat.flat_search(x, np.ones((1, 4), np.float32), 12)
    assert I[0].tolist() == list(range(10)) + [-1, -1]
    assert D[0, 10] == np.float32(-3.4028234663852886e38)


def test_subset_labels_are_positions():
    x = flat.synth(100, 16, 1)
    q = flat.synth(1, 16, 2)
    rows = np.array([50, 3, 99, 7], np.int64)
    D, I = flat.flat_search(x, q, 4, rows=rows)
    Dfull = x[rows] @ q[0]
    assert I[0].tolist() == np.argsort(-Dfull, kind="stable").tolist()


def test_normalize_l2_semantics():
    x = flat.synth(50, 33, 9)
    x[7] = 0
    y = x.copy()
    flat.normalize_l2(y)
    assert not y[7].any()
    n = np.linalg.norm(np.delete(y, 7, 0).astype(np.float64), axis=1)
    np.testing.assert_allclose(n, 1.0, atol=1e-6)
    np.testing.assert_allclose(np.delete(y, 7, 0), np.delete(x, 7, 0) / np.linalg.norm(
        np.delete(x, 7, 0).astype(np.float64), axis=1, keepdims=True), atol=1e-7)


def test_l2_metric():
    x = flat.synth(500, 24, 5)
    q = flat.synth(2, 24, 6)
    D, I = flat.flat_search(x, q, 5, metric=flat.METRIC_L2)
    for i in range(2):
        d2 = ((x.astype(np.float64) - q[i].astype(np.float64)) ** 2).sum(1)
        order = np.lexsort((np.arange(500), d2))[:5]
        assert np.array_equal(I[i], order)
        np.testing.assert_allclose(D[i], d2[order], atol=1e-5)


def test_synth_known_answers():
    """Known-answer vectors of the synthetic stream: host and device generators must both keep
    producing exactly these bits (the device side is compared with the host side on the GPU)."""
    v = flat.synth(2, 4, 1234, 0)
    want = np.array([[0.6426162719726562, 0.1812286376953125, -0.1288604736328125, -0.1969451904296875],
                     [-0.00726318359375, -0.18012237548828125, 0.43199920654296875, 0.14659881591796875]], np.float32)
    assert v.tobytes() == want.tobytes(), v.tolist()
    w = flat.synth(1, 3, 5678, (1 << 33) + 5)
    want2 = np.array([[0.05548095703125, 0.0550079345703125, 0.2867279052734375]], np.float32)
    assert w.tobytes() == want2.tobytes(), w.tolist()
    big = flat.synth(4096, 512, 1234)
    assert abs(float(big.mean())) < 2e-3 and abs(float(big.std()) - 0.2887) < 5e-3  # Irwin-Hall(4)/2 spread
