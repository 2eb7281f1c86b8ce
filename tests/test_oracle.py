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


def test_synth_families_known_answers_and_shape():
    """The two unfriendly families of the synthetic stream ("a numpy.random.rand-style all-positive variant
    (what the reference tests use, tests/test_sharded_multithreaded_operations.py:22) as a stress case for near-ties"; and a
    clustered one with exact / near duplicates): known-answer vectors (the device generator is held against the host's on the
    GPU) and the properties the certified passes are measured against."""
    v = flat.synth(2, 4, 1234 | flat.SYNTH_POSITIVE, 7)
    want = np.array([[0.431235134601593, 0.19885194301605225, 0.1625680923461914, 0.5594544410705566],
                     [0.5369088053703308, 0.8322985768318176, 0.5937770009040833, 0.6396514773368835]], np.float32)
    assert v.tobytes() == want.tobytes(), v.tolist()
    w = flat.synth(1, 3, 5678 | flat.SYNTH_CLUSTERED, (1 << 33) + 5)
    want2 = np.array([[-0.3460564613342285, 0.20024538040161133, -0.6225290298461914]], np.float32)
    assert w.tobytes() == want2.tobytes(), w.tolist()
    pos = flat.synth(20000, 64, 1234 | flat.SYNTH_POSITIVE)
    assert pos.min() >= 0.0 and pos.max() < 1.0 and abs(float(pos.mean()) - 0.5) < 5e-3
    flat.normalize_l2(pos)
    cos = pos[:200] @ pos[200:400].T
    assert 0.5 < float(cos.min()) and float(cos.max()) < 0.95 and abs(float(cos.mean()) - 0.75) < 0.02   # a narrow cone around 0.75
    clu = flat.synth(100000, 32, 1234 | flat.SYNTH_CLUSTERED)
    assert 5 <= 100000 - np.unique(clu, axis=0).shape[0] <= 400        # exact duplicates exist (noise-free rows sharing a centre)
    flat.normalize_l2(clu)
    q = flat.synth(8, 32, 5678 | flat.SYNTH_CLUSTERED)                 # another seed: the SAME centres
    flat.normalize_l2(q)
    s = np.sort(clu @ q.T, axis=0)[::-1]
    assert (s[0] > 0.99).all() and (s[0] - s[9] < 3e-3).all()          # ten rows within 3e-3 of the best: near-ties by design


def test_block_search_and_merge_equal_the_sequential_scan():
    """oracle_flat_search_block + oracle_merge_topk (what tests/bigcheck.py streams a 10M-row device corpus through) must
    return bit for bit what the per-query sequential scan returns: IP and L2, ragged block sizes, duplicate rows (ties
    by id across blocks), k larger than a block, row selections, any thread count."""
    n, d, nq = 7000, 96, 37
    x = flat.synth(n, d, 1234)
    flat.normalize_l2(x)
    x[5000:5003] = x[17]        # exact duplicates in another block: ties must resolve to the lower id
    q = flat.synth(nq, d, 5678)
    q[3] = x[17]
    keep = (np.arange(n) % 3 != 1).astype(np.uint8)
    for metric in (flat.METRIC_IP, flat.METRIC_L2):
        for k, cuts in ((10, (0, 1000, 1001, 4096, n)), (300, (0, 200, 6000, n))):
            want = flat.flat_search(x, q, k, metric=metric)
            parts = [flat.flat_search_block(x[a:b], q, k, id_base=a, metric=metric, nthreads=t + 1)
                     for t, (a, b) in enumerate(zip(cuts[:-1], cuts[1:]))]
            got = flat.merge_topk(parts, metric=metric)
            assert np.array_equal(got[1], want[1]) and got[0].tobytes() == want[0].tobytes()
        rows = np.nonzero(keep)[0]
        Dw, Pw = flat.flat_search(x, q, 10, metric=metric, rows=rows)
        Dg, Ig = flat.flat_search_block(x, q, 10, metric=metric, keep=keep, nthreads=3)
        assert np.array_equal(rows[Pw], Ig) and Dg.tobytes() == Dw.tobytes()
    D, I = flat.flat_search_block(x[:4], q[:2], 6)      # fewer rows than k: -1 / -FLT_MAX padding
    assert (I[:, 4:] == -1).all() and (D[:, 4:] == np.float32(-3.4028234663852886e38)).all()


class _HostIndex:
    """get_rows over a host matrix: lets tests/bigcheck.py's comparison logic run without a GPU."""

    def __init__(self, x):
        self.x, self.d = x, x.shape[1]

    def get_rows(self, row0, n, out=None):
        if out is None:
            return self.x[row0:row0 + n].copy()
        out[:] = self.x[row0:row0 + n]
        return out


def test_fullsize_comparison_logic_accepts_near_ties_and_rejects_wrong_rows():
    import pytest
    import bigcheck
    n, d, k = 5000, 64, 10
    x = flat.synth(n, d, 1234)
    flat.normalize_l2(x)
    q = flat.synth(6, d, 5678)
    flat.normalize_l2(q)
    x[4000] = x[int(flat.flat_search(x, q[:1], k)[1][0, k - 1])]   # a duplicate of query 0's k-th row: an exact tie
    idx = _HostIndex(x)
    (oracle,), _ = bigcheck.oracle_topk_streamed(idx, n, q, k, block=1024)
    Do, Io = oracle
    assert np.array_equal(Io, flat.flat_search(x, q, k)[1])
    rec = bigcheck.compare(idx, q, Do.copy(), Io.copy(), Do, Io, "identical")
    assert rec["queries_with_id_differences"] == 0
    # the tie resolved the other way round: accepted and counted
    I2 = Io.copy()
    assert 4000 not in I2[0]
    I2[0, k - 1] = 4000
    rec = bigcheck.compare(idx, q, Do.copy(), I2, Do, Io, "tie")
    assert rec["queries_with_id_differences"] == 1 and rec["adjudicated_near_ties"] == 1
    # a row that is simply not among the best
    worst = int(np.argmin(x @ q[1]))
    I3 = Io.copy()
    I3[1, k - 1] = worst
    with pytest.raises(AssertionError, match="not a near-tie"):
        bigcheck.compare(idx, q, Do.copy(), I3, Do, Io, "wrong row")
    # a wrong distance
    D4 = Do.copy()
    D4[2, 0] += 3e-4
    with pytest.raises(AssertionError, match="distances differ"):
        bigcheck.compare(idx, q, D4, Io.copy(), Do, Io, "wrong distance")
    # a selection (filtered search): rows outside it never appear
    keep = (np.arange(n) % 2 == 0).astype(np.uint8)
    (sel,), _ = bigcheck.oracle_topk_streamed(idx, n, q, k, keeps=(keep,), block=999)
    assert (sel[1] % 2 == 0).all()
