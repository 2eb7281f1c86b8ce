"""Oracle comparison AT the BASELINE sizes (configs 2 and 3; config 4 is in test_config4_gpu.py): every search path the
library picks for 1 / 32 / 128 / 256 queries per call is held against the CPU oracle on the full corpus — 1,000 queries at
1M x 512, 1,000 at 10M x 512, 512 at config 5's 10M x 384 (1,000 queries per config, seeds 1234 / 5678) — id for id, every id difference
adjudicated in float64 (tests/bigcheck.py).  Reference call site: minivectordb/vector_database.py:497
(``index.search(embedding, search_k)``), filtered branch :508-523.
"""
import numpy as np
import pytest

from oracle import flat

import bigcheck

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native(gpu):
    from minivectordb_amd import _native
    assert _native.device_count() >= 1
    return _native


def _queries(nq, d, seed=5678):
    q = flat.synth(nq, d, seed)
    flat.normalize_l2(q)
    return q


def _in_chunks(search, q, per_call):
    out = [search(q[a:a + per_call]) for a in range(0, q.shape[0], per_call)]
    return np.concatenate([o[0] for o in out]), np.concatenate([o[1] for o in out])


def _bitmap(keep):
    bits = np.zeros((keep.shape[0] + 63) // 64 * 64, dtype=np.uint8)
    bits[:keep.shape[0]] = keep
    return np.packbits(bits, bitorder="little").view(np.uint64)


def test_config2_1M_x_512_1000_queries_vs_oracle(native):
    """BASELINE config 2 at full size: 1,000 queries, k = 10, through the single-query scan and through calls of 32 / 128 /
    256 / 1,000 queries (the fp32-MFMA, bf16-split and fp16-nomination passes and their certified re-runs)."""
    n, d, k, nq = 1_000_000, 512, 10, 1000
    idx = native.FlatIndex(d)
    idx.reserve(n)
    idx.add_synthetic(n, 1234, normalize=True)
    q = _queries(nq, d)
    (oracle,), cost = bigcheck.oracle_topk_streamed(idx, n, q, k)
    Do, Io = oracle
    for per_call in (1, 32, 128, 256, 1000):
        D, I = _in_chunks(lambda qs: idx.search(qs, k), q, per_call)
        rec = bigcheck.compare(idx, q, D, I, Do, Io, f"config2 1M x 512, {per_call} queries per call")
        bigcheck.report(dict(rec, oracle_cost=cost))
    # un-normalised queries through the fused prologue (what find_most_similar sends, vector_database.py:475)
    D, I = _in_chunks(lambda qs: idx.search(qs * np.float32(3.25), k, normalize_q=True), q[:200], 1)
    bigcheck.report(bigcheck.compare(idx, q[:200], D, I, Do[:200], Io[:200], "config2 1M x 512, fused query normalisation"))
    idx.close()


def test_config2_1M_x_512_l2_and_filters_vs_oracle(native):
    """The extensions at config 2's size: L2 metric (north_star names it next to IP) and the filtered branch as a row
    list, a bitmap and a resident row set, 200 queries each, single and batched."""
    n, d, k, nq = 1_000_000, 512, 10, 200
    q = _queries(nq, d)
    idx = native.FlatIndex(d)
    idx.reserve(n)
    idx.add_synthetic(n, 1234, normalize=True)
    rs = np.random.RandomState(11)
    keep_half = (rs.rand(n) < 0.5).astype(np.uint8)
    keep_few = (rs.rand(n) < 0.03).astype(np.uint8)
    (all_rows, half, few), cost = bigcheck.oracle_topk_streamed(idx, n, q, k, keeps=(None, keep_half, keep_few))
    words = _bitmap(keep_half)
    for per_call in (1, 32, 128):
        D, I = _in_chunks(lambda qs: idx.search_masked(qs, k, words), q, per_call)
        bigcheck.report(bigcheck.compare(idx, q, D, I, *half, f"config2 bitmap keeping 50 %, {per_call} queries per call"))
    for keep, want, name in ((keep_half, half, "50 %"), (keep_few, few, "3 %")):
        rows = np.nonzero(keep)[0].astype(np.int64)
        rowset = idx.rowset(rows)
        for per_call in (1, 64):
            D, I = _in_chunks(lambda qs: idx.search_rowset(qs, k, rowset), q, per_call)
            bigcheck.report(bigcheck.compare(idx, q, D, I, *want,
                                             f"config2 resident row set keeping {name}, {per_call} queries per call"))
        D, P = _in_chunks(lambda qs: idx.search_subset(qs, k, rows), q[:50], 1)   # labels = positions in the list
        bigcheck.report(bigcheck.compare(idx, q[:50], D, rows[P], want[0][:50], want[1][:50],
                                         f"config2 row list keeping {name}, per query"))
        rowset.close()
    idx.close()

    l2 = native.FlatIndex(d, metric=native.METRIC_L2)
    l2.reserve(n)
    l2.add_synthetic(n, 1234, normalize=True)
    (oracle,), cost = bigcheck.oracle_topk_streamed(l2, n, q, k, metric=flat.METRIC_L2)
    for per_call in (1, 8, 32, 64, 128, 200):
        D, I = _in_chunks(lambda qs: l2.search(qs, k), q, per_call)
        bigcheck.report(bigcheck.compare(l2, q, D, I, *oracle, f"config2 L2 metric, {per_call} queries per call",
                                         metric=flat.METRIC_L2))
    l2.close()


def test_config3_10M_x_512_every_pass_vs_oracle(native):
    """BASELINE config 3 (the headline) at full size, BASELINE.md section 4's 1,000 queries: all of them against the oracle over
    all 10M rows ONE PER CALL (the headline kernel) and 256 per call; the first 256 also 32 and 128 per call and under a bitmap
    keeping half of the rows."""
    n, d, k, nq, nb = 10_000_000, 512, 10, 1000, 256
    idx = native.FlatIndex(d)
    idx.reserve(n)
    idx.add_synthetic(n, 1234, normalize=True)
    q = _queries(nq, d)
    keep = (np.random.RandomState(3).rand(n) < 0.5).astype(np.uint8)
    (everything,), cost = bigcheck.oracle_topk_streamed(idx, n, q, k)
    (half,), cost_half = bigcheck.oracle_topk_streamed(idx, n, q[:nb], k, keeps=(keep,))
    Do, Io = everything
    reruns = native.split_rerun_count()
    D, I = _in_chunks(lambda qs: idx.search(qs, k), q, 1)
    bigcheck.report(dict(bigcheck.compare(idx, q, D, I, Do, Io, "config3 10M x 512, 1 query per call"), oracle_cost=cost))
    D, I = _in_chunks(lambda qs: idx.search(qs, k), q, 256)
    bigcheck.report(bigcheck.compare(idx, q, D, I, Do, Io, "config3 10M x 512, 256 queries per call"))
    for per_call in (32, 128):
        D, I = _in_chunks(lambda qs: idx.search(qs, k), q[:nb], per_call)
        bigcheck.report(bigcheck.compare(idx, q[:nb], D, I, Do[:nb], Io[:nb], f"config3 10M x 512, {per_call} queries per call"))
    words = _bitmap(keep)
    D, I = _in_chunks(lambda qs: idx.search_masked(qs, k, words), q[:40], 1)
    bigcheck.report(dict(bigcheck.compare(idx, q[:40], D, I, half[0][:40], half[1][:40], "config3 bitmap keeping 50 %, 1 query per call"),
                         oracle_cost=cost_half))
    for per_call in (32, 128):
        D, I = _in_chunks(lambda qs: idx.search_masked(qs, k, words), q[:nb], per_call)
        bigcheck.report(bigcheck.compare(idx, q[:nb], D, I, *half, f"config3 bitmap keeping 50 %, {per_call} queries per call"))
    # the opt-in single-query route over the fp16 shadow (mvdb_index_set_option "shadow_single_query"): the first 100 queries
    idx.set_option("shadow_single_query", 1)
    D, I = _in_chunks(lambda qs: idx.search(qs, k), q[:100], 1)
    idx.set_option("shadow_single_query", 0)
    bigcheck.report(bigcheck.compare(idx, q[:100], D, I, Do[:100], Io[:100], "config3 10M x 512, 1 query per call over the fp16 shadow (opt-in)"))
    bigcheck.report({"what": "config3 certified-pass re-runs during the comparison",
                     "chunks_rerun": native.split_rerun_count() - reruns})
    idx.close()


@pytest.mark.parametrize("family,name", [(0, "zero-mean"), (flat.SYNTH_CLUSTERED, "clustered")])
def test_config5_10M_x_384_knn_leg_vs_oracle(native, family, name):
    """BASELINE config 5's kNN leg at full size — 10M x 384, k = 10: the fp16-shadow kernel at d = 384 is its own instantiation
    (flat_scan_h16_kernel<24, ...>), as are the d = 384 forms of the single-query scan and of the exact fp32-MFMA pass.  Queries:
    the 256 golden embeddings of the config-5 batch (transformers' own output for tests/golden/encoder_golden.npz case 8 —
    what the encoder hands the search) + 256 queries of the corpus' own family; 1 / 32 / 128 / 256 per call, id for id against
    the streamed oracle, on the zero-mean stream and on the clustered family (where certificates are refused and the rescue /
    exact re-run tiers answer)."""
    from encoder_cases import load_cases
    n, d, k = 10_000_000, 384, 10
    case = next(c for c in load_cases() if c["B"] == 256)
    assert case["emb"].shape == (256, d)
    qs = flat.synth(256, d, 5678 | family)
    flat.normalize_l2(qs)
    q = np.ascontiguousarray(np.concatenate([case["emb"].astype(np.float32), qs]))
    nq = q.shape[0]
    idx = native.FlatIndex(d)
    idx.reserve(n)
    idx.add_synthetic(n, 1234 | family, normalize=True)
    (oracle,), cost = bigcheck.oracle_topk_streamed(idx, n, q, k)
    Do, Io = oracle
    gap = float(np.median(Do[:, 0] - Do[:, k - 1]))
    pick = np.r_[0:64, 256:320]    # one per call: 64 golden + 64 synthetic
    D, I = _in_chunks(lambda x: idx.search(x, k), q[pick], 1)
    bigcheck.report(dict(bigcheck.compare(idx, q[pick], D, I, Do[pick], Io[pick], f"config5 {name} 10M x 384, 1 query per call"),
                         oracle_cost=cost, median_gap_best_to_kth=gap))
    for per_call in (32, 128, 256):
        before = native.split_rerun_count()
        D, I = _in_chunks(lambda x: idx.search(x, k), q, per_call)
        rec = bigcheck.compare(idx, q, D, I, Do, Io, f"config5 {name} 10M x 384, {per_call} queries per call")
        bigcheck.report(dict(rec, chunks_rerun=native.split_rerun_count() - before, calls=(nq + per_call - 1) // per_call))
    assert idx.shadow_rows == n   # the batches did run over the fp16 shadow
    idx.close()


def test_config3_10M_x_512_l2_vs_oracle(native):
    """The L2 metric at the headline size (north_star: "flat inner-product / L2 kNN"): 128 queries against the oracle over all
    10M rows, one per call (the exact GEMV scan), 8 per call (fp32-MFMA pass) and 32 / 128 per call (round 4: the inner
    product's certified passes over the fp16 shadow, L2 re-score, norm-range certificate, device-gated exact re-run)."""
    n, d, k, nq = 10_000_000, 512, 10, 128
    idx = native.FlatIndex(d, metric=native.METRIC_L2)
    idx.reserve(n)
    idx.add_synthetic(n, 1234, normalize=True)
    q = _queries(nq, d)
    (oracle,), cost = bigcheck.oracle_topk_streamed(idx, n, q, k, metric=flat.METRIC_L2)
    Do, Io = oracle
    reruns = native.split_rerun_count()
    D, I = _in_chunks(lambda qs: idx.search(qs, k), q[:32], 1)
    bigcheck.report(dict(bigcheck.compare(idx, q[:32], D, I, Do[:32], Io[:32], "config3 L2 10M x 512, 1 query per call",
                                          metric=flat.METRIC_L2), oracle_cost=cost))
    for per_call in (8, 32, 128):
        D, I = _in_chunks(lambda qs: idx.search(qs, k), q, per_call)
        bigcheck.report(bigcheck.compare(idx, q, D, I, Do, Io, f"config3 L2 10M x 512, {per_call} queries per call",
                                         metric=flat.METRIC_L2))
    assert idx.shadow_rows == n   # the batches did run over the fp16 shadow
    bigcheck.report({"what": "config3 L2 certified-pass re-runs during the comparison",
                     "chunks_rerun": native.split_rerun_count() - reruns})
    idx.close()


def test_config2_1M_x_512_l2_mixed_norms_vs_oracle(native):
    """L2 over UN-normalised rows at config 2's size (the squared norms spread over a few per cent: far beyond what the norm-range
    certificate tolerates, so batches nominate by q.x - |x|^2 / 2 with per-row offsets, round 4): 128 queries of assorted
    norms against the streamed oracle, 1 / 8 / 32 / 128 per call, and under a bitmap keeping half of the rows."""
    n, d, k, nq = 1_000_000, 512, 10, 128
    idx = native.FlatIndex(d, metric=native.METRIC_L2)
    idx.reserve(n)
    idx.add_synthetic(n, 1234, normalize=False)
    q = flat.synth(nq, d, 5678)
    q *= np.exp(np.random.RandomState(1).uniform(np.log(0.3), np.log(2.0), size=(nq, 1))).astype(np.float32)
    keep = (np.random.RandomState(2).rand(n) < 0.5).astype(np.uint8)
    (everything, half), cost = bigcheck.oracle_topk_streamed(idx, n, q, k, keeps=(None, keep), metric=flat.METRIC_L2)
    reruns = native.split_rerun_count()
    for per_call in (1, 8, 32, 128):
        D, I = _in_chunks(lambda qs: idx.search(qs, k), q, per_call)
        bigcheck.report(dict(bigcheck.compare(idx, q, D, I, *everything, f"config2 L2 mixed norms, {per_call} queries per call",
                                              metric=flat.METRIC_L2), oracle_cost=cost))
    assert idx.shadow_rows == n
    words = _bitmap(keep)
    for per_call in (1, 64):
        D, I = _in_chunks(lambda qs: idx.search_masked(qs, k, words), q, per_call)
        bigcheck.report(bigcheck.compare(idx, q, D, I, *half, f"config2 L2 mixed norms under a 50 % bitmap, {per_call} queries per call",
                                         metric=flat.METRIC_L2))
    bigcheck.report({"what": "config2 L2 mixed norms: certified-pass re-runs", "chunks_rerun": native.split_rerun_count() - reruns})
    idx.close()


@pytest.mark.parametrize("family,name", [(flat.SYNTH_POSITIVE, "all-positive"), (flat.SYNTH_CLUSTERED, "clustered")])
@pytest.mark.parametrize("n", [1_000_000, 10_000_000])
def test_certified_passes_on_unfriendly_corpora_vs_oracle(native, n, family, name):
    """Every full-size record above is on the zero-mean stream, where the 10th and 64th scores of a query are 6e-3 apart and the
    certified passes (fp16 nomination + exact fp32 re-score + margin test, eps(512) = 1.04e-3) always certify.  Here: the
    reference's own kind of test vector (numpy.random.rand rows, tests/test_sharded_multithreaded_operations.py:22 — SURVEY.md
    section 8(d)), and a clustered corpus with thousands of rows within 1e-3 of a query's best, exact duplicates included.  256
    queries of the same family, 1 / 32 / 128 / 256 per call and the opt-in single-query shadow route, id for id against the
    streamed oracle; what failed certification is re-run exactly, and how often that happens is reported."""
    d, k, nq = 512, 10, 256
    idx = native.FlatIndex(d)
    idx.reserve(n)
    idx.add_synthetic(n, 1234 | family, normalize=True)
    q = flat.synth(nq, d, 5678 | family)
    flat.normalize_l2(q)
    (oracle,), cost = bigcheck.oracle_topk_streamed(idx, n, q, k)
    Do, Io = oracle
    gap = float(np.median(Do[:, 0] - Do[:, k - 1]))
    nsingle = 64 if n > 1_000_000 else 256
    D, I = _in_chunks(lambda qs: idx.search(qs, k), q[:nsingle], 1)
    bigcheck.report(dict(bigcheck.compare(idx, q[:nsingle], D, I, Do[:nsingle], Io[:nsingle], f"{name} {n} x 512, 1 query per call"),
                         oracle_cost=cost, median_gap_best_to_kth=gap))
    for per_call in (32, 128, 256):
        before = native.split_rerun_count()
        D, I = _in_chunks(lambda qs: idx.search(qs, k), q, per_call)
        rec = bigcheck.compare(idx, q, D, I, Do, Io, f"{name} {n} x 512, {per_call} queries per call")
        bigcheck.report(dict(rec, chunks_rerun=native.split_rerun_count() - before, calls=(nq + per_call - 1) // per_call))
    idx.set_option("shadow_single_query", 1)
    before = native.split_rerun_count()
    D, I = _in_chunks(lambda qs: idx.search(qs, k), q[:nsingle], 1)
    idx.set_option("shadow_single_query", 0)
    rec = bigcheck.compare(idx, q[:nsingle], D, I, Do[:nsingle], Io[:nsingle], f"{name} {n} x 512, 1 query per call over the fp16 shadow (opt-in)")
    bigcheck.report(dict(rec, chunks_rerun=native.split_rerun_count() - before, calls=nsingle))
    idx.close()
