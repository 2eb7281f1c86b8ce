"""The tile flags of the certified batch pass (round 6), restated on the CPU.

The pass's main launches raise, per query, the bit of every 32-row tile whose best approximate score is NOT below

    thr - flag_coef |q|,     flag_coef = (2.25 half_eps(d) + 4 d 2^-24) max|x|        (mvdb.hip: launch_half_pass)

where thr is the threshold the scanning wave holds for that query when it reaches the tile (the larger of the phase's floor —
the 16th best approximate score of the rows before the phase — and the 16th score of the block's own list).  Should the query's
certificate be refused, its rescue launch walks only the flagged tiles (+ the seed launch's) and admits rows with

    a(x) >= F = nextafter(t - m |q|, -inf) - e |q|,      t = the k-th fp32 re-score among the 64 nominees, m = 2 d 2^-24 max|x|,
                                                          e = half_eps(d) max|x|.

The claim that makes this exact: every row with a(x) >= F lies in a flagged tile (or a seed tile), for k <= 16.  This file
replays the pass's bookkeeping in numpy — fp16 images, phases, per-block 16-deep lists, folds, the 64 nominees, the fp32
re-score — on corpora built to put hundreds of rows inside the band, and checks the claim row by row (with the band set to zero
the first case fails: rows the rescue pass must see sit in tiles whose best score is just below the running threshold; the
worst-case band itself is the derivation's, DESIGN.md 4.3g — random data does not come near it).  (The GPU side of the
same statement: tests/test_flat_gpu.py::test_rescue_launches_skip_tiles_no_refused_query_flagged — same bits with and without
the tile lists.)
"""
import numpy as np
import pytest

from test_split_bound import half_eps

KEEP, RESCORE = 16, 64


def _fp16_image(v, scale):
    return (v.astype(np.float64) * scale).astype(np.float16).astype(np.float64)


def _pow2_scale(bound):
    _, e = np.frexp(bound)
    return np.ldexp(1.0, 15 - e)


def _replay(x, q, k, blocks, seed_tiles, growth, last_growth):
    """One query through the pass.  Returns (flagged tiles incl. seed, F, a) with a the approximate scores of every row."""
    n, d = x.shape
    B = float(np.sqrt((x.astype(np.float64) ** 2).sum(1)).max())                 # the index' row-norm bound
    sx, sq = _pow2_scale(B), _pow2_scale(float(np.abs(q).max()))
    a = ((_fp16_image(x, sx) @ _fp16_image(q, sq)) / (sx * sq)).astype(np.float32)  # one product, exact scales
    qn = float(np.sqrt((q.astype(np.float64) ** 2).sum()))
    e, m = half_eps(d) * B * (1 + 1e-6), 2.0 * d * 2.0 ** -24 * B * (1 + 1e-6)
    coef = (2.25 * half_eps(d) + 4.0 * d * 2.0 ** -24) * B * (1 + 1e-6)
    band = np.float32(coef) * np.float32(qn)
    ntiles = (n + 31) // 32
    flagged = np.zeros(ntiles, dtype=bool)
    flagged[:seed_tiles] = True                                                    # never scanned by a flag-writing launch
    # seed launch: every score of the first tiles; the running nominees = their 16 best, the first floor = the 16th
    base = np.sort(a[: seed_tiles * 32])[::-1][:KEEP]
    floor = np.float32(base[KEEP - 1]) if base.size >= KEEP else np.float32(-np.inf)
    # phases, planned backwards (launch_half_pass)
    ends, b, g = [], ntiles, last_growth
    while b > seed_tiles:
        ends.append(b)
        b = (b + g - 1) // g
        g = growth
        if b <= 2 * seed_tiles:
            break
    covered, lists = seed_tiles, []
    for p in range(len(ends) - 1, -1, -1):
        lists = [np.empty(0, np.float32) for _ in range(blocks)]
        for blk in range(blocks):
            thr = floor
            for t in range(covered + blk, ends[p], blocks):                         # the block's tiles, in its order
                s = a[t * 32:(t + 1) * 32]
                mx = s.max()
                if not (mx < thr - band):                                           # the flag test, thr BEFORE the tile
                    flagged[t] = True
                cand = s[s >= thr] if np.isfinite(thr) else s                       # (ties kept: the kernel's key order drops some — fewer inserts only raise thr later)
                if cand.size:
                    merged = np.sort(np.concatenate([lists[blk], cand]))[::-1][:KEEP]
                    lists[blk] = merged
                    if merged.size >= KEEP:
                        thr = max(thr, np.float32(merged[KEEP - 1]))
        covered = ends[p]
        if p > 0:                                                                   # fold: the running 16 best, the next floor
            base = np.sort(np.concatenate([base] + lists))[::-1][:KEEP]
            floor = np.float32(base[KEEP - 1]) if base.size >= KEEP else np.float32(-np.inf)
    # certification: R = the 64 best candidates by a(), re-scored in fp32; t = the k-th re-score
    pool = np.sort(np.concatenate([base] + lists))[::-1][:RESCORE]
    rows = np.flatnonzero(a >= pool[-1])                                            # rows of the pool (ties: a superset — raises t at most)
    rows = rows[np.argsort(-a[rows], kind="stable")][:RESCORE]
    r = (x[rows].astype(np.float32) @ q.astype(np.float32)).astype(np.float32)
    t = np.sort(r)[::-1][k - 1]
    fl = np.nextafter(np.float32(t - np.float32(m) * np.float32(qn)), np.float32(-np.inf))
    F = np.float32(fl - np.float32(e) * np.float32(qn))
    return flagged, F, a


def _clustered(rng, n, d, centres, noise):
    c = rng.standard_normal((centres, d)).astype(np.float32)
    c /= np.linalg.norm(c, axis=1, keepdims=True)
    x = c[rng.integers(0, centres, n)] + noise * rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float32), c


@pytest.mark.parametrize("d,k,noise", [(128, 10, 2e-4), (256, 16, 1e-4), (128, 1, 5e-4), (384, 10, 3e-5), (128, 12, 0.0)])
def test_every_row_the_rescue_pass_would_admit_lies_in_a_flagged_tile(d, k, noise):
    rng = np.random.default_rng(d * 1000 + k)
    n, centres, blocks, seed_tiles = 24_000, 12, 8, 8
    x, c = _clustered(rng, n, d, centres, noise)
    inside_total = unflagged_total = 0
    for qi in range(6):
        q = c[qi % centres] + (noise if noise else 1e-4) * rng.standard_normal(d).astype(np.float32)
        q = (q / np.linalg.norm(q)).astype(np.float32)
        for growth, last_growth in ((16, 6), (6, 4), (2, 2)):
            flagged, F, a = _replay(x, q, k, blocks, seed_tiles, growth, last_growth)
            admitted = np.flatnonzero(a >= F)
            assert admitted.size >= k
            missing = admitted[~flagged[admitted // 32]]
            assert missing.size == 0, (qi, growth, missing[:5], a[missing[:5]], F)
            inside_total += admitted.size
            unflagged_total += int((~flagged).sum())
    # the test must not be vacuous: the bands held far more rows than the 64 nominees, and tiles were skipped
    assert inside_total > 6 * 3 * 200, inside_total
    assert unflagged_total > 0


def test_flags_on_a_friendly_corpus_are_few():
    """Zero-mean rows: a query flags a few dozen tiles of hundreds, and still every admitted row is covered."""
    rng = np.random.default_rng(7)
    n, d, k = 32_000, 128, 10
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    q = rng.standard_normal(d).astype(np.float32)
    q /= np.linalg.norm(q)
    flagged, F, a = _replay(x, q, k, blocks=8, seed_tiles=8, growth=16, last_growth=6)
    admitted = np.flatnonzero(a >= F)
    assert flagged[admitted // 32].all()
    assert flagged.sum() < 0.35 * flagged.size, flagged.sum()
