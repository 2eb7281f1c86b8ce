"""The layer-walking launch (csrc/encoder_walk.hpp: <= 128 token slots, the reference's one-sentence-per-call shape,
minivectordb/embedding_model.py:62-71) against the float64 restatement and against the per-op kernels of the same library.
Tolerances as test_encoder_gpu.py: 2e-5 on the unit-norm embeddings, 1e-4 on hidden states (values up to ~6)."""
import numpy as np
import pytest

from oracle import encoder as E

pytestmark = pytest.mark.gpu


def _model(cfg, weights, **kw):
    import torch
    from minivectordb_amd.embedding_model import GpuEncoder
    return GpuEncoder(cfg, {k: torch.from_numpy(v) for k, v in weights.items()}, device=0, **kw)


SHAPES = [("tiny", 1, 5), ("tiny", 3, 9), ("tiny", 4, 16), ("hd64", 2, 17), ("hd64", 1, 64), ("xlmr-tiny", 3, 11), ("h96", 2, 23),
          ("e5-small-dims", 1, 7), ("e5-small-dims", 1, 16), ("e5-small-dims", 1, 33), ("e5-small-dims", 1, 64),
          ("e5-small-dims", 3, 21), ("e5-small-dims", 4, 16), ("xlmr-large-dims", 1, 12), ("xlmr-large-dims", 2, 30),
          ("xlmr-large-dims", 1, 64), ("e5-small-dims", 1, 100), ("e5-small-dims", 1, 128), ("e5-small-dims", 5, 25), ("e5-small-dims", 128, 1),
          ("xlmr-large-dims", 1, 97), ("xlmr-large-dims", 2, 64), ("hd64", 3, 40), ("h96", 5, 23)]


@pytest.mark.parametrize("name,B,S", SHAPES, ids=[f"{n}-{b}x{s}" for n, b, s in SHAPES])
def test_walk_matches_float64(name, B, S, gpu):
    import torch
    cfg = E.make_config(name)
    w = E.make_weights(cfg, 31)
    ids, mask = E.make_inputs(cfg, B, S, 32)
    enc = _model(cfg, w)
    assert enc.walks(B, S)
    h64, e64 = E.numpy_forward(cfg, w, ids, mask)
    for compute in (2, 0):          # one exact-fp32 launch serves both modes
        emb = enc.forward(ids, mask, compute=compute)
        np.testing.assert_allclose(emb, e64, atol=2e-5, rtol=0)
    dev = torch.device("cuda", 0)
    out, hidden = enc.forward_device(torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev), want_hidden=True)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), emb)
    hidden = hidden.cpu().numpy()
    m = mask.astype(bool)
    np.testing.assert_allclose(hidden[m], h64[m], atol=1e-4, rtol=0)
    assert not hidden[~m].any()
    # arbitrary (non-prefix) masks
    if S > 4:
        mask2 = mask.copy()
        mask2[:, 2] = 0
        _, e64b = E.numpy_forward(cfg, w, ids, mask2)
        np.testing.assert_allclose(enc.forward(ids, mask2), e64b, atol=2e-5, rtol=0)
    # repeated launches re-arm the barrier words: identical bits
    for _ in range(3):
        assert np.array_equal(enc.forward(ids, mask), emb)
    enc.close()


def test_walk_row_is_independent_of_the_batch(gpu):
    """A sentence's embedding is bit-for-bit the same alone, in any small batch, at any position and under any padding
    width (the reference semantics are B = 1)."""
    cfg = E.make_config("e5-small-dims")
    enc = _model(cfg, E.make_weights(cfg, 21))
    ids, mask = E.make_inputs(cfg, 4, 16, 22)
    ref = enc.forward(ids, mask)
    for b in range(4):
        n = int(mask[b].sum())
        alone = enc.forward(ids[b:b + 1, :n], mask[b:b + 1, :n])
        assert np.array_equal(alone[0], ref[b]), b
    assert np.array_equal(enc.forward(ids[::-1].copy(), mask[::-1].copy()), ref[::-1])
    assert np.array_equal(enc.forward(ids[1:3], mask[1:3]), ref[1:3])
    enc.close()


def test_walk_agrees_with_the_per_op_kernels(gpu, monkeypatch):
    """Same weights through the per-op kernel chain (MVDB_ENCODER_WALK=0, both modes): rounding-level agreement, and the
    cls pooling of the bge-m3 shape."""
    cfg = E.make_config("e5-small-dims")
    w = E.make_weights(cfg, 5)
    ids, mask = E.make_inputs(cfg, 2, 30, 6)
    walk = _model(cfg, w)
    got = walk.forward(ids, mask)
    walk.close()
    monkeypatch.setenv("MVDB_ENCODER_WALK", "0")
    chain = _model(cfg, w)
    assert not chain.walks(2, 30)
    for compute in (0, 2):
        np.testing.assert_allclose(got, chain.forward(ids, mask, compute=compute), atol=3e-6, rtol=0)
    chain.close()
    monkeypatch.delenv("MVDB_ENCODER_WALK")
    xcfg = E.make_config("xlmr-large-dims")
    xw = E.make_weights(xcfg, 12)
    xi, xm = E.make_inputs(xcfg, 2, 20, 13)
    enc = _model(xcfg, xw, pooling="cls")
    h64, _ = E.numpy_forward(xcfg, xw, xi, xm)
    cls = h64[:, 0] / np.linalg.norm(h64[:, 0], axis=1, keepdims=True)
    np.testing.assert_allclose(enc.forward(xi, xm), cls, atol=2e-5, rtol=0)
    enc.close()


def test_walk_all_padding_sentence_is_nan_like_the_reference(gpu):
    cfg = E.make_config("tiny")
    enc = _model(cfg, E.make_weights(cfg, 3))
    ids, mask = E.make_inputs(cfg, 3, 9, 4)
    mask[1] = 0
    out = enc.forward(ids, mask, compute=0)
    assert np.isnan(out[1]).all() and np.isfinite(out[[0, 2]]).all()   # 0 / 0 in average_pool (embedding_model.py:50-53)
    _, e64 = E.numpy_forward(cfg, E.make_weights(cfg, 3), ids[[0, 2]], mask[[0, 2]])
    np.testing.assert_allclose(out[[0, 2]], e64, atol=2e-5, rtol=0)
    enc.close()


def test_walk_fuzz_random_batches_and_masks(gpu):
    """60 random small batches (1..12 sentences, 1..128 token slots, ragged lengths, holes in the masks, all-padding rows) on
    four model shapes — head widths 32 and 64, XLM-R position ids, a width that is not a multiple of 64 — against the float64
    restatement, and each batch row against its own one-sentence forward bit for bit."""
    rs = np.random.RandomState(2024)
    encs = {}
    for trial in range(60):
        name = ("tiny", "hd64", "xlmr-tiny", "h96")[trial % 4]
        cfg = E.make_config(name)
        if name not in encs:
            w = E.make_weights(cfg, 40 + trial)
            encs[name] = (_model(cfg, w), w)
        enc, w = encs[name]
        B = int(rs.randint(1, 13))
        smax = min(128 // B, cfg["max_position_embeddings"] - 2)
        S = int(rs.randint(1, smax + 1))
        ids, mask = E.make_inputs(cfg, B, S, 1000 + trial)
        if S > 3 and trial % 3 == 0:
            mask[rs.randint(0, B), rs.randint(1, S)] = 0          # a hole
        if B > 2 and trial % 5 == 0:
            mask[rs.randint(1, B)] = 0                           # a sentence of padding only
        assert enc.walks(B, S)
        live = mask.sum(axis=1) > 0
        with np.errstate(invalid="ignore", divide="ignore"):
            _, e64 = E.numpy_forward(cfg, w, ids[live], mask[live])
        got = enc.forward(ids, mask)
        np.testing.assert_allclose(got[live], e64, atol=2e-5, rtol=0, err_msg=f"trial {trial}: {name} {B}x{S}")
        assert np.isnan(got[~live]).all()
        b = int(np.flatnonzero(live)[rs.randint(0, live.sum())])
        assert np.array_equal(enc.forward(ids[b:b + 1], mask[b:b + 1])[0], got[b]), (trial, b)
    for enc, _ in encs.values():
        enc.close()


def test_walk_launches_from_two_streams_are_ordered(gpu):
    """Two callers, two streams, one encoder: the walking launches share the encoder's phase counters and buffers, so the
    library orders them on the device (an event behind every launch, waited for by the next) — overlapping grids would
    corrupt each other's counters and never finish."""
    import torch
    cfg = E.make_config("e5-small-dims")
    w = E.make_weights(cfg, 77)
    enc = _model(cfg, w)
    dev = torch.device("cuda", 0)
    ids, mask = E.make_inputs(cfg, 2, 40, 78)
    _, e64 = E.numpy_forward(cfg, w, ids, mask)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    ia = [torch.from_numpy(ids[b:b + 1]).to(dev) for b in range(2)]
    ma = [torch.from_numpy(mask[b:b + 1]).to(dev) for b in range(2)]
    torch.cuda.synchronize()
    outs = []
    for rep in range(40):
        for b, st in ((0, s1), (1, s2)):
            with torch.cuda.stream(st):
                outs.append((b, enc.forward_device(ia[b], ma[b])[0]))
    torch.cuda.synchronize()
    for b, o in outs:
        np.testing.assert_allclose(o.cpu().numpy()[0], e64[b], atol=2e-5, rtol=0)
    enc.close()


def test_walk_launches_of_two_encoders_are_ordered_on_the_device(gpu):
    """Two encoders of one process, two streams: their walking launches are persistent grids that spin on their own
    workgroups — resident together they could each hold CUs the other's missing workgroups need.  The library orders every
    walking launch of a device behind the one before (the large shape asks for every CU: two of those cannot coexist)."""
    import torch
    dev = torch.device("cuda", 0)
    cfgs = [E.make_config("xlmr-large-dims"), E.make_config("e5-small-dims")]
    ws = [E.make_weights(cfgs[0], 5), E.make_weights(cfgs[1], 6)]
    encs = [_model(cfgs[0], ws[0]), _model(cfgs[1], ws[1])]
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    data, want = [], []
    for cfg, w in zip(cfgs, ws):
        ids, mask = E.make_inputs(cfg, 1, 24, 17)
        data.append((torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)))
        want.append(E.numpy_forward(cfg, w, ids, mask)[1])
    torch.cuda.synchronize()
    outs = []
    for rep in range(30):
        for i in (0, 1):
            with torch.cuda.stream(streams[i]):
                outs.append((i, encs[i].forward_device(*data[i])[0]))
    torch.cuda.synchronize()
    for i, o in outs:
        np.testing.assert_allclose(o.cpu().numpy(), want[i], atol=2e-5, rtol=0)
    for e in encs:
        e.close()
