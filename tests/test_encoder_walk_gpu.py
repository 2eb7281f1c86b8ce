"""The layer-walking launch (csrc/encoder_walk.hpp: <= 64 token slots since round 6, the reference's one-sentence-per-call shape,
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
    # the walking launch serves 64 token slots (the per-op kernels win beyond since round 6); the shapes that hand over stay in
    # the list: the same float64 restatement holds for whichever kernels answer
    assert enc.walks(B, S) == (B * S <= 64)
    h64, e64 = E.numpy_forward(cfg, w, ids, mask)
    embs = {}
    for compute in (0, 2):          # one exact-fp32 launch serves both modes
        embs[compute] = emb = enc.forward(ids, mask, compute=compute)
        np.testing.assert_allclose(emb, e64, atol=2e-5, rtol=0)
    if enc.walks(B, S):
        assert np.array_equal(embs[0], embs[2])
    dev = torch.device("cuda", 0)
    out, hidden = enc.forward_device(torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev), want_hidden=True)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), emb)   # (the default mode, 2: the last one above)
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
    """60 random small batches (1..12 sentences, 1..64 token slots, ragged lengths, holes in the masks, all-padding rows) on
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
        smax = min(64 // B, cfg["max_position_embeddings"] - 2)
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


def test_walk_abandoned_by_its_deadline_falls_back_to_the_per_op_kernels(gpu, monkeypatch):
    """Bounded waits (encoder_walk.hpp, Args::deadline): with a deadline of 0 us the first wait that finds its phase incomplete
    abandons the launch.  The launch must END (no hang), count itself, poison its output — and the host entry must return the
    embedding all the same, from the per-op kernels, within the same call.  Then the encoder stays on the per-op kernels for
    a while and finally walks again."""
    import torch
    monkeypatch.setenv("MVDB_WALK_DEADLINE_US", "0")
    cfg = E.make_config("e5-small-dims")
    w = E.make_weights(cfg, 41)
    ids, mask = E.make_inputs(cfg, 1, 40, 42)
    _, e64 = E.numpy_forward(cfg, w, ids, mask)
    enc = _model(cfg, w)
    assert enc.walks(1, 40)
    got = enc.forward(ids, mask)
    np.testing.assert_allclose(got, e64, atol=2e-5, rtol=0)
    st = enc.walk_stats()
    assert st["aborts"] == 1 and st["fallbacks"] == 1, st
    # the next calls do not pay the deadline again: they are served by the per-op kernels, identical bits
    for _ in range(5):
        assert np.array_equal(enc.forward(ids, mask), got)
    st = enc.walk_stats()
    assert st["aborts"] == 1 and st["fallbacks"] == 1 and 0 < st["suspended_calls"] < 256, st
    # the device entry: the abandoned launch poisons `out` and raises the overflow word; rerun_on_overflow=True recovers
    enc2 = _model(cfg, w)
    dev = torch.device("cuda", 0)
    ti, tm = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
    out, _ = enc2.forward_device(ti, tm)
    torch.cuda.synchronize()
    assert torch.isnan(out).all() and int(enc2.overflow_flag().item()) == 1
    assert enc2.walk_stats()["aborts"] == 1
    out, _ = enc2.forward_device(ti, tm, rerun_on_overflow=True)   # (suspended by now: per-op kernels)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), e64, atol=2e-5, rtol=0)
    enc.close()
    enc2.close()
    # the same on a GOLDEN case (transformers' own BertModel output, tests/golden/encoder_golden.npz: the smallest one, 3 x 9 slots):
    # the abandoned launch's fallback returns the golden embeddings
    from encoder_cases import load_cases
    gc = next(c for c in load_cases() if c["B"] * c["S"] <= 64)
    gcfg = E.make_config(gc["name"])
    enc4 = _model(gcfg, E.make_weights(gcfg, gc["wseed"]))
    assert enc4.walks(gc["B"], gc["S"])
    np.testing.assert_allclose(enc4.forward(gc["ids"], gc["mask"]), gc["emb"], atol=2e-5, rtol=0)
    assert enc4.walk_stats()["aborts"] == 1 and enc4.walk_stats()["fallbacks"] == 1
    enc4.close()
    # with the default deadline nothing is abandoned and the walker is back after the suspension
    monkeypatch.delenv("MVDB_WALK_DEADLINE_US")
    enc3 = _model(cfg, w)
    for _ in range(300):
        g3 = enc3.forward(ids, mask)
    np.testing.assert_allclose(g3, e64, atol=2e-5, rtol=0)
    assert enc3.walk_stats() == {"aborts": 0, "fallbacks": 0, "suspended_calls": 0}
    enc3.close()


def test_walk_launches_from_two_host_threads_and_two_encoders_do_not_overlap(gpu):
    """Two host threads, each with its own encoder and stream, enqueue walking launches as fast as they can: the per-device
    order (wait for the previous launch's event, launch, record) is taken under ONE lock, so the two threads cannot both
    enqueue behind the same predecessor and put two persistent grids on the device at once (round-5 advisor finding)."""
    import threading
    import torch
    dev = torch.device("cuda", 0)
    cfgs = [E.make_config("xlmr-large-dims"), E.make_config("e5-small-dims")]
    ws = [E.make_weights(cfgs[0], 8), E.make_weights(cfgs[1], 9)]
    encs = [_model(cfgs[0], ws[0]), _model(cfgs[1], ws[1])]
    data, want, outs, errs = [], [], [[], []], []
    for cfg, w in zip(cfgs, ws):
        ids, mask = E.make_inputs(cfg, 1, 20, 19)
        data.append((torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev), ids, mask))
        want.append(E.numpy_forward(cfg, w, ids, mask)[1])
    torch.cuda.synchronize()

    def run(i):
        try:
            st = torch.cuda.Stream(dev)
            with torch.cuda.stream(st):
                for rep in range(60):
                    outs[i].append(encs[i].forward_device(data[i][0], data[i][1])[0])
                    if rep % 10 == 9:    # and the host entry in between (its own stream, waits for its launch)
                        outs[i].append(torch.from_numpy(encs[i].forward(data[i][2], data[i][3])).to(dev))
            st.synchronize()
        except Exception as ex:  # noqa: BLE001
            errs.append(ex)

    th = [threading.Thread(target=run, args=(i,)) for i in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
        assert not t.is_alive(), "walking launches of two threads wedged"
    assert not errs, errs
    torch.cuda.synchronize()
    for i in (0, 1):
        for o in outs[i]:
            np.testing.assert_allclose(o.cpu().numpy(), want[i], atol=2e-5, rtol=0)
        assert encs[i].walk_stats()["aborts"] == 0
        encs[i].close()


_CHILD = r"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, {root!r})
from oracle import encoder as E
from minivectordb_amd.embedding_model import GpuEncoder
cfg = E.make_config("e5-small-dims")
w = E.make_weights(cfg, 61)
enc = GpuEncoder(cfg, {{k: torch.from_numpy(v) for k, v in w.items()}}, device=0)
rs = np.random.RandomState(62)
lens = rs.randint(4, 61, size={n})
sents = [rs.randint(5, cfg["vocab_size"], size=(1, int(L))).astype(np.int32) for L in lens]
enc.forward(sents[0], np.ones_like(sents[0]))
open({ready!r} + str(os.getpid()), "w").close()
while len([f for f in os.listdir(os.path.dirname({ready!r})) if f.startswith(os.path.basename({ready!r}))]) < {procs}:
    time.sleep(0.01)
t0 = time.perf_counter()
out = np.concatenate([enc.forward(s, np.ones_like(s)) for s in sents])
dt = time.perf_counter() - t0
np.save({out!r} + str(os.getpid()) + ".npy", out)
print(json.dumps({{"pid": os.getpid(), "seconds": dt, **enc.walk_stats()}}))
"""


def test_two_processes_embedding_on_one_gpu_both_finish_with_identical_results(gpu, tmp_path):
    """The ordinary way to serve a vector database: two worker processes, one GPU, each embedding one sentence per call.  Their
    walking launches are persistent grids; the advisory lock keyed by the GPU's UUID keeps two of them from being in flight
    together, the bounded waits keep a launch that cannot complete from wedging the GPU.  Both children embed the same 2,000
    sentences concurrently: both finish, and both return bit for bit what a process alone returns."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def spawn(tag, procs, n=2000):
        ready, out = str(tmp_path / f"ready_{tag}_"), str(tmp_path / f"out_{tag}_")
        code = _CHILD.format(root=root, n=n, ready=ready, procs=procs, out=out)
        ps = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(procs)]
        res = []
        for p in ps:
            try:
                so, se = p.communicate(timeout=600)
            except subprocess.TimeoutExpired:
                for q in ps:
                    q.kill()
                raise AssertionError("a child embedding on a shared GPU did not finish")
            assert p.returncode == 0, se[-2000:]
            res.append(json.loads(so.strip().splitlines()[-1]))
        return res, [np.load(out + str(r["pid"]) + ".npy") for r in res]

    alone, ref = spawn("alone", 1)
    both, outs = spawn("both", 2)
    for o in outs:
        assert np.array_equal(o, ref[0])
    print("one process alone:", alone, "two sharing the GPU:", both)


def test_walk_shape_under_stream_capture_takes_the_capturable_kernels(gpu):
    """A walking launch cannot take part in the per-device ordering while the caller captures the stream (round-5 advisor
    finding): a captured forward of a walk-sized shape is recorded as the per-op kernels, and replays correctly next to eager
    walking launches of another encoder."""
    import torch
    dev = torch.device("cuda", 0)
    cfg = E.make_config("e5-small-dims")
    w = E.make_weights(cfg, 51)
    ids, mask = E.make_inputs(cfg, 1, 30, 52)
    _, e64 = E.numpy_forward(cfg, w, ids, mask)
    enc, other = _model(cfg, w), _model(cfg, w)
    ti, tm = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
    assert enc.walks(1, 30)
    st = torch.cuda.Stream(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        # allocation is not capturable: size the per-op workspace (and build the fp16 weight images) with an eager batch that
        # does not walk and holds more tokens
        big_ids, big_mask = E.make_inputs(cfg, 40, 30, 53)
        enc.forward_device(torch.from_numpy(big_ids).to(dev), torch.from_numpy(big_mask).to(dev))
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
            out, _ = enc.forward_device(ti, tm)
    for _ in range(3):
        g.replay()
        other.forward(ids, mask)     # eager walking launches of another encoder meanwhile
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), e64, atol=2e-5, rtol=0)
    assert enc.walk_stats()["aborts"] == 0 and other.walk_stats()["aborts"] == 0
    enc.close()
    other.close()


@pytest.mark.parametrize("deadline_us", [None, 1000, 200])
def test_walk_under_cu_pressure_from_another_stream_never_hangs(gpu, monkeypatch, deadline_us):
    """The walking launch needs all of its workgroups resident.  Here a foreign stream keeps every CU busy with large GEMMs
    (torch.mm, ~10 ms each, queued back to back) while sentences are embedded one per call: whether the launch gets its CUs in
    time or its bounded waits abandon it and the per-op kernels answer, every call returns the right embedding — and returns.
    With the deadline cut to 1 ms / 0.2 ms the waits do run out in the middle of a forward, at whatever phase the starved
    workgroups have reached (the deadline-0 test above abandons at the very first wait only)."""
    import time
    import torch
    if deadline_us is not None:
        monkeypatch.setenv("MVDB_WALK_DEADLINE_US", str(deadline_us))
    dev = torch.device("cuda", 0)
    cfg = E.make_config("e5-small-dims")
    w = E.make_weights(cfg, 91)
    enc = _model(cfg, w)
    ids, mask = E.make_inputs(cfg, 1, 48, 92)
    _, e64 = E.numpy_forward(cfg, w, ids, mask)
    a = torch.randn((8192, 8192), device=dev)
    b = torch.randn((8192, 8192), device=dev)
    side = torch.cuda.Stream(dev)
    enc.forward(ids, mask)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lat = []
    for rep in range(12):
        with torch.cuda.stream(side):
            for _ in range(4):
                torch.mm(a, b)             # ~40 ms of every CU on the other stream
        t1 = time.perf_counter()
        got = enc.forward(ids, mask)
        lat.append(time.perf_counter() - t1)
        np.testing.assert_allclose(got, e64, atol=2e-5, rtol=0)
    torch.cuda.synchronize()
    st = enc.walk_stats()
    print(f"under CU pressure (deadline {deadline_us or 20000} us): 12 forwards in {time.perf_counter() - t0:.2f} s, slowest {max(lat) * 1e3:.1f} ms, walker stats {st}")
    assert max(lat) < 5.0
    enc.close()
