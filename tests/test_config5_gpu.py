"""BASELINE config 5 under -m gpu: "e5-small embedding forward (batch 256 sentences) on MI355X MFMA + 10M x 384 kNN
end-to-end".  The 256-sentence batch is a committed golden (transformers' BertModel on CPU, tests/golden/
make_encoder_golden.py); the embeddings never leave the device between mvdb_encoder_forward_device and
mvdb_index_search_device; the corpus is the synthetic stream (10M x 384, generated on the device) with copies of
8 of the embeddings planted behind it.  Reference path: minivectordb/embedding_model.py:62-71 ->
vector_database.py:473-497."""
import numpy as np
import pytest

import bigcheck
from encoder_cases import load_cases
from oracle import encoder as E

pytestmark = pytest.mark.gpu


def test_config5_encoder_batch256_then_knn_10m_x_384(gpu):
    import torch
    from minivectordb_amd import _native as native
    from minivectordb_amd.embedding_model import GpuEncoder

    case = next(c for c in load_cases() if c["B"] == 256)
    assert case["name"] == "e5-small-dims" and case["S"] == 32
    cfg = E.make_config(case["name"])
    w = E.make_weights(cfg, case["wseed"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(dev))
    enc = GpuEncoder(cfg, {k: torch.from_numpy(v) for k, v in w.items()}, device=0)
    ids = torch.from_numpy(case["ids"]).to(dev)
    mask = torch.from_numpy(case["mask"]).to(dev)
    emb, _ = enc.forward_device(ids, mask)
    torch.cuda.synchronize()
    emb_host = emb.cpu().numpy()
    np.testing.assert_allclose(emb_host, case["emb"], atol=2e-5, rtol=0)      # vs transformers, all 256 rows

    n, d, k, B = 10_000_000, 384, 10, 256
    idx = native.FlatIndex(d, device=0)
    idx.reserve(n + 8)
    idx.add_synthetic(n, 1234, normalize=True)
    needles = [0, 1, 31, 100, 127, 128, 200, 255]
    idx.add(emb_host[needles], normalize=True)                                 # rows n .. n+7
    assert idx.ntotal == n + 8

    D = torch.empty((B, k), dtype=torch.float32, device=dev)
    I = torch.empty((B, k), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    reruns = native.split_rerun_count()
    native.prof_enable(True)
    try:
        for name in ("ip_scan_split_seed", "ip_scan_split", "ip_scan_half_seed", "ip_scan_half"):
            native.prof_read(name)
        # the chain: encoder output tensor -> search, same stream, nothing touches the host in between
        emb2, _ = enc.forward_device(ids, mask)
        idx.search_device(emb2.data_ptr(), B, k, D.data_ptr(), I.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        assert native.prof_read("ip_scan_half_seed")[0] == 1, "256 queries = ONE certified fp16-nomination pass over the corpus"
    finally:
        native.prof_enable(False)
    assert native.split_rerun_count() == reruns, "random corpus: every query must certify"
    Dh, Ih = D.cpu().numpy(), I.cpu().numpy()
    for j, b in enumerate(needles):                                            # needles first
        assert Ih[b, 0] == n + j, (b, Ih[b, :3], Dh[b, :3])
        assert abs(Dh[b, 0] - 1.0) < 1e-5
    assert (Ih >= 0).all() and (np.diff(Dh, axis=1) <= 0).all()
    for b in range(B):
        assert len(set(Ih[b].tolist())) == k
    # ALL 256 result rows against the CPU oracle streamed over the whole corpus (10M synthetic rows + the 8 needles): id for
    # id, distances within 1e-4, id differences adjudicated in float64 (tests/bigcheck.py) — the queries are the embeddings
    # the encoder produced on the device
    (oracle,), cost = bigcheck.oracle_topk_streamed(idx, n + 8, emb_host, k)
    rec = bigcheck.compare(idx, emb_host, Dh, Ih, *oracle, "config5 encoder -> 256-query kNN over 10M x 384 (+ 8 needles), device chain")
    bigcheck.report(dict(rec, oracle_cost=cost))
    # and the batch agrees with the exact fp32 single-query scan on a sample
    for b in [0, 1, 77, 128, 129, 255]:
        D1, I1 = idx.search(emb_host[b], k)
        assert np.array_equal(I1[0], Ih[b]), (b, I1[0], Ih[b])
        np.testing.assert_allclose(D1[0], Dh[b], atol=2e-6, rtol=0)
    # ---- the whole chain as ONE hipGraph: encoder forward -> 256 queries -> certified kNN --------------------------
    # mvdb_index_search_device never reads the device back and the encoder joins an outer capture with plain launches, so
    # the two calls record into one graph; the replay must reproduce the eager ids and scores bit for bit
    cur = torch.cuda.current_stream()
    g = torch.cuda.CUDAGraph()
    emb_g = torch.empty_like(emb2)
    Dg, Ig = torch.empty_like(D), torch.empty_like(I)
    for _ in range(2):   # eager calls on this stream size the workspaces (allocation is not capturable)
        e_, _ = enc.forward_device(ids, mask)
        emb_g.copy_(e_)
        idx.search_device(emb_g.data_ptr(), B, k, Dg.data_ptr(), Ig.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=cur, capture_error_mode="thread_local"):
        e_, _ = enc.forward_device(ids, mask)
        emb_g.copy_(e_)
        idx.search_device(emb_g.data_ptr(), B, k, Dg.data_ptr(), Ig.data_ptr(), stream=stream)
    Dg.zero_()
    Ig.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(Ig.cpu().numpy(), Ih) and np.array_equal(Dg.cpu().numpy(), Dh)
    import time
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(10):
        e_, _ = enc.forward_device(ids, mask)
        idx.search_device(e_.data_ptr(), B, k, Dg.data_ptr(), Ig.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    t_eager = (time.perf_counter() - t0) / 10
    print(f"config 5 (S = 32, 10M x 384): one graph {t_graph * 1e3:.3f} ms, two eager calls {t_eager * 1e3:.3f} ms")
    idx.close()
    enc.close()
