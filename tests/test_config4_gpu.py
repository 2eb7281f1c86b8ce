"""BASELINE config 4 under -m gpu: "80M x 512 fp32 sharded across 8 x MI355X, RCCL all-gather top-k, k = 10".
The driver's fresh box has ONE GPU, so two halves are checked here:

  * the whole 80M x 512 corpus (163.84 GB of the 288 GB) resident on one GPU — every row of the config is scanned
    by every search path, needles planted past row 80,000,000 must come back first, nq = 1 / 32 / 128 must agree;
  * bench.py's N > 1 code path (local scan -> all-gather of packed top-k -> k-way merge kernel -> global q/s) as
    two child processes sharing the GPU over gloo (RCCL refuses two ranks on one device), merged ids compared with
    a single index holding both shards.

The 8-GPU run itself is the driver's (SCALE_rNN.json).  Reference contract: top-k of the union of all shards
(minivectordb/sharded_vector_database.py:598-662)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import flat

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config4_whole_corpus_80m_x_512_on_one_gpu(gpu):
    from minivectordb_amd import _native as native
    n, d, k, nq = 80_000_000, 512, 10, 128
    idx = native.FlatIndex(d, device=0)
    idx.reserve(n + nq)     # one allocation: growing by reallocation would need old + new side by side
    for first in range(0, n, 10_000_000):   # the 8 shards of the config, back to back
        idx.add_synthetic(10_000_000, 1234, first_row=first, normalize=True)
    q = flat.synth(nq, d, 5678)
    flat.normalize_l2(q)
    idx.add(q, normalize=True)              # needles: rows n .. n+nq-1 hold the queries themselves
    assert idx.ntotal == n + nq
    want = n + np.arange(nq)
    reruns = native.split_rerun_count()
    results = {}
    for label, qs in (("batch128", q), ("batch32", q[:32]), ("single", q[:1])):
        D, I = idx.search(qs, k)
        assert np.array_equal(I[:, 0], want[:len(qs)]), (label, I[:4, :3])
        assert np.allclose(D[:, 0], 1.0, atol=1e-5)
        assert (I >= 0).all() and (I < n + nq).all() and (np.diff(D, axis=1) <= 0).all()
        for i in (0, len(qs) - 1):          # returned scores == float64 dot products of the rows fetched back
            rows = np.stack([idx.get_rows(int(r), 1)[0] for r in I[i]])
            ref = rows.astype(np.float64) @ qs[i].astype(np.float64)
            assert np.abs(ref - D[i]).max() <= 1e-4, (label, i)
        results[label] = (D, I)
    assert native.split_rerun_count() == reruns
    # the three paths (split-precision 128, split-precision 32, exact fp32 GEMV) agree id for id
    assert np.array_equal(results["batch128"][1][:32], results["batch32"][1])
    assert np.array_equal(results["batch128"][1][:1], results["single"][1])
    np.testing.assert_allclose(results["batch128"][0][:32], results["batch32"][0], atol=2e-6, rtol=0)
    # rows from every one of the 8 shards show up among the results (the scan really covers 80M rows)
    shards = set((results["batch128"][1][:, 1:] // 10_000_000).ravel().tolist())
    assert shards >= set(range(8)), shards
    # 100 fresh queries against the CPU oracle over ALL 80M rows ("100 at 80M"), streamed to the host in
    # 1M-row blocks: the single-query scan and one 100-query call (tests/bigcheck.py)
    import bigcheck
    q100 = flat.synth(100, d, 5678, first_row=4096)
    flat.normalize_l2(q100)
    (oracle,), cost = bigcheck.oracle_topk_streamed(idx, n + nq, q100, k)
    single = [idx.search(q100[i], k) for i in range(100)]
    D1, I1 = np.concatenate([s[0] for s in single]), np.concatenate([s[1] for s in single])
    bigcheck.report(dict(bigcheck.compare(idx, q100, D1, I1, *oracle, "config4 80M x 512 on one GPU, 1 query per call"),
                         oracle_cost=cost))
    Db, Ib = idx.search(q100, k)
    bigcheck.report(bigcheck.compare(idx, q100, Db, Ib, *oracle, "config4 80M x 512 on one GPU, 100 queries per call"))
    # deleting an EARLY row at this size: the 164 GB tail moves up one row, in place, through the 512 MiB staging buffer
    # (until round 4 the compaction wanted a temporary as large as the tail, which this GPU does not have left)
    probe = [7, 8, 40_000_000, n + nq - 2]
    before = {r: idx.get_rows(r, 1)[0].copy() for r in probe + [6, n + nq - 1]}
    idx.remove_rows(np.array([7], dtype=np.int64))
    assert idx.ntotal == n + nq - 1
    assert idx.get_rows(6, 1)[0].tobytes() == before[6].tobytes()
    assert idx.get_rows(7, 1)[0].tobytes() == before[8].tobytes()
    assert idx.get_rows(n + nq - 2, 1)[0].tobytes() == before[n + nq - 1].tobytes()
    D, I = idx.search(q[:3], k)
    assert np.array_equal(I[:, 0], want[:3] - 1) and np.allclose(D[:, 0], 1.0, atol=1e-5)
    D, I = idx.search(q[:1], k)
    assert I[0, 0] == n - 1
    idx.close()


def _two_rank_bench(tmp_path, rows, steps, warmup, d=512, k=10, dump=None, **extra_env):
    """bench.py --gpus 2 as two ranks sharing GPU 0 over gloo; returns (returncode, stdout, stderr)."""
    env = dict(os.environ, MVDB_BENCH_SHARE_GPU="1", MVDB_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    port = 23000 + (os.getpid() + len(extra_env) * 131 + rows) % 4000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(steps),
           "--warmup", str(warmup), "--rows", str(rows), "--dim", str(d), "--k", str(k)]
    if dump:
        cmd += ["--dump", dump]
    # own session + faulthandler: if the two ranks ever hang, SIGABRT makes every Python process of the job print where
    import signal
    env["PYTHONFAULTHANDLER"] = "1"
    proc = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                            start_new_session=True)
    try:
        so, se = proc.communicate(timeout=int(os.environ.get("MVDB_TEST_SUBPROCESS_TIMEOUT", "600")))
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGABRT)
        try:
            so, se = proc.communicate(timeout=20)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)
            so, se = proc.communicate()
        pytest.fail("two-rank bench.py hung; tracebacks of its processes:\n" + so[-1500:] + se[-6000:])
    return proc.returncode, so, se


def test_two_rank_bench_survives_a_failed_native_route(gpu, tmp_path):
    """bench.py --gpus N when the in-library RCCL route cannot be brought up (injected: this box has one GPU): the run falls
    back, loudly, to torch.distributed's all-gather and still prints ONE bench line, which names the failure; with
    MVDB_BENCH_REQUIRE_NATIVE=1 it prints ONE JSON error line instead and exits 3."""
    rc, so, se = _two_rank_bench(tmp_path, 200_000, 6, 2, MVDB_BENCH_TEST_NATIVE_FAILURE="1")
    assert rc == 0, so[-2000:] + se[-4000:]
    lines = [l for l in so.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["collective"].startswith("torch.distributed")
    assert "injected" in out["native_route_error"]
    assert "injected" in se                      # every rank said so on stderr
    rc, so, se = _two_rank_bench(tmp_path, 200_000, 6, 2, MVDB_BENCH_TEST_NATIVE_FAILURE="1", MVDB_BENCH_REQUIRE_NATIVE="1")
    assert rc != 0
    lines = [l for l in so.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (lines, se[-2000:])
    err = json.loads(lines[0])
    assert "injected" in err["error"] and err["rank"] == 0 and err["n_gpus"] == 2 and "hint" in err


def test_config4_two_rank_bench_path_on_a_shared_gpu(gpu, tmp_path):
    from minivectordb_amd import _native as native
    rows, d, k, steps, warmup = 1_000_000, 512, 10, 12, 2
    dump = str(tmp_path / "dump.npz")
    rc, so, se = _two_rank_bench(tmp_path, rows, steps, warmup, d=d, k=k, dump=dump)
    assert rc == 0, so[-2000:] + se[-4000:]
    line = [l for l in so.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["corpus_rows"] == 2 * rows
    # headline = queries/s over the WHOLE corpus (not shard passes summed over ranks)
    assert out["value"] == pytest.approx(steps / (out["ms_per_step"] * steps / 1e3), rel=1e-3)
    assert out["shard_passes_per_s"] == pytest.approx(2 * out["value"], rel=1e-3)
    assert out["roofline"]["peak"] == 16000.0 and len(out["roofline"]["per_rank_avg_launch_ms"]) == 2
    assert out["collective"].startswith("torch.distributed")   # gloo group: the RCCL route is not selectable here
    assert out["native_route_error"] is None
    z = np.load(dump)
    assert int(z["world"]) == 2 and int(z["rows_per_rank"]) == rows
    # the same corpus in ONE index: rank r generated rows [r * rows, (r + 1) * rows) of stream 1234
    idx = native.FlatIndex(d, device=0)
    idx.add_synthetic(rows, 1234, first_row=0, normalize=True)
    idx.add_synthetic(rows, 1234, first_row=rows, normalize=True)
    qall = flat.synth(warmup + steps, d, 5678)
    flat.normalize_l2(qall)
    seen_shards = set()
    for s in range(z["I"].shape[0]):
        D1, I1 = idx.search(qall[warmup + s], k)
        assert np.array_equal(z["I"][s, 0], I1[0]), (s, z["I"][s, 0], I1[0])
        np.testing.assert_allclose(z["D"][s, 0], D1[0], atol=2e-6, rtol=0)
        seen_shards |= set((I1[0] // rows).tolist())
    assert seen_shards == {0, 1}
    idx.close()
