"""world_size-2 gloo test (CPU) of the row-partitioned search plumbing in
minivectordb_amd/distributed.py: shard ranges, label offsets, the packed all-gather layout and the
gather order.  The per-shard scan and the k-way merge are HIP kernels and cannot run here, so both
are injected: the local search is the CPU oracle over the rank's row range, the merge a numpy
restatement of mvdb_merge_topk_device.  The result on every rank must equal the oracle's top-k over
the UNION of the shards (the reference's contract: it stacks all shards into one index,
minivectordb/sharded_vector_database.py:45-71, :598-662)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, d, k, nq, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from minivectordb_amd.distributed import PackedTopK, ShardedSearcher, shard_ranges
    from oracle import flat

    x = flat.synth(n, d, 1234)
    flat.normalize_l2(x)
    q = flat.synth(nq, d, 5678)
    flat.normalize_l2(q)
    first, cnt = shard_ranges(n, world)[rank]
    shard = np.ascontiguousarray(x[first:first + cnt])

    def local_search(qt, D, I, label_offset):
        Dl, Il = flat.flat_search(shard, qt.numpy(), k)
        D.copy_(torch.from_numpy(Dl))
        I.copy_(torch.from_numpy(np.where(Il >= 0, Il + label_offset, -1)))

    def merge(gathered, D_out, I_out):
        for qi in range(gathered.nq):
            cands = []
            for l in range(gathered.world):
                Dl, Il = gathered.views(l)
                for j in range(k):
                    if int(Il[qi, j]) >= 0:
                        cands.append((-float(Dl[qi, j]), int(Il[qi, j])))
            cands.sort()
            for j in range(k):
                if j < len(cands):
                    D_out[qi, j] = -cands[j][0]
                    I_out[qi, j] = cands[j][1]
                else:
                    D_out[qi, j] = -3.4028234663852886e38
                    I_out[qi, j] = -1

    s = ShardedSearcher(None, k, rank=rank, world=world, label_offset=first, device=torch.device("cpu"),
                        local_search=local_search, merge=merge)
    D, I = s.search_device(torch.from_numpy(q))
    np.save(os.path.join(out_dir, f"D{rank}.npy"), D.numpy())
    np.save(os.path.join(out_dir, f"I{rank}.npy"), I.numpy())
    # layout facts the HIP merge kernel relies on
    p = PackedTopK(nq, k, torch.device("cpu"), world)
    assert p.nbytes % 16 == 0 and p.stride_I * 8 == p.nbytes and p.stride_D * 4 == p.nbytes
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,k,nq", [(1001, 10, 3), (64, 5, 1)])
def test_two_rank_sharded_search_equals_union(tmp_path, n, k, nq):
    from oracle import flat
    world, d = 2, 64
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, n, d, k, nq, str(tmp_path)), nprocs=world, join=True)
    x = flat.synth(n, d, 1234)
    flat.normalize_l2(x)
    q = flat.synth(nq, d, 5678)
    flat.normalize_l2(q)
    Dw, Iw = flat.flat_search(x, q, k)
    for r in range(world):
        D = np.load(tmp_path / f"D{r}.npy")
        I = np.load(tmp_path / f"I{r}.npy")
        assert np.array_equal(I, Iw), (r, I, Iw)
        assert np.array_equal(D, Dw)


def test_shard_ranges_and_files(tmp_path):
    from minivectordb_amd.distributed import shard_ranges
    from minivectordb_amd.sharded_vector_database import shard_files_for_rank
    assert shard_ranges(10, 3) == [(0, 4), (4, 3), (7, 3)]
    assert shard_ranges(80_000_000, 8)[7] == (70_000_000, 10_000_000)
    for i in (0, 1, 2, 10, 3):
        (tmp_path / f"shard_{i}.pkl").write_bytes(b"")
    got = [shard_files_for_rank(str(tmp_path), r, 2) for r in range(2)]
    assert got == [["shard_0.pkl", "shard_1.pkl", "shard_2.pkl"], ["shard_3.pkl", "shard_10.pkl"]]


# ---- DistributedShardedVectorDatabase: SPMD find_most_similar over a reference-format db_shards/ -------
QUERIES = [
    dict(k=5),
    dict(k=12, metadata_filter={"bucket": 3}),
    dict(k=4, or_filters=[{"colour": "red"}, {"bucket": 1}], exclude_filter={"bucket": 4}),
    dict(k=7, metadata_filter={"price": {"$gte": 40}}, autocut=True),
    dict(k=3, metadata_filter={"nokey": 1}),
    dict(k=64),
    dict(k=90),                                        # > 64 results and > rows per shard: the sorted-merge path
    dict(k=80, metadata_filter={"colour": "green"}),   # more than the filter passes
]


def _np_merge(gathered, D_out, I_out):
    k = gathered.k
    for qi in range(gathered.nq):
        cands = []
        for l in range(gathered.world):
            Dl, Il = gathered.views(l)
            cands += [(-float(Dl[qi, j]), int(Il[qi, j])) for j in range(k) if int(Il[qi, j]) >= 0]
        cands.sort()
        for j in range(k):
            D_out[qi, j] = -cands[j][0] if j < len(cands) else -3.4028234663852886e38
            I_out[qi, j] = cands[j][1] if j < len(cands) else -1


def _build_shards(path, monkeypatch, n=95, d=48):
    from minivectordb_amd import ShardedVectorDatabase, _native
    from oracle import flat
    from oracle_backend import OracleIndex
    monkeypatch.setattr(_native, "FlatIndex", OracleIndex)  # undone at the end of THIS test: the stand-in never outlives it
    db = ShardedVectorDatabase(storage_dir=path, shard_size=10)
    x = flat.synth(n, d, 321)
    colours = ["red", "green", "blue"]
    db.store_embeddings_batch([f"id{i}" for i in range(n)], list(x),
                              [{"bucket": i % 5, "price": i, "colour": colours[i % 3]} for i in range(n)])
    return db


def _dist_worker(rank, world, port, path, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pickle
    from minivectordb_amd.distributed import DistributedShardedVectorDatabase
    from oracle import flat
    from oracle_backend import OracleIndex
    holder = {}

    def local_search(q, D, I, label_offset, rows=None, normalize_q=False):
        # CPU stand-in of mvdb_index_search_device / mvdb_index_search_subset_device(map_labels=1): same contract
        index, k = holder["db"].index, D.shape[1]
        miss_d, miss_i = np.float32(-3.4028234663852886e38), -1
        Dl = np.full((q.shape[0], k), miss_d, np.float32)
        Il = np.full((q.shape[0], k), miss_i, np.int64)
        if rows is None:
            kk = min(k, index.ntotal)
            if kk:
                d_, i_ = index.search(q.numpy(), kk, normalize_q=normalize_q)
                Dl[:, :kk], Il[:, :kk] = d_, np.where(i_ >= 0, i_ + label_offset, -1)
        elif len(rows):
            r = rows.numpy() if torch.is_tensor(rows) else rows.rows   # a device row list, or the index's resident row set
            kk = min(k, r.size)
            d_, p_ = index.search_subset(q.numpy(), kk, r, normalize_q=normalize_q)
            Dl[:, :kk], Il[:, :kk] = d_, np.where(p_ >= 0, r[np.maximum(p_, 0)] + label_offset, -1)
        D.copy_(torch.from_numpy(Dl))
        I.copy_(torch.from_numpy(Il))

    db = DistributedShardedVectorDatabase(path, device=torch.device("cpu"), index_factory=OracleIndex,
                                          local_search=local_search, merge=_np_merge)
    holder["db"] = db
    assert db._searcher(5).collective == "torch.distributed.all_gather_into_tensor"
    assert db.world == world and db.local_rows > 0
    # a rank unpickles ONLY its own shard files; the bookkeeping of the others arrives by all_gather_object
    from minivectordb_amd.sharded_vector_database import shard_files_for_rank
    assert db.files_opened == shard_files_for_rank(path, rank, world) and len(db.files_opened) == 5
    assert len(db.unique_ids) == 95 and len(db.metadata) == 95
    # every per-k searcher shares the database's ONE exchange route
    assert db._searcher(7)._collective is db._searcher(5)._collective is db._collective
    # an EXPLICIT request for the native RCCL route must raise where it cannot be brought up (gloo group, CPU
    # tensors) instead of silently measuring the torch route — on every rank together, no rank left in a collective
    from minivectordb_amd.distributed import Collective
    with pytest.raises(RuntimeError, match="MVDB_COLLECTIVE=native"):
        Collective(rank, world, torch.device("cpu"), want="native")
    assert Collective(rank, world, torch.device("cpu")).name == Collective.TORCH
    q = flat.synth(len(QUERIES), 48, 654)
    res = []
    for i, kw in enumerate(QUERIES):
        ids, dists, metas = db.find_most_similar(q[i], **kw)
        res.append((list(ids), [float(v) for v in dists], list(metas)))
    # a repeated filter evaluates nothing and uploads nothing: its local rows are a resident row set of this rank's index
    built, uploads = db.rowsets_built, [c for c in db.index.calls if c[0] == "rowset"]
    assert built == len(uploads) >= 3
    for i, kw in enumerate(QUERIES):
        again = db.find_most_similar(q[i], **kw)
        assert list(again[0]) == res[i][0]
    assert db.rowsets_built == built and [c for c in db.index.calls if c[0] == "rowset"] == uploads
    # the batch form: every rank passes the same queries, element j is the single-query answer under that filter
    for i, kw in enumerate(QUERIES):
        many = db.find_most_similar_batch(q[:4], **kw)
        assert len(many) == 4
        assert list(many[i][0]) == res[i][0] if i < 4 else True
        for j in range(4):
            one = db.find_most_similar(q[j], **kw)
            assert list(many[j][0]) == list(one[0]) and list(many[j][2]) == list(one[2]), (kw, j)
    with open(os.path.join(out_dir, f"res{rank}.pkl"), "wb") as f:
        pickle.dump((res, db.first_row, db.local_rows), f)
    db.close()
    dist.barrier()
    dist.destroy_process_group()


def test_distributed_database_equals_single_process(tmp_path, monkeypatch):
    import pickle
    from oracle import flat
    path = str(tmp_path / "shards")
    ref = _build_shards(path, monkeypatch)
    world = 2
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_dist_worker, args=(world, port, path, str(tmp_path)), nprocs=world, join=True)
    q = flat.synth(len(QUERIES), 48, 654)
    want = []
    for i, kw in enumerate(QUERIES):
        ids, dists, metas = ref.find_most_similar(q[i], **kw)
        want.append((list(ids), [float(v) for v in dists], list(metas)))
    spans = []
    for r in range(world):
        res, first, count = pickle.load(open(tmp_path / f"res{r}.pkl", "rb"))
        spans.append((first, count))
        for got, exp in zip(res, want):
            assert got[0] == exp[0]
            np.testing.assert_allclose(got[1], exp[1], atol=1e-6)
            assert got[2] == exp[2]
    assert spans[0] == (0, 50) and spans[1] == (50, 45)  # whole shard files per rank: 5 + 5 files of 10 rows
