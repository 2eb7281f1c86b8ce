"""GPU tests of the multi-GPU exchange pieces that can run on ONE device (the driver's box): the in-library RCCL
communicator at world = 1 (dlopen binding, ncclCommInitRank, ncclAllGather on a stream), the device-resident subset
search with global labels, and DistributedShardedVectorDatabase's device path against the single-process class on
the same reference-format shard directory (minivectordb/sharded_vector_database.py:598-662)."""
import numpy as np
import pytest

from oracle import flat

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native(gpu):
    from minivectordb_amd import _native
    return _native


def test_native_rccl_communicator_single_rank(native):
    import torch
    dev = torch.device("cuda", 0)
    assert native.lib().mvdb_comm_available() == 0, native.last_error()
    uid = native.Comm.unique_id()
    assert isinstance(uid, bytes) and len(uid) == 128 and any(uid)
    comm = native.Comm(uid, 0, 1, device=0)
    assert native.lib().mvdb_comm_world(comm._h) == 1 and native.lib().mvdb_comm_rank(comm._h) == 0
    src = torch.arange(4096, dtype=torch.uint8, device=dev)
    dst = torch.zeros(4096, dtype=torch.uint8, device=dev)
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        comm.allgather(src.data_ptr(), dst.data_ptr(), 4096, stream=s.cuda_stream)
    s.synchronize()
    assert torch.equal(src, dst)
    with pytest.raises(ValueError):
        native.Comm(uid, 3, 2, device=0)
    comm.close()


def test_subset_search_device_with_global_labels(native):
    import torch
    dev = torch.device("cuda", 0)
    n, d, k, nq = 30000, 384, 10, 3
    x = flat.synth(n, d, 1234)
    flat.normalize_l2(x)
    q = flat.synth(nq, d, 5678)
    idx = native.FlatIndex(d)
    idx.add(x)
    rs = np.random.RandomState(1)
    rows = np.sort(rs.choice(n, 5000, replace=False)).astype(np.int64)
    qn = q.copy()
    flat.normalize_l2(qn)
    Dw, Pw = flat.flat_search(x, qn, k, rows=rows)
    q_t = torch.from_numpy(q).to(dev)
    r_t = torch.from_numpy(rows).to(dev)
    D = torch.empty((nq, k), dtype=torch.float32, device=dev)
    I = torch.empty((nq, k), dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    idx.search_subset_device(q_t.data_ptr(), nq, k, r_t.data_ptr(), rows.size, D.data_ptr(), I.data_ptr(), stream=st,
                             normalize_q=True, map_labels=True, label_offset=7_000_000)
    torch.cuda.synchronize()
    assert np.array_equal(I.cpu().numpy(), rows[Pw] + 7_000_000)
    np.testing.assert_allclose(D.cpu().numpy(), Dw, atol=2e-6, rtol=0)
    # positions (the host entry point's convention) + offset
    idx.search_subset_device(q_t.data_ptr(), nq, k, r_t.data_ptr(), rows.size, D.data_ptr(), I.data_ptr(), stream=st,
                             normalize_q=True, map_labels=False, label_offset=5)
    torch.cuda.synchronize()
    assert np.array_equal(I.cpu().numpy(), Pw + 5)
    # fewer rows than k, and an empty list: faiss' missing-result convention
    few = torch.from_numpy(rows[:4]).to(dev)
    idx.search_subset_device(q_t.data_ptr(), nq, k, few.data_ptr(), 4, D.data_ptr(), I.data_ptr(), stream=st,
                             normalize_q=True, map_labels=True, label_offset=100)
    torch.cuda.synchronize()
    Ih = I.cpu().numpy()
    assert (Ih[:, 4:] == -1).all() and set(Ih[0, :4].tolist()) == set((rows[:4] + 100).tolist())
    idx.search_subset_device(q_t.data_ptr(), nq, k, 0, 0, D.data_ptr(), I.data_ptr(), stream=st, map_labels=True)
    torch.cuda.synchronize()
    assert (I.cpu().numpy() == -1).all() and (D.cpu().numpy() < -3e38).all()
    idx.close()


def test_distributed_database_device_path_equals_single_process(native, tmp_path):
    """world = 1 on the GPU: same answers as ShardedVectorDatabase for full, filtered, k > 64 and k > rows searches
    (the N > 1 plumbing around it is covered by tests/test_distributed_gloo.py)."""
    import torch
    from minivectordb_amd import ShardedVectorDatabase
    from minivectordb_amd.distributed import DistributedShardedVectorDatabase
    path = str(tmp_path / "shards")
    n, d = 950, 96
    ref = ShardedVectorDatabase(storage_dir=path, shard_size=100)
    x = flat.synth(n, d, 321)
    colours = ["red", "green", "blue"]
    ref.store_embeddings_batch([f"id{i}" for i in range(n)], list(x),
                               [{"bucket": i % 5, "price": i, "colour": colours[i % 3]} for i in range(n)])
    db = DistributedShardedVectorDatabase(path, rank=0, world=1, device=torch.device("cuda", 0))
    assert db.local_rows == n and db.first_row == 0
    queries = [dict(k=5), dict(k=12, metadata_filter={"bucket": 3}),
               dict(k=4, or_filters=[{"colour": "red"}, {"bucket": 1}], exclude_filter={"bucket": 4}),
               dict(k=7, metadata_filter={"price": {"$gte": 400}}, autocut=True), dict(k=3, metadata_filter={"nokey": 1}),
               dict(k=64), dict(k=100), dict(k=999), dict(k=400, metadata_filter={"colour": "green"})]
    q = flat.synth(len(queries), d, 654)
    for i, kw in enumerate(queries):
        got = db.find_most_similar(q[i], **kw)
        want = ref.find_most_similar(q[i], **kw)
        assert list(got[0]) == list(want[0]), (kw, got[0][:5], want[0][:5])
        np.testing.assert_allclose(np.array(got[1], dtype=np.float64), np.array(want[1], dtype=np.float64), atol=2e-6)
        assert list(got[2]) == list(want[2])


def test_native_rccl_route_through_the_distributed_database(native, tmp_path):
    """The WHOLE N-GPU exchange path on the one-GPU box: a one-rank RCCL process group, `Collective(want="native")`
    brought up the way an 8-rank job brings it up (librccl bound by dlopen, agreement all-reduces, unique id through
    broadcast_object_list, ncclCommInitRank, probe gather), then every `find_most_similar` of
    DistributedShardedVectorDatabase runs local scan -> ncclAllGather (mvdb_allgather_topk) -> merge kernel
    (`exchange_always`).  Results must equal the single-process class; a filter's local rows go up once."""
    import os
    import torch
    import torch.distributed as dist
    from minivectordb_amd import ShardedVectorDatabase
    from minivectordb_amd.distributed import Collective, DistributedShardedVectorDatabase
    path = str(tmp_path / "shards")
    n, d = 2000, 128
    ref = ShardedVectorDatabase(storage_dir=path, shard_size=256)
    x = flat.synth(n, d, 77)
    ref.store_embeddings_batch([f"id{i}" for i in range(n)], list(x),
                               [{"bucket": i % 7, "price": i, "tag": "even" if i % 2 == 0 else "odd"} for i in range(n)])
    port = 24000 + os.getpid() % 3000
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        db = DistributedShardedVectorDatabase(path, device=torch.device("cuda", 0), collective="native",
                                              exchange_always=True)
        assert db._collective.name == Collective.NATIVE and db._collective.comm is not None
        queries = [dict(k=5), dict(k=10, metadata_filter={"bucket": 3}), dict(k=10, exclude_filter={"tag": "odd"}),
                   dict(k=6, or_filters=[{"bucket": 1}, {"bucket": 2}], exclude_filter={"price": 8}),
                   dict(k=70), dict(k=300, metadata_filter={"tag": "even"}), dict(k=3, metadata_filter={"nokey": 1}),
                   dict(k=9, metadata_filter={"price": {"$lt": 100}}, autocut=True)]
        q = flat.synth(len(queries), d, 654)
        for rounds in range(2):   # the second round is answered from the resident row sets
            for i, kw in enumerate(queries):
                got = db.find_most_similar(q[i], **kw)
                want = ref.find_most_similar(q[i], **kw)
                assert list(got[0]) == list(want[0]), (kw, got[0][:5], want[0][:5])
                np.testing.assert_allclose(np.array(got[1], dtype=np.float64), np.array(want[1], dtype=np.float64),
                                           atol=2e-6)
                assert list(got[2]) == list(want[2])
            if rounds == 0:
                built = db.rowsets_built
                assert built >= 5
        assert db.rowsets_built == built
        # several queries in one collective call: element j is the single-query answer
        for kw in queries:
            many = db.find_most_similar_batch(q[:5], **kw)
            assert len(many) == 5
            for j in range(5):
                want = ref.find_most_similar(q[j], **kw)
                assert list(many[j][0]) == list(want[0]), (kw, j)
                np.testing.assert_allclose(np.array(many[j][1], dtype=np.float64), np.array(want[1], dtype=np.float64), atol=2e-6)
                assert list(many[j][2]) == list(want[2])
        assert db._searcher(5).collective.startswith("ncclAllGather")
        db.close()
    finally:
        dist.destroy_process_group()
