"""The threaded store / search / delete stress of tests/test_threads_and_rerank.py (shaped like the reference's
tests/test_multithreaded_operations.py and tests/test_sharded_multithreaded_operations.py: 5 writers, 5 searchers — plain and
filtered —, 1 deleter) with the REAL device index behind the drop-in classes: the index's reader / writer lock, the mutators'
own stream, the write-generation stamps of the resident row sets and the coalesced uploads all run under contention."""
import pytest

from test_threads_and_rerank import _stress

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("round_", range(3))
def test_threaded_store_search_delete_flat_on_the_device(tmp_path, gpu, round_):
    from minivectordb_amd import VectorDatabase
    db = VectorDatabase(storage_file=str(tmp_path / "t.pkl"))
    _stress(db, db.delete_embedding)
    assert db.id_map == {row: uid for uid, row in db.inverse_id_map.items()}


@pytest.mark.parametrize("round_", range(2))
def test_threaded_store_search_delete_sharded_on_the_device(tmp_path, gpu, round_):
    from minivectordb_amd import ShardedVectorDatabase
    db = ShardedVectorDatabase(storage_dir=str(tmp_path / "shards"), shard_size=77)
    _stress(db, lambda uid: db.delete_embeddings_batch([uid]))
    assert [db.unique_ids[r] for r in range(len(db.unique_ids))] == sorted(db.inverse_id_map, key=db.inverse_id_map.get)
    db2 = ShardedVectorDatabase(storage_dir=str(tmp_path / "shards"), shard_size=77)
    assert sorted(db2.unique_ids) == sorted(db.unique_ids)


def test_find_most_similar_batch_on_the_device(tmp_path, gpu):
    """find_most_similar_batch through the real index at a size where 8+ queries take the certified batch passes over the fp16
    shadow (120k x 256), plain and under a metadata filter (the filter's resident row set): per query the ids of
    find_most_similar, distances within 2e-6 (a batch pass returns fp32 re-scores, the single query the scan's sums)."""
    import numpy as np
    from oracle import flat
    from minivectordb_amd import VectorDatabase
    n, d = 120_000, 256
    x = flat.synth(n, d, 91)
    q = flat.synth(40, d, 92)
    db = VectorDatabase(storage_file=str(tmp_path / "b.pkl"))
    db.store_embeddings_batch(list(range(n)), x, [{"bucket": i % 4} for i in range(n)])
    for kwargs in ({"k": 10}, {"k": 10, "metadata_filter": {"bucket": 1}}, {"k": 5, "exclude_filter": {"bucket": 0}}):
        many = db.find_most_similar_batch(q, **kwargs)
        assert len(many) == 40
        for i in (0, 7, 19, 39):
            one = db.find_most_similar(q[i], **kwargs)
            assert list(many[i][0]) == list(one[0]), (kwargs, i)
            np.testing.assert_allclose(np.asarray(many[i][1]), np.asarray(one[1]), atol=2e-6, rtol=0)
            assert list(many[i][2]) == list(one[2])
    assert db.index.shadow_rows == n          # the unfiltered batch went through the shadow pass
