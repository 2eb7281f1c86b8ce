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
