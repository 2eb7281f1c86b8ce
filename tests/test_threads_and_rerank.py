"""Host-logic tests that need no GPU: a threaded store/search/delete stress run shaped like the
reference's tests/test_multithreaded_operations.py (5 writer threads, 5 searcher threads, 1 deleter;
asserts the final bookkeeping), and hybrid_rerank_results.  Device back end = oracle stand-in."""
import threading

import pytest

from oracle import flat
from oracle_backend import OracleIndex


@pytest.fixture
def oracle_backend(monkeypatch):
    from minivectordb_amd import _native
    monkeypatch.setattr(_native, "FlatIndex", OracleIndex)


def _stress(db, delete):
    d, per = 64, 120
    x = flat.synth(5 * per, d, 77)
    errs = []

    def writer(w):
        try:
            for i in range(per):
                uid = w * per + i
                db.store_embedding(uid, x[uid], {"w": w, "i": i})
        except Exception as e:  # pragma: no cover
            errs.append(("w", e))

    def searcher(sidx):
        try:
            q = flat.synth(40, d, 1000 + sidx)
            for i in range(40):
                ids, dist, meta = db.find_most_similar(q[i], k=7)
                assert len(ids) == len(dist) == len(meta) <= 7
                # (like the reference, a search racing a delete may map a row that has just been
                # renumbered, or hold a row list that the delete has shortened — the reference's
                # `self.embeddings[list(filtered)]` raises IndexError there, the device library
                # ValueError; only the shape of successful results is asserted while writers run)
                try:
                    ids, dist, meta = db.find_most_similar(q[i], k=5, metadata_filter={"w": sidx})
                except (ValueError, IndexError):
                    continue
                assert len(ids) == len(dist) == len(meta) <= 5
        except Exception as e:  # pragma: no cover
            errs.append(("s", e))

    deleted = []

    def deleter():
        import time
        try:
            for uid in range(0, 5 * per, 7):
                for _ in range(2000):
                    if uid in db.inverse_id_map:
                        break
                    time.sleep(0.001)
                delete(uid)
                deleted.append(uid)
        except Exception as e:  # pragma: no cover
            errs.append(("d", e))

    ts = [threading.Thread(target=writer, args=(w,)) for w in range(5)]
    ts += [threading.Thread(target=searcher, args=(s,)) for s in range(5)]
    ts += [threading.Thread(target=deleter)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    n = 5 * per - len(deleted)
    assert db.embeddings.shape[0] == n == len(db.metadata) == len(db.inverse_id_map)
    # quiescent again: filters are exact
    for w in range(5):
        ids, dist, meta = db.find_most_similar(x[3], k=9, metadata_filter={"w": w})
        assert len(ids) == 9 and all(m["w"] == w for m in meta)
    # every surviving id is findable, at its own row
    ids, dist, meta = db.find_most_similar(x[3], k=1)
    assert ids[0] == 3 and abs(float(dist[0]) - 1.0) < 1e-5
    for uid in deleted:
        assert uid not in db.inverse_id_map
    rows = sorted(db.inverse_id_map.values())
    assert rows == list(range(n))


def test_threaded_store_search_delete_flat(tmp_path, oracle_backend):
    from minivectordb_amd import VectorDatabase
    db = VectorDatabase(storage_file=str(tmp_path / "t.pkl"))
    _stress(db, db.delete_embedding)
    assert db.id_map == {row: uid for uid, row in db.inverse_id_map.items()}


def test_threaded_store_search_delete_sharded(tmp_path, oracle_backend):
    from minivectordb_amd import ShardedVectorDatabase
    db = ShardedVectorDatabase(storage_dir=str(tmp_path / "shards"), shard_size=77)
    _stress(db, lambda uid: db.delete_embeddings_batch([uid]))
    assert [db.unique_ids[r] for r in range(len(db.unique_ids))] == sorted(
        db.inverse_id_map, key=db.inverse_id_map.get)
    db2 = ShardedVectorDatabase(storage_dir=str(tmp_path / "shards"), shard_size=77)
    assert sorted(db2.unique_ids) == sorted(db.unique_ids)


def test_hybrid_rerank_results(tmp_path, oracle_backend):
    from minivectordb_amd import VectorDatabase
    db = VectorDatabase(storage_file=str(tmp_path / "h.pkl"))
    sentences = ["i like dogs", "the stock market fell", "dogs are animals", "cats and dogs", "quantum physics"]
    scores = [0.9, 0.2, 0.7, 0.6, 0.1]
    s, c = db.hybrid_rerank_results(sentences, scores, "dogs", k=3)
    assert len(s) == len(c) == 3 and s[0] == "i like dogs"
    assert set(s) <= set(sentences)
    s0, c0 = db.hybrid_rerank_results([], [], "dogs", k=3)
    assert list(s0) == [] and list(c0) == []
    # weights are honoured: all weight on the search score reproduces the search order
    s2, c2 = db.hybrid_rerank_results(sentences, scores, "dogs", k=5, weights=(1.0, 0.0, 0.0))
    assert list(s2) == ["i like dogs", "dogs are animals", "cats and dogs", "the stock market fell", "quantum physics"]
    # reference quirk kept on purpose (vector_database.py:429-432): sentences and scores are stacked into
    # ONE string array and the scores are sorted AS STRINGS, so '36.0' ranks above '100.0'
    s3, c3 = db.hybrid_rerank_results(sentences, scores, "quantum physics", k=1, weights=(0.0, 0.0, 1.0))
    assert s3[0] == "dogs are animals" and str(c3[0]) == "36.0"


def test_get_vector_result_survives_delete(oracle_backend, tmp_path):
    """get -> delete -> re-store (rename) must store the vector that was read (the growable host
    matrix is edited in place, so a live view would silently turn into the following row).  The reference's
    np.delete allocates a new array, an earlier get_vector result keeps its data (vector_database.py:104,126)."""
    import numpy as np
    from minivectordb_amd import VectorDatabase
    db = VectorDatabase(storage_file=str(tmp_path / "db.pkl"))
    x = flat.synth(4, 32, 5)
    for uid, row in zip("abcd", x):
        db.store_embedding(uid, row)
    db.find_most_similar(x[0], k=2)          # build: rows are normalised now
    v = db.get_vector("b")
    want = v.copy()
    db.delete_embedding("b")
    assert np.array_equal(v, want)
    db.store_embedding("b2", v)
    db.find_most_similar(x[0], k=2)
    assert np.allclose(db.get_vector("b2"), want, atol=1e-6)
    assert not np.allclose(db.get_vector("b2"), db.get_vector("c"), atol=1e-3)


def test_partial_ratio_hand_computed_values():
    """Values worked out by hand from rapidfuzz's published algorithm (windows of the shorter string's length plus
    the overhanging ones, Indel ratio 2 LCS / (len a + len b), thefuzz's int(round(.))); see minivectordb_amd/_fuzz.py.
    The library itself is absent here: parity unpinned."""
    from minivectordb_amd._fuzz import partial_ratio
    assert partial_ratio("this is a test", "this is a test!") == 100      # thefuzz README example
    assert partial_ratio("abcd", "xxabcdxx") == 100
    assert partial_ratio("abcd", "abxcd") == 75       # best windows "abxc" / "bxcd": LCS 3 -> 6 / 8
    assert partial_ratio("abxcd", "abcd") == 75       # argument order does not matter
    assert partial_ratio("abc", "bcd") == 80          # window "bc": LCS 2 -> 4 / 5
    assert partial_ratio("abc", "xyz") == 0
    assert partial_ratio("", "abc") == 0 and partial_ratio("abc", "") == 0 and partial_ratio(None, "a") == 0
    assert partial_ratio("dogs", "i like dogs") == 100
    assert partial_ratio("dogs", "cats and dog") == 86  # window " dog" -> LCS 3 -> 6/8 = 75; overhang "dog" -> 6/7 = 85.7
