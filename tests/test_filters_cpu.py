"""Differential tests of the incremental host bookkeeping (minivectordb_amd/_dbcore.py, round 4) against NAIVE restatements
of the reference's semantics, under random interleavings of stores, deletes and filtered queries:

  * `_get_filtered_indices` (value index maintained through writes, sorted row arrays, symbolic "all rows" / "all rows but")
    vs a per-row evaluation of the reference's filter rules (minivectordb/vector_database.py:157-386: AND clauses, OR clauses
    intersected with them, exclusions; a key matches only rows whose metadata HAS the key; the first operator of a dict);
  * the id index across more than 4096 deletes (handle compaction: the value index must start over);
  * `ShardedVectorDatabase`'s first-fit cursor vs the reference's scan over every shard (sharded_vector_database.py:98-102),
    including deletes that reopen earlier shards, and the files it leaves behind.
Device back end = the oracle stand-in (tests/oracle_backend.py); no GPU.
"""
import operator
import os
import pickle

import numpy as np
import pytest

from oracle import flat
from oracle_backend import OracleIndex


@pytest.fixture
def oracle_backend(monkeypatch):
    from minivectordb_amd import _native
    monkeypatch.setattr(_native, "FlatIndex", OracleIndex)


_OPS = {"$gt": operator.gt, "$gte": operator.ge, "$lt": operator.lt, "$lte": operator.le, "$ne": operator.ne,
        "$in": lambda field, operand: operand in field}


def _matches(meta, key, value, operators_allowed=True):
    """Does a row with metadata `meta` satisfy `key: value`?  Only rows that HAVE the key are ever looked at."""
    if key not in meta:
        return False
    field = meta[key]
    if operators_allowed and isinstance(value, dict):
        name = next(iter(value))
        return bool(_OPS[name](field, value[name]))
    return field == value


def _naive(metadata, metadata_filters, exclude_filter, or_filters):
    """The reference's pipeline (vector_database.py:354-386) row by row; returns the sorted selected rows."""
    n = len(metadata)
    as_list = lambda f: [f] if isinstance(f, dict) else f  # noqa: E731
    chosen = None if metadata_filters else set(range(n))
    if metadata_filters:
        for clause in as_list(metadata_filters):
            for key, value in clause.items():
                rows = {r for r in range(n) if _matches(metadata[r], key, value)}
                chosen = rows if chosen is None else chosen & rows
                if not chosen:
                    break
    if or_filters:
        clauses = [c for c in as_list(or_filters) if c]
        if clauses:
            either = set()
            for clause in clauses:
                for key, value in clause.items():
                    either |= {r for r in range(n) if _matches(metadata[r], key, value)}
            chosen = either if chosen is None else chosen & either
    if exclude_filter:
        for clause in as_list(exclude_filter):
            for key, value in clause.items():
                chosen = chosen - {r for r in range(n) if _matches(metadata[r], key, value, operators_allowed=False)}
                if not chosen:
                    break
    return sorted(chosen if chosen is not None else set())


def _random_meta(rs):
    meta = {}
    if rs.rand() < 0.9:
        meta["bucket"] = int(rs.randint(0, 6))
    if rs.rand() < 0.7:
        meta["colour"] = ["red", "green", "blue", None][rs.randint(0, 4)]
    if rs.rand() < 0.5:
        meta["price"] = float(rs.randint(0, 50)) if rs.rand() < 0.8 else int(rs.randint(0, 50))   # 3 == 3.0 must match
    if rs.rand() < 0.3:
        meta["tags"] = [["a", "b"], ["b"], ["c", "a"], []][rs.randint(0, 4)]                       # unhashable values
    if rs.rand() < 0.1:
        meta["flag"] = bool(rs.randint(0, 2))                                                      # True == 1
    return meta


def _random_filters(rs):
    def clause():
        kind = rs.randint(0, 8)
        if kind == 0:
            return {"bucket": int(rs.randint(0, 7))}
        if kind == 1:
            return {"colour": ["red", "green", "blue", None, "mauve"][rs.randint(0, 5)]}
        if kind == 2:
            return {"price": {["$gt", "$gte", "$lt", "$lte", "$ne"][rs.randint(0, 5)]: int(rs.randint(0, 50))}}
        if kind == 3:
            return {"tags": [["a", "b"], ["b"], ["zzz"]][rs.randint(0, 3)]}
        if kind == 4:
            return {"tags": {"$in": ["a", "b", "c", "q"][rs.randint(0, 4)]}}
        if kind == 5:
            return {"bucket": int(rs.randint(0, 6)), "colour": ["red", "green"][rs.randint(0, 2)]}
        if kind == 6:
            return {"flag": 1}
        return {"nokey": 1}
    mf = ex = orf = None
    if rs.rand() < 0.6:
        mf = clause() if rs.rand() < 0.5 else [clause() for _ in range(rs.randint(1, 3))]
    if rs.rand() < 0.4:
        c = clause()
        while any(isinstance(v, dict) for v in c.values()):   # exclusions compare a dict by equality: keep them plain here
            c = clause()
        ex = c if rs.rand() < 0.5 else [c]
    if rs.rand() < 0.4:
        orf = [clause() for _ in range(rs.randint(1, 3))]
        if rs.rand() < 0.2:
            orf.append({})
    return mf, ex, orf


def _check_filters(db, rs, how_many):
    for _ in range(how_many):
        mf, ex, orf = _random_filters(rs)
        got = db._get_filtered_indices(mf, ex, orf)
        want = _naive(db.metadata, mf, ex, orf)
        assert got.materialize().tolist() == want, (mf, ex, orf)
        assert len(got) == len(want) and bool(got) == bool(want)


@pytest.mark.parametrize("kind", ["flat", "sharded"])
def test_filters_follow_random_writes(tmp_path, oracle_backend, kind):
    from minivectordb_amd import ShardedVectorDatabase, VectorDatabase
    rs = np.random.RandomState(7 if kind == "flat" else 8)
    d = 16
    if kind == "flat":
        db = VectorDatabase(storage_file=str(tmp_path / "f.pkl"))
        delete = db.delete_embedding
    else:
        db = ShardedVectorDatabase(storage_dir=str(tmp_path / "s"), shard_size=37)
        delete = lambda uid: db.delete_embeddings_batch([uid])  # noqa: E731
    alive, nxt = [], 0
    x = flat.synth(4000, d, 99)
    for step in range(900):
        r = rs.rand()
        if r < 0.55 or len(alive) < 5:
            uid = nxt if nxt % 4 else f"s{nxt}"
            db.store_embedding(uid, x[nxt], _random_meta(rs))
            alive.append(uid)
            nxt += 1
        elif r < 0.7:
            ids = [nxt + j for j in range(rs.randint(1, 6))]
            db.store_embeddings_batch(ids, [x[i] for i in ids], [_random_meta(rs) for _ in ids])
            alive += ids
            nxt += len(ids)
        else:
            uid = alive.pop(rs.randint(len(alive)))
            delete(uid)
        if step % 7 == 0:
            _check_filters(db, rs, 6)
        if step % 50 == 0:   # the public views stay what the reference would show
            assert list(db.inverse_id_map) == alive and list(db.inverse_id_map.values()) == list(range(len(alive)))
            assert len(db.metadata) == len(alive)
            for key, holders in db.inverted_index.items():
                assert holders and holders == {u for u, m in zip(alive, db.metadata) if key in m}
    # searches go through the same selections: every hit satisfies its filter, in score order
    ids, dist, metas = db.find_most_similar(x[3], k=50, metadata_filter={"bucket": 2}, exclude_filter={"colour": "red"})
    assert all(m.get("bucket") == 2 and m.get("colour") != "red" for m in metas)
    assert len(ids) == min(50, len(_naive(db.metadata, {"bucket": 2}, {"colour": "red"}, None)))
    assert list(dist) == sorted(dist, reverse=True)


def test_filters_survive_handle_compaction(tmp_path, oracle_backend):
    """More than 4096 deletes: the id index renumbers its handles (epoch moves) and the value index starts over."""
    from minivectordb_amd import VectorDatabase
    rs = np.random.RandomState(3)
    db = VectorDatabase(storage_file=str(tmp_path / "c.pkl"))
    n, d = 6000, 8
    x = flat.synth(n, d, 5)
    metas = [_random_meta(rs) for _ in range(n)]
    db.store_embeddings_batch(list(range(n)), list(x), metas)
    _check_filters(db, rs, 10)          # builds the value index
    epoch = db._ids.epoch
    doomed = rs.permutation(n)[:4500]
    for j, uid in enumerate(doomed.tolist()):
        db.delete_embedding(uid)
        if j % 600 == 0:
            _check_filters(db, rs, 4)
    assert db._ids.epoch > epoch
    _check_filters(db, rs, 20)
    db.store_embedding("late", x[0], {"bucket": 2, "colour": "red"})
    _check_filters(db, rs, 10)
    assert db._get_filtered_indices({"bucket": 2, "colour": "red"}, None, None).materialize().tolist()[-1] == len(db.metadata) - 1


def test_filter_error_behaviour_is_the_references(tmp_path, oracle_backend):
    from minivectordb_amd import VectorDatabase
    db = VectorDatabase(storage_file=str(tmp_path / "e.pkl"))
    x = flat.synth(4, 8, 1)
    db.store_embeddings_batch([1, 2, 3, 4], list(x), [{"a": 1}, {"a": None}, {"b": 2}, {"a": 3, "b": 2}])
    with pytest.raises(ValueError, match="Invalid operator: \\$bad"):
        db.find_most_similar(x[0], metadata_filter={"a": {"$bad": 1}})
    with pytest.raises(TypeError):   # None > 1: the comparison's own error propagates, as in the reference
        db.find_most_similar(x[0], metadata_filter={"a": {"$gt": 1}})
    with pytest.raises(TypeError, match="NoneType"):   # [{}] selects nothing, the exclusion then trips over None (:380-384)
        db.find_most_similar(x[0], metadata_filter=[{}], exclude_filter={"a": 1})
    assert db.find_most_similar(x[0], metadata_filter=[{}]) == ([], [], [])
    # a clause that empties the selection stops reading ITS OWN remaining keys (:291-292): the bad operator is never seen
    assert db.find_most_similar(x[0], metadata_filter={"a": 99, "b": {"$bad": 1}}) == ([], [], [])
    ids, _, _ = db.find_most_similar(x[0], k=4, exclude_filter={"a": {"$gt": 0}})   # exclusions compare a dict by equality
    assert sorted(ids) == [1, 2, 3, 4]


def _naive_shard_for_next_row(box_item_map, shard_size):
    for shard_id, items in box_item_map.items():
        if len(items) < shard_size:
            return shard_id
    return len(box_item_map)


def test_shard_cursor_equals_the_scan(tmp_path, oracle_backend):
    from minivectordb_amd import ShardedVectorDatabase
    rs = np.random.RandomState(12)
    path = str(tmp_path / "shards")
    db = ShardedVectorDatabase(storage_dir=path, shard_size=9)
    model = {}            # shard id -> ids, maintained by the reference's rule
    x = flat.synth(3000, 8, 4)
    alive, nxt = [], 0

    def place(uid):
        sid = _naive_shard_for_next_row(model, 9)
        model.setdefault(sid, []).append(uid)

    for step in range(400):
        r = rs.rand()
        if r < 0.45 or len(alive) < 10:
            db.store_embedding(nxt, x[nxt], {"i": nxt})
            place(nxt)
            alive.append(nxt)
            nxt += 1
        elif r < 0.7:
            ids = list(range(nxt, nxt + rs.randint(1, 25)))
            db.store_embeddings_batch(ids, [x[i] for i in ids], [{"i": i} for i in ids])
            for uid in ids:
                place(uid)
            alive += ids
            nxt += len(ids)
        else:
            kill = [alive.pop(rs.randint(len(alive))) for _ in range(min(len(alive) - 1, rs.randint(1, 12)))]
            db.delete_embeddings_batch(kill)
            for sid in model:
                model[sid] = [u for u in model[sid] if u not in kill]
        assert {k: list(v) for k, v in db.box_item_map.items()} == model, step
    assert db.unique_ids == alive
    # the files are what a fresh instance stacks back together: shard order, rows in file order
    files = sorted((f for f in os.listdir(path)), key=lambda f: int(f.split("_")[1].split(".")[0]))
    stacked = []
    for f in files:
        with open(os.path.join(path, f), "rb") as fh:
            shard = pickle.load(fh)
        assert shard["unique_ids"] == model[int(f.split("_")[1].split(".")[0])]
        assert shard["embeddings"].shape[0] == len(shard["unique_ids"]) == len(shard["metadata"])
        assert shard["inverted_index"] == ({"i": set(shard["unique_ids"])} if shard["unique_ids"] else {})
        stacked += shard["unique_ids"]
    again = ShardedVectorDatabase(storage_dir=path, shard_size=9)
    assert again.unique_ids == stacked and sorted(stacked) == sorted(alive)
    ids, _, metas = again.find_most_similar(x[alive[0]], k=1)
    assert ids[0] == alive[0] and metas[0] == {"i": alive[0]}


def test_find_most_similar_batch_equals_one_query_at_a_time(tmp_path, monkeypatch):
    """find_most_similar_batch (an extension: several queries under one filter in one call) returns, per query, exactly what
    find_most_similar returns — ids, distances, metadata, the empty-result and autocut conventions — for both classes."""
    import numpy as np
    from minivectordb_amd import ShardedVectorDatabase, VectorDatabase, _native
    from oracle_backend import OracleIndex
    monkeypatch.setattr(_native, "FlatIndex", OracleIndex)
    rs = np.random.RandomState(3)
    n, d = 300, 16
    x = rs.randn(n, d).astype(np.float32)
    meta = [{"bucket": i % 5, "tag": "a" if i % 2 else "b"} for i in range(n)]
    q = rs.randn(9, d).astype(np.float32)
    for db in (VectorDatabase(storage_file=str(tmp_path / "f.pkl")),
               ShardedVectorDatabase(storage_dir=str(tmp_path / "s"), shard_size=64)):
        assert db.find_most_similar_batch(q, k=3) == [([], [], [])] * 9          # empty database
        db.store_embeddings_batch([f"id{i}" for i in range(n)], x, meta)
        for kwargs in ({}, {"metadata_filter": {"bucket": 2}}, {"exclude_filter": {"tag": "a"}},
                       {"or_filters": [{"bucket": 1}, {"bucket": 4}], "k": 7}, {"metadata_filter": {"bucket": 99}},
                       {"k": 12, "autocut": True}):
            many = db.find_most_similar_batch(q, **kwargs)
            assert len(many) == 9
            for i in range(9):
                one = db.find_most_similar(q[i], **kwargs)
                assert list(many[i][0]) == list(one[0]) and list(many[i][2]) == list(one[2]), (kwargs, i)
                np.testing.assert_allclose(np.asarray(many[i][1], dtype=np.float64), np.asarray(one[1], dtype=np.float64), atol=1e-6)
                assert type(many[i][0]) is type(one[0])
        with __import__("pytest").raises(ValueError):
            db.find_most_similar_batch(q[0], k=3)
