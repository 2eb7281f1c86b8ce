"""Scenario definitions (inputs only) for the golden fixtures.  The expected outputs are recorded by
make_golden.py from the REFERENCE's own classes; these workloads are this repo's own, modelled on
the situations the reference's tests exercise (G1 config-1 search, G2 filters,
G3 autocut, G4 colinear ties, G5 renumbering after deletes, sharded bookkeeping)."""

D = 512


def q(row, d=D, seed=5678):
    return {"synth": [seed, row, d]}


def _items(n, d=D, seed=1234, first=0):
    colours = ["red", "green", "blue", "amber"]
    metas = []
    for i in range(first, first + n):
        m = {"bucket": i % 10, "price": round(i * 0.75, 2), "colour": colours[i % 4], "name": f"item-{i}"}
        if i % 3 == 0:
            m["tags"] = ["sale", f"t{i % 7}"]
        if i % 5 == 0:
            m["stock"] = i % 11
        metas.append(m)
    return {"op": "store_batch", "ids": list(range(first, first + n)),
            "vecs": {"synth_block": [seed, first, n, d]}, "metas": metas}


def config1():
    """BASELINE config 1: 1k x 512, k = 5 through find_most_similar (+ filter variety on the same db)."""
    ops = [{"op": "wipe", "path": "c1.pkl"}, {"op": "open", "kind": "flat", "path": "c1.pkl"}, _items(1000)]
    for r in range(8):
        ops.append({"op": "search", "q": q(r), "k": 5})
    ops += [
        {"op": "search", "q": q(20), "k": 10},
        {"op": "search", "q": q(21), "k": 64},
        {"op": "search", "q": q(22), "k": 100},
        {"op": "search", "q": q(23), "k": 5, "filter": {"bucket": 3}},
        {"op": "search", "q": q(24), "k": 5, "filter": {"bucket": 3, "colour": "blue"}},
        {"op": "search", "q": q(25), "k": 7, "filter": [{"bucket": 4}, {"colour": "red"}]},
        {"op": "search", "q": q(26), "k": 5, "filter": {"price": {"$gte": 600}}},
        {"op": "search", "q": q(27), "k": 5, "filter": {"price": {"$lt": 30.5}}},
        {"op": "search", "q": q(28), "k": 5, "filter": {"price": {"$gt": 100, "$lt": 110}}},  # 2nd operator ignored
        {"op": "search", "q": q(29), "k": 5, "filter": {"colour": {"$ne": "red"}, "bucket": {"$lte": 1}}},
        {"op": "search", "q": q(30), "k": 5, "filter": {"tags": {"$in": "sale"}}},
        {"op": "search", "q": q(31), "k": 5, "filter": {"tags": {"$in": "t3"}, "stock": {"$gte": 0}}},
        {"op": "search", "q": q(32), "k": 5, "or": [{"bucket": 1}, {"bucket": 2, "colour": "amber"}]},
        {"op": "search", "q": q(33), "k": 5, "or": {"name": "item-17"}},
        {"op": "search", "q": q(34), "k": 5, "filter": {"colour": "green"}, "or": [{"bucket": 5}, {"bucket": 7}]},
        {"op": "search", "q": q(35), "k": 5, "exclude": {"colour": "red"}},
        {"op": "search", "q": q(36), "k": 5, "exclude": [{"colour": "red"}, {"bucket": 2}]},
        {"op": "search", "q": q(37), "k": 5, "filter": {"bucket": 6}, "exclude": {"colour": "blue"}},
        {"op": "search", "q": q(38), "k": 5, "filter": {"bucket": 6}, "or": [{"colour": "blue"}, {}],
         "exclude": {"name": "item-6"}},
        {"op": "search", "q": q(39), "k": 5, "filter": {"nokey": 1}},
        {"op": "search", "q": q(40), "k": 5, "filter": {}},
        {"op": "search", "q": q(41), "k": 5, "or": [{}]},
        {"op": "search", "q": q(42), "k": 5, "exclude": {"nokey": 1}},
        {"op": "search", "q": q(43), "k": 5, "filter": {"price": {"$between": [1, 2]}}},  # invalid operator
        {"op": "search", "q": q(44), "k": 5, "or": [{"price": {"$regex": "x"}}]},
        {"op": "search", "q": q(45), "k": 2000},  # k > n
        {"op": "search", "q": q(46), "k": 999, "filter": {"bucket": {"$lt": 3}}},  # large k on a subset (300 rows)
        {"op": "search", "q": q(47), "k": 5, "filter": {"stock": {"$gt": 4}}, "autocut": True},
        {"op": "search", "q": {"synth": [1234, 77, D], "scale": 3.5}, "k": 5},  # a stored row, un-normalised
        {"op": "search", "q": {"synth": [1234, 77, D], "scale": 3.5}, "k": 5, "autocut": True},
        {"op": "get_vector", "id": 77},
        {"op": "get_vector", "id": 5000},
        {"op": "state"},
    ]
    return ops


def colinear():
    """2-d colinear rows: exact/near score ties, k > n, filters with ties (reference tests' vectors)."""
    rows = [[0.5, 0.5], [0.1, 0.1], [0.7, 0.7], [0.5, -0.4], [0.2, 0.9], [0.9, 0.2], [-0.3, -0.3], [0.0, 1.0]]
    ops = [{"op": "wipe", "path": "co.pkl"}, {"op": "open", "kind": "flat", "path": "co.pkl"},
           {"op": "search", "q": {"list": [1.0, 1.0]}, "k": 3}]  # empty database
    for i, r in enumerate(rows):
        ops.append({"op": "store", "id": i + 1, "vec": {"list": r},
                    "meta": {"group": "a" if i % 2 == 0 else "b", "rank": i}})
    ops += [
        {"op": "search", "q": {"list": [1.0, 1.0]}, "k": 3},
        {"op": "search", "q": {"list": [1.0, 1.0]}, "k": 20},
        {"op": "search", "q": {"list": [0.2, 0.9]}, "k": 4, "filter": {"group": "a"}},
        {"op": "search", "q": {"list": [1.0, 0.0]}, "k": 4, "or": [{"group": "b"}]},
        {"op": "search", "q": {"list": [1.0, 1.0]}, "k": 8, "autocut": True},
        {"op": "search", "q": {"list": [0.0, 0.0]}, "k": 3},  # zero query
        {"op": "store", "id": 1, "vec": {"list": [1.0, 2.0]}},  # duplicate id
        {"op": "store", "id": 99, "vec": {"list": [0.0, 0.0]}},  # zero row: normalisation leaves it
        {"op": "search", "q": {"list": [1.0, 1.0]}, "k": 20},
        {"op": "get_vector", "id": 99},
        {"op": "get_vector", "id": 1},
        {"op": "state"},
    ]
    return ops


def deletes():
    """Renumbering after deletes, interleaved with searches and new stores (incremental device sync)."""
    d = 64
    ops = [{"op": "wipe", "path": "del.pkl"}, {"op": "open", "kind": "flat", "path": "del.pkl"}]
    for i in range(1, 7):
        ops.append({"op": "store", "id": i, "vec": {"synth": [7, i, d]}, "meta": {"v": i, "odd": i % 2}})
    ops += [
        {"op": "search", "q": q(1, d), "k": 3},
        {"op": "delete", "id": 2},
        {"op": "state"},
        {"op": "search", "q": q(1, d), "k": 6},
        {"op": "delete", "id": 2},  # already gone
        {"op": "store", "id": "seven", "vec": {"synth": [7, 7, d]}, "meta": {"v": 7, "odd": 1}},
        {"op": "delete", "id": 1},
        {"op": "delete", "id": "seven"},
        {"op": "store", "id": 8, "vec": {"synth": [7, 8, d]}, "meta": {"v": 8, "odd": 0}},
        {"op": "search", "q": q(2, d), "k": 10, "filter": {"odd": 1}},
        {"op": "search", "q": q(2, d), "k": 10},
        {"op": "state"},
        {"op": "persist"},
        {"op": "reopen"},
        {"op": "state"},
        {"op": "search", "q": q(2, d), "k": 10},
        {"op": "get_vector", "id": 8},
    ]
    for i in (3, 4, 5, 6, 8):
        ops.append({"op": "delete", "id": i})
    ops += [{"op": "state"}, {"op": "search", "q": q(2, d), "k": 3}]
    return ops


def autocut():
    """Result lists with and without a > 20 % relative drop."""
    d = 128
    ops = [{"op": "wipe", "path": "ac.pkl"}, {"op": "open", "kind": "flat", "path": "ac.pkl"}]
    # rows 0-2 close to the query direction (seed 5678 row 0), the rest unrelated
    for i, w in enumerate([6.0, 5.0, 4.5]):
        ops.append({"op": "store", "id": f"near{i}", "vec": {"synth": [11, i, d], "add": [5678, 0, w]},
                    "meta": {"kind": "near"}})
    for i in range(10):
        ops.append({"op": "store", "id": f"far{i}", "vec": {"synth": [12, i, d]}, "meta": {"kind": "far"}})
    ops += [
        {"op": "search", "q": q(0, d), "k": 8},
        {"op": "search", "q": q(0, d), "k": 8, "autocut": True},
        {"op": "search", "q": q(0, d), "k": 3, "autocut": True},
        {"op": "search", "q": q(0, d), "k": 1, "autocut": True},
        {"op": "search", "q": q(0, d), "k": 8, "autocut": True, "filter": {"kind": "far"}},
        {"op": "autocut_scores", "scores": [0.9, 0.85, 0.5, 0.45]},
        {"op": "autocut_scores", "scores": [0.9, 0.8, 0.7, 0.6]},
        {"op": "autocut_scores", "scores": [0.5, 0.5, 0.1]},
        {"op": "autocut_scores", "scores": [0.3, -0.1, -0.2]},
        {"op": "autocut_scores", "scores": [0.7, 0.7]},
    ]
    return ops


def sharded():
    """Shard-file bookkeeping (shard_size = 7), single + batch stores, batch delete, reload, search."""
    d = 96
    ops = [{"op": "wipe", "path": "shards"},
           {"op": "open", "kind": "sharded", "path": "shards", "kw": {"shard_size": 7}}]
    for i in range(5):
        ops.append({"op": "store", "id": f"s{i}", "vec": {"synth": [21, i, d]}, "meta": {"part": i % 2, "n": i}})
    b = _items(30, d=d, seed=22, first=100)
    ops += [
        b,
        {"op": "state"},
        {"op": "search", "q": q(3, d), "k": 5},
        {"op": "search", "q": q(3, d), "k": 5, "filter": {"bucket": 1}},
        {"op": "search", "q": q(4, d), "k": 6, "or": [{"part": 1}, {"colour": "red"}], "exclude": {"bucket": 4}},
        {"op": "search", "q": q(5, d), "k": 50},
        {"op": "get_vector", "id": "s0"},
        {"op": "delete_batch", "ids": ["s1", 101, 110, 129]},
        {"op": "state"},
        {"op": "search", "q": q(3, d), "k": 5},
        {"op": "delete_batch", "ids": ["nope"]},
        {"op": "delete_batch", "ids": []},
        {"op": "store_batch", "ids": [500, 501], "vecs": {"synth_block": [23, 0, 2, d]}},
        {"op": "store_batch", "ids": [600], "vecs": {"synth_block": [23, 5, 2, d]}},  # length mismatch
        {"op": "store", "id": 500, "vec": {"synth": [23, 9, d]}},  # duplicate
        {"op": "state"},
        {"op": "reopen"},
        {"op": "state"},
        {"op": "search", "q": q(3, d), "k": 5},
        {"op": "search", "q": q(6, d), "k": 4, "filter": {"price": {"$gt": 80}}, "autocut": True},
        {"op": "delete_batch", "ids": "s0"},  # bare id is wrapped into a list
        {"op": "search", "q": q(3, d), "k": 3},
    ]
    return ops


def mixed_dims():
    """Odd dimensions (padding path) and batch-argument errors."""
    ops = [{"op": "wipe", "path": "md.pkl"}, {"op": "open", "kind": "flat", "path": "md.pkl"},
           {"op": "store_batch", "ids": list(range(40)), "vecs": {"synth_block": [31, 0, 40, 3]},
            "metas": [{"i": i} for i in range(40)]},
           {"op": "search", "q": q(0, 3), "k": 5},
           {"op": "search", "q": q(1, 3), "k": 5, "filter": {"i": {"$lt": 10}}},
           {"op": "store_batch", "ids": [100, 101], "vecs": {"synth_block": [31, 100, 2, 3]}, "metas": [{"i": 100}]},
           {"op": "store_batch", "ids": [0], "vecs": {"synth_block": [31, 200, 1, 3]}},
           {"op": "store_batch", "ids": [200, 201], "vecs": {"synth_block": [31, 300, 2, 3]}},
           {"op": "search", "q": q(2, 3), "k": 50},
           {"op": "state"}]
    return ops


def _fuzz(kind, seed, nops=140):
    """Seeded random mix of stores, batch stores, deletes and searches with random filters (flat or
    sharded): exercises interleavings the hand-written scenarios do not."""
    import random
    rnd = random.Random(seed)
    d = 48
    path = f"fuzz_{kind}_{seed}" + ("" if kind == "sharded" else ".pkl")
    kw = {"shard_size": 9} if kind == "sharded" else {}
    ops = [{"op": "wipe", "path": path}, {"op": "open", "kind": kind, "path": path, "kw": kw}]
    live, next_id = [], 0
    colours = ["red", "green", "blue"]

    def meta(i):
        m = {"g": i % 4, "score": (i * 37) % 101, "colour": colours[i % 3]}
        if i % 2:
            m["tags"] = [f"t{i % 5}", "x"]
        return m

    def rand_filter():
        c = rnd.randrange(7)
        if c == 0:
            return {"filter": {"g": rnd.randrange(4)}}
        if c == 1:
            return {"filter": {"score": {rnd.choice(["$gt", "$gte", "$lt", "$lte", "$ne"]): rnd.randrange(101)}}}
        if c == 2:
            return {"or": [{"colour": rnd.choice(colours)}, {"g": rnd.randrange(4)}]}
        if c == 3:
            return {"exclude": {"colour": rnd.choice(colours)}}
        if c == 4:
            return {"filter": {"tags": {"$in": f"t{rnd.randrange(5)}"}}, "exclude": [{"g": rnd.randrange(4)}]}
        if c == 5:
            return {"filter": [{"g": rnd.randrange(4)}, {"colour": rnd.choice(colours)}],
                    "or": {"score": {"$gte": rnd.randrange(101)}}}
        return {}

    for step in range(nops):
        r = rnd.random()
        if r < 0.30 or not live:
            ops.append({"op": "store", "id": next_id, "vec": {"synth": [900 + seed, next_id, d]}, "meta": meta(next_id)})
            live.append(next_id)
            next_id += 1
        elif r < 0.40:
            n = rnd.randrange(1, 12)
            ids = list(range(next_id, next_id + n))
            ops.append({"op": "store_batch", "ids": ids, "vecs": {"synth_block": [900 + seed, next_id, n, d]},
                        "metas": [meta(i) for i in ids]})
            live += ids
            next_id += n
        elif r < 0.52:
            victim = live.pop(rnd.randrange(len(live)))
            if kind == "sharded":
                extra = [live.pop(rnd.randrange(len(live)))] if live and rnd.random() < 0.4 else []
                ops.append({"op": "delete_batch", "ids": [victim] + extra})
            else:
                ops.append({"op": "delete", "id": victim})
        elif r < 0.56:
            ops.append({"op": "state"})
        elif r < 0.60 and kind == "flat":
            ops += [{"op": "persist"}, {"op": "reopen"}]
        elif r < 0.60:
            ops.append({"op": "reopen"})
        else:
            op = {"op": "search", "q": q(rnd.randrange(10_000), d), "k": rnd.choice([1, 3, 5, 10, 70])}
            op.update(rand_filter())
            if rnd.random() < 0.2:
                op["autocut"] = True
            ops.append(op)
    ops.append({"op": "state"})
    return ops


def delete_everything(kind):
    """Delete every row, reload from disk, search the empty database, store again (the situations of the
    reference's test_index_then_delete_everything_and_reload and of deleting unknown + known ids together)."""
    d = 32
    path = "wipeout" if kind == "sharded" else "wipeout.pkl"
    kw = {"shard_size": 4} if kind == "sharded" else {}
    ops = [{"op": "wipe", "path": path}, {"op": "open", "kind": kind, "path": path, "kw": kw}]
    for i in range(9):
        ops.append({"op": "store", "id": f"k{i}", "vec": {"synth": [41, i, d]}, "meta": {"even": i % 2 == 0}})
    ops.append({"op": "search", "q": q(0, d), "k": 4})
    if kind == "sharded":
        ops += [{"op": "delete_batch", "ids": ["k0", "missing"]},  # a known and an unknown id together
                {"op": "delete_batch", "ids": [f"k{i}" for i in range(9)]}]
    else:
        ops += [{"op": "delete", "id": f"k{i}"} for i in range(9)]
    ops += [{"op": "state"}, {"op": "search", "q": q(0, d), "k": 4},
            {"op": "search", "q": q(0, d), "k": 4, "filter": {"even": True}}]
    if kind == "flat":
        ops.append({"op": "persist"})
    ops += [{"op": "reopen"}, {"op": "state"}, {"op": "search", "q": q(1, d), "k": 4},
            {"op": "store", "id": "again", "vec": {"synth": [41, 100, d]}, "meta": {"even": True}},
            {"op": "search", "q": q(1, d), "k": 4}, {"op": "search", "q": q(1, d), "k": 4, "filter": {"even": True}},
            {"op": "state"}]
    return ops


def migrate():
    """VectorDatabase -> ShardedVectorDatabase via _convert_from_non_sharded_db, then search / reload."""
    d = 40
    ops = [{"op": "wipe", "path": "mig.pkl"}, {"op": "wipe", "path": "mig_shards"},
           {"op": "open", "kind": "flat", "path": "mig.pkl"}, _items(23, d=d, seed=51)]
    ops += [{"op": "search", "q": q(2, d), "k": 5},
            {"op": "migrate", "path": "mig_shards", "kw": {"shard_size": 5}},
            {"op": "state"}, {"op": "search", "q": q(2, d), "k": 5},
            {"op": "search", "q": q(3, d), "k": 5, "filter": {"colour": "blue"}},
            {"op": "reopen"}, {"op": "state"}, {"op": "search", "q": q(2, d), "k": 5}]
    return ops


SCENARIOS = {
    "migrate": migrate,
    "delete_everything_flat": lambda: delete_everything("flat"),
    "delete_everything_sharded": lambda: delete_everything("sharded"),
    "fuzz_flat_1": lambda: _fuzz("flat", 1),
    "fuzz_flat_2": lambda: _fuzz("flat", 2),
    "fuzz_sharded_3": lambda: _fuzz("sharded", 3),
    "config1": config1,
    "colinear": colinear,
    "deletes": deletes,
    "autocut": autocut,
    "sharded": sharded,
    "mixed_dims": mixed_dims,
}
