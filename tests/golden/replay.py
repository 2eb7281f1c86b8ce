"""Scenario replay engine shared by tests/golden/make_golden.py (drives the REFERENCE classes, in
the build container only) and tests/test_golden_*.py (drives minivectordb_amd).

A scenario is a JSON-serialisable list of operations on one database object; `run` returns one
JSON-serialisable record per operation.  Vectors are described, not stored: {"synth": [seed, row,
d]} (oracle synthetic stream, bit-exact everywhere), {"list": [...]} or {"synth": ..., "scale": s}.
"""
import os
import shutil

import numpy as np

from oracle import flat


def vec(spec):
    if "list" in spec:
        return np.array(spec["list"], dtype=np.float32)
    seed, row, d = spec["synth"]
    v = flat.synth(1, d, seed, row)[0]
    if "scale" in spec:
        v = (v * np.float32(spec["scale"])).astype(np.float32)
    if "add" in spec:  # nudge towards another synthetic row: controlled similarity
        seed2, row2, w = spec["add"]
        v = (v + np.float32(w) * flat.synth(1, d, seed2, row2)[0]).astype(np.float32)
    return v


def vecs(spec):
    if "rows" in spec:
        return [vec(s) for s in spec["rows"]]
    seed, first, n, d = spec["synth_block"]
    return list(flat.synth(n, d, seed, first))


def _plain(x):
    """numpy scalars / tuples -> JSON-able python values."""
    if isinstance(x, (np.floating,)):
        return float(x)
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    if isinstance(x, dict):
        return {str(k): _plain(v) for k, v in x.items()}
    if isinstance(x, set):
        return sorted(_plain(v) for v in x)
    if isinstance(x, np.ndarray):
        return [_plain(v) for v in x.tolist()]
    return x


def _search_record(res):
    ids, dist, meta = res
    return {
        "ids": _plain(ids), "dist": [float(v) for v in dist], "meta": _plain(meta),
        "types": [type(ids).__name__, type(dist).__name__, type(meta).__name__,
                  type(dist[0]).__name__ if len(dist) else None],
    }


def run(make_db, ops, workdir):
    """make_db(kind, path, **kw) -> database object.  kind in {"flat", "sharded"}."""
    db = None
    out = []
    last_args = None
    for op in ops:
        name = op["op"]
        rec = {"op": name}
        try:
            if name == "open":
                path = os.path.join(workdir, op["path"])
                last_args = (op["kind"], path, op.get("kw", {}))
                db = make_db(*last_args[:2], **last_args[2])
            elif name == "reopen":
                db = make_db(*last_args[:2], **last_args[2])
            elif name == "migrate":  # flat -> sharded through _convert_from_non_sharded_db
                path = os.path.join(workdir, op["path"])
                last_args = ("sharded", path, op.get("kw", {}))
                new_db = make_db(*last_args[:2], **last_args[2])
                new_db._convert_from_non_sharded_db(db)
                db = new_db
            elif name == "wipe":
                path = os.path.join(workdir, op["path"])
                if os.path.isdir(path):
                    shutil.rmtree(path)
                elif os.path.exists(path):
                    os.remove(path)
            elif name == "store":
                if "meta" in op:
                    db.store_embedding(op["id"], vec(op["vec"]), op["meta"])
                else:
                    db.store_embedding(op["id"], vec(op["vec"]))
            elif name == "store_batch":
                args = [op["ids"], vecs(op["vecs"])]
                if "metas" in op:
                    args.append([dict(m) for m in op["metas"]])
                db.store_embeddings_batch(*args)
            elif name == "delete":
                db.delete_embedding(op["id"])
            elif name == "delete_batch":
                db.delete_embeddings_batch(op["ids"])
            elif name == "persist":
                db.persist_to_disk()
            elif name == "search":
                kw = {}
                for src, dst in (("filter", "metadata_filter"), ("exclude", "exclude_filter"), ("or", "or_filters"),
                                 ("k", "k"), ("autocut", "autocut")):
                    if src in op:
                        kw[dst] = op[src]
                rec.update(_search_record(db.find_most_similar(vec(op["q"]), **kw)))
            elif name == "get_vector":
                rec["vector"] = [float(v) for v in db.get_vector(op["id"])]
            elif name == "autocut_scores":
                rec["cut"] = _plain(db.autocut_scores([np.float32(v) for v in op["scores"]]))
            elif name == "state":
                rec["n"] = 0 if db.embeddings is None else int(db.embeddings.shape[0])
                rec["embedding_size"] = _plain(db.embedding_size)
                rec["inverse_id_map"] = [[_plain(k), int(v)] for k, v in db.inverse_id_map.items()]
                if hasattr(db, "id_map"):
                    rec["id_map"] = [[int(k), _plain(v)] for k, v in sorted(db.id_map.items())]
                if hasattr(db, "unique_ids"):
                    rec["unique_ids"] = _plain(db.unique_ids)
                    rec["box_item_map"] = [[int(k), _plain(v)] for k, v in sorted(db.box_item_map.items())]
                    rec["shard_files"] = sorted(os.listdir(db.storage_dir))
                rec["inverted_index"] = {str(k): sorted(_plain(v) for v in s) for k, s in db.inverted_index.items()}
                rec["metadata"] = _plain(db.metadata)
            else:
                raise RuntimeError(f"unknown op {name}")
        except (ValueError, IndexError, KeyError, TypeError) as e:
            rec["error"] = type(e).__name__
            rec["message"] = str(e)
        out.append(rec)
    return out
