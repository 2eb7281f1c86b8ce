#!/usr/bin/env python3
"""Generate tests/golden/golden_scenarios.json by replaying tests/golden/scenarios.py through the
REFERENCE's own classes (minivectordb.vector_database.VectorDatabase and
minivectordb.sharded_vector_database.ShardedVectorDatabase, imported from /root/reference).

Runs ONLY in the build container (where /root/reference is mounted); the GPU box and the test
suite read the committed JSON and never import the reference.

What this pins and what it does not:
  * The reference's arithmetic lives in faiss-cpu, which is absent from this image, so the
    reference modules are imported with `faiss` bound to a stand-in whose normalize_L2 /
    IndexFlatIP.{add,search} call this repo's CPU oracle (oracle/flat_oracle.c), and `thefuzz`
    bound to a trivial stand-in (only needed for the import to succeed).
  * Everything ABOVE that boundary is the reference's real code running for real: filter sets,
    search_k clamping, the full/filtered branch choice, sub-index row order, id and metadata
    mapping, tuple/list return types, autocut, renumbering after deletes, shard assignment and
    shard-file bookkeeping, error types and messages.
  * Scores in the fixture are therefore "oracle arithmetic through reference plumbing": parity of
    the arithmetic with faiss itself stays UNPINNED.
"""
import json
import os
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402

from oracle import flat  # noqa: E402
import replay  # noqa: E402
import scenarios  # noqa: E402

REFERENCE = "/root/reference"


def install_stand_ins():
    faiss = types.ModuleType("faiss")

    def normalize_L2(x):
        assert x.dtype == np.float32 and x.flags["C_CONTIGUOUS"]
        flat.normalize_l2(x)

    class IndexFlatIP:
        def __init__(self, d):
            self.d = d
            self.x = np.zeros((0, d), dtype=np.float32)

        @property
        def ntotal(self):
            return self.x.shape[0]

        def add(self, x):
            self.x = np.ascontiguousarray(np.vstack([self.x, np.asarray(x, dtype=np.float32)]))

        def search(self, q, k):
            return flat.flat_search(self.x, q, k)

    faiss.normalize_L2 = normalize_L2
    faiss.IndexFlatIP = IndexFlatIP
    sys.modules["faiss"] = faiss

    thefuzz = types.ModuleType("thefuzz")
    fuzz = types.ModuleType("thefuzz.fuzz")
    fuzz.partial_ratio = lambda a, b: 0
    thefuzz.fuzz = fuzz
    sys.modules["thefuzz"] = thefuzz
    sys.modules["thefuzz.fuzz"] = fuzz


def main():
    install_stand_ins()
    sys.path.insert(0, REFERENCE)
    from minivectordb.vector_database import VectorDatabase
    from minivectordb.sharded_vector_database import ShardedVectorDatabase

    def make_db(kind, path, **kw):
        if kind == "flat":
            return VectorDatabase(storage_file=path)
        return ShardedVectorDatabase(storage_dir=path, **kw)

    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, build in scenarios.SCENARIOS.items():
            ops = build()
            out[name] = {"ops": ops, "expected": replay.run(make_db, ops, tmp)}
            errs = sum(1 for r in out[name]["expected"] if "error" in r)
            print(f"{name}: {len(ops)} ops, {errs} recorded errors")
    path = os.path.join(HERE, "golden_scenarios.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
