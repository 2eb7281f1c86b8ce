"""Host logic of the drop-in classes against the golden records made from the REFERENCE classes
(tests/golden/make_golden.py).  The device back end is replaced by the oracle-backed stand-in
(tests/oracle_backend.py), so every record must match EXACTLY — ids, float32 scores bit for bit,
metadata, return types, error types and messages, id maps, shard bookkeeping."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))

import golden_compare  # noqa: E402
import replay  # noqa: E402
from oracle_backend import OracleIndex  # noqa: E402

GOLDEN = golden_compare.load()


@pytest.fixture
def oracle_backend(monkeypatch):
    from minivectordb_amd import _native
    monkeypatch.setattr(_native, "FlatIndex", OracleIndex)


def make_db(kind, path, **kw):
    from minivectordb_amd import ShardedVectorDatabase, VectorDatabase
    if kind == "flat":
        return VectorDatabase(storage_file=path)
    return ShardedVectorDatabase(storage_dir=path, **kw)


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_scenario_matches_reference(name, tmp_path, oracle_backend):
    sc = GOLDEN[name]
    got = replay.run(make_db, sc["ops"], str(tmp_path))
    golden_compare.compare(got, sc["expected"], tol=3e-7, exact=True)
