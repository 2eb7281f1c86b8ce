"""The same golden scenarios through the real HIP back end (C-ABI): ids identical (away from fp32
ties), scores within 1e-4 of the reference-plumbing + oracle-arithmetic records."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))

import golden_compare  # noqa: E402
import replay  # noqa: E402

pytestmark = pytest.mark.gpu
GOLDEN = golden_compare.load()


def make_db(kind, path, **kw):
    from minivectordb_amd import ShardedVectorDatabase, VectorDatabase
    if kind == "flat":
        return VectorDatabase(storage_file=path)
    return ShardedVectorDatabase(storage_dir=path, **kw)


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_scenario_matches_reference_on_gpu(name, tmp_path, gpu):
    sc = GOLDEN[name]
    got = replay.run(make_db, sc["ops"], str(tmp_path))
    golden_compare.compare(got, sc["expected"], tol=1e-4, exact=False)
