import os
import sys

import pytest

# The float64 adjudication of the oracle is thousands of SMALL numpy products; on a 128-core box the default BLAS / OpenMP
# thread pools make each of them ~0.1 s (measured: 14 fuzz cases in 200 s instead of ~900).  Bound them unless the caller
# has chosen — before numpy loads where possible, through threadpoolctl otherwise.
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")
if "numpy" in sys.modules:
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=int(os.environ["OMP_NUM_THREADS"]))
    except Exception:  # pragma: no cover - best effort
        pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must run on a GPU box: fail loudly instead of skipping when the device is missing."""
    assert _has_gpu(), "this test is marked gpu but no HIP device is visible"
    return 0


def assert_product_native():
    """The GPU tests compare the HIP library with the oracle: refuse to run them on a stand-in.  `_native.FlatIndex`
    must be the product class and the in-tree libmvdb.so must be mapped into this process."""
    from minivectordb_amd import _native
    assert _native.FlatIndex.__module__ == "minivectordb_amd._native", \
        f"_native.FlatIndex is {_native.FlatIndex!r}: a test left a stand-in behind"
    _native.lib()
    with open("/proc/self/maps") as f:
        maps = f.read()
    assert os.path.realpath(_native.LIB_PATH) in maps, f"{_native.LIB_PATH} is not mapped into this process"
    return _native


@pytest.fixture(autouse=True)
def _gpu_tests_run_on_the_product(request):
    if request.node.get_closest_marker("gpu") is not None:
        assert_product_native()
    yield
