"""The C-ABI library: loads, exports every symbol include/mvdb.h declares, and fails loudly
(never falls back) when no HIP device is present.  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "mvdb.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mvdb_[a-z0-9_]+)\s*\(", text)))


def test_header_functions_are_exported_and_bound():
    from minivectordb_amd import _native
    lib = _native.lib()
    names = declared_functions()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/mvdb.h but not exported by libmvdb.so"
        assert name in _native.PROTOTYPES, f"{name} has no ctypes prototype"
    for name in _native.PROTOTYPES:
        assert name in names, f"{name} bound in _native.py but not declared in include/mvdb.h"


def test_abi_version_and_error_channel():
    from minivectordb_amd import _native
    lib = _native.lib()
    assert lib.mvdb_abi_version() == 1
    h = ctypes.c_void_p()
    rc = lib.mvdb_index_create(0, 0, 0, ctypes.byref(h))
    assert rc == _native.ERR_ARG and "dimension" in _native.last_error()
    rc = lib.mvdb_index_create(8, 7, 0, ctypes.byref(h))
    assert rc == _native.ERR_ARG and "metric" in _native.last_error()


def test_library_is_in_tree_and_is_hip_code():
    from minivectordb_amd import _native
    assert os.path.dirname(_native.LIB_PATH).startswith(ROOT)
    blob = open(_native.LIB_PATH, "rb").read()
    assert b"gfx950" in blob, "libmvdb.so carries no gfx950 code object"


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the no-device behaviour is checked on CPU-only hosts")
    from minivectordb_amd import _native
    with pytest.raises(RuntimeError, match="no HIP device"):
        _native.device_count()
    with pytest.raises(RuntimeError, match="no HIP device|no CPU fallback"):
        _native.FlatIndex(16)
    import numpy as np
    from minivectordb_amd import VectorDatabase
    db = VectorDatabase(storage_file=os.path.join("/tmp", f"mvdb_nofallback_{os.getpid()}.pkl"))
    db.store_embedding(1, np.ones(8, np.float32))
    with pytest.raises(RuntimeError):
        db.find_most_similar(np.ones(8, np.float32))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "minivectordb_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "liboracle" not in text, f
