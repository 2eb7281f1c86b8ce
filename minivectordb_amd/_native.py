"""ctypes binding of libmvdb.so (include/mvdb.h).

The library is the ONLY compute back end of this package: there is no CPU fallback.  Loading
fails loudly when the in-tree shared object is missing, and every compute entry point raises
``RuntimeError`` when no HIP device is usable.

PyTorch-ROCm bundles its own HIP runtime under the SONAME ``libamdhip64.so.7``; libmvdb.so must bind to
THAT instance, so that device pointers and streams can be exchanged with torch tensors (torch is used
only for device memory, streams and torch.distributed — never for the search arithmetic) and so that a
later ``import torch`` does not bring a second HIP runtime into the process.  If torch is already
imported its runtime is loaded; otherwise torch's ``lib/libamdhip64.so`` is pre-loaded by path WITHOUT
importing torch (a VectorDatabase-only user pays no 1-2 s ``import torch`` on the first query).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmvdb.so")
LIB_PATH = os.environ.get("MVDB_LIBMVDB", LIB_PATH)  # diagnostics only: e.g. the ablation build (make ABLATE=1)

METRIC_IP = 0
METRIC_L2 = 1

ERR_ARG = 1
ERR_HIP = 2
ERR_NODEVICE = 3
ERR_OOM = 4

_lib = None

c_f32p = ctypes.POINTER(ctypes.c_float)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_vp = ctypes.c_void_p


class EncoderCfg(ctypes.Structure):
    """mirror of mvdb_encoder_cfg"""
    _fields_ = [
        ("vocab_size", ctypes.c_int),
        ("hidden", ctypes.c_int),
        ("layers", ctypes.c_int),
        ("heads", ctypes.c_int),
        ("intermediate", ctypes.c_int),
        ("max_positions", ctypes.c_int),
        ("type_vocab", ctypes.c_int),
        ("position_offset", ctypes.c_int),
        ("ln_eps", ctypes.c_float),
        ("pooling", ctypes.c_int),
    ]


# name -> (restype, argtypes); the single source of truth for tests that check the export table
PROTOTYPES = {
    "mvdb_last_error": (ctypes.c_char_p, []),
    "mvdb_abi_version": (ctypes.c_int, []),
    "mvdb_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "mvdb_index_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(c_vp)]),
    "mvdb_index_free": (ctypes.c_int, [c_vp]),
    "mvdb_index_reload_env": (ctypes.c_int, [c_vp]),
    "mvdb_index_set_option": (ctypes.c_int, [c_vp, ctypes.c_char_p, ctypes.c_longlong]),
    "mvdb_index_reset": (ctypes.c_int, [c_vp]),
    "mvdb_index_ntotal": (ctypes.c_int64, [c_vp]),
    "mvdb_index_shadow_rows": (ctypes.c_int64, [c_vp]),
    "mvdb_index_dim": (ctypes.c_int, [c_vp]),
    "mvdb_index_device": (ctypes.c_int, [c_vp]),
    "mvdb_index_reserve": (ctypes.c_int, [c_vp, ctypes.c_int64]),
    "mvdb_index_add": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, ctypes.c_int]),
    "mvdb_index_add_device": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, ctypes.c_int]),
    "mvdb_index_add_synthetic": (ctypes.c_int, [c_vp, ctypes.c_int64, ctypes.c_uint64, ctypes.c_int64, ctypes.c_int]),
    "mvdb_index_get_rows": (ctypes.c_int, [c_vp, ctypes.c_int64, ctypes.c_int64, c_vp]),
    "mvdb_index_remove_rows": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64]),
    "mvdb_index_search": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp, c_vp]),
    "mvdb_index_search_device": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int64, c_vp, c_vp, c_vp]),
    "mvdb_index_search_subset": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp,
                                                ctypes.c_int64, c_vp, c_vp]),
    "mvdb_index_search_subset_device": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp,
                                                       ctypes.c_int64, ctypes.c_int, ctypes.c_int64, c_vp, c_vp, c_vp]),
    "mvdb_index_search_masked": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp, ctypes.c_int,
                                                c_vp, c_vp]),
    "mvdb_index_search_masked_device": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp,
                                                       ctypes.c_int, ctypes.c_int64, c_vp, c_vp, c_vp]),
    "mvdb_rowset_create": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(c_vp)]),
    "mvdb_rowset_size": (ctypes.c_int64, [c_vp]),
    "mvdb_rowset_is_bitmap": (ctypes.c_int, [c_vp]),
    "mvdb_rowset_free": (ctypes.c_int, [c_vp]),
    "mvdb_index_search_rowset": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp, c_vp, c_vp]),
    "mvdb_index_search_rowset_device": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp,
                                                       ctypes.c_int64, c_vp, c_vp, c_vp]),
    "mvdb_comm_available": (ctypes.c_int, []),
    "mvdb_comm_unique_id": (ctypes.c_int, [c_vp]),
    "mvdb_comm_create": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(c_vp)]),
    "mvdb_comm_free": (ctypes.c_int, [c_vp]),
    "mvdb_comm_rank": (ctypes.c_int, [c_vp]),
    "mvdb_comm_world": (ctypes.c_int, [c_vp]),
    "mvdb_allgather_topk": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int64, c_vp]),
    "mvdb_merge_topk_device": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp,
                                              ctypes.c_int64, c_vp, ctypes.c_int64, c_vp, c_vp, ctypes.c_int, c_vp]),
    "mvdb_normalize_l2": (ctypes.c_int, [c_vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    "mvdb_synth_fill_device": (ctypes.c_int, [c_vp, ctypes.c_int64, ctypes.c_int, ctypes.c_uint64, ctypes.c_int64,
                                              ctypes.c_int, ctypes.c_int, c_vp]),
    "mvdb_split_rerun_count": (ctypes.c_int64, []),
    "mvdb_rescue_tile_stats": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "mvdb_half_eps": (ctypes.c_double, [ctypes.c_int]),
    "mvdb_half_max_queries": (ctypes.c_int, [ctypes.c_int]),
    "mvdb_prof_enable": (ctypes.c_int, [ctypes.c_int]),
    "mvdb_prof_read": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64),
                                      ctypes.POINTER(ctypes.c_double)]),
    "mvdb_prof_symbol": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]),
    # encoder (encoder.hip)
    "mvdb_encoder_weight_count": (ctypes.c_int, [ctypes.POINTER(EncoderCfg)]),
    "mvdb_encoder_weight_name": (ctypes.c_char_p, [ctypes.POINTER(EncoderCfg), ctypes.c_int]),
    "mvdb_encoder_create": (ctypes.c_int, [ctypes.POINTER(EncoderCfg), ctypes.POINTER(c_vp), ctypes.c_int,
                                           ctypes.POINTER(c_vp)]),
    "mvdb_encoder_free": (ctypes.c_int, [c_vp]),
    "mvdb_encoder_gemm_tile_form": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    "mvdb_encoder_splitk_planes": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "mvdb_index_single_route_suspensions": (ctypes.c_longlong, [c_vp]),
    "mvdb_encoder_walks": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int]),
    "mvdb_encoder_walk_stats": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp]),
    "mvdb_encoder_overflow_flag": (c_vp, [c_vp]),
    "mvdb_encoder_forward": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp]),
    "mvdb_encoder_forward_device": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                   c_vp, c_vp, c_vp]),
}


def _bind_torch_hip_runtime():
    """One HIP runtime per process (module docstring): torch's own, loaded before libmvdb.so resolves its
    ``libamdhip64.so.7`` dependency."""
    import sys
    if "torch" in sys.modules or os.environ.get("MVDB_IMPORT_TORCH_FIRST") == "1":
        import torch  # noqa: F401
        return
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return  # no torch in this environment: the system ROCm runtime serves
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
            return
        except OSError:
            pass
    import torch  # noqa: F401  (layout not recognised: the slow, certain way)


def lib():
    """Load libmvdb.so (once).  Raises if the in-tree build is missing — never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP library first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C minivectordb_amd/csrc). "
            "minivectordb_amd has no CPU fallback.")
    _bind_torch_hip_runtime()
    L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return _lib


def last_error():
    msg = lib().mvdb_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc):
    """Map a non-zero status to the Python exception the drop-in layer promises."""
    if rc == 0:
        return
    msg = last_error()
    if rc == ERR_ARG:
        raise ValueError(msg)
    if rc == ERR_OOM:
        raise MemoryError(msg)
    raise RuntimeError(msg)


def device_count():
    n = ctypes.c_int(0)
    check(lib().mvdb_device_count(ctypes.byref(n)))
    return n.value


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


class FlatIndex:
    """Thin object wrapper over mvdb_index*: the device-resident flat matrix + search."""

    def __init__(self, d, metric=METRIC_IP, device=0):
        self._h = ctypes.c_void_p()
        self.d = int(d)
        self.metric = metric
        self.device = device
        check(lib().mvdb_index_create(self.d, metric, device, ctypes.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().mvdb_index_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    @property
    def ntotal(self):
        return int(lib().mvdb_index_ntotal(self._h))

    @property
    def shadow_rows(self):
        """Rows in the fp16 shadow the batch passes stream (0: none yet / not applicable)."""
        return int(lib().mvdb_index_shadow_rows(self._h))

    def reset(self):
        check(lib().mvdb_index_reset(self._h))

    def reserve(self, n):
        check(lib().mvdb_index_reserve(self._h, int(n)))

    def set_option(self, name, value):
        """Per-index switch set in code (include/mvdb.h: mvdb_index_set_option): "shadow_single_query", "half_shadow",
        "compact_bytes"."""
        check(lib().mvdb_index_set_option(self._h, str(name).encode(), int(value)))

    def reload_env(self):
        """Re-read the MVDB_* hooks (the library reads them once, when the index is created)."""
        check(lib().mvdb_index_reload_env(self._h))

    def add(self, x, normalize=False):
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError(f"expected [n,{self.d}] float32, got {x.shape}")
        check(lib().mvdb_index_add(self._h, _ptr(x), x.shape[0], int(bool(normalize))))

    def add_device(self, ptr, n, normalize=False):
        check(lib().mvdb_index_add_device(self._h, ctypes.c_void_p(ptr), int(n), int(bool(normalize))))

    def add_synthetic(self, n, seed, first_row=0, normalize=True):
        check(lib().mvdb_index_add_synthetic(self._h, int(n), int(seed), int(first_row), int(bool(normalize))))

    def get_rows(self, row0, n, out=None):
        """Rows [row0, row0+n) as float32 [n,d]; `out` (C-contiguous float32 [n,d]) is filled in place
        when given (no temporary for multi-GB read-backs)."""
        if out is None:
            out = np.empty((int(n), self.d), dtype=np.float32)
        elif not (isinstance(out, np.ndarray) and out.dtype == np.float32 and out.flags["C_CONTIGUOUS"]
                  and out.shape == (int(n), self.d)):
            raise ValueError("out must be a C-contiguous float32 array of shape [n, d]")
        check(lib().mvdb_index_get_rows(self._h, int(row0), int(n), _ptr(out)))
        return out

    def remove_rows(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        check(lib().mvdb_index_remove_rows(self._h, _ptr(rows), rows.shape[0]))

    def search(self, q, k, normalize_q=False):
        q = np.ascontiguousarray(np.atleast_2d(np.asarray(q, dtype=np.float32)))
        if q.shape[1] != self.d:
            raise ValueError(f"query dimension {q.shape[1]} != index dimension {self.d}")
        nq = q.shape[0]
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        check(lib().mvdb_index_search(self._h, _ptr(q), nq, int(k), int(bool(normalize_q)), _ptr(D), _ptr(I)))
        return D, I

    def search_subset(self, q, k, rows, normalize_q=False):
        q = np.ascontiguousarray(np.atleast_2d(np.asarray(q, dtype=np.float32)))
        if q.shape[1] != self.d:
            raise ValueError(f"query dimension {q.shape[1]} != index dimension {self.d}")
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        nq = q.shape[0]
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        check(lib().mvdb_index_search_subset(self._h, _ptr(q), nq, int(k), int(bool(normalize_q)), _ptr(rows),
                                             rows.shape[0], _ptr(D), _ptr(I)))
        return D, I

    def rowset(self, rows, excluded=False):
        """The listed rows (or, excluded=True, every row BUT them) made resident on the device for repeated searches under
        one filter: see RowSet."""
        return RowSet(self, rows, excluded)

    def search_rowset(self, q, k, rowset, normalize_q=False):
        """Search a resident RowSet; labels are ROW NUMBERS of the index."""
        q = np.ascontiguousarray(np.atleast_2d(np.asarray(q, dtype=np.float32)))
        if q.shape[1] != self.d:
            raise ValueError(f"query dimension {q.shape[1]} != index dimension {self.d}")
        nq = q.shape[0]
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        check(lib().mvdb_index_search_rowset(self._h, _ptr(q), nq, int(k), int(bool(normalize_q)), rowset._h, _ptr(D),
                                             _ptr(I)))
        return D, I

    def search_rowset_device(self, q_ptr, nq, k, rowset, D_ptr, I_ptr, stream=0, normalize_q=False, label_offset=0):
        """Device-pointer variant of search_rowset (labels: row numbers + label_offset); enqueues on `stream` and returns."""
        check(lib().mvdb_index_search_rowset_device(
            self._h, ctypes.c_void_p(q_ptr), int(nq), int(k), int(bool(normalize_q)), rowset._h, int(label_offset),
            ctypes.c_void_p(D_ptr), ctypes.c_void_p(I_ptr), ctypes.c_void_p(stream)))

    def search_masked(self, q, k, mask_words, normalize_q=False, labels="rows"):
        """Search the rows whose bit is set in `mask_words` (uint64[(ntotal + 63) // 64], bit r & 63 of word r >> 6 = row r;
        `pack_row_mask` builds it).  labels="rows": row numbers; "positions": positions in the ascending list of the
        selected rows (what search_subset returns for that list)."""
        q = np.ascontiguousarray(np.atleast_2d(np.asarray(q, dtype=np.float32)))
        if q.shape[1] != self.d:
            raise ValueError(f"query dimension {q.shape[1]} != index dimension {self.d}")
        mask_words = np.ascontiguousarray(mask_words, dtype=np.uint64)
        if mask_words.shape[0] < (self.ntotal + 63) // 64:
            raise ValueError("the mask has fewer than (ntotal + 63) // 64 words")
        nq = q.shape[0]
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        check(lib().mvdb_index_search_masked(self._h, _ptr(q), nq, int(k), int(bool(normalize_q)), _ptr(mask_words),
                                             {"positions": 0, "rows": 1}[labels], _ptr(D), _ptr(I)))
        return D, I

    def search_masked_device(self, q_ptr, nq, k, mask_ptr, D_ptr, I_ptr, stream=0, normalize_q=False, labels="rows",
                             label_offset=0):
        """Device-pointer variant of search_masked."""
        check(lib().mvdb_index_search_masked_device(
            self._h, ctypes.c_void_p(q_ptr), int(nq), int(k), int(bool(normalize_q)), ctypes.c_void_p(mask_ptr),
            {"positions": 0, "rows": 1}[labels], int(label_offset), ctypes.c_void_p(D_ptr), ctypes.c_void_p(I_ptr),
            ctypes.c_void_p(stream)))

    def search_device(self, q_ptr, nq, k, D_ptr, I_ptr, stream=0, normalize_q=False, label_offset=0):
        """All buffers are device pointers (ints); enqueues on `stream` and returns."""
        check(lib().mvdb_index_search_device(self._h, ctypes.c_void_p(q_ptr), int(nq), int(k),
                                             int(bool(normalize_q)), int(label_offset), ctypes.c_void_p(D_ptr),
                                             ctypes.c_void_p(I_ptr), ctypes.c_void_p(stream)))


    def search_subset_device(self, q_ptr, nq, k, rows_ptr, m, D_ptr, I_ptr, stream=0, normalize_q=False,
                             map_labels=False, label_offset=0):
        """Device-pointer variant of search_subset (rows already validated and resident on the device)."""
        check(lib().mvdb_index_search_subset_device(
            self._h, ctypes.c_void_p(q_ptr), int(nq), int(k), int(bool(normalize_q)), ctypes.c_void_p(rows_ptr), int(m),
            int(bool(map_labels)), int(label_offset), ctypes.c_void_p(D_ptr), ctypes.c_void_p(I_ptr),
            ctypes.c_void_p(stream)))


class RowSet:
    """mvdb_rowset*: a filter's rows resident on the device (a bitmap for dense / excluded sets, a row list otherwise).
    Valid until the index it was built on gains or loses a row."""

    def __init__(self, index, rows, excluded=False):
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        self._h = ctypes.c_void_p()
        check(lib().mvdb_rowset_create(index._h, _ptr(rows), rows.shape[0], int(bool(excluded)), ctypes.byref(self._h)))

    def __len__(self):
        return int(lib().mvdb_rowset_size(self._h))

    @property
    def is_bitmap(self):
        return bool(lib().mvdb_rowset_is_bitmap(self._h))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().mvdb_rowset_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """RCCL communicator owned by libmvdb.so (mvdb_comm*): one all-gather of packed top-k blocks per query batch."""

    def __init__(self, unique_id, rank, world, device=0):
        self._h = ctypes.c_void_p()
        buf = (ctypes.c_ubyte * 128).from_buffer_copy(bytes(unique_id))
        check(lib().mvdb_comm_create(ctypes.cast(buf, ctypes.c_void_p), int(rank), int(world), int(device),
                                     ctypes.byref(self._h)))
        self.rank, self.world = int(rank), int(world)

    @staticmethod
    def unique_id():
        buf = (ctypes.c_ubyte * 128)()
        check(lib().mvdb_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)))
        return bytes(buf)

    def allgather(self, local_ptr, gathered_ptr, nbytes_per_rank, stream=0):
        check(lib().mvdb_allgather_topk(self._h, ctypes.c_void_p(local_ptr), ctypes.c_void_p(gathered_ptr),
                                        int(nbytes_per_rank), ctypes.c_void_p(stream)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().mvdb_comm_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def normalize_l2(x, device=0):
    """In-place faiss.normalize_L2 equivalent on the GPU for a C-contiguous float32 [n,d] array."""
    if not (isinstance(x, np.ndarray) and x.dtype == np.float32 and x.flags["C_CONTIGUOUS"] and x.ndim == 2):
        raise ValueError("normalize_l2 needs a C-contiguous float32 [n,d] ndarray")
    check(lib().mvdb_normalize_l2(_ptr(x), x.shape[0], x.shape[1], device))
    return x


def prof_enable(on=True):
    check(lib().mvdb_prof_enable(int(bool(on))))


def pack_row_mask(n, rows=None, excluded=None):
    """uint64 words of the row bitmap search_masked takes: the listed `rows` set, or every row but `excluded`."""
    bits = np.zeros((n + 63) // 64 * 64, dtype=np.uint8)
    if rows is not None:
        bits[np.fromiter(rows, dtype=np.int64, count=len(rows)) if not isinstance(rows, np.ndarray) else rows] = 1
    else:
        bits[:n] = 1
        if excluded is not None and len(excluded):
            bits[np.fromiter(excluded, dtype=np.int64, count=len(excluded)) if not isinstance(excluded, np.ndarray)
                 else excluded] = 0
    return np.packbits(bits, bitorder="little").view(np.uint64)


def split_rerun_count():
    return int(lib().mvdb_split_rerun_count())


def rescue_tile_stats():
    """(tiles the rescue launches were handed, tiles they would have scanned without the tile flags) since the library was loaded."""
    a, b = ctypes.c_int64(0), ctypes.c_int64(0)
    check(lib().mvdb_rescue_tile_stats(ctypes.byref(a), ctypes.byref(b)))
    return int(a.value), int(b.value)


def half_eps(d):
    return float(lib().mvdb_half_eps(int(d)))


def half_max_queries(d):
    return int(lib().mvdb_half_max_queries(int(d)))


def encoder_splitk_planes(tokens, n, k, compute_units=256):
    """Planes a small batch's [tokens, n] = A [tokens, k] W^T GEMM of the encoder is split into over K (0: not split)."""
    return int(lib().mvdb_encoder_splitk_planes(int(tokens), int(n), int(k), int(compute_units)))


def encoder_gemm_tile_form(tokens, n, compute_units=256):
    """256 / 192: the 256-row tile form of the split-precision GEMM applies to `tokens` packed tokens; 0: it does not."""
    return int(lib().mvdb_encoder_gemm_tile_form(int(tokens), int(n), int(compute_units)))


def prof_symbol(name):
    """Kernel instantiation last launched under profiling label `name` ("" if none)."""
    buf = ctypes.create_string_buffer(256)
    check(lib().mvdb_prof_symbol(name.encode(), buf, 256))
    return buf.value.decode()


def prof_read(name):
    n = ctypes.c_int64(0)
    ms = ctypes.c_double(0.0)
    check(lib().mvdb_prof_read(name.encode(), ctypes.byref(n), ctypes.byref(ms)))
    return n.value, ms.value
