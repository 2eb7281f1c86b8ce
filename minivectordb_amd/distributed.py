"""Row-partitioned search across GPUs: one process per GPU, one RCCL all-gather per query batch.

The reference's ``ShardedVectorDatabase`` shards only its pickle files and searches ONE
concatenated matrix (minivectordb/sharded_vector_database.py:45-71, :79-84, :598-662); the result
contract is therefore "top-k of the union".  Here each rank keeps a contiguous row range resident
in its own HBM and answers from it; global top-k  ⊆  union of per-shard top-k, so the exchange is
one all-gather of ``k`` (score, label) pairs per query and shard — 12*k bytes per rank and query,
latency-bound on xGMI — followed by a k-way merge that every rank runs redundantly (no second
collective).  With world == 1 no collective is initialised or issued.

The collective is RCCL's ``ncclAllGather`` issued from inside libmvdb.so (``mvdb_allgather_topk``, a communicator
created from a unique id that rank 0 draws and the launcher's process group hands around) — torch is then not on the
search path at all.  ``MVDB_COLLECTIVE=torch`` (or a non-RCCL process group, e.g. the gloo tests) routes the same
buffer through ``torch.distributed.all_gather_into_tensor`` instead; which one ran is reported by ``.collective``.

The local scan and the merge are injected callables so that the exchange plumbing (packing layout,
label offsets, gather order) is exercised by world_size-2 ``gloo`` tests on CPU, where the HIP
kernels cannot run; the defaults bind to libmvdb.so and have no CPU fallback.
"""
import os
import sys
import ctypes

import torch
import torch.distributed as dist


def _align16(nbytes):
    return (nbytes + 15) // 16 * 16


class PackedTopK:
    """One rank's result block laid out for a single all-gather: [I: nq*k int64 | D: nq*k fp32],
    both parts padded to 16 bytes."""

    def __init__(self, nq, k, device, world=1):
        self.nq, self.k = nq, k
        self.i_bytes = _align16(nq * k * 8)
        self.d_bytes = _align16(nq * k * 4)
        self.nbytes = self.i_bytes + self.d_bytes
        self.buf = torch.zeros(world * self.nbytes, dtype=torch.uint8, device=device)
        self.world = world

    def views(self, slot=0):
        base = slot * self.nbytes
        I = self.buf[base:base + self.nq * self.k * 8].view(torch.int64).view(self.nq, self.k)
        D = self.buf[base + self.i_bytes:base + self.i_bytes + self.nq * self.k * 4].view(torch.float32).view(
            self.nq, self.k)
        return D, I

    @property
    def stride_I(self):
        return self.nbytes // 8

    @property
    def stride_D(self):
        return self.nbytes // 4


def shard_ranges(total_rows, world):
    """Contiguous row ranges, remainder spread over the first ranks: [(first, count), ...]."""
    base, rem = divmod(int(total_rows), int(world))
    out, first = [], 0
    for r in range(world):
        cnt = base + (1 if r < rem else 0)
        out.append((first, cnt))
        first += cnt
    return out


class ShardedSearcher:
    """Search this rank's shard, all-gather the per-shard top-k, merge.

    index          minivectordb_amd._native.FlatIndex holding this rank's rows (or None when
                   `local_search` is injected)
    label_offset   global row number of this rank's first row (default rank * rows_per_rank)
    local_search   callable(q, D_view, I_view, label_offset, rows=None, normalize_q=False): fills the views
                   (device tensors); `rows` = int64 device tensor of LOCAL row numbers to restrict the scan to
    merge          callable(gathered: PackedTopK, D_out, I_out): merges `world` lists
    """

    def __init__(self, index, k, rank=0, world=1, rows_per_rank=None, label_offset=None, device=None, group=None,
                 metric=0, local_search=None, merge=None, collective=None):
        self.index, self.k, self.rank, self.world = index, int(k), int(rank), int(world)
        self.group = group
        self.metric = metric
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        if label_offset is None:
            label_offset = self.rank * int(rows_per_rank or 0)
        self.label_offset = int(label_offset)
        self._local_search = local_search or self._hip_local_search
        self._merge = merge or self._hip_merge
        self._bufs = {}
        self._comm = None
        self.collective = "none"
        if self.world > 1:
            self.collective = self._pick_collective(collective)

    # ---- the exchange: ncclAllGather inside libmvdb.so, or the process group's all-gather -------
    def _pick_collective(self, want):
        want = want or os.environ.get("MVDB_COLLECTIVE")
        on_gpu = self.device.type == "cuda"
        backend = dist.get_backend(self.group) if dist.is_initialized() else None
        if want is None:
            want = "native" if (on_gpu and backend == "nccl") else "torch"
        if want == "torch":
            return "torch.distributed.all_gather_into_tensor"
        if want != "native":
            raise ValueError(f"MVDB_COLLECTIVE must be 'native' or 'torch' (got {want!r})")
        from . import _native
        # Agree FIRST: ncclCommInitRank blocks until every rank has joined, so a rank that cannot bind librccl must be
        # known to all before anybody enters it.
        ok = torch.tensor([1 if _native.lib().mvdb_comm_available() == 0 else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if int(ok.item()) == 0:
            print(f"[mvdb] rank {self.rank}: librccl could not be bound on every rank ({_native.last_error()}); "
                  "using torch.distributed.all_gather_into_tensor", file=sys.stderr, flush=True)
            return "torch.distributed.all_gather_into_tensor"
        try:
            # rank 0 draws the RCCL unique id; the launcher's process group is only the side channel for it
            box = [_native.Comm.unique_id() if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0) if self.group else 0,
                                       group=self.group)
            self._comm = _native.Comm(box[0], self.rank, self.world, device=self.device.index or 0)
            # one probe gather: every rank must see every rank's block in rank order
            mine = torch.full((16,), self.rank, dtype=torch.uint8, device=self.device)
            allb = torch.full((16 * self.world,), 255, dtype=torch.uint8, device=self.device)
            stream = torch.cuda.current_stream().cuda_stream
            self._comm.allgather(mine.data_ptr(), allb.data_ptr(), 16, stream=stream)
            torch.cuda.synchronize()
            want_t = torch.arange(self.world, dtype=torch.uint8, device=self.device).repeat_interleave(16)
            good = torch.tensor([1 if torch.equal(allb, want_t) else 0], dtype=torch.int32, device=self.device)
            dist.all_reduce(good, op=dist.ReduceOp.MIN, group=self.group)
            if int(good.item()) == 0:
                raise RuntimeError("probe all-gather returned the wrong blocks on some rank")
            return "ncclAllGather (mvdb_allgather_topk, libmvdb.so)"
        except Exception as e:  # both routes are RCCL over xGMI; say loudly which one is in use
            print(f"[mvdb] rank {self.rank}: native RCCL communicator unavailable ({e}); "
                  "using torch.distributed.all_gather_into_tensor", file=sys.stderr, flush=True)
            self._comm = None
            return "torch.distributed.all_gather_into_tensor"

    def _all_gather(self, gathered, local):
        if self._comm is not None:
            self._comm.allgather(local.buf.data_ptr(), gathered.buf.data_ptr(), local.nbytes,
                                 stream=torch.cuda.current_stream().cuda_stream)
        else:
            dist.all_gather_into_tensor(gathered.buf, local.buf, group=self.group)

    # ---- defaults: HIP kernels through the C-ABI ------------------------------------------------
    def _hip_local_search(self, q, D, I, label_offset, rows=None, normalize_q=False):
        stream = torch.cuda.current_stream().cuda_stream
        k = D.shape[1]
        if rows is None:
            self.index.search_device(q.data_ptr(), q.shape[0], k, D.data_ptr(), I.data_ptr(), stream=stream,
                                     label_offset=label_offset, normalize_q=normalize_q)
        else:
            self.index.search_subset_device(q.data_ptr(), q.shape[0], k, rows.data_ptr(), rows.shape[0],
                                            D.data_ptr(), I.data_ptr(), stream=stream, normalize_q=normalize_q,
                                            map_labels=True, label_offset=label_offset)

    def _hip_merge(self, gathered, D_out, I_out):
        from . import _native
        D0, I0 = gathered.views(0)
        _native.check(_native.lib().mvdb_merge_topk_device(
            self.metric, self.world, gathered.nq, self.k, ctypes.c_void_p(D0.data_ptr()), gathered.stride_D,
            ctypes.c_void_p(I0.data_ptr()), gathered.stride_I, ctypes.c_void_p(D_out.data_ptr()),
            ctypes.c_void_p(I_out.data_ptr()), self.device.index or 0,
            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def _buffers(self, nq):
        b = self._bufs.get(nq)
        if b is None:
            local = PackedTopK(nq, self.k, self.device, 1)
            gathered = PackedTopK(nq, self.k, self.device, self.world) if self.world > 1 else None
            D_out = torch.empty((nq, self.k), dtype=torch.float32, device=self.device)
            I_out = torch.empty((nq, self.k), dtype=torch.int64, device=self.device)
            b = (local, gathered, D_out, I_out)
            self._bufs[nq] = b
        return b

    def search_device(self, q, rows=None, normalize_q=False):
        """q: [nq, d] float32 tensor on this rank's device (identical on every rank); rows: None, or an int64
        device tensor of this rank's LOCAL row numbers to restrict the scan to (may be empty).
        Returns (D [nq,k], I [nq,k]) device tensors holding the GLOBAL top-k (valid until the next
        call with the same nq).  Nothing on this path touches the host."""
        nq = q.shape[0]
        local, gathered, D_out, I_out = self._buffers(nq)
        D_loc, I_loc = local.views(0)
        if rows is None and not normalize_q:
            self._local_search(q, D_loc, I_loc, self.label_offset)
        else:
            self._local_search(q, D_loc, I_loc, self.label_offset, rows=rows, normalize_q=normalize_q)
        if self.world == 1:
            return D_loc, I_loc
        self._all_gather(gathered, local)
        self._merge(gathered, D_out, I_out)
        return D_out, I_out

    def close(self):
        if self._comm is not None:
            self._comm.close()
            self._comm = None


class DistributedShardedVectorDatabase:
    """Multi-GPU, read-only serving of a reference ``db_shards/`` directory: the drop-in
    ``find_most_similar`` of ``ShardedVectorDatabase`` with the stacked matrix row-partitioned over the
    ranks of a ``torch.distributed`` job (one process per GPU).

    Every rank loads the bookkeeping of ALL shard files (ids, metadata, inverted index — global row
    numbers are the reference's stacking order, sharded_vector_database.py:45-71) but keeps only the
    embeddings of its own contiguous run of shard files (``shard_files_for_rank``) resident in HBM.
    ``find_most_similar`` is SPMD: every rank calls it with the same arguments and gets the same,
    global answer — local scan (full, or restricted to the filtered rows this rank owns) written straight
    into the packed exchange block on the device, ONE all-gather of the per-shard top-k, k-way merge on
    every rank, one copy of the k results to the host.  Exact score ties resolve to the lower global row
    number.  Writes are not supported in this mode (build the directory with ``ShardedVectorDatabase``);
    world * k <= 16384 (one block sorts the gathered lists in LDS).

    `index_factory` / `local_search` / `merge` exist for the world_size-2 ``gloo`` test on CPU.
    """

    def __init__(self, storage_dir='db_shards', rank=None, world=None, device=None, group=None,
                 index_factory=None, local_search=None, merge=None):
        import pickle
        from collections import defaultdict

        import numpy as np

        from ._dbcore import FilterAndRerankMixin, _AllRows
        from .sharded_vector_database import shard_files_for_rank

        self._np = np
        self._AllRows = _AllRows
        if world is None:
            world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank(group) if world > 1 else 0
        self.rank, self.world, self.group = int(rank), int(world), group
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.storage_dir = storage_dir

        files = [f for f in os.listdir(storage_dir) if f.endswith('.pkl')]
        files.sort(key=lambda x: int(x.split('_')[1].split('.')[0]))
        mine = set(shard_files_for_rank(storage_dir, self.rank, self.world))
        self.metadata, self.unique_ids = [], []
        self.inverted_index = defaultdict(set)
        pieces, self.first_row, self.local_rows = [], None, 0
        for fname in files:
            with open(os.path.join(storage_dir, fname), 'rb') as f:
                data = pickle.load(f)
            if fname in mine:
                if self.first_row is None:
                    self.first_row = len(self.unique_ids)
                pieces.append(np.asarray(data['embeddings'], dtype=np.float32))
                self.local_rows += len(data['unique_ids'])
            self.metadata.extend(data['metadata'])
            self.unique_ids.extend(data['unique_ids'])
            for key, value in data['inverted_index'].items():
                self.inverted_index[key].update(value)
            del data
        if self.first_row is None:
            self.first_row = len(self.unique_ids)
        self.inverse_id_map = {uid: i for i, uid in enumerate(self.unique_ids)}
        self.embedding_size = pieces[0].shape[1] if pieces else None
        if self.world > 1:  # ranks without rows still need the dimension
            dims = [None] * self.world
            dist.all_gather_object(dims, self.embedding_size, group=group)
            self.embedding_size = next((x for x in dims if x), None)

        # the reference's filter engine, bound to this object's bookkeeping
        class _Filters(FilterAndRerankMixin):
            pass
        self._filters = _Filters()
        self._filters.inverted_index = self.inverted_index
        self._filters.inverse_id_map = self.inverse_id_map
        self._filters.metadata = self.metadata
        self._filters.hash_vectorizer = None

        self.index = None
        if self.embedding_size is not None:
            if index_factory is None:
                from . import _native
                index_factory = lambda d: _native.FlatIndex(d, device=self.device.index or 0)  # noqa: E731
            self.index = index_factory(self.embedding_size)
            if pieces:
                self.index.add(np.ascontiguousarray(np.concatenate(pieces, axis=0)), normalize=True)
        self._local_search = local_search
        self._merge = merge
        self._searchers = {}

    def autocut_scores(self, score_list):
        return self._filters.autocut_scores(score_list)

    def _searcher(self, k):
        s = self._searchers.get(k)
        if s is None:
            s = ShardedSearcher(self.index, k, rank=self.rank, world=self.world, label_offset=self.first_row,
                                device=self.device, group=self.group, local_search=self._local_search,
                                merge=self._merge)
            self._searchers[k] = s
        return s

    def find_most_similar(self, embedding, metadata_filter=None, exclude_filter=None, or_filters=None, k=5,
                          autocut=False):
        np = self._np
        n_total = len(self.unique_ids)
        if n_total == 0 or self.index is None:
            return [], [], []
        q = np.array([np.array(embedding, dtype=np.float32)])  # query prep of sharded_vector_database.py:602-604
        filtered = self._filters._get_filtered_indices(metadata_filter, exclude_filter, or_filters)
        if not filtered:
            return [], [], []
        search_k = min(k, len(filtered))
        if self.world * search_k > 16384:
            raise NotImplementedError("DistributedShardedVectorDatabase merges at most 16384 / world results per query")
        q_dev = torch.from_numpy(q).to(self.device)
        rows = None
        if len(filtered) != n_total:
            # the filtered rows this rank owns, as LOCAL row numbers, ascending (ties -> lower global row)
            lo, hi = self.first_row, self.first_row + self.local_rows
            mine = np.array(sorted(r for r in filtered if lo <= r < hi), dtype=np.int64) - lo
            rows = torch.from_numpy(mine).to(self.device)
        Dg, Ig = self._searcher(search_k).search_device(q_dev, rows=rows, normalize_q=True)
        Dg, Ig = Dg.cpu().numpy()[0], Ig.cpu().numpy()[0]
        found = [(self.unique_ids[i], d, self.metadata[i]) for i, d in zip(Ig, Dg) if i >= 0]
        ids, distances, metadatas = zip(*found) if found else ([], [], [])
        if autocut and len(distances) > 1:
            remove = self.autocut_scores(distances)
            if remove:
                ids = [ids[i] for i in range(len(ids)) if i not in remove]
                distances = [distances[i] for i in range(len(distances)) if i not in remove]
                metadatas = [metadatas[i] for i in range(len(metadatas)) if i not in remove]
        return ids, distances, metadatas
