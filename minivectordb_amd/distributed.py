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


class Collective:
    """The exchange route of one process group, decided ONCE and shared by every searcher that uses the group:
    RCCL's ``ncclAllGather`` issued from inside libmvdb.so (``native``: one communicator per ``Collective``), or the
    process group's own ``all_gather_into_tensor`` (``torch``).

    want = None (or ``MVDB_COLLECTIVE`` unset): native on an RCCL group with GPU tensors, torch otherwise; a native
    route that cannot be brought up falls back to torch, loudly, on EVERY rank together.
    want = "native" asked for EXPLICITLY (argument or ``MVDB_COLLECTIVE=native``): failure raises on every rank — a
    broken native route must not be masked by the fallback (the bench line's `collective` field would be the only tell).
    """

    TORCH = "torch.distributed.all_gather_into_tensor"
    NATIVE = "ncclAllGather (mvdb_allgather_topk, libmvdb.so)"

    def __init__(self, rank, world, device, group=None, want=None, always=False):
        """always: bring the route up even at world == 1 (a one-rank process group must exist) — how the one-GPU test
        box runs ncclCommInitRank, the probe gather and the agreement logic that an N-GPU job runs."""
        self.rank, self.world, self.device, self.group = int(rank), int(world), device, group
        self.comm = None
        self.name = "none"
        self.details = {}
        if self.world > 1 or always:
            self.name = self._pick(want)

    def _agree(self, ok):
        """MIN over the ranks of a local 0/1 — every rank must reach this call whatever happened to it before."""
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return int(t.item()) == 1

    def _pick(self, want):
        want = want or os.environ.get("MVDB_COLLECTIVE")
        explicit = want == "native"
        on_gpu = self.device.type == "cuda"
        backend = dist.get_backend(self.group) if dist.is_initialized() else None
        if want is None:
            want = "native" if (on_gpu and backend == "nccl") else "torch"
        if want == "torch":
            return self.TORCH
        if want != "native":
            raise ValueError(f"MVDB_COLLECTIVE must be 'native' or 'torch' (got {want!r})")
        from . import _native

        def give_up(why):
            if self.comm is not None:
                self.comm.close()
                self.comm = None
            if explicit:
                raise RuntimeError(f"MVDB_COLLECTIVE=native was requested but the RCCL route of libmvdb.so is "
                                   f"unavailable on rank {self.rank}: {why}")
            print(f"[mvdb] rank {self.rank}: native RCCL communicator unavailable ({why}); using {self.TORCH}",
                  file=sys.stderr, flush=True)
            return self.TORCH

        # Agree FIRST: ncclCommInitRank blocks until every rank has joined, so a rank that cannot bind librccl must be
        # known to all before anybody enters it.
        bound = on_gpu and _native.lib().mvdb_comm_available() == 0
        if not self._agree(bound):
            return give_up("librccl could not be bound on every rank" + ("" if bound else f" ({_native.last_error()})"))
        # From here on a rank that fails still takes part in every collective below: the outcome is decided by ONE
        # MIN all-reduce that all ranks reach, so no rank can end up on a different route than its peers.
        err = None
        box = [None]
        try:
            # rank 0 draws the RCCL unique id; the launcher's process group is only the side channel for it
            if self.rank == 0:
                box[0] = _native.Comm.unique_id()
        except Exception as e:
            err = e
        dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0) if self.group else 0, group=self.group)
        if not self._agree(box[0] is not None):
            return give_up(f"no unique id ({err})")
        good = False
        try:
            self.comm = _native.Comm(box[0], self.rank, self.world, device=self.device.index or 0)
            # one probe gather: every rank must see every rank's block in rank order
            mine = torch.full((16,), self.rank, dtype=torch.uint8, device=self.device)
            allb = torch.full((16 * self.world,), 255, dtype=torch.uint8, device=self.device)
            self.comm.allgather(mine.data_ptr(), allb.data_ptr(), 16, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            want_t = torch.arange(self.world, dtype=torch.uint8, device=self.device).repeat_interleave(16)
            good = bool(torch.equal(allb, want_t))
            if not good:
                err = "probe all-gather returned the wrong blocks"
        except Exception as e:
            err = e
        if not self._agree(good):
            return give_up(err if err is not None else "another rank failed to bring the communicator up")
        self.details = {"unique_id_from": "rank 0 via broadcast_object_list", "probe": "16-byte blocks, rank order verified"}
        return self.NATIVE

    def all_gather(self, gathered, local):
        if self.comm is not None:
            self.comm.allgather(local.buf.data_ptr(), gathered.buf.data_ptr(), local.nbytes,
                                stream=torch.cuda.current_stream().cuda_stream)
        else:
            dist.all_gather_into_tensor(gathered.buf, local.buf, group=self.group)

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None


class ShardedSearcher:
    """Search this rank's shard, all-gather the per-shard top-k, merge.

    index          minivectordb_amd._native.FlatIndex holding this rank's rows (or None when
                   `local_search` is injected)
    label_offset   global row number of this rank's first row (default rank * rows_per_rank)
    local_search   callable(q, D_view, I_view, label_offset, rows=None, normalize_q=False): fills the views
                   (device tensors); `rows` = int64 device tensor of LOCAL row numbers to restrict the scan to
    merge          callable(gathered: PackedTopK, D_out, I_out): merges `world` lists
    collective     None / "native" / "torch" (a route to bring up for this searcher), or a `Collective` that several
                   searchers share (it is then NOT closed by this searcher)
    exchange_always  run the all-gather and the merge even at world == 1 (one-GPU test of the whole exchange path)

    `rows` of search_device / local_search: None, an int64 device tensor of LOCAL row numbers, or a resident row set of
    the index (`_native.RowSet`: a filter's local rows kept on the device across queries).
    """

    def __init__(self, index, k, rank=0, world=1, rows_per_rank=None, label_offset=None, device=None, group=None,
                 metric=0, local_search=None, merge=None, collective=None, exchange_always=False):
        self.index, self.k, self.rank, self.world = index, int(k), int(rank), int(world)
        self._always = bool(exchange_always)
        self.group = group
        self.metric = metric
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        if label_offset is None:
            label_offset = self.rank * int(rows_per_rank or 0)
        self.label_offset = int(label_offset)
        self._local_search = local_search or self._hip_local_search
        self._merge = merge or self._hip_merge
        self._bufs = {}
        self._owns_collective = not isinstance(collective, Collective)
        self._collective = collective if isinstance(collective, Collective) else Collective(
            self.rank, self.world, self.device, group=group, want=collective, always=self._always)
        self.collective = self._collective.name

    def _all_gather(self, gathered, local):
        self._collective.all_gather(gathered, local)

    # ---- defaults: HIP kernels through the C-ABI ------------------------------------------------
    def _hip_local_search(self, q, D, I, label_offset, rows=None, normalize_q=False):
        stream = torch.cuda.current_stream().cuda_stream
        k = D.shape[1]
        if rows is None:
            self.index.search_device(q.data_ptr(), q.shape[0], k, D.data_ptr(), I.data_ptr(), stream=stream,
                                     label_offset=label_offset, normalize_q=normalize_q)
        elif not torch.is_tensor(rows):   # a resident row set: nothing to upload
            self.index.search_rowset_device(q.data_ptr(), q.shape[0], k, rows, D.data_ptr(), I.data_ptr(), stream=stream,
                                            normalize_q=normalize_q, label_offset=label_offset)
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
            gathered = PackedTopK(nq, self.k, self.device, self.world) if (self.world > 1 or self._always) else None
            D_out = torch.empty((nq, self.k), dtype=torch.float32, device=self.device)
            I_out = torch.empty((nq, self.k), dtype=torch.int64, device=self.device)
            b = (local, gathered, D_out, I_out)
            self._bufs[nq] = b
        return b

    def search_device(self, q, rows=None, normalize_q=False):
        """q: [nq, d] float32 tensor on this rank's device (identical on every rank); rows: None, an int64 device
        tensor of this rank's LOCAL row numbers to restrict the scan to (may be empty), or a resident row set.
        Returns (D [nq,k], I [nq,k]) device tensors holding the GLOBAL top-k (valid until the next
        call with the same nq).  Nothing on this path touches the host."""
        nq = q.shape[0]
        local, gathered, D_out, I_out = self._buffers(nq)
        D_loc, I_loc = local.views(0)
        if rows is None and not normalize_q:
            self._local_search(q, D_loc, I_loc, self.label_offset)
        else:
            self._local_search(q, D_loc, I_loc, self.label_offset, rows=rows, normalize_q=normalize_q)
        if self.world == 1 and not self._always:
            return D_loc, I_loc
        self._all_gather(gathered, local)
        self._merge(gathered, D_out, I_out)
        return D_out, I_out

    def close(self):
        if self._owns_collective:
            self._collective.close()
        self._bufs = {}


class DistributedShardedVectorDatabase:
    """Multi-GPU, read-only serving of a reference ``db_shards/`` directory: the drop-in
    ``find_most_similar`` of ``ShardedVectorDatabase`` with the stacked matrix row-partitioned over the
    ranks of a ``torch.distributed`` job (one process per GPU).

    Every rank unpickles ONLY its own contiguous run of shard files (``shard_files_for_rank``; `files_opened` lists
    them) — their embeddings go to its HBM, their bookkeeping (ids, metadata, inverted index) is exchanged file by
    file (one bounded ``all_gather_object`` per round of files, over a gloo side group when the job's group is RCCL, so
    the pickles never stage through HBM) until every rank holds the bookkeeping of ALL shards (global row numbers are the
    reference's stacking order, sharded_vector_database.py:45-71) without any rank reading another rank's embeddings
    from disk (at BASELINE config 4 that would be 164 GB read eight times).
    ``find_most_similar`` is SPMD: every rank calls it with the same arguments and gets the same,
    global answer — local scan (full, or restricted to the filtered rows this rank owns) written straight
    into the packed exchange block on the device, ONE all-gather of the per-shard top-k, k-way merge on
    every rank, one copy of the k results to the host.  A filter's LOCAL rows stay resident on the device
    (`mvdb_rowset`, one per filter expression, as `_dbcore` keeps them for the single-GPU classes): a repeated filter
    evaluates nothing and uploads nothing.  Exact score ties resolve to the lower global row
    number.  Writes are not supported in this mode (build the directory with ``ShardedVectorDatabase``);
    world * k <= 16384 (one block sorts the gathered lists in LDS).  The exchange route (RCCL communicator) is
    brought up once per database and shared by the per-k searchers; `close()` releases it.

    `index_factory` / `local_search` / `merge` exist for the world_size-2 ``gloo`` test on CPU; `collective`
    ("native" | "torch") and `exchange_always` for the one-GPU test of the RCCL route at world = 1.  `fast_single_query`:
    VectorDatabase's switch, for every rank's local index.
    """

    def __init__(self, storage_dir='db_shards', rank=None, world=None, device=None, group=None,
                 index_factory=None, local_search=None, merge=None, collective=None, exchange_always=False,
                 fast_single_query=False):
        import pickle
        from collections import defaultdict

        import numpy as np

        from ._dbcore import FilterAndRerankMixin, _IdIndex
        from .sharded_vector_database import shard_files_for_rank

        self._np = np
        if world is None:
            world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank(group) if world > 1 else 0
        self.rank, self.world, self.group = int(rank), int(world), group
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.storage_dir = storage_dir

        mine = shard_files_for_rank(storage_dir, self.rank, self.world)
        self.files_opened = []
        book, pieces = [], []   # book: this rank's files in order, (unique_ids, metadata, inverted_index)
        for fname in mine:
            with open(os.path.join(storage_dir, fname), 'rb') as f:
                data = pickle.load(f)
            self.files_opened.append(fname)
            pieces.append(np.asarray(data['embeddings'], dtype=np.float32))
            book.append((data['unique_ids'], data['metadata'], dict(data['inverted_index'])))
            del data
        self.local_rows = sum(len(b[0]) for b in book)
        dim = pieces[0].shape[1] if pieces else None
        per_rank, dims = self._exchange_bookkeeping(book, dim)
        self.metadata, unique_ids = [], []
        self.inverted_index = defaultdict(set)
        self.first_row = 0
        for r, rbook in enumerate(per_rank):
            if r == self.rank:
                self.first_row = len(unique_ids)
            for uids, metas, inv in rbook:
                self.metadata.extend(metas)
                unique_ids.extend(uids)
                for key, value in inv.items():
                    self.inverted_index[key].update(value)
        self.embedding_size = next((d for d in dims if d), None)  # ranks without rows still need the dimension
        del per_rank, book

        # the single-GPU classes' filter engine (value index, symbolic selections), bound to this object's bookkeeping
        class _Filters(FilterAndRerankMixin):
            pass
        self._filters = _Filters()
        self._filters._ids = _IdIndex(unique_ids)
        self._filters.inverted_index = self.inverted_index
        self._filters.metadata = self.metadata
        self._filters.hash_vectorizer = None
        self.unique_ids = self._filters._ids.uids

        self.index = None
        if self.embedding_size is not None:
            if index_factory is None:
                from . import _native
                index_factory = lambda d: _native.FlatIndex(d, device=self.device.index or 0)  # noqa: E731
            self.index = index_factory(self.embedding_size)
            if fast_single_query:   # single queries through the certified fp16-shadow pass too (VectorDatabase's switch)
                self.index.set_option("shadow_single_query", 1)
            if pieces:
                self.index.add(np.ascontiguousarray(np.concatenate(pieces, axis=0)), normalize=True)
        self._local_search = local_search
        self._merge = merge
        self._searchers = {}
        self._always = bool(exchange_always)
        self._local_sets = {}   # filter expression -> (global hit count, this rank's rows: None = all / a resident row set)
        self.rowsets_built = 0
        # pinned staging: the query goes up and the k results come down without a pageable bounce
        self._q_pin = self._q_dev = None
        if self.device.type == "cuda" and self.embedding_size is not None:
            self._q_pin = torch.empty((1, self.embedding_size), dtype=torch.float32).pin_memory()
            self._q_dev = torch.empty((1, self.embedding_size), dtype=torch.float32, device=self.device)
        self._out_pin = {}
        # ONE exchange route for the whole database (search_k = min(k, hits) changes with every filter: a communicator per
        # distinct k would put ncclCommInitRank on the query path and never free it)
        self._collective = Collective(self.rank, self.world, self.device, group=group, want=collective, always=self._always)

    @property
    def inverse_id_map(self):
        return self._filters._ids.inverse_dict()

    def _exchange_bookkeeping(self, book, dim):
        """Every rank's per-file (ids, metadata, inverted index), in rank order, WITHOUT one giant object gather: round i
        gathers the i-th file of every rank (a rank that has run out contributes None), so the transient is `world` shard
        files' worth of pickles per round.  On an RCCL job the rounds run over a gloo side group (host memory; an
        `all_gather_object` on RCCL stages every pickle through device tensors padded to the largest)."""
        if self.world == 1:
            return [book], [dim]
        side = self.group
        try:
            if self.group is None and dist.get_backend() == "nccl":
                side = dist.new_group(backend="gloo")
        except Exception as e:   # no gloo in this build: the job's own group serves, chunked all the same
            print(f"[mvdb] rank {self.rank}: gloo side group unavailable ({e}); bookkeeping goes over the job's group",
                  file=sys.stderr, flush=True)
            side = self.group
        heads = [None] * self.world
        dist.all_gather_object(heads, (len(book), dim), group=side)
        per_rank = [[] for _ in range(self.world)]
        for i in range(max(h[0] for h in heads)):
            pieces = [None] * self.world
            dist.all_gather_object(pieces, book[i] if i < len(book) else None, group=side)
            for r, piece in enumerate(pieces):
                if piece is not None:
                    per_rank[r].append(piece)
        if side is not self.group and side is not None:
            try:
                dist.destroy_process_group(side)
            except Exception:
                pass
        return per_rank, [h[1] for h in heads]

    def autocut_scores(self, score_list):
        return self._filters.autocut_scores(score_list)

    def _searcher(self, k):
        s = self._searchers.get(k)
        if s is None:
            if len(self._searchers) >= 64:   # per-k exchange buffers only: cheap to drop and rebuild
                self._searchers.clear()
            s = ShardedSearcher(self.index, k, rank=self.rank, world=self.world, label_offset=self.first_row,
                                device=self.device, group=self.group, local_search=self._local_search,
                                merge=self._merge, collective=self._collective, exchange_always=self._always)
            self._searchers[k] = s
        return s

    def close(self):
        """Release the exchange route (RCCL communicator) and the per-k buffers; the device index stays usable."""
        for s in self._searchers.values():
            s.close()
        self._searchers = {}
        self._local_sets = {}
        self._collective.close()

    def _local_rows_of(self, metadata_filter, exclude_filter, or_filters):
        """(global number of rows the filter selects, this rank's share of them) — the share is None (every local row) or
        a row set resident on this rank's device; evaluated and uploaded ONCE per filter expression (the database is
        read-only in this mode, so the entry never goes stale)."""
        try:
            key = repr((metadata_filter, exclude_filter, or_filters))
        except Exception:
            key = None
        hit = self._local_sets.get(key) if key is not None else None
        if hit is not None:
            return hit
        chosen = self._filters._get_filtered_indices(metadata_filter, exclude_filter, or_filters)
        rows = None
        if len(chosen) and not chosen.everything:
            mine = chosen.local(self.first_row, self.first_row + self.local_rows)
            if not mine.everything:   # (an exclusion may leave this rank's rows whole)
                rows = (self.index.rowset(mine.gone, excluded=True) if mine.gone is not None
                        else self.index.rowset(mine.rows))
                self.rowsets_built += 1
        entry = (len(chosen), rows)
        if key is not None:
            if len(self._local_sets) >= 16:
                self._local_sets.clear()
            self._local_sets[key] = entry
        return entry

    def _host_results(self, Dg, Ig):
        """The k merged results of one query on the host: pinned buffers + one stream wait on a GPU."""
        if self.device.type != "cuda":
            return Dg.numpy()[0], Ig.numpy()[0]
        k = Dg.shape[1]
        pin = self._out_pin.get(k)
        if pin is None:
            pin = self._out_pin[k] = (torch.empty((1, k), dtype=torch.float32).pin_memory(),
                                      torch.empty((1, k), dtype=torch.int64).pin_memory())
        pin[0].copy_(Dg, non_blocking=True)
        pin[1].copy_(Ig, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return pin[0].numpy()[0].copy(), pin[1].numpy()[0].copy()

    def find_most_similar(self, embedding, metadata_filter=None, exclude_filter=None, or_filters=None, k=5,
                          autocut=False):
        np = self._np
        n_total = len(self.unique_ids)
        if n_total == 0 or self.index is None:
            return [], [], []
        q = np.array([np.array(embedding, dtype=np.float32)])  # query prep of sharded_vector_database.py:602-604
        if metadata_filter or exclude_filter or or_filters:
            hits, rows = self._local_rows_of(metadata_filter, exclude_filter, or_filters)
        else:
            hits, rows = n_total, None
        if not hits:
            return [], [], []
        search_k = min(k, hits)
        if self.world * search_k > 16384:
            raise NotImplementedError("DistributedShardedVectorDatabase merges at most 16384 / world results per query")
        if self._q_pin is not None:
            self._q_pin.copy_(torch.from_numpy(q))
            self._q_dev.copy_(self._q_pin, non_blocking=True)
            q_dev = self._q_dev
        else:
            q_dev = torch.from_numpy(q).to(self.device)
        Dg, Ig = self._searcher(search_k).search_device(q_dev, rows=rows, normalize_q=True)
        Dg, Ig = self._host_results(Dg, Ig)
        return self._package_row(Ig, Dg, autocut)

    def _package_row(self, Ig, Dg, autocut):
        """One query's merged (global row, distance) pairs -> (ids, distances, metadatas) in the reference's conventions."""
        found = [(self.unique_ids[i], d, self.metadata[i]) for i, d in zip(Ig, Dg) if i >= 0]
        ids, distances, metadatas = zip(*found) if found else ([], [], [])
        if autocut and len(distances) > 1:
            remove = self.autocut_scores(distances)
            if remove:
                ids = [ids[i] for i in range(len(ids)) if i not in remove]
                distances = [distances[i] for i in range(len(distances)) if i not in remove]
                metadatas = [metadatas[i] for i in range(len(metadatas)) if i not in remove]
        return ids, distances, metadatas

    def find_most_similar_batch(self, embeddings, metadata_filter=None, exclude_filter=None, or_filters=None, k=5,
                                autocut=False):
        """Several queries under ONE filter in one collective call (every rank passes the same queries): element i is what
        ``find_most_similar(embeddings[i], ...)`` returns.  One local batch search per rank, one all-gather of the packed
        per-shard top-k of all queries, one merge (the shape `bench.py --nq N` times)."""
        np = self._np
        queries = np.ascontiguousarray(np.asarray(embeddings, dtype=np.float32))
        if queries.ndim != 2:
            raise ValueError("embeddings must be a 2-D array-like, one query per row")
        nq = queries.shape[0]
        n_total = len(self.unique_ids)
        empty = [([], [], []) for _ in range(nq)]
        if nq == 0 or n_total == 0 or self.index is None:
            return empty
        if metadata_filter or exclude_filter or or_filters:
            hits, rows = self._local_rows_of(metadata_filter, exclude_filter, or_filters)
        else:
            hits, rows = n_total, None
        if not hits:
            return empty
        search_k = min(k, hits)
        if self.world * search_k > 16384:
            raise NotImplementedError("DistributedShardedVectorDatabase merges at most 16384 / world results per query")
        q_dev = torch.from_numpy(queries).to(self.device)
        Dg, Ig = self._searcher(search_k).search_device(q_dev, rows=rows, normalize_q=True)
        Dh, Ih = Dg.cpu().numpy(), Ig.cpu().numpy()
        return [self._package_row(Ih[i], Dh[i], autocut) for i in range(nq)]
