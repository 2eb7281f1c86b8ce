"""Host-side state shared by VectorDatabase and ShardedVectorDatabase.

The reference keeps ids, metadata and filters in plain Python containers that it walks or rebuilds on every call
(minivectordb/vector_database.py:139-152 renumbering per delete, :157-386 a Python pass over every id per filtered
query, :356 an O(N) set per unfiltered query, :42-47 a full index rebuild after any write; the sharded class duplicates
all of it, sharded_vector_database.py:289-596).  This module keeps the same SEMANTICS — which rows a filter selects,
what errors it raises, what the public attributes show — on structures that are maintained incrementally:

  _RowStore     rows live on the device; rows stored since the last query wait in host blocks (one upload per build)
  _IdIndex      id <-> row through stable handles (no renumbering loop on delete)
  _ValueIndex   per metadata key, value -> handles, appended to on store, filtered through the deleted handles on use
  _Selection    what a filter leaves: everything / everything but a few rows / a sorted row list
  resident row sets (`mvdb_rowset`) cached per filter expression and write generation

Nothing here is on the GPU path; the arithmetic is libmvdb's.  One deliberate difference: a filtered search enumerates its
rows in ASCENDING row order (exact score ties resolve to the lower row, as in the unfiltered search), where the
reference's ``list(set_of_rows)`` (vector_database.py:510) follows CPython's hash-table order.
"""
import bisect
from array import array
from operator import ge, gt, le, lt, ne

import numpy as np

_OPERATORS = {
    "$gt": gt,
    "$gte": ge,
    "$lt": lt,
    "$lte": le,
    "$ne": ne,
    "$in": lambda field, operand: operand in field,  # "operand in metadata value", as the reference
}

_NO_ROWS = np.empty(0, dtype=np.int64)


def _sorted_rows(rows):
    """Any iterable of row numbers -> sorted unique int64 array."""
    if isinstance(rows, np.ndarray):
        return np.unique(rows.astype(np.int64, copy=False))
    return np.array(sorted(set(rows)), dtype=np.int64)


class _Selection:
    """The rows a filter leaves of the n stored ones — never an O(n) Python object:

    rows is None, gone is None   every row        (the reference builds set(range(n)) per query, vector_database.py:356)
    gone = sorted int64 array    every row but these (an exclude-filter over everything; searched as a resident bitmap)
    rows = sorted int64 array    exactly these
    """
    __slots__ = ("n", "rows", "gone")

    def __init__(self, n, rows=None, gone=None):
        self.n = int(n)
        self.rows = rows
        self.gone = gone if gone is not None and len(gone) else None

    def __len__(self):
        if self.rows is not None:
            return int(self.rows.shape[0])
        return self.n - (int(self.gone.shape[0]) if self.gone is not None else 0)

    def __bool__(self):
        return len(self) > 0

    @property
    def everything(self):
        return self.rows is None and self.gone is None

    def materialize(self):
        """The selected rows as a sorted int64 array (O(n) for the symbolic forms: diagnostics and tests only)."""
        if self.rows is not None:
            return self.rows
        every = np.arange(self.n, dtype=np.int64)
        return every if self.gone is None else np.setdiff1d(every, self.gone, assume_unique=True)

    def local(self, lo, hi):
        """The part of the selection inside rows [lo, hi), renumbered from lo (a rank's share in the row-partitioned
        search)."""
        if self.rows is not None:
            a, b = np.searchsorted(self.rows, (lo, hi))
            return _Selection(hi - lo, rows=self.rows[a:b] - lo)
        if self.gone is not None:
            a, b = np.searchsorted(self.gone, (lo, hi))
            return _Selection(hi - lo, gone=self.gone[a:b] - lo)
        return _Selection(hi - lo)


class _RowStore:
    """The stacked embedding matrix of a database, WITHOUT a host mirror of what the device already holds.

    Rows [0, synced) live — normalised — in the device index only; rows stored since the last index build wait,
    un-normalised, in `pending` host blocks and are uploaded by the next build (`flush`, ONE add however many blocks).
    `materialize` reads the device rows back when somebody asks for the whole matrix (the reference exposes it as
    ``self.embeddings`` and pickles it), `row` fetches one row (``get_vector``), `delete` compacts the device matrix
    and/or drops pending rows.  The reference keeps one numpy matrix that it re-stacks on every insert (``np.vstack``,
    vector_database.py:72) and copies on every delete (``np.delete``, :126).
    """

    def __init__(self, d):
        self.d = d
        self.synced = 0      # leading rows resident on the device
        self.pending = []    # float32 [m_i, d] blocks stored since the last build, in row order
        self.npending = 0
        self._cache = None   # materialised matrix, valid until the next mutation

    @classmethod
    def adopt(cls, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        m = cls(arr.shape[1])
        if arr.shape[0]:
            m.pending.append(arr)
            m.npending = arr.shape[0]
        return m

    @property
    def n(self):
        return self.synced + self.npending

    def append(self, rows):
        rows = np.asarray(rows, dtype=np.float32)
        if rows.ndim == 1:
            rows = rows[None, :]
        if rows.shape[1] != self.d:
            # same failure mode as np.vstack in the reference
            raise ValueError(
                f"all the input array dimensions except for the concatenation axis must match exactly, "
                f"but along dimension 1, the array at index 0 has size {self.d} and the array at index 1 "
                f"has size {rows.shape[1]}")
        self.pending.append(rows)
        self.npending += rows.shape[0]
        self._cache = None

    DIRECT_UPLOAD_BYTES = 8 << 20

    def flush(self, index):
        """Upload the pending rows (normalised on the device, vector_database.py:45-46) and forget the host copies:
        one `mvdb_index_add` per run of small blocks — 1,000 single-row stores followed by a query cost one upload and one
        wait on the index's own stream, not 1,000."""
        if not self.pending:
            return
        # runs of SMALL blocks are stacked into one upload; a block of >= 8 MiB goes up as it stands (stacking ten 200 MB
        # batches cost 155 ms of host memcpy before a 37 ms upload)
        groups, small = [], []
        for block in self.pending:
            if block.nbytes >= self.DIRECT_UPLOAD_BYTES:
                if small:
                    groups.append(small)
                    small = []
                groups.append([block])
            else:
                small.append(block)
        if small:
            groups.append(small)
        done = 0
        try:
            for group in groups:
                block = group[0] if len(group) == 1 else np.concatenate(group, axis=0)
                index.add(block, normalize=True)    # a failed add leaves ITS rows (and the later ones) pending
                self.synced += block.shape[0]
                done += len(group)
        finally:
            self.pending = self.pending[done:]
            self.npending = sum(b.shape[0] for b in self.pending)
            self._cache = None

    def delete(self, rows, index):
        """Remove the given stacked row numbers (np.delete semantics: later rows move up)."""
        rows = sorted(int(r) for r in rows)
        dev = [r for r in rows if r < self.synced]
        host = [r - self.synced for r in rows if r >= self.synced]
        if host:
            stacked = self.pending[0] if len(self.pending) == 1 else np.vstack(self.pending)
            kept = np.delete(stacked, host, axis=0)
            self.pending = [kept] if kept.shape[0] else []
            self.npending = kept.shape[0]
        if dev:
            index.remove_rows(dev)
            self.synced -= len(dev)
        self._cache = None

    def row(self, r, index):
        """A fresh copy of stacked row r (the reference hands out a view of an array that every write REPLACES, so
        an earlier result never changes under the caller, vector_database.py:72,104,126)."""
        if r < self.synced:
            return index.get_rows(r, 1)[0]
        r -= self.synced
        for block in self.pending:
            if r < block.shape[0]:
                return block[r].copy()
            r -= block.shape[0]
        raise IndexError("row out of range")

    def materialize(self, index):
        if self._cache is None:
            out = np.empty((self.n, self.d), dtype=np.float32)
            if self.synced:
                index.get_rows(0, self.synced, out=out[:self.synced])
            at = self.synced
            for block in self.pending:
                out[at:at + block.shape[0]] = block
                at += block.shape[0]
            self._cache = out
        return self._cache


class _IdIndex:
    """row <-> unique id bookkeeping with O(tail memmove) deletes, shared by both database classes.

    The reference keeps dicts (``id_map`` row -> id, ``inverse_id_map`` id -> row; the sharded class a ``unique_ids``
    list) and rebuilds them over every row at every delete (vector_database.py:139-152,
    sharded_vector_database.py:229-241): a Python loop over the whole database.  Here the order lives in ONE list
    (``uids``, row -> id: ``list.pop`` is a C memmove) and ids map to stable HANDLES (insertion counters); the current
    row of a handle is the handle minus the number of deleted handles below it (bisect over a short sorted list).
    Handles are renumbered only by `compact` (every 4096 deletes), which bumps `epoch` — whoever stores handles
    (`_ValueIndex`) starts over then.  The plain dicts the reference exposes are produced on demand (`inverse_dict`,
    `row_dict`) and cached until the next write; producing them does NOT touch the handles.
    """

    def __init__(self, uids=()):
        self.uids = list(uids)
        self.handle = {u: i for i, u in enumerate(self.uids)}
        self.next = len(self.uids)
        self.deleted = []       # sorted handles removed since the last compaction
        self.epoch = 0
        self._deleted_arr = None
        self._row_dict = None
        self._inverse = None

    def __len__(self):
        return len(self.uids)

    def __contains__(self, uid):
        return uid in self.handle

    def _touched(self):
        self._row_dict = None
        self._inverse = None

    def append(self, uid):
        self.handle[uid] = self.next
        self.next += 1
        self.uids.append(uid)
        self._touched()

    def extend(self, uids):
        """append() for a whole batch (a later duplicate id takes the handle over, as one append after the other would)."""
        uids = list(uids)
        self.handle.update(zip(uids, range(self.next, self.next + len(uids))))
        self.next += len(uids)
        self.uids.extend(uids)
        self._touched()

    def row(self, uid):
        h = self.handle[uid]
        return h - bisect.bisect_left(self.deleted, h) if self.deleted else h

    def pop(self, uid):
        r = self.row(uid)
        bisect.insort(self.deleted, self.handle.pop(uid))
        self._deleted_arr = None
        self.uids.pop(r)
        self._touched()
        if len(self.deleted) > 4096:
            self.compact()
        return r

    def remove_many(self, uids):
        """Drop several ids; returns their rows (as they were BEFORE the call), ascending.  A handful go one by one; a
        large batch is one pass over the id list."""
        uids = list(dict.fromkeys(uids))
        if len(uids) <= 32:
            rows = sorted(self.row(u) for u in uids)
            for r, u in sorted(((self.row(u), u) for u in uids), reverse=True):
                bisect.insort(self.deleted, self.handle.pop(u))
                self.uids.pop(r)
            self._deleted_arr = None
            self._touched()
            if len(self.deleted) > 4096:
                self.compact()
            return rows
        rows = sorted(self.row(u) for u in uids)
        doomed = set(rows)
        self.uids = [u for r, u in enumerate(self.uids) if r not in doomed]
        self.compact()
        return rows

    def compact(self):
        self.handle = {u: i for i, u in enumerate(self.uids)}
        self.next = len(self.uids)
        self.deleted = []
        self._deleted_arr = None
        self.epoch += 1
        self._touched()

    def rows_of(self, handles):
        """Sorted int64 handles -> the rows of those still alive (sorted, since rows are monotone in handles)."""
        if not self.deleted or not handles.shape[0]:
            return handles
        if self._deleted_arr is None:
            self._deleted_arr = np.array(self.deleted, dtype=np.int64)
        dead = self._deleted_arr
        below = np.searchsorted(dead, handles)
        alive = dead[np.minimum(below, dead.shape[0] - 1)] != handles
        return (handles - below)[alive]

    def inverse_dict(self):
        """id -> row as a plain dict (what the reference calls inverse_id_map), in row order."""
        if self._inverse is None:
            self._inverse = self.handle if not self.deleted else {u: r for r, u in enumerate(self.uids)}
        return self._inverse

    def row_dict(self):
        """row -> id as a plain dict (the reference's id_map)."""
        if self._row_dict is None:
            self._row_dict = dict(enumerate(self.uids))
        return self._row_dict


_NO_FIELD = object()


class _ValueIndex:
    """Equality filters without the reference's pass over every id per query (vector_database.py:283-302, :332-345).

    Per metadata key (built on the first filter that names it, then maintained by `note_store` / `note_delete`):
    value -> the HANDLES of the rows whose metadata[key] == value, ascending (handles only grow, so appending keeps the
    order), plus the rows whose value is unhashable (compared with == at query time).  A delete does not edit the handle
    arrays: dead handles are filtered out through `_IdIndex.rows_of` when the entry is used and disappear for good when
    the id index compacts (its `epoch` moves: the whole value index is rebuilt on demand).

    CONTRACT: metadata dicts are treated as immutable once stored — the reference re-reads ``self.metadata[row]`` at
    every query, so an in-place edit of a stored dict is honoured there and not seen here.
    """

    def __init__(self, ids):
        self.epoch = ids.epoch
        self.keys = {}   # key -> (by_value: {value: array('q') of handles}, odd: {handle: unhashable field})

    def _build(self, key, uids, ids, metadata):
        by_value, odd = {}, {}
        if 2 * len(uids) >= len(metadata) == len(ids.uids):
            # most rows carry the key: walk the rows in order — handles ascend with the rows, so nothing is sorted and no id is
            # looked up (a million rows: ~0.15 s instead of ~0.3 s)
            missing = _NO_FIELD
            handle_of = ids.handle
            handles = range(len(metadata)) if not ids.deleted and ids.next == len(metadata) else [handle_of[u] for u in ids.uids]
            for h, meta in zip(handles, metadata):
                field = meta.get(key, missing)
                if field is missing:
                    continue
                try:
                    slot = by_value.get(field)
                    if slot is None:
                        slot = by_value[field] = array('q')
                    slot.append(h)
                except TypeError:   # list / dict metadata value
                    odd[h] = field
            self.keys[key] = (by_value, odd)
            return by_value, odd
        pairs = []
        for uid in uids:
            if uid in ids.handle:
                pairs.append((ids.handle[uid], metadata[ids.row(uid)].get(key, None)))
        pairs.sort(key=lambda p: p[0])
        for h, field in pairs:
            try:
                slot = by_value.get(field)
                if slot is None:
                    slot = by_value[field] = array('q')
                slot.append(h)
            except TypeError:   # list / dict metadata value
                odd[h] = field
        self.keys[key] = (by_value, odd)
        return by_value, odd

    def rows_equal(self, key, value, owner):
        """Sorted rows whose metadata[key] == value, or None when `value` is unhashable (the caller walks instead)."""
        try:
            hash(value)
        except TypeError:
            return None
        entry = self.keys.get(key)
        if entry is None:
            entry = self._build(key, list(owner.inverted_index.get(key, ())), owner._ids, owner.metadata)
        by_value, odd = entry
        if value != value:  # NaN never equals anything under the reference's `==`; a dict lookup would match it by identity
            return _NO_ROWS
        slot = by_value.get(value)
        rows = owner._ids.rows_of(np.frombuffer(slot, dtype=np.int64).copy()) if slot else _NO_ROWS
        if odd:
            extra = [h for h, field in odd.items() if field == value]
            if extra:
                rows = np.union1d(rows, owner._ids.rows_of(np.array(sorted(extra), dtype=np.int64)))
        return rows

    def note_store(self, handle, meta):
        for key, field in meta.items():
            entry = self.keys.get(key)
            if entry is None:
                continue   # built when a filter first names the key
            try:
                slot = entry[0].get(field)
                if slot is None:
                    slot = entry[0][field] = array('q')
                slot.append(handle)
            except TypeError:
                entry[1][handle] = field

    def note_delete(self, handle, meta):
        for key in meta:
            entry = self.keys.get(key)
            if entry is not None:
                entry[1].pop(handle, None)   # (hashable values: the dead handle is filtered out at query time)


class FilterAndRerankMixin:
    """Needs: self.inverted_index, self._ids (_IdIndex), self.metadata, self.hash_vectorizer, self._mat (_RowStore),
    self.index, self.embedding_size, self._device, self.lock."""

    # ---- device mirror -----------------------------------------------------------------------------
    def _build_index(self):
        """Bring the device matrix up to date (caller holds the lock).

        Reference: IndexFlatIP(d); normalize_L2(self.embeddings) in place; index.add (vector_database.py:42-47).
        Only rows stored since the last build are uploaded; they are normalised on the device, which from then on
        is their only home (`_RowStore`): ``get_vector`` / ``embeddings`` / ``persist_to_disk`` read them back, so the
        in-place normalisation the reference performs on its host matrix stays visible.
        """
        from . import _native
        if self.index is None:
            self.index = _native.FlatIndex(self.embedding_size, metric=_native.METRIC_IP, device=self._device)
            if self.__dict__.get("_fast_single_query"):
                self.index.set_option("shadow_single_query", 1)
        if self._mat.n > 0:
            self._mat.flush(self.index)
            self._embeddings_changed = False

    # ---- shared ingest / delete bookkeeping ------------------------------------------------------------
    def _admit(self, unique_ids, vectors, metadata_dicts):
        """Append rows + bookkeeping common to both database classes (caller holds the lock, has
        validated ids).  `vectors` is a sequence of float32 1-D arrays.  Returns the first new row."""
        if self.embedding_size is None:
            self.embedding_size = vectors[0].shape[0]
        if self._mat is None:
            self._mat = _RowStore(self.embedding_size)
        first = self._mat.n
        if len(vectors):
            if isinstance(vectors, np.ndarray) and vectors.ndim == 2:   # a batch that arrived as ONE float32 matrix: no per-row work
                self._mat.append(vectors[0] if vectors.shape[0] == 1 else vectors)
            else:
                self._mat.append(vectors[0] if len(vectors) == 1 else np.vstack(vectors))
        self.metadata.extend(metadata_dicts)
        values = self._live_value_index()
        handle = self._ids.next
        self._ids.extend(unique_ids)
        inverted = self.inverted_index
        for uid, meta in zip(unique_ids, metadata_dicts):
            if meta:
                for key in meta:
                    inverted[key].add(uid)
                if values is not None:
                    values.note_store(handle, meta)
            handle += 1
        self._note_write()
        self._embeddings_changed = True
        return first

    def _expel(self, unique_ids):
        """Bookkeeping of a delete (caller holds the lock, ids exist): id maps, metadata list, inverted index — touching
        only what belongs to the doomed ids (the reference walks every inverted-index key per deleted id and rebuilds its
        lists and dicts over all rows, vector_database.py:128-152, sharded_vector_database.py:229-241) — and the device /
        pending rows.  Returns the removed rows, ascending."""
        unique_ids = list(dict.fromkeys(unique_ids))
        values = self._live_value_index()
        for uid in unique_ids:
            meta = self.metadata[self._ids.row(uid)]
            if values is not None and meta:
                values.note_delete(self._ids.handle[uid], meta)
            for key in meta:
                holders = self.inverted_index.get(key)
                if holders is not None:
                    holders.discard(uid)
                    if not holders:
                        del self.inverted_index[key]
        rows = self._ids.remove_many(unique_ids)
        if len(rows) <= 32:
            for r in reversed(rows):
                del self.metadata[r]
        else:
            doomed = set(rows)
            self.metadata[:] = [m for r, m in enumerate(self.metadata) if r not in doomed]
        self._mat.delete(rows, self.index)
        self._note_write()
        self._embeddings_changed = True
        return rows

    def _row_count(self):
        return len(self._ids)

    def _note_write(self):
        """Every store / delete: cached row sets and the exposed dicts belong to the previous state.  (The value index is
        NOT dropped: it is maintained incrementally, `_admit` / `_expel`.)"""
        self.__dict__["_write_gen"] = self.__dict__.get("_write_gen", 0) + 1
        # built against the previous rows: dropped, not closed — a search running outside the lock may still hold one
        # (its device memory goes when the last reference does)
        self.__dict__["_rowsets"] = {}

    _invalidate_filter_cache = _note_write

    def _live_value_index(self):
        """The value index if it exists and its handles are current, else None (it is rebuilt by the next filter)."""
        vi = self.__dict__.get("_values")
        if vi is not None and vi.epoch != self._ids.epoch:
            vi = self.__dict__["_values"] = None
        return vi

    # ---- search plumbing ---------------------------------------------------------------------------------
    def _nearest_rows(self, embedding, metadata_filter, exclude_filter, or_filters, k):
        """Device half of find_most_similar: [(row, score)] best first, rows of the stacked matrix.
        Query prep, lazy device sync, filter evaluation, k clamp and the full / filtered branch follow
        vector_database.py:470-523; the arithmetic is libmvdb's."""
        if self._mat is None:
            return []
        query = np.array([np.array(embedding, dtype=np.float32)])  # [1, d]; normalised on the device
        return self._nearest_rows_many(query, metadata_filter, exclude_filter, or_filters, k)[0]

    def find_most_similar_batch(self, embeddings, metadata_filter=None, exclude_filter=None, or_filters=None, k=5,
                                autocut=False):
        """Several queries under ONE filter in one call (no reference counterpart: the reference's API is one query per call):
        element i of the returned list is what ``find_most_similar(embeddings[i], ...)`` returns.  The queries share corpus
        passes on the device (2+ queries: the certified batch passes, also under a filter's resident row set) — 10M x 512:
        128 queries in 1.8 ms against 2.84 ms for one."""
        queries = np.ascontiguousarray(np.asarray(embeddings, dtype=np.float32))
        if queries.ndim != 2:
            raise ValueError("embeddings must be a 2-D array-like, one query per row")
        if queries.shape[0] == 0 or self._mat is None:
            return [([], [], []) for _ in range(queries.shape[0])]
        uids = self._ids.uids
        out = []
        for found in self._nearest_rows_many(queries, metadata_filter, exclude_filter, or_filters, k):
            hits = []
            for row, score in found:
                try:  # a row a concurrent delete has just renumbered away is skipped, as in find_most_similar
                    hits.append((uids[row], score, self.metadata[row]))
                except (KeyError, IndexError):
                    pass
            out.append(self._package(hits, autocut))
        return out

    def _nearest_rows_many(self, query, metadata_filter, exclude_filter, or_filters, k):
        """_nearest_rows for a [nq, d] float32 matrix of queries: one list of (row, score) per query."""
        nq = query.shape[0]
        filtered = bool(metadata_filter or exclude_filter or or_filters)
        key = None
        if filtered:
            try:
                key = repr((metadata_filter, exclude_filter, or_filters))
            except Exception:
                key = None
        for attempt in range(3):
            with self.lock:
                if self._embeddings_changed:
                    self._build_index()
                index, n_rows = self.index, self._mat.n
                gen = self.__dict__.get("_write_gen", 0)
                hit = self.__dict__.get("_rowsets", {}).get(key) if key is not None else None
                if hit is not None and hit[0] == gen and hit[1] is index:
                    count, rowset, wanted = hit[2], hit[3], None
                else:
                    wanted = self._get_filtered_indices(metadata_filter, exclude_filter, or_filters)
                    count, rowset = len(wanted), None
            if not count or index is None:
                return [[] for _ in range(nq)]
            take = min(k, count)
            try:
                if count == n_rows:
                    scores, rows = index.search(query, take, normalize_q=True)
                else:
                    if rowset is None:
                        rowset = self._resident_rowset(index, wanted, key, gen)
                    scores, rows = index.search_rowset(query, take, rowset, normalize_q=True)
                break
            except ValueError:
                # another thread deleted rows between the filter and the search (the reference would still be searching
                # its old index object): evaluate the filter again on the current rows
                if attempt == 2:
                    raise
        return [[(int(r), s) for r, s in zip(rows[i], scores[i]) if r != -1] for i in range(nq)]

    def _resident_rowset(self, index, wanted, key, gen):
        """The filtered rows as a device-resident row set (`mvdb_rowset`), kept until the next write: the reference
        gathers the filtered rows into a throw-away index for EVERY query (vector_database.py:508-523); here consecutive
        queries under one filter upload nothing and do not even evaluate the filter again.  An exclude-filter over
        everything travels as the few excluded rows and lives as a bitmap.  The entry is stamped with the write
        generation captured — under the lock — together with the filter's rows: a set built from rows that a concurrent
        write has since outdated is used for THIS query only and never enters the cache."""
        if wanted.gone is not None:
            rowset = index.rowset(wanted.gone, excluded=True)
        else:
            rowset = index.rowset(wanted.rows)
        if key is not None:
            with self.lock:
                if self.__dict__.get("_write_gen", 0) == gen and self.index is index:
                    cache = self.__dict__.setdefault("_rowsets", {})
                    if len(cache) >= 16:  # a handful of filters in rotation; each holds device memory until its last user drops it
                        cache.clear()
                    cache[key] = (gen, index, len(wanted), rowset)
        return rowset

    def _package(self, hits, autocut):
        """[(id, score, metadata)] -> (ids, distances, metadatas): tuples, three empty lists when there
        is nothing, lists after an autocut (the reference's conventions, vector_database.py:526-536)."""
        if not hits:
            return [], [], []
        ids, distances, metadatas = zip(*hits)
        if autocut and len(distances) > 1:
            dropped = set(self.autocut_scores(distances))
            if dropped:
                keep = [i for i in range(len(ids)) if i not in dropped]
                return [ids[i] for i in keep], [distances[i] for i in keep], [metadatas[i] for i in keep]
        return ids, distances, metadatas

    # ---- metadata filters (semantics of vector_database.py:157-386) ------------------------------------
    def _rows_matching(self, key, value, operators_allowed=True):
        """Sorted rows whose metadata[key] satisfies `value`: plain equality, or {"$op": operand} where operators are
        allowed — only the FIRST operator of the dict is honoured and an unknown one is a ValueError, as in the
        reference (:163-176); exclude-filters compare a dict value by equality (:330-345)."""
        predicate = None
        if operators_allowed and isinstance(value, dict):
            name = next(iter(value))
            test = _OPERATORS.get(name)
            if test is None:
                raise ValueError(f"Invalid operator: {name}")
            operand = value[name]
            predicate = lambda field: test(field, operand)  # noqa: E731
        else:
            values = self._live_value_index()
            if values is None:
                values = self.__dict__["_values"] = _ValueIndex(self._ids)
            rows = values.rows_equal(key, value, self)
            if rows is not None:
                return rows
            predicate = lambda field: field == value  # noqa: E731  (unhashable operand: walk)
        # operators (and unhashable operands) look at every row that has the key, like the reference; an exception of
        # the comparison itself ($gt against None, $in on a number) propagates, as there
        ids, found = self._ids, []
        holders = self.inverted_index.get(key, ())
        if 2 * len(holders) >= len(self.metadata) == len(ids.uids):
            # most rows carry the key: one pass over the rows in order (no id lookups, nothing to sort)
            return np.fromiter((row for row, meta in enumerate(self.metadata) if key in meta and predicate(meta[key])),
                               dtype=np.int64)
        for uid in list(holders):
            if uid in ids.handle:
                row = ids.row(uid)
                if predicate(self.metadata[row].get(key, None)):
                    found.append(row)
        return _sorted_rows(found)

    def _get_filtered_indices(self, metadata_filters, exclude_filter, or_filters):
        """The reference's filter pipeline (vector_database.py:354-386) on sorted row arrays: AND clauses, then OR
        clauses intersected with them, then exclusions.  Returns a `_Selection`.  Control flow that decides which errors
        surface is kept: an AND pass stops reading the keys of one clause once nothing is left (:291-292), an exclusion
        likewise (:347-348), a filter list holding only empty dicts leaves "no selection" behind, which an exclusion then
        trips over with the reference's TypeError."""
        n = self._row_count()
        as_list = lambda f: [f] if isinstance(f, dict) else f  # noqa: E731
        chosen = None        # None: no AND/OR clause has selected anything yet
        if metadata_filters:
            for clause in as_list(metadata_filters):
                for key, value in clause.items():
                    rows = self._rows_matching(key, value)
                    chosen = rows if chosen is None else np.intersect1d(chosen, rows, assume_unique=True)
                    if not len(chosen):
                        break
        selects_all = not metadata_filters   # the reference starts from every row only when no AND filter was given
        if or_filters:
            clauses = [c for c in as_list(or_filters) if c]
            if clauses:
                either = _NO_ROWS
                for clause in clauses:
                    for key, value in clause.items():
                        either = np.union1d(either, self._rows_matching(key, value))
                chosen = either if (chosen is None or selects_all) else np.intersect1d(chosen, either, assume_unique=True)
                selects_all = False
        if exclude_filter:
            if chosen is None and not selects_all:
                # `None -= set` in the reference (metadata_filter=[{}] with an exclusion)
                for clause in as_list(exclude_filter):
                    for key, value in clause.items():
                        self._rows_matching(key, value, operators_allowed=False)
                        raise TypeError("unsupported operand type(s) for -=: 'NoneType' and 'set'")
                return _Selection(n, rows=_NO_ROWS)   # (exclusions without a single key: nothing was ever selected)
            gone = _NO_ROWS
            left = n if chosen is None else len(chosen)
            for clause in as_list(exclude_filter):
                for key, value in clause.items():
                    rows = self._rows_matching(key, value, operators_allowed=False)
                    if chosen is None:
                        gone = np.union1d(gone, rows)
                        left = n - len(gone)
                    else:
                        chosen = np.setdiff1d(chosen, rows, assume_unique=True)
                        left = len(chosen)
                    if not left:
                        break
            if chosen is None:
                return _Selection(n, gone=gone)
        if chosen is None:
            return _Selection(n) if selects_all else _Selection(n, rows=_NO_ROWS)
        return _Selection(n, rows=chosen)

    # ---- hybrid rerank (vector_database.py:388-441): host-side string work, not on the GPU path ---------
    def _fetch_hash_text_features(self, text):
        if self.hash_vectorizer is None:
            from sklearn.feature_extraction.text import HashingVectorizer
            self.hash_vectorizer = HashingVectorizer(ngram_range=(1, 6), analyzer='char', n_features=64)
        counts = self.hash_vectorizer.fit_transform([text]).toarray()
        return np.sum(counts, axis=0).tolist()

    def _calculate_text_hash_scores(self, query, documents):
        """Cosine between the character n-gram hash vectors of the query and of each document."""
        if len(documents) == 0:
            return []
        unit = lambda v: np.asarray(v) / np.linalg.norm(v)  # noqa: E731
        q = unit(self._fetch_hash_text_features(query))
        return [np.dot(q, unit(self._fetch_hash_text_features(doc))) for doc in documents]

    def _calculate_fuzzy_ratios(self, query, documents):
        from ._fuzz import partial_ratio
        return [partial_ratio(query, doc) for doc in documents]

    def hybrid_rerank_results(self, sentences, search_scores, query, k=5, weights=(0.80, 0.15, 0.05)):
        """Blend the search scores with the two lexical scores and re-rank (vector_database.py:413-441).  Kept quirk:
        sentences and blended scores share ONE numpy string table, so the ranking sorts the scores AS STRINGS and the
        returned scores are strings; any failure falls back to the first k inputs."""
        try:
            lexical = self._calculate_text_hash_scores(query, sentences)
            fuzzy = self._calculate_fuzzy_ratios(query, sentences)
            if len(lexical) == 0:
                return sentences[:k], search_scores[:k]
            w_search, w_lexical, w_fuzzy = weights
            blended = w_search * np.array(search_scores) + w_lexical * np.array(lexical) + w_fuzzy * np.array(fuzzy)
            table = np.column_stack((np.array(sentences), np.array(blended)))
            table = table[table[:, 1].argsort()[::-1]]
            return tuple(table[:, 0])[:k], tuple(table[:, 1])[:k]
        except Exception:
            return sentences[:k], search_scores[:k]

    # ---- autocut (vector_database.py:443-464): numpy.float32 arithmetic element by element, as the reference -----------
    def autocut_scores(self, score_list):
        """Indices to drop: everything after the steepest relative drop between neighbours, if that drop exceeds 20 %."""
        drops = [(score_list[i - 1] - score_list[i]) / score_list[i - 1] for i in range(1, len(score_list))]
        steepest = max(drops)
        if steepest > 0.2:
            return list(range(drops.index(steepest) + 1, len(score_list)))
        return []
