"""Host-side logic shared by VectorDatabase and ShardedVectorDatabase: the growable host matrix,
the Mongo-like metadata filter engine, autocut and hybrid rerank.  Pure Python/numpy bookkeeping
with the semantics of the reference (minivectordb/vector_database.py:157-464, duplicated verbatim in
minivectordb/sharded_vector_database.py:289-596); none of it is on the GPU path.
"""
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


class _AllRows:
    """Symbolic 'every stored row' (what the reference builds as an O(N) set per query)."""

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __bool__(self):
        return self.n > 0

    def materialize(self):
        return set(range(self.n))


class _AllRowsExcept:
    """Symbolic 'every stored row but these' — what an exclude-filter leaves of `_AllRows` (the reference subtracts from an
    O(N) Python set per query, vector_database.py:354-386).  Searched as a resident bitmap (`mvdb_rowset_create(...,
    excluded=1)`): n / 8 bytes up the wire, one full-rate pass over the corpus."""

    def __init__(self, n, removed):
        self.n = n
        self.removed = {r for r in removed if 0 <= r < n}

    def __len__(self):
        return self.n - len(self.removed)

    def __bool__(self):
        return len(self) > 0

    def materialize(self):
        return set(range(self.n)) - self.removed


class _RowStore:
    """The stacked embedding matrix of a database, WITHOUT a host mirror of what the device already holds.

    Rows [0, synced) live — normalised — in the device index only; rows stored since the last index build wait,
    un-normalised, in `pending` host blocks and are uploaded by the next build (`flush`).  `materialize` reads the
    device rows back when somebody asks for the whole matrix (the reference exposes it as ``self.embeddings`` and
    pickles it), `row` fetches one row (``get_vector``), `delete` compacts the device matrix and/or drops pending
    rows.  The reference keeps one numpy matrix that it re-stacks on every insert (``np.vstack``,
    vector_database.py:72) and copies on every delete (``np.delete``, :126); a growable host mirror of that (round 1)
    still paid a 2 GB copy for an append after a batch load and a host memmove per delete at 1M x 512.
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

    def flush(self, index):
        """Upload the pending rows (normalised on the device, vector_database.py:45-46) and forget the host copies."""
        # a block leaves `pending` the moment it is on the device: if a later add raises (hipMalloc while growing),
        # the rows already uploaded are not uploaded again by the next build
        while self.pending:
            block = self.pending[0]
            index.add(block, normalize=True)
            self.pending.pop(0)
            self.npending -= block.shape[0]
            self.synced += block.shape[0]
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
    """row <-> unique id bookkeeping of VectorDatabase with O(tail memmove) deletes.

    The reference keeps two dicts (``id_map`` row -> id, ``inverse_id_map`` id -> row) and rebuilds both over
    ``sorted(id_map)`` at every delete (vector_database.py:139-152): a Python loop over every row behind the deleted
    one, ~0.4 s at 1M rows.  Here the order lives in ONE list (``uids``, row -> id: ``list.pop`` is a C memmove) and
    ids map to stable HANDLES (insertion counters); the current row of a handle is the handle minus the number of
    deleted handles below it (bisect over a short sorted list).  The two dicts the reference exposes are produced on
    demand (`inverse_dict`, `row_dict`) and cached until the next write.
    """

    def __init__(self, uids=()):
        import bisect
        self._bisect = bisect
        self.uids = list(uids)
        self.handle = {u: i for i, u in enumerate(self.uids)}
        self.next = len(self.uids)
        self.deleted = []       # sorted handles removed since the last compaction
        self._row_dict = None

    def __len__(self):
        return len(self.uids)

    def __contains__(self, uid):
        return uid in self.handle

    def append(self, uid):
        self.handle[uid] = self.next
        self.next += 1
        self.uids.append(uid)
        self._row_dict = None

    def row(self, uid):
        h = self.handle[uid]
        return h - self._bisect.bisect_left(self.deleted, h) if self.deleted else h

    def pop(self, uid):
        r = self.row(uid)
        self._bisect.insort(self.deleted, self.handle.pop(uid))
        self.uids.pop(r)
        self._row_dict = None
        if len(self.deleted) > 4096:
            self.compact()
        return r

    def compact(self):
        self.handle = {u: i for i, u in enumerate(self.uids)}
        self.next = len(self.uids)
        self.deleted = []

    def inverse_dict(self):
        """id -> row as a plain dict (what the reference calls inverse_id_map; the filter engine walks it)."""
        if self.deleted:
            self.compact()
        return self.handle

    def row_dict(self):
        """row -> id as a plain dict (the reference's id_map)."""
        if self._row_dict is None:
            self._row_dict = dict(enumerate(self.uids))
        return self._row_dict


class _ExclusionProbe:
    """Collects what an exclude filter would remove from 'all rows' without building that set."""

    def __init__(self):
        self.removed = set()

    def __isub__(self, rows):
        self.removed |= rows
        return self

    def __bool__(self):
        return True


class FilterAndRerankMixin:
    """Needs: self.inverted_index, self.inverse_id_map, self.metadata, self.hash_vectorizer,
    self._mat (_RowStore), self.index, self.embedding_size, self._device."""

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
        if self._mat.n > 0:
            self._mat.flush(self.index)
            self._embeddings_changed = False

    # ---- shared ingest / search plumbing ------------------------------------------------------------
    def _admit(self, unique_ids, vectors, metadata_dicts):
        """Append rows + bookkeeping common to both database classes (caller holds the lock, has
        validated ids).  `vectors` is a sequence of float32 1-D arrays.  Returns the first new row."""
        if self.embedding_size is None:
            self.embedding_size = vectors[0].shape[0]
        if self._mat is None:
            self._mat = _RowStore(self.embedding_size)
        first = self._mat.n
        if len(vectors):
            self._mat.append(vectors[0] if len(vectors) == 1 else np.vstack(vectors))
        self.metadata.extend(metadata_dicts)
        self._note_ids(unique_ids, first)
        for uid, meta in zip(unique_ids, metadata_dicts):
            for key in meta:
                self.inverted_index[key].add(uid)
        self._invalidate_filter_cache()
        self._embeddings_changed = True
        return first

    def _row_count(self):
        return len(self.inverse_id_map)

    def _note_ids(self, unique_ids, first_row):
        """id -> row bookkeeping of newly admitted rows (the sharded class keeps the reference's plain dict)."""
        for row, uid in enumerate(unique_ids, start=first_row):
            self.inverse_id_map[uid] = row

    def _subset_order(self, wanted):
        """Enumeration of a filtered row set handed to the device (positions come back).  The flat class
        uses list(set) like the reference's `self.embeddings[list(filtered)]` (vector_database.py:510)."""
        return list(wanted)

    def _nearest_rows(self, embedding, metadata_filter, exclude_filter, or_filters, k):
        """Device half of find_most_similar: [(row, score)] best first, rows of the stacked matrix.
        Query prep, lazy device sync, filter evaluation, k clamp and the full / filtered branch follow
        vector_database.py:470-523; the arithmetic is libmvdb's."""
        if self._mat is None:
            return []
        query = np.array([np.array(embedding, dtype=np.float32)])  # [1, d]; normalised on the device
        if self._embeddings_changed:
            with self.lock:
                self._build_index()
        with self.lock:
            wanted = self._get_filtered_indices(metadata_filter, exclude_filter, or_filters)
            index, n_rows = self.index, self._mat.n
        if not wanted or index is None:
            return []
        take = min(k, len(wanted))
        if len(wanted) == n_rows:
            scores, rows = index.search(query, take, normalize_q=True)
            return [(int(r), s) for r, s in zip(rows[0], scores[0]) if r != -1]
        for attempt in range(3):
            rowset = self._resident_rowset(index, wanted, (metadata_filter, exclude_filter, or_filters))
            try:
                scores, rows = index.search_rowset(query, take, rowset, normalize_q=True)
                break
            except ValueError:
                # another thread deleted rows between the filter and the search (the reference would still be searching
                # its old index object): evaluate the filter again on the current rows
                if attempt == 2:
                    raise
                with self.lock:
                    self._invalidate_filter_cache()
                    wanted = self._get_filtered_indices(metadata_filter, exclude_filter, or_filters)
                    index = self.index
                if not wanted or index is None:
                    return []
                take = min(k, len(wanted))
        return [(int(r), s) for r, s in zip(rows[0], scores[0]) if r != -1]

    def _resident_rowset(self, index, wanted, filters):
        """The filtered rows as a device-resident row set (`mvdb_rowset`), kept until the next write: the reference
        gathers the filtered rows into a throw-away index for EVERY query (vector_database.py:508-523); here consecutive
        queries under one filter upload nothing.  An exclude-filter over everything travels as the few excluded rows and
        lives as a bitmap; a row list keeps the reference's enumeration order (`_subset_order`: ties resolve as there)."""
        cache = self.__dict__.setdefault("_rowsets", {})
        try:
            key = repr(filters)
        except Exception:
            key = None
        hit = cache.get(key) if key is not None else None
        if hit is not None and hit[0] is index:
            return hit[1]
        if isinstance(wanted, _AllRowsExcept):
            rowset = index.rowset(np.fromiter(wanted.removed, dtype=np.int64, count=len(wanted.removed)), excluded=True)
        else:
            rowset = index.rowset(self._subset_order(wanted))
        if key is not None:
            if len(cache) >= 16:  # a handful of filters in rotation; each holds device memory until its last user drops it
                cache.clear()
            cache[key] = (index, rowset)
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
        """Rows whose metadata[key] satisfies `value` (plain equality, or {"$op": operand} — only the
        FIRST operator of the dict is honoured, as in the reference)."""
        predicate = None
        if operators_allowed and isinstance(value, dict):
            op = next(iter(value))
            operand = value[op]
            func = _OPERATORS.get(op)
            if func is None:
                raise ValueError(f"Invalid operator: {op}")
            predicate = lambda field: func(field, operand)  # noqa: E731
        if predicate is None:
            fast = self._rows_equal_cached(key, value)
            if fast is not None:
                return fast
        rows = set()
        inverse = self.inverse_id_map
        try:
            for uid in self.inverted_index.get(key, set()).copy():
                if uid not in inverse:
                    continue
                row = inverse[uid]
                field = self.metadata[row].get(key, None)
                if (predicate(field) if predicate is not None else field == value):
                    rows.add(row)
        except KeyError:
            rows = set()
        return rows

    def _rows_equal_cached(self, key, value):
        """Equality filters through a per-key value index built on first use and dropped on every write
        (`_invalidate_filter_cache`).  CONTRACT: metadata dicts are treated as immutable once stored — the reference
        re-reads ``self.metadata[row]`` at every query, so an in-place edit of a stored dict is honoured there and is
        NOT seen here until the next store / delete (re-checking every hit costs 4 ms per query at 10,000 hits, ten
        times the search itself).  The reference walks every id that has `key` for each query
        (vector_database.py:283-302: O(N) Python per filtered search); the index keeps, per value, the
        matching rows IN THAT SAME ORDER, so the returned set is built by the same sequence of
        insertions (identical tie order downstream).  Returns None when the value is unhashable."""
        try:
            hash(value)
        except TypeError:
            return None
        cache = self.__dict__.setdefault("_value_index", {})
        entry = cache.get(key)
        if entry is None:
            by_value, unhashable = {}, []
            inverse = self.inverse_id_map
            for uid in self.inverted_index.get(key, set()).copy():
                if uid not in inverse:
                    continue
                row = inverse[uid]
                field = self.metadata[row].get(key, None)
                try:
                    by_value.setdefault(field, []).append(row)
                except TypeError:  # list / dict metadata value: compared with == at query time
                    unhashable.append((row, field))
            entry = cache[key] = (by_value, unhashable)
        by_value, unhashable = entry
        if value != value:  # NaN never equals anything under the reference's `==`; a dict lookup matches it by identity
            return set()
        rows = set(by_value.get(value, ()))
        for row, field in unhashable:
            if field == value:
                rows.add(row)
        return rows

    def _invalidate_filter_cache(self):
        self.__dict__["_value_index"] = {}
        # built against the previous rows: dropped, not closed — a search running outside the lock may still hold one
        # (its device memory goes when the last reference does)
        self.__dict__["_rowsets"] = {}

    def _apply_or_filter(self, or_filters):
        result_indices = set()
        for clause in or_filters:
            for key, value in clause.items():
                result_indices |= self._rows_matching(key, value)
        return result_indices

    def _apply_and_filter(self, and_filters, filtered_indices):
        for clause in and_filters:
            for key, value in clause.items():
                rows = self._rows_matching(key, value)
                if filtered_indices is None:
                    filtered_indices = rows
                else:
                    filtered_indices &= rows
                if not filtered_indices:
                    break
        return filtered_indices

    def _apply_exclude_filter(self, exclude_filter, filtered_indices):
        for clause in exclude_filter:
            for key, value in clause.items():
                filtered_indices -= self._rows_matching(key, value, operators_allowed=False)
                if not filtered_indices:
                    break
        return filtered_indices

    def _get_filtered_indices(self, metadata_filters, exclude_filter, or_filters):
        filtered_indices = _AllRows(self._row_count()) if not metadata_filters else None

        if isinstance(metadata_filters, dict):
            metadata_filters = [metadata_filters]

        if metadata_filters:
            filtered_indices = self._apply_and_filter(metadata_filters, filtered_indices)

        if or_filters:
            if isinstance(or_filters, dict):
                or_filters = [or_filters]
            or_filters = [or_filter for or_filter in or_filters if or_filter]
            if or_filters:
                temp_indices = self._apply_or_filter(or_filters)
                if filtered_indices is None or isinstance(filtered_indices, _AllRows):
                    filtered_indices = temp_indices
                else:
                    filtered_indices &= temp_indices

        if exclude_filter:
            if isinstance(exclude_filter, dict):
                exclude_filter = [exclude_filter]
            if isinstance(filtered_indices, _AllRows):
                # only pay the O(N) set when something is actually excluded
                probe = self._apply_exclude_filter(exclude_filter, _ExclusionProbe())
                if probe.removed:
                    filtered_indices = _AllRowsExcept(filtered_indices.n, probe.removed)
            else:
                filtered_indices = self._apply_exclude_filter(exclude_filter, filtered_indices)

        return filtered_indices if filtered_indices is not None else set()

    # ---- hybrid rerank (vector_database.py:388-441): host-side string work, not on the GPU path ---------
    def _fetch_hash_text_features(self, text):
        if self.hash_vectorizer is None:
            from sklearn.feature_extraction.text import HashingVectorizer
            self.hash_vectorizer = HashingVectorizer(ngram_range=(1, 6), analyzer='char', n_features=64)
        X = self.hash_vectorizer.fit_transform([text])
        return np.sum(X.toarray(), axis=0).tolist()

    def _calculate_text_hash_scores(self, query, documents):
        if len(documents) == 0:
            return []
        query_vector = self._fetch_hash_text_features(query)
        documents_vectors = [self._fetch_hash_text_features(doc) for doc in documents]
        query_vector /= np.linalg.norm(query_vector)
        return [np.dot(query_vector, doc_vector / np.linalg.norm(doc_vector)) for doc_vector in documents_vectors]

    def _calculate_fuzzy_ratios(self, query, documents):
        from ._fuzz import partial_ratio
        return [partial_ratio(query, doc) for doc in documents]

    def hybrid_rerank_results(self, sentences, search_scores, query, k=5, weights=(0.80, 0.15, 0.05)):
        try:
            text_hash_scores = self._calculate_text_hash_scores(query, sentences)
            fuzzy_scores = self._calculate_fuzzy_ratios(query, sentences)

            if len(text_hash_scores) == 0:
                return sentences[:k], search_scores[:k]

            search_weight, text_hash_weight, fuzzy_weight = weights
            combined_scores = (search_weight * np.array(search_scores) + text_hash_weight * np.array(text_hash_scores)
                               + fuzzy_weight * np.array(fuzzy_scores))

            combined_results = np.column_stack((np.array(sentences), np.array(combined_scores)))
            combined_results = combined_results[combined_results[:, 1].argsort()[::-1]]
            sentences, combined_scores = zip(*combined_results)
            return sentences[:k], combined_scores[:k]
        except Exception:
            return sentences[:k], search_scores[:k]

    # ---- autocut (vector_database.py:443-464): kept in numpy.float32 arithmetic -----------------------------
    def autocut_scores(self, score_list):
        score_decreases = []
        for i in range(1, len(score_list)):
            score_decreases.append((score_list[i - 1] - score_list[i]) / score_list[i - 1])

        max_score_decrease = max(score_decreases)

        if max_score_decrease > 0.2:
            return list(range(score_decreases.index(max_score_decrease) + 1, len(score_list)))

        return []

