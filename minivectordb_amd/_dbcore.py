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


class _HostMatrix:
    """float32 [n,d] matrix with amortised append; `.view` is the ndarray the reference exposes."""

    def __init__(self, d):
        self.d = d
        self.n = 0
        self.buf = np.zeros((0, d), dtype=np.float32)

    @classmethod
    def adopt(cls, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        m = cls(arr.shape[1])
        m.buf = arr
        m.n = arr.shape[0]
        return m

    @property
    def view(self):
        return self.buf[:self.n]

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
        need = self.n + rows.shape[0]
        if need > self.buf.shape[0]:
            cap = max(need, int(self.buf.shape[0] * 1.5) + 16)
            nb = np.empty((cap, self.d), dtype=np.float32)
            nb[:self.n] = self.buf[:self.n]
            self.buf = nb
        self.buf[self.n:need] = rows
        self.n = need

    def delete(self, rows):
        rows = list(rows)
        if len(rows) == 1:
            # one row (delete_embedding): shift the tail in place instead of re-copying the matrix
            r = int(rows[0])
            tail = self.n - 1 - r
            if tail > 0:
                import ctypes
                row_bytes = self.d * 4
                base = self.buf.ctypes.data
                ctypes.memmove(base + r * row_bytes, base + (r + 1) * row_bytes, tail * row_bytes)
            self.n -= 1
            return
        keep = np.ones(self.n, dtype=bool)
        keep[np.asarray(rows, dtype=np.int64)] = False
        kept = self.buf[:self.n][keep]
        self.buf = np.ascontiguousarray(kept)
        self.n = kept.shape[0]


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
    self._mat, self.index, self._synced_rows, self.embedding_size, self._device."""

    # ---- device mirror -----------------------------------------------------------------------------
    def _build_index(self):
        """Bring the device matrix up to date with the host matrix (caller holds the lock).

        Reference: IndexFlatIP(d); normalize_L2(self.embeddings) in place; index.add
        (vector_database.py:42-47).  Only rows appended since the last build are uploaded; they are
        normalised on the device and read back so the host matrix shows the same normalised rows.
        """
        from . import _native
        if self.index is None:
            self.index = _native.FlatIndex(self.embedding_size, metric=_native.METRIC_IP, device=self._device)
        n = self._mat.n
        if n > 0:
            if self._synced_rows < n:
                start = self._synced_rows
                self.index.add(self._mat.buf[start:n], normalize=True)
                self.index.get_rows(start, n - start, out=self._mat.buf[start:n])  # in-place side effect
                self._synced_rows = n
            self._embeddings_changed = False

    # ---- shared ingest / search plumbing ------------------------------------------------------------
    def _admit(self, unique_ids, vectors, metadata_dicts):
        """Append rows + bookkeeping common to both database classes (caller holds the lock, has
        validated ids).  `vectors` is a sequence of float32 1-D arrays.  Returns the first new row."""
        if self.embedding_size is None:
            self.embedding_size = vectors[0].shape[0]
        if self._mat is None:
            self._mat = _HostMatrix(self.embedding_size)
        first = self._mat.n
        if len(vectors):
            self._mat.append(vectors[0] if len(vectors) == 1 else np.vstack(vectors))
        self.metadata.extend(metadata_dicts)
        for row, uid in enumerate(unique_ids, start=first):
            self.inverse_id_map[uid] = row
        for uid, meta in zip(unique_ids, metadata_dicts):
            for key in meta:
                self.inverted_index[key].add(uid)
        self._invalidate_filter_cache()
        self._embeddings_changed = True
        return first

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
        subset = self._subset_order(wanted)
        scores, positions = index.search_subset(query, take, subset, normalize_q=True)
        return [(int(subset[p]), s) for p, s in zip(positions[0], scores[0]) if p != -1]

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
        try:
            for uid in self.inverted_index.get(key, set()).copy():
                if uid not in self.inverse_id_map:
                    continue
                row = self.inverse_id_map[uid]
                field = self.metadata[row].get(key, None)
                if (predicate(field) if predicate is not None else field == value):
                    rows.add(row)
        except KeyError:
            rows = set()
        return rows

    def _rows_equal_cached(self, key, value):
        """Equality filters through a per-key value index built on first use and dropped on every write
        (`_invalidate_filter_cache`).  The reference walks every id that has `key` for each query
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
            for uid in self.inverted_index.get(key, set()).copy():
                if uid not in self.inverse_id_map:
                    continue
                row = self.inverse_id_map[uid]
                field = self.metadata[row].get(key, None)
                try:
                    by_value.setdefault(field, []).append(row)
                except TypeError:  # list / dict metadata value: compared with == at query time
                    unhashable.append((row, field))
            entry = cache[key] = (by_value, unhashable)
        by_value, unhashable = entry
        rows = set(by_value.get(value, ()))
        for row, field in unhashable:
            if field == value:
                rows.add(row)
        return rows

    def _invalidate_filter_cache(self):
        self.__dict__["_value_index"] = {}

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
        filtered_indices = _AllRows(len(self.inverse_id_map)) if not metadata_filters else None

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
                    filtered_indices = filtered_indices.materialize() - probe.removed
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

