"""VectorDatabase — drop-in for ``minivectordb.vector_database.VectorDatabase`` whose numeric
path (normalise, add, flat inner-product search, filtered search) runs on an MI355X through
libmvdb.so instead of faiss-cpu.

Same public methods, arguments, return types and error behaviour as the reference class
(minivectordb/vector_database.py:7-548); host bookkeeping (ids, metadata, inverted index, filters,
autocut, pickle format) is Python as in the reference.  What changed underneath:

* ``faiss.IndexFlatIP`` + ``faiss.normalize_L2``  ->  one device-resident matrix (``FlatIndex``);
  rows are normalised on the device and the normalised rows are copied back into the host matrix,
  reproducing the reference's in-place side effect (vector_database.py:45) that ``get_vector`` and
  ``persist_to_disk`` expose.
* A write no longer forces an O(N*d) rebuild at the next query (vector_database.py:477-479,
  :42-47): appended rows are uploaded incrementally, deleted rows are compacted on the device.
* The host matrix grows geometrically instead of ``np.vstack`` per insert (:72, :107).
* With no filter the reference materialises ``set(inverse_id_map.values())`` per query (:356);
  here "all rows" is represented symbolically.
* The filtered branch searches the listed rows of the resident corpus in place
  (``mvdb_index_search_subset``) instead of gathering them into a throw-away index (:510-514).

There is no CPU fallback: searching requires the HIP library and a GPU.
"""
import os
import pickle
import threading
from collections import defaultdict
import numpy as np

from ._dbcore import FilterAndRerankMixin, _HostMatrix


class VectorDatabase(FilterAndRerankMixin):
    def __init__(self, storage_file='db.pkl', device=0):
        self.hash_vectorizer = None  # built lazily (sklearn) by hybrid_rerank_results
        self.embedding_size = None
        self.storage_file = storage_file
        self._mat = None
        self.metadata = []  # Stores dictionaries of metadata
        self.id_map = {}  # Maps embedding row number to unique id
        self.inverse_id_map = {}  # Maps unique id to embedding row number
        self.inverted_index = defaultdict(set)  # Inverted index for metadata
        self.index = None  # device-resident FlatIndex (created on first build)
        self._synced_rows = 0  # leading host rows mirrored (normalised) on the device
        self._embeddings_changed = False
        self._device = device
        self.lock = threading.Lock()
        self._load_database()

    # ---- the reference exposes the matrix as a plain attribute -----------------------------------
    @property
    def embeddings(self):
        return None if self._mat is None else self._mat.view

    @embeddings.setter
    def embeddings(self, value):
        self._mat = None if value is None else _HostMatrix.adopt(value)
        self._synced_rows = 0
        if self.index is not None:
            self.index.reset()
        self._embeddings_changed = True

    def _convert_ndarray_float32(self, ndarray):
        return np.array(ndarray, dtype=np.float32)

    def _convert_ndarray_float32_batch(self, ndarrays):
        return [np.array(arr, dtype=np.float32) for arr in ndarrays]

    # ---- persistence: same pickle layout as the reference (vector_database.py:28-40, :538-548) ----
    def _load_database(self):
        if os.path.exists(self.storage_file):
            with self.lock:
                with open(self.storage_file, 'rb') as f:
                    data = pickle.load(f)
                emb = data['embeddings']
                self._mat = None if emb is None else _HostMatrix.adopt(emb)
                self.embedding_size = emb.shape[1] if emb is not None else None
                self.metadata = data['metadata']
                self.id_map = data['id_map']
                self.inverse_id_map = data['inverse_id_map']
                self.inverted_index = data.get('inverted_index', defaultdict(set))
                self._invalidate_filter_cache()
                self._synced_rows = 0
                if self.embedding_size is not None:
                    self._build_index()

    def persist_to_disk(self):
        with self.lock:
            with open(self.storage_file, 'wb') as f:
                data = {
                    'embeddings': None if self._mat is None else np.array(self._mat.view),
                    'metadata': self.metadata,
                    'id_map': self.id_map,
                    'inverse_id_map': self.inverse_id_map,
                    'inverted_index': self.inverted_index
                }
                pickle.dump(data, f)

    # ---- ingest / delete (vector_database.py:49-155) -------------------------------------------------
    def get_vector(self, unique_id):
        with self.lock:
            if unique_id not in self.inverse_id_map:
                raise ValueError("Unique ID does not exist.")
            row_num = self.inverse_id_map[unique_id]
            # a copy: the reference hands out a view of an array that np.delete/np.vstack REPLACE on every
            # write (vector_database.py:72,126), so an earlier result never changes under the caller; the
            # growable buffer here is edited in place
            return self._mat.view[row_num].copy()

    def store_embedding(self, unique_id, embedding, metadata_dict={}):
        with self.lock:
            if unique_id in self.inverse_id_map:
                raise ValueError("Unique ID already exists.")
            row = self._admit([unique_id], [self._convert_ndarray_float32(embedding)], [metadata_dict])
            self.id_map[row] = unique_id

    def store_embeddings_batch(self, unique_ids, embeddings, metadata_dicts=[]):
        with self.lock:
            if any(uid in self.inverse_id_map for uid in unique_ids):
                raise ValueError("Unique ID already exists.")
            vectors = self._convert_ndarray_float32_batch(embeddings)
            # like the reference: a partial metadata list is an error, an empty one means "no metadata"
            if 0 < len(metadata_dicts) < len(unique_ids):
                raise ValueError("Metadata dictionaries must be provided for all unique IDs.")
            if metadata_dicts == []:
                metadata_dicts = [{} for _ in unique_ids]
            first = self._admit(unique_ids, vectors, metadata_dicts)
            self.id_map.update(zip(range(first, first + len(vectors)), unique_ids))

    def delete_embedding(self, unique_id):
        if unique_id not in self.inverse_id_map:
            raise ValueError("Unique ID does not exist.")

        with self.lock:
            row_num = self.inverse_id_map[unique_id]
            self._mat.delete([row_num])
            if row_num < self._synced_rows:
                # keep the device mirror aligned with np.delete's renumbering
                self.index.remove_rows([row_num])
                self._synced_rows -= 1
            metadata_to_delete = self.metadata.pop(row_num)

            for key in metadata_to_delete:
                self.inverted_index[key].discard(unique_id)
                if not self.inverted_index[key]:
                    del self.inverted_index[key]

            del self.inverse_id_map[unique_id]

            # renumber: rows after the deleted one move up by one.  Same resulting maps as the
            # reference's rebuild over sorted(id_map) (:139-152), done in place over the shifted tail only.
            n_old = len(self.id_map)
            for old_index in range(row_num + 1, n_old):
                uid = self.id_map[old_index]
                self.id_map[old_index - 1] = uid
                self.inverse_id_map[uid] = old_index - 1
            del self.id_map[n_old - 1]

            self._invalidate_filter_cache()
            self._embeddings_changed = True

    # ---- search (vector_database.py:466-536) -----------------------------------------------------------------
    def find_most_similar(self, embedding, metadata_filter=None, exclude_filter=None, or_filters=None, k=5,
                          autocut=False):
        """ or_filters could be a list of dictionaries, where each dictionary contains key-value pairs for OR
        filters, or a single dictionary, which is equivalent to a list with a single dictionary."""
        hits = []
        for row, score in self._nearest_rows(embedding, metadata_filter, exclude_filter, or_filters, k):
            try:  # a row a concurrent delete has just renumbered away is skipped, as in the reference
                hits.append((self.id_map[row], score, self.metadata[row]))
            except (KeyError, IndexError):
                pass
        return self._package(hits, autocut)
