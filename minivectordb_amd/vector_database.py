"""VectorDatabase — drop-in for ``minivectordb.vector_database.VectorDatabase`` whose numeric
path (normalise, add, flat inner-product search, filtered search) runs on an MI355X through
libmvdb.so instead of faiss-cpu.

Same public methods, arguments, return types and error behaviour as the reference class
(minivectordb/vector_database.py:7-548); host bookkeeping (ids, metadata, inverted index, filters,
autocut, pickle format) is Python as in the reference.  What changed underneath:

* ``faiss.IndexFlatIP`` + ``faiss.normalize_L2``  ->  one device-resident matrix (``FlatIndex``);
  rows are normalised on the device, which is then their ONLY home (``_dbcore._RowStore``: no host
  mirror).  ``get_vector`` / ``embeddings`` / ``persist_to_disk`` read the device rows back, so the
  reference's in-place normalisation side effect (vector_database.py:45) stays visible.
* A write no longer forces an O(N*d) rebuild at the next query (vector_database.py:477-479,
  :42-47): rows stored since the last build wait in host blocks and are uploaded incrementally,
  deleted rows are compacted on the device (no ``np.vstack`` per insert, :72, :107).
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

from ._dbcore import FilterAndRerankMixin, _IdIndex, _RowStore


class VectorDatabase(FilterAndRerankMixin):
    def __init__(self, storage_file='db.pkl', device=0, fast_single_query=False):
        """storage_file: as in the reference (vector_database.py:8).  device: HIP device ordinal.  fast_single_query: every
        find_most_similar call (one query) nominates over an fp16 copy of the rows and is re-scored and certified in fp32 — the
        same ids and distances, about half the time per query from 500,000 rows x 256 / 384 / 512 on, 50 % more device memory
        (include/mvdb.h: mvdb_index_set_option "shadow_single_query")."""
        self._fast_single_query = bool(fast_single_query)
        self.hash_vectorizer = None  # built lazily (sklearn) by hybrid_rerank_results
        self.embedding_size = None
        self.storage_file = storage_file
        self._mat = None  # _RowStore: device-resident rows + rows waiting for the next build
        self.metadata = []  # Stores dictionaries of metadata
        self._ids = _IdIndex()  # row <-> unique id (what the reference keeps as id_map / inverse_id_map)
        self.inverted_index = defaultdict(set)  # Inverted index for metadata
        self.index = None  # device-resident FlatIndex (created on first build)
        self._embeddings_changed = False
        self._device = device
        self.lock = threading.Lock()
        self._load_database()

    # ---- the reference exposes the matrix and both id maps as plain attributes --------------------
    @property
    def embeddings(self):
        return None if self._mat is None else self._mat.materialize(self.index)

    @embeddings.setter
    def embeddings(self, value):
        self._mat = None if value is None else _RowStore.adopt(value)
        if self.index is not None:
            self.index.reset()
        self._embeddings_changed = True

    @property
    def id_map(self):
        """Maps embedding row number to unique id (a dict rebuilt on demand after a write)."""
        return self._ids.row_dict()

    @property
    def inverse_id_map(self):
        """Maps unique id to embedding row number (a dict brought up to date on demand after a delete)."""
        return self._ids.inverse_dict()

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
                self._mat = None if emb is None else _RowStore.adopt(emb)
                self.embedding_size = emb.shape[1] if emb is not None else None
                self.metadata = data['metadata']
                id_map = data['id_map']
                self._ids = _IdIndex(id_map[i] for i in range(len(id_map)))
                self.inverted_index = data.get('inverted_index', defaultdict(set))
                self.__dict__.pop("_values", None)
                self._note_write()
                if self.embedding_size is not None:
                    self._build_index()

    def persist_to_disk(self):
        with self.lock:
            with open(self.storage_file, 'wb') as f:
                data = {
                    'embeddings': None if self._mat is None else np.array(self._mat.materialize(self.index)),
                    'metadata': self.metadata,
                    'id_map': dict(self._ids.row_dict()),
                    'inverse_id_map': dict(self._ids.inverse_dict()),
                    'inverted_index': self.inverted_index
                }
                pickle.dump(data, f)

    # ---- ingest / delete (vector_database.py:49-155) -------------------------------------------------
    def get_vector(self, unique_id):
        with self.lock:
            if unique_id not in self._ids:
                raise ValueError("Unique ID does not exist.")
            return self._mat.row(self._ids.row(unique_id), self.index)

    def store_embedding(self, unique_id, embedding, metadata_dict={}):
        with self.lock:
            if unique_id in self._ids:
                raise ValueError("Unique ID already exists.")
            self._admit([unique_id], [self._convert_ndarray_float32(embedding)], [metadata_dict])

    def store_embeddings_batch(self, unique_ids, embeddings, metadata_dicts=[]):
        with self.lock:
            if any(uid in self._ids for uid in unique_ids):
                raise ValueError("Unique ID already exists.")
            if isinstance(embeddings, np.ndarray) and embeddings.ndim == 2 and len(embeddings) > 1:
                vectors = np.array(embeddings, dtype=np.float32)   # one copy of the whole batch instead of one array per row
            else:
                vectors = self._convert_ndarray_float32_batch(embeddings)
            # like the reference: a partial metadata list is an error, an empty one means "no metadata"
            if 0 < len(metadata_dicts) < len(unique_ids):
                raise ValueError("Metadata dictionaries must be provided for all unique IDs.")
            if metadata_dicts == []:
                metadata_dicts = [{} for _ in unique_ids]
            self._admit(unique_ids, vectors, metadata_dicts)

    def delete_embedding(self, unique_id):
        if unique_id not in self._ids:
            raise ValueError("Unique ID does not exist.")

        with self.lock:
            # rows behind the deleted one move up by one — on the device (tail compaction), in the metadata list and in
            # the id list (C memmoves); the reference rebuilds both of its dicts row by row (:139-152).  Only the keys
            # of the deleted row's own metadata are visited in the inverted index (:128-137).
            self._expel([unique_id])

    # ---- search (vector_database.py:466-536) -----------------------------------------------------------------
    def find_most_similar(self, embedding, metadata_filter=None, exclude_filter=None, or_filters=None, k=5,
                          autocut=False):
        """ or_filters could be a list of dictionaries, where each dictionary contains key-value pairs for OR
        filters, or a single dictionary, which is equivalent to a list with a single dictionary."""
        hits = []
        uids = self._ids.uids
        for row, score in self._nearest_rows(embedding, metadata_filter, exclude_filter, or_filters, k):
            try:  # a row a concurrent delete has just renumbered away is skipped, as in the reference
                hits.append((uids[row], score, self.metadata[row]))
            except (KeyError, IndexError):
                pass
        return self._package(hits, autocut)
