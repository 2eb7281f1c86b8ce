"""ShardedVectorDatabase — drop-in for ``minivectordb.sharded_vector_database.ShardedVectorDatabase``.

In the reference "sharded" means the PICKLE FILES are sharded (``shard_{i}.pkl`` of at most
``shard_size`` rows, sharded_vector_database.py:98-102, :134-178); at load every shard is stacked
into ONE matrix and ONE IndexFlatIP is searched (:45-71, :79-84, :598-662).  This class keeps that
contract — same files, same arguments, same results — and searches the stacked matrix resident in
HBM through libmvdb.so.  Spreading the rows over several GPUs (one process per GPU, RCCL
all-gather of per-shard top-k) is ``minivectordb_amd.distributed.ShardedSearcher``; shard files map
to ranks whole (``shard_files_for_rank``).

Deliberate fix: ``get_vector`` indexes the shard's array with the row's
position INSIDE that shard; the reference uses the global row number (:91-96), which is only right
for shard 0.
"""
import os
import pickle
import threading
from collections import defaultdict

import numpy as np

from ._dbcore import FilterAndRerankMixin, _RowStore


def shard_files_for_rank(storage_dir, rank, world):
    """Whole shard files owned by `rank` when a stored database is spread over `world` GPUs:
    contiguous runs of shard ids, so global row order == reference stacking order."""
    files = [f for f in os.listdir(storage_dir) if f.endswith('.pkl')]
    files.sort(key=lambda x: int(x.split('_')[1].split('.')[0]))
    base, rem = divmod(len(files), world)
    start = rank * base + min(rank, rem)
    return files[start:start + base + (1 if rank < rem else 0)]


class ShardedVectorDatabase(FilterAndRerankMixin):
    def __init__(self, storage_dir='db_shards', shard_size=5000, device=0):
        self.hash_vectorizer = None
        self.embedding_size = None
        self.storage_dir = storage_dir
        self.shard_size = shard_size
        self._mat = None
        self.metadata = []
        self.unique_ids = []
        self.inverse_id_map = {}
        self.inverted_index = defaultdict(set)
        self.index = None
        self._embeddings_changed = False
        self._device = device
        self.lock = threading.Lock()
        self.box_item_map = {}
        self.inverse_box_item_map = {}
        self._load_database()

    @property
    def embeddings(self):
        return None if self._mat is None else self._mat.materialize(self.index)

    @embeddings.setter
    def embeddings(self, value):
        self._mat = None if value is None else _RowStore.adopt(value)
        if self.index is not None:
            self.index.reset()
        self._embeddings_changed = True

    def _convert_from_non_sharded_db(self, non_sharded_db_object):
        embeddings = non_sharded_db_object.embeddings
        metadata = non_sharded_db_object.metadata
        unique_ids = [non_sharded_db_object.id_map[i] for i in range(len(embeddings))]
        self.store_embeddings_batch(unique_ids, embeddings, metadata)
        del non_sharded_db_object

    def _convert_ndarray_float32(self, ndarray):
        return np.array(ndarray, dtype=np.float32)

    def _convert_ndarray_float32_batch(self, ndarrays):
        return [np.array(arr, dtype=np.float32) for arr in ndarrays]

    # ---- shard files (layout of sharded_vector_database.py:134-178) -----------------------------------
    def _shard_path(self, shard_id):
        return os.path.join(self.storage_dir, f'shard_{shard_id}.pkl')

    def _read_shard(self, shard_id):
        path = self._shard_path(shard_id)
        if os.path.exists(path):
            with open(path, 'rb') as f:
                data = pickle.load(f)
            data['inverted_index'] = defaultdict(set, data['inverted_index'])
            return data
        return {'embeddings': np.zeros((0, self.embedding_size), dtype=np.float32), 'metadata': [],
                'unique_ids': [], 'inverted_index': defaultdict(set)}

    def _write_shard(self, shard_id, data):
        out = dict(data)
        out['inverted_index'] = dict(data['inverted_index'])  # plain dict on disk, as the reference
        with open(self._shard_path(shard_id), 'wb') as f:
            pickle.dump(out, f)

    def _load_database(self):
        if not os.path.exists(self.storage_dir):
            os.makedirs(self.storage_dir)

        shard_files = [f for f in os.listdir(self.storage_dir) if f.endswith('.pkl')]
        shard_files.sort(key=lambda x: int(x.split('_')[1].split('.')[0]))

        self.inverted_index = defaultdict(set)
        pieces = []
        for shard_file in shard_files:
            with self.lock:
                with open(os.path.join(self.storage_dir, shard_file), 'rb') as f:
                    data = pickle.load(f)
                pieces.append(np.asarray(data['embeddings'], dtype=np.float32))
                self.metadata.extend(data['metadata'])
                self.unique_ids.extend(data['unique_ids'])
                for key, value in data['inverted_index'].items():
                    self.inverted_index[key].update(value)
                self._update_box_item_map(data['unique_ids'], shard_file)
        if pieces:
            # one concatenation instead of the reference's vstack per shard (O(shards^2) copying)
            self._mat = _RowStore.adopt(np.concatenate(pieces, axis=0))

        self.inverse_id_map = {uid: i for i, uid in enumerate(self.unique_ids)}

        if self._mat is not None and self._mat.n > 0:
            self.embedding_size = self._mat.d
            with self.lock:
                self._build_index()

    def _update_box_item_map(self, unique_ids, shard_file):
        shard_id = int(os.path.basename(shard_file).split('_')[1].split('.')[0])
        self.box_item_map[shard_id] = unique_ids
        for uid in unique_ids:
            self.inverse_box_item_map[uid] = shard_id

    def get_vector(self, unique_id):
        with self.lock:
            if unique_id not in self.inverse_id_map:
                raise ValueError("Unique ID does not exist.")
            shard_id = self.inverse_box_item_map[unique_id]
            with open(self._shard_path(shard_id), 'rb') as f:
                data = pickle.load(f)
            return data['embeddings'][data['unique_ids'].index(unique_id)]

    def _get_available_shard_id(self):
        for shard_id, items in self.box_item_map.items():
            if len(items) < self.shard_size:
                return shard_id
        return len(self.box_item_map)

    # ---- ingest (sharded_vector_database.py:104-132, :243-287) -----------------------------------------
    def _assign_to_shards(self, unique_ids, vectors, metadata_dicts):
        """First non-full shard (dict order) per row, then ONE rewrite per touched shard file
        (sharded_vector_database.py:98-102, :276-287)."""
        groups = defaultdict(list)
        for uid, vec, meta in zip(unique_ids, vectors, metadata_dicts):
            shard_id = self._get_available_shard_id()
            groups[shard_id].append((uid, vec, meta))
            self.box_item_map.setdefault(shard_id, []).append(uid)
            self.inverse_box_item_map[uid] = shard_id
        for shard_id, items in groups.items():
            uids, vecs, metas = zip(*items)
            self._persist_to_shard_multiple(shard_id, list(uids), list(vecs), list(metas))

    def store_embedding(self, unique_id, embedding, metadata_dict={}):
        with self.lock:
            if unique_id in self.inverse_id_map:
                raise ValueError("Unique ID already exists.")
            vector = self._convert_ndarray_float32(embedding)
            self._admit([unique_id], [vector], [metadata_dict])
            self.unique_ids.append(unique_id)
            self._assign_to_shards([unique_id], [vector], [metadata_dict])

    def _persist_to_shard(self, shard_id, unique_id, embedding, metadata_dict):
        self._persist_to_shard_multiple(shard_id, [unique_id], [embedding], [metadata_dict])

    def _persist_to_shard_multiple(self, shard_id, unique_ids, embeddings, metadata_dicts):
        data = self._read_shard(shard_id)
        data['embeddings'] = np.vstack([data['embeddings']] + [np.atleast_2d(e) for e in embeddings])
        data['metadata'].extend(metadata_dicts)
        data['unique_ids'].extend(unique_ids)
        for metadata_dict, unique_id in zip(metadata_dicts, unique_ids):
            for key in metadata_dict:
                data['inverted_index'][key].add(unique_id)
        self._write_shard(shard_id, data)

    def store_embeddings_batch(self, unique_ids: list, embeddings, metadata_dicts=[]):
        with self.lock:
            if len(unique_ids) != len(embeddings):
                raise ValueError("Number of unique IDs must match number of embeddings.")
            vectors = self._convert_ndarray_float32_batch(embeddings)
            for uid in unique_ids:
                if uid in self.inverse_id_map:
                    raise ValueError(f"Unique ID {uid} already exists.")
            missing = len(unique_ids) - len(metadata_dicts)
            if missing > 0:  # pads the CALLER's list in place, like the reference (:260-261)
                metadata_dicts.extend({} for _ in range(missing))
            self._admit(unique_ids, vectors, metadata_dicts)
            self.unique_ids.extend(unique_ids)
            self._assign_to_shards(unique_ids, vectors, metadata_dicts)

    # ---- delete (sharded_vector_database.py:180-241) ------------------------------------------------------
    def _remove_embeddings_from_shard(self, shard_id, unique_ids):
        with open(self._shard_path(shard_id), 'rb') as f:
            data = pickle.load(f)

        doomed = set(unique_ids)
        keep = [i for i, uid in enumerate(data['unique_ids']) if uid not in doomed]
        data['embeddings'] = data['embeddings'][keep]
        data['metadata'] = [data['metadata'][i] for i in keep]
        data['unique_ids'] = [data['unique_ids'][i] for i in keep]

        for uid in doomed:
            for key, ids in list(data['inverted_index'].items()):
                if uid in ids:
                    ids.discard(uid)
                    if not ids:
                        del data['inverted_index'][key]

        with open(self._shard_path(shard_id), 'wb') as f:
            pickle.dump(data, f)

        self.box_item_map[shard_id] = data['unique_ids']
        for uid in doomed:
            del self.inverse_box_item_map[uid]

    def delete_embeddings_batch(self, unique_ids):
        with self.lock:
            if not isinstance(unique_ids, list):
                unique_ids = [unique_ids]

            if not unique_ids:
                raise ValueError("No unique IDs provided.")

            if not all(uid in self.inverse_id_map for uid in unique_ids):
                raise ValueError("One or more unique IDs do not exist.")

            unique_ids = [uid for uid in unique_ids if uid is not None]

            shard_groups = defaultdict(list)
            for unique_id in unique_ids:
                shard_groups[self.inverse_box_item_map[unique_id]].append(unique_id)
            for shard_id, shard_unique_ids in shard_groups.items():
                self._remove_embeddings_from_shard(shard_id, shard_unique_ids)

            doomed = set(unique_ids)
            rows = sorted({self.inverse_id_map[uid] for uid in doomed})
            self._mat.delete(rows, self.index)
            keep = [i for i, uid in enumerate(self.unique_ids) if uid not in doomed]
            self.metadata = [self.metadata[i] for i in keep]
            self.unique_ids = [self.unique_ids[i] for i in keep]

            for uid in unique_ids:
                for key, ids in list(self.inverted_index.items()):
                    ids.discard(uid)
                    if not ids:
                        del self.inverted_index[key]

            self.inverse_id_map = {uid: i for i, uid in enumerate(self.unique_ids)}
            self._invalidate_filter_cache()
            self._embeddings_changed = True

    # ---- search (sharded_vector_database.py:598-662) --------------------------------------------------------
    def _subset_order(self, wanted):
        return np.array(list(wanted), dtype=np.int32)  # the int32 row list the reference feeds np.take (:636)

    def find_most_similar(self, embedding, metadata_filter=None, exclude_filter=None, or_filters=None, k=5,
                          autocut=False):
        hits = [(self.unique_ids[row], score, self.metadata[row])
                for row, score in self._nearest_rows(embedding, metadata_filter, exclude_filter, or_filters, k)]
        return self._package(hits, autocut)
