"""ShardedVectorDatabase — drop-in for ``minivectordb.sharded_vector_database.ShardedVectorDatabase``.

In the reference "sharded" means the PICKLE FILES are sharded (``shard_{i}.pkl`` of at most
``shard_size`` rows, sharded_vector_database.py:98-102, :134-178); at load every shard is stacked
into ONE matrix and ONE IndexFlatIP is searched (:45-71, :79-84, :598-662).  This class keeps that
contract — same files, same arguments, same results — and searches the stacked matrix resident in
HBM through libmvdb.so.  Spreading the rows over several GPUs (one process per GPU, RCCL
all-gather of per-shard top-k) is ``minivectordb_amd.distributed.ShardedSearcher``; shard files map
to ranks whole (``shard_files_for_rank``).

Host bookkeeping is this package's own (``_dbcore``): ids live in an ``_IdIndex`` (stable handles, no renumbering loop),
a delete touches the doomed rows' own metadata keys only, equality filters read an incrementally maintained value index,
and the first shard with room is found from a cursor instead of a scan over every shard per stored row.  What the
reference rebuilds over ALL rows on every delete — ``metadata``, ``unique_ids``, ``inverse_id_map`` and a walk over every
inverted-index key per deleted id (sharded_vector_database.py:225-241) — is not rebuilt; the shard FILE a delete touches
is still rewritten whole, as there (:180-204): the files are the persistence contract.

Deliberate fix: ``get_vector`` indexes the shard's array with the row's
position INSIDE that shard; the reference uses the global row number (:91-96), which is only right
for shard 0.
"""
import os
import pickle
import threading
from collections import defaultdict

import numpy as np

from ._dbcore import FilterAndRerankMixin, _IdIndex, _RowStore


def _shard_number(file_name):
    return int(os.path.basename(file_name).split('_')[1].split('.')[0])


def _shard_file_names(storage_dir):
    """``shard_<i>.pkl`` files of a directory in shard order (the reference's stacking order, :41-42)."""
    return sorted((f for f in os.listdir(storage_dir) if f.endswith('.pkl')), key=_shard_number)


def shard_files_for_rank(storage_dir, rank, world):
    """Whole shard files owned by `rank` when a stored database is spread over `world` GPUs:
    contiguous runs of shard ids, so global row order == reference stacking order."""
    files = _shard_file_names(storage_dir)
    base, rem = divmod(len(files), world)
    start = rank * base + min(rank, rem)
    return files[start:start + base + (1 if rank < rem else 0)]


class ShardedVectorDatabase(FilterAndRerankMixin):
    def __init__(self, storage_dir='db_shards', shard_size=5000, device=0, fast_single_query=False):
        """storage_dir, shard_size: as in the reference (sharded_vector_database.py:9).  device, fast_single_query: see
        VectorDatabase."""
        self._fast_single_query = bool(fast_single_query)
        self.hash_vectorizer = None
        self.embedding_size = None
        self.storage_dir = storage_dir
        self.shard_size = shard_size
        self._mat = None
        self.metadata = []
        self._ids = _IdIndex()
        self.inverted_index = defaultdict(set)
        self.index = None
        self._embeddings_changed = False
        self._device = device
        self.lock = threading.Lock()
        self.box_item_map = {}          # shard id -> ids stored in that file, file order
        self.inverse_box_item_map = {}  # id -> shard id
        self._shard_order = []          # shard ids in box_item_map (= first-fit) order
        self._first_open = 0            # shards before this position in _shard_order are full
        self._load_database()

    # ---- what the reference exposes as plain attributes ----------------------------------------------------
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
    def unique_ids(self):
        """Stacked row -> id (the list itself: row order, kept by the id index)."""
        return self._ids.uids

    @property
    def inverse_id_map(self):
        """id -> stacked row as a plain dict in row order (brought up to date on demand after a delete)."""
        return self._ids.inverse_dict()

    def _convert_from_non_sharded_db(self, non_sharded_db_object):
        rows = non_sharded_db_object.embeddings
        ids_by_row = non_sharded_db_object.id_map
        self.store_embeddings_batch([ids_by_row[r] for r in range(len(rows))], rows, non_sharded_db_object.metadata)
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
        """Stack every shard file in shard order (sharded_vector_database.py:37-71): ONE concatenation of the pieces
        (the reference re-stacks the growing matrix per shard), ids and metadata appended per file, the inverted index
        merged, the file -> ids maps filled."""
        os.makedirs(self.storage_dir, exist_ok=True)
        self.inverted_index = defaultdict(set)
        pieces, ids = [], []
        for name in _shard_file_names(self.storage_dir):
            with self.lock:
                with open(os.path.join(self.storage_dir, name), 'rb') as f:
                    shard = pickle.load(f)
            pieces.append(np.asarray(shard['embeddings'], dtype=np.float32))
            self.metadata.extend(shard['metadata'])
            ids.extend(shard['unique_ids'])
            for key, holders in shard['inverted_index'].items():
                self.inverted_index[key].update(holders)
            self._update_box_item_map(shard['unique_ids'], name)
        self._ids = _IdIndex(ids)
        self._first_open = 0
        if pieces:
            self._mat = _RowStore.adopt(np.concatenate(pieces, axis=0))
        if self._mat is not None and self._mat.n > 0:
            self.embedding_size = self._mat.d
            with self.lock:
                self._build_index()

    def _update_box_item_map(self, unique_ids, shard_file):
        shard_id = _shard_number(shard_file)
        if shard_id not in self.box_item_map:
            self._shard_order.append(shard_id)
        self.box_item_map[shard_id] = unique_ids
        self.inverse_box_item_map.update(dict.fromkeys(unique_ids, shard_id))

    def get_vector(self, unique_id):
        with self.lock:
            if unique_id not in self._ids:
                raise ValueError("Unique ID does not exist.")
            shard = self._read_shard(self.inverse_box_item_map[unique_id])
            return shard['embeddings'][shard['unique_ids'].index(unique_id)]

    def _get_available_shard_id(self):
        """First shard, in the order the shards entered `box_item_map`, that holds fewer than `shard_size` ids; when all
        are full, the NUMBER of shards (the reference's rule, sharded_vector_database.py:98-102, which it evaluates by
        scanning every shard for every stored row).  Shards only fill up between deletes, so a cursor over the order
        does: everything before it is full."""
        order, at = self._shard_order, self._first_open
        while at < len(order) and len(self.box_item_map[order[at]]) >= self.shard_size:
            at += 1
        self._first_open = at
        return order[at] if at < len(order) else len(self.box_item_map)

    # ---- ingest (sharded_vector_database.py:104-132, :243-287) -----------------------------------------
    def _assign_to_shards(self, unique_ids, vectors, metadata_dicts):
        """One shard per row by first fit, then ONE rewrite per touched shard file (:276-287)."""
        if isinstance(vectors, np.ndarray) and vectors.ndim == 2 and len(unique_ids) == vectors.shape[0] == len(metadata_dicts):
            # a batch that arrived as one matrix: first fit places RUNS of consecutive rows (a shard takes rows until it is
            # full, then the next one does) — the same placement, one slice per touched shard instead of one array per row
            unique_ids = list(unique_ids)
            at, total = 0, len(unique_ids)
            while at < total:
                shard_id = self._get_available_shard_id()
                if shard_id not in self.box_item_map:
                    self.box_item_map[shard_id] = []
                    self._shard_order.append(shard_id)
                take = min(max(1, self.shard_size - len(self.box_item_map[shard_id])), total - at)
                run = unique_ids[at:at + take]
                self.box_item_map[shard_id].extend(run)
                self.inverse_box_item_map.update(dict.fromkeys(run, shard_id))
                self._persist_to_shard_multiple(shard_id, run, vectors[at:at + take], metadata_dicts[at:at + take])
                at += take
            return
        placed = defaultdict(lambda: ([], [], []))
        for uid, vec, meta in zip(unique_ids, vectors, metadata_dicts):
            shard_id = self._get_available_shard_id()
            if shard_id not in self.box_item_map:
                self.box_item_map[shard_id] = []
                self._shard_order.append(shard_id)
            self.box_item_map[shard_id].append(uid)
            self.inverse_box_item_map[uid] = shard_id
            for column, item in zip(placed[shard_id], (uid, vec, meta)):
                column.append(item)
        for shard_id, (uids, vecs, metas) in placed.items():
            self._persist_to_shard_multiple(shard_id, uids, vecs, metas)

    def store_embedding(self, unique_id, embedding, metadata_dict={}):
        with self.lock:
            if unique_id in self._ids:
                raise ValueError("Unique ID already exists.")
            vector = self._convert_ndarray_float32(embedding)
            self._admit([unique_id], [vector], [metadata_dict])
            self._assign_to_shards([unique_id], [vector], [metadata_dict])

    def _persist_to_shard(self, shard_id, unique_id, embedding, metadata_dict):
        self._persist_to_shard_multiple(shard_id, [unique_id], [embedding], [metadata_dict])

    def _persist_to_shard_multiple(self, shard_id, unique_ids, embeddings, metadata_dicts):
        shard = self._read_shard(shard_id)
        if isinstance(embeddings, np.ndarray) and embeddings.ndim == 2:
            shard['embeddings'] = np.vstack([shard['embeddings'], embeddings])
        else:
            shard['embeddings'] = np.vstack([shard['embeddings']] + [np.atleast_2d(e) for e in embeddings])
        shard['metadata'].extend(metadata_dicts)
        shard['unique_ids'].extend(unique_ids)
        for uid, meta in zip(unique_ids, metadata_dicts):
            for key in meta:
                shard['inverted_index'][key].add(uid)
        self._write_shard(shard_id, shard)

    def store_embeddings_batch(self, unique_ids: list, embeddings, metadata_dicts=[]):
        with self.lock:
            if len(unique_ids) != len(embeddings):
                raise ValueError("Number of unique IDs must match number of embeddings.")
            if isinstance(embeddings, np.ndarray) and embeddings.ndim == 2 and len(embeddings) > 1:
                vectors = np.array(embeddings, dtype=np.float32)   # one copy of the whole batch instead of one array per row
            else:
                vectors = self._convert_ndarray_float32_batch(embeddings)
            for uid in unique_ids:
                if uid in self._ids:
                    raise ValueError(f"Unique ID {uid} already exists.")
            missing = len(unique_ids) - len(metadata_dicts)
            if missing > 0:  # pads the CALLER's list in place, like the reference (:260-261)
                metadata_dicts.extend({} for _ in range(missing))
            self._admit(unique_ids, vectors, metadata_dicts)
            self._assign_to_shards(unique_ids, vectors, metadata_dicts)

    # ---- delete (sharded_vector_database.py:180-241) ------------------------------------------------------
    def _remove_embeddings_from_shard(self, shard_id, unique_ids):
        """Rewrite one shard file without the given ids.  The shard's inverted index loses each id under the keys of that
        id's OWN metadata (the reference tests every key of the shard per id, :195-200)."""
        with open(self._shard_path(shard_id), 'rb') as f:
            shard = pickle.load(f)
        doomed = set(unique_ids)
        stays = np.fromiter((uid not in doomed for uid in shard['unique_ids']), dtype=bool, count=len(shard['unique_ids']))
        for uid, meta, stay in zip(shard['unique_ids'], shard['metadata'], stays):
            if stay:
                continue
            for key in meta:
                holders = shard['inverted_index'].get(key)
                if holders is not None:
                    holders.discard(uid)
                    if not holders:
                        del shard['inverted_index'][key]
        shard['embeddings'] = shard['embeddings'][stays]
        shard['metadata'] = [m for m, stay in zip(shard['metadata'], stays) if stay]
        shard['unique_ids'] = [u for u, stay in zip(shard['unique_ids'], stays) if stay]
        with open(self._shard_path(shard_id), 'wb') as f:
            pickle.dump(shard, f)
        self.box_item_map[shard_id] = shard['unique_ids']
        for uid in doomed:
            del self.inverse_box_item_map[uid]

    def delete_embeddings_batch(self, unique_ids):
        with self.lock:
            if not isinstance(unique_ids, list):
                unique_ids = [unique_ids]
            if not unique_ids:
                raise ValueError("No unique IDs provided.")
            if any(uid not in self._ids for uid in unique_ids):
                raise ValueError("One or more unique IDs do not exist.")
            unique_ids = [uid for uid in unique_ids if uid is not None]

            by_shard = defaultdict(list)
            for uid in unique_ids:
                by_shard[self.inverse_box_item_map[uid]].append(uid)
            for shard_id, shard_ids in by_shard.items():
                self._remove_embeddings_from_shard(shard_id, shard_ids)
            self._first_open = 0   # room has appeared in shards the cursor had passed

            # stacked rows, ids, metadata, inverted index, device matrix: only what belongs to the doomed ids moves
            self._expel(unique_ids)

    # ---- search (sharded_vector_database.py:598-662) --------------------------------------------------------
    def find_most_similar(self, embedding, metadata_filter=None, exclude_filter=None, or_filters=None, k=5,
                          autocut=False):
        uids = self._ids.uids
        hits = []
        for row, score in self._nearest_rows(embedding, metadata_filter, exclude_filter, or_filters, k):
            try:  # a row a concurrent delete has just renumbered away is skipped, as in VectorDatabase.find_most_similar
                hits.append((uids[row], score, self.metadata[row]))
            except (KeyError, IndexError):
                pass
        return self._package(hits, autocut)
