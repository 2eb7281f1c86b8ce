"""Host bookkeeping that replaced the reference's per-write rebuilds (minivectordb_amd/_dbcore.py): `_IdIndex`
(id <-> row with handle arithmetic instead of renumbering loops, vector_database.py:139-152) and `_RowStore` (device
rows + pending host rows instead of np.vstack / np.delete on one host matrix, :72, :126) against naive models."""
import numpy as np
import pytest

from minivectordb_amd._dbcore import _IdIndex, _RowStore
from oracle_backend import OracleIndex


def test_id_index_matches_naive_renumbering():
    rs = np.random.RandomState(0)
    ids, model = _IdIndex(), []
    nxt = 0
    for step in range(6000):
        op = rs.rand()
        if op < 0.55 or not model:
            uid = f"u{nxt}" if nxt % 3 else nxt
            nxt += 1
            ids.append(uid)
            model.append(uid)
        else:
            uid = model[rs.randint(len(model))]
            assert ids.row(uid) == model.index(uid)
            r = ids.pop(uid)
            assert r == model.index(uid)
            model.remove(uid)
            assert uid not in ids
        if step % 500 == 0:
            for uid in (model[0], model[len(model) // 2], model[-1]):
                assert ids.row(uid) == model.index(uid) and uid in ids
    assert ids.uids == model and len(ids) == len(model)
    assert ids.row_dict() == dict(enumerate(model))
    assert ids.inverse_dict() == {u: i for i, u in enumerate(model)}
    for uid in model[::97]:
        assert ids.row(uid) == model.index(uid)
    # compaction keeps everything consistent and appends continue after it
    ids.append("tail")
    assert ids.row("tail") == len(model)


def test_row_store_matches_numpy_model():
    rs = np.random.RandomState(1)
    d = 8
    idx = OracleIndex(d)
    st = _RowStore(d)
    model_raw = np.zeros((0, d), np.float32)     # what the reference's matrix would hold
    synced = 0

    def normed(a):
        out = a.copy()
        nz = np.linalg.norm(out, axis=1) > 0
        out[nz] /= np.linalg.norm(out[nz], axis=1, keepdims=True)
        return out

    for step in range(300):
        op = rs.rand()
        if op < 0.5:
            rows = rs.randn(rs.randint(1, 5), d).astype(np.float32)
            st.append(rows if rows.shape[0] > 1 else rows[0])
            model_raw = np.vstack([model_raw, rows])
        elif op < 0.7:
            st.flush(idx)
            model_raw[synced:] = normed(model_raw[synced:])   # the in-place normalisation of a build
            synced = model_raw.shape[0]
        elif model_raw.shape[0] > 2:
            kill = sorted(set(rs.randint(0, model_raw.shape[0], size=rs.randint(1, 3)).tolist()))
            st.delete(kill, idx)
            synced -= sum(1 for r in kill if r < synced)
            model_raw = np.delete(model_raw, kill, axis=0)
        assert st.n == model_raw.shape[0] and st.synced == synced == idx.ntotal
        if step % 10 == 0 and st.n:
            np.testing.assert_allclose(st.materialize(idx), model_raw, atol=1e-6)
            r = rs.randint(st.n)
            got = st.row(r, idx)
            np.testing.assert_allclose(got, model_raw[r], atol=1e-6)
            got[:] = 123.0                     # a copy: the store must not change under the caller's edit
            np.testing.assert_allclose(st.row(r, idx), model_raw[r], atol=1e-6)


def test_row_store_uploads_big_blocks_as_they_stand(monkeypatch):
    """flush: runs of small pending blocks are stacked into one add, a block of DIRECT_UPLOAD_BYTES or more goes up alone;
    an add that fails leaves its rows and the later ones pending, the earlier ones synced."""
    d = 8
    monkeypatch.setattr(_RowStore, "DIRECT_UPLOAD_BYTES", 10 * d * 4)   # 10 rows
    rs = np.random.RandomState(2)
    blocks = [rs.randn(m, d).astype(np.float32) for m in (1, 3, 12, 2, 10, 1, 1)]

    class Recorder(OracleIndex):
        def __init__(self, d, fail_at=None):
            super().__init__(d)
            self.sizes, self.fail_at = [], fail_at

        def add(self, x, normalize=False):
            if self.fail_at is not None and len(self.sizes) == self.fail_at:
                raise RuntimeError("device allocation failed")
            self.sizes.append(x.shape[0])
            return super().add(x, normalize=normalize)

    idx, st = Recorder(d), _RowStore(d)
    for b in blocks:
        st.append(b)
    st.flush(idx)
    assert idx.sizes == [4, 12, 2, 10, 2] and st.synced == 30 == idx.ntotal and st.npending == 0
    want = np.vstack(blocks)
    want /= np.linalg.norm(want, axis=1, keepdims=True)
    np.testing.assert_allclose(st.materialize(idx), want, atol=1e-6)

    idx, st = Recorder(d, fail_at=2), _RowStore(d)
    for b in blocks:
        st.append(b)
    with pytest.raises(RuntimeError):
        st.flush(idx)
    assert idx.sizes == [4, 12] and st.synced == 16 == idx.ntotal and st.npending == 14 and st.n == 30
    idx.fail_at = None
    st.flush(idx)
    assert st.synced == 30 and st.npending == 0
    np.testing.assert_allclose(st.materialize(idx), want, atol=1e-6)


def test_pack_row_mask_takes_lists_sets_and_arrays():
    """`excluded=np.ndarray` used to die on `if excluded:`; an empty array must mean "exclude nothing"."""
    from minivectordb_amd._native import pack_row_mask
    full = pack_row_mask(70)
    assert full.dtype == np.uint64 and full.shape == (2,) and int(full[0]) == 2 ** 64 - 1 and int(full[1]) == 2 ** 6 - 1
    for excluded in ([1, 65], {1, 65}, np.array([1, 65]), np.array([65, 1], dtype=np.int32)):
        m = pack_row_mask(70, excluded=excluded)
        assert int(m[0]) == 2 ** 64 - 1 - 2 and int(m[1]) == 2 ** 6 - 1 - 2
    assert np.array_equal(pack_row_mask(70, excluded=np.array([], dtype=np.int64)), full)
    assert np.array_equal(pack_row_mask(70, excluded=[]), full)
    m = pack_row_mask(70, rows=np.array([0, 64, 69]))
    assert int(m[0]) == 1 and int(m[1]) == 1 + 2 ** 5


def test_fast_single_query_kwarg_reaches_the_index(tmp_path, monkeypatch):
    """VectorDatabase(fast_single_query=True) / ShardedVectorDatabase(...): the device index is created with the
    "shadow_single_query" option set (include/mvdb.h: mvdb_index_set_option); the default leaves it alone."""
    from minivectordb_amd import ShardedVectorDatabase, VectorDatabase, _native
    monkeypatch.setattr(_native, "FlatIndex", OracleIndex)
    x = np.random.RandomState(0).randn(6, 8).astype(np.float32)
    for make in (lambda **kw: VectorDatabase(storage_file=str(tmp_path / f"a{len(kw)}.pkl"), **kw),
                 lambda **kw: ShardedVectorDatabase(storage_dir=str(tmp_path / f"s{len(kw)}"), shard_size=4, **kw)):
        db = make(fast_single_query=True)
        db.store_embeddings_batch(list(range(6)), x)
        db.find_most_similar(x[0], k=2)
        assert ("set_option", "shadow_single_query", 1) in db.index.calls
        db = make()
        db.store_embeddings_batch(list(range(6)), x)
        db.find_most_similar(x[0], k=2)
        assert not any(c[0] == "set_option" for c in db.index.calls if isinstance(c, tuple))


def test_matrix_batches_are_stored_exactly_like_row_lists(tmp_path, monkeypatch):
    """store_embeddings_batch fast paths (a batch that arrives as ONE float32 matrix: no per-row arrays, shard placement by runs):
    the shard files are byte-identical to those of the per-row path, the bookkeeping and the searches equal — both classes,
    batches that straddle shard boundaries, a partly filled last shard, a single-row batch."""
    import filecmp
    import os
    from minivectordb_amd import ShardedVectorDatabase, VectorDatabase, _native
    monkeypatch.setattr(_native, "FlatIndex", OracleIndex)
    rs = np.random.RandomState(5)
    n, d = 1000, 12
    x = rs.randn(n, d).astype(np.float32)
    meta = [{"bucket": i % 7, "i": i} if i % 3 else {} for i in range(n)]
    cuts = [0, 1, 130, 131, 640, 1000]
    dbs = []
    for as_matrix in (True, False):
        sdir = str(tmp_path / f"s{int(as_matrix)}")
        sh = ShardedVectorDatabase(storage_dir=sdir, shard_size=64)
        fl = VectorDatabase(storage_file=str(tmp_path / f"f{int(as_matrix)}.pkl"))
        for a, b in zip(cuts[:-1], cuts[1:]):
            block = x[a:b] if as_matrix else [row for row in x[a:b]]
            sh.store_embeddings_batch([f"u{i}" for i in range(a, b)], block, list(meta[a:b]))
            fl.store_embeddings_batch([f"u{i}" for i in range(a, b)], block, list(meta[a:b]))
        dbs.append((sh, fl, sdir))
    (sh1, fl1, d1), (sh0, fl0, d0) = dbs
    names = sorted(os.listdir(d1))
    assert names == sorted(os.listdir(d0)) and len(names) == 16
    assert all(filecmp.cmp(os.path.join(d1, f), os.path.join(d0, f), shallow=False) for f in names)
    assert sh1.box_item_map == sh0.box_item_map and sh1.inverse_box_item_map == sh0.inverse_box_item_map
    assert sh1.unique_ids == sh0.unique_ids and sh1.metadata == sh0.metadata
    assert dict(sh1.inverted_index) == dict(sh0.inverted_index) and dict(fl1.inverted_index) == dict(fl0.inverted_index)
    assert fl1.inverse_id_map == fl0.inverse_id_map and fl1.metadata == fl0.metadata
    assert np.array_equal(fl1.embeddings, fl0.embeddings) and np.array_equal(sh1.embeddings, sh0.embeddings)
    for a, b in ((sh1, sh0), (fl1, fl0)):
        r1 = a.find_most_similar(x[77], k=5, metadata_filter={"bucket": 0})
        r0 = b.find_most_similar(x[77], k=5, metadata_filter={"bucket": 0})
        assert list(r1[0]) == list(r0[0]) and list(r1[2]) == list(r0[2])
