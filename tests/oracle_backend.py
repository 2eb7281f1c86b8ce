"""Test-only stand-in for minivectordb_amd._native.FlatIndex built on the CPU oracle.

Injected by CPU tests (monkeypatch) so that the HOST logic of the drop-in classes — filters, id
maps, incremental device sync, autocut, pickle formats — can be exercised without a GPU.  The
product never imports this module and has no such fallback.
"""
import numpy as np

from oracle import flat


class _OracleRowSet:
    def __init__(self, rows, n, gen):
        self.rows, self.n, self.gen = rows, n, gen

    def __len__(self):
        return len(self.rows)

    def close(self):
        pass


class OracleIndex:
    def __init__(self, d, metric=0, device=0):
        self.d, self.metric, self.device = int(d), metric, device
        self.x = np.zeros((0, self.d), dtype=np.float32)
        self.calls = []

    @property
    def ntotal(self):
        return self.x.shape[0]

    def close(self):
        pass

    def set_option(self, name, value):
        self.calls.append(("set_option", name, int(value)))

    def reset(self):
        self.x = np.zeros((0, self.d), dtype=np.float32)

    def reserve(self, n):
        pass

    def add(self, x, normalize=False):
        x = np.ascontiguousarray(x, dtype=np.float32).copy()
        assert x.ndim == 2 and x.shape[1] == self.d
        if normalize:
            flat.normalize_l2(x)
        self.x = np.ascontiguousarray(np.vstack([self.x, x]))
        self.calls.append(("add", x.shape[0]))

    def get_rows(self, row0, n, out=None):
        assert 0 <= row0 and row0 + n <= self.x.shape[0]
        if out is None:
            return self.x[row0:row0 + n].copy()
        out[...] = self.x[row0:row0 + n]
        return out

    def remove_rows(self, rows):
        rows = np.asarray(rows, dtype=np.int64)
        assert len(set(rows.tolist())) == len(rows)
        self.x = np.ascontiguousarray(np.delete(self.x, rows, 0))
        self.renumbered = getattr(self, "renumbered", 0) + 1
        self.calls.append(("remove", len(rows)))

    def search(self, q, k, normalize_q=False):
        self.calls.append(("search", k))
        q = np.atleast_2d(np.asarray(q, dtype=np.float32))
        x = self.x
        if x.shape[0] == 0:
            return (np.full((q.shape[0], k), -3.4028234663852886e38, np.float32),
                    np.full((q.shape[0], k), -1, np.int64))
        return flat.flat_search(x, q, k, metric=self.metric, normalize_q=normalize_q)

    def rowset(self, rows, excluded=False):
        """Stand-in of _native.RowSet: the rows in search order (an excluded set: every other row, ascending)."""
        rows = np.asarray(rows, dtype=np.int64)
        if excluded:
            keep = np.ones(self.x.shape[0], dtype=bool)
            keep[rows] = False
            rows = np.flatnonzero(keep).astype(np.int64)
        self.calls.append(("rowset", len(rows)))
        return _OracleRowSet(rows, self.x.shape[0], getattr(self, "renumbered", 0))

    def search_rowset(self, q, k, rowset, normalize_q=False):
        if rowset.n > self.x.shape[0] or rowset.gen != getattr(self, "renumbered", 0):
            raise ValueError("the row set was built for another state of the index")  # what mvdb_index_search_rowset reports
        D, P = self.search_subset(q, k, rowset.rows, normalize_q=normalize_q)
        return D, np.where(P >= 0, rowset.rows[np.maximum(P, 0)], -1)

    def search_subset(self, q, k, rows, normalize_q=False):
        self.calls.append(("subset", k, len(rows)))
        q = np.atleast_2d(np.asarray(q, dtype=np.float32))
        x = self.x  # one snapshot: a concurrent remove_rows replaces self.x
        rows = np.asarray(rows, dtype=np.int64)
        if rows.size and (rows.min() < 0 or rows.max() >= x.shape[0]):
            raise ValueError("subset row out of range")  # what mvdb_index_search_subset reports
        return flat.flat_search(x, q, k, metric=self.metric, normalize_q=normalize_q, rows=rows)
