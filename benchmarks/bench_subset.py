#!/usr/bin/env python3
"""Filtered (subset) search on a resident 10M x 512 corpus, DEVICE side only: the row list is already on the device
(mvdb_index_search_subset_device / mvdb_index_search_masked_device), queries too; time = the enqueue-to-completion of the
search alone.  Reports TB/s of rows touched (rows x d x 4 bytes / time).  One JSON line per case.
MVDB_BENCH_ROWS / MVDB_BENCH_DIM override the corpus shape."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from minivectordb_amd import _native as native  # noqa: E402


def main():
    n, d, k = int(os.environ.get("MVDB_BENCH_ROWS", 10_000_000)), int(os.environ.get("MVDB_BENCH_DIM", 512)), 10
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    idx = native.FlatIndex(d)
    idx.reserve(n)
    idx.add_synthetic(n, 1234, normalize=True)
    q = torch.empty((1, d), dtype=torch.float32, device=dev)
    native.check(native.lib().mvdb_synth_fill_device(q.data_ptr(), 1, d, 5678, 0, 1, 0, stream.cuda_stream))
    D = torch.empty((1, k), dtype=torch.float32, device=dev)
    I = torch.empty((1, k), dtype=torch.int64, device=dev)
    rs = np.random.RandomState(0)

    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    dt = timed(lambda: idx.search_device(q.data_ptr(), 1, k, D.data_ptr(), I.data_ptr(), stream=stream.cuda_stream))
    print(json.dumps({"what": "full scan", "rows": n, "ms": round(dt * 1e3, 3), "TBps": round(n * d * 4 / dt / 1e12, 2)}), flush=True)
    have_mask = hasattr(idx, "search_masked_device")
    for frac, order in ((0.001, "sorted"), (0.01, "sorted"), (0.1, "sorted"), (0.1, "random"), (0.5, "sorted"), (0.9, "sorted"),
                        (0.99, "sorted")):
        m = int(n * frac)
        rows = rs.choice(n, m, replace=False)
        if order == "sorted":
            rows = np.sort(rows)
        rows_t = torch.from_numpy(rows.astype(np.int64)).to(dev)
        dt = timed(lambda: idx.search_subset_device(q.data_ptr(), 1, k, rows_t.data_ptr(), m, D.data_ptr(), I.data_ptr(),
                                                    stream=stream.cuda_stream))
        rec = {"what": f"row list, {frac:g} of the rows ({order})", "rows": m, "list_ms": round(dt * 1e3, 3),
               "list_TBps_rows_touched": round(m * d * 4 / dt / 1e12, 2)}
        if have_mask and order == "sorted":
            bits = np.zeros((n + 63) // 64 * 64, dtype=np.uint8)
            bits[rows] = 1
            mask_t = torch.from_numpy(np.packbits(bits, bitorder="little").view(np.uint64).copy()).to(dev)
            dtm = timed(lambda: idx.search_masked_device(q.data_ptr(), 1, k, mask_t.data_ptr(), D.data_ptr(), I.data_ptr(),
                                                         stream=stream.cuda_stream))
            rec.update({"mask_ms": round(dtm * 1e3, 3), "mask_TBps_rows_touched": round(m * d * 4 / dtm / 1e12, 2),
                        "mask_TBps_corpus": round(n * d * 4 / dtm / 1e12, 2)})
        print(json.dumps(rec), flush=True)
    idx.close()


if __name__ == "__main__":
    main()
