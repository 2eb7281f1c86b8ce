#!/usr/bin/env python3
"""Where does the fp16-shadow nomination pass (128-query form, padded) beat the exact passes for SMALL batches?
Device-side time per call (queries resident), rows x queries grid; MVDB_SPLIT_SCAN_MIN_NQ=2 forces the certified pass."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from minivectordb_amd import _native as native

dev = torch.device("cuda", 0)
d, k = int(os.environ.get("PROBE_D", "512")), 10
st = torch.cuda.Stream(dev)
torch.cuda.set_stream(st)
for n in (100_000, 1_000_000, 10_000_000):
    idx = native.FlatIndex(d)
    idx.reserve(n)
    idx.add_synthetic(n, 1234, normalize=True)
    q = torch.empty((256, d), dtype=torch.float32, device=dev)
    native.check(native.lib().mvdb_synth_fill_device(q.data_ptr(), 256, d, 5678, 0, 1, 0, st.cuda_stream))
    D = torch.empty((256, k), dtype=torch.float32, device=dev)
    I = torch.empty((256, k), dtype=torch.int64, device=dev)
    for nq in (1, 2, 4, 8, 13, 16, 24, 32, 48):
        row = {"rows": n, "nq": nq}
        for label, env in (("default", None), ("certified_from_2", "2")):
            if env is None:
                os.environ.pop("MVDB_SPLIT_SCAN_MIN_NQ", None)
            else:
                os.environ["MVDB_SPLIT_SCAN_MIN_NQ"] = env
            idx.reload_env()
            for _ in range(5):
                idx.search_device(q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), stream=st.cuda_stream)
            torch.cuda.synchronize()
            reps = 30
            t0 = time.perf_counter()
            for _ in range(reps):
                idx.search_device(q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), stream=st.cuda_stream)
            torch.cuda.synchronize()
            row[label + "_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 4)
        print(json.dumps(row), flush=True)
    idx.close()
