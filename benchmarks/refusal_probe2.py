"""Where a refused 256-query call spends its time on the clustered corpus (1M and 10M rows x 512): wall time per call and
device time per kernel family (the library's own event pairs), 4 calls each."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from minivectordb_amd import _native as native
dev = torch.device("cuda", 0)
d, k = 512, 10
fam = 2 << 56
FAM = ("ip_scan", "ip_scan_scores", "ip_scan_mfma", "ip_scan_mfma_masked", "ip_scan_gemm", "ip_scan_half", "ip_scan_half_seed", "ip_scan_rescue",
       "ip_scan_rerun")
for n in (1_000_000, 10_000_000):
    idx = native.FlatIndex(d)
    idx.reserve(n)
    idx.add_synthetic(n, 1234 | fam, normalize=True)
    stream = torch.cuda.current_stream().cuda_stream
    for nq in (256, 32):
        q = torch.empty((256, d), dtype=torch.float32, device=dev)
        native.check(native.lib().mvdb_synth_fill_device(q.data_ptr(), 256, d, 5678 | fam, 0, 1, 0, stream))
        D = torch.empty((256, k), dtype=torch.float32, device=dev)
        I = torch.empty((256, k), dtype=torch.int64, device=dev)
        def run(a, m):
            idx.search_device(q[a:a+m].data_ptr(), m, k, D[a:a+m].data_ptr(), I[a:a+m].data_ptr(), stream=stream)
        run(0, nq); torch.cuda.synchronize()
        native.prof_enable(True)
        for rep in range(3):
            for f in FAM: native.prof_read(f)
            r0 = native.split_rerun_count()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            run(0, nq)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            rec = {}
            for f in FAM:
                c, ms = native.prof_read(f)
                if c: rec[f] = [c, round(ms, 3)]
            print(json.dumps({"rows": n, "nq": nq, "ms": round(dt*1e3, 3), "refused_chunks": native.split_rerun_count() - r0, "kernels [launches, ms]": rec}), flush=True)
        native.prof_enable(False)
    idx.close() if hasattr(idx, "close") else None
    del idx
