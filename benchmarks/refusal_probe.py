import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from minivectordb_amd import _native as native
dev = torch.device("cuda", 0)
n, d, k = 10_000_000, 512, 10
fam = 2 << 56
idx = native.FlatIndex(d)
idx.reserve(n)
idx.add_synthetic(n, 1234 | fam, normalize=True)
nq = 256
q = torch.empty((nq, d), dtype=torch.float32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
native.check(native.lib().mvdb_synth_fill_device(q.data_ptr(), nq, d, 5678 | fam, 0, 1, 0, stream))
D = torch.empty((nq, k), dtype=torch.float32, device=dev)
I = torch.empty((nq, k), dtype=torch.int64, device=dev)
FAM = ("ip_scan", "ip_scan_mfma", "ip_scan_gemm", "ip_scan_half", "ip_scan_half_seed", "ip_scan_rescue", "ip_scan_rerun")
def run(a, m):
    idx.search_device(q[a:a+m].data_ptr(), m, k, D[a:a+m].data_ptr(), I[a:a+m].data_ptr(), stream=stream)
run(0, 32); torch.cuda.synchronize()
native.prof_enable(True)
for a in range(0, 256, 32):
    for f in FAM: native.prof_read(f)
    r0 = native.split_rerun_count()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(a, 32)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    rec = {}
    for f in FAM:
        c, ms = native.prof_read(f)
        if c: rec[f] = [c, round(ms, 3)]
    print(a, "ms", round(dt*1e3, 2), "refused chunks", native.split_rerun_count() - r0, rec)
