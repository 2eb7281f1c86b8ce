"""Average launch time of the split-precision batch kernel (and whichever other scan kernels ran) at one shape.
Usage: python benchmarks/split_probe.py [rows] [dim] [nq] [k] [steps]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minivectordb_amd import _native as native

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 128
k = int(sys.argv[4]) if len(sys.argv) > 4 else 10
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
idx = native.FlatIndex(d)
idx.add_synthetic(n, 1234, normalize=True)
q = torch.empty((nq, d), dtype=torch.float32, device="cuda")
native.check(native.lib().mvdb_synth_fill_device(q.data_ptr(), nq, d, 5678, 0, 1, 0, 0))
D = torch.empty((nq, k), dtype=torch.float32, device="cuda")
I = torch.empty((nq, k), dtype=torch.int64, device="cuda")
st = torch.cuda.Stream()
import time
with torch.cuda.stream(st):
    for _ in range(2):
        idx.search_device(q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), stream=st.cuda_stream)
    torch.cuda.synchronize()
    native.prof_enable(True)
    for name in ("ip_scan", "ip_scan_mfma", "ip_scan_gemm", "ip_scan_split", "ip_scan_split32", "ip_scan_split_seed"):
        native.prof_read(name)
    t0 = time.perf_counter()
    for _ in range(steps):
        idx.search_device(q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), stream=st.cuda_stream)
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) / steps * 1e3
out = {"rows": n, "dim": d, "nq": nq, "k": k, "dbg": os.environ.get("MVDB_SPLIT_DBG", "0")}
for name in ("ip_scan_mfma", "ip_scan_gemm", "ip_scan_split", "ip_scan_split32", "ip_scan_split_seed"):
    c, ms = native.prof_read(name)
    if c:
        out[name] = round(ms / c, 4)
out["call_ms"] = round(wall_ms, 4)
out["reruns"] = native.split_rerun_count()
print(json.dumps(out))
