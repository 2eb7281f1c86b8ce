#!/usr/bin/env python3
"""The rescue tier under a switch (MVDB_DISABLE_TILE_SKIP=1: the rescue launches scan the whole shadow; round 6's launch-shape A/B
used MVDB_RESCUE_FORM): clustered corpus (PROBE_FAMILY=0: zero-mean, 1: all-positive), 256 queries per call; wall time per call,
the launches' own durations from the library's profiling hooks, tiles listed.  usage: rescue_form_probe.py [rows] [dim] [queries per call]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minivectordb_amd import _native as native
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
k = 10
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 256
fam = int(os.environ.get("PROBE_FAMILY", "2")) << 56
idx = native.FlatIndex(d)
idx.reserve(n)
idx.add_synthetic(n, 1234 | fam, normalize=True)
stream = torch.cuda.current_stream().cuda_stream
q = torch.empty((nq, d), dtype=torch.float32, device=dev)
native.check(native.lib().mvdb_synth_fill_device(q.data_ptr(), nq, d, 5678 | fam, 0, 1, 0, stream))
D = torch.empty((nq, k), dtype=torch.float32, device=dev)
I = torch.empty((nq, k), dtype=torch.int64, device=dev)
for _ in range(5):
    idx.search_device(q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), stream=stream)
torch.cuda.synchronize()
reps = 30
t0 = time.perf_counter()
for _ in range(reps):
    idx.search_device(q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), stream=stream)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / reps * 1e3
native.prof_enable(True)
for name in ("ip_scan_rescue", "ip_scan_half", "ip_scan_half_seed", "ip_scan_rerun"):
    native.prof_read(name)
for _ in range(10):
    idx.search_device(q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), stream=stream)
torch.cuda.synchronize()
out = {"rows": n, "d": d, "nq": nq, "tile_skip": os.environ.get("MVDB_DISABLE_TILE_SKIP", "0") != "1", "family": os.environ.get("PROBE_FAMILY", "2"),
       "call_ms": round(wall, 3), "checksum": int(I.sum().item()), "dsum": float(D.double().sum().item())}
listed, total = native.rescue_tile_stats()
out["rescue_tiles_listed_of_total"] = [listed, total, round(listed / max(total, 1), 4)]
for name in ("ip_scan_rescue", "ip_scan_half", "ip_scan_half_seed", "ip_scan_rerun"):
    l, ms = native.prof_read(name)
    out[name] = [l, round(ms / max(l, 1), 4)]
native.prof_enable(False)
print(json.dumps(out), flush=True)
