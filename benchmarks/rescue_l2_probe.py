#!/usr/bin/env python3
"""256 queries per call on the clustered corpus (every certificate refused at 1M rows), inner product against L2: wall time per
call and device time per tier, rescue pass on / off.  usage: rescue_l2_probe.py [rows] [d]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minivectordb_amd import _native as native
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
k, nq, fam = 10, 256, 2 << 56
dev = torch.device("cuda", 0)
FAM = ("ip_scan", "ip_scan_half", "ip_scan_half_seed", "ip_scan_rescue", "ip_scan_rerun")
for metric, name in ((native.METRIC_IP, "ip"), (native.METRIC_L2, "l2")):
    for rescue in (1, 0):
        os.environ["MVDB_DISABLE_RESCUE"] = "0" if rescue else "1"
        idx = native.FlatIndex(d, metric=metric)
        idx.reserve(n)
        idx.add_synthetic(n, 1234 | fam, normalize=True)
        stream = torch.cuda.current_stream().cuda_stream
        q = torch.empty((nq, d), dtype=torch.float32, device=dev)
        native.check(native.lib().mvdb_synth_fill_device(q.data_ptr(), nq, d, 5678 | fam, 0, 1, 0, stream))
        D = torch.empty((nq, k), dtype=torch.float32, device=dev)
        I = torch.empty((nq, k), dtype=torch.int64, device=dev)
        run = lambda: idx.search_device(q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), stream=stream)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        native.prof_enable(True)
        for f in FAM: native.prof_read(f)
        r0 = native.split_rerun_count()
        run(); torch.cuda.synchronize()
        rec = {f: [native.prof_read(f)[0], round(native.prof_read(f)[1], 3)] for f in ()}
        tiers = {}
        for f in FAM:
            c, ms = native.prof_read(f)
            if c: tiers[f] = [c, round(ms, 3)]
        native.prof_enable(False)
        print(json.dumps({"rows": n, "d": d, "metric": name, "rescue_pass": bool(rescue), "ms_per_256_query_call": round(min(ts) * 1e3, 3),
                          "refused_chunks": native.split_rerun_count() - r0, "device_ms [launches, ms]": tiers}), flush=True)
        idx.close()
os.environ.pop("MVDB_DISABLE_RESCUE", None)
