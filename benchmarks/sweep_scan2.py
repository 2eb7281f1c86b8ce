#!/usr/bin/env python3
"""Finer sweep around the best point of sweep_scan.py (interleaved repeats to separate noise)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
points = [(v, b) for v in (101, 201, 301, 401, 601, 801) for b in (1, 2, 3)]
res = {}
for rep in range(2):
    for variant, bpc in points:
        env = dict(os.environ, MVDB_SCAN_VARIANT=str(variant), MVDB_SCAN_BLOCKS_PER_CU=str(bpc))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "150", "--warmup", "20",
                              "--no-cpu-baseline"], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(variant, bpc, "FAILED", out.stderr[-300:])
            continue
        j = json.loads(line[-1])
        res.setdefault((variant, bpc), []).append((j["roofline"]["achieved"], j["value"], j["p50_latency_ms"]))
for (variant, bpc), v in sorted(res.items()):
    print(f"variant={variant:3d} blocks/CU={bpc} GB/s={[x[0] for x in v]} qps={[x[1] for x in v]} p50={[x[2] for x in v]}")
