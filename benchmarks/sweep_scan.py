#!/usr/bin/env python3
"""Coarse sweep of the d = 512 scan kernel's tuning hooks (MVDB_SCAN_VARIANT = U*10 + NT,
MVDB_SCAN_BLOCKS_PER_CU), one bench.py process per point.  Prints QPS / GB/s per point."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
points = [(-1, 0)] + [(v, 0) for v in (20, 21, 40, 80, 81)] + [(-1, b) for b in (2, 3, 4, 5, 6)] + [(81, 4), (81, 3), (-1, 0)]
rows = sys.argv[1] if len(sys.argv) > 1 else "10000000"
for variant, bpc in points:
    env = dict(os.environ)
    if variant >= 0:
        env["MVDB_SCAN_VARIANT"] = str(variant)
    if bpc:
        env["MVDB_SCAN_BLOCKS_PER_CU"] = str(bpc)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "100", "--warmup", "20",
                          "--no-cpu-baseline", "--rows", rows], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(variant, bpc, "FAILED", out.stderr[-300:])
        continue
    j = json.loads(line[-1])
    print(f"variant={variant:3d} blocks/CU={bpc} qps={j['value']:8.2f} ms/step={j['ms_per_step']:.4f} "
          f"scan_ms={j['roofline']['avg_launch_ms']:.4f} GB/s={j['roofline']['achieved']:.0f} "
          f"frac={j['roofline']['frac']:.4f} p50={j['p50_latency_ms']:.3f}", flush=True)
