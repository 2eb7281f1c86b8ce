#!/usr/bin/env python3
"""U x blocks/CU sweep of the GEMV scan for other widths (MVDB_SCAN_U hook), ~5 GB corpora."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in (384, 256, 1024, 64):
    rows = 5_120_000_000 // (d * 4)
    for u in (1, 2, 4, 8):
        for b in (2, 3, 4):
            env = dict(os.environ, MVDB_SCAN_U=str(u), MVDB_SCAN_BLOCKS_PER_CU=str(b))
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "100", "--warmup", "10",
                                  "--no-cpu-baseline", "--dim", str(d), "--rows", str(rows)], env=env,
                                 capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if line:
                j = json.loads(line[-1])
                print(f"d={d} U={u} blocks/CU={b} GB/s={j['roofline']['achieved']:.0f} qps={j['value']:.1f}", flush=True)
