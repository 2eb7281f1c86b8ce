"""Timing ablations of flat_scan_split128_kernel (MVDB_SPLIT128_DBG; results of DBG != 0 are invalid):
main-launch ms at 10M x 512, 128 queries.  usage: split128_probe.py [rows] [dim]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = sys.argv[1] if len(sys.argv) > 1 else "10000000"
dim = sys.argv[2] if len(sys.argv) > 2 else "512"
VARIANTS = ((0, "full"), (1024, "DMA issued back to back (not spread between MFMAs)"), (64, "nomination never entered"),
            (576, "never nominate, exchange in registers (no LDS, no barrier)"), (68, "never nominate, no DMA"),
            (580, "never nominate, no DMA, exchange in registers"), (2, "no MFMA"), (4, "no DMA"))
if len(sys.argv) > 3:
    VARIANTS = tuple(v for v in VARIANTS if str(v[0]) in sys.argv[3].split(","))
for dbg, label in VARIANTS:
    env = dict(os.environ, MVDB_SPLIT128_DBG=str(dbg))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--nq", "128", "--steps", "20", "--warmup", "3",
                        "--rows", rows, "--dim", dim, "--no-cpu-baseline"], env=env, capture_output=True, text=True)
    try:
        d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        r = d["roofline"]
        print(json.dumps({"dbg": dbg, "what": label, "main_launches_ms_per_pass": round(r["avg_launch_ms"] * r["launches_per_corpus_pass"], 4),
                          "ms_per_step": d["ms_per_step"], "GBps": r["achieved"]}), flush=True)
    except Exception as e:
        print(dbg, label, "failed", p.stderr[-500:])
