cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03i; mkdir -p $OUT
SECONDS=0; python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$?"
echo "wall seconds: $SECONDS"
python3 - <<'PY'
import json
r=json.load(open("gpurun_out/r03i/bench_default.json"))
print({k: r[k] for k in ("value","ms_per_step","p50_latency_ms")}, r["roofline"]["frac"], r["cpu_baseline"]["value"])
e=r["encoder"]
if "error" in e: print(e)
else:
    for s in e["shapes"]: print(s["S"], s["ragged"], s["fp16x3"]["ms"], s["fp16x3"]["roofline"]["frac"], s["fp32"]["ms"], s["fp32"]["roofline"]["frac"])
    print(e["cpu_baseline"])
    print(r["config5"])
PY
