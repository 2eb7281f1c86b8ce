cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03k; mkdir -p $OUT
python3 benchmarks/bench_config5.py 2>/dev/null | cut -c1-400
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2>$OUT/bench.err; python3 -c "
import json; r=json.load(open('$OUT/bench.json')); print(r['value'], r['roofline']['frac']); print(json.dumps(r['config5'])[:900])"
bash benchmarks/r03_run14.sh
