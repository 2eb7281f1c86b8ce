cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03e; mkdir -p $OUT
export MVDB_BENCH_COMPUTE=2 MVDB_BENCH_REPS=20 MVDB_BENCH_S=128,512
python3 benchmarks/bench_encoder.py > $OUT/enc_new.jsonl 2>> $OUT/bench.err
MVDB_GEMM_X3_BIG=0 python3 benchmarks/bench_encoder.py > $OUT/enc_nobig.jsonl 2>> $OUT/bench.err
MVDB_GEMM_X3_BIG=0 MVDB_GEMM_X3_BM128N192=1 python3 benchmarks/bench_encoder.py > $OUT/enc_n192.jsonl 2>> $OUT/bench.err
python3 - <<'PY'
import json
rows={}
for f in ("new","nobig","n192"):
    for l in open(f"gpurun_out/r03e/enc_{f}.jsonl"):
        r=json.loads(l); rows.setdefault((r["S"],r["ragged"]),{})[f]=r["ms"]
for k in sorted(rows): print(k, rows[k])
PY
