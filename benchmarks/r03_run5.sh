cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03d; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_encoder_gpu.py tests/test_encoder_tiles.py -m gpu -q -x > $OUT/pytest_encoder.txt 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_encoder.txt
export MVDB_BENCH_COMPUTE=2 MVDB_BENCH_REPS=20 MVDB_BENCH_S=32,64,128,256,512
python3 benchmarks/bench_encoder.py > $OUT/enc_new.jsonl 2>> $OUT/bench.err
MVDB_GEMM_LN_FUSED=0 python3 benchmarks/bench_encoder.py > $OUT/enc_noln.jsonl 2>> $OUT/bench.err
MVDB_GEMM_LN_BM=32 python3 benchmarks/bench_encoder.py > $OUT/enc_bm32.jsonl 2>> $OUT/bench.err
MVDB_GEMM_LN_BM=64 python3 benchmarks/bench_encoder.py > $OUT/enc_bm64.jsonl 2>> $OUT/bench.err
MVDB_GEMM_LN_BM=128 python3 benchmarks/bench_encoder.py > $OUT/enc_bm128.jsonl 2>> $OUT/bench.err
python3 - <<'PY'
import json
rows={}
for f in ("new","noln","bm32","bm64","bm128"):
    for l in open(f"gpurun_out/r03d/enc_{f}.jsonl"):
        r=json.loads(l); rows.setdefault((r["S"],r["ragged"]),{})[f]=r["ms"]
for k in sorted(rows): print(k, rows[k])
PY
