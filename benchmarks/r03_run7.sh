cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03f; mkdir -p $OUT
export MVDB_BENCH_COMPUTE=2 MVDB_BENCH_REPS=30 MVDB_BENCH_S=32,64
python3 benchmarks/bench_encoder.py > $OUT/enc_new.jsonl 2>> $OUT/bench.err
MVDB_GEMM_X3_BIG=0 python3 benchmarks/bench_encoder.py > $OUT/enc_nobig.jsonl 2>> $OUT/bench.err
MVDB_GEMM_X3_BIG=0 MVDB_GEMM_X3_BM128N192=1 python3 benchmarks/bench_encoder.py > $OUT/enc_n192.jsonl 2>> $OUT/bench.err
MVDB_GEMM_X3_BIG=1 python3 benchmarks/bench_encoder.py > $OUT/enc_big1.jsonl 2>> $OUT/bench.err
python3 - <<'PY'
import json
rows={}
for f in ("new","nobig","n192","big1"):
    for l in open(f"gpurun_out/r03f/enc_{f}.jsonl"):
        r=json.loads(l); rows.setdefault((r["S"],r["ragged"]),{})[f]=r["ms"]
for k in sorted(rows): print(k, rows[k])
PY
cd /tmp && export TMPDIR=/tmp
MVDB_GEMM_X3_BIG=0 MVDB_GEMM_X3_BM128N192=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/e1 -- python3 $GRAFT_REPO_ROOT/benchmarks/bench_encoder_s32.py 30 > /dev/null 2>&1
grep -E "gemm_x3" $(find /tmp/e1 -name "*kernel_stats.csv" | head -1) | sed 's/_ZN12_GLOBAL__N_1//; s/EEEvPK[^"]*"//' | cut -d, -f1,2,4 | head -5
