cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03b; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_encoder_gpu.py tests/test_encoder_tiles.py tests/test_load_path_gpu.py -m gpu -q > $OUT/pytest_encoder.txt 2>&1; echo "pytest rc=$?"
tail -6 $OUT/pytest_encoder.txt
python3 benchmarks/r03_det.py 2>&1 | tail -8
export MVDB_BENCH_COMPUTE=2 MVDB_BENCH_REPS=20
python3 benchmarks/bench_encoder.py > $OUT/enc_new.jsonl 2>> $OUT/bench.err
MVDB_GEMM_LN_FUSED=0 python3 benchmarks/bench_encoder.py > $OUT/enc_noln.jsonl 2>> $OUT/bench.err
MVDB_GEMM_LN_FUSED=0 MVDB_ATTENTION_IMG=0 python3 benchmarks/bench_encoder.py > $OUT/enc_old.jsonl 2>> $OUT/bench.err
MVDB_GEMM_LN_BM=64 python3 benchmarks/bench_encoder.py > $OUT/enc_bm64.jsonl 2>> $OUT/bench.err
for f in new noln old bm64; do echo "== $f"; cut -c1-110 $OUT/enc_$f.jsonl; done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/enc_s32 -- python3 $R/benchmarks/bench_encoder_s32.py 30 > /dev/null 2>&1
cp $(find /tmp/enc_s32 -name "*kernel_stats.csv" | head -1) $R/$OUT/encoder_s32_kernel_stats.csv
MVDB_S32_S=512 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/enc_s512 -- python3 $R/benchmarks/bench_encoder_s32.py 5 > /dev/null 2>&1
cp $(find /tmp/enc_s512 -name "*kernel_stats.csv" | head -1) $R/$OUT/encoder_s512_kernel_stats.csv
head -6 $R/$OUT/encoder_s32_kernel_stats.csv | cut -c1-150
head -6 $R/$OUT/encoder_s512_kernel_stats.csv | cut -c1-150
