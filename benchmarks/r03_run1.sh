# round-3 run 1: lane-op probe, encoder parity with the fused kernels, A/B of the new encoder paths, kernel traces
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03a
mkdir -p $OUT
cd $R
./benchmarks/micro/probe_lane_ops > $OUT/probe_lane_ops.txt 2>&1; echo "probe rc=$?" | tee -a $OUT/log.txt
timeout 900 python3 -m pytest tests/test_encoder_gpu.py tests/test_encoder_tiles.py -m gpu -x -q > $OUT/pytest_encoder.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/log.txt
tail -5 $OUT/pytest_encoder.txt
export MVDB_BENCH_COMPUTE=2 MVDB_BENCH_REPS=20
python3 benchmarks/bench_encoder.py > $OUT/enc_new.jsonl 2>> $OUT/bench.err
MVDB_GEMM_LN_FUSED=0 python3 benchmarks/bench_encoder.py > $OUT/enc_noln.jsonl 2>> $OUT/bench.err
MVDB_ATTENTION_IMG=0 python3 benchmarks/bench_encoder.py > $OUT/enc_noimg.jsonl 2>> $OUT/bench.err
MVDB_GEMM_LN_FUSED=0 MVDB_ATTENTION_IMG=0 python3 benchmarks/bench_encoder.py > $OUT/enc_old.jsonl 2>> $OUT/bench.err
MVDB_GEMM_LN_BM=64 python3 benchmarks/bench_encoder.py > $OUT/enc_bm64.jsonl 2>> $OUT/bench.err
MVDB_GEMM_LN_BM=32 MVDB_BENCH_S=512 python3 benchmarks/bench_encoder.py > $OUT/enc_bm32_s512.jsonl 2>> $OUT/bench.err
for f in new noln noimg old bm64 bm32_s512; do echo "== $f"; cut -c1-150 $OUT/enc_$f.jsonl; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/enc_s32 -- python3 $R/benchmarks/bench_encoder_s32.py 30 > /dev/null 2>&1
cp $(find /tmp/enc_s32 -name "*kernel_stats.csv" | head -1) $OUT/encoder_s32_kernel_stats.csv
MVDB_S32_S=512 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/enc_s512 -- python3 $R/benchmarks/bench_encoder_s32.py 5 > /dev/null 2>&1
cp $(find /tmp/enc_s512 -name "*kernel_stats.csv" | head -1) $OUT/encoder_s512_kernel_stats.csv
head -8 $OUT/encoder_s32_kernel_stats.csv | cut -c1-200
head -8 $OUT/encoder_s512_kernel_stats.csv | cut -c1-200
