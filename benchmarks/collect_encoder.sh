# Encoder-only subset of collect_profiles.sh (after a change that only touches encoder.hip): same files, same names.
# usage: bash benchmarks/collect_encoder.sh <tag>
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/benchmarks/bench_config5.py > $OUT/${TAG}_config5_end_to_end.jsonl 2>> $OUT/bench.err
python3 $R/benchmarks/bench_encoder.py > $OUT/${TAG}_encoder_bench.jsonl 2>> $OUT/bench.err
MVDB_BENCH_MODEL=e5-large python3 $R/benchmarks/bench_encoder.py > $OUT/${TAG}_encoder_large_bench.jsonl 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/enc_s32 -- python3 $R/benchmarks/bench_encoder_s32.py 30 > /dev/null 2>&1
cp $(find /tmp/enc_s32 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_encoder_s32_kernel_stats.csv
bash $R/benchmarks/prof_encoder_x3.sh $TAG/enc_pmc > /dev/null 2>&1
cp $OUT/enc_pmc/pmc_summary.txt $OUT/${TAG}_encoder_s32_pmc.txt
python3 $R/benchmarks/bench_dropin.py > $OUT/${TAG}_dropin_1M.json 2>> $OUT/bench.err
ls -la $OUT
