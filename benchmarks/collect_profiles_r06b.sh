# Round 6, late: what changed after collect_profiles_r06.sh (encoder split-K rule) — the default bench line, the one-sentence
# latencies, the mid-batch probe, the kernel census.  usage: bash benchmarks/collect_profiles_r06b.sh
TAG=r06
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG}b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err
python3 $R/benchmarks/long_sentence_probe.py --variants default --lengths 8,16,32,64,96,128,129,192,256,384,512 > $OUT/${TAG}_encoder_single_sentence.jsonl 2>> $OUT/bench.err
python3 $R/benchmarks/long_sentence_probe.py --variants default --large --lengths 8,16,32,64,65,96,129,192,256,384,512 > $OUT/${TAG}_encoder_single_sentence_large.jsonl 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/enc_one -- python3 $R/benchmarks/long_sentence_probe.py --variants default --lengths 32,256 --calls 100 > /dev/null 2>&1
cp $(find /tmp/enc_one -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_encoder_single_sentence_kernel_stats.csv
(python3 $R/benchmarks/mid_batch_probe.py; python3 $R/benchmarks/mid_batch_probe.py --large) > $OUT/${TAG}_encoder_mid_batches.txt 2>> $OUT/bench.err
python3 $R/benchmarks/kernel_census.py > $OUT/${TAG}_kernel_census.json
ls -la $OUT
