# Round-6 evidence, collected on the GPU box into gpurun_out/r06/ (copied into profiles/ afterwards).
# usage: MVDB_GIT_HEAD=$(git rev-parse --short HEAD) bash benchmarks/collect_profiles_r06.sh [quick]
TAG=r06
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err
for nq in 32 128 256; do
  python3 $R/bench.py --nq $nq --steps 60 --warmup 10 --no-cpu-baseline --no-encoder > $OUT/${TAG}_bench_nq$nq.json 2>> $OUT/bench.err
done
python3 $R/bench.py --nq 256 --dim 384 --steps 60 --warmup 10 --no-cpu-baseline --no-encoder > $OUT/${TAG}_bench_nq256_d384.json 2>> $OUT/bench.err
python3 $R/bench.py --nq 128 --dim 128 --steps 60 --warmup 10 --no-cpu-baseline --no-encoder > $OUT/${TAG}_bench_nq128_d128.json 2>> $OUT/bench.err
python3 $R/benchmarks/refusal_probe2.py > $OUT/${TAG}_refusal_probe.txt 2>&1
python3 $R/bench.py --rows 1000000 --steps 500 --warmup 50 --no-cpu-baseline --no-encoder > $OUT/${TAG}_config2_1M.json 2>> $OUT/bench.err
# the headline kernel: kernel trace + separate PMC passes (FETCH_SIZE / WRITE_SIZE), as the guide prescribes
nq=1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$nq -- python3 $R/bench.py --nq $nq --steps 200 --warmup 10 --no-cpu-baseline --no-encoder > $OUT/${TAG}_final_nq${nq}_bench_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/fe_$nq -- python3 $R/bench.py --nq $nq --steps 20 --warmup 5 --no-cpu-baseline --no-encoder > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/wr_$nq -- python3 $R/bench.py --nq $nq --steps 20 --warmup 5 --no-cpu-baseline --no-encoder > /dev/null 2>&1
(cd $R/profiles && python3 summarize_pmc.py ${TAG}_final_nq$nq /tmp/tr_$nq /tmp/fe_$nq /tmp/wr_$nq $MVDB_GIT_HEAD) > $OUT/summarize_nq$nq.log 2>&1
mv $R/profiles/${TAG}_final_nq${nq}_kernel_stats.csv $R/profiles/${TAG}_final_nq${nq}_pmc_summary.json $OUT/ 2>/dev/null
# ONE sentence per call: walker and (129+ tokens) the per-op chain, e5-small and large shapes
python3 $R/benchmarks/long_sentence_probe.py --variants default --lengths 8,16,32,64,96,128,192,256,384,512 > $OUT/${TAG}_encoder_single_sentence.jsonl 2>> $OUT/bench.err
python3 $R/benchmarks/long_sentence_probe.py --variants default --large --lengths 8,16,32,64,96,128,192,256,384,512 > $OUT/${TAG}_encoder_single_sentence_large.jsonl 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/enc_one -- python3 $R/benchmarks/long_sentence_probe.py --variants default --lengths 32,256 --calls 100 > /dev/null 2>&1
cp $(find /tmp/enc_one -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_encoder_single_sentence_kernel_stats.csv
python3 $R/benchmarks/walk_two_process_probe.py > $OUT/${TAG}_walk_two_process.jsonl 2>> $OUT/bench.err
python3 $R/benchmarks/walk_two_process_probe.py --large --n 1000 >> $OUT/${TAG}_walk_two_process.jsonl 2>> $OUT/bench.err
python3 $R/benchmarks/dispatch_census.py > $OUT/${TAG}_dispatch_census.json 2>> $OUT/bench.err
if [ "$1" != "quick" ]; then
python3 $R/benchmarks/bench_dropin.py > $OUT/${TAG}_dropin_1M.json 2>> $OUT/bench.err
python3 $R/benchmarks/bench_variants.py > $OUT/${TAG}_secondary_paths.jsonl 2>> $OUT/bench.err
# fuzzers on the final tree (the dispatcher changed this round: one certified generation)
python3 $R/benchmarks/fuzz_parity.py 601 240 > $OUT/fuzz_a.txt 2>&1
python3 $R/benchmarks/fuzz_parity.py 602 200 split > $OUT/fuzz_b.txt 2>&1
python3 $R/benchmarks/fuzz_parity.py 603 200 masked > $OUT/fuzz_c.txt 2>&1
python3 $R/benchmarks/fuzz_parity.py 604 200 shadow > $OUT/fuzz_d.txt 2>&1
python3 $R/benchmarks/fuzz_mutations.py 605 200 > $OUT/fuzz_e.txt 2>&1
(for f in a b c d e; do echo "== fuzz_$f"; tail -4 $OUT/fuzz_$f.txt; done) > $OUT/${TAG}_fuzz_parity.txt
fi
ls -la $OUT
