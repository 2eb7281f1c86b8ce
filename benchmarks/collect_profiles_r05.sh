# Round-5 evidence, collected on the GPU box into gpurun_out/r05/ (copied into profiles/ afterwards).
# usage: MVDB_GIT_HEAD=$(git rev-parse --short HEAD) bash benchmarks/collect_profiles_r05.sh
TAG=r05
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err
for nq in 32 128 256; do
  python3 $R/bench.py --nq $nq --steps 60 --warmup 10 --no-cpu-baseline --no-encoder > $OUT/${TAG}_bench_nq$nq.json 2>> $OUT/bench.err
done
python3 $R/bench.py --nq 256 --dim 384 --steps 60 --warmup 10 --no-cpu-baseline --no-encoder > $OUT/${TAG}_bench_nq256_d384.json 2>> $OUT/bench.err
python3 $R/bench.py --nq 128 --k 32 --steps 60 --warmup 10 --no-cpu-baseline --no-encoder > $OUT/${TAG}_bench_nq128_k32.json 2>> $OUT/bench.err
python3 $R/benchmarks/refusal_probe2.py > $OUT/${TAG}_refusal_probe.txt 2>&1
python3 $R/benchmarks/ties_probe.py > $OUT/${TAG}_ties_probe.txt 2>&1
python3 $R/bench.py --rows 1000000 --steps 500 --warmup 50 --no-cpu-baseline --no-encoder > $OUT/${TAG}_config2_1M.json 2>> $OUT/bench.err
# the headline kernel: kernel trace + separate PMC passes (FETCH_SIZE / WRITE_SIZE), as the guide prescribes
nq=1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$nq -- python3 $R/bench.py --nq $nq --steps 200 --warmup 10 --no-cpu-baseline --no-encoder > $OUT/${TAG}_final_nq${nq}_bench_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/fe_$nq -- python3 $R/bench.py --nq $nq --steps 20 --warmup 5 --no-cpu-baseline --no-encoder > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/wr_$nq -- python3 $R/bench.py --nq $nq --steps 20 --warmup 5 --no-cpu-baseline --no-encoder > /dev/null 2>&1
(cd $R/profiles && python3 summarize_pmc.py ${TAG}_final_nq$nq /tmp/tr_$nq /tmp/fe_$nq /tmp/wr_$nq $MVDB_GIT_HEAD) > $OUT/summarize_nq$nq.log 2>&1
mv $R/profiles/${TAG}_final_nq${nq}_kernel_stats.csv $R/profiles/${TAG}_final_nq${nq}_pmc_summary.json $OUT/ 2>/dev/null
# ONE sentence per call: the layer-walking launch under the kernel trace (one encoder kernel per forward), e5-small and large shapes
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/enc_one -- python3 $R/benchmarks/bench_encoder_single.py > $OUT/${TAG}_encoder_single_sentence_under_rocprof.json 2>/dev/null
cp $(find /tmp/enc_one -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_encoder_single_sentence_kernel_stats.csv
python3 $R/benchmarks/bench_encoder_single.py > $OUT/${TAG}_encoder_single_sentence.json 2>> $OUT/bench.err
python3 $R/benchmarks/bench_encoder_single.py --large > $OUT/${TAG}_encoder_single_sentence_large.json 2>> $OUT/bench.err
MVDB_ENCODER_WALK=0 python3 $R/benchmarks/bench_encoder_single.py > $OUT/${TAG}_encoder_single_sentence_per_op_chain.json 2>> $OUT/bench.err
# the 256 x 32-token batch (unchanged kernels): launches per layer for the record
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/enc_s32 -- python3 $R/benchmarks/bench_encoder_s32.py 30 > /dev/null 2>&1
cp $(find /tmp/enc_s32 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_encoder_s32_kernel_stats.csv
python3 $R/benchmarks/bench_dropin.py > $OUT/${TAG}_dropin_1M.json 2>> $OUT/bench.err
python3 $R/benchmarks/bench_variants.py > $OUT/${TAG}_secondary_paths.jsonl 2>> $OUT/bench.err
# fuzzers on the final tree (the scan shapes and the dispatcher changed this round)
python3 $R/benchmarks/fuzz_parity.py 501 240 > $OUT/fuzz_a.txt 2>&1
python3 $R/benchmarks/fuzz_parity.py 502 200 split > $OUT/fuzz_b.txt 2>&1
python3 $R/benchmarks/fuzz_parity.py 503 200 masked > $OUT/fuzz_c.txt 2>&1
python3 $R/benchmarks/fuzz_parity.py 504 200 shadow > $OUT/fuzz_d.txt 2>&1
python3 $R/benchmarks/fuzz_mutations.py 505 200 > $OUT/fuzz_e.txt 2>&1
(for f in a b c d e; do echo "== fuzz_$f"; tail -4 $OUT/fuzz_$f.txt; done) > $OUT/${TAG}_fuzz_parity.txt
# ablation build: the walker's phase timeline
ABL=$R/minivectordb_amd/lib/libmvdb_ablate.so
if [ -f $ABL ]; then
  (for S in 8 32 64 128; do MVDB_LIBMVDB=$ABL python3 $R/benchmarks/walk_trace.py $S; done; for S in 16 64; do MVDB_LIBMVDB=$ABL python3 $R/benchmarks/walk_trace.py $S large; done) > $OUT/${TAG}_walk_phase_timeline.jsonl 2>> $OUT/bench.err
fi
ls -la $OUT
