R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/benchmarks/fuzz_parity.py 601 240 > $OUT/fuzz_a.txt 2>&1
python3 $R/benchmarks/fuzz_parity.py 602 200 split > $OUT/fuzz_b.txt 2>&1
python3 $R/benchmarks/fuzz_parity.py 603 200 masked > $OUT/fuzz_c.txt 2>&1
python3 $R/benchmarks/fuzz_parity.py 604 200 shadow > $OUT/fuzz_d.txt 2>&1
python3 $R/benchmarks/fuzz_mutations.py 605 200 > $OUT/fuzz_e.txt 2>&1
(for f in a b c d e; do echo "== fuzz_$f"; tail -2 $OUT/fuzz_$f.txt; done) > $OUT/r05_fuzz_parity.txt
python3 $R/bench.py > $OUT/r05_bench_default.json 2> $OUT/bench_default.err
python3 $R/benchmarks/refusal_probe2.py > $OUT/r05_refusal_probe.txt 2>&1
python3 $R/bench.py --nq 256 --steps 60 --warmup 10 --no-cpu-baseline --no-encoder > $OUT/r05_bench_nq256.json 2>> $OUT/bench.err
python3 $R/bench.py --nq 32 --steps 60 --warmup 10 --no-cpu-baseline --no-encoder > $OUT/r05_bench_nq32.json 2>> $OUT/bench.err
python3 $R/bench.py --rows 1000000 --nq 32 --steps 200 --warmup 20 --no-cpu-baseline --no-encoder > $OUT/r05_bench_1M_nq32.json 2>> $OUT/bench.err
