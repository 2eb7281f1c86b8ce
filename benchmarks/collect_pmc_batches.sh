# rocprofv3 kernel trace + separate PMC passes of the batch bench commands (128 / 32 queries per call), as collect_profiles_r05.sh does for nq = 1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_batches
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for nq in 128 32; do
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$nq -- python3 $R/bench.py --nq $nq --steps 60 --warmup 10 --no-cpu-baseline --no-encoder > $OUT/r05_final_nq${nq}_bench_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/fe_$nq -- python3 $R/bench.py --nq $nq --steps 20 --warmup 5 --no-cpu-baseline --no-encoder > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/wr_$nq -- python3 $R/bench.py --nq $nq --steps 20 --warmup 5 --no-cpu-baseline --no-encoder > /dev/null 2>&1
(cd $R/profiles && python3 summarize_pmc.py r05_final_nq$nq /tmp/tr_$nq /tmp/fe_$nq /tmp/wr_$nq $MVDB_GIT_HEAD) > $OUT/summarize_nq$nq.log 2>&1
mv $R/profiles/r05_final_nq${nq}_kernel_stats.csv $R/profiles/r05_final_nq${nq}_pmc_summary.json $OUT/ 2>/dev/null
done
ls -la $OUT
