# Round 6, after the tile flags: kernel trace + separate FETCH_SIZE / WRITE_SIZE passes of the certified batch pass at 32 / 128 / 256
# queries per call, and of a refused 256-query call on the clustered corpus (the rescue launch over its tile list).
# usage: MVDB_GIT_HEAD=... bash benchmarks/collect_pmc_batches_r06b.sh
TAG=r06
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG}pmc_b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for nq in 32 128 256; do
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$nq -- python3 $R/bench.py --nq $nq --steps 100 --warmup 10 --no-cpu-baseline --no-encoder > $OUT/${TAG}_final_nq${nq}_bench_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/fe_$nq -- python3 $R/bench.py --nq $nq --steps 20 --warmup 5 --no-cpu-baseline --no-encoder > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/wr_$nq -- python3 $R/bench.py --nq $nq --steps 20 --warmup 5 --no-cpu-baseline --no-encoder > /dev/null 2>&1
(cd $R/profiles && python3 summarize_pmc.py ${TAG}_final_nq$nq /tmp/tr_$nq /tmp/fe_$nq /tmp/wr_$nq $MVDB_GIT_HEAD) > $OUT/summarize_nq$nq.log 2>&1
mv $R/profiles/${TAG}_final_nq${nq}_kernel_stats.csv $R/profiles/${TAG}_final_nq${nq}_pmc_summary.json $OUT/ 2>/dev/null
done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rt_tr -- python3 $R/benchmarks/rescue_trace.py 6 > $OUT/rescue_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/rt_fe -- python3 $R/benchmarks/rescue_trace.py 3 > $OUT/rescue_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/rt_wr -- python3 $R/benchmarks/rescue_trace.py 3 > $OUT/rescue_write.log 2>&1
(cd $R/profiles && python3 summarize_pmc.py ${TAG}_rescue_clustered /tmp/rt_tr /tmp/rt_fe /tmp/rt_wr $MVDB_GIT_HEAD) > $OUT/summarize_rescue.log 2>&1
mv $R/profiles/${TAG}_rescue_clustered_kernel_stats.csv $R/profiles/${TAG}_rescue_clustered_pmc_summary.json $OUT/ 2>/dev/null
ls -la $OUT
