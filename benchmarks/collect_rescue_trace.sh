# kernel trace + separate PMC passes of a refused 256-query call on the clustered corpus (the rescue launch's HBM bytes)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/rescue_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rt_tr -- python3 $R/benchmarks/rescue_trace.py 6 > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/rt_fe -- python3 $R/benchmarks/rescue_trace.py 3 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/rt_wr -- python3 $R/benchmarks/rescue_trace.py 3 > $OUT/write.log 2>&1
(cd $R/profiles && python3 summarize_pmc.py r05_rescue_clustered /tmp/rt_tr /tmp/rt_fe /tmp/rt_wr $MVDB_GIT_HEAD) > $OUT/summarize.log 2>&1
mv $R/profiles/r05_rescue_clustered_kernel_stats.csv $R/profiles/r05_rescue_clustered_pmc_summary.json $OUT/ 2>/dev/null
ls -la $OUT
