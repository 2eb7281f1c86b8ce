cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03c; mkdir -p $OUT
for d in 0 1 2 3 4; do
  for S in 32 512; do
    n=30; [ $S = 512 ] && n=5
    MVDB_GEMM_X3_DBG=$d MVDB_S32_S=$S rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_${d}_$S -- python3 $R/benchmarks/bench_encoder_s32.py $n > /dev/null 2>&1
    f=$(find /tmp/abl_${d}_$S -name "*kernel_stats.csv" | head -1)
    echo "== dbg=$d S=$S"; grep -E "gemm_x3" $f | sed 's/_ZN12_GLOBAL__N_1//; s/EEEvPK[^"]*"//' | cut -d, -f1,2,4 | head -4
  done
done
