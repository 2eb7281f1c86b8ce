# Collects the round's evidence on the GPU box into gpurun_out/<tag>/ (copied into profiles/ afterwards):
# bench lines, rocprofv3 kernel traces and separate PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA busy) per batch size.
# usage: bash benchmarks/collect_profiles.sh <tag>
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err
for nq in 32 128 256; do
  python3 $R/bench.py --nq $nq --steps 60 --warmup 10 --no-cpu-baseline > $OUT/${TAG}_bench_nq$nq.json 2>> $OUT/bench.err
done
python3 $R/bench.py --nq 128 --dim 384 --steps 60 --warmup 10 --no-cpu-baseline > $OUT/${TAG}_bench_nq128_d384.json 2>> $OUT/bench.err
python3 $R/bench.py --nq 256 --dim 384 --steps 60 --warmup 10 --no-cpu-baseline > $OUT/${TAG}_bench_nq256_d384.json 2>> $OUT/bench.err
python3 $R/bench.py --nq 128 --k 32 --steps 60 --warmup 10 --no-cpu-baseline > $OUT/${TAG}_bench_nq128_k32.json 2>> $OUT/bench.err
python3 $R/bench.py --nq 128 --dim 640 --rows 8000000 --steps 60 --warmup 10 --no-cpu-baseline > $OUT/${TAG}_bench_nq128_d640.json 2>> $OUT/bench.err
python3 $R/bench.py --nq 128 --dim 768 --rows 5000000 --steps 60 --warmup 10 --no-cpu-baseline --no-encoder > $OUT/${TAG}_bench_nq128_d768.json 2>> $OUT/bench.err
python3 $R/bench.py --nq 128 --dim 1024 --rows 5000000 --steps 60 --warmup 10 --no-cpu-baseline --no-encoder > $OUT/${TAG}_bench_nq128_d1024.json 2>> $OUT/bench.err
python3 $R/bench.py --nq 8 --steps 100 --warmup 10 --no-cpu-baseline --no-encoder > $OUT/${TAG}_bench_nq8.json 2>> $OUT/bench.err
# the same batches on the fp32 rows (round 3's kernels): A/B on this box
for nq in 8 32 128 256; do
  MVDB_DISABLE_HALF_SHADOW=1 python3 $R/bench.py --nq $nq --steps 60 --warmup 10 --no-cpu-baseline --no-encoder > $OUT/${TAG}_bench_nq${nq}_no_shadow.json 2>> $OUT/bench.err
done
python3 $R/bench.py --rows 1000000 --steps 500 --warmup 50 --no-cpu-baseline --no-encoder > $OUT/${TAG}_config2_1M.json 2>> $OUT/bench.err
for nq in 1 32 128 256; do
  steps=60; [ $nq = 1 ] && steps=200
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$nq -- python3 $R/bench.py --nq $nq --steps $steps --warmup 10 --no-cpu-baseline --no-encoder > $OUT/${TAG}_final_nq${nq}_bench_under_rocprof.json 2>/dev/null
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/fe_$nq -- python3 $R/bench.py --nq $nq --steps 20 --warmup 5 --no-cpu-baseline --no-encoder > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/wr_$nq -- python3 $R/bench.py --nq $nq --steps 20 --warmup 5 --no-cpu-baseline --no-encoder > /dev/null 2>&1
  (cd $R/profiles && python3 summarize_pmc.py ${TAG}_final_nq$nq /tmp/tr_$nq /tmp/fe_$nq /tmp/wr_$nq $MVDB_GIT_HEAD) > $OUT/summarize_nq$nq.log 2>&1
  mv $R/profiles/${TAG}_final_nq${nq}_kernel_stats.csv $R/profiles/${TAG}_final_nq${nq}_pmc_summary.json $OUT/ 2>/dev/null
done
for nq in 128 256; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/mf_$nq -- python3 $R/bench.py --nq $nq --steps 20 --warmup 5 --no-cpu-baseline --no-encoder > /dev/null 2>&1
  python3 $R/profiles/summarize_mfma.py /tmp/mf_$nq $OUT/${TAG}_mfma_util_nq$nq.json > $OUT/mfma_nq$nq.log 2>&1
done
python3 $R/benchmarks/bench_config5.py > $OUT/${TAG}_config5_end_to_end.jsonl 2>> $OUT/bench.err
python3 $R/benchmarks/bench_encoder.py > $OUT/${TAG}_encoder_bench.jsonl 2>> $OUT/bench.err
MVDB_BENCH_MODEL=e5-large python3 $R/benchmarks/bench_encoder.py > $OUT/${TAG}_encoder_large_bench.jsonl 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/enc_s32 -- python3 $R/benchmarks/bench_encoder_s32.py 30 > /dev/null 2>&1
cp $(find /tmp/enc_s32 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_encoder_s32_kernel_stats.csv
MVDB_S32_S=512 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/enc_s512 -- python3 $R/benchmarks/bench_encoder_s32.py 5 > /dev/null 2>&1
cp $(find /tmp/enc_s512 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_encoder_s512_kernel_stats.csv
MVDB_BENCH_COMPUTE=2 MVDB_BENCH_S=32,64,128,256,512 python3 $R/benchmarks/bench_encoder.py > $OUT/${TAG}_encoder_seq_sweep.jsonl 2>> $OUT/bench.err
MVDB_GEMM_LN_FUSED=0 MVDB_ATTENTION_IMG=0 MVDB_GEMM_X3_PERSIST=0 MVDB_GEMM_X3_SPREAD_SMALL=0 MVDB_GEMM_LN_SPREAD=0 MVDB_BENCH_COMPUTE=2 MVDB_BENCH_S=32,512 python3 $R/benchmarks/bench_encoder.py > $OUT/${TAG}_encoder_bench_r02_paths.jsonl 2>> $OUT/bench.err
bash $R/benchmarks/prof_encoder_x3.sh $TAG/enc_pmc > /dev/null 2>&1
cp $OUT/enc_pmc/pmc_summary.txt $OUT/${TAG}_encoder_s32_pmc.txt
python3 $R/benchmarks/scale_check.py > $OUT/${TAG}_scale_check_80M.json 2>> $OUT/bench.err
python3 $R/benchmarks/bench_dropin.py > $OUT/${TAG}_dropin_1M.json 2>> $OUT/bench.err
python3 $R/benchmarks/bench_variants.py > $OUT/${TAG}_secondary_paths.jsonl 2>> $OUT/bench.err
python3 $R/benchmarks/small_batch_probe.py > $OUT/${TAG}_small_batch_routing.jsonl 2>> $OUT/bench.err
python3 $R/benchmarks/bench_subset.py > $OUT/${TAG}_subset_device_side.jsonl 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/subs -- python3 $R/benchmarks/bench_subset.py > /dev/null 2>&1
cp $(find /tmp/subs -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_subset_kernel_stats.csv
bash $R/benchmarks/half_probe.sh > $OUT/half_probe.log 2>&1
cp $R/gpurun_out/half_probe.jsonl $OUT/${TAG}_half_pass_dims.jsonl
# ablation build (make -C minivectordb_amd/csrc ABLATE=1 before the run): per-tile timelines and the K-loop / epilogue split
ABL=$R/minivectordb_amd/lib/libmvdb_ablate.so
if [ -f $ABL ]; then
  MVDB_LIBMVDB=$ABL MVDB_GEMM_X3_DBG=5 python3 $R/benchmarks/x3_timeline.py 512 > $OUT/${TAG}_x3_timeline_s512.txt 2>&1
  MVDB_LIBMVDB=$ABL MVDB_GEMM_X3_DBG=5 MVDB_GEMM_X3_SPREAD=0 python3 $R/benchmarks/x3_timeline.py 512 > $OUT/${TAG}_x3_timeline_s512_dma_burst.txt 2>&1
  MVDB_LIBMVDB=$ABL MVDB_GEMM_X3_DBG=6 python3 $R/benchmarks/x3_timeline.py 512 > $OUT/${TAG}_x3_timeline_s512_mfma16x16x32_standin.txt 2>&1
  : > $OUT/${TAG}_x3_ablations.txt
  for d in 0 3 4; do
    for S in 32 512; do
      n=30; [ $S = 512 ] && n=5
      MVDB_LIBMVDB=$ABL MVDB_GEMM_LN_FUSED=2 MVDB_GEMM_X3_DBG=$d MVDB_S32_S=$S rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_${d}_$S -- python3 $R/benchmarks/bench_encoder_s32.py $n > /dev/null 2>&1
      f=$(find /tmp/abl_${d}_$S -name "*kernel_stats.csv" | head -1)
      echo "== MVDB_GEMM_X3_DBG=$d (0 full, 3 K loop only, 4 epilogue only), B = 256, S = $S: kernel, calls, average ns" >> $OUT/${TAG}_x3_ablations.txt
      grep -E "gemm_x3" $f | sed 's/_ZN12_GLOBAL__N_1//; s/EEEvPK[^"]*"//' | cut -d, -f1,2,4 >> $OUT/${TAG}_x3_ablations.txt
    done
  done
fi
ls -la $OUT
