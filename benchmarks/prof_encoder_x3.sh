# PMC passes over the encoder's GEMM kernels at B = 256, S = 32 (benchmarks/bench_encoder_s32.py): per-launch averages.
# usage (GPU box): bash benchmarks/prof_encoder_x3.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-enc_pmc}
mkdir -p $OUT
SETS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum;SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY;SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"
IFS=';' read -ra ARR <<< "$SETS"
: > $OUT/pmc_summary.txt
for set in "${ARR[@]}"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/epmc_${tag} -- python3 $R/benchmarks/bench_encoder_s32.py ${MVDB_PROF_ITERS:-6} > /tmp/epmc_${tag}.log 2>&1
  python3 - <<PY >> $OUT/pmc_summary.txt
import csv, glob, collections
f = glob.glob("/tmp/epmc_${tag}/**/*_counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        n = r["Kernel_Name"]
        if "gemm" in n or "attention" in n or "ln_kernel" in n:
            key = n.split("(")[0].replace("void (anonymous namespace)::", "")[:40] + "|grid" + r.get("Grid_Size", "?")
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(agg.items()):
    print(k, {n: round(sum(v) / len(v) / 1e6, 3) for n, v in c.items()}, "(millions per launch)", len(next(iter(c.values()))), "launches")
PY
done
cat $OUT/pmc_summary.txt
