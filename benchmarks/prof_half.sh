# PMC passes over the certified pass's main launches, flat_scan_h16_kernel (bench.py --nq N --dim D): per corpus pass sums.
# usage (GPU box): bash benchmarks/prof_half.sh "128 256" 512 [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
NQS=${1:-"128 256"}
DIM=${2:-512}
OUT=$R/gpurun_out/${3:-half_pmc}
mkdir -p $OUT
SETS="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES;SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM;SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS"
IFS=';' read -ra ARR <<< "$SETS"
STEPS=6
for nq in $NQS; do
  [ -z "$SKIP_TRACE" ] && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/htr_${nq} -- python3 $R/bench.py --nq $nq --dim $DIM --steps $STEPS --warmup 2 --no-cpu-baseline > $OUT/bench_nq${nq}_d${DIM}.json 2>/dev/null
  python3 - <<PY >> $OUT/pmc_summary_d${DIM}.txt
import csv, glob
f = glob.glob("/tmp/htr_${nq}/**/*kernel_stats.csv", recursive=True)
for fn in f:
    for r in csv.DictReader(open(fn)):
        if "half" in r["Name"] or "seed" in r["Name"] or "certify" in r["Name"]:
            print("nq=${nq} trace", r["Name"][:70], "calls", r["Calls"], "avg_ns", r["AverageNs"], "total_ns", r["TotalDurationNs"])
PY
  for set in "${ARR[@]}"; do
    tag=$(echo $set | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/hpmc_${nq}_${tag} -- python3 $R/bench.py --nq $nq --dim $DIM --steps $STEPS --warmup 2 --no-cpu-baseline > /tmp/hpmc_${nq}_${tag}.log 2>&1
    python3 - <<PY >> $OUT/pmc_summary_d${DIM}.txt
import csv, glob, collections
f = glob.glob("/tmp/hpmc_${nq}_${tag}/**/*_counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        if "flat_scan_h16_kernel" in r["Kernel_Name"]:
            agg["main"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    # launches: (warmup + steps + latency loop) passes x 3 phases; report the sum per corpus pass
    print("nq=${nq}", {n: round(sum(v) / (len(v) / 3.0) / 1e6, 3) for n, v in c.items()}, "(millions per corpus pass, 3 main launches)")
PY
  done
done
cat $OUT/pmc_summary_d${DIM}.txt
