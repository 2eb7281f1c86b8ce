# PMC passes over flat_scan_split128_kernel (bench.py --nq 128) for the ablation variants; prints per-launch averages.
# usage (GPU box): bash benchmarks/prof_split128.sh "0 2 4" "SET1;SET2;..."
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${3:-r2c}
mkdir -p $OUT
DBGS=${1:-"0 2 4"}
SETS=${2:-"SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES;SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM;SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC"}
IFS=';' read -ra ARR <<< "$SETS"
for dbg in $DBGS; do
  for set in "${ARR[@]}"; do
    tag=$(echo $set | cut -d' ' -f1)
    MVDB_SPLIT128_DBG=$dbg timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_${dbg}_${tag} -- python3 $R/bench.py --nq 128 --steps 6 --warmup 2 --no-cpu-baseline > /tmp/pmc_${dbg}_${tag}.log 2>&1
    python3 - <<PY >> $OUT/pmc_summary.txt
import csv, glob, collections
f = glob.glob("/tmp/pmc_${dbg}_${tag}/**/*_counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        if "split128" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print("dbg=${dbg}", {n: round(sum(v)/len(v)/1e6, 2) for n, v in c.items()}, "(millions per launch)")
PY
  done
done
cat $OUT/pmc_summary.txt
