# PMC pass over the encoder forward at B = 256, S = 32: MFMA-pipe busy cycles and clock per kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r2enc}
mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d /tmp/enc_pmc -- python3 $R/benchmarks/bench_encoder_s32.py 10 > /tmp/enc_pmc.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/enc_pmc2 -- python3 $R/benchmarks/bench_encoder_s32.py 10 > /tmp/enc_pmc2.log 2>&1
python3 - <<PY | tee $OUT/encoder_s32_pmc.txt
import csv, glob, collections
for d in ("/tmp/enc_pmc", "/tmp/enc_pmc2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", kv[1].get("SQ_ACTIVE_INST_ANY", [0])))):
        n = len(next(iter(c.values())))
        if n < 10: continue
        print(k, n, {m: round(sum(v) / len(v) / 1e3, 1) for m, v in c.items()}, "(thousands per launch)")
PY
