# rocprofv3 kernel trace of the encoder forward at B = 256, S = 32 (BASELINE config 5 shape), plain launches (no graph)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r2enc}
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/enc_trace_${1:-r2enc} -- python3 $R/benchmarks/bench_encoder_s32.py 30 $2 > /tmp/enc_trace.log 2>&1
f=$(find /tmp/enc_trace_${1:-r2enc} -name "*_kernel_stats.csv" | head -1)
cp $f $OUT/encoder_s32_kernel_stats.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print(f'{float(r["TotalDurationNs"])/tot*100:5.1f}%  calls {r["Calls"]:>5}  avg {float(r["AverageNs"])/1e3:8.1f} us  {r["Name"][:95]}')
print("total per forward (ms):", tot / 30 / 1e6)
PY
