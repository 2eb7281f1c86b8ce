#!/bin/bash
# A/B runs of the fp16 nomination pass (half_scan.hip) on one MI355X: writes gpurun_out/half_probe.jsonl
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/half_probe.jsonl
: > $out
run() { # label, env..., -- bench args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  line=$(env "${envs[@]}" python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1)
  echo "{\"label\": \"$label\", \"bench\": $line}" >> $out
}
run "d512 nq128" X=1 -- --nq 128 --steps 20 --warmup 3

run "d512 nq256" X=1 -- --nq 256 --steps 20 --warmup 3
run "d512 nq128 on the fp32 rows" MVDB_DISABLE_HALF_SHADOW=1 -- --nq 128 --steps 20 --warmup 3
run "d512 nq256 on the fp32 rows" MVDB_DISABLE_HALF_SHADOW=1 -- --nq 256 --steps 20 --warmup 3
run "d512 nq64" X=1 -- --nq 64 --steps 20 --warmup 3
run "d384 nq128" X=1 -- --nq 128 --dim 384 --steps 20 --warmup 3
run "d384 nq256" X=1 -- --nq 256 --dim 384 --steps 20 --warmup 3
run "d384 nq256 on the fp32 rows" MVDB_DISABLE_HALF_SHADOW=1 -- --nq 256 --dim 384 --steps 20 --warmup 3
run "d256 nq256" X=1 -- --nq 256 --dim 256 --steps 20 --warmup 3
run "d1024 nq128" X=1 -- --nq 128 --dim 1024 --rows 5000000 --steps 20 --warmup 3
run "d1024 nq128 on the fp32 rows" MVDB_DISABLE_HALF_SHADOW=1 -- --nq 128 --dim 1024 --rows 5000000 --steps 20 --warmup 3
run "d768 nq128" X=1 -- --nq 128 --dim 768 --rows 5000000 --steps 20 --warmup 3
run "d768 nq128 on the fp32 rows" MVDB_DISABLE_HALF_SHADOW=1 -- --nq 128 --dim 768 --rows 5000000 --steps 20 --warmup 3
python - <<'PY'
import json
for l in open("gpurun_out/half_probe.jsonl"):
    r = json.loads(l); b = r["bench"]; rf = b["roofline"]
    print(f'{r["label"]:32s} {b["value"]:10.0f} q/s  {b["ms_per_step"]:7.3f} ms/step  kernel {rf["kernel"]:26s} frac {rf["frac"]:.3f}  launches/pass {rf.get("launches_per_corpus_pass")}')
PY
