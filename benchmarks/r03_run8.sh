cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03g; mkdir -p $OUT
timeout 1500 python3 -m pytest tests/test_flat_gpu.py tests/test_config5_gpu.py tests/test_config4_gpu.py tests/test_exchange_gpu.py tests/test_golden_gpu.py -m gpu -q -x > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest.txt | cut -c1-200
python3 bench.py --nq 256 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_nq256.json 2>$OUT/bench_nq256.err; python3 -c "
import json; r=json.load(open('$OUT/bench_nq256.json')); print('nq256', r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['avg_launch_ms'])"
python3 bench.py --nq 128 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_nq128.json 2>$OUT/bench_nq128.err; python3 -c "
import json; r=json.load(open('$OUT/bench_nq128.json')); print('nq128', r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['avg_launch_ms'])"
