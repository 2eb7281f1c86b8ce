cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03l; mkdir -p $OUT
export MVDB_TEST_SUBPROCESS_TIMEOUT=180
timeout 900 python3 -m pytest tests/test_flat_gpu.py tests/test_golden_gpu.py tests/test_exchange_gpu.py -m gpu -q -x -k "masked or row_sets or exclude or golden or subset or exchange" > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest.txt | cut -c1-220
python3 benchmarks/bench_subset.py 2>/dev/null | cut -c1-200
python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-encoder | python3 -c "import json,sys; r=json.load(sys.stdin); print(r['value'], r['roofline']['frac'], r['roofline']['avg_launch_ms'])"
