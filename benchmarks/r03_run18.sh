cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03n; mkdir -p $OUT
export MVDB_TEST_SUBPROCESS_TIMEOUT=180
timeout 1500 python3 -m pytest tests/test_flat_gpu.py tests/test_golden_gpu.py -m gpu -q -x -k "split_precision_batch or margin or rerun or golden or distributed or capturable or large_batch or outside" > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest.txt | cut -c1-220



