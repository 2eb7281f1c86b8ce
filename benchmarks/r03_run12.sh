cd $GRAFT_REPO_ROOT
export MVDB_TEST_SUBPROCESS_TIMEOUT=150
OUT=gpurun_out/r03h; mkdir -p $OUT
timeout 1200 python3 -m pytest tests/test_flat_gpu.py tests/test_config5_gpu.py tests/test_config4_gpu.py tests/test_exchange_gpu.py tests/test_golden_gpu.py -m gpu -q -x > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -40 $OUT/pytest.txt | cut -c1-220
