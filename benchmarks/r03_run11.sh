cd $GRAFT_REPO_ROOT
export MVDB_TEST_SUBPROCESS_TIMEOUT=120
echo "== capture test + two-rank"
timeout 400 python3 -m pytest tests/test_flat_gpu.py tests/test_config4_gpu.py -m gpu -q -x -k "capturable or two_rank" 2>&1 | tail -4 | cut -c1-200
echo "== config5 + two-rank"
timeout 400 python3 -m pytest tests/test_config5_gpu.py tests/test_config4_gpu.py -m gpu -q -x -k "config5 or two_rank" 2>&1 | tail -4 | cut -c1-200
