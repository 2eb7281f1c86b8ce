cd $GRAFT_REPO_ROOT
timeout 420 python3 -m pytest tests/test_config4_gpu.py -m gpu -q -x 2>&1 | tail -15 | cut -c1-300
echo "---- second: only the two-rank test"
timeout 300 python3 -m pytest tests/test_config4_gpu.py -m gpu -q -x -k two_rank 2>&1 | tail -5 | cut -c1-300
