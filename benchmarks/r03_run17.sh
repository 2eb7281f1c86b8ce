cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03m; mkdir -p $OUT
export MVDB_TEST_SUBPROCESS_TIMEOUT=180
timeout 1500 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest.txt | cut -c1-220
