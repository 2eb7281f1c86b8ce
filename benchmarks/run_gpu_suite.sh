cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_fulltest; mkdir -p $OUT
timeout 2400 python3 -m pytest tests -m gpu -q -x --durations=15 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"
tail -25 $OUT/pytest_gpu.txt
