cd $GRAFT_REPO_ROOT
for v in "" "MVDB_GEMM_LN_FUSED=0" "MVDB_ATTENTION_IMG=0" "MVDB_GEMM_LN_FUSED=0 MVDB_ATTENTION_IMG=0"; do echo "=== $v"; env $v python3 benchmarks/r03_det.py 2>&1 | tail -9; done
