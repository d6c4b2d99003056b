cd $GRAFT_REPO_ROOT
ISCC_HIP_OPTS="mfma=0" python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_bench_contract.py 2>&1 | tail -25
