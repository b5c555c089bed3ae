cd $GRAFT_REPO_ROOT; timeout 900 python -m pytest tests/test_gpu_gemv4.py tests/test_gpu_path.py -q -x 2>&1 | tail -2
