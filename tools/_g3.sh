cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_narrow_gpu.py -q 2>&1 | tail -5
