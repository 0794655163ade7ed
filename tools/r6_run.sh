cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "switch" 2>&1 | tail -5 | cut -c1-250
