cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/ -m gpu -q > gpurun_out/r6/gpu_tests.txt 2>&1; tail -8 gpurun_out/r6/gpu_tests.txt
