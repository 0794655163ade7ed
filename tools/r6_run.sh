cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/glue_attribution.py --workload fauna --steps 3 --out gpurun_out/r06_glue_attribution_fauna.json > gpurun_out/r06_glue_fauna.log 2>&1 || tail -5 gpurun_out/r06_glue_fauna.log
timeout 1500 python -m pytest tests/ -m gpu -q > gpurun_out/r06_gpu_tests.txt 2>&1; tail -8 gpurun_out/r06_gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
