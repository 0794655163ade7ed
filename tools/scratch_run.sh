python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_gpu.log
bash tools/prof_bench.sh r03c 20 5 > /dev/null 2>&1
grep -E "rs_|cv_|sk_|tp_|dm_|nr_|gb_fwd|aa_an" gpurun_out/r03c_kernel_summary.txt | head -30
python bench.py --no-cpu-baseline --networks fast > gpurun_out/r03c_bench.json 2> gpurun_out/r03c_bench.err
