python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/pytest_gpu.log | head
bash tools/prof_bench.sh r03c 20 5 > /dev/null 2>&1
grep -E "^(rs_|cv_|sk_|tp_|dm_|nr_|gb_|aa_|ca_|sh_)" gpurun_out/r03c_kernel_summary.txt | head -40
python bench.py --no-cpu-baseline --networks fast > gpurun_out/r03c_bench.json 2> gpurun_out/r03c_bench.err
