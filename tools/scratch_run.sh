python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/pytest_gpu.log | head
python bench.py --no-cpu-baseline --networks fast > gpurun_out/r03c_bench.json 2> gpurun_out/r03c_bench.err
