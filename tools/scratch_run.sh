set -x
bash tools/run_bench_lines.sh r03 > gpurun_out/r03_lines.log 2>&1
bash tools/prof_bench.sh r03_bench 20 5 > /dev/null 2>&1
bash tools/pmc_traffic.sh r03 > gpurun_out/r03_pmc.log 2>&1; tail -3 gpurun_out/r03_pmc.log
python bench.py > gpurun_out/r03_bench_default2.json 2> /dev/null
