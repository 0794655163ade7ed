bash tools/run_bench_lines.sh r03 > gpurun_out/r03_lines.log 2>&1
bash tools/prof_bench.sh r03_bench 20 5 > /dev/null 2>&1
bash tools/pmc_traffic.sh r03 > gpurun_out/r03_pmc.log 2>&1; tail -2 gpurun_out/r03_pmc.log
