#!/bin/bash
# Everything under profiles/ for one round, in the order that makes the committed lines self-consistent (run on the GPU box from the repo
# root, then copy gpurun_out/<tag>_* into profiles/):   bash tools/regen_profiles.sh r03
TAG=${1:-r03}
mkdir -p gpurun_out
bash tools/pmc_traffic.sh $TAG > gpurun_out/${TAG}_pmc.log 2>&1
cp gpurun_out/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json   # (on the box: bench.py then finds the stamp of the sources it runs on)
bash tools/run_bench_lines.sh $TAG > gpurun_out/${TAG}_lines.log 2>&1
bash tools/prof_bench.sh ${TAG}_bench 20 5 > /dev/null 2>&1
tail -3 gpurun_out/${TAG}_lines.log
