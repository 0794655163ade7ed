#!/bin/bash
# Everything under profiles/ for one round, in the order that makes the committed lines self-consistent (run on the GPU box from the repo
# root, then copy gpurun_out/<tag>_* into profiles/):   bash tools/regen_profiles.sh r04
TAG=${1:-r06}
mkdir -p gpurun_out
bash tools/pmc_traffic.sh $TAG > gpurun_out/${TAG}_pmc.log 2>&1
cp gpurun_out/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json   # (on the box: bench.py then finds the stamp of the sources it runs on)
bash tools/run_bench_lines.sh $TAG > gpurun_out/${TAG}_lines.log 2>&1
bash tools/prof_bench.sh ${TAG}_bench 20 5 > /dev/null 2>&1
bash tools/prof_bench.sh ${TAG}_bcc51s 20 5 --grid bcc51s > /dev/null 2>&1
# (round 5) rocprofv3 summaries of the other committed lines too: R = 128, ponymation (rendered and as configured), fauna, the trained-like mesh, the long run
bash tools/prof_bench.sh ${TAG}_grid128 10 3 --grid-res 128 > /dev/null 2>&1
bash tools/prof_bench.sh ${TAG}_ponymation 10 3 --workload ponymation > /dev/null 2>&1
bash tools/prof_bench.sh ${TAG}_ponymation_norender 10 3 --workload ponymation --no-render --batch 20 --frames 10 > /dev/null 2>&1
bash tools/prof_bench.sh ${TAG}_fauna 20 5 --workload fauna > /dev/null 2>&1
bash tools/prof_bench.sh ${TAG}_spiky 20 5 --mesh spiky > /dev/null 2>&1
bash tools/prof_bench.sh ${TAG}_long400 400 100 > /dev/null 2>&1
python tools/bench_dmtet.py --grid kuhn64 bcc51s kuhn128 bcc102s --surf --passes plain auto --json gpurun_out/${TAG}_dmtet_grids.json > gpurun_out/${TAG}_dmtet_grids.txt 2>&1
python tools/numbering_diag.py --grids kuhn64 kuhn64s bcc51 bcc51s --json gpurun_out/${TAG}_numbering.json > /dev/null 2>&1
python tools/long_run_diag.py 600 2>&1 | grep "^[0-9]" > gpurun_out/${TAG}_long_run_diag.txt
bash tools/pmc_issue.sh $TAG > /dev/null 2>&1          # issue / stall / parked shares of every kernel's wave cycles
python 3danimals_amd/csrc/build.py --profile > /dev/null 2>&1 && python tools/kernel_phases.py 2>/dev/null | grep -v Warning > gpurun_out/${TAG}_kernel_phases.txt   # phase stamps inside the kernels, real step
(hipcc --offload-arch=gfx950 -O3 -std=c++17 -DA3D_PROFILE -fhip-fp32-correctly-rounded-divide-sqrt -I include -I 3danimals_amd/csrc tools/skin_phases/phases.hip 3danimals_amd/csrc/common.hip -o gpurun_out/skin_phases 2>/dev/null && (gpurun_out/skin_phases; gpurun_out/skin_phases 16 23800) > gpurun_out/${TAG}_skin_phases.txt 2>&1; rm -f gpurun_out/skin_phases)
# (round 6) who launched what: every GPU kernel of the step attributed to the code that launched it, for the three workloads
for w in magicpony fauna ponymation; do python tools/glue_attribution.py --workload $w --steps 3 --out gpurun_out/${TAG}_glue_attribution_$w.json > /dev/null 2>&1; done
# (round 6) eight ranks on ONE GPU (gloo): the N = 8 launch path and the host-side contention of eight Python processes, measured; no scaling claim
python bench.py --gpus 8 --backend gloo --share-gpu --steps 10 --warmup 3 --no-cpu-baseline --no-fingerprint > gpurun_out/${TAG}_bench_ranks8_one_gpu.json 2> gpurun_out/${TAG}_bench_ranks8_one_gpu.err
tail -3 gpurun_out/${TAG}_lines.log
