#!/bin/bash
# Kernel-level timing of the HIP path with rocprofv3 (run on the GPU box from the repo root):
#   bash tools/prof_kernels.sh <tag>        -> gpurun_out/<tag>_kernels.txt
set -e
TAG=${1:-kern}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o $TAG -- python $REPO/tools/microbench.py --iters 20 > /tmp/prof_$TAG.log 2>&1 || (tail -20 /tmp/prof_$TAG.log; exit 1)
python $REPO/tools/rocprof_summary.py /tmp/prof_$TAG/${TAG}_kernel_stats.csv 1 0 | sed -n '/a3d HIP kernels/,/top 0/p' > $REPO/gpurun_out/${TAG}_kernels.txt
cat $REPO/gpurun_out/${TAG}_kernels.txt
