#!/bin/bash
# rocprofv3 kernel trace of the default bench command (run on the GPU box from the repo root):
#   bash tools/prof_bench.sh <tag> [steps] [warmup] [extra bench.py arguments ...]
#   -> gpurun_out/<tag>_kernel_stats.csv, gpurun_out/<tag>_kernel_summary.txt
set -e
TAG=${1:-bench}
STEPS=${2:-20}
WARM=${3:-5}
shift 3 2>/dev/null || shift $#
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o $TAG -- python $REPO/bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-kernel-timing --no-fingerprint --networks fast "$@" > /tmp/prof_$TAG.log 2>&1 || (tail -20 /tmp/prof_$TAG.log; exit 1)
grep '^{"metric"' /tmp/prof_$TAG.log | tail -1 > $REPO/gpurun_out/${TAG}_bench_line.json   # (the JSON line, not the profiler's last log line)
cp /tmp/prof_$TAG/${TAG}_kernel_stats.csv $REPO/gpurun_out/${TAG}_kernel_stats.csv
python $REPO/tools/rocprof_summary.py /tmp/prof_$TAG/${TAG}_kernel_stats.csv $((STEPS + WARM)) 30 > $REPO/gpurun_out/${TAG}_kernel_summary.txt
head -60 $REPO/gpurun_out/${TAG}_kernel_summary.txt
