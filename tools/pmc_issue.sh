#!/bin/bash
# Issue / stall / parked shares of every hot-path kernel's wave cycles (tools/pmc_issue.py), one rocprofv3 --pmc pass over the bench step.
#   bash tools/pmc_issue.sh <tag> [bench.py arguments]   -> gpurun_out/<tag>_pmc_issue.txt
set -e
TAG=${1:-pmc}
shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_issue_$TAG
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d /tmp/pmc_issue_$TAG -o issue -- \
  python $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-fingerprint --no-extra-legs --networks fast "$@" > /tmp/pmc_issue_$TAG.log 2>&1 || (tail -20 /tmp/pmc_issue_$TAG.log; exit 1)
python $REPO/tools/pmc_issue.py /tmp/pmc_issue_$TAG > $REPO/gpurun_out/${TAG}_pmc_issue.txt
cat $REPO/gpurun_out/${TAG}_pmc_issue.txt
