#!/bin/bash
# HBM traffic counters of the bench workload, two separate rocprofv3 --pmc passes (run on the GPU box from the repo root):
#   bash tools/pmc_traffic.sh <tag>   -> gpurun_out/<tag>_pmc_traffic.json, gpurun_out/<tag>_pmc_{fetch,write}_size.txt
set -e
TAG=${1:-pmc}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_${TAG}_$C
  rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_${TAG}_$C -o $C -- python $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-fingerprint --networks fast > /tmp/pmc_${TAG}_$C.log 2>&1 || (tail -20 /tmp/pmc_${TAG}_$C.log; exit 1)
done
python $REPO/tools/pmc_summary.py /tmp/pmc_${TAG}_FETCH_SIZE rs_ gb_ ip_ aa_ ca_ dm_ sk_ nr_ tp_ ss_ bn_ cv_ sh_ ls_ fl_ he_ gm_ xf_ > $REPO/gpurun_out/${TAG}_pmc_fetch_size.txt
python $REPO/tools/pmc_summary.py /tmp/pmc_${TAG}_WRITE_SIZE rs_ gb_ ip_ aa_ ca_ dm_ sk_ nr_ tp_ ss_ bn_ cv_ sh_ ls_ fl_ he_ gm_ xf_ > $REPO/gpurun_out/${TAG}_pmc_write_size.txt
CAL=$(ls $REPO/profiles/*_pmc_calibration.json 2>/dev/null | tail -1)
python $REPO/tools/pmc_traffic.py /tmp/pmc_${TAG}_FETCH_SIZE /tmp/pmc_${TAG}_WRITE_SIZE $REPO/gpurun_out/${TAG}_pmc_traffic.json $CAL
