#!/bin/bash
# Calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on known byte counts (run on the GPU box from the repo root):
#   bash tools/pmc_calibrate.sh <tag>   -> gpurun_out/<tag>_pmc_calibration.json
set -e
TAG=${1:-pmc}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out
[ -f $REPO/tools/pmc_calib/libpmc_calib.so ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o $REPO/tools/pmc_calib/libpmc_calib.so $REPO/tools/pmc_calib/calib.hip
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_${TAG}_$C
  rocprofv3 --pmc $C --output-format csv -d /tmp/cal_${TAG}_$C -o $C -- python $REPO/tools/pmc_calib/run.py 3 > /tmp/cal_${TAG}_$C.log 2>&1 || (tail -20 /tmp/cal_${TAG}_$C.log; exit 1)
done
python $REPO/tools/pmc_calibration.py /tmp/cal_${TAG}_FETCH_SIZE /tmp/cal_${TAG}_WRITE_SIZE $REPO/gpurun_out/${TAG}_pmc_calibration.json
