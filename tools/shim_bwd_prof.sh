#!/bin/bash
# kernel durations of the modular backward kernels, tile-scatter (A3D_EXP=0) against the per-pixel forms (A3D_EXP=140), experiment library
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export A3D_LIB=$REPO/3danimals_amd/lib/liba3d_hip_exp.so
for e in ${EXPS:-0 140}; do
  rm -rf /tmp/bp_$e
  A3D_EXP=$e timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bp_$e -o bp -- python $REPO/tools/shim_bwd_bench.py > /tmp/bp_$e.log 2>&1
  echo "== A3D_EXP=$e"; grep -E "us/call|covered" /tmp/bp_$e.log
  f=/tmp/bp_$e/bp_kernel_stats.csv
  if [ -f $f ]; then python - <<PY
import csv
for r in csv.DictReader(open("$f")):
    if any(k in r["Name"] for k in ("rs_bwd", "ip_bwd", "fill", "memset", "Fill")): print(r["Name"][:40], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
else echo "no stats file"; ls /tmp/bp_$e; fi
done
