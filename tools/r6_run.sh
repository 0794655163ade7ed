set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dmtet or surface or speculative or DMTet" > gpurun_out/r4/tests.txt 2>&1; tail -3 gpurun_out/r4/tests.txt
python tools/glue_attribution.py --workload magicpony --steps 3 --out gpurun_out/r4/glue_magicpony.json > /dev/null 2>&1
export A3D_LIB=$R/3danimals_amd/lib/liba3d_hip_exp.so
cd /tmp && export TMPDIR=/tmp
for e in 0 45; do for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pn_$e_$C
  A3D_EXP=$e rocprofv3 --pmc $C --output-format csv -d /tmp/pn_${e}_$C -o $C -- python $R/tools/bench_normals.py > /tmp/pn.log 2>&1 || tail -5 /tmp/pn.log
  python $R/tools/pmc_summary.py /tmp/pn_${e}_$C nr_ > $R/gpurun_out/r4/normals_exp${e}_$C.txt
done; done
cd $R
head -20 gpurun_out/r4/normals_exp*_*.txt
