cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r7
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "xfm or stagewise or full_size or pinned or gradients or overlay or animal_model or render_mesh or e2e or fauna or ponymation or normals" > gpurun_out/r7/gpu_tests.txt 2>&1; tail -6 gpurun_out/r7/gpu_tests.txt
python tools/glue_attribution.py --workload magicpony --steps 3 --out gpurun_out/r7/glue_magicpony.json > /dev/null 2>&1
python - <<'P'
import json
d=json.load(open('gpurun_out/r7/glue_magicpony.json'))
for k in ('in_scope_a3d_us_per_step','in_scope_glue_us_per_step','in_scope_total_us_per_step','glue_launches_per_step','a3d_in_scope_launches_per_step'): print(k,d[k])
for e in d['glue_top']: print(' ',e['where'][:70],e['launches_per_step'],e['us_per_step'])
P
