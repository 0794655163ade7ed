cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r8
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "flow or ponymation or xfm or render_mesh" > gpurun_out/r8/gpu_tests.txt 2>&1; tail -6 gpurun_out/r8/gpu_tests.txt | cut -c1-200
for w in ponymation fauna; do
python tools/glue_attribution.py --workload $w --steps 3 --out gpurun_out/r8/glue_$w.json > /dev/null 2>&1
python - $w <<'P'
import json,sys
d=json.load(open(f'gpurun_out/r8/glue_{sys.argv[1]}.json'))
print(sys.argv[1])
for k in ('in_scope_a3d_us_per_step','in_scope_glue_us_per_step','in_scope_total_us_per_step','glue_launches_per_step','a3d_in_scope_launches_per_step'): print(k,d[k])
for e in d['glue_top']: print(' ',e['where'][:70],e['launches_per_step'],e['us_per_step'])
P
done
