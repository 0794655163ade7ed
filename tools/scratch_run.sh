mkdir -p gpurun_out/diag
for w in magicpony fauna ponymation; do
  n=4; [ $w = ponymation ] && n=8
  timeout 900 python tools/parity_diag.py --workload $w --steps 0 35 --n $n --tuned > gpurun_out/diag/$w.jsonl 2> gpurun_out/diag/$w.err
  echo $w $?; tail -3 gpurun_out/diag/$w.err
done
