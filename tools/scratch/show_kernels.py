import json, sys
d = json.loads(sys.stdin.read())
pat = sys.argv[1] if len(sys.argv) > 1 else ""
print(d["value"], d["ms_per_step"], {k: v["mean_us"] for k, v in d["kernels"].items() if pat in k})
