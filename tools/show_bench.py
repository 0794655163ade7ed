#!/usr/bin/env python
"""Summary of one bench.py JSON line:  python tools/show_bench.py gpurun_out/b1.json"""
import json
import sys

d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(d["value"], d["unit"], "| in scope:", r["in_scope"], "| with f3:", r.get("with_f3_losses"))
print("dominant:", r["kernel"], r["launch_us"], "us", r["frac"], "| parity:", d.get("parity"), "| host syncs:", (d.get("host_syncs") or {}).get("per_step"))
tot = 0.0
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["mean_us"] * kv[1]["launches_per_step"]):
    t = v["mean_us"] * v["launches_per_step"]
    tot += t
    print(f"{k:30s} {v['mean_us']:7.1f} x{v['launches_per_step']:5.1f} = {t:7.1f} us  {v.get('GBps')} GB/s")
print("sum", round(tot, 1))
