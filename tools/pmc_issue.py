#!/usr/bin/env python
"""Where the waves of each hot-path kernel spend their cycles, from ONE rocprofv3 --pmc pass over the bench step (SQ counters):

    bash tools/pmc_issue.sh <tag>      (on the GPU box)  ->  gpurun_out/<tag>_pmc_issue.txt

SQ_WAVE_CYCLES = ACTIVE_INST_ANY + WAIT_INST_ANY + WAIT_ANY (quad-cycles, summed over waves; /opt/skills/guides/MI355X_MICROARCH.md):
  issue  = ACTIVE_INST_ANY / WAVE_CYCLES   a wave was issuing (of which valu = ACTIVE_INST_VALU / WAVE_CYCLES)
  stall  = WAIT_INST_ANY / WAVE_CYCLES     it had an instruction but the pipe / a hazard held it
  parked = WAIT_ANY / WAVE_CYCLES          it sat at s_waitcnt / s_barrier (memory, LDS, other waves)
and VALU instructions per wave.  A kernel whose waves are mostly parked is bound by round trips; one with a high issue share by the
instructions it issues -- fewer instructions is then the lever (the skinning launches, round 4), not fewer bytes.
usage: python tools/pmc_issue.py <dir-with-counter_collection.csv> [kernel-name-prefix ...]
"""
import csv
import glob
import sys
from collections import defaultdict

path = sys.argv[1]
subs = sys.argv[2:] or ["rs_", "gb_", "ip_", "aa_", "ca_", "dm_", "sk_", "nr_", "tp_", "cv_", "sh_", "ls_", "bn_"]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = (r.get("Kernel_Name") or r.get("Kernel Name") or "").replace("(anonymous namespace)::", "")
        short = name.split("(")[0][:44]
        if any(short.startswith(s) or (" " + s) in short for s in subs):
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"{'kernel':46s} {'calls':>5s} {'waves':>8s} {'VALU/wave':>9s} {'issue':>6s} {'valu':>6s} {'stall':>6s} {'parked':>6s}")
rows = []
for k, v in acc.items():
    m = lambda c: sum(v.get(c, [0.0])) / max(len(v.get(c, [1.0])), 1)
    wc = m("SQ_WAVE_CYCLES")
    if wc <= 0:
        continue
    rows.append((m("SQ_WAVE_CYCLES"), k, len(v["SQ_WAVE_CYCLES"]), m("SQ_WAVES"), m("SQ_INSTS_VALU") / max(m("SQ_WAVES"), 1.0), m("SQ_ACTIVE_INST_ANY") / wc,
                 m("SQ_ACTIVE_INST_VALU") / wc, m("SQ_WAIT_INST_ANY") / wc, m("SQ_WAIT_ANY") / wc))
for _, k, n, waves, ipw, issue, valu, stall, parked in sorted(rows, reverse=True):
    print(f"{k:46s} {n:5d} {waves:8.0f} {ipw:9.0f} {issue:6.2f} {valu:6.2f} {stall:6.2f} {parked:6.2f}")
