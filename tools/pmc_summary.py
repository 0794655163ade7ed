#!/usr/bin/env python
"""Per-kernel HBM traffic from a rocprofv3 --pmc counter_collection.csv (FETCH_SIZE / WRITE_SIZE, in KiB per dispatch).

usage: python tools/pmc_summary.py <dir-or-csv> [name-substring ...]
Applies the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced reads by 2x
(so a range [raw, 2*raw] is printed); WRITE_SIZE is uncalibrated.
"""
import csv
import glob
import sys
from collections import defaultdict

path = sys.argv[1]
subs = sys.argv[2:] or ["rs_", "gb_", "ip_", "aa_", "dm_", "sk_", "nr_", "ss_"]
files = [path] if path.endswith(".csv") else glob.glob(path + "/**/*counter_collection.csv", recursive=True)
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name") or r.get("Kernel Name") or ""
        if not any(s in name for s in subs):
            continue
        acc[name.replace("(anonymous namespace)::", "").split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"{'kernel':42s} {'calls':>5s} {'FETCH MB (raw..x2)':>22s} {'WRITE MB':>10s}")
for k, v in sorted(acc.items()):
    fe, wr = v.get("FETCH_SIZE", []), v.get("WRITE_SIZE", [])
    n = max(len(fe), len(wr))
    fm = sum(fe) / max(len(fe), 1) * 1024 / 1e6 if fe else float("nan")
    wm = sum(wr) / max(len(wr), 1) * 1024 / 1e6 if wr else float("nan")
    print(f"{k:42s} {n:5d} {fm:10.2f}..{2*fm:<10.2f} {wm:10.2f}")
