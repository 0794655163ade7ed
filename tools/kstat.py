#!/usr/bin/env python
"""print calls/avg/min/max (us) of kernels whose name contains a substring, from a rocprofv3 *_kernel_stats.csv"""
import csv, glob, sys
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_stats.csv", recursive=True) if not path.endswith(".csv") else [path]
for f in files:
    for r in csv.DictReader(open(f)):
        if any(s in r["Name"] for s in sys.argv[2:]):
            print(f"{r['Name'][:48]:48s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f} us")
