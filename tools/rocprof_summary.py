#!/usr/bin/env python
"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a short, committed summary (profiles/).

usage: python tools/rocprof_summary.py <kernel_stats.csv> <steps_profiled> [top_n]
"""
import csv
import re
import sys


def short(name):
    if name.startswith("Cijk_"):
        mt = re.search(r"MT(\d+x\d+x\d+)", name)
        return "rocBLAS/Tensile GEMM f32 " + (name[5:14]) + " MT" + (mt.group(1) if mt else "?")
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"at::native::", "", name)
    m = re.match(r"void\s+([A-Za-z_0-9:]+)<(.{0,80})", name)
    if m:
        return (m.group(1) + "<" + m.group(2))[:110]
    return name[:110]


def category(name):
    name = name.replace("(anonymous namespace)::", "")
    if name.startswith("Cijk_") or "rocblas" in name:
        return "torch GEMM (MLPs)"
    for tag in ("rs_", "ip_", "aa_", "ca_", "dm_", "sk_", "nr_", "gb_", "tp_", "bn_", "cv_", "sh_", "ss_", "ls_", "fl_", "he_", "gm_"):
        if name.startswith(tag) or name.startswith("void " + tag):
            return "a3d HIP kernels"
    if "rocclr" in name or "fillBuffer" in name.lower():
        return "memcpy/memset"
    return "torch elementwise/index/reduce"


def main():
    path, steps = sys.argv[1], float(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    rows = list(csv.DictReader(open(path)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    cats = {}
    for r in rows:
        c = category(r["Name"])
        cats[c] = cats.get(c, 0.0) + float(r["TotalDurationNs"])
    print(f"# rocprofv3 kernel stats: {len(rows)} distinct kernels, {total/1e6:.2f} ms GPU time over ~{steps:g} profiled steps "
          f"(incl. warm-up/setup) = {total/1e6/steps:.2f} ms/step")
    print("\n## by category")
    for c, t in sorted(cats.items(), key=lambda x: -x[1]):
        print(f"{c:36s} {t/1e6:9.2f} ms  {100*t/total:5.1f} %")
    print("\n## a3d HIP kernels (all)")
    print(f"{'kernel':64s} {'calls':>6s} {'avg us':>9s} {'min us':>9s} {'max us':>9s} {'total ms':>9s}")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        if category(r["Name"]) == "a3d HIP kernels":
            print(f"{short(r['Name'])[:64]:64s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} {float(r['MinNs'])/1e3:9.1f} "
                  f"{float(r['MaxNs'])/1e3:9.1f} {float(r['TotalDurationNs'])/1e6:9.2f}")
    print(f"\n## top {top} kernels overall")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
        print(f"{short(r['Name'])[:84]:84s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.2f} %")


if __name__ == "__main__":
    main()
