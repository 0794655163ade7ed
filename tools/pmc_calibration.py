#!/usr/bin/env python
"""FETCH_SIZE / WRITE_SIZE reported by rocprofv3 for the known-byte-count kernels of tools/pmc_calib against what they move.

usage: python tools/pmc_calibration.py <fetch-dir> <write-dir> <out.json>
"""
import csv
import glob
import json
import sys
from collections import defaultdict

GiB = float(1 << 30)
# kernel -> (bytes read, bytes written as the program sees them, lines (64 B) it must touch on the memory side, note)
KNOWN = {
    "cal_fill16": (0, GiB, GiB, "16-byte streaming stores, 1 GiB"),
    "cal_fill4": (0, GiB, GiB, "4-byte streaming stores (64 lanes = 4 lines), 1 GiB"),
    "cal_read16": (GiB, 0, GiB, "16-byte streaming loads, 1 GiB"),
    "cal_store_per_line": (0, 4.0 * (1 << 20), 64.0 * (1 << 20), "one 4-byte store per 64-byte line, 1M lines"),
    "cal_gather4": (4.0 * (1 << 20) + 4.0 * (1 << 20), 0, 64.0 * (1 << 20) + 4.0 * (1 << 20), "1M random 4-byte loads from 1 GiB (+ 4 MB of indices streamed)"),
    "cal_atomic_min64": (1.6e6, 8.0 * 4e5, 64.0 * 4e5, "4e5 fire-and-forget 64-bit atomicMin on 8 MB of keys (+ 1.6 MB of indices)"),
    "cal_atomic_rows": (0.56e6, 64.0 * 1.4e5, 64.0 * 1.4e5, "1.4e5 float atomicAdd rows, 16 adjacent lanes per 64-byte row (+ 0.56 MB of indices)"),
}


def load(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = (r.get("Kernel_Name") or r.get("Kernel Name") or "").split("(")[0].replace("void ", "").strip()
            tot[name] += float(r["Counter_Value"]) * 1024.0  # KiB -> bytes
            cnt[name] += 1
    return {k: tot[k] / cnt[k] for k in tot}


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k, (rd, wr, lines, note) in KNOWN.items():
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        out[k] = dict(pattern=note, program_read_MB=round(rd / 1e6, 2), program_write_MB=round(wr / 1e6, 2), lines_touched_MB=round(lines / 1e6, 2),
                      FETCH_SIZE_MB=round(f / 1e6, 2), WRITE_SIZE_MB=round(w / 1e6, 2),
                      fetch_over_read=None if not rd else round(f / rd, 3), write_over_write=None if not wr else round(w / wr, 3),
                      write_over_lines=None if not wr else round(w / lines, 3))
        print(k, json.dumps(out[k]))
    wide = out["cal_fill16"]["write_over_write"] or 1.0
    factors = dict(
        write_wide_store=wide, write_narrow_store=out["cal_fill4"]["write_over_write"], fetch_wide_load=out["cal_read16"]["fetch_over_read"],
        write_bytes_per_scattered_4B_store=round(out["cal_store_per_line"]["WRITE_SIZE_MB"] * 1e6 / (1 << 20), 1),
        fetch_bytes_per_random_4B_load=round((out["cal_gather4"]["FETCH_SIZE_MB"] * 1e6 - 4.0 * (1 << 20) * (out["cal_read16"]["fetch_over_read"] or 0.5)) / (1 << 20), 1),
        write_bytes_per_atomic_min64=round(out["cal_atomic_min64"]["WRITE_SIZE_MB"] * 1e6 / 4e5, 1),
        fetch_bytes_per_atomic_min64=round((out["cal_atomic_min64"]["FETCH_SIZE_MB"] * 1e6) / 4e5, 1),
        write_bytes_per_coalesced_row_atomic=round(out["cal_atomic_rows"]["WRITE_SIZE_MB"] * 1e6 / 1.4e5, 1),
        fetch_bytes_per_coalesced_row_atomic=round(out["cal_atomic_rows"]["FETCH_SIZE_MB"] * 1e6 / 1.4e5, 1))
    json.dump(dict(note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on the known-byte-count kernels of tools/pmc_calib (gfx950, ROCm 7.2); "
                        "per dispatch, mean of 3", kernels=out, factors=factors), open(sys.argv[3], "w"), indent=1)
    print(json.dumps(factors, indent=1))


if __name__ == "__main__":
    main()
