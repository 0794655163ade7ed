"""Where the microseconds of each hot-path kernel go INSIDE the real step: per-phase durations over the work-groups of one launch.

    python 3danimals_amd/csrc/build.py --profile            # liba3d_hip_prof.so: the library with -DA3D_PROFILE (a3d_common.h: A3D_STAMP)
    python tools/kernel_phases.py [--only gb_bwd sh_bwd ..] [bench.py-style scene arguments: --grid-res, --batch, --workload]

The instrumented library is loaded instead of the product's (A3D_LIB); thread 0 of every work-group of the selected kernel stamps the
100 MHz wall clock at the phase boundaries the source marks (A3D_STAMP(kernel, slot)), one kernel at a time, during one training step of
the synthetic scene bench.py times.  Printed per kernel: work-groups that started, the spread of their start times, first start -> last
stamp (the launch minus its dispatch latency), and per phase the median / p90 / max duration over the work-groups that reached it.
"""
import argparse
import ctypes
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROF = os.path.join(ROOT, "3danimals_amd", "lib", "liba3d_hip_prof.so")
os.environ["A3D_LIB"] = PROF

import numpy as np  # noqa: E402
import torch  # noqa: E402

# name -> (translation unit, kernel id, {slot: label of the phase that ENDS at that slot})
KERNELS = {
    "dm_sign": ("dmtet", 0, {5: "whole kernel"}),
    "dm_count_cull": ("dmtet", 1, {1: "group fields of the sign plane (2 dependent loads) + ballots", 2: "barrier", 5: "blocks that hold crossings", 6: "acknowledgements + ticket (folded scan)", 7: "scan tail (the last work-group only)"}),
    "dm_emit": ("dmtet", 3, {1: "counts from the device + clear", 5: "a slab with surface: planes -> rows -> vertices / faces"}),
    "dm_bwd": ("dmtet", 4, {5: "whole kernel"}),
    "sk_fwd": ("skin", 0, {1: "links + bones into LDS", 2: "chain products", 3: "logits of the first group + barrier", 5: "softmax, blend, store"}),
    "sk_bwd": ("skin", 1, {1: "stage + weights + g_v (phase 1)", 2: "matrix phase", 3: "tiles to LDS + barrier", 4: "share of g_T + barrier", 5: "chain adjoint + atomics"}),
    "rs_tri": ("raster", 0, {1: "set-up: indices, vertices, box, prefix (-> barrier)", 2: "pooled fragment tests + atomics", 3: "barrier (big boxes listed)",
                             5: "tile stage of the big boxes"}),
    "rs_resolve": ("raster", 1, {5: "whole kernel"}),
    "rs_resolve_cover": ("raster", 3, {1: "keys, count published, winner's gathers, texel, look-up of the earlier counts", 5: "list entry + G-buffer row (uncovered: -1)"}),
    "gb_cover_fwd": ("gbuffer", 0, {1: "texel + block offset (-> barrier)", 5: "row: 3 gathers x 3 arrays, stores (uncovered: -1)"}),
    "gb_bwd": ("gbuffer", 1, {1: "loads + pixel adjoint", 2: "DPP merges", 3: "stage into LDS lists (-> barrier)", 4: "slot compaction (-> barrier)", 5: "list walks + row atomics"}),
    "sh_bwd": ("shade", 0, {1: "per-point adjoint + stores", 5: "per-image row reduction + atomics"}),
    "nr_face_bwd": ("normals", 0, {5: "whole kernel"}),
    "nr_sum_bwd": ("normals", 1, {5: "whole kernel"}),
    "ca_compose": ("antialias", 0, {1: "pixel -> source map into LDS (-> barrier; generic path)", 3: "movers of the second buffer: loads + stores",
                                    4: "analysis work-groups (whole)", 5: "movers of the first buffer (4 floats per pixel: whole; else loads + stores)"}),
    "ca_blend": ("antialias", 1, {1: "segment offsets", 5: "blends (atomics)"}),
    "ca_gather": ("antialias", 2, {1: "list -> LDS (-> barrier)", 5: "gather + store"}),
    "eb": ("bones", 0, {1: "A: centroids, spine ends, planar copy", 2: "B: first select (4 radix passes)", 3: "B: compaction + second select",
                        4: "C: feet of the quadrants", 5: "D: joints and bones"}),
    "ca_bwd": ("antialias", 3, {1: "segment offsets", 5: "records: colour adjoints + edge adjoints (atomics)"}),
}
MAX_WG = 65536


def report(name, st, labels):
    started = st[:, 0] > 0
    n = int(started.sum())
    if n == 0:
        print(f"{name}: no work-group stamped (kernel not launched in this step?)")
        return
    s = st[started].astype(np.int64)
    t0 = s[:, 0].min()
    last = s[:, :6].max()
    line = f"{name}: {n} work-groups, starts spread over {(s[:, 0].max() - t0) * 0.01:.2f} us, first start -> last stamp {(last - t0) * 0.01:.2f} us"
    if (s[:, 6] > 0).any() and (s[:, 7] > s[:, 6]).any():
        ok = (s[:, 7] > s[:, 6]) & (s[:, 5] > s[:, 0])
        mhz = (s[ok, 7] - s[ok, 6]) / ((s[ok, 5] - s[ok, 0]) * 0.01)
        line += f", shader clock {np.median(mhz):.0f} MHz"
    print(line)
    slots = sorted(labels)
    prev = 0
    for k in slots:
        # the previous stamp this work-group wrote (phases may be skipped by early exits)
        have = s[:, k] > 0
        if not have.any():
            print(f"    {labels[k]:62s} (no work-group reached it)")
            continue
        before = np.zeros(have.sum(), dtype=np.int64)
        for j in [x for x in [0] + slots if x < k]:
            cand = s[have, j]
            before = np.maximum(before, np.where(cand <= s[have, k], cand, 0))
        d = np.sort((s[have, k] - before) * 0.01)
        print(f"    {labels[k]:62s} median {d[len(d) // 2]:6.2f}  p90 {d[len(d) * 9 // 10]:6.2f}  max {d[-1]:6.2f} us   ({have.sum()} work-groups)")
        prev = k
    life = np.sort((s[:, :6].max(axis=1) - s[:, 0]) * 0.01)
    print(f"    {'work-group lifetime (first -> last stamp)':62s} median {life[len(life) // 2]:6.2f}  p90 {life[len(life) * 9 // 10]:6.2f}  p99 {life[len(life) * 99 // 100]:6.2f}  max {life[-1]:6.2f} us"
          f"   (sum over work-groups {life.sum():.0f} us)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="+", default=None, choices=sorted(KERNELS))
    ap.add_argument("--grid-res", type=int, default=64)
    ap.add_argument("--grid", default=None)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--workload", default="magicpony")
    ap.add_argument("--mesh", default="quadruped", choices=("quadruped", "spiky"))
    ap.add_argument("--steps", type=int, default=25, help="warm-up steps before the stamped ones (bench.py's default warm-up + steps)")
    args = ap.parse_args()
    if not os.path.exists(PROF):
        raise SystemExit(f"{PROF} is missing: python 3danimals_amd/csrc/build.py --profile")
    pipeline = importlib.import_module("3danimals_amd.pipeline")
    _lib = importlib.import_module("3danimals_amd._lib")
    handle = _lib.lib()
    dev = torch.device("cuda:0")
    scene = pipeline.SyntheticScene(grid_res=args.grid_res, batch=args.batch, resolution=(256, 256), device=dev, seed=0, workload=args.workload,
                                    deform=args.workload == "magicpony", grid=args.grid, mesh=args.mesh)
    for _ in range(args.steps):
        scene.step(backward=True, optimizer_step=True)
    torch.cuda.synchronize()
    buf = torch.zeros((MAX_WG, 8), dtype=torch.int64, device=dev)
    for name in args.only or KERNELS:
        tu, kid, labels = KERNELS[name]
        setter = getattr(handle, f"a3d_profile_set_{tu}")
        setter.restype, setter.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]
        buf.zero_()
        torch.cuda.synchronize()
        assert setter(buf.data_ptr(), kid) == 0
        scene.step(backward=True, optimizer_step=True)
        torch.cuda.synchronize()
        assert setter(None, -1) == 0
        report(name, buf.cpu().numpy(), labels)


if __name__ == "__main__":
    main()
