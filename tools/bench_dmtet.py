"""Event-timed a3d_dmtet_count / whole extraction (count + emit + read-back) on synthetic grids.

    python tools/bench_dmtet.py [--grid kuhn64 kuhn128 bcc51s bcc102s] [--iters 50] [--surf]

``kuhnR``: the spatially numbered Kuhn grid of R^3 cells; ``bccR``: the BCC lattice (Quartet's family) in its generator's numbering;
``bccRs``: the same with a random vertex numbering, shuffled rows and permuted row entries -- the numbering an external mesher's file
has (the reference's data/tets/{128,256}_tets.npz: bcc51s ~ the "128" class, bcc102s ~ the "256" class).
Every line says which count pass ran (``dmtet_pass``).
"""
import argparse
import importlib
import json
import os
import re
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_grid(name, tetgrid):
    return tetgrid.named_grid(name)[:2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", nargs="+", default=["kuhn64", "kuhn128"])
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--surf", action="store_true")
    ap.add_argument("--sdf", default="quadruped", choices=["quadruped", "ellipsoid", "none", "noise"])
    ap.add_argument("--passes", nargs="+", default=["plain", "auto"], help="plain: no static tables; ordered: the ranked lists forced; auto: what ops.dmtet_extract picks")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    a3d = importlib.import_module("3danimals_amd")
    ops = importlib.import_module("3danimals_amd.ops")
    dm = importlib.import_module("3danimals_amd.model.geometry.dmtet")
    syn = importlib.import_module("3danimals_amd.synthetic")
    dev = torch.device("cuda:0")
    records = []
    for name in args.grid:
        p, t = make_grid(name, a3d.tetgrid)
        pos, tets = torch.from_numpy(p).to(dev), torch.from_numpy(t).long().to(dev)
        scale = 7.0 / float((pos.amax(0) - pos.amin(0)).max())
        if args.sdf == "quadruped":
            sdf = syn.quadruped_sdf((pos * scale).cpu(), 0.2, noise=0.0)
        elif args.sdf == "ellipsoid":
            sdf = syn.ellipsoid_sdf((pos * scale).cpu(), noise=0.0)
        elif args.sdf == "none":
            sdf = -torch.ones(pos.shape[0])
        else:
            sdf = torch.randn(pos.shape[0])
        sdf = sdf.to(dev).contiguous().float()
        for label in args.passes:
            topo = dm.TetGridTopology(tets, positions=pos)
            if label == "plain":
                topo.WORD_GROUPS = topo.SPATIAL_ORDER = False
            elif label == "ordered":
                topo.WORD_GROUPS = False
            Ne, Nt, Nv = topo.edges32.shape[0], topo.tets32.shape[0], pos.shape[0]

            def extract():
                return ops.dmtet_extract(pos, sdf, topo, surface_vertices=args.surf, for_backward=args.surf)

            out = extract()
            which = topo.count_pass()
            rec = dict(grid=name, Nv=Nv, Ne=Ne, Nt=Nt, requested=label, dmtet_pass=which, V=int(out[0].shape[0]), F=int(out[1].shape[0]))
            for what, fn in (("count", lambda: ops.dmtet_count_only(pos, sdf, topo, surface_vertices=args.surf)), ("extract", extract)):
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                rec[what + "_us"] = round(1e3 * e0.elapsed_time(e1) / args.iters, 1)
            _lib = importlib.import_module("3danimals_amd._lib")
            with _lib.KernelTimer() as kt:  # per entry point, live HIP events (what bench.py reports)
                for _ in range(args.iters):
                    extract()
            rec["entry_us"] = {k: round(1e3 * ms, 1) for k, (n, ms) in kt.summary().items()}
            if (wr := topo.words_read(sdf)) is not None:
                rec["words_read"] = dict(edge=wr[0], edge_words=wr[1], tet=wr[2], tet_words=wr[3])
            print(json.dumps(rec), flush=True)
            records.append(rec)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(records, f, indent=1)


if __name__ == "__main__":
    main()
