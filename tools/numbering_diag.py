"""What the numbering of the tet grid costs the kernels AFTER DMTet: the same geometry in the generator's numbering, with the vertex
numbering scrambled, with the tet rows shuffled, with both (tetgrid.named_grid suffixes v / t / s) -- surface vertices and triangles
inherit the order of the grid's edges / tets, so a randomly numbered grid gives a randomly ordered mesh.

    python tools/numbering_diag.py [--grids kuhn64 kuhn64v kuhn64t kuhn64s] [--steps 10]
"""
import argparse
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grids", nargs="+", default=["kuhn64", "kuhn64v", "kuhn64t", "kuhn64s"])
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    pipeline = importlib.import_module("3danimals_amd.pipeline")
    ops = importlib.import_module("3danimals_amd.ops")
    L = importlib.import_module("3danimals_amd._lib")
    gt = importlib.import_module("3danimals_amd.gemm_tuning")
    gt.enable()
    seen = []
    init = ops.AAAnalysis.__init__

    def hooked(self, *a, **k):
        init(self, *a, **k)
        seen.append(self)

    ops.AAAnalysis.__init__ = hooked
    out = {}
    for g in args.grids:
        scene = pipeline.SyntheticScene(grid=g, batch=16, resolution=(256, 256), device="cuda", seed=0, workload="magicpony", deform=True)
        for _ in range(3):
            scene.step()
        seen.clear()
        with L.KernelTimer() as timer:
            for _ in range(args.steps):
                scene.step()
        t = {k: round(1e3 * ms, 1) for k, (n, ms) in timer.summary().items() if k.startswith(("a3d_dmtet", "a3d_skin", "a3d_rast", "a3d_cover", "a3d_gbuffer", "a3d_shade", "a3d_composite", "a3d_normals"))}
        prior = scene.last["prior"]
        rec = dict(V=int(prior.v_pos.shape[1]), F=int(prior.t_pos_idx.shape[1]), aa_records=int(seen[-1].count.sum()) if seen else None,
                   covered=int((scene.last["rast"][..., 3] > 0).sum()), in_scope_us=round(sum(t.values()), 1), entry_us=t)
        out[g] = rec
        print(g, json.dumps(rec), flush=True)
        del scene
    if args.json:
        json.dump(out, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
