#!/usr/bin/env python
"""Tune the fp32 GEMMs of the path's MLPs with PyTorch TunableOp and write 3danimals_amd/tunableop/gfx950_fp32.csv (GPU box).

    python tools/tune_gemms.py [--lo 163840 --hi 245760]

Sweeps the point-count buckets the render path can produce (multiples of render.POINT_BUCKET) through the texture and DINO
fields (forward + backward), the SDF field over the whole grid (forward) and over the surface buckets (forward + backward,
and the eikonal double backward), at the network sizes of config/model/magicpony.yaml.
"""
import argparse
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lo", type=int, default=163840)
    ap.add_argument("--hi", type=int, default=245760)
    ap.add_argument("--grid-res", type=int, default=64)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    tuning = importlib.import_module("3danimals_amd.gemm_tuning")
    pipeline = importlib.import_module("3danimals_amd.pipeline")
    render = importlib.import_module("3danimals_amd.model.render.render")
    dmtet = importlib.import_module("3danimals_amd.model.geometry.dmtet")
    out = args.out or tuning.TUNED_FILE
    if os.path.exists(out):
        os.remove(out)
    tuning.enable(tuning=True, filename=out)
    torch.cuda.tunable.set_max_tuning_duration(30)
    torch.cuda.tunable.set_max_tuning_iterations(20)
    dev = torch.device("cuda:0")
    scene = pipeline.SyntheticScene(grid_res=args.grid_res, batch=16, resolution=(256, 256), device=dev, seed=0)
    t0 = time.time()
    feat = scene.feat.detach()
    for m in range(args.lo, args.hi + 1, render.POINT_BUCKET):
        x = torch.rand(m, 3, device=dev, requires_grad=True)
        idx = torch.randint(0, feat.shape[0], (m,), device=dev).sort().values
        f = feat.clone().requires_grad_(True)  # per-image rows + point->image index, as render._shade_points passes them
        (scene.netTexture.sample(x, feat=f, feat_index=idx).sum() + scene.netDINO.sample(x).sum()).backward()
        print(f"bucket {m}: {time.time() - t0:.0f}s", flush=True)
    geo = scene.netShape
    with torch.no_grad():
        geo.get_sdf(geo.verts)
    for n in range(dmtet.SURFACE_BUCKET, 16 * dmtet.SURFACE_BUCKET + 1, dmtet.SURFACE_BUCKET):
        geo.get_sdf(torch.rand(n, 3, device=dev)).sum().backward()
    geo.mesh_verts = torch.rand(6000, 3, device=dev)
    ((geo.get_sdf_gradient().norm(dim=-1) - 1) ** 2).mean().backward()
    for _ in range(3):  # whatever else the real step multiplies (light MLP, camera transforms ...)
        scene.step()
    torch.cuda.synchronize()
    n = sum(1 for _ in open(out)) if os.path.exists(out) else 0
    print("wrote", out, n, "lines in", f"{time.time() - t0:.0f}s")


if __name__ == "__main__":
    main()
