"""Pixel-box statistics of the trained-like mesh (bench.py --mesh spiky) for a few parameter sets of pipeline.SPIKES, next to the step-600
statistics of the real long run it stands for (profiles/r04_long_run_diag.txt):  python tools/spiky_diag.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
p = importlib.import_module("3danimals_amd.pipeline")
_lib = importlib.import_module("3danimals_amd._lib")
SETS = [None, {}]  # the quadruped, the shipped parameters
if len(sys.argv) > 1:  # e.g.  tau=0.93,length2=0.4  omega2=3,tau2=0.6
    SETS = [dict((kv.split("=")[0], float(kv.split("=")[1])) for kv in a.split(",")) for a in sys.argv[1:]]
for q in SETS:
    s = p.SyntheticScene(grid_res=64, batch=16, resolution=(256, 256), device=torch.device("cuda:0"), seed=0, workload="magicpony", deform=True,
                         mesh="quadruped" if q is None else "spiky", spikes=q)
    for _ in range(3):
        s.step(backward=True, optimizer_step=False)
    with _lib.KernelTimer() as t:
        for _ in range(5):
            s.step(backward=True, optimizer_step=False)
    torch.cuda.synchronize()
    summ = t.summary()
    rast, clip, tri = s.last["rast"], s.last["points"]["clip"], s.last["shape"].t_pos_idx[0]
    ndc = clip[..., :2] / clip[..., 3:].clamp(min=1e-6)
    c = ((ndc * 0.5 + 0.5) * 256)[:, tri]
    ext = c.amax(2) - c.amin(2)
    area = (ext[..., 0].clamp(0, 256) + 1) * (ext[..., 1].clamp(0, 256) + 1)
    us = lambda pre: round(sum(1e3 * v[1] for k, v in summ.items() if k.startswith(pre)), 1)
    scope = round(sum(1e3 * v[1] * v[0] / 5 for k, v in summ.items() if k.startswith(("a3d_dmtet_", "a3d_skin_", "a3d_normals_", "a3d_rast_", "a3d_cover_", "a3d_gbuffer_", "a3d_shade_", "a3d_composite_aa_"))), 1)
    print(q, "| V", clip.shape[1], "covered", int((rast[..., 3] > 0).sum()), "box px: mean", round(float(area.mean()), 1), "max", round(float(area.max())),
          "sum/1e6", round(float(area.sum()) / 1e6, 2), "frac>64", round(float((area > 64).float().mean()), 3), "frac>512", round(float((area > 512).float().mean()), 4),
          "| us: rast_fwd", us("a3d_rast_fwd"), "ca_fwd", us("a3d_composite_aa_fwd"), "ca_bwd", us("a3d_composite_aa_bwd"), "gb_bwd", us("a3d_gbuffer_bwd"), "in-scope", scope, flush=True)
    del s
