"""What the synthetic training does to the mesh over many steps (why late steps cost more):  python tools/long_run_diag.py [steps]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
p = importlib.import_module("3danimals_amd.pipeline")
_lib = importlib.import_module("3danimals_amd._lib")
s = p.SyntheticScene(grid_res=64, batch=16, resolution=(256, 256), device=torch.device("cuda:0"), seed=0, workload="magicpony", deform=True)
for i in range(steps + 1):
    timed = i % 100 == 0
    if timed:
        with _lib.KernelTimer() as t:
            s.step(backward=True, optimizer_step=True)
        torch.cuda.synchronize()
        summ = t.summary()
        rast = s.last["rast"]
        pts = s.last["points"]
        clip = pts["clip"]
        tri = s.last["shape"].t_pos_idx[0]
        ndc = clip[..., :2] / clip[..., 3:].clamp(min=1e-6)
        px = (ndc * 0.5 + 0.5) * 256
        c = px[:, tri]  # [B,F,3,2]
        ext = (c.amax(2) - c.amin(2))
        area = (ext[..., 0].clamp(0, 256) + 1) * (ext[..., 1].clamp(0, 256) + 1)
        print(i, "V", clip.shape[1], "F", tri.shape[0], "covered", int((rast[..., 3] > 0).sum()), "min w", float(clip[..., 3].min()),
              "box px: mean", round(float(area.mean()), 1), "max", round(float(area.max())), "sum/1e6", round(float(area.sum()) / 1e6, 2),
              "| rast us", [round(1e3 * v[1], 1) for k, v in summ.items() if k.startswith("a3d_rast_fwd")], "ndc max", round(float(ndc.abs().max()), 2), flush=True)
    else:
        s.step(backward=True, optimizer_step=True)
