"""Oracle: BASELINE config 1 -- the geometry-only path (no raster) on CPU.  TEST INFRASTRUCTURE ONLY.

"DMTet marching-tets on 32-res tet grid + LBS skinning, random SDF/bones, PyTorch CPU (plumbing, no raster)" (BASELINE.json
configs[0]; inputs as in BASELINE.md section 2/3): Kuhn R=32 grid at scale 7, sdf = ellipsoid init + N(0, 0.01^2) (seed 0),
estimate_bones(8 body bones, 4 legs x 3, 'z_minmax_y+'), articulation ~ U(-0.25, 0.25) rad [16,1,20,3], temperature 0.05;
step = DMTet -> vertex normals (B=1) -> bones -> skinning (B=16) -> vertex normals (B=16) -> backward to sdf and angles.
The survey measured the reference's own code on this workload at 208.8 ms/step (8 threads, build container).
"""
import time
from importlib import import_module

import torch

from . import dmtet_ref, mesh_ref, skinning_ref


def make_inputs(res=32, batch=16, seed=0, scale=7.0):
    a3d = import_module("3danimals_amd")
    v, t = a3d.tetgrid.kuhn_grid(res)
    pos = torch.from_numpy(v) * scale
    return dict(pos=pos, tets=torch.from_numpy(t), sdf=a3d.synthetic.ellipsoid_sdf(pos, scale, 0.01, seed=seed),
                arti=a3d.synthetic.seeded((batch, 1, 20, 3), seed + 3, -0.25, 0.25), temperature=0.05,
                w_verts=None, batch=batch)


def cpu_step(inp, backward=True):
    """One forward(+backward) of the geometry path on CPU with the oracle; returns dict(seconds, V, F, loss)."""
    sk = import_module("3danimals_amd.model.geometry.skinning")  # estimate_bones: device-agnostic torch host logic
    t0 = time.perf_counter()
    sdf = inp["sdf"].clone().requires_grad_(backward)
    arti = inp["arti"].clone().requires_grad_(backward)
    verts, faces, _, _ = dmtet_ref.marching_tets(inp["pos"], sdf, inp["tets"])
    n1 = mesh_ref.vertex_normals(verts[None], faces)
    bones, tree, _ = sk.estimate_bones(verts[None, None].detach(), n_body_bones=8, n_legs=4, n_leg_bones=3, body_bones_mode="z_minmax_y+")
    posed, _ = skinning_ref.skinning(verts[None, None], bones, tree, arti, inp["temperature"])
    posed = posed.view(inp["batch"], -1, 3)
    n2 = mesh_ref.vertex_normals(posed, faces)
    loss = (posed ** 2).mean() + (n2[..., 1]).mean() + (n1[..., 2]).mean()
    if backward:
        loss.backward()
    return dict(seconds=time.perf_counter() - t0, V=int(verts.shape[0]), F=int(faces.shape[0]), loss=float(loss.detach()),
                grad_sdf=sdf.grad, grad_arti=arti.grad)
