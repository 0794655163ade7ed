"""Locate the pixel where the HIP step and the oracle step disagree (run on the GPU box)."""
import importlib, os, sys, copy
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mesh_ref, render_ref, raster_ref
pipeline = importlib.import_module("3danimals_amd.pipeline")
ops = importlib.import_module("3danimals_amd.ops")
dev = torch.device("cuda:0")
scene = pipeline.SyntheticScene(grid_res=16, batch=3, resolution=(64, 64), device=dev, seed=0, net_width=32, net_layers=3, feat_dim=16, embedder_freq=4)
out = scene.step(backward=True, optimizer_step=False)
cpu = lambda t: t.detach().float().cpu()
prior, shape = scene.last["prior"], scene.last["shape"]
faces = prior.t_pos_idx[0].cpu()
posed = cpu(shape.v_pos)
clip = render_ref.xfm_points(posed, cpu(scene.mvp)).contiguous()
rast_o = raster_ref.rasterize(clip, faces.int(), (64, 64))
rast_h = cpu(ops.rasterize(clip.to(dev), faces.to(dev), (64, 64)))
print("rast ids equal:", bool((rast_o[..., 3] == rast_h[..., 3]).all()), "max |rast diff|", float((rast_o - rast_h).abs().max()))
print("last rast equal to recomputed:", bool((cpu(scene.last["rast"])[..., 3] == rast_h[..., 3]).all()))
col = torch.rand(3, 64, 64, 4)
aa_o = raster_ref.antialias(col, rast_o, clip, faces.int())
aa_h = cpu(ops.antialias(col.to(dev), rast_h.to(dev), clip.to(dev), faces.to(dev)))
d = (aa_o - aa_h).abs()
print("antialias max diff", float(d.max()), "at", np.unravel_index(int(d.argmax()), d.shape))
scene.netLight.light_params = None
tex, dino, lgt = (copy.deepcopy(m).cpu() for m in (scene.netTexture, scene.netDINO, scene.netLight))
with torch.no_grad():
    shaded, dino_pred = render_ref.render_mesh(posed, faces, mesh_ref.vertex_normals(posed, faces), cpu(scene.mvp), cpu(scene.w2c), cpu(scene.campos), tex, lgt,
                                               scene.resolution, background=cpu(scene.background), feat=cpu(scene.feat), render_modes=("shaded", "dino_pred"),
                                               prior_v_pos=cpu(prior.v_pos), dino_net=dino)
e1 = (shaded - cpu(out["shaded"])).abs(); e2 = (dino_pred - cpu(out["dino_pred"])).abs()
print("shaded err", float(e1.max()), np.unravel_index(int(e1.argmax()), e1.shape), "dino err", float(e2.max()), np.unravel_index(int(e2.argmax()), e2.shape))
b, c, y, x = np.unravel_index(int(e1.argmax()), e1.shape)
print("oracle", shaded[b, :, y, x], "hip", cpu(out["shaded"])[b, :, y, x])
print("rast around:", rast_h[b, y-1:y+2, x-1:x+2, 3])
print("normals diff", float((mesh_ref.vertex_normals(posed, faces) - cpu(shape.v_nrm)).abs().max()))
