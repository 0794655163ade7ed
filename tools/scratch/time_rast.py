import sys, importlib, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pipeline = importlib.import_module("3danimals_amd.pipeline")
ops = importlib.import_module("3danimals_amd.ops")
ru = importlib.import_module("3danimals_amd.model.render.renderutils")
dev = torch.device("cuda:0")
dbg = os.environ.pop("A3D_RAST_DBG", None)
scene = pipeline.SyntheticScene(grid_res=64, batch=16, resolution=(256, 256), device=dev, seed=0, net_width=32, net_layers=3, feat_dim=16, embedder_freq=4)
scene.step(backward=False)
prior, shape = scene.last["prior"], scene.last["shape"]
tri = prior.t_pos_idx[0]
clip = ru.xfm_points(shape.v_pos, scene.mvp).detach().contiguous()
if dbg: os.environ["A3D_RAST_DBG"] = dbg
for _ in range(30):
    r = ops.rasterize(clip, tri, (256, 256))
torch.cuda.synchronize()
print("done", float((r[..., 3] > 0).float().mean()))
