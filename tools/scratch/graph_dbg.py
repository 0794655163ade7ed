import sys, importlib, time, faulthandler, os
faulthandler.enable()
sys.path.insert(0, "/root/repo")
import torch
tuning = importlib.import_module("3danimals_amd.gemm_tuning")
pipeline = importlib.import_module("3danimals_amd.pipeline")
dm = importlib.import_module("3danimals_amd.model.geometry.dmtet")
if os.environ.get("TUNE", "1") == "1":
    tuning.enable()
scene = pipeline.SyntheticScene(grid_res=32, batch=2, resolution=(64, 64), device="cuda", seed=0, net_width=64, net_layers=3)
geo = scene.netShape
geo.getMesh(jitter_grid=False)
pts = torch.rand(10000, 3, device="cuda") * 4 - 2
print("graphed call", flush=True)
out = geo._graphed_sdf_gradient(pts)
torch.cuda.synchronize()
print("captured + replayed fwd", flush=True)
out.sum().backward()
torch.cuda.synchronize()
print("replayed bwd ok", [float(p.grad.abs().max()) for p in geo.mlp.parameters() if p.grad is not None][:3], flush=True)
