import sys, importlib
sys.path.insert(0, "/root/repo")
import torch
from oracle import check
pipeline = importlib.import_module("3danimals_amd.pipeline")
scene = pipeline.SyntheticScene(grid_res=16, batch=3, resolution=(64, 64), device="cuda", seed=0, net_width=32, net_layers=3, feat_dim=16, embedder_freq=4)
out = scene.step(backward=True, optimizer_step=False)
print(check.compare_step(scene, out))
