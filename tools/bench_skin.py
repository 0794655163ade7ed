"""Isolated timing of skinning (fused pose vs separate launches), forward and backward, on bench-sized inputs."""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sk = importlib.import_module("3danimals_amd.model.geometry.skinning")
ops = importlib.import_module("3danimals_amd.ops")
syn = importlib.import_module("3danimals_amd.synthetic")
pipeline = importlib.import_module("3danimals_amd.pipeline")
dev = torch.device("cuda:0")
scene = pipeline.SyntheticScene(grid_res=int(os.environ.get("RES", "64")), batch=16, resolution=(64, 64), device=dev, seed=0, net_width=32, net_layers=3, feat_dim=16, embedder_freq=4)
scene.step(backward=False)
prior = scene.last["prior"]
v = prior.v_pos[None].detach().clone().requires_grad_(True)
arti = scene.arti.detach().clone().requires_grad_(True)
def run(fused, bwd):
    sk.FUSED_POSE = fused
    out, aux = sk.skinning(v, scene.bones, scene.kinematic_tree, arti, output_posed_bones=False, temperature=0.05)
    if bwd:
        out.backward(g)
g = torch.randn(16, 1, v.shape[2], 3, device=dev)
for fused in ((True,) if os.environ.get("FUSED_ONLY") else (False, True)):
    for bwd in (False, True):
        for _ in range(5): run(fused, bwd)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): run(fused, bwd)
        b.record(); torch.cuda.synchronize()
        print("fused" if fused else "split", "fwd+bwd" if bwd else "fwd", round(a.elapsed_time(b) / 50 * 1e3, 1), "us/iter (host-bound if > kernels)")
