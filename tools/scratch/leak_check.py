"""Does device memory stay flat over a few hundred training steps (caches, graphs, autograd references)?"""
import sys, importlib
sys.path.insert(0, "/root/repo")
import torch
tuning = importlib.import_module("3danimals_amd.gemm_tuning")
pipeline = importlib.import_module("3danimals_amd.pipeline")
tuning.enable()
scene = pipeline.SyntheticScene(grid_res=64, batch=16, resolution=(256, 256), device="cuda", seed=0)
for i in range(301):
    out = scene.step()
    if i % 50 == 0:
        torch.cuda.synchronize()
        print(i, "loss %.4f" % float(out["loss"]), "alloc %.1f MB" % (torch.cuda.memory_allocated() / 1e6), "reserved %.1f MB" % (torch.cuda.memory_reserved() / 1e6),
              "V", scene.last["prior"].v_pos.shape[1], flush=True)
