"""Where do the blocking device->host copies of one step come from?"""
import sys, importlib, traceback, collections
sys.path.insert(0, "/root/repo")
import torch
tuning = importlib.import_module("3danimals_amd.gemm_tuning")
pipeline = importlib.import_module("3danimals_amd.pipeline")
tuning.enable()
scene = pipeline.SyntheticScene(grid_res=64, batch=16, resolution=(256, 256), device="cuda", seed=0)
for _ in range(3):
    scene.step()
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
import warnings
hits = collections.Counter()
def showwarning(message, category, filename, lineno, file=None, line=None):
    st = traceback.extract_stack()
    frames = [f for f in st if "/root/repo/" in f.filename and "sync_sources" not in f.filename]
    key = " <- ".join(f"{f.filename.split('/root/repo/')[1]}:{f.lineno}" for f in frames[-3:][::-1])
    hits[key] += 1
warnings.showwarning = showwarning
warnings.simplefilter("always")
scene.step()
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("default")
for k, v in hits.most_common():
    print(v, k)
print("total", sum(hits.values()))
