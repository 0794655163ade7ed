import sys, os, importlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch.profiler import profile, ProfilerActivity, record_function
pipeline = importlib.import_module("3danimals_amd.pipeline")
render = importlib.import_module("3danimals_amd.model.render.render")
importlib.import_module("3danimals_amd.gemm_tuning").enable()
dev = torch.device("cuda:0")
scene = pipeline.SyntheticScene(grid_res=64, batch=16, resolution=(256, 256), device=dev, seed=0)
for _ in range(5): scene.step()
# wrap stages
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        with record_function(label):
            return f(*a, **k)
    setattr(obj, name, g)
wrap(scene.netShape, "getMesh", "S:getMesh(sdf+dmtet+normals)")
wrap(scene.netTexture, "sample", "S:texture_mlp")
wrap(scene.netDINO, "sample", "S:dino_mlp")
wrap(scene, "losses", "S:losses")
wrap(scene.netShape, "get_sdf_gradient", "S:eikonal")
wrap(scene.optimizer, "step", "S:adam")
wrap(render, "_shade_points", "S:shade_points(total)")
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3): scene.step()
    torch.cuda.synchronize()
ev = prof.key_averages()
rows = [(e.key, e.device_time_total/3e3, e.self_device_time_total/3e3, e.count/3) for e in ev if e.key.startswith("S:")]
for r in sorted(rows, key=lambda r:-r[1]): print(f"{r[0]:40s} device_total {r[1]:8.2f} ms/step  calls {r[3]:.0f}")
tot = sum(e.self_device_time_total for e in ev)/3e3
print("total device ms/step", round(tot,2))
top = sorted(ev, key=lambda e:-e.self_device_time_total)[:22]
for e in top: print(f"{e.key[:70]:70s} {e.self_device_time_total/3e3:8.2f} ms/step x{e.count/3:.0f}")
