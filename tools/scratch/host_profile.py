"""Host-side view of one training step: which torch ops cost CPU time / launches (torch.profiler)."""
import sys, importlib, time
sys.path.insert(0, "/root/repo")
import torch
from torch.profiler import profile, ProfilerActivity, record_function
tuning = importlib.import_module("3danimals_amd.gemm_tuning")
pipeline = importlib.import_module("3danimals_amd.pipeline")
tuning.enable()
scene = pipeline.SyntheticScene(grid_res=64, batch=16, resolution=(256, 256), device="cuda", seed=0)
for _ in range(5):
    scene.step()
torch.cuda.synchronize()
# phase wall times with syncs
import collections
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        scene.step()
    torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.self_cpu_time_total)[:40]
tot_cpu = sum(e.self_cpu_time_total for e in ka) / 3e3
print("total self CPU ms/step %.2f" % tot_cpu)
for e in rows:
    print("%-60s n/step %6.1f  cpu ms/step %6.3f  cuda ms/step %6.3f" % (e.key[:60], e.count / 3, e.self_cpu_time_total / 3e3, e.self_device_time_total / 3e3))
