import sys, importlib
sys.path.insert(0, "/root/repo")
import torch
from torch.profiler import profile, ProfilerActivity
tuning = importlib.import_module("3danimals_amd.gemm_tuning")
pipeline = importlib.import_module("3danimals_amd.pipeline")
tuning.enable()
scene = pipeline.SyntheticScene(grid_res=64, batch=16, resolution=(256, 256), device="cuda", seed=0)
for _ in range(5):
    scene.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3):
        scene.step()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
skip = ("aten::mm", "aten::bmm", "aten::addmm", "aten::_addmm_activation")
rows = [e for e in ka if e.self_device_time_total > 0 and e.key not in skip]
rows = [e for e in rows if e.key.startswith("aten::") or e.key.startswith("_") or "Backward" in e.key]
rows.sort(key=lambda e: -e.count)
print("non-GEMM device ms/step %.2f in %.0f launches" % (sum(e.self_device_time_total for e in rows) / 3e3, sum(e.count for e in rows) / 3))
for e in rows[:60]:
    print("%-34s n/step %5.1f  us/call %7.1f  ms/step %6.3f  %s" % (e.key[:34], e.count / 3, e.self_device_time_total / max(e.count, 1), e.self_device_time_total / 3e3, str(e.input_shapes)[:100]))
