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
rows = [e for e in ka if e.key in ("aten::mm", "aten::addmm", "aten::bmm", "aten::relu_", "aten::relu", "aten::threshold_backward", "aten::cat", "aten::mul", "aten::sum", "aten::copy_", "aten::add", "aten::index_select", "aten::sin", "aten::cos")]
rows.sort(key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in ka) / 3e3
print("total device ms/step %.2f" % tot)
for e in rows[:45]:
    print("%-26s n/step %5.1f  dev us/call %8.1f  ms/step %6.3f  %s" % (e.key, e.count / 3, e.self_device_time_total / max(e.count, 1), e.self_device_time_total / 3e3, str(e.input_shapes)[:110]))
