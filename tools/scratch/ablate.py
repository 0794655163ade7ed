import sys, importlib, time
sys.path.insert(0, "/root/repo")
import torch
tuning = importlib.import_module("3danimals_amd.gemm_tuning")
pipeline = importlib.import_module("3danimals_amd.pipeline")
tuning.enable()
scene = pipeline.SyntheticScene(grid_res=64, batch=16, resolution=(256, 256), device="cuda", seed=0)
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("full step            %.2f ms" % t(lambda: scene.step()))
print("no eikonal           %.2f ms" % t(lambda: scene.step(sdf_reg=False)))
print("no optimizer         %.2f ms" % t(lambda: scene.step(optimizer_step=False)))
print("forward only (nograd)%.2f ms" % t(lambda: scene.step(backward=False)))
with torch.no_grad():
    print("getMesh nograd       %.2f ms" % t(lambda: scene.netShape.getMesh(jitter_grid=True)))
def sdf_only():
    m = scene.netShape.getMesh(jitter_grid=True)
    m.v_pos.sum().backward()
print("getMesh + backward   %.2f ms" % t(sdf_only))
def eik():
    scene.netShape.getMesh(jitter_grid=True)
    ((scene.netShape.get_sdf_gradient().norm(dim=-1) - 1) ** 2).mean().backward()
print("getMesh + eikonal bwd %.2f ms" % t(eik))
