import sys, importlib, time
sys.path.insert(0, "/root/repo")
import torch
tuning = importlib.import_module("3danimals_amd.gemm_tuning")
pipeline = importlib.import_module("3danimals_amd.pipeline")
dm = importlib.import_module("3danimals_amd.model.geometry.dmtet")
tuning.enable()
scene = pipeline.SyntheticScene(grid_res=64, batch=16, resolution=(256, 256), device="cuda", seed=0)
geo = scene.netShape
geo.getMesh(jitter_grid=False)
# numerics: same points through both paths
pts = torch.rand(10000, 3, device="cuda") * 4 - 2
def loss_of(G): return ((G.norm(dim=-1) - 1) ** 2).mean()
def eager(p):
    p = p.clone().requires_grad_(True)
    y = geo.get_sdf(pts=p)
    return torch.autograd.grad([y], p, grad_outputs=torch.ones_like(y), create_graph=True)[0]
params = [q for q in geo.mlp.parameters()]
Ge = eager(pts); ge = torch.autograd.grad(loss_of(Ge), params, allow_unused=True)
Gg = geo._graphed_sdf_gradient(pts); gg = torch.autograd.grad(loss_of(Gg), params, allow_unused=True)
print("G maxdiff", (Ge - Gg).abs().max().item())
for (n, _), a, b in zip(geo.mlp.named_parameters(), ge, gg):
    if a is None or b is None:
        print(n, "None:", a is None, b is None); continue
    print(n, "grad maxdiff %.3e of %.3e" % ((a - b).abs().max().item(), a.abs().max().item()))
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def eik():
    ((geo.get_sdf_gradient().norm(dim=-1) - 1) ** 2).mean().backward()
dm.GRAPH_SDF_GRADIENT = True
print("eikonal graphed %.3f ms" % t(eik))
dm.GRAPH_SDF_GRADIENT = False
print("eikonal eager   %.3f ms" % t(eik))
dm.GRAPH_SDF_GRADIENT = True
print("full step graphed %.2f ms" % t(lambda: scene.step()))
dm.GRAPH_SDF_GRADIENT = False
print("full step eager   %.2f ms" % t(lambda: scene.step()))
