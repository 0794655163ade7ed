import sys, os, importlib, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pipeline = importlib.import_module("3danimals_amd.pipeline")
render = importlib.import_module("3danimals_amd.model.render.render")
dm = importlib.import_module("3danimals_amd.model.geometry.dmtet")
sk = importlib.import_module("3danimals_amd.model.geometry.skinning")
mesh = importlib.import_module("3danimals_amd.model.render.mesh")
ops = importlib.import_module("3danimals_amd.ops")
importlib.import_module("3danimals_amd.gemm_tuning").enable()
dev = torch.device("cuda:0")
scene = pipeline.SyntheticScene(grid_res=64, batch=16, resolution=(256, 256), device=dev, seed=0)
for _ in range(5): scene.step()
acc = {}
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t=time.perf_counter(); r=f(*a, **k); acc[label]=acc.get(label,0)+time.perf_counter()-t; return r
    setattr(obj, name, g)
wrap(scene.netShape, "getMesh", "getMesh(host incl. dmtet sync)")
wrap(ops, "dmtet_extract", "  dmtet_extract (count+sync+emit)")
wrap(scene.netShape, "_get_mesh_surface_backward", "  surface_backward total")
wrap(sk, "skinning", "skinning")
wrap(mesh, "make_mesh", "make_mesh (x2)")
wrap(render, "render_mesh", "render_mesh total")
wrap(render, "_covered_pixels", "  covered_pixels (sync)")
wrap(render, "_shade_points", "  shade_points (MLPs enqueue)")
wrap(scene, "losses", "losses")
wrap(scene.netShape, "get_sdf_gradient", "eikonal")
wrap(scene.optimizer, "step", "adam")
N=10
torch.cuda.synchronize(); t0=time.perf_counter()
tb=0
for _ in range(N):
    t=time.perf_counter()
    out = scene(jitter=True, sdf_reg=True)
    tf=time.perf_counter()-t
    scene.optimizer.zero_grad(set_to_none=True)
    t=time.perf_counter(); out["loss"].backward(); tb+=time.perf_counter()-t
    scene.optimizer.step()
    acc["forward total"]=acc.get("forward total",0)+tf
torch.cuda.synchronize(); wall=(time.perf_counter()-t0)/N*1e3
print(f"wall {wall:.2f} ms/step")
acc["backward (host)"]=tb
for k,v in acc.items(): print(f"{k:40s} {v/N*1e3:7.2f} ms")
