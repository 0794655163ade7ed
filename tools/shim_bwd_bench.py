"""Raw C-ABI timing of a3d_rast_bwd / a3d_interp_bwd on the bench workload's tensors (events over back-to-back calls)."""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pipeline = importlib.import_module("3danimals_amd.pipeline")
ops = importlib.import_module("3danimals_amd.ops")
_lib = importlib.import_module("3danimals_amd._lib")
ru = importlib.import_module("3danimals_amd.model.render.renderutils")
dev = torch.device("cuda:0")
mesh = os.environ.get("MESH", "quadruped")
scene = pipeline.SyntheticScene(grid_res=64, batch=16, resolution=(256, 256), device=dev, seed=0, net_width=32, net_layers=3, feat_dim=16, embedder_freq=4, mesh=mesh)
scene.step(backward=False)
prior, shape = scene.last["prior"], scene.last["shape"]
B, V, F, H, W = 16, prior.v_pos.shape[1], prior.t_pos_idx.shape[1], 256, 256
tri32 = ops.tri_int32(prior.t_pos_idx[0])
clip = ru.xfm_points(shape.v_pos, scene.mvp).detach().contiguous()
rast = ops.rasterize(clip, prior.t_pos_idx[0], (H, W)).contiguous()
g_rast = torch.rand_like(rast)
g_clip = torch.empty_like(clip)
p, st = ops.ptr, ops.stream
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print("covered", int((rast[..., 3] > 0).sum()), "V", V, "F", F)
print("rast_bwd us/call", round(t(lambda: _lib.call("a3d_rast_bwd", p(g_rast), p(rast), p(clip), B, p(tri32), B, V, F, H, W, p(g_clip), st())), 2))
for C, attr in ((3, shape.v_pos.detach().contiguous()), (3, prior.v_pos.detach().contiguous()), (16, torch.rand(B, V, 16, device=dev))):
    ab = attr.shape[0]
    g_out = torch.rand(B, H, W, C, device=dev)
    g_attr = torch.empty_like(attr)
    g_r = torch.empty_like(rast)
    print(f"interp_bwd C{C} attr_batch {ab} us/call", round(t(lambda: _lib.call("a3d_interp_bwd", p(g_out), p(attr), ab, C, p(rast), p(tri32), B, V, F, H, W, p(g_attr), p(g_r), st())), 2))
