import ctypes, importlib, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["A3D_LIB"] = os.path.join(ROOT, "3danimals_amd", "lib", "liba3d_hip_prof.so")
import kernel_phases as kp
pipeline = importlib.import_module("3danimals_amd.pipeline"); ops = importlib.import_module("3danimals_amd.ops"); _lib = importlib.import_module("3danimals_amd._lib")
ru = importlib.import_module("3danimals_amd.model.render.renderutils")
dev = torch.device("cuda:0")
scene = pipeline.SyntheticScene(grid_res=64, batch=16, resolution=(256, 256), device=dev, seed=0, net_width=32, net_layers=3, feat_dim=16, embedder_freq=4)
scene.step(backward=False)
prior, shape = scene.last["prior"], scene.last["shape"]
B, V, F, H, W = 16, prior.v_pos.shape[1], prior.t_pos_idx.shape[1], 256, 256
tri32 = ops.tri_int32(prior.t_pos_idx[0]); clip = ru.xfm_points(shape.v_pos, scene.mvp).detach().contiguous()
rast = ops.rasterize(clip, prior.t_pos_idx[0], (H, W)).contiguous(); g_rast = torch.rand_like(rast); g_clip = torch.empty_like(clip)
handle = _lib.lib(); setter = handle.a3d_profile_set_raster; setter.restype, setter.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]
buf = torch.zeros((kp.MAX_WG, 8), dtype=torch.int64, device=dev)
call = lambda: _lib.call("a3d_rast_bwd", ops.ptr(g_rast), ops.ptr(rast), ops.ptr(clip), B, ops.ptr(tri32), B, V, F, H, W, ops.ptr(g_clip), ops.stream())
for _ in range(3): call()
torch.cuda.synchronize(); assert setter(buf.data_ptr(), 4) == 0; call(); torch.cuda.synchronize(); assert setter(None, -1) == 0
kp.report("rs_bwd", buf.cpu().numpy(), {1: "loads + adjoint", 2: "merge", 3: "barrier + entries + slot + stage/link", 4: "barrier", 5: "flush"})

attr = shape.v_pos.detach().contiguous(); g_out = torch.rand(B, H, W, 3, device=dev); g_attr = torch.empty_like(attr); g_r = torch.empty_like(rast)
setter = handle.a3d_profile_set_interp; setter.restype, setter.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]
call = lambda: _lib.call("a3d_interp_bwd", ops.ptr(g_out), ops.ptr(attr), B, 3, ops.ptr(rast), ops.ptr(tri32), B, V, F, H, W, ops.ptr(g_attr), ops.ptr(g_r), ops.stream())
for _ in range(3): call()
buf.zero_(); torch.cuda.synchronize(); assert setter(buf.data_ptr(), 0) == 0; call(); torch.cuda.synchronize(); assert setter(None, -1) == 0
kp.report("ip_bwd C3", buf.cpu().numpy(), {1: "loads + rows", 2: "merge", 3: "barrier + entries + slot + stage/link", 4: "barrier", 5: "flush"})
