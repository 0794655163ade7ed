"""Isolated timing of the rasteriser on the bench-sized mesh (kernel times: run under rocprofv3 --kernel-trace --stats).

    RES=64 python tools/bench_rast.py
"""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if os.environ.get("RAST_DBG"):  # the measurement knobs live in the experiment build only (csrc/build.py --exp)
    os.environ.setdefault("A3D_LIB", os.path.join(ROOT, "3danimals_amd", "lib", "liba3d_hip_exp.so"))
ops = importlib.import_module("3danimals_amd.ops")
pipeline = importlib.import_module("3danimals_amd.pipeline")
render = importlib.import_module("3danimals_amd.model.render.render")
dev = torch.device("cuda:0")
scene = pipeline.SyntheticScene(grid_res=int(os.environ.get("RES", "64")), batch=16, resolution=(256, 256), device=dev, seed=0, net_width=32, net_layers=3,
                                feat_dim=16, embedder_freq=4, mesh=os.environ.get("MESH", "quadruped"))
scene.step(backward=False)
clip = scene.last["points"]["clip"].contiguous()
tri = scene.last["prior"].t_pos_idx[0]
print("clip", tuple(clip.shape), "faces", tri.shape[0], flush=True)
os.environ["A3D_EXP"] = os.environ.get("RAST_DBG", "0")  # (debug stage knob, if the library was built with one)
for _ in range(5):
    ops.rasterize(clip, tri, (256, 256))
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50):
    rast = ops.rasterize(clip, tri, (256, 256))
b.record(); torch.cuda.synchronize()
print("RAST_DBG", os.environ.get("RAST_DBG", "0"), "MESH", os.environ.get("MESH", "quadruped"), "RES", os.environ.get("RES", "64"), "rasterize", round(a.elapsed_time(b) / 50 * 1e3, 1), "us/iter; covered",
      float((rast[..., 3] > 0).float().mean()), "checksum", float(rast[..., 3].double().sum()))
