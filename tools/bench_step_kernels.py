"""Kernel times of the in-scope kernels inside the bench step (run under rocprofv3 --kernel-trace --stats; A3D_EXP is set AFTER the
warm-up steps, for libraries built with a debug stage knob).

    RES=64 STEPS=20 python tools/bench_step_kernels.py
"""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if os.environ.get("DBG"):  # the measurement knobs live in the experiment build only (csrc/build.py --exp)
    os.environ.setdefault("A3D_LIB", os.path.join(ROOT, "3danimals_amd", "lib", "liba3d_hip_exp.so"))
pipeline = importlib.import_module("3danimals_amd.pipeline")
dev = torch.device("cuda:0")
scene = pipeline.SyntheticScene(grid_res=int(os.environ.get("RES", "64")), batch=16, resolution=(256, 256), device=dev, seed=0, workload=os.environ.get("WORKLOAD", "magicpony"), deform=os.environ.get("WORKLOAD", "magicpony") == "magicpony",
                                mesh=os.environ.get("MESH", "quadruped"))
for _ in range(3):
    scene.step(backward=True, optimizer_step=True)
torch.cuda.synchronize()
os.environ["A3D_EXP"] = os.environ.get("DBG", "0")
for _ in range(int(os.environ.get("STEPS", "20"))):
    scene.step(backward=True, optimizer_step=False)
torch.cuda.synchronize()
print("done")
