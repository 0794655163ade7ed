import sys, importlib
sys.path.insert(0, "/root/repo")
import torch
pipeline = importlib.import_module("3danimals_amd.pipeline")
render = importlib.import_module("3danimals_amd.model.render.render")
for batch in (16,):
    for rank in range(8):
        scene = pipeline.SyntheticScene(grid_res=64, batch=batch, resolution=(256, 256), device="cuda", seed=1000 * rank)
        out = scene.step(backward=False)
        rast = scene.last["rast"]
        P = int((rast[..., 3] > 0).sum())
        b = render.POINT_BUCKET
        print("rank", rank, "P", P, "bucket", -(-P // b) * b, "V", scene.last["prior"].v_pos.shape, flush=True)
        del scene
