"""Stage-wise parity report of one workload step at any size (oracle/check.compare_step), before and after a number of Adam steps.

  python tools/parity_diag.py --workload fauna --steps 0 35 --n 4
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="magicpony")
    ap.add_argument("--grid-res", type=int, default=64)
    ap.add_argument("--resolution", type=int, default=256)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--steps", type=int, nargs="+", default=[0])
    ap.add_argument("--n", type=int, default=4, help="frames the oracle re-does")
    ap.add_argument("--tuned", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    args = ap.parse_args()
    from oracle import check

    pipeline = importlib.import_module("3danimals_amd.pipeline")
    if args.tuned:
        importlib.import_module("3danimals_amd.gemm_tuning").enable()
    batch = args.batch if args.batch is not None else (8 if args.workload == "ponymation" else 16)
    frames = args.frames if args.workload == "ponymation" else 1
    scene = pipeline.SyntheticScene(grid_res=args.grid_res, batch=batch, resolution=(args.resolution, args.resolution), device="cuda", seed=0,
                                    workload=args.workload, num_frames=frames, deform=(args.workload == "magicpony"))
    done = 0
    for target in sorted(args.steps):
        while done < target:
            scene.step()
            done += 1
        out = scene.step(backward=True, optimizer_step=False)
        torch.cuda.synchronize()
        t = time.perf_counter()
        rep = check.compare_step(scene, out, n_images=args.n, end_to_end=not args.no_e2e)
        rep["after_steps"], rep["checker_seconds"], rep["pass"] = done, round(time.perf_counter() - t, 1), check.passes(rep)
        print(json.dumps(rep), flush=True)


if __name__ == "__main__":
    main()
