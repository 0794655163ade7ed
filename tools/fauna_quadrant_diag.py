"""How close the Fauna step's bone estimation comes to an EMPTY leg quadrant over a long run (estimate_bones guards the quadrants with a
device-side assert -- skinning.py: foot_of -- and a failed device assert ends the process as an "HSA hardware exception" on ROCm:
tools/assert_async_probe.py).  python tools/fauna_quadrant_diag.py [steps] [seed]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
p = importlib.import_module("3danimals_amd.pipeline")
sk = importlib.import_module("3danimals_amd.model.geometry.skinning")
s = p.SyntheticScene(grid_res=64, batch=16, resolution=(256, 256), device=torch.device("cuda:0"), seed=seed, workload="fauna")
sk.QUADRANT_POPULATION = []
lowest = None
for i in range(steps + 1):
    s.step(backward=True, optimizer_step=True)
    if i % 50 == 0 or i == steps:
        pop = torch.stack(sk.QUADRANT_POPULATION).cpu()  # [calls, 4]
        sk.QUADRANT_POPULATION.clear()
        lo = pop.min(0).values
        lowest = lo if lowest is None else torch.minimum(lowest, lo)
        print(f"step {i}: V {s.last['prior'].v_pos.shape[1]}  vertices per leg quadrant, min over the last {pop.shape[0]} estimates: {lo.tolist()}  (run minimum {lowest.tolist()})", flush=True)
