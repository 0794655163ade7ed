"""Caching-allocator footprint of the training step over many iterations:  python tools/mem_growth.py [steps] [every]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
every = int(sys.argv[2]) if len(sys.argv) > 2 else 100
p = importlib.import_module("3danimals_amd.pipeline")
s = p.SyntheticScene(grid_res=64, batch=16, resolution=(256, 256), device=torch.device("cuda:0"), seed=0, workload="magicpony", deform=True)
for i in range(steps + 1):
    s.step(backward=True, optimizer_step=True)
    if i % every == 0:
        torch.cuda.synchronize()
        st = torch.cuda.memory_stats()
        print(i, "allocated MB", round(torch.cuda.memory_allocated() / 1e6), "peak", round(torch.cuda.max_memory_allocated() / 1e6), "reserved MB",
              round(torch.cuda.memory_reserved() / 1e6), "segments", st.get("segment.all.current"), "inactive split MB",
              round(st.get("inactive_split_bytes.all.current", 0) / 1e6), flush=True)
