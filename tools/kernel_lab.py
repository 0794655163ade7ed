#!/usr/bin/env python
"""Kernel bisection on the bench workload's own tensors (GPU box):  python tools/kernel_lab.py gbuffer_bwd 0 1 2 3

For each value of the A3D_EXP knob (read by the C side on every call) the named entry point is enqueued 20 times between two
HIP events, 5 rounds, best average reported.  Knobs are temporary instrumentation inside the kernels under study.
"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def best_us(fn, reps=20, rounds=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / reps)
    return best


def main():
    what, knobs = sys.argv[1], [int(x) for x in sys.argv[2:]] or [0]
    pipeline = importlib.import_module("3danimals_amd.pipeline")
    ops = importlib.import_module("3danimals_amd.ops")
    L = importlib.import_module("3danimals_amd._lib")
    ru = importlib.import_module("3danimals_amd.model.render.renderutils")
    dev = torch.device("cuda:0")
    scene = pipeline.SyntheticScene(grid_res=64, batch=16, resolution=(256, 256), device=dev, seed=0, net_width=32, net_layers=3, feat_dim=16,
                                    embedder_freq=4, mesh=os.environ.get("A3D_LAB_MESH", "quadruped"))
    scene.step(backward=False)
    prior, shape = scene.last["prior"], scene.last["shape"]
    B, V, F, H, W = 16, prior.v_pos.shape[1], prior.t_pos_idx.shape[1], 256, 256
    tri = prior.t_pos_idx[0]
    tri32 = ops.tri_int32(tri)
    clip = ru.xfm_points(shape.v_pos, scene.mvp).detach().contiguous()
    rast = ops.rasterize(clip, tri, (H, W)).detach()
    pix = ops.covered_pixels(rast)
    P = pix.shape[0]
    nrm = ops.vertex_normals(shape.v_pos.detach(), tri)
    vpos, pv = shape.v_pos.detach().contiguous(), prior.v_pos.detach().contiguous()
    print(f"V={V} F={F} P={P}")
    ptr, stream = L.ptr, L.stream
    if what == "gbuffer_bwd":
        g = torch.rand(P, 12, device=dev)
        rows = torch.empty(B * V, 16, device=dev)
        fn = lambda: L.call("a3d_gbuffer_bwd", ptr(g), ptr(rast), ptr(tri32), ptr(pix), P, ptr(vpos), ptr(nrm), ptr(pv), 1, ptr(clip), B, V, F, H, W,
                            ptr(rows), 0, 1, None, 0, None, None, stream())
    elif what == "rast_fwd":
        out = torch.empty(B, H, W, 4, device=dev)
        scratch = torch.empty(L.lib().a3d_rast_scratch_bytes(B, H, W), dtype=torch.uint8, device=dev)
        fn = lambda: L.call("a3d_rast_fwd", ptr(clip), B, ptr(tri32), B, V, F, H, W, ptr(out), ptr(scratch), 0, None, stream())
    elif what == "aa_analyze":
        topo = ops.aa_topology(tri32, V)
        cap = L.lib().a3d_aa_capacity(B, H, W)
        work = torch.empty(cap, 4, dtype=torch.int32, device=dev)
        count = torch.empty(L.lib().a3d_aa_shards(), dtype=torch.int32, device=dev)
        screen = torch.empty(B, V, 2, device=dev)
        fn = lambda: L.call("a3d_aa_analyze", ptr(rast), ptr(clip), B, ptr(tri32), ptr(topo.opp), B, V, F, H, W, ptr(screen), ptr(work),
                            cap, ptr(count), 0, stream())
    else:
        raise SystemExit("unknown entry point")
    for k in knobs:
        os.environ["A3D_EXP"] = str(k)
        print(f"{what} A3D_EXP={k}: {best_us(fn):8.1f} us")


if __name__ == "__main__":
    main()
