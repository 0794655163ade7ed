#!/usr/bin/env python
"""Per-entry-point timing of the HIP path on the bench workload's own tensors (GPU box).

    python tools/microbench.py [--batch 16] [--grid-res 64] [--iters 50]

Builds the SyntheticScene, runs one step to obtain realistic inputs (mesh, clip positions, rast ...), then times
each op in isolation with HIP events (median over --iters), printing microseconds and algorithmic GB/s.  The backward rows go through
torch.autograd.grad, whose host-side dispatch (~100 us) is longer than the kernels: run it under `rocprofv3 --kernel-trace --stats` for
kernel times (e.g. rs_bwd_kernel 39 us, ip_bwd_kernel 52 us, gb_bwd_kernel 21 us), or see bench.py's per-entry-point events.
"""
import argparse
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        ts.append((a, b))
    torch.cuda.synchronize()
    v = sorted(x.elapsed_time(y) for x, y in ts)
    return v[len(v) // 2] * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--grid-res", type=int, default=64)
    ap.add_argument("--resolution", type=int, default=256)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--small-nets", action="store_true", default=True)
    args = ap.parse_args()
    pipeline = importlib.import_module("3danimals_amd.pipeline")
    ops = importlib.import_module("3danimals_amd.ops")
    ru = importlib.import_module("3danimals_amd.model.render.renderutils")
    sk = importlib.import_module("3danimals_amd.model.geometry.skinning")
    bench = importlib.import_module("bench")
    dev = torch.device("cuda:0")
    scene = pipeline.SyntheticScene(grid_res=args.grid_res, batch=args.batch, resolution=(args.resolution,) * 2, device=dev, seed=0, net_width=32,
                                    net_layers=3, feat_dim=16, embedder_freq=4)
    scene.step(backward=False)
    prior, shape = scene.last["prior"], scene.last["shape"]
    B, V, F, H, W = args.batch, prior.v_pos.shape[1], prior.t_pos_idx.shape[1], args.resolution, args.resolution
    geo = scene.netShape
    dims = dict(B=B, V=V, F=F, H=H, W=W, Nv=geo.verts.shape[0], Ne=geo.topology.edges32.shape[0], Nt=geo.topology.tets32.shape[0], K=20)
    print("workload:", dims)
    tri = prior.t_pos_idx[0]
    tri32 = ops.tri_int32(tri)
    clip = ru.xfm_points(shape.v_pos, scene.mvp).detach().contiguous()
    rast = ops.rasterize(clip, tri, (H, W))
    cover = float((rast[..., 3] > 0).float().mean())
    print(f"coverage {cover:.3f}")
    rows = []

    def add(name, fn):
        us = timeit(fn, args.iters)
        rows.append((name, us, None))

    sdf, pos = geo.current_sdf.detach(), geo.current_pos.detach()
    add("a3d_dmtet_count+emit", lambda: ops.dmtet(pos, sdf, geo.topology))
    add("a3d_rast_fwd", lambda: ops.rasterize(clip, tri, (H, W)))
    clip_g = clip.clone().requires_grad_(True)
    r = ops.rasterize(clip_g, tri, (H, W))
    g = torch.rand_like(r)
    add("a3d_rast_bwd", lambda: torch.autograd.grad(r, clip_g, g, retain_graph=True))
    vp = shape.v_pos.detach().clone().requires_grad_(True)
    add("a3d_interp_fwd[C3]", lambda: ops.interpolate(vp.detach(), rast, tri))
    o = ops.interpolate(vp, rast, tri)
    go = torch.rand_like(o)
    add("a3d_interp_bwd[C3]", lambda: torch.autograd.grad(o, vp, go, retain_graph=True))
    pv = prior.v_pos.detach().clone().requires_grad_(True)
    o2 = ops.interpolate(pv, rast, tri)
    add("a3d_interp_bwd[C3] (shared attr)", lambda: torch.autograd.grad(o2, pv, go, retain_graph=True))
    add("a3d_normals_fwd", lambda: ops.vertex_normals(vp.detach(), tri))
    n = ops.vertex_normals(vp, tri)
    gn = torch.rand_like(n)
    add("a3d_normals_bwd", lambda: torch.autograd.grad(n, vp, gn, retain_graph=True))
    topo = ops.aa_topology(tri32, V)
    add("a3d_aa_topology", lambda: ops.AATopology(tri32, V))
    add("a3d_aa_analyze", lambda: ops.AAAnalysis(rast, clip, topo))
    an = ops.AAAnalysis(rast, clip, topo)
    print("aa records:", int(an.count.sum().item()))
    for C in (4, 17):
        col = torch.rand(B, H, W, C, device=dev, requires_grad=True)
        add(f"a3d_aa_fwd[C{C}]", lambda: ops.antialias(col.detach(), rast, clip, tri, analysis=an))
        oc = ops.antialias(col, rast, clip_g, tri, analysis=an)
        gc = torch.rand_like(oc)
        add(f"a3d_aa_bwd[C{C}]", lambda: torch.autograd.grad(oc, [col, clip_g], gc, retain_graph=True))
    M = sk.bone_transforms_torch(scene.bones, scene.kinematic_tree, scene.arti.detach())
    T = M[:, :, :3, :].reshape(B, 20, 12).contiguous().requires_grad_(True)
    bones = scene.bones.reshape(1, 20, 2, 3)
    add("a3d_skin_fwd", lambda: ops.skin(pv.detach(), bones, T.detach(), 0.05))
    so = ops.skin(pv, bones, T, 0.05)
    gs = torch.rand_like(so)
    add("a3d_skin_bwd", lambda: torch.autograd.grad(so, [pv, T], gs, retain_graph=True))
    add("torch bone_transforms", lambda: sk.bone_transforms_torch(scene.bones, scene.kinematic_tree, scene.arti.detach()))
    chain = sk._chain_index32(scene.kinematic_tree, dev)
    if chain is not None:
        ang = scene.arti.detach().reshape(B, 20, 3).clone().requires_grad_(True)
        add("a3d_bone_transforms_fwd", lambda: ops.bone_transforms(bones.reshape(1, 20, 6), ang.detach(), chain))
        Mo = ops.bone_transforms(bones.reshape(1, 20, 6), ang, chain)
        gM = torch.rand_like(Mo)
        add("a3d_bone_transforms_bwd", lambda: torch.autograd.grad(Mo, ang, gM, retain_graph=True))
    add("a3d_mesh_topology", lambda: ops.mesh_topology(tri32, V))
    add("a3d_normals_adjacency", lambda: ops.VertexFaceAdjacency(tri32, V))
    pix = ops.covered_pixels(rast)
    dims["P"] = int(pix.shape[0])
    add("a3d_cover_count+emit", lambda: ops.covered_pixels(rast))
    nrm = ops.vertex_normals(shape.v_pos.detach(), tri)
    gin = [t.detach().clone().requires_grad_(True) for t in (clip, shape.v_pos, nrm, prior.v_pos)]
    add("a3d_gbuffer_fwd", lambda: ops.gbuffer(clip, shape.v_pos.detach(), nrm, prior.v_pos.detach(), rast, tri, pix))
    gb = ops.gbuffer(*gin, rast, tri, pix)
    ggb = torch.rand_like(gb)
    add("a3d_gbuffer_bwd", lambda: torch.autograd.grad(gb, gin, ggb, retain_graph=True))
    print(f"{'op':40s} {'us':>9s} {'alg MB':>9s} {'GB/s':>9s} {'% of 8TB/s':>10s}")
    for name, us, ab in rows:
        if "+" in name:
            ab = sum(bench.algorithmic_bytes(x, dims) for x in ("a3d_dmtet_count", "a3d_dmtet_emit"))
        base = name.split(" (")[0]
        ab = ab if ab is not None else bench.algorithmic_bytes(base, dims)
        if ab:
            print(f"{name:40s} {us:9.1f} {ab/1e6:9.2f} {ab/us/1e3:9.1f} {100*ab/us/1e3/8000:9.2f}%")
        else:
            print(f"{name:40s} {us:9.1f}")


if __name__ == "__main__":
    main()
