"""Event-timed a3d_dmtet_count / a3d_dmtet_emit on the synthetic grids: culled against plain count pass.

    python tools/bench_dmtet.py [--res 64 128] [--iters 50]
"""
import argparse
import importlib
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, nargs="+", default=[64, 128])
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--surf", action="store_true")
    ap.add_argument("--sdf", default="quadruped", choices=["quadruped", "ellipsoid", "none", "noise"])
    args = ap.parse_args()
    a3d = importlib.import_module("3danimals_amd")
    ops = importlib.import_module("3danimals_amd.ops")
    dm = importlib.import_module("3danimals_amd.model.geometry.dmtet")
    syn = importlib.import_module("3danimals_amd.synthetic")
    _lib = importlib.import_module("3danimals_amd._lib")
    dev = torch.device("cuda:0")
    ops.DMTET_CULL_MIN_VERTS = 0
    for res in args.res:
        p, t = a3d.tetgrid.kuhn_grid(res)
        pos, tets = torch.from_numpy(p).to(dev), torch.from_numpy(t).long().to(dev)
        topo = dm.TetGridTopology(tets)
        scale = 7.0 / float((pos.amax(0) - pos.amin(0)).max())
        if args.sdf == "quadruped":
            sdf = syn.quadruped_sdf((pos * scale).cpu(), 0.2, noise=0.0)
        elif args.sdf == "ellipsoid":
            sdf = syn.ellipsoid_sdf((pos * scale).cpu(), noise=0.0)
        elif args.sdf == "none":
            sdf = -torch.ones(pos.shape[0])
        else:
            sdf = torch.randn(pos.shape[0])
        sdf = sdf.to(dev).contiguous().float()
        if groups_stats := topo.word_groups():
            sign16 = (sdf > 0).cpu()
            pad = (-sign16.shape[0]) % 16
            f = torch.cat([sign16, torch.zeros(pad, dtype=torch.bool)]).reshape(-1, 16)
            state = torch.where(f.all(1), 2, torch.where(f.any(1), 1, 0))  # 0 out, 2 in, 1 mixed
            for name, tab in zip(("edge", "tet"), groups_stats):
                tab = tab.cpu().long()
                st = torch.where(tab >= 0, state[tab.clamp(min=0)], torch.ones_like(tab))
                skip = ((st == 0).all(1) | (st == 2).all(1)).reshape(-1, 16)
                print(f"R={res} {name} words skipped {skip.float().mean():.4f}, blocks fully skipped {skip.all(1).float().mean():.4f}")
        Ne, Nt, Nv = topo.edges32.shape[0], topo.tets32.shape[0], pos.shape[0]
        scratch = torch.empty(_lib.lib().a3d_dmtet_scratch_bytes(Ne, Nt), dtype=torch.uint8, device=dev)
        counts = torch.empty(6, dtype=torch.int32, device=dev)
        groups = topo.word_groups()
        # --surf: with the bit plane of the surface-adjacent grid vertices (the count call's fourth scan work-group), as the training step
        vscratch = torch.zeros(_lib.lib().a3d_dmtet_vertex_scratch_bytes(Nv), dtype=torch.uint8, device=dev) if args.surf else None
        for label, gr in (("plain", None), ("culled", groups)):
            if label == "culled" and gr is None:
                continue

            def run():
                ops.call("a3d_dmtet_count", ops.ptr(sdf), ops.ptr(topo.edges32), ops.ptr(topo.tets32), Ne, Nt, ops.ptr(scratch), ops.ptr(counts),
                         ops.ptr(vscratch), 1, Nv, ops.ptr(gr[0]) if gr else None, ops.ptr(gr[1]) if gr else None, None, 0, ops.stream())

            for _ in range(5):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            print(f"R={res} Nv={Nv} Ne={Ne} Nt={Nt} count[{label}]: {1e3 * e0.elapsed_time(e1) / args.iters:.1f} us  counts={counts.tolist()}", flush=True)


if __name__ == "__main__":
    main()
