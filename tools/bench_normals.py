"""Isolated timing of vertex normals (fwd / bwd) on the bench mesh, sorted (a3d_mesh_topology) vs extraction-built lists."""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
ops = importlib.import_module("3danimals_amd.ops")
L = importlib.import_module("3danimals_amd._lib")
pipeline = importlib.import_module("3danimals_amd.pipeline")
dev = torch.device("cuda:0")
scene = pipeline.SyntheticScene(grid_res=64, batch=16, resolution=(64, 64), device=dev, seed=0, net_width=32, net_layers=3, feat_dim=16, embedder_freq=4)
scene.step(backward=False)
prior, shape = scene.last["prior"], scene.last["shape"]
tri32 = ops.tri_int32(prior.t_pos_idx)
V = prior.v_pos.shape[1]
adj_x = ops._adj_cache.peek(tri32)
adj_s, _ = ops.mesh_topology(tri32.clone(), V)
print("V", V, "F", tri32.shape[0], "extraction lists present:", adj_x is not None)
big = torch.empty(64 << 20, device=dev)
for name, adj in (("sorted", adj_s), ("extraction", adj_x)):
    if adj is None: continue
    for B, v0 in ((16, shape.v_pos.detach()), (1, prior.v_pos.detach())):
        v = v0.clone().requires_grad_(True)
        g = torch.randn_like(v)
        def run():
            n = ops._Normals.apply(v, tri32, adj)
            n.backward(g)
        for _ in range(5): run()
        torch.cuda.synchronize()
        with L.KernelTimer() as t:
            for _ in range(30):
                big.zero_()  # keep the GPU busy so the events do not include host launch latency
                run()
        print(name, B, {k: round(v[1] * 1e3, 1) for k, v in t.summary().items()})
