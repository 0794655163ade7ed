"""Worker of test_two_process_ddp_on_one_gpu_replays_captured_graphs: rank r of 2, both on cuda:0, DDP over gloo (two RCCL ranks
cannot share a device).  Mirrors bench.py's order: build the scene, capture the HIP graphs, only then initialise torch.distributed."""
import importlib
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    torch.cuda.set_device(0)
    pipeline = importlib.import_module("3danimals_amd.pipeline")
    scene = pipeline.SyntheticScene(grid_res=16, batch=2, resolution=(64, 64), device="cuda:0", seed=0, data_seed=100 * rank, net_width=32,
                                    net_layers=3, feat_dim=16, embedder_freq=4)
    scene.netShape.capture_sdf_gradient_graph()
    n_graphs = len(scene.netShape._sdf_gradient_graphs)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    module = torch.nn.parallel.DistributedDataParallel(scene, device_ids=[0], broadcast_buffers=False)
    losses = []
    for _ in range(3):
        torch.manual_seed(11)  # same grid jitter / regulariser samples on both ranks
        losses.append(float(scene.step(module=module)["loss"]))
    sums = {n: float(p.grad.double().sum()) for n, p in scene.named_parameters()}
    weights = {n: float(p.detach().double().sum()) for n, p in scene.named_parameters()}
    json.dump(dict(rank=rank, n_graphs=n_graphs, graphs_after=len(scene.netShape._sdf_gradient_graphs), losses=losses, grad_sums=sums,
                   weight_sums=weights), open(out, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
