"""World-size-2 data-parallel test on CPU (gloo): the N>1 path of bench.py without a GPU.

Each rank takes its shard of the batch, runs forward+backward of the step (the CPU oracle stands in for the HIP kernels --
tests may use it as the engine), the MLP gradients are averaged with the same helper bench.py falls back to, and the result
must equal the single-process gradient of the whole batch.  Also checks the barrier-bracketed max-over-ranks timing.
"""
import importlib
import os
import socket
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


PER_IMAGE = ("mvp", "w2c", "campos", "feat", "arti", "image_gt", "dino_gt", "mask_gt", "mask_dt", "mask_valid", "background")


def _shard(st, lo, hi):
    out = dict(st)
    for k in PER_IMAGE:
        out[k] = st[k][lo:hi]
    out["n"] = hi - lo
    return out


def _worker(rank, world, port, result):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from oracle import step_ref

        du = importlib.import_module("3danimals_amd.dist_util")
        st = step_ref.synthetic_state(grid_res=12, n=4, resolution=(32, 32), seed=0)  # identical weights on every rank
        lo, hi = du.shard_range(4, rank, world)
        local = step_ref.cpu_step(_shard(st, lo, hi), backward=True)
        nets = [st["sdf_mlp"], st["tex"], st["dino"], st["lgt"]]
        params = [p for m in nets for p in m.parameters()]
        assert all(p.grad is not None for p in params)  # find_unused_parameters=False must be fine
        du.allreduce_mean_grads(params, bucket_bytes=1 << 14)  # several buckets
        reduced = [p.grad.clone() for p in params]
        # timing helper: rank 1 is slower; everyone must report the max
        elapsed = du.timed_steps(lambda: time.sleep(0.02 * (rank + 1)), steps=3, warmup=1)
        if rank == 0:
            full = step_ref.cpu_step(st, backward=True)
            ok = True
            for p, g in zip(params, reduced):
                scale = float(p.grad.abs().max()) + 1e-12
                ok &= bool((p.grad - g).abs().max() <= 2e-4 * scale + 1e-7)
            # per-image leaves: d(mean over 4)/d(image) = (1/2) d(mean over this rank's 2)/d(image)
            ok &= bool(torch.allclose(full["grads"]["arti"][lo:hi], local["grads"]["arti"] / world, rtol=1e-3, atol=1e-6))
            ok &= abs(float(full["loss"]) - float(local["loss"])) < 10  # different shards, same scale
            result.put((ok, elapsed))
        else:
            result.put((True, elapsed))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_average_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for ok, _ in results)
    times = [t for _, t in results]
    assert abs(times[0] - times[1]) < 1e-6 and times[0] >= 3 * 0.04 * 0.9  # both report the slower rank's time


def test_shard_range_covers_batch():
    du = importlib.import_module("3danimals_amd.dist_util")
    for n in (1, 7, 16, 128):
        for w in (1, 2, 3, 8):
            spans = [du.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_bench_plain_command_spawns_its_own_ranks_dry_launch():
    """`python bench.py --gpus 2` with no launcher around it (WORLD_SIZE unset): bench.py re-runs itself under torch.distributed.run, the
    ranks rendezvous on 127.0.0.1, all-reduce, and rank 0's JSON line is the ONLY thing on stdout (other output is passed to stderr).
    --dry-launch skips the GPU work, so the launch / collect plumbing of the 8-GPU driver run is exercised here on CPU."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-launch", "--steps", "3"],
                       capture_output=True, text=True, timeout=300, env=env, cwd="/tmp")
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-2000:]
    lines = p.stdout.strip().splitlines()
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["allreduce_sum"] == 3.0 and line["steps"] == 3
    assert "self-launch" in line["launcher"] and "stdout chatter" in p.stderr
    # and under an external launcher (the driver's documented command shape) the same line comes straight from rank 0
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-launch"],
                       capture_output=True, text=True, timeout=300, env=env, cwd="/tmp")
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["rccl_ranks"] == 2 and "launcher" not in line
    # a wrong --gpus under a launcher is refused before any collective
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--dry-launch"], capture_output=True, text=True, timeout=120,
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), cwd="/tmp")
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr
