#!/usr/bin/env python
"""Benchmark of the reconstruct-and-render hot path:  python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): train_magicpony_horse-like synthetic step,
batch 16 per GPU @ 256x256, forward + backward + Adam -- DMTet on a Kuhn R=64 grid (the stand-in for the reference's
"128" Quartet grid), skinning with 20 bones, rasterise / interpolate / antialias, the texture / DINO /
light / SDF MLPs at the reference's sizes, photometric + mask + DINO-feature losses.  See 3danimals_amd/pipeline.py.

N > 1: launched by the driver through torch.distributed.run, one rank per GPU, DDP over RCCL (gradient all-reduce of
the MLP parameters; the hot-path kernels themselves exchange nothing: images shard over the batch) -> weak scaling.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the dominant hot-path HIP kernel: algorithmic bytes per launch / mean launch duration measured live
                  with HIP events on the launch stream, against the 8 TB/s HBM peak;
  cpu_baseline -- the CPU oracle (kind "port") timed on this box's host cores on a bounded sample of the batch.
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_FP32_PEAK_TFLOPS = 157.3  # dense fp32-input MFMA (v_mfma_f32_32x32x2_f32), same guide


def algorithmic_bytes(name, d):
    """Algorithmic HBM bytes of ONE call of a C-ABI entry point (DESIGN.md 'Kernels').

    ``name`` may carry a channel tag, e.g. 'a3d_interp_fwd[C3]'.  Index buffers shared over the batch count once.
    """
    B, V, F, HW, Nv, Ne, Nt, K = d["B"], d["V"], d["F"], d["H"] * d["W"], d["Nv"], d["Ne"], d["Nt"], d["K"]
    C = int(name.split("[C")[1].split("]")[0]) if "[C" in name else 0
    base = name.split("[")[0]
    Pp = -(-int(d.get("P", 0)) // 8192) * 8192  # render.POINT_BUCKET
    table = {
        "a3d_dmtet_count": 4 * Nv + 8 * Ne + 16 * Nt,
        "a3d_dmtet_emit": 16 * Nv + 8 * Ne + 4 * Ne + 16 * Nt + 24 * Nt + 16 * V + 48 * F,
        "a3d_dmtet_bwd": 12 * V + 4 * V + 8 * V + 4 * Nv,
        "a3d_skin_fwd": 12 * V + 12 * B * V,
        "a3d_skin_bwd": 12 * B * V + 12 * V + 12 * V + 48 * B * K,
        "a3d_normals_adjacency": 24 * F + 4 * V,  # triangle list in, CSR out
        "a3d_normals_fwd": 4 * V + 24 * F + B * (36 * F + 24 * V),  # CSR + indices once; per image position gathers, acc + nrm out
        "a3d_normals_bwd": 4 * V + 24 * F + B * (36 * F + 36 * F + 60 * V),
        "a3d_rast_fwd": B * (16 * V + 16 * HW) + 12 * F,
        "a3d_rast_bwd": B * (32 * HW + 16 * V),
        "a3d_interp_fwd": B * (16 * HW + 4 * C * HW),
        "a3d_interp_bwd": B * (16 * HW + 4 * C * HW + 16 * HW + 4 * C * V),
        "a3d_gbuffer_fwd": int(d.get("P", 0)) * (8 + 16 + 48),
        "a3d_gbuffer_bwd": int(d.get("P", 0)) * (8 + 16 + 48) + B * V * (36 + 16),
        "a3d_mesh_topology": 12 * F + (4 * V + 12 * F) + 12 * F,  # triangle list in; CSR + opposite-vertex table out
        "a3d_rows_segsum": 4 * Pp * C + 4 * B * C,  # P here = the padded point list the fields see
        "a3d_rows_add_relu_fwd": 8 * Pp * C,  # y read + written in place; rows[B,C] stay in L2
        "a3d_rows_add_relu_bwd": 12 * Pp * C + 4 * B * C,  # g, y in; g_pre out; per-image sums
        "a3d_gemm_nn_relumask": Pp * 256 * 4 * 3 + 256 * 256 * 4,  # A in, X (mask) in, C out; weight from L2.  Compute bound: see flops below
        "a3d_harmonic_embed_fwd": Pp * (12 + 4 * 64),  # texture field's input stage (n = 10: 64 columns); the DINO field's is 52 wide
        "a3d_harmonic_embed_bwd": Pp * (4 * 64 + 12 + 12),
        "a3d_recon_losses_fwd": B * HW * (16 + 64 + 12 + 64 + 12 + 1),  # shaded, dino(16), image_gt, dino_gt, three masks; 'both' out
        "a3d_recon_losses_bwd": B * HW * (16 + 64 + 12 + 64 + 12 + 1 + 16 + 64),
        "a3d_cover_count": 4 * B * HW,  # the id channel of the raster buffer
        "a3d_cover_emit": 4 * B * HW + 8 * int(d.get("P", 0)),
        "a3d_shade_fwd": int(d.get("P", 0)) * (48 + 68 + 12 + 12 + 4 + 12),
        "a3d_shade_bwd": int(d.get("P", 0)) * (48 + 68 + 12 + 28 + 48 + 68 + 12),
        "a3d_bone_transforms_fwd": B * K * (12 + 48) + 24 * K,
        "a3d_bone_transforms_bwd": B * K * (12 + 48 + 12) + 24 * K,
        "a3d_aa_topology": 12 * F + 12 * F,
        "a3d_aa_analyze": B * 16 * HW,
        "a3d_aa_fwd": B * 8 * C * HW,
        "a3d_aa_bwd": B * 8 * C * HW + B * 16 * V,
    }
    return table.get(base)


def algorithmic_flops(name, d):
    """Flops of ONE call for the compute-bound entry points (None for the bandwidth-bound ones)."""
    Pp = -(-int(d.get("P", 0)) // 8192) * 8192
    return {"a3d_gemm_nn_relumask": 2 * Pp * 256 * 256}.get(name.split("[")[0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--grid-res", type=int, default=64)
    ap.add_argument("--resolution", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the HIP-event per-kernel pass (roofline = null); for PMC runs")
    ap.add_argument("--no-tuned-gemms", action="store_true", help="do not load the shipped TunableOp results for the torch MLPs")
    ap.add_argument("--cpu-sample-images", type=int, default=4)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs an MI355X; the HIP hot path has no CPU fallback"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pipeline = importlib.import_module("3danimals_amd.pipeline")
    L = importlib.import_module("3danimals_amd._lib")
    L.lib()  # fail loudly, now, if the HIP library is missing
    tuned = importlib.import_module("3danimals_amd.gemm_tuning").enable() if not args.no_tuned_gemms else False

    scene = pipeline.SyntheticScene(grid_res=args.grid_res, batch=args.batch, resolution=(args.resolution, args.resolution), device=dev,
                                    seed=0, data_seed=1000 * rank)
    scene.netShape.capture_sdf_gradient_graph()  # HIP graphs are captured before any RCCL thread exists; the steps only replay them
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm
    # weak scaling = fixed work per GPU: every rank renders the same 16 poses / cameras (hence the same number of covered pixels and
    # the same GEMM shapes) against its own image features and target images, so the all-reduced gradients differ per rank.
    module = None
    if world > 1:
        module = torch.nn.parallel.DistributedDataParallel(scene, device_ids=[local_rank], broadcast_buffers=False, gradient_as_bucket_view=True)

    # W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize on both sides, MAX over ranks
    du = importlib.import_module("3danimals_amd.dist_util")
    elapsed = du.timed_steps(lambda: scene.step(module=module), args.steps, args.warmup, device=dev)
    images = world * args.batch * args.steps

    # ---- per-kernel timing pass (same workload, separate from the headline timing so the events do not perturb it)
    roofline, kernels = None, {}
    if rank == 0 and not args.no_kernel_timing:
        import contextlib

        # rank 0 alone re-runs a few steps under HIP-event timers; with DDP that must not enqueue collectives the other ranks
        # never join, hence no_sync() (local gradients only) and no optimizer step
        with L.KernelTimer() as timer, (module.no_sync() if module is not None else contextlib.nullcontext()):
            for _ in range(min(args.steps, 10)):
                scene.step(module=module, optimizer_step=(world == 1))
        prior, shape = scene.last["prior"], scene.last["shape"]
        dims = dict(B=args.batch, V=int(prior.v_pos.shape[1]), F=int(prior.t_pos_idx.shape[1]), H=args.resolution, W=args.resolution,
                    Nv=int(scene.netShape.verts.shape[0]), Ne=int(scene.netShape.topology.edges32.shape[0]),
                    Nt=int(scene.netShape.topology.tets32.shape[0]), K=int(scene.bones.shape[2]),
                    P=int((scene.last["rast"][..., 3] > 0).sum()) if "rast" in scene.last else 0)
        total_ms = 0.0
        for name, (count, mean_ms) in sorted(timer.summary().items()):
            per_step = count / min(args.steps, 10)
            ab = algorithmic_bytes(name, dims)
            kernels[name] = dict(launches_per_step=round(per_step, 2), mean_us=round(mean_ms * 1e3, 2),
                                 algorithmic_MB=None if ab is None else round(ab / 1e6, 3),
                                 GBps=None if ab is None else round(ab / (mean_ms * 1e-3) / 1e9, 1))
            total_ms += mean_ms * per_step
        # dominant = most time per step among the C-ABI entry points (all have a byte model)
        cand = {k: v for k, v in kernels.items() if v["GBps"] is not None}
        dom = max(cand, key=lambda k: cand[k]["mean_us"] * cand[k]["launches_per_step"])
        # bytes-weighted aggregate over the whole HIP path: sum(algorithmic bytes) / sum(time), per step
        tot_b = sum(v["algorithmic_MB"] * v["launches_per_step"] for v in cand.values()) * 1e6
        tot_t = sum(v["mean_us"] * v["launches_per_step"] for v in cand.values()) * 1e-6
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")  # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (committed)
        if os.path.exists(pmc):
            rec = json.load(open(pmc))["per_call"].get(dom.split("[")[0])
            traffic = None if rec is None else round(rec["traffic_MB"] * 1e6)
        hip_path = dict(ms_per_step=round(total_ms, 3), algorithmic_MB_per_step=round(tot_b / 1e6, 1),
                        GBps=round(tot_b / tot_t / 1e9, 1), frac=round(tot_b / tot_t / 1e9 / HBM_PEAK_GBS, 4))
        flops = algorithmic_flops(dom, dims)
        if flops is not None:  # the one compute-bound kernel of the path: fp32 MFMA GEMM (157.3 TFLOP/s dense fp32 MFMA peak, MI355X guide)
            tf = flops / (cand[dom]["mean_us"] * 1e-6) / 1e12
            roofline = dict(kernel=dom, bound="mfma", achieved=round(tf, 1), peak=MFMA_FP32_PEAK_TFLOPS, unit="TFLOP/s",
                            frac=round(tf / MFMA_FP32_PEAK_TFLOPS, 4), traffic=traffic, launch_us=cand[dom]["mean_us"],
                            algorithmic_flops_per_launch=flops, algorithmic_bytes_per_launch=round(cand[dom]["algorithmic_MB"] * 1e6),
                            hbm_GBps=cand[dom]["GBps"], hip_path=hip_path, mesh=dims)
        else:
            roofline = dict(kernel=dom, bound="hbm", achieved=cand[dom]["GBps"], peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=round(cand[dom]["GBps"] / HBM_PEAK_GBS, 4), traffic=traffic, launch_us=cand[dom]["mean_us"],
                            algorithmic_bytes_per_launch=round(cand[dom]["algorithmic_MB"] * 1e6), hip_path=hip_path, mesh=dims)
        # the streaming kernel with the most time per step, for the HBM side of the picture
        mem = {k: v for k, v in cand.items() if algorithmic_flops(k, dims) is None}
        if mem:
            top = max(mem, key=lambda k: mem[k]["mean_us"] * mem[k]["launches_per_step"])
            roofline["top_hbm_kernel"] = dict(kernel=top, achieved=mem[top]["GBps"], unit="GB/s", frac=round(mem[top]["GBps"] / HBM_PEAK_GBS, 4),
                                              launch_us=mem[top]["mean_us"])

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import step_ref

        threads = min(os.cpu_count(), 32)  # torch-CPU stops scaling (and thrashes) far below the 256 logical cores of the GPU box
        torch.set_num_threads(threads)
        n = max(1, min(args.cpu_sample_images, args.batch))
        st = step_ref.snapshot(scene, n)
        res = step_ref.cpu_step(st, backward=True)
        cpu_baseline = dict(value=round(n / res["seconds"], 4), unit="images/s", cores=threads, kind="port",
                            sample=f"oracle/step_ref.cpu_step fwd+bwd on {n} of the {args.batch} images of this workload "
                                   f"(Kuhn R={args.grid_res} DMTet, LBS, {args.resolution}x{args.resolution} raster+shade+antialias, losses), "
                                   f"1 run, torch {threads} threads of {os.cpu_count()} logical cores, {res['seconds']:.1f} s")

    config1 = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # BASELINE config 1 (geometry only, no raster): CPU oracle beside the HIP path on the same inputs
        from oracle import geometry_ref

        dmtet_mod = importlib.import_module("3danimals_amd.model.geometry.dmtet")
        inp = geometry_ref.make_inputs(res=32, batch=16, seed=0)
        for _ in range(2):
            geometry_ref.cpu_step(inp)
        cpu_s = sorted(geometry_ref.cpu_step(inp)["seconds"] for _ in range(5))[2]
        gin = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
        topo = dmtet_mod.TetGridTopology(gin["tets"])
        for _ in range(3):
            res1 = pipeline.geometry_config1_step(gin, topo)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(20):
            pipeline.geometry_config1_step(gin, topo)
        torch.cuda.synchronize()
        hip_s = (time.perf_counter() - t1) / 20
        config1 = dict(workload="DMTet (Kuhn R=32) + normals + estimate_bones + LBS (B=16, K=20) + normals, fwd+bwd, no raster",
                       mesh=dict(V=res1["V"], F=res1["F"]), cpu_port_ms=round(cpu_s * 1e3, 2), cpu_images_per_s=round(16 / cpu_s, 1),
                       cpu_threads=threads, hip_ms=round(hip_s * 1e3, 3), hip_images_per_s=round(16 / hip_s, 1),
                       note="estimate_bones (host logic with read-backs, once per epoch in training) is inside both timings")

    if rank == 0:
        line = {
            "metric": "train images/sec fwd+bwd @256x256 b16",
            "value": round(images / elapsed, 3),
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "train_magicpony_horse-like synthetic step: DMTet(Kuhn R=%d)+LBS(20 bones)+raster/interp/antialias "
                                   "+ SDF/texture/DINO/light MLPs + photometric/mask/DINO losses, fwd+bwd+Adam" % args.grid_res,
                       "batch_per_gpu": args.batch, "global_batch": world * args.batch, "resolution": [args.resolution, args.resolution],
                       "grid": f"kuhn{args.grid_res}", "parallelism": f"dp{world}", "tuned_mlp_gemms": bool(tuned),
                       "per_rank_data": "same poses/cameras on every rank (equal work per GPU), per-rank image features and targets"},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "config1_geometry": config1,
            "kernels": kernels,
        }
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:  # the ONE json line is the last thing written to stdout (after RCCL's own start-up / tear-down chatter)
        sys.stdout.flush()
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
