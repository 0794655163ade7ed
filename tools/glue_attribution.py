#!/usr/bin/env python
"""Who launched every GPU kernel of a step?  

``in_scope_us_per_step`` of bench.py sums the C-ABI entry points only.  The path's own Python modules (3danimals_amd/model/geometry,
model/render, ops.py) also launch torch kernels -- contiguous copies, slices and their padded gradients, small matmuls, gradient
accumulations -- and those were in nobody's figure.  This tool runs a few steps under torch.profiler with

  * every C-ABI call bracketed by a ``a3d_call/<entry point>`` range (ops.call is wrapped),
  * every public function of the path's modules bracketed by ``a3d_path/<module.function>``,
  * the networks' entry points (hostnets.*, DirectionalLight, DMTetGeometry.get_sdf*) by ``a3d_net/<...>``,
  * the consumer-side reads of the path's outputs (regulariser) by ``a3d_consumer/<...>``,

and attributes each GPU kernel (and memcpy / memset) of the trace to the innermost range that encloses its LAUNCH on the launching
thread.  Kernels launched by the autograd engine are attributed through the sequence number of the backward node that launched them
to the forward op that created the node, i.e. to the range that forward op ran in -- a SliceBackward belongs to whoever sliced.

Categories (us per step):
  a3d_in_scope   kernels of the in-scope entry points (bench.IN_SCOPE)            -- what in_scope_us_per_step approximates with events
  glue_fwd/bwd   torch kernels launched inside a3d_path ranges (forward / through their autograd nodes): THE PATH'S OWN GLUE
  f3_losses      a3d_recon_losses_* / a3d_flow_loss_* 
  f3_glue        torch kernels inside the f3 wrapper functions
  networks       a3d_net ranges, torch or HIP (model/networks: out of scope)
  optimizer      Optimizer.step
  consumer       a3d_consumer ranges
  other          the synthetic scene's own arithmetic (loss weighting, leaf bookkeeping, zero_grad ...)

Usage:  python tools/glue_attribution.py [--workload magicpony] [--steps 3] [--out gpurun_out/glue.json]
"""
import argparse
import bisect
import contextlib
import functools
import importlib
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

IN_SCOPE = ("a3d_dmtet_", "a3d_skin_", "a3d_bone_transforms_", "a3d_normals_", "a3d_mesh_topology", "a3d_rast_", "a3d_interp_", "a3d_cover_",
            "a3d_gbuffer_", "a3d_shade_", "a3d_aa_", "a3d_composite_aa_", "a3d_mask_aa_", "a3d_xfm_", "a3d_estimate_bones")
F3 = ("a3d_recon_losses_", "a3d_flow_loss_")

PATH_FUNCTIONS = [  # (module under 3danimals_amd, attribute path)
    ("model.geometry.dmtet", "DMTet.__call__"), ("model.geometry.dmtet", "DMTetGeometry.getMesh"),
    ("model.geometry.dmtet", "DMTetGeometry._get_mesh_surface_backward"),
    ("model.geometry.skinning", "estimate_bones"), ("model.geometry.skinning", "skinning"), ("model.geometry.skinning", "bone_transforms_torch"),
    ("model.render.mesh", "make_mesh"), ("model.render.mesh", "auto_normals"), ("model.render.mesh", "Mesh.deform"),
    ("model.render.mesh", "Mesh.normals_job"), ("model.render.mesh", "Mesh.take_normals"),
    ("model.render.render", "render_mesh"), ("model.render.render", "render_layer"), ("model.render.render", "shade"),
    ("model.render.render", "_shade_points"), ("model.render.render", "interpolate"), ("model.render.render", "_render_mesh_layers"),
    ("model.render.renderutils.ops", "xfm_points"), ("model.render.renderutils", "xfm_points"),
]
F3_FUNCTIONS = [("ops", "reconstruction_losses"), ("ops", "flow_loss")]
NET_FUNCTIONS = [
    ("hostnets", "CoordMLP.forward"), ("hostnets", "CoordMLP.sample"), ("hostnets", "CoordMLP_Mod.forward"), ("hostnets", "CoordMLP_Mod.sample"),
    ("hostnets", "MLP.forward"), ("hostnets", "MLP_Mod.forward"), ("model.render.light", "DirectionalLight.forward"),
    ("model.geometry.dmtet", "DMTetGeometry.get_sdf"), ("model.geometry.dmtet", "DMTetGeometry.get_sdf_gradient"),
]
CONSUMER_FUNCTIONS = [("pipeline", "prior_normal_regulariser")]
SCENE_FUNCTIONS = [("pipeline", "_SyntheticGeometry.get_sdf"), ("pipeline", "synthetic_spikes"), ("pipeline", "SyntheticScene.random_view_mask")]  # input generators


def _resolve(pkg, mod, path):
    m = importlib.import_module(f"{pkg}.{mod}")
    parts = path.split(".")
    owner = m
    for p in parts[:-1]:
        owner = getattr(owner, p)
    return owner, parts[-1]


@contextlib.contextmanager
def installed_ranges(pkg="3danimals_amd"):
    """Wrap the listed functions (and ops.call) in torch.profiler.record_function ranges; restore on exit."""
    import torch
    from torch.profiler import record_function

    undo = []

    def wrap(owner, name, label):
        if not hasattr(owner, name):
            return
        raw = owner.__dict__[name] if isinstance(owner, type) and name in owner.__dict__ else getattr(owner, name)
        kind = None
        fn = raw
        if isinstance(raw, staticmethod):
            kind, fn = staticmethod, raw.__func__
        elif isinstance(raw, classmethod):
            kind, fn = classmethod, raw.__func__
        elif isinstance(raw, property):
            return

        @functools.wraps(fn)
        def ranged(*a, **k):
            with record_function(label):
                return fn(*a, **k)

        undo.append((owner, name, raw))
        setattr(owner, name, kind(ranged) if kind else ranged)

    for group, prefix in ((PATH_FUNCTIONS, "a3d_path/"), (F3_FUNCTIONS, "a3d_f3/"), (NET_FUNCTIONS, "a3d_net/"), (CONSUMER_FUNCTIONS, "a3d_consumer/"),
                          (SCENE_FUNCTIONS, "a3d_scene/")):
        for mod, path in group:
            try:
                owner, name = _resolve(pkg, mod, path)
            except (ImportError, AttributeError):
                continue
            wrap(owner, name, f"{prefix}{mod.split('.')[-1]}.{path}")
    # every C-ABI entry point: ops.call is the one name the Functions use
    ops = importlib.import_module(f"{pkg}.ops")
    L = importlib.import_module(f"{pkg}._lib")
    raw_call = L.call

    def ranged_call(name, *args, tag=""):
        with record_function(f"a3d_call/{name}{tag}"):
            return raw_call(name, *args, tag=tag)

    undo.append((ops, "call", ops.call))
    undo.append((L, "call", L.call))
    ops.call = ranged_call
    L.call = ranged_call
    try:
        yield
    finally:
        for owner, name, raw in reversed(undo):
            setattr(owner, name, raw)
        del torch


class _ThreadTree:
    """CPU-side events of one thread (cpu_op + user_annotation), properly nested: innermost-enclosing queries by time."""

    def __init__(self, events):
        ev = sorted(events, key=lambda e: (e["ts"], -e.get("dur", 0.0)))
        self.ev = ev
        self.ts = [e["ts"] for e in ev]
        self.end = [e["ts"] + e.get("dur", 0.0) for e in ev]
        self.parent = [-1] * len(ev)
        stack = []
        for i in range(len(ev)):
            while stack and self.end[stack[-1]] < self.ts[i]:
                stack.pop()
            self.parent[i] = stack[-1] if stack else -1
            stack.append(i)

    def innermost(self, t):
        i = bisect.bisect_right(self.ts, t) - 1
        while i >= 0 and self.end[i] < t:
            i = self.parent[i]
        return i

    def ancestors(self, i):
        while i >= 0:
            yield i, self.ev[i]
            i = self.parent[i]


LABELS = ("a3d_call/", "a3d_path/", "a3d_f3/", "a3d_net/", "a3d_consumer/", "a3d_scene/", "Optimizer.step")


def _label_of(tree, i):
    """(innermost label range name or None, innermost autograd evaluate_function event or None) walking up from event index i."""
    label = node = None
    for _, e in tree.ancestors(i):
        n = e["name"]
        if label is None and n.startswith(LABELS):
            label = n
        if node is None and n.startswith("autograd::engine::evaluate_function"):
            node = e
        if label is not None:
            break
    return label, node


def attribute_trace(trace, steps):
    """chrome-trace dict -> report (see the module docstring)."""
    events = trace["traceEvents"]
    by_tid, runtime, fwd_by_seq = {}, {}, {}
    gpu = []
    for e in events:
        cat = e.get("cat")
        if cat in ("cpu_op", "user_annotation"):
            by_tid.setdefault(e["tid"], []).append(e)
        elif cat in ("cuda_runtime", "cuda_driver"):
            c = e.get("args", {}).get("correlation")
            if c is not None:
                runtime[c] = e
        elif cat in ("kernel", "gpu_memcpy", "gpu_memset"):
            gpu.append(e)
    trees = {tid: _ThreadTree(evs) for tid, evs in by_tid.items()}
    for tid, tree in trees.items():
        for i, e in enumerate(tree.ev):
            a = e.get("args", {})
            if e.get("cat") == "cpu_op" and a.get("Sequence number", -1) >= 0 and a.get("Fwd thread id", 0) == 0:
                fwd_by_seq[a["Sequence number"]] = (tid, i)  # (several ops share a number; the last one in time created the node)

    def classify(label, via_bwd):
        if label is None:
            return "other"
        if label.startswith("a3d_call/"):
            name = label[len("a3d_call/"):]
            if name.startswith(IN_SCOPE):
                return "a3d_in_scope"
            if name.startswith(F3):
                return "f3_losses"
            if name.startswith("a3d_bw_probe"):
                return "other"
            return "networks"
        if label.startswith("a3d_path/"):
            return "glue_bwd" if via_bwd else "glue_fwd"
        if label.startswith("a3d_f3/"):
            return "f3_glue"
        if label.startswith("a3d_net/"):
            return "networks"
        if label.startswith("a3d_consumer/"):
            return "consumer"
        if label.startswith("a3d_scene/"):
            return "other"
        if label.startswith("Optimizer.step"):
            return "optimizer"
        return "other"

    totals, detail, unplaced = {}, {}, 0
    for k in gpu:
        a = k.get("args", {})
        rt = runtime.get(a.get("correlation"))
        label = node = None
        via_bwd = False
        if rt is not None and rt["tid"] in trees:
            tree = trees[rt["tid"]]
            i = tree.innermost(rt["ts"])
            label, node = _label_of(tree, i)
            if label is None and node is not None:
                # launched by the autograd engine outside any range of ours: the forward op that created this node decides
                hit = fwd_by_seq.get(node.get("args", {}).get("Sequence number", -1))
                if hit is not None:
                    label = _enclosing_non_call(trees[hit[0]], hit[1])
                via_bwd = True
        else:
            unplaced += 1
        cat = classify(label, via_bwd)
        dur = float(k.get("dur", 0.0))
        totals[cat] = totals.get(cat, 0.0) + dur
        where = (label or "-") + ((" <- " + node["name"].split(": ")[-1]) if (via_bwd and node is not None) else "")
        key = (cat, k["name"][:90], where)
        d = detail.setdefault(key, [0, 0.0])
        d[0] += 1
        d[1] += dur
    per_step = {c: round(v / steps, 2) for c, v in sorted(totals.items())}
    rows = [dict(category=c, kernel=n, where=w, launches_per_step=round(cnt / steps, 2), us_per_step=round(us / steps, 2))
            for (c, n, w), (cnt, us) in detail.items()]
    rows.sort(key=lambda r: -r["us_per_step"])
    top = lambda cats, n: [r for r in rows if r["category"] in cats][:n]
    glue = per_step.get("glue_fwd", 0.0) + per_step.get("glue_bwd", 0.0)
    launches = lambda cats: round(sum(r["launches_per_step"] for r in rows if r["category"] in cats), 1)
    return dict(steps=steps, us_per_step=per_step, in_scope_a3d_us_per_step=per_step.get("a3d_in_scope", 0.0), in_scope_glue_us_per_step=round(glue, 2),
                in_scope_total_us_per_step=round(per_step.get("a3d_in_scope", 0.0) + glue, 2),
                glue_launches_per_step=launches(("glue_fwd", "glue_bwd")), a3d_in_scope_launches_per_step=launches(("a3d_in_scope",)),
                gpu_us_per_step=round(sum(totals.values()) / steps, 1), a3d_top=top(("a3d_in_scope", "f3_losses"), 60), glue_top=top(("glue_fwd", "glue_bwd"), 80), f3_glue_top=top(("f3_glue",), 6),
                other_top=top(("other", "consumer"), 8), gpu_events_without_launch_site=unplaced,
                note="kernel durations from torch.profiler's trace (the same clock rocprofv3 reads); a kernel belongs to the innermost range around its "
                     "LAUNCH; engine-launched kernels follow their backward node's sequence number to the forward op that created it")


def _enclosing_non_call(tree, i):
    for _, e in tree.ancestors(i):
        n = e["name"]
        if n.startswith(LABELS) and not n.startswith("a3d_call/"):
            return n
    return None


def profile_steps(step_fn, steps=3, keep_trace=None):
    """Run ``steps`` calls of ``step_fn`` under torch.profiler with the ranges installed -> report."""
    import torch
    from torch.profiler import ProfilerActivity, profile

    with installed_ranges():
        step_fn()  # (one step outside the trace with the wrappers on: first-use caches)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            for _ in range(steps):
                step_fn()
            torch.cuda.synchronize()
    path = keep_trace or os.path.join(tempfile.mkdtemp(prefix="a3d_trace_"), "trace.json")
    prof.export_chrome_trace(path)
    with open(path) as f:
        trace = json.load(f)
    if keep_trace is None:
        os.remove(path)
    return attribute_trace(trace, steps)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="magicpony", choices=("magicpony", "fauna", "ponymation"))
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--grid-res", type=int, default=64)
    ap.add_argument("--mesh", default="quadruped")
    ap.add_argument("--out", default=None)
    ap.add_argument("--keep-trace", default=None)
    args = ap.parse_args()
    import torch

    pipeline = importlib.import_module("3danimals_amd.pipeline")
    importlib.import_module("3danimals_amd.gemm_tuning").enable()
    batch = args.batch if args.batch is not None else (8 if args.workload == "ponymation" else 16)
    scene = pipeline.SyntheticScene(grid_res=args.grid_res, batch=batch, resolution=(256, 256), device="cuda", seed=0, workload=args.workload,
                                    num_frames=8 if args.workload == "ponymation" else 1, deform=args.workload == "magicpony", mesh=args.mesh)
    scene.netShape.capture_sdf_gradient_graph()
    for _ in range(5):
        scene.step()
    torch.cuda.synchronize()
    rep = profile_steps(lambda: scene.step(), args.steps, args.keep_trace)
    rep["workload"] = args.workload
    text = json.dumps(rep, indent=1)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write(text)
    print(text)


if __name__ == "__main__":
    main()
