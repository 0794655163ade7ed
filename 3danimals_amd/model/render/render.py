"""render_mesh / render_layer / shade / render_uv -- public API of /root/reference/model/render/render.py.

Same signatures, same return contract (a list of NCHW tensors in ``render_modes`` order; an unknown mode raises KeyError
inside shade like the reference, render.py:127-128), same arithmetic; underneath, the nvdiffrast calls become HIP kernels:

* one triangle-parallel rasterisation pass (csrc/raster.hip: 64-bit atomicMin on depth|id, then a resolve) instead of the OpenGL
  depth peeler (layers > 0 peel behind the previous layer's (depth, id): render_mesh(num_layers > 1) takes the general dense path);
* on the training path (spp 1, no tangent/depth mode) ONE fused kernel builds the G-buffer of the covered pixels only
  (csrc/cover.hip + csrc/gbuffer.hip: world position, face normal, smooth normal, canonical position; its backward also carries
  the rasteriser's gradient), ONE kernel does the shading arithmetic (csrc/shade.hip), and the texture / DINO fields see the
  covered pixels only -- output-identical, because uncovered pixels are composited with alpha 0 (render.py:261-262);
* otherwise the modular path: interpolation of each attribute (csrc/interp.hip), the geometric normal as the reference's
  per-face attribute with a [[f,f,f]] index buffer (render.py:185-191);
* tangents are neither computed nor interpolated unless the 'tangent' mode asks for them (dead otherwise, a4);
* ONE silhouette analysis per render (csrc/antialias.hip), applied to every buffer that the reference antialiases
  separately (render.py:311-315).

The texture / DINO / light MLPs (``material.sample``, ``dino_net.sample``, ``lgt.shade``) stay PyTorch modules.
"""
from __future__ import annotations

import os

import torch

from ... import ops
from ..._lib import fp32_region
from . import light, util
from . import renderutils as ru

ANTIALIASED_MODES = ("shaded", "flow", "dino_pred", "depth", "shading")  # reference render.py:311
FUSED_COMPOSITE = True  # sparse buffers: composite + antialias as one op (ops.composite_antialias); False = torch composite + ops.antialias
SHADE_COVERED_ONLY = True  # evaluate the texture / DINO MLPs on rasterised pixels only (output-identical; see shade())
POINT_BUCKET = 8192  # pad the covered-point list seen by the MLPs to a multiple of this (0 = off)
LAST_RAST = [None]
LAST_POINTS = [None]  # introspection hook like LAST_RAST: what the fused path handed from stage to stage in the last render_mesh call
FUSED_GBUFFER = True  # build the G-buffer of the covered pixels with one fused HIP kernel (csrc/gbuffer.hip)
FUSED_COVER_GBUFFER = True  # ... and the covered-pixel list in the same launch (a3d_cover_gbuffer_fwd) instead of a3d_cover_emit before it
DEFER_RESOLVE = os.environ.get("A3D_DEFER_RESOLVE", "1") != "0"  # ... and the rasteriser's resolve in that launch too (a3d_rast_resolve_gbuffer_fwd)
DEFER_ANALYSIS = os.environ.get("A3D_DEFER_ANALYSIS", "1") != "0"  # the silhouette analysis as extra work-groups of the compositor's first launch
FUSED_FLOW_DELTA = os.environ.get("A3D_FUSED_FLOW_DELTA", "1") != "0"  # 'flow': the per-vertex motion to the next frame as one launch each way (ops.flow_delta)
FUSED_MASK_RENDER = True  # a render without material, light and feature field whose only mode is 'shaded' skips the G-buffer: ops.mask_antialias
FUSED_SHADING = True
ALIAS_POSITIONS = os.environ.get("A3D_ALIAS_POSITIONS", "1") != "0"  # the clip transform's node feeds the G-buffer's position attribute too (one accumulation launch less)
FIELD_INPUTS_FROM_GBUFFER = os.environ.get("A3D_FIELD_INPUTS", "1") != "0"  # the fields' padded input rows + image index written by the G-buffer launch (round 6)
SHADE_IN_COMPOSITOR = os.environ.get("A3D_SHADE_IN_COMPOSITOR", "1") != "0"  # no a3d_shade_fwd launch when only the composited colour reads its output  # shading normal + camera normal + directional light of the covered pixels in one HIP kernel (csrc/shade.hip)


# Shading only the covered pixels makes the sizes of the networks' activations follow the silhouette ([P,256] rows: ~5 GB per step at
# B = 16), where the reference's dense [B,H,W] inputs never change size.  The caching allocator keeps every block it ever handed out:
# when P outgrows a bucket, the ~40 activation blocks of the old size are too small for the new requests and stay cached for good
# -- measured on a steadily growing shape: 5.5 GB reserved after the first step, 208 GB after 1200, for 1 GB in use and a 10 GB peak
# (tools/mem_growth.py).  So whenever a point count appears that no earlier step had, the cache is given back to the driver if it
# exceeds ALLOCATOR_TRIM_RATIO x the peak of what was ever in use (a device synchronisation and a few hundred hipFree calls, at most
# once per new size; 0 = never, env A3D_ALLOCATOR_TRIM_RATIO).  The peak is tracked HERE (memory_allocated sampled at every check and
# at every new size), not read from max_memory_allocated: a caller that resets torch's peak statistics every epoch would otherwise make
# every new size look like a reason to trim.  Under data-parallel training the ranks meet a new size at different steps and the rank
# that trims (milliseconds) keeps the others waiting at the next all-reduce -- at most once per size per rank, and only while its
# cache is three times what it ever needed; ranks that must not stall set the ratio to 0 and cap the cache with
# PYTORCH_HIP_ALLOC_CONF=garbage_collection_threshold instead.
ALLOCATOR_TRIM_RATIO = float(os.environ.get("A3D_ALLOCATOR_TRIM_RATIO", "3"))
# (round 5) WHEN the cache is given back: 'inline' = where the new size is met, between the G-buffer and the networks of that forward (a
# caller that only overlays model/render gets this: it has no other hook); 'step_end' = the render only records the wish and the
# training loop calls allocator_trim_at_step_end() after its optimiser step (pipeline.SyntheticScene.step does): no synchronisation
# and no hipFree in the middle of a forward, the step's own activations are back in the cache by then (more of it can go), and under
# data-parallel training the millisecond falls between two steps instead of between a rank's forward and the others' all-reduce.
ALLOCATOR_TRIM_MODE = os.environ.get("A3D_ALLOCATOR_TRIM_MODE", "inline")
_point_counts_seen = set()
_in_use_peak = {}
_trim_wanted = set()


def allocator_trim_at_step_end(device=None):
    """Give the cache back now if a render of this step asked for it (ALLOCATOR_TRIM_MODE 'step_end').  -> True when it did."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _trim_wanted:
        return False
    _trim_wanted.discard(key)
    torch.cuda.empty_cache()
    return True


_matmul_fp32 = fp32_region(torch.matmul)  # camera algebra of the path: float32 also inside the caller's autocast region (see _lib.fp32_region)


def _trim_allocator_cache(n_points, device):
    if ALLOCATOR_TRIM_RATIO <= 0 or device.type != "cuda":
        return False
    key = device.index if device.index is not None else torch.cuda.current_device()
    peak = _in_use_peak[key] = max(_in_use_peak.get(key, 1 << 30), torch.cuda.memory_allocated(device))  # (a counter read: no synchronisation)
    if n_points in _point_counts_seen or torch.cuda.is_current_stream_capturing():
        return False
    if len(_point_counts_seen) >= 4096:
        _point_counts_seen.clear()
    _point_counts_seen.add(n_points)
    # (this call sits after the G-buffer and before the networks of a step: the step's own activations are not allocated yet, the
    # peak is what earlier steps needed -- the previous step's networks are sampled by the NEXT call's memory_allocated only while
    # their graph is alive, hence also torch's own figure, which cannot be lower than what this process really peaked at since its
    # last reset)
    in_use_peak = max(peak, torch.cuda.max_memory_allocated(device))
    _in_use_peak[key] = in_use_peak
    if torch.cuda.memory_reserved(device) <= ALLOCATOR_TRIM_RATIO * in_use_peak:
        return False
    if ALLOCATOR_TRIM_MODE == "step_end":
        _trim_wanted.add(key)
        return False
    torch.cuda.empty_cache()
    return True


def interpolate(attr, rast, attr_idx, rast_db=None):
    """dr.interpolate wrapper of the reference (render.py:23-24): (values, None) -- or (values, pixel differentials of every attribute)
    when ``rast_db`` is given, which no caller on this path does."""
    out = ops.interpolate(attr.contiguous(), rast, attr_idx)
    return out, (None if rast_db is None else ops.interpolate_da(attr.contiguous(), rast, attr_idx, rast_db, "all"))


def shade(gb_pos, gb_geometric_normal, gb_normal, gb_tangent, gb_tex_pos, w2c, view_pos, lgt, material, bsdf, feat=None, render_modes=None,
          two_sided_shading=True, delta_xy_interp=None, dino_net=None, class_vector=None, cover=None):
    """Per-pixel shading (reference render.py:30-132): texture / DINO field lookups, shading normal, directional light.

    ``cover`` (bool [B,H,W], optional, not in the reference signature): evaluate everything on the covered pixels only
    and scatter into dense buffers.  Output-identical after compositing -- uncovered pixels are overwritten by
    lerp(bg, ., 0) (render.py:261-262) and antialias only mixes composited colours -- but the two coordinate MLPs,
    which dominate the FLOPs of the whole step, see 3-5x fewer points.
    """
    if cover is not None and (render_modes is None or "depth" not in render_modes):
        return _shade_covered(gb_pos, gb_geometric_normal, gb_normal, gb_tangent, gb_tex_pos, w2c, view_pos, lgt, material, bsdf, feat,
                              render_modes, two_sided_shading, delta_xy_interp, dino_net, class_vector, cover)
    if material is not None:
        all_tex = material.sample(gb_tex_pos, feat=feat)
    else:
        all_tex = torch.ones(*gb_pos.shape[:-1], 9, device=gb_pos.device)
    kd, ks = all_tex[..., :3], all_tex[..., 3:6]
    dino_pred = dino_net.sample(gb_tex_pos, feat=class_vector) if dino_net is not None else None
    alpha = torch.ones_like(kd[..., 0:1])

    # the reference discards the MLP's normal channels here (render.py:71): no perturbation
    gb_normal = ru.prepare_shading_normal(gb_pos, view_pos, None, gb_normal, gb_tangent, gb_geometric_normal,
                                          two_sided_shading=two_sided_shading, opengl=True, use_python=True)
    b, h, w, _ = gb_normal.shape
    cam_normal = util.safe_normalize(_matmul_fp32(gb_normal.view(b, -1, 3), w2c[:, :3, :3].transpose(2, 1))).view(b, h, w, 3)

    bsdf = _resolve_bsdf(bsdf, material)
    shading = None
    if lgt is None:
        shaded_col = kd
    elif isinstance(lgt, light.EnvironmentLight):
        raise NotImplementedError("EnvironmentLight is outside the hot path")
    else:
        shaded_col, shading = lgt.shade(feat, kd, cam_normal)

    depth = None
    if render_modes is not None and "depth" in render_modes:
        hom = torch.cat([gb_pos, torch.ones_like(gb_pos[..., :1])], dim=-1)
        depth = _matmul_fp32(hom.view(b, -1, 4), w2c.transpose(-1, -2)).view(b, h, w, 4)[..., 2]
        dmin, dmax = depth.amin(dim=(1, 2), keepdim=True), depth.amax(dim=(1, 2), keepdim=True)
        depth = ((depth - dmin) / (dmax - dmin)).unsqueeze(-1)

    buffers = _collect(render_modes, shaded_col, kd, ks, gb_normal, gb_geometric_normal, gb_tangent, shading, delta_xy_interp, dino_pred, depth)
    if render_modes is not None:  # an unknown or unavailable mode raises KeyError here, exactly like the reference (render.py:127-128)
        return {mode: torch.cat((buffers[mode], alpha), dim=-1) for mode in render_modes}
    return {"shaded": torch.cat((shaded_col, alpha), dim=-1)}


def _resolve_bsdf(bsdf, material):
    assert bsdf is not None or material.bsdf is not None, "Material must specify a BSDF type"
    bsdf = bsdf if bsdf is not None else material.bsdf
    if bsdf == "pbr":
        raise NotImplementedError("bsdf='pbr' needs an EnvironmentLight (reference render.py:83-87); no config uses it")
    assert bsdf == "diffuse", "Invalid BSDF '%s'" % bsdf
    return bsdf


def _collect(render_modes, shaded_col, kd, ks, normal, geo_normal, tangent, shading, flow, dino_pred, depth):
    """The buffer dict of the reference (render.py:110-125), restricted to what was asked for."""
    wanted = set(render_modes) if render_modes is not None else {"shaded"}
    buffers = {"shaded": shaded_col}
    if "kd" in wanted:
        buffers["kd"] = kd
    if "ks" in wanted:
        buffers["ks"] = ks
    if "normal" in wanted:
        buffers["normal"] = (normal + 1.0) * 0.5
    if "geo_normal" in wanted:
        buffers["geo_normal"] = (geo_normal + 1.0) * 0.5
    if "tangent" in wanted and tangent is not None:
        buffers["tangent"] = (tangent + 1.0) * 0.5
    if shading is not None:
        buffers["shading"] = shading
    if flow is not None:
        buffers["flow"] = flow
    if dino_pred is not None:
        buffers["dino_pred"] = dino_pred
    if depth is not None:
        buffers["depth"] = depth
    return buffers


def _shade_covered(gb_pos, gb_geometric_normal, gb_normal, gb_tangent, gb_tex_pos, w2c, view_pos, lgt, material, bsdf, feat, render_modes,
                   two_sided_shading, delta_xy_interp, dino_net, class_vector, cover):
    """shade() on the covered pixels only, from dense G-buffers (generic path: any mode, any attribute set)."""
    b, h, w, _ = gb_pos.shape
    pix = torch.nonzero(cover.reshape(-1)).squeeze(1)  # [P]; one host sync for P
    take = lambda t: None if t is None else t.reshape(b * h * w, t.shape[-1]).index_select(0, pix)
    pos, geo, nrm, tng, tex_pos, flow = map(take, (gb_pos, gb_geometric_normal, gb_normal, gb_tangent, gb_tex_pos, delta_xy_interp))
    return _shade_points(pos, geo, nrm, tng, tex_pos, flow, pix, (b, h, w), w2c, view_pos, lgt, material, bsdf, feat, render_modes,
                         two_sided_shading, dino_net, class_vector)


class SparseBuffers(dict):
    """mode -> [>= P,C] values at the covered pixels ``pix`` (flat indices into [B,H,W]); what the fused path hands to the compositor.
    A value may carry more rows than the list (the fields' point list is padded to a bucket size; the padding rows are never read):
    ``self[mode]`` returns the P rows, ``peek(mode)`` the tensor as it is (what the compositor takes, so that no row slice -- and no
    zero-padded gradient of one -- stands between the fields and the compositor)."""

    def __init__(self, pix, bhw, inv=None):
        super().__init__()
        self.pix, self.bhw, self.inv = pix, bhw, inv  # inv: pixel -> row (int32 [B*H*W], -1 = uncovered) when the list came with it
        self.shade_recipe = None  # ops.ShadeRecipe: self['shaded'] has not been computed (the fused compositor does it on the fly)

    def __contains__(self, mode):
        return super().__contains__(mode) or (mode == "shaded" and self.shade_recipe is not None)

    def __getitem__(self, mode):
        # a caller that indexes the colour directly (outside render_mesh, which hands the recipe to the compositor) gets the stand-alone
        # shading launch's result
        if mode == "shaded" and self.shade_recipe is not None and not super().__contains__(mode):
            super().__setitem__(mode, self.shade_recipe.materialize()[2])
        vals = super().__getitem__(mode)
        return vals if vals.shape[0] == self.pix.shape[0] else vals[: self.pix.shape[0]]

    def peek(self, mode):
        """The stored tensor (possibly with padding rows), without materialising a deferred colour."""
        return super().__getitem__(mode)

    def dense(self, mode):
        """[B,H,W,C+1] with alpha 1 on covered pixels, zeros elsewhere (the layout render_layer returns in the reference)."""
        b, h, w = self.bhw
        vals = self[mode]
        out = torch.zeros(b * h * w, vals.shape[-1] + 1, dtype=vals.dtype, device=vals.device)
        return out.index_copy(0, self.pix, torch.cat((vals, torch.ones_like(vals[:, :1])), dim=-1)).view(b, h, w, -1)


def _shade_points(pos, geo, nrm, tng, tex_pos, flow, pix, bhw, w2c, view_pos, lgt, material, bsdf, feat, render_modes, two_sided_shading,
                  dino_net, class_vector, sparse=False, gb=None, inv=None, tex_in=None, img_p=None):
    """The arithmetic of shade() (reference render.py:30-132) on compact [P,.] arrays; scatters into dense [B,H,W,C+1]
    buffers (zeros, alpha 0, where nothing was rasterised).  ``pix`` = flat pixel indices of the P points.
    ``tex_in`` [Pp,3] / ``img_p`` [Pp] (from the G-buffer launch, ops.covered_gbuffer(field_inputs=...)): the fields' input rows and
    point -> image index, already padded to the bucket size; without them they are derived here with torch ops."""
    b, h, w = bhw
    dev = pix.device
    n_pts = pix.shape[0]
    # The two coordinate MLPs see the point list padded to a multiple of POINT_BUCKET rows (zeros; outputs sliced off again):
    # the GEMM shapes then repeat from step to step, which is what rocBLAS/hipBLASLt kernel selection and TunableOp key on.
    if tex_in is not None:
        tex_rows, img_rows = tex_in, img_p
        n_pad = tex_in.shape[0] - n_pts
    else:
        img_rows = torch.div(pix, h * w, rounding_mode="floor")
        n_pad = (-n_pts) % POINT_BUCKET if POINT_BUCKET else 0
        tex_rows = tex_pos
        if n_pad:
            img_rows = torch.cat((img_rows, img_rows.new_full((n_pad,), b - 1)))  # padding rows ride with the last image: the index stays non-decreasing
            tex_rows = torch.nn.functional.pad(tex_pos, (0, 0, 0, n_pad))
    img = img_rows if n_pad == 0 else img_rows[:n_pts]
    _trim_allocator_cache(n_pts + n_pad, pix.device)

    def field(net, x, f):
        # networks that understand (per-image rows, point -> image index) get that instead of a [P,C] per-point copy of the feature
        if f is not None and f.shape[0] == b and getattr(net, "indexed_feat", False):
            return net.sample(x, feat=f, feat_index=img_rows)
        return net.sample(x, feat=None if f is None else (_rows_per_point(f, img_rows, b) if f.shape[0] == b else f.expand(n_pts + n_pad, -1)))

    # (the fields' outputs keep their padding rows: consumers that need exactly P rows slice, the fused compositor does not)
    all_tex = field(material, tex_rows, feat) if material is not None else torch.ones(n_pts, 9, device=dev)
    dino_rows = field(dino_net, tex_rows, class_vector) if dino_net is not None else None

    _resolve_bsdf(bsdf, material)
    if lgt is not None and isinstance(lgt, light.EnvironmentLight):
        raise NotImplementedError("EnvironmentLight is outside the hot path")
    modes = render_modes if render_modes is not None else ["shaded"]
    view = view_pos.reshape(-1, 3)
    light_rows = lgt(feat) if lgt is not None else None  # DirectionalLight.forward: [B,5] = direction(3), ambient, diffuse (light.py:176-184)

    # ---- the fused training path: nothing is computed here.  The compositor computes the colour on the fly and its backward node runs
    # the shading adjoint (ops.ShadeRecipe / ops.shade_composite_antialias): no [B,17] table, no kd / ks / row slices, no launch.
    if (gb is not None and FUSED_SHADING and SHADE_IN_COMPOSITOR and sparse and inv is not None and lgt is not None and material is not None
            and w2c.dim() == 3 and all(m in ("shaded", "dino_pred", "flow", "kd", "ks") for m in modes)):
        recipe = ops.ShadeRecipe(gb, w2c, view, light_rows, all_tex, two_sided_shading, img_rows)
        LAST_POINTS[0] = dict(pix=pix, gb=gb.detach(), all_tex=all_tex.detach(), dino=None if dino_rows is None else dino_rows.detach(),
                              light=light_rows.detach(), flow=None if flow is None else flow.detach())
        out = SparseBuffers(pix, (b, h, w), inv)
        out.shade_recipe = recipe
        lazy = {"dino_pred": dino_rows, "flow": flow, "kd": lambda: all_tex[:n_pts, :3], "ks": lambda: all_tex[:n_pts, 3:6]}
        for mode in modes:
            if mode == "shaded":
                continue
            val = lazy[mode]
            if val is None:
                raise KeyError(mode)  # like the reference: a mode whose buffer was never produced (render.py:127-128)
            out[mode] = val() if callable(val) else val
        return out

    all_tex = all_tex[:n_pts] if n_pad else all_tex
    kd, ks = all_tex[..., :3], all_tex[..., 3:6]
    dino_pred = None if dino_rows is None else (dino_rows[:n_pts] if n_pad else dino_rows)
    # the narrow per-image quantities (camera rotation 9, view position 3, light parameters 5) travel to the points as ONE gather
    cols = [w2c[:, :3, :3].reshape(-1, 9).expand(b, 9), view.expand(b, 3)]
    if lgt is not None:
        cols.append(light_rows)
    per_image = torch.cat(cols, dim=-1)  # [B, 12 | 17]
    shading = None
    if gb is not None and FUSED_SHADING:  # one HIP kernel each way for the ~30 (+~70 backward) elementwise launches below; the kernels
        # read the image's row through the point -> image index (no [P,17] copy) and reduce its gradient per image themselves
        if lgt is None:
            nrm, shaded_col = ops.shade_points(gb, per_image, None, two_sided_shading, img=img), kd
        else:
            nrm, shading, shaded_col = ops.shade_points(gb, per_image, kd, two_sided_shading, img=img)
    else:
        per_point = _rows_per_point(per_image, img, b)
        rot, view_p = per_point[:, 0:9].reshape(-1, 3, 3), per_point[:, 9:12]
        nrm = ru.prepare_shading_normal(pos, view_p, None, nrm, tng, geo, two_sided_shading=two_sided_shading, opengl=True, use_python=True)
        cam_normal = util.safe_normalize((rot * nrm[:, None, :]).sum(-1))  # per-point 3x3 . 3 as elementwise work, not P tiny GEMMs
        if lgt is None:
            shaded_col = kd
        else:
            params = per_point[:, 12:17]
            shading = params[:, 3:4] + params[:, 4:5] * torch.clamp(util.dot(params[:, :3], cam_normal), min=0.0)
            shaded_col = shading * kd

    if gb is not None:  # references only (no copies): the stage-wise checkers re-do every stage from the previous stage's HIP output
        LAST_POINTS[0] = dict(pix=pix, gb=gb.detach(), all_tex=all_tex.detach() if material is not None else None,
                              dino=None if dino_pred is None else dino_pred.detach(), light=None if light_rows is None else light_rows.detach(),
                              flow=None if flow is None else flow.detach())
    buffers = _collect(render_modes, shaded_col, kd, ks, nrm, geo, tng, shading, flow, dino_pred, None)
    out = SparseBuffers(pix, (b, h, w), inv)
    for mode in modes:
        out[mode] = buffers[mode]  # KeyError for an unknown / unavailable mode, like the reference (render.py:127-128)
    return out if sparse else {mode: out.dense(mode) for mode in out}


def _rows_per_point(t, img, b):
    return ops.rows_per_point(t, img)


PIXEL_TILE = 8  # covered pixels are listed tile by tile (0 = plain row-major order)


def _covered_pixels(rast):
    """Flat indices (b*H + y)*W + x of the covered pixels, image-major and, inside an image, in 8x8-tile order: consecutive
    entries then touch the same few triangles, which is what the G-buffer kernels' gathers and LDS scatter aggregation like.
    Ballot/prefix-sum compaction on the GPU (csrc/cover.hip)."""
    return ops.covered_pixels(rast, tile=PIXEL_TILE)


FUSED_GBUFFER_MODES = frozenset(("shaded", "kd", "ks", "normal", "geo_normal", "shading", "dino_pred", "flow"))


def render_layer(rast, rast_deriv, mesh, w2c, view_pos, material, lgt, resolution, spp, msaa, bsdf, feat, render_modes=None, prior_mesh=None,
                 two_sided_shading=True, delta_xy=None, dino_net=None, class_vector=None, clip=None, sparse=False, v_pos_attr=None):
    """G-buffer interpolation + shading of one depth layer (reference render.py:139-221).

    ``clip`` (the [B,V,4] clip-space vertices, not in the reference signature) enables the fused path: one HIP kernel builds
    the G-buffer of the covered pixels and its backward also carries the rasteriser's gradient (csrc/gbuffer.hip).  Without
    it, or for modes that need other attributes (flow, tangent, depth) or spp > 1, the generic interpolate path runs.
    """
    full_res = [resolution[0] * spp, resolution[1] * spp]
    if prior_mesh is None:
        prior_mesh = mesh
    render_modes = render_modes if render_modes is not None else ["shaded"]
    tri = mesh.t_pos_idx[0]
    assert mesh.v_nrm is not None

    fused = (clip is not None and SHADE_COVERED_ONLY and spp == 1 and not ({"tangent", "depth"} & set(render_modes))
             and clip.shape[0] == mesh.v_pos.shape[0] and mesh.t_nrm_idx.data_ptr() == mesh.t_pos_idx.data_ptr())
    if fused:
        b, h, w = rast.shape[:3]
        flow = None
        flow_fused = "flow" in render_modes and delta_xy.shape[-1] <= 3  # the one extra attribute of the sequence models rides in the same kernels
        if FUSED_COVER_GBUFFER and PIXEL_TILE == 8 and h % 8 == 0 and w % 8 == 0:
            # the covered-pixel list and its G-buffer rows from ONE launch (one host sync before it: the number of covered pixels)
            # (... and what the fields take from it -- canonical positions as dense rows, point -> image index, padded -- from the same launch)
            res = ops.covered_gbuffer(clip, mesh.v_pos if v_pos_attr is None else v_pos_attr, mesh.v_nrm, prior_mesh.v_pos, rast, tri,
                                      extra=delta_xy if flow_fused else None,
                                      field_inputs=POINT_BUCKET if FIELD_INPUTS_FROM_GBUFFER else None)
            (gb, flow, pix, inv), rest = (res[:4], res[4:]) if flow_fused else ((res[0], None, res[1], res[2]), res[3:])
            tex_in, img_p = rest if rest else (None, None)
        else:
            tex_in = img_p = None
            pix, inv = ops.covered_pixels(rast, tile=PIXEL_TILE, return_inverse=True)  # one host sync for the number of covered pixels
            if flow_fused:
                gb, flow = ops.gbuffer(clip, mesh.v_pos, mesh.v_nrm, prior_mesh.v_pos, rast, tri, pix, extra=delta_xy)  # [P,12], [P,2]
            else:
                gb = ops.gbuffer(clip, mesh.v_pos, mesh.v_nrm, prior_mesh.v_pos, rast, tri, pix)  # [P,12]
        if "flow" in render_modes and not flow_fused:
            flow = interpolate(delta_xy, rast, tri)[0].reshape(b * h * w, -1).index_select(0, pix)
        return _shade_points(gb[:, 0:3], gb[:, 3:6], gb[:, 6:9], None, gb[:, 9:12], flow, pix, (b, h, w), w2c, view_pos, lgt, material, bsdf, feat,
                             render_modes, two_sided_shading, dino_net, class_vector, sparse=sparse, gb=gb, inv=inv, tex_in=tex_in, img_p=img_p)

    rast_s = util.scale_img_nhwc(rast, resolution, mag="nearest", min="nearest") if (spp > 1 and msaa) else rast
    gb_pos, _ = interpolate(mesh.v_pos, rast_s, tri)
    v0, v1, v2 = mesh.v_pos[:, tri[:, 0]], mesh.v_pos[:, tri[:, 1]], mesh.v_pos[:, tri[:, 2]]
    face_normals = util.safe_normalize(torch.cross(v1 - v0, v2 - v0, dim=-1))
    gb_geometric_normal, _ = interpolate(face_normals, rast_s, _face_index_buffer(tri))
    gb_normal, _ = interpolate(mesh.v_nrm, rast_s, mesh.t_nrm_idx[0])
    gb_tangent = None
    if "tangent" in render_modes:  # only then are tangents ever observable (perturbed_nrm is None, render.py:71)
        gb_tangent, _ = interpolate(mesh.v_tng, rast_s, mesh.t_tng_idx[0])
    delta_xy_interp = interpolate(delta_xy, rast_s, tri)[0] if "flow" in render_modes else None
    gb_tex_pos, _ = interpolate(prior_mesh.v_pos, rast_s, tri)  # canonical position = texture coordinate (render.py:209)

    buffers = shade(gb_pos, gb_geometric_normal, gb_normal, gb_tangent, gb_tex_pos, w2c, view_pos, lgt, material, bsdf, feat=feat,
                    render_modes=render_modes, two_sided_shading=two_sided_shading, delta_xy_interp=delta_xy_interp, dino_net=dino_net,
                    class_vector=class_vector, cover=(rast_s[..., 3] > 0) if SHADE_COVERED_ONLY else None)
    if spp > 1 and msaa:
        for key in buffers.keys():
            buffers[key] = util.scale_img_nhwc(buffers[key], full_res, mag="nearest", min="nearest")
    return buffers


_face_idx_cache = ops._IdentityCache()


def _face_index_buffer(tri):
    """[[f,f,f]] int32 so a per-face attribute can go through interpolate (reference render.py:190); cached per topology."""
    return _face_idx_cache.get(tri, lambda t: torch.arange(t.shape[0], dtype=torch.int32, device=t.device)[:, None].repeat(1, 3).contiguous())


def render_mesh(ctx, mesh, mtx_in, w2c, view_pos, material, lgt, resolution, spp=1, num_layers=1, msaa=False, background=None, bsdf=None,
                feat=None, render_modes=None, prior_mesh=None, two_sided_shading=True, dino_net=None, num_frames=None, class_vector=None):
    """Rasterise, shade, composite over the background and antialias (reference render.py:228-337).

    ``ctx`` is accepted for signature compatibility (dr.RasterizeGLContext() in the reference, AnimalModel.py:235-236)
    and unused: the HIP rasteriser keeps no context.
    """
    assert mesh.t_pos_idx.shape[1] > 0, "Got empty training triangle mesh (unrecoverable discontinuity)"
    assert background is None or (background.shape[1] == resolution[0] and background.shape[2] == resolution[1])
    render_modes = render_modes if render_modes is not None else ["shaded"]
    dev = mesh.v_pos.device
    full_res = [resolution[0] * spp, resolution[1] * spp]
    mtx_in = torch.tensor(mtx_in, dtype=torch.float32, device=dev) if not torch.is_tensor(mtx_in) else mtx_in
    view_pos = torch.tensor(view_pos, dtype=torch.float32, device=dev) if not torch.is_tensor(view_pos) else view_pos
    view_pos = view_pos[:, None, None, :] if view_pos.dim() == 2 else view_pos

    v_pos_attr = v_pos_nrm = clip_aa = None
    if ALIAS_POSITIONS and spp == 1 and num_layers == 1 and mesh.v_pos.is_cuda and ru.ops._hip_xfm_ok(mesh.v_pos, mtx_in) and mesh.v_pos.shape[0] == mtx_in.shape[0] \
            and not torch.is_autocast_enabled():
        # the clip transform's node hands the positions out once more for the G-buffer's position attribute: their two gradients (through
        # clip space, through the interpolated position) meet inside that node's ONE backward launch, not in an accumulation kernel
        # (... and a third time for the pending vertex normals, whose backward is the positions' third gradient)
        # (... and the clip positions themselves a second time, for the antialiasing, whose gradient w.r.t. them is their second one)
        v_pos_clip, v_pos_attr, v_pos_nrm, clip_aa = ops.xfm_points(mesh.v_pos, mtx_in, alias=3)  # [B,V,4]
    else:
        v_pos_clip = ru.xfm_points(mesh.v_pos, mtx_in, use_python=True)  # [B,V,4]

    delta_xy = None
    if "flow" in render_modes and FUSED_FLOW_DELTA and v_pos_clip.is_cuda and v_pos_clip.dtype == torch.float32 and num_frames > 1 and v_pos_clip.shape[0] % num_frames == 0 \
            and not torch.is_autocast_enabled():
        delta_xy = ops.flow_delta(v_pos_clip, num_frames)  # the expression below as ONE launch each way (a3d_flow_delta_*)
    elif "flow" in render_modes:  # 2-D motion of each vertex to the next frame (render.py:281-288)
        ndc = v_pos_clip[..., :2] / v_pos_clip[..., -1:]
        ndc = ndc.view(-1, num_frames, *ndc.shape[1:])
        delta_xy = ndc[:, 1:] - ndc[:, :-1]
        delta_xy = torch.cat([delta_xy, torch.zeros_like(delta_xy[:, :1])], dim=1).view(-1, *ndc.shape[2:])

    tri = mesh.t_pos_idx[0]
    clip_f = v_pos_clip.float()
    if num_layers != 1:  # depth peeling: never used by a config (AnimalModel.py:247); the general, dense path
        return _render_mesh_layers(mesh, clip_f, tri, w2c, view_pos, material, lgt, resolution, spp, num_layers, msaa, background, bsdf, feat,
                                   render_modes, prior_mesh, two_sided_shading, dino_net, class_vector, delta_xy)
    job = mesh.normals_job() if hasattr(mesh, "normals_job") else None  # pending auto_normals: extra work-groups of the triangle launch
    mask_only = (FUSED_MASK_RENDER and FUSED_COMPOSITE and material is None and lgt is None and dino_net is None and list(render_modes) == ["shaded"]
                 and spp == 1 and (background is None or not background.requires_grad))
    # when render_layer's fused path is certain to come next, the rasteriser's resolve waits for it: ONE launch then writes the texels,
    # the covered-pixel list and the G-buffer rows (ops.DEFER_RESOLVE; the conditions are render_layer's own)
    defer = (DEFER_RESOLVE and not mask_only and FUSED_GBUFFER and FUSED_COVER_GBUFFER and SHADE_COVERED_ONLY and PIXEL_TILE == 8 and spp == 1
             and not ({"tangent", "depth"} & set(render_modes)) and clip_f.shape[0] == mesh.v_pos.shape[0]
             and mesh.t_nrm_idx.data_ptr() == mesh.t_pos_idx.data_ptr() and full_res[0] % 8 == 0 and full_res[1] % 8 == 0)
    rast = ops.rasterize(clip_f, tri, full_res, normals_job=job, defer_resolve=defer)
    if job is not None:
        mesh.take_normals(job, graph_pos=v_pos_nrm)
    LAST_RAST[0] = rast.detach()  # introspection hook for benchmarks / debugging (coverage, ids)
    LAST_POINTS[0] = None
    if mask_only:
        _resolve_bsdf(bsdf, material)  # the reference's asserts come first (render.py:79,85): bsdf None without a material, 'pbr' without a light
        # shade() gives every covered pixel kd = (1,1,1) here (render.py:57-60, :84-85 with lgt None): the image is the coverage, and the only
        # gradient is the silhouette's.  Fauna's random-view mask (Fauna.py:111-173).  No covered-pixel list, no G-buffer, no read-back.
        bg = None if background is None else torch.cat((background, torch.zeros_like(background[..., 0:1])), dim=-1)
        tri32 = ops.tri_int32(tri)
        analysis = ops.AAAnalysis(rast, clip_f, ops.aa_topology(tri32, clip_f.shape[1]), defer=DEFER_ANALYSIS)
        LAST_POINTS[0] = dict(clip=clip_f.detach())
        return [ops.mask_antialias(rast, clip_f, bg, analysis, 3).permute(0, 3, 1, 2)]
    try:
        rendered = render_layer(rast, None, mesh, w2c, view_pos, material, lgt, resolution, spp, msaa, bsdf, feat=feat, render_modes=render_modes,
                                prior_mesh=prior_mesh, two_sided_shading=two_sided_shading, delta_xy=delta_xy, dino_net=dino_net,
                                class_vector=class_vector, clip=clip_f if FUSED_GBUFFER else None, sparse=True, v_pos_attr=v_pos_attr)
    except BaseException:
        if defer:  # the consumer the deferred resolve was promised failed: nothing may be left pending (keys pinned, texels unwritten)
            ops.drop_pending_resolve(rast)
        raise

    if background is not None and spp > 1:
        background = util.scale_img_nhwc(background, full_res, mag="nearest", min="nearest")
    # the reference appends a zero alpha channel to the background on every call (render.py:254-256): the fused compositor reads the
    # 3-channel tensor as it is (a3d_ca_buffer.bg_channels); only the torch branches further down build the 4-channel copy
    bg4 = [None]

    def background4():
        if bg4[0] is None:
            bg4[0] = (torch.cat((background, torch.zeros_like(background[..., 0:1])), dim=-1) if background is not None
                      else torch.zeros(1, full_res[0], full_res[1], 4, dtype=torch.float32, device=dev))
        return bg4[0]

    analysis = None
    # composite + antialias fused (csrc/antialias.hip): one pass over the image per buffer, same values as the two steps further down;
    # the buffers go through the op two at a time (the colour and the feature image of a training step share its launches)
    fused = {}
    can_fuse = isinstance(rendered, SparseBuffers) and rendered.inv is not None and FUSED_COMPOSITE and (background is None or not background.requires_grad)
    fuse_keys = [k for k in dict.fromkeys(render_modes) if can_fuse and k in rendered and k in ANTIALIASED_MODES]
    if fuse_keys:
        tri32 = ops.tri_int32(tri)
        analysis = ops.AAAnalysis(rast, clip_f, ops.aa_topology(tri32, clip_f.shape[1]), defer=DEFER_ANALYSIS)  # runs inside the first compositor call

        def bg_of(k):
            if k not in ("shaded", "geo_normal", "shading") or background is None:
                return None
            return background4()[..., 2:] if k == "shading" else background  # ('shading': the last two channels of the 4-channel background)

        # channels handed out per mode (render.py:320-331): the op returns the view, so no slice (and no padded gradient of one) follows it
        keep_of = {"flow": 2, "dino_pred": -1, "shading": 1, "depth": 1}

        def keep(k, vals):
            n = keep_of.get(k)
            return None if n is None else (vals.shape[-1] if n == -1 else n)

        recipe = getattr(rendered, "shade_recipe", None)
        # (the antialiasing differentiates w.r.t. the clip positions: it takes them as the clip transform's OTHER clip output, same storage)
        clip_of_aa = clip_f if clip_aa is None or clip_aa.dtype != clip_f.dtype else clip_aa
        for i in range(0, len(fuse_keys), 2):
            ka, kb = fuse_keys[i], (fuse_keys[i + 1] if i + 1 < len(fuse_keys) else None)
            vb = rendered.peek(kb) if (kb is not None and not (kb == "shaded" and recipe is not None)) else (rendered[kb] if kb is not None else None)
            if ka == "shaded" and recipe is not None and not dict.__contains__(rendered, "shaded"):  # (only the FIRST buffer of a call can be shaded on the fly)
                res = ops.shade_composite_antialias(recipe, rendered.pix, rendered.inv, bg_of(ka), clip_of_aa, analysis, vals2=vb,
                                                    background2=None if kb is None else bg_of(kb), keep2=None if kb is None else keep(kb, vb))
            else:
                va = rendered.peek(ka) if dict.__contains__(rendered, ka) else rendered[ka]
                res = ops.composite_antialias(va, rendered.pix, rendered.inv, bg_of(ka), clip_of_aa, analysis, vals2=vb,
                                              background2=None if kb is None else bg_of(kb), keep=keep(ka, va), keep2=None if kb is None else keep(kb, vb))
            if kb is None:
                fused[ka] = res
            else:
                fused[ka], fused[kb] = res
    if LAST_POINTS[0] is not None:
        LAST_POINTS[0]["clip"] = clip_f.detach()
    out_buffers = []
    for key in render_modes:
        if key not in rendered:
            out_buffers.append(None)
            continue
        fused_aa = key in fused
        if fused_aa:  # (already antialiased, and already cut to the channels the mode returns)
            out_buffers.append(fused[key].permute(0, 3, 1, 2))
            continue
        background = background4()
        coverage = (rast[..., -1:] > 0).float()
        if isinstance(rendered, SparseBuffers):
            # coverage is 0 or 1, for which lerp(bg, [rgb,1], alpha) (render.py:261-262) returns exactly bg or exactly [rgb,1]:
            # start from the background and overwrite the covered pixels -- same bits, a third of the passes over the image
            vals = rendered[key]
            c1 = vals.shape[-1] + 1
            if key in ("shaded", "geo_normal", "shading"):
                bgk = background[..., 2:] if (key == "shading" and background.shape[-1] == 4) else background
                accum = bgk.expand(rast.shape[0], -1, -1, -1).clone()
            else:
                accum = torch.zeros(*rast.shape[:3], c1, dtype=torch.float32, device=dev)
            accum.view(-1, c1).index_copy_(0, rendered.pix, torch.cat((vals, torch.ones_like(vals[:, :1])), dim=-1))
        else:
            buf = rendered[key]
            bg = background if key in ("shaded", "geo_normal", "shading") else torch.zeros_like(buf)
            if key == "shading" and bg.shape[-1] == 4:
                bg = bg[..., 2:]
            alpha = coverage * buf[..., -1:]
            accum = torch.lerp(bg, torch.cat((buf[..., :-1], torch.ones_like(buf[..., -1:])), dim=-1), alpha)  # render.py:261-262
        if key in ANTIALIASED_MODES:
            if analysis is None:
                tri32 = ops.tri_int32(tri)
                analysis = ops.AAAnalysis(rast, clip_f, ops.aa_topology(tri32, clip_f.shape[1]))
            accum = ops.antialias(accum.contiguous().float(), rast, clip_f, tri, analysis=analysis)
        out = util.avg_pool_nhwc(accum, spp) if spp > 1 else accum
        if key in ("kd", "ks", "normal", "geo_normal", "tangent"):
            out = out[..., :3]
        elif key in ("shading", "depth"):
            out = out[..., :1]
        elif key == "flow":
            out = out[..., :2]
        elif key == "dino_pred":
            out = out[..., :-1]
        out_buffers.append(out.permute(0, 3, 1, 2))
    return out_buffers


def _slice_mode(key, out):
    """Channel selection per render mode (reference render.py:320-331)."""
    if key in ("kd", "ks", "normal", "geo_normal", "tangent"):
        return out[..., :3]
    if key in ("shading", "depth"):
        return out[..., :1]
    if key == "flow":
        return out[..., :2]
    if key == "dino_pred":
        return out[..., :-1]
    return out


def _render_mesh_layers(mesh, clip_f, tri, w2c, view_pos, material, lgt, resolution, spp, num_layers, msaa, background, bsdf, feat, render_modes,
                        prior_mesh, two_sided_shading, dino_net, class_vector, delta_xy):
    """render_mesh with ``num_layers`` depth layers (reference render.py:258-268, 290-337): every layer is rasterised behind the previous
    one (DepthPeeler), shaded, and the layers are composited back to front, antialiasing after each layer with that layer's geometry."""
    dev = clip_f.device
    full_res = [resolution[0] * spp, resolution[1] * spp]
    layers, prev = [], None
    for _ in range(num_layers):
        rast = ops.rasterize(clip_f, tri, full_res, prev=prev)
        prev = rast.detach()
        rendered = render_layer(rast, None, mesh, w2c, view_pos, material, lgt, resolution, spp, msaa, bsdf, feat=feat, render_modes=render_modes,
                                prior_mesh=prior_mesh, two_sided_shading=two_sided_shading, delta_xy=delta_xy, dino_net=dino_net,
                                class_vector=class_vector, clip=None, sparse=False)
        layers.append((rendered, rast, ops.AAAnalysis(rast, clip_f, ops.aa_topology(ops.tri_int32(tri), clip_f.shape[1]))))
    LAST_RAST[0] = layers[0][1].detach()
    if background is not None:
        if spp > 1:
            background = util.scale_img_nhwc(background, full_res, mag="nearest", min="nearest")
        background = torch.cat((background, torch.zeros_like(background[..., 0:1])), dim=-1)
    else:
        background = torch.zeros(1, full_res[0], full_res[1], 4, dtype=torch.float32, device=dev)
    out_buffers = []
    for key in render_modes:
        if key not in layers[0][0]:
            out_buffers.append(None)
            continue
        accum = background if key in ("shaded", "geo_normal", "shading") else torch.zeros_like(layers[0][0][key])
        if key == "shading" and accum.shape[-1] == 4:
            accum = accum[..., 2:]
        for buffers, rast, analysis in reversed(layers):
            buf = buffers[key]
            alpha = (rast[..., -1:] > 0).float() * buf[..., -1:]
            accum = torch.lerp(accum, torch.cat((buf[..., :-1], torch.ones_like(buf[..., -1:])), dim=-1), alpha)
            if key in ANTIALIASED_MODES:
                accum = ops.antialias(accum.contiguous().float(), rast, clip_f, tri, analysis=analysis)
        out = util.avg_pool_nhwc(accum, spp) if spp > 1 else accum
        out_buffers.append(_slice_mode(key, out).permute(0, 3, 1, 2))
    return out_buffers


def render_uv(ctx, mesh, resolution, mlp_texture, feat=None):
    """Texture-space bake used by the OBJ/MTL export (reference render.py:342-360)."""
    uv_clip = mesh.v_tex * 2.0 - 1.0
    uv_clip4 = torch.cat((uv_clip, torch.zeros_like(uv_clip[..., 0:1]), torch.ones_like(uv_clip[..., 0:1])), dim=-1)
    rast = ops.rasterize(uv_clip4.contiguous(), mesh.t_tex_idx[0], resolution)
    gb_pos, _ = interpolate(mesh.v_pos, rast, mesh.t_pos_idx[0])
    all_tex = mlp_texture.sample(gb_pos, feat=feat)
    assert all_tex.shape[-1] in (9, 10), "Combined kd_ks_normal must be 9 or 10 channels"
    return (rast[..., -1:] > 0).float(), all_tex[..., :-6], all_tex[..., -6:-3], util.safe_normalize(all_tex[..., -3:])
