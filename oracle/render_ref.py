"""Oracle: render_mesh / render_layer / shade, restating /root/reference/model/render/render.py.

TEST INFRASTRUCTURE ONLY.  shade() and the clip transform are pinned against goldens captured from
the imported reference (tests/golden/shade_*.npz, xfm_*.npz); the three nvdiffrast operators inside
come from oracle/raster_ref.py (parity unpinned).  All tensors CPU, float32.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import mesh_ref, raster_ref

NORMAL_THRESHOLD = 0.1  # reference renderutils/bsdf.py:13


def xfm_points(points, matrix):
    """reference renderutils/ops.py:524-525: pad(points,1) @ matrix^T -> [B,V,4]."""
    return torch.matmul(F.pad(points, (0, 1), value=1.0), matrix.transpose(1, 2))


def _dot(a, b):
    return (a * b).sum(-1, keepdim=True)


def shading_normal(pos, view_pos, smooth_nrm, geom_nrm, two_sided=True):
    """prepare_shading_normal with perturbed_nrm=None (reference render.py:71-72, bsdf.py:28-51).

    With the (0,0,1) perturbation the tangent frame drops out: shading normal == normalised smooth
    normal; what remains is the two-sided flip and the bend toward the geometric normal.
    """
    n = F.normalize(smooth_nrm, dim=-1)
    view = F.normalize(view_pos - pos, dim=-1)
    # _perturb_normal with (0,0,1): tng*0 - bitng*0 + n*1, re-normalised (bsdf.py:38-44)
    n = F.normalize(n, dim=-1)
    g = geom_nrm
    if two_sided:
        front = _dot(g, view) > 0
        n = torch.where(front, n, -n)
        g = torch.where(front, g, -g)
    t = torch.clamp(_dot(view, n) / NORMAL_THRESHOLD, 0, 1)
    return torch.lerp(g, n, t)


def shade(gb_pos, gb_geo_nrm, gb_nrm, gb_tex_pos, w2c, view_pos, lgt, material, feat=None, render_modes=("shaded",),
          two_sided=True, flow=None, dino_net=None, class_vector=None):
    """reference render.py:30-132 for bsdf='diffuse' (the only one any config uses)."""
    if material is not None:
        all_tex = material.sample(gb_tex_pos, feat=feat)
    else:
        all_tex = torch.ones(*gb_pos.shape[:-1], 9)
    kd, ks = all_tex[..., :3], all_tex[..., 3:6]
    dino = dino_net.sample(gb_tex_pos, feat=class_vector) if dino_net is not None else None
    alpha = torch.ones_like(kd[..., :1])
    nrm = shading_normal(gb_pos, view_pos, gb_nrm, gb_geo_nrm, two_sided)
    b, h, w, _ = nrm.shape
    cam_nrm = mesh_ref.safe_normalize(torch.matmul(nrm.view(b, -1, 3), w2c[:, :3, :3].transpose(2, 1))).view(b, h, w, 3)
    shading = None
    if lgt is None:
        shaded = kd
    else:
        shaded, shading = lgt.shade(feat, kd, cam_nrm)
    depth = None
    if "depth" in render_modes:
        hom = torch.cat([gb_pos, torch.ones_like(gb_pos[..., :1])], -1)
        cam = torch.matmul(hom.view(b, -1, 4), w2c.transpose(-1, -2)).view(b, h, w, 4)
        depth = cam[..., 2]
        dmin, dmax = depth.amin((1, 2), keepdim=True), depth.amax((1, 2), keepdim=True)
        depth = ((depth - dmin) / (dmax - dmin)).unsqueeze(-1)
    buffers = {"shaded": shaded, "kd": kd, "ks": ks, "normal": (nrm + 1) * 0.5, "geo_normal": (gb_geo_nrm + 1) * 0.5}
    if shading is not None:
        buffers["shading"] = shading
    if flow is not None:
        buffers["flow"] = flow
    if dino is not None:
        buffers["dino_pred"] = dino
    if depth is not None:
        buffers["depth"] = depth
    return {m: torch.cat((buffers[m], alpha), -1) for m in render_modes}  # KeyError for unknown modes (render.py:127-128)


def render_mesh(v_pos, faces, v_nrm, mtx, w2c, view_pos, material, lgt, resolution, background=None, feat=None,
                render_modes=("shaded",), prior_v_pos=None, two_sided=True, dino_net=None, num_frames=None,
                class_vector=None, clip=None, taps=None):
    """reference render.py:228-337 with spp=1, num_layers=1 (the only setting used, AnimalModel.py:245-248).

    v_pos [B,V,3], faces [F,3] int64, v_nrm [B,V,3]; returns the list of NCHW buffers in render_modes order.
    Checker hooks (oracle/check.py): ``clip`` replaces the clip transform's result (stage-wise parity: the next stage starts from
    the HIP stage's own output); ``taps`` (a dict) receives the intermediate buffers.
    """
    assert faces.shape[0] > 0, "Got empty training triangle mesh (unrecoverable discontinuity)"
    H, W = resolution
    view_pos = view_pos[:, None, None, :] if view_pos.dim() == 2 else view_pos
    clip = xfm_points(v_pos, mtx) if clip is None else clip  # :278
    flow_attr = None
    if "flow" in render_modes:  # :281-288
        ndc = clip[..., :2] / clip[..., -1:]
        ndc = ndc.view(-1, num_frames, *ndc.shape[1:])
        dxy = ndc[:, 1:] - ndc[:, :-1]
        dxy = torch.cat([dxy, torch.zeros_like(dxy[:, :1])], 1)
        flow_attr = dxy.view(-1, *dxy.shape[2:])
    tri = faces.int()
    rast = raster_ref.rasterize(clip, tri, (H, W))  # :292-294
    # values: the rasteriser's own stored (u, v) -- what dr.interpolate reads; gradients: d(u, v)/d(clip) of the same triangle ids
    # (the recomputed pair differs from the stored one by rounding only, but on sliver triangles that reaches 1e-4 in the normals)
    uv = raster_ref.barycentrics(clip, tri, rast)
    uv = rast[..., :2] + (uv - uv.detach())
    rast_d = torch.cat([uv.clamp(0, 1), rast[..., 2:]], -1)
    gb_pos = raster_ref.interpolate(v_pos, rast_d, tri)  # :182
    fn = mesh_ref.face_normals(v_pos, faces)  # :185-188
    fidx = torch.arange(faces.shape[0], dtype=torch.int32)[:, None].repeat(1, 3)
    gb_geo = raster_ref.interpolate(fn, rast_d, fidx)  # :190-191
    gb_nrm = raster_ref.interpolate(v_nrm, rast_d, tri)  # :195
    gb_flow = raster_ref.interpolate(flow_attr, rast_d, tri) if flow_attr is not None else None  # :199-200
    prior = v_pos if prior_v_pos is None else prior_v_pos
    gb_tex = raster_ref.interpolate(prior, rast_d, tri)  # :209
    if taps is not None:
        taps.update(clip=clip, rast=rast, gb=torch.cat([gb_pos, gb_geo, gb_nrm, gb_tex], -1), flow=gb_flow)
    buffers = shade(gb_pos, gb_geo, gb_nrm, gb_tex, w2c, view_pos, lgt, material, feat, render_modes, two_sided, gb_flow,
                    dino_net, class_vector)
    if background is not None:  # :299-304
        bg4 = torch.cat((background, torch.zeros_like(background[..., :1])), -1)
    else:
        bg4 = torch.zeros(1, H, W, 4)
    opp = raster_ref.edge_opposites(tri.numpy())
    outs = []
    for key in render_modes:  # :306-335
        if key not in buffers:
            outs.append(None)
            continue
        buf = buffers[key]
        aa = key in ("shaded", "flow", "dino_pred", "depth", "shading")
        bg = bg4 if key in ("shaded", "geo_normal", "shading") else torch.zeros_like(buf)
        if key == "shading" and background is not None:
            bg = bg[..., 2:]
        cover = (rast[..., -1:] > 0).float() * buf[..., -1:]
        acc = torch.lerp(bg.expand_as(buf), torch.cat((buf[..., :-1], torch.ones_like(buf[..., -1:])), -1), cover)  # :261-262
        if aa:
            acc = raster_ref.antialias(acc.contiguous(), rast, clip, tri, opp)  # :264-267
        if key in ("kd", "ks", "normal", "geo_normal"):
            acc = acc[..., :3]
        elif key in ("shading", "depth"):
            acc = acc[..., :1]
        elif key == "flow":
            acc = acc[..., :2]
        elif key == "dino_pred":
            acc = acc[..., :-1]
        outs.append(acc.permute(0, 3, 1, 2))
    return outs
