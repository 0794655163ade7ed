"""Oracle: rasterise / interpolate / antialias (the nvdiffrast operator API as the reference uses it).

TEST INFRASTRUCTURE ONLY.  **PARITY UNPINNED** -- see raster_ref.c and oracle/__init__.py: the
reference calls nvdiffrast (render.py:24,264-267,292-294), which is absent; this restates the
operator semantics of SURVEY.md Appendix A.  Coverage / triangle ids come from the scalar C
rasteriser (raster_ref.c); everything differentiable is torch-CPU so tests can autograd through it.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_LIB = None


def build(force: bool = False) -> str:
    """gcc the C restatement into oracle/_build/libraster_ref.so (strict IEEE: no contraction)."""
    os.makedirs(_BUILD, exist_ok=True)
    src = os.path.join(_HERE, "raster_ref.c")
    out = os.path.join(_BUILD, "libraster_ref.so")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", src, "-o", out, "-lm"])
    return out


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.a3d_ref_rasterize.restype = ctypes.c_int
        _LIB.a3d_ref_rasterize.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
        _LIB.a3d_ref_rasterize_peel.restype = ctypes.c_int
        _LIB.a3d_ref_rasterize_peel.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p]
    return _LIB


def rasterize(pos: torch.Tensor, tri: torch.Tensor, resolution, prev: torch.Tensor = None) -> torch.Tensor:
    """dr.rasterize / DepthPeeler layer 0 -> rast [B,H,W,4] = (u, v, z/w, tri_id+1); with ``prev`` (the previous layer) the next
    depth layer: per pixel the nearest fragment strictly behind the previous one in (z/w, id) order."""
    H, W = int(resolution[0]), int(resolution[1])
    p = pos.detach().to(torch.float32).contiguous()
    t = tri.to(torch.int32).contiguous()
    B, V = p.shape[0], p.shape[1]
    out = torch.empty(B, H, W, 4, dtype=torch.float32)
    if prev is None:
        rc = _lib().a3d_ref_rasterize(p.data_ptr(), B, t.data_ptr(), B, V, t.shape[0], H, W, out.data_ptr())
    else:
        pv = prev.detach().to(torch.float32).contiguous()
        assert tuple(pv.shape) == (B, H, W, 4)
        rc = _lib().a3d_ref_rasterize_peel(p.data_ptr(), B, t.data_ptr(), B, V, t.shape[0], H, W, pv.data_ptr(), out.data_ptr())
    assert rc == 0
    return out


def barycentrics(pos: torch.Tensor, tri: torch.Tensor, rast: torch.Tensor) -> torch.Tensor:
    """Differentiable (u,v) of the stored triangle ids (unclamped); used to check rasterise backward.

    u = a0/(a0+a1+a2), v = a1/(...), a_i from q_i = (x_i - fx w_i, y_i - fy w_i)  (Appendix A).
    """
    B, H, W, _ = rast.shape
    ids = rast[..., 3].long() - 1
    hit = ids >= 0
    t = tri.long()[ids.clamp(min=0)]  # [B,H,W,3]
    bidx = torch.arange(B)[:, None, None, None].expand_as(t)
    P = pos[bidx, t]  # [B,H,W,3,4]
    fx = ((torch.arange(W, dtype=torch.float32) + 0.5) * (2.0 / W) - 1.0)[None, None, :]
    fy = ((torch.arange(H, dtype=torch.float32) + 0.5) * (2.0 / H) - 1.0)[None, :, None]
    qx = P[..., 0] - fx[..., None] * P[..., 3]
    qy = P[..., 1] - fy[..., None] * P[..., 3]
    a0 = qx[..., 1] * qy[..., 2] - qy[..., 1] * qx[..., 2]
    a1 = qx[..., 2] * qy[..., 0] - qy[..., 2] * qx[..., 0]
    a2 = qx[..., 0] * qy[..., 1] - qy[..., 0] * qx[..., 1]
    s = a0 + a1 + a2
    s = torch.where(hit, s, torch.ones_like(s))
    uv = torch.stack([a0 / s, a1 / s], -1)
    return torch.where(hit[..., None], uv, torch.zeros_like(uv))


def interpolate(attr: torch.Tensor, rast: torch.Tensor, tri: torch.Tensor) -> torch.Tensor:
    """dr.interpolate(attr, rast, tri)[0]: u*A0 + v*A1 + (1-u-v)*A2, zero where empty.

    attr [B,V,C] or [1,V,C] (broadcast over the batch, render.py:209).  Differentiable wrt attr and
    rast[...,:2].
    """
    if attr.dim() == 2:
        attr = attr[None]
    B = rast.shape[0]
    ids = rast[..., 3].long() - 1
    hit = (ids >= 0)[..., None]
    t = tri.long()[ids.clamp(min=0)]
    if attr.shape[0] == 1:
        A = attr[0][t]  # [B,H,W,3,C]
    else:
        bidx = torch.arange(B)[:, None, None, None].expand_as(t)
        A = attr[bidx, t]
    u, v = rast[..., 0:1], rast[..., 1:2]
    out = u * A[..., 0, :] + v * A[..., 1, :] + (1.0 - u - v) * A[..., 2, :]
    return torch.where(hit, out, torch.zeros_like(out))


def edge_opposites(tri: np.ndarray) -> np.ndarray:
    """opp[f,i] = vertex opposite edge i of face f in the adjacent face, or -1 (mesh boundary).

    Edge i of a face (v0,v1,v2) is the one NOT containing v_i: (v_{i+1}, v_{i+2}).  Each undirected
    edge keeps, per traversal direction, the lowest and the highest (face*4+corner) code that uses it;
    a face's neighbour across an edge is the lowest code of the opposite direction, else -- a mesh that
    is not consistently wound -- a different face of the same direction (the lowest code, or the
    highest when the face itself is the lowest), else none.  nvdiffrast keeps the first two opposite
    vertices that arrive for an undirected edge whatever the winding (antialias.cu evhashInsert /
    evhashFind); on a 2-manifold both rules name the one other face.  (Specification shared with
    aa_topology in 3danimals_amd/csrc/topo_common.h.  Until round 3 only the lowest code per direction
    was kept: of two faces traversing a shared edge the same way only one found the other.)
    """
    tri = np.asarray(tri, dtype=np.int64)
    F = tri.shape[0]
    table = {}
    for f in range(F):
        for i in range(3):
            a, b = int(tri[f, (i + 1) % 3]), int(tri[f, (i + 2) % 3])
            key, d = ((a, b), 0) if a < b else ((b, a), 1)
            code = f * 4 + i
            slot = table.setdefault(key, [None, None, None, None])
            if slot[d] is None or code < slot[d]:
                slot[d] = code
            if slot[2 + d] is None or code > slot[2 + d]:
                slot[2 + d] = code
    opp = np.full((F, 3), -1, dtype=np.int32)
    for f in range(F):
        for i in range(3):
            a, b = int(tri[f, (i + 1) % 3]), int(tri[f, (i + 2) % 3])
            if a == b:
                continue
            key, d = ((a, b), 0) if a < b else ((b, a), 1)
            slot = table[key]
            code = f * 4 + i
            other = slot[1 - d]
            if other is None:
                other = slot[d] if slot[d] != code else (slot[2 + d] if slot[2 + d] != code else None)
            if other is not None:
                opp[f, i] = tri[other // 4, other % 4]
    return opp


def _same_sign(a, b):
    return torch.signbit(a) == torch.signbit(b)


def antialias(color: torch.Tensor, rast: torch.Tensor, pos: torch.Tensor, tri: torch.Tensor, opp=None) -> torch.Tensor:
    """dr.antialias(color, rast, pos, tri) -- analytic silhouette antialiasing (Appendix A).

    For every horizontally / vertically adjacent pixel pair with different triangle ids: take the
    nearer covered pixel's triangle; among its edges that straddle the line joining the two pixel
    centres pick the one whose crossing is farthest along the pair direction; if that edge is a
    silhouette edge (no neighbour, or the neighbour's opposite vertex lies on the same side as the
    triangle's own third vertex), is steeper than 45 degrees w.r.t. the pair direction, and crosses
    at distance dc in (-1/16, 1+1/16) from the triangle's pixel: alpha = +-(0.5 - clamp(dc,0,1)) and
    the pixel on the short side is blended toward the other.  Differentiable wrt color and pos.
    """
    B, H, W, C = color.shape
    tri_l = tri.long()
    if opp is None:
        opp = edge_opposites(tri.cpu().numpy())
    opp_l = torch.as_tensor(opp).long()
    ids = rast[..., 3].long() - 1
    zw = rast[..., 2]
    out = color.clone()
    flat_out = out.view(-1, C)
    flat_col = color.reshape(-1, C)
    xh, yh = W * 0.5, H * 0.5
    for d in (0, 1):
        if d == 0:
            id0, id1 = ids[:, :, :-1], ids[:, :, 1:]
            z0, z1 = zw[:, :, :-1], zw[:, :, 1:]
        else:
            id0, id1 = ids[:, :-1, :], ids[:, 1:, :]
            z0, z1 = zw[:, :-1, :], zw[:, 1:, :]
        sel = (id0 != id1).nonzero()
        if sel.shape[0] == 0:
            continue
        b, y, x = sel[:, 0], sel[:, 1], sel[:, 2]
        t0, t1 = id0[b, y, x], id1[b, y, x]
        zz0, zz1 = z0[b, y, x], z1[b, y, x]
        t = torch.where(t0 >= 0, t0, t1)
        both = (t0 >= 0) & (t1 >= 0)
        t = torch.where(both, torch.where(zz0 < zz1, t0, t1), t)
        use1 = t == t1
        px = x + torch.where(use1, 1 - d, 0)
        py = y + torch.where(use1, d, 0)
        ds = torch.where(use1, -1.0, 1.0)
        vi = tri_l[t]  # [n,3]
        oi = opp_l[t]
        oi = torch.where(oi >= 0, oi, vi)  # no neighbour -> own vertex -> always silhouette
        pb = b if pos.shape[0] > 1 else torch.zeros_like(b)
        P = pos[pb[:, None], vi]  # [n,3,4]
        O = pos[pb[:, None], oi]
        fx = (px.float() + 0.5 - xh)[:, None]
        fy = (py.float() + 0.5 - yh)[:, None]
        X = P[..., 0] / P[..., 3] * xh - fx
        Y = P[..., 1] / P[..., 3] * yh - fy
        OX = O[..., 0] / O[..., 3] * xh - fx
        OY = O[..., 1] / O[..., 3] * yh - fy
        x0, x1, x2 = X.unbind(-1)
        y0, y1, y2 = Y.unbind(-1)
        bb = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0)
        w0 = (x1 - OX[:, 0]) * (y2 - OY[:, 0]) - (x2 - OX[:, 0]) * (y1 - OY[:, 0])
        w1 = (x2 - OX[:, 1]) * (y0 - OY[:, 1]) - (x0 - OX[:, 1]) * (y2 - OY[:, 1])
        w2 = (x0 - OX[:, 2]) * (y1 - OY[:, 2]) - (x1 - OX[:, 2]) * (y0 - OY[:, 2])
        sil = torch.stack([_same_sign(w0, bb), _same_sign(w1, bb), _same_sign(w2, bb)], -1)
        if d == 1:  # pair direction becomes the first coordinate
            X, Y = Y, X
            x0, x1, x2 = X.unbind(-1)
            y0, y1, y2 = Y.unbind(-1)
        # edge i joins vertices (i+1, i+2)
        xa = torch.stack([x1, x2, x0], -1)
        ya = torch.stack([y1, y2, y0], -1)
        xb = torch.stack([x2, x0, x1], -1)
        yb = torch.stack([y2, y0, y1], -1)
        dxe, dye = xb - xa, yb - ya
        num = ds[:, None] * (xa * dye - ya * dxe)
        straddle = ~_same_sign(ya, yb)
        ratio = torch.where(straddle, num / torch.where(straddle, dye, torch.ones_like(dye)), torch.full_like(num, -float("inf")))
        # argmax, ties to the lowest index
        di = torch.zeros_like(t)
        best = ratio[:, 0]
        for k in (1, 2):
            better = ratio[:, k] > best
            di = torch.where(better, k, di)
            best = torch.where(better, ratio[:, k], best)
        pick = lambda v: v.gather(1, di[:, None])[:, 0]
        ok = pick(sil) & pick(straddle) & (pick(dye).abs() >= pick(dxe).abs())
        dc = best
        eps = 0.0625
        ok = ok & (dc > -eps) & (dc < 1.0 + eps)
        if not bool(ok.any()):
            continue
        dcc = dc.clamp(0.0, 1.0)
        alpha = ds * (0.5 - dcc)
        p0 = (b * H + y) * W + x
        p1 = p0 + (1 if d == 0 else W)
        alpha = torch.where(ok, alpha, torch.zeros_like(alpha))
        dst = torch.where(alpha > 0, p0, p1)
        contrib = alpha[:, None] * (flat_col[p1] - flat_col[p0])
        contrib = contrib[ok]
        flat_out = flat_out.index_add(0, dst[ok], contrib)
    return flat_out.view(B, H, W, C)
