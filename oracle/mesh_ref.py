"""Oracle: vertex normals / tangents, restating /root/reference/model/render/mesh.py.

TEST INFRASTRUCTURE ONLY.  Pinned against tests/golden/mesh_*.npz.
"""
from __future__ import annotations

import torch


def _dot(a, b):
    return (a * b).sum(-1, keepdim=True)


def safe_normalize(x, eps: float = 1e-20):
    """reference render/util.py:28-32: x / sqrt(clamp(x.x, eps))."""
    return x / torch.sqrt(torch.clamp(_dot(x, x), min=eps))


def vertex_normals(v_pos: torch.Tensor, faces: torch.Tensor) -> torch.Tensor:
    """auto_normals, reference mesh.py:276-304.

    v_pos [B,V,3]; faces [F,3] int64.  Area-weighted (un-normalised) face
    normals are summed onto their three corners, near-zero sums are replaced by
    (0,0,1) and the result is safe-normalised.
    """
    i0, i1, i2 = faces[:, 0], faces[:, 1], faces[:, 2]
    v0, v1, v2 = v_pos[:, i0], v_pos[:, i1], v_pos[:, i2]
    fn = torch.cross(v1 - v0, v2 - v0, dim=-1)  # :285
    acc = torch.zeros_like(v_pos)
    for idx in (i0, i1, i2):  # :289-293
        acc = acc.index_add(1, idx, fn)
    default = torch.tensor([0.0, 0.0, 1.0], dtype=v_pos.dtype)
    acc = torch.where(_dot(acc, acc) > 1e-20, acc, default)  # :296-298
    return safe_normalize(acc)  # :299


def vertex_tangents(v_pos, faces, v_tex, uv_idx, v_nrm) -> torch.Tensor:
    """compute_tangents, reference mesh.py:310-350 (normal indices == position indices)."""
    pos = [v_pos[:, faces[:, i]] for i in range(3)]
    tex = [v_tex[:, uv_idx[:, i]] for i in range(3)]
    uve1, uve2 = tex[1] - tex[0], tex[2] - tex[0]
    pe1, pe2 = pos[1] - pos[0], pos[2] - pos[0]
    nom = pe1 * uve2[..., 1:2] - pe2 * uve1[..., 1:2]  # :333
    denom = uve1[..., 0:1] * uve2[..., 1:2] - uve1[..., 1:2] * uve2[..., 0:1]  # :334
    tang = nom / torch.where(denom > 0.0, torch.clamp(denom, min=1e-6), torch.clamp(denom, max=-1e-6))  # :337
    tsum = torch.zeros_like(v_nrm)
    cnt = torch.zeros_like(v_nrm)
    for i in range(3):  # :340-343
        tsum = tsum.index_add(1, faces[:, i], tang)
        cnt = cnt.index_add(1, faces[:, i], torch.ones_like(tang))
    t = tsum / cnt  # :344
    t = safe_normalize(t)  # :347
    return safe_normalize(t - _dot(t, v_nrm) * v_nrm)  # :348


def face_normals(v_pos: torch.Tensor, faces: torch.Tensor) -> torch.Tensor:
    """Normalised geometric normals used for the bent-normal trick, reference render.py:185-188."""
    v0, v1, v2 = v_pos[:, faces[:, 0]], v_pos[:, faces[:, 1]], v_pos[:, faces[:, 2]]
    return safe_normalize(torch.cross(v1 - v0, v2 - v0, dim=-1))
