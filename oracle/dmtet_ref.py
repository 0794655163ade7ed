"""Oracle: marching tetrahedra, restating /root/reference/model/geometry/dmtet.py.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned against goldens
captured from the imported reference (tests/golden/dmtet_*.npz).

The topology part is numpy (integer work, bit-exact bar); the vertex
interpolation is torch so tests can take d(verts)/d(sdf) by autograd.
This deliberately follows the reference's *own* formulation (sort + unique over
the edges of the valid tets) and NOT the static-edge prefix-sum formulation the
HIP kernels use, so that a parity test between the two means something.
"""
from __future__ import annotations

import numpy as np
import torch

# reference dmtet.py:26-43 (edge-slot triples per occupancy case, -1 padded)
TRIANGLE_TABLE = np.array(
    [
        [-1, -1, -1, -1, -1, -1],
        [1, 0, 2, -1, -1, -1],
        [4, 0, 3, -1, -1, -1],
        [1, 4, 2, 1, 3, 4],
        [3, 1, 5, -1, -1, -1],
        [2, 3, 0, 2, 5, 3],
        [1, 4, 0, 1, 5, 4],
        [4, 2, 5, -1, -1, -1],
        [4, 5, 2, -1, -1, -1],
        [4, 1, 0, 4, 5, 1],
        [3, 2, 0, 3, 5, 2],
        [1, 3, 5, -1, -1, -1],
        [4, 1, 2, 4, 3, 1],
        [3, 0, 4, -1, -1, -1],
        [2, 0, 1, -1, -1, -1],
        [-1, -1, -1, -1, -1, -1],
    ],
    dtype=np.int64,
)
# reference dmtet.py:45
NUM_TRIANGLES = np.array([0, 1, 1, 2, 1, 2, 2, 1, 1, 2, 2, 1, 2, 1, 1, 0], dtype=np.int64)
# reference dmtet.py:46
EDGE_SLOTS = np.array([0, 1, 0, 2, 0, 3, 1, 2, 1, 3, 2, 3], dtype=np.int64)


def uv_atlas(num_tets: int):
    """Per-tet uv quads, reference dmtet.py:69-84 (depends only on Nt)."""
    n = int(np.ceil(np.sqrt((2 * num_tets + 1) // 2)))
    lin = torch.linspace(0, 1 - (1 / n), n, dtype=torch.float32)
    ty, tx = torch.meshgrid(lin, lin, indexing="ij")
    pad = 0.9 / n
    uvs = torch.stack([tx, ty, tx + pad, ty, tx + pad, ty + pad, tx, ty + pad], dim=-1).view(-1, 2)
    return uvs, n


def topology(sdf: np.ndarray, tets: np.ndarray):
    """Integer part of DMTet.__call__ (reference dmtet.py:104-121,133-151,86-96).

    Returns dict(interp_v int64 [V,2], faces int64 [F,3], uv_idx int64 [F,3]).
    """
    sdf = np.asarray(sdf).reshape(-1)
    tets = np.asarray(tets, dtype=np.int64)
    occ = sdf > 0  # :106 strict
    occ4 = occ[tets]  # [Nt,4]
    nin = occ4.sum(-1)
    valid = (nin > 0) & (nin < 4)  # :107-109
    vt = tets[valid]
    # all 6 edges of the valid tets, each sorted (min,max)  (:112-113, sort_edges :59-67)
    e = vt[:, EDGE_SLOTS].reshape(-1, 2)
    e = np.stack([e.min(-1), e.max(-1)], -1)
    # lexicographic unique with inverse (:115)
    if e.shape[0]:
        uniq, inv = np.unique(e, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
    else:
        uniq, inv = np.zeros((0, 2), np.int64), np.zeros((0,), np.int64)
    crossing = occ[uniq].sum(-1) == 1  # :118
    rank = np.full(uniq.shape[0], -1, dtype=np.int64)
    rank[crossing] = np.arange(int(crossing.sum()))  # :119-120
    idx_map = rank[inv].reshape(-1, 6)  # :121,133
    interp_v = uniq[crossing]  # :123

    case = (occ4[valid].astype(np.int64) * (1 << np.arange(4))).sum(-1)  # :135-136
    ntri = NUM_TRIANGLES[case]  # :137
    one, two = ntri == 1, ntri == 2
    f1 = np.take_along_axis(idx_map[one], TRIANGLE_TABLE[case[one]][:, :3], axis=1).reshape(-1, 3)
    f2 = np.take_along_axis(idx_map[two], TRIANGLE_TABLE[case[two]][:, :6], axis=1).reshape(-1, 3)
    faces = np.concatenate([f1, f2], 0)  # :140-143 (1-triangle tets first)

    gid = np.arange(tets.shape[0], dtype=np.int64)[valid]  # :146-147
    g2 = np.stack([gid[two] * 2, gid[two] * 2 + 1], -1).reshape(-1)
    face_gidx = np.concatenate([gid[one] * 2, g2], 0)  # :148-151
    # map_uv index part (:86-96); _idx(t,N) == t
    tet_idx = face_gidx // 2
    tri_idx = face_gidx % 2
    uv_idx = np.stack([tet_idx * 4, tet_idx * 4 + tri_idx + 1, tet_idx * 4 + tri_idx + 2], -1).reshape(-1, 3)
    return dict(interp_v=interp_v, faces=faces, uv_idx=uv_idx)


def interpolate_verts(pos: torch.Tensor, sdf: torch.Tensor, interp_v: torch.Tensor) -> torch.Tensor:
    """Zero-crossing placement, reference dmtet.py:124-131 (same op order)."""
    p = pos[interp_v.reshape(-1)].reshape(-1, 2, 3)
    s = sdf.reshape(-1)[interp_v.reshape(-1)].reshape(-1, 2, 1)
    s = torch.cat([s[:, :1], -s[:, 1:]], 1)  # second endpoint negated (:126)
    den = s.sum(1, keepdim=True)  # s_a - s_b (:128)
    w = torch.flip(s, [1]) / den  # (-s_b/den, s_a/den) (:130)
    return (p * w).sum(1)  # :131


def marching_tets(pos: torch.Tensor, sdf: torch.Tensor, tets: torch.Tensor):
    """Full DMTet.__call__ -> (verts, faces, uvs, uv_idx)  (reference dmtet.py:104-155)."""
    topo = topology(sdf.detach().cpu().numpy(), tets.cpu().numpy())
    interp_v = torch.from_numpy(topo["interp_v"])
    verts = interpolate_verts(pos, sdf, interp_v)
    uvs, _ = uv_atlas(tets.shape[0])
    return verts, torch.from_numpy(topo["faces"]), uvs, torch.from_numpy(topo["uv_idx"])


def ellipsoid_init_sdf(pts: torch.Tensor, grid_scale: float) -> torch.Tensor:
    """init_sdf == 'ellipsoid', reference dmtet.py:246-250."""
    rxy = grid_scale * 0.15
    xs, ys, zs = pts.unbind(-1)
    return rxy - torch.stack([xs, ys, zs / 2], -1).norm(dim=-1, keepdim=True)


def sphere_init_sdf(pts: torch.Tensor, grid_scale: float) -> torch.Tensor:
    """init_sdf == 'sphere', reference dmtet.py:241-244."""
    return grid_scale * 0.25 - pts.norm(dim=-1, keepdim=True)
