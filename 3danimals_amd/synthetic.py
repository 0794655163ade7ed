"""Synthetic inputs shaped like the reference's (no dataset / checkpoint can be fetched).

Used by tests, ``bench.py`` and ``__graft_entry__.smoke()``; sizes follow SURVEY.md section 8d.
All generators are deterministic given the seed and run on CPU (results are moved by the caller).
"""
from __future__ import annotations

import math

import numpy as np
import torch


def ellipsoid_sdf(pts: torch.Tensor, grid_scale: float = 7.0, noise: float = 0.01, seed: int = 0) -> torch.Tensor:
    """The reference's ``init_sdf='ellipsoid'`` (dmtet.py:246-250) plus N(0, noise^2): BASELINE config 1."""
    g = torch.Generator().manual_seed(seed)
    xs, ys, zs = pts.unbind(-1)
    sdf = grid_scale * 0.15 - torch.stack([xs, ys, zs / 2], -1).norm(dim=-1)
    return sdf + noise * torch.randn(sdf.shape, generator=g)


def _capsule(p, a, b, r):
    a = torch.tensor(a, dtype=p.dtype)
    b = torch.tensor(b, dtype=p.dtype)
    ab = b - a
    t = ((p - a) @ ab / (ab @ ab)).clamp(0, 1)
    return r - (p - (a + t[..., None] * ab)).norm(dim=-1)


def quadruped_sdf(pts: torch.Tensor, leg_radius: float = 0.2, noise: float = 0.0, seed: int = 0) -> torch.Tensor:
    """A horse-like union (positive inside): body ellipsoid + neck + head + four legs.

    y is up, the spine runs along z, x=0 is the symmetry plane -- the conventions
    ``estimate_bones`` assumes (skinning.py:142-153).  All four leg quadrants are populated.
    """
    p = pts
    rad = torch.tensor([0.5, 0.55, 1.25], dtype=p.dtype)
    ctr = torch.tensor([0.0, 0.45, 0.0], dtype=p.dtype)
    body = (1.0 - ((p - ctr) / rad).norm(dim=-1)) * 0.5
    parts = [body, _capsule(p, (0, 0.7, 1.0), (0, 1.3, 1.55), 0.27), _capsule(p, (0, 1.3, 1.55), (0, 1.25, 1.95), 0.22)]
    for sx in (-1, 1):
        for sz in (-1, 1):
            parts.append(_capsule(p, (0.3 * sx, 0.3, 0.85 * sz), (0.33 * sx, -1.15, 0.9 * sz), leg_radius))
    sdf = torch.stack(parts, 0).amax(0)
    if noise > 0:
        g = torch.Generator().manual_seed(seed)
        sdf = sdf + noise * torch.randn(sdf.shape, generator=g)
    return sdf


def perspective(fovy_deg: float = 25.0, aspect: float = 1.0, n: float = 0.1, f: float = 1000.0) -> torch.Tensor:
    """gluPerspective with the y flip of the reference (render/util.py:189-194)."""
    y = math.tan(math.radians(fovy_deg) / 2)
    return torch.tensor(
        [[1 / (y * aspect), 0, 0, 0], [0, 1 / -y, 0, 0], [0, 0, -(f + n) / (f - n), -(2 * f * n) / (f - n)], [0, 0, -1, 0]],
        dtype=torch.float32,
    )


def random_cameras(batch: int, seed: int = 0, cam_z: float = 10.0, fov: float = 25.0, max_xy: float = 0.2 * 2.22,
                   max_z: float = 0.5 * 2.22):
    """(mvp, w2c, campos) for ``batch`` look-at views around the object.

    Mirrors what ``get_camera_extrinsics_from_pose`` produces (InstancePredictorBase.py:606-621):
    w2c = [R | t] with the object pushed ``cam_z`` in front of the camera, mvp = proj @ w2c,
    campos = -R^T t.  Rotations: yaw uniform in [0,2pi), pitch in +-20 degrees.
    """
    g = torch.Generator().manual_seed(seed)
    yaw = torch.rand(batch, generator=g) * 2 * math.pi
    pitch = (torch.rand(batch, generator=g) - 0.5) * math.radians(40)
    cy, sy, cp, sp = yaw.cos(), yaw.sin(), pitch.cos(), pitch.sin()
    zero, one = torch.zeros(batch), torch.ones(batch)
    ry = torch.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy], -1).view(batch, 3, 3)
    rx = torch.stack([one, zero, zero, zero, cp, -sp, zero, sp, cp], -1).view(batch, 3, 3)
    R = rx @ ry
    t = torch.stack([(torch.rand(batch, generator=g) * 2 - 1) * max_xy, (torch.rand(batch, generator=g) * 2 - 1) * max_xy,
                     (torch.rand(batch, generator=g) * 2 - 1) * max_z], -1)
    t = t + torch.tensor([0.0, 0.0, -cam_z])
    w2c = torch.eye(4).repeat(batch, 1, 1)
    w2c[:, :3, :3] = R
    w2c[:, :3, 3] = t
    mvp = perspective(fov) @ w2c
    campos = -(R.transpose(1, 2) @ t[..., None])[..., 0]
    return mvp, w2c, campos


def seeded(shape, seed: int, low: float = 0.0, high: float = 1.0) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) * (high - low) + low


def np_rng(seed: int) -> np.random.Generator:
    return np.random.default_rng(seed)
