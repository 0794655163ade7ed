"""Geometry helpers mirroring /root/reference/model/geometry/util.py (public API kept; torch, any device).

``line_segment_distance`` is what the skinning kernel fuses (csrc/skin.hip); the torch form stays for callers
that import it directly.
"""
import torch


def line_segment_distance(a, b, points, sqrt=True):
    """Distance from ``points`` [..., N, D] to the segments a-b [..., D] (reference geometry/util.py:30-53)."""
    a, b = a[..., None, :], b[..., None, :]
    ab = b - a
    denom = torch.clamp((ab * ab).sum(-1, keepdim=True), min=1e-6)
    t = (((points - a) * ab).sum(-1, keepdim=True) / denom).clamp(0.0, 1.0)
    closest = a + t * ab
    dist = ((closest - points) ** 2).sum(-1)
    return torch.sqrt(dist + 1e-6) if sqrt else dist


def sample_farthest_points(pts, k, return_index=False):
    """Greedy farthest-point sampling, pts [B,3,N] -> [B,3,k] (reference geometry/util.py:5-27; unused by default)."""
    b, c, n = pts.shape
    idx = torch.randint(n, [b], device=pts.device)
    chosen = [idx]
    picked = pts.gather(2, idx[:, None, None].expand(b, c, 1))
    dist = (picked - pts).norm(dim=1)
    for _ in range(1, k):
        idx = dist.argmax(dim=1)
        chosen.append(idx)
        picked = pts.gather(2, idx[:, None, None].expand(b, c, 1))
        dist = torch.minimum(dist, (picked - pts).norm(dim=1))
    indexes = torch.stack(chosen, 1)
    out = pts.gather(2, indexes[:, None, :].expand(b, c, k))
    return (out, indexes) if return_index else out
