"""Oracle: linear-blend skinning, restating /root/reference/model/geometry/skinning.py:369-439.

TEST INFRASTRUCTURE ONLY.  Pinned against tests/golden/skinning_*.npz.
Bone by bone, chain link by chain link -- the slow, obviously-right way.
"""
from __future__ import annotations

import torch


def segment_distance(a, b, pts):
    """reference geometry/util.py:30-53: sqrt(|closest(a,b,p)-p|^2 + 1e-6)."""
    a, b = a[..., None, :], b[..., None, :]
    ab = b - a
    t = ((pts - a) * ab).sum(-1, keepdim=True) / torch.clamp((ab * ab).sum(-1, keepdim=True), min=1e-6)
    t = t.clamp(0.0, 1.0)
    s = a + t * ab
    return torch.sqrt(((s - pts) ** 2).sum(-1) + 1e-6)


def bone_weights(bones, verts, temperature):
    """reference skinning.py:16-22: softmax over the bone axis of -dist/T.  -> [K,B,F,V]."""
    d = torch.stack([segment_distance(bones[:, :, k, 0], bones[:, :, k, 1], verts) for k in range(bones.shape[2])])
    return torch.softmax(-d / temperature, dim=0)


def rest_rotation(forward):
    """reference skinning.py:251-270 (right = (1,0,0); columns right, up, forward)."""
    f = torch.nn.functional.normalize(forward, dim=-1)
    right = torch.tensor([1.0, 0.0, 0.0], dtype=f.dtype).expand_as(f)
    up = torch.nn.functional.normalize(torch.cross(f, right, dim=-1), dim=-1)
    right = torch.cross(up, f, dim=-1)
    up = torch.nn.functional.normalize(up, dim=-1)
    return torch.stack([right, up, f], dim=-1)


def euler_xyz(angles):
    """reference skinning.py:285-340, convention 'XYZ': R = Rx @ Ry @ Rz."""
    x, y, z = angles.unbind(-1)
    o, zr = torch.ones_like(x), torch.zeros_like(x)
    rx = torch.stack([o, zr, zr, zr, x.cos(), -x.sin(), zr, x.sin(), x.cos()], -1).reshape(*x.shape, 3, 3)
    ry = torch.stack([y.cos(), zr, y.sin(), zr, o, zr, -y.sin(), zr, y.cos()], -1).reshape(*x.shape, 3, 3)
    rz = torch.stack([z.cos(), -z.sin(), zr, z.sin(), z.cos(), zr, zr, zr, o], -1).reshape(*x.shape, 3, 3)
    return rx @ ry @ rz


def _affine(rot=None, trans=None, n=1):
    m = torch.eye(4).repeat(n, 1, 1)
    if rot is not None:
        m[:, :3, :3] = rot
    if trans is not None:
        m[:, :3, 3] = trans
    return m


def _rigid_inverse(m):
    r, t = m[:, :3, :3], m[:, :3, 3]
    inv = torch.eye(4).repeat(len(m), 1, 1)
    inv[:, :3, :3] = r.transpose(1, 2)
    inv[:, :3, 3] = -(r.transpose(1, 2) @ t[..., None])[..., 0]
    return inv


def bone_transforms(bones, kinematic_tree, angles):
    """Per-bone world transforms M[k] of shape [B*F,4,4] (reference skinning.py:389-417).

    For bone k the chain is [k, parents...] leaf -> root, and each link i applies
    Rest_i . Rot_i . Rest_i^-1 on the left.
    """
    B, F = angles.shape[:2]
    out = {}
    for bone_id, _ in kinematic_tree:
        parents = [p for p, ch in kinematic_tree if bone_id in ch]
        chain = (parents + [bone_id])[::-1]
        m = torch.eye(4)[None]
        for i in chain:
            joint = bones[:, :, i, 0].reshape(-1, 3)
            vec = (bones[:, :, i, 1] - bones[:, :, i, 0]).reshape(-1, 3)
            rest = _affine(rest_rotation(vec), joint, n=len(joint))
            rot = _affine(euler_xyz(angles[:, :, i].reshape(-1, 3)), None, n=B * F)
            m = rest @ (rot @ (_rigid_inverse(rest) @ m))
        out[bone_id] = m
    return out


def skinning(v_pos, bones, kinematic_tree, angles, temperature=1.0, output_posed_bones=False):
    """reference skinning.py:369-439.  v_pos [1|B,1|F,V,3], bones [1|B,1|F,K,2,3], angles [B,F,K,3]."""
    B, F = angles.shape[:2]
    w = bone_weights(bones, v_pos.detach(), temperature)  # [K,b,f,V]
    mats = bone_transforms(bones, kinematic_tree, angles)
    v4 = torch.cat([v_pos, torch.ones_like(v_pos[..., :1])], -1).reshape(-1, v_pos.shape[-2], 4)
    posed = bones.clone()
    if output_posed_bones and (posed.shape[0] != B or posed.shape[1] != F):
        posed = posed.repeat(B, F, 1, 1, 1)
    acc = 0
    for bone_id, _ in kinematic_tree:
        m = mats[bone_id]
        moved = (v4 @ m.transpose(-2, -1))[..., :3].reshape(B, F, -1, 3)
        if output_posed_bones:
            b4 = torch.cat([posed[:, :, bone_id].reshape(B * F, 2, 3), torch.ones(B * F, 2, 1)], -1)
            posed[:, :, bone_id] = (b4 @ m.transpose(-2, -1))[..., :3].reshape(B, F, 2, 3)
        acc = acc + w[bone_id][..., None] * moved
    aux = {"bones_pred": bones, "vertices_to_bones": w}
    if output_posed_bones:
        aux["posed_bones"] = posed
    return acc, aux
