"""Bone estimation and linear-blend skinning -- public API of /root/reference/model/geometry/skinning.py.

``skinning()`` (reference :369-439) spends its time in ~24k tiny torch ops: for each of the K bones it walks the
kinematic chain composing 4x4 matrices one link at a time, then transforms a full copy of the vertex array.  Here

* the K per-bone link transforms  L_i = Rest_i . Rot(euler_i) . Rest_i^-1  are built for all bones at once and
  the chains are composed level by level with batched 4x4 matmuls (a handful of torch ops, autograd for free:
  gradients reach ``deform_params`` through them);
* the per-vertex work -- segment distances, softmax over bones, weighted sum of K affine maps, and its backward
  incl. the reduction of the 3x4 transform gradients over the vertices -- is one HIP kernel each way
  (csrc/skin.hip).

``estimate_bones()`` (reference :50-248, @no_grad, once per epoch) is restated on torch and works on any device.
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn.functional as F

from ... import ops
from ..._lib import fp32_region
from . import util  # noqa: F401  (public attribute of the reference module)


# ------------------------------------------------------------------------------------------------ kinematic chains
def build_kinematic_chain(n_bones, start_bone_idx):
    """Leaf-to-root chain of ``n_bones`` bones starting at ``start_bone_idx`` (reference :25-37).

    Returns (bones_to_joints, kinematic_chain, dependent_bones); parents come first in ``kinematic_chain``.
    """
    bones_to_joints, chain, below = [], [], []
    for i in range(n_bones):
        bone = start_bone_idx + i
        bones_to_joints.append((i + 1, i))
        chain.insert(0, (bone, below))
        below = below + [bone]
    return bones_to_joints, chain, below


QUADRANT_POPULATION = None  # a list: estimate_bones appends the four leg quadrants' vertex counts (device tensor) on every call


def update_body_kinematic_chain(kinematic_chain, leg_kinematic_chain, body_bone_idx, leg_bone_idxs, attach_legs_to_body=True):
    """Hang a leg under the body bone it attaches to and every ancestor of it (reference :40-46; mutates the lists)."""
    if attach_legs_to_body:
        for bone_idx, dependents in kinematic_chain:
            if bone_idx == body_bone_idx or body_bone_idx in dependents:
                dependents += leg_bone_idxs
    return kinematic_chain + leg_kinematic_chain


def children_to_parents(kinematic_tree):
    """[(bone, [children])] -> [(bone, [parents])] (reference :273-282)."""
    return [(b, [p for p, ch in kinematic_tree if b in ch]) for b, _ in kinematic_tree]


def _joints_to_bones(joints, pairs):
    return torch.stack([torch.stack([joints[:, :, a], joints[:, :, b]], dim=2) for a, b in pairs], dim=2)


def _masked_quantiles(x, mask, qs):
    """torch.quantile(x[mask], q) (linear interpolation) for each q, without boolean indexing: no host sync, static shapes."""
    flat = torch.where(mask, x, torch.full_like(x, float("inf"))).reshape(-1).sort().values
    n = mask.sum()
    out = []
    for q in qs:
        pos = q * (n - 1).to(x.dtype)
        lo = pos.floor()
        hi = pos.ceil()
        idx = torch.stack((lo, hi)).long().clamp(min=0)  # (a 0-dim tensor index would be read back to the host)
        ab = flat.index_select(0, idx)
        out.append(torch.lerp(ab[0], ab[1], pos - lo))
    return out


DEVICE_ESTIMATE_BONES = os.environ.get("A3D_DEVICE_ESTIMATE_BONES", "1") != "0"  # estimate_bones as one HIP launch (a3d_estimate_bones); off: the torch restatement below


def _body_chain(n_body_bones):
    """(bones_to_joints, kinematic_chain) of the spine (reference :128-141)."""
    half = n_body_bones // 2
    bones_to_joints, kinematic_chain, bone = [], [], 0
    below = []
    for i in range(half):  # point_a -> mid
        bones_to_joints.append((i + 1, i))
        kinematic_chain.insert(0, (bone, below))
        below = below + [bone]
        bone += 1
    below = []
    for i in range(n_body_bones - 1, half - 1, -1):  # point_b -> mid
        bones_to_joints.append((i, i + 1))
        kinematic_chain.insert(0, (bone, below))
        below = below + [bone]
        bone += 1
    return bones_to_joints, kinematic_chain


def _estimate_bones_device(seq_shape, n_body_bones, n_leg_bones, body_bones_mode, compute_kinematic_chain, aux, attach_legs_to_body,
                           legs_to_body_joint_indices, bone_y_threshold):
    """estimate_bones through a3d_estimate_bones (csrc/bones.hip): ONE launch for the ~245 torch launches of the restatement below; the
    kinematic chain -- a Python structure -- is built here from the two attachment joints the launch hands back (the one read-back of a
    chain rebuild, as before; none with a cached chain).  None: a cached ``aux`` this form does not cover (the caller takes the torch path)."""
    from ..._lib import defer_check, read_back

    b2j_body, chain_body = _body_chain(n_body_bones)
    leg_b2j = [(i + 1, i) for i in range(n_leg_bones)]
    if compute_kinematic_chain:
        given = list(legs_to_body_joint_indices) if legs_to_body_joint_indices is not None else [None] * 4
        attach = [(-1 if given[0] is None else int(given[0])), (-1 if given[1] is None else int(given[1])), -1, -1]  # legs 2 / 3 reuse 1 / 0 (:213-216)
    else:
        if list(aux["bones_to_joints"]) != b2j_body or (n_leg_bones > 0 and any(list(l["leg_bones_to_joints"]) != leg_b2j for l in aux["legs"])):
            return None
        attach = [int(l["body_bone_idx"]) for l in aux["legs"]] if n_leg_bones > 0 else [0, 0, 0, 0]
    bones, nearest, ok = ops.estimate_bones_device(seq_shape, n_body_bones, n_leg_bones, body_bones_mode == "z_minmax_y+", bone_y_threshold, attach)
    if n_leg_bones > 0:
        defer_check(ok[0], "estimate_bones: no vertex in a leg quadrant (the reference drops into pdb here, skinning.py:183)")
    if not compute_kinematic_chain:
        return bones
    aux = {"bones_to_joints": b2j_body}
    kinematic_chain = chain_body
    if n_leg_bones > 0:
        idx = [attach[0], attach[1]]
        if attach[0] < 0 or attach[1] < 0:  # THE read-back of a chain rebuild (it also carries the quadrant flag deferred above)
            got = read_back(nearest).tolist()
            idx = [int(got[0]), int(got[1])]
        if legs_to_body_joint_indices is None:
            legs_to_body_joint_indices = [None, None, None, None]
        leg_auxs, start = [], n_body_bones
        for i in range(4):
            body_bone_idx = idx[1] if i == 2 else (idx[0] if i == 3 else idx[i])
            legs_to_body_joint_indices[i] = body_bone_idx  # written back into the caller's list, as the reference does (:220)
            b2j, leg_chain, leg_ids = build_kinematic_chain(n_leg_bones, start_bone_idx=start)
            kinematic_chain = update_body_kinematic_chain(kinematic_chain, leg_chain, body_bone_idx, leg_ids, attach_legs_to_body)
            leg_auxs.append({"body_bone_idx": body_bone_idx, "leg_bones_to_joints": b2j})
            start += n_leg_bones
        aux["legs"] = leg_auxs
    aux["kinematic_chain"] = kinematic_chain
    return bones, kinematic_chain, aux


# ------------------------------------------------------------------------------------------------ bone estimation
@torch.no_grad()
@fp32_region
def estimate_bones(seq_shape, n_body_bones, resample=False, n_legs=4, n_leg_bones=0, body_bones_mode="z_minmax", compute_kinematic_chain=True,
                   aux=None, attach_legs_to_body=True, legs_to_body_joint_indices=None, bone_y_threshold=None):
    """Heuristic skeleton from the rest-pose vertices (reference :50-248).

    seq_shape [B,F,V,3] -> bones [B,F,K,2,3] (+ kinematic_chain, aux when ``compute_kinematic_chain``).
    Spine: the two extreme-z vertices (optionally only among those not far below the centroid), snapped to the
    x=0 symmetry plane, joined through the (lifted) centroid by ``n_body_bones`` bones.  Legs: lowest vertex of
    each top-view quadrant, joined to the nearest-in-z (or prescribed) body joint by ``n_leg_bones`` bones.
    """
    if resample:
        b, _, n, _ = seq_shape.shape
        pts = util.sample_farthest_points(seq_shape.reshape(-1, n, 3).transpose(1, 2), n // 4)
        seq_shape = pts.transpose(1, 2).reshape(b, -1, n // 4, 3)

    if (DEVICE_ESTIMATE_BONES and QUADRANT_POPULATION is None and body_bones_mode in ("z_minmax", "z_minmax_y+")
            and ops.estimate_bones_device_ok(seq_shape, n_body_bones, n_leg_bones, n_legs)):
        out = _estimate_bones_device(seq_shape, n_body_bones, n_leg_bones, body_bones_mode, compute_kinematic_chain, aux, attach_legs_to_body,
                                     legs_to_body_joint_indices, bone_y_threshold)
        if out is not None:
            return out

    def pick(idx):
        return seq_shape.gather(2, idx[..., None, None].expand(-1, -1, 1, 3)).squeeze(2)

    if body_bones_mode == "z_minmax":
        point_a, point_b = pick(seq_shape[..., 2].argmax(dim=2)), pick(seq_shape[..., 2].argmin(dim=2))
    elif body_bones_mode == "z_minmax_y+":
        centroid = seq_shape.mean(2)
        upper = (seq_shape[..., 1] > (centroid[:, :, None, 1] - 0.5)).float()
        z = seq_shape[..., 2]
        point_a = pick((z * upper + (-1e6) * (1 - upper)).argmax(2))
        point_b = pick((z * upper + 1e6 * (1 - upper)).argmin(2))
    else:
        raise NotImplementedError
    point_a[..., 0] = 0
    point_b[..., 0] = 0
    mid_point = seq_shape.mean(2)
    mid_point[..., 0] = 0
    if n_leg_bones > 0:
        mid_point[..., 1] += 0.5

    assert n_body_bones % 2 == 0
    n_joints = n_body_bones + 1
    blend = torch.linspace(0.0, 1.0, math.ceil(n_joints / 2), device=point_a.device)[None, None, :, None]
    joints_a = point_a[:, :, None] * (1 - blend) + mid_point[:, :, None] * blend
    joints_b = point_b[:, :, None] * blend + mid_point[:, :, None] * (1 - blend)
    joints = torch.cat([joints_a[:, :, :-1], joints_b], 2)

    if compute_kinematic_chain:
        aux = {}
        half = n_body_bones // 2
        bones_to_joints, kinematic_chain, bone = [], [], 0
        below = []
        for i in range(half):  # point_a -> mid
            bones_to_joints.append((i + 1, i))
            kinematic_chain.insert(0, (bone, below))
            below = below + [bone]
            bone += 1
        below = []
        for i in range(n_body_bones - 1, half - 1, -1):  # point_b -> mid
            bones_to_joints.append((i, i + 1))
            kinematic_chain.insert(0, (bone, below))
            below = below + [bone]
            bone += 1
        aux["bones_to_joints"] = bones_to_joints
    else:
        bones_to_joints, kinematic_chain = aux["bones_to_joints"], aux["kinematic_chain"]
    bones_pred = _joints_to_bones(joints, bones_to_joints)

    if n_leg_bones > 0:
        assert n_legs == 4
        xs, ys, zs = seq_shape.unbind(-1)
        if bone_y_threshold is None:
            margin = (xs.quantile(0.95) - xs.quantile(0.05)) * 0.2
            quadrants = [(xs > margin) & (zs > 0), (xs > margin) & (zs < 0), (xs < -margin) & (zs < 0), (xs < -margin) & (zs > 0)]
        else:  # Fauna variant: centre the quadrants on the lower part of the body
            low = ys < ys.quantile(bone_y_threshold)
            xq, zq = _masked_quantiles(xs, low, (0.5, 0.95, 0.05)), _masked_quantiles(zs, low, (0.5, 0.95, 0.05))
            x0, z0 = xq[0], zq[0]  # == xs[low].quantile(.) of the reference (:160-166), without the data-dependent shape
            mx, mz = (xq[1] - xq[2]) * 0.2, (zq[1] - zq[2]) * 0.2
            quadrants = [(xs - x0 > mx) & (zs - z0 > mz), (xs - x0 > mx) & (zs < z0), (xs - x0 < -mx) & (zs < z0),
                         (xs - x0 < -mx) & (zs - z0 > mz)]

        def foot_of(quadrant):
            """Lowest vertex of the quadrant (first one on ties, like indexing the masked subset; skinning.py:177-186), all (b, f) at
            once on the tensor's device.  An EMPTY quadrant (the reference drops into pdb there, skinning.py:183) raises: at once on the
            CPU; on the GPU the flag stays on the device and travels in the next read-back the path performs anyway (_lib.defer_check
            -- NOT a device-side assert: a failed torch._assert_async ends a ROCm process as an anonymous "HSA hardware exception").
            Until then the foot of an empty quadrant is vertex 0: a wrong skeleton for one step, never an out-of-range index."""
            if not seq_shape.is_cuda and not bool(quadrant.any(dim=-1).all()):
                raise RuntimeError("estimate_bones: no vertex in a leg quadrant (the reference drops into pdb here, skinning.py:183)")
            y_in = torch.where(quadrant, ys, torch.full_like(ys, float("inf")))
            return torch.gather(seq_shape, 2, y_in.argmin(dim=-1)[..., None, None].expand(-1, -1, 1, 3))  # [B,F,1,3]

        if QUADRANT_POPULATION is not None:  # diagnostic hook (tools/fauna_quadrant_diag.py): vertices per leg quadrant, kept on the device
            QUADRANT_POPULATION.append(torch.stack([q.sum(dim=-1).min() for q in quadrants]))
        if seq_shape.is_cuda:  # ONE flag for the four quadrants (one stack, one any, one all -- not four of each)
            from ..._lib import defer_check

            defer_check(torch.stack(quadrants).any(dim=-1).all(),
                        "estimate_bones: no vertex in a leg quadrant (the reference drops into pdb here, skinning.py:183)")
        feet = [foot_of(q) for q in quadrants]
        ramp = torch.linspace(0.0, 1.0, n_leg_bones + 1, device=seq_shape.device)[None, None, :, None]
        if legs_to_body_joint_indices is None:
            legs_to_body_joint_indices = [None, None, None, None]
        if compute_kinematic_chain:
            # attachment joint = the body joint nearest in z to the foot, fixed by the first instance and shared by all (:187-192); legs
            # 2 / 3 reuse the joints of legs 1 / 0 (:213-216).  Whatever is not prescribed is found on the device and read back in ONE
            # transfer -- the kinematic chain is a Python structure that depends on it, so this read-back is inherent when the chain is
            # rebuilt (once per epoch; Fauna: every iteration); with a cached chain (compute_kinematic_chain=False) nothing is read back.
            need = [i for i in (0, 1) if legs_to_body_joint_indices[i] is None]
            if need:
                nearest = torch.stack([torch.argmin((bones_pred[0, 0, :, 1, 2] - feet[i][0, 0, 0, 2]).abs()) for i in need])
                if nearest.is_cuda:  # (the ONE read-back of a chain rebuild also carries the quadrant flags deferred above)
                    from ..._lib import read_back

                    nearest = read_back(nearest)
                nearest = nearest.tolist()
                for i, j in zip(need, nearest):
                    legs_to_body_joint_indices[i] = int(j)
        start = n_body_bones
        leg_bones_all = []
        leg_auxs = [] if compute_kinematic_chain else aux["legs"]
        for i in range(4):
            if compute_kinematic_chain:
                body_bone_idx = legs_to_body_joint_indices[1] if i == 2 else (legs_to_body_joint_indices[0] if i == 3 else legs_to_body_joint_indices[i])
                legs_to_body_joint_indices[i] = body_bone_idx  # written back into the caller's list, as the reference does (:220)
                leg_b2j, leg_chain, leg_ids = build_kinematic_chain(n_leg_bones, start_bone_idx=start)
                kinematic_chain = update_body_kinematic_chain(kinematic_chain, leg_chain, body_bone_idx, leg_ids, attach_legs_to_body)
                leg_auxs.append({"body_bone_idx": body_bone_idx, "leg_bones_to_joints": leg_b2j})
                start += n_leg_bones
            else:
                body_bone_idx, leg_b2j = leg_auxs[i]["body_bone_idx"], leg_auxs[i]["leg_bones_to_joints"]
            leg_joints = feet[i] * (1 - ramp) + bones_pred[:, :, body_bone_idx, 1][:, :, None, :] * ramp
            leg_bones_all.append(_joints_to_bones(leg_joints, leg_b2j))
        all_bones = torch.cat([bones_pred] + leg_bones_all, dim=2)
    else:
        all_bones = bones_pred

    if compute_kinematic_chain:
        aux["kinematic_chain"] = kinematic_chain
        if n_leg_bones > 0:
            aux["legs"] = leg_auxs
        return all_bones.detach(), kinematic_chain, aux
    return all_bones.detach()


# ------------------------------------------------------------------------------------------------ rotations (public API)
def _axis_angle_rotation(axis: str, angle: torch.Tensor) -> torch.Tensor:
    """Rotation about one coordinate axis, [...] -> [...,3,3] (PyTorch3D convention; reference :285-312)."""
    c, s = torch.cos(angle), torch.sin(angle)
    o, z = torch.ones_like(angle), torch.zeros_like(angle)
    if axis == "X":
        flat = (o, z, z, z, c, -s, z, s, c)
    elif axis == "Y":
        flat = (c, z, s, z, o, z, -s, z, c)
    elif axis == "Z":
        flat = (c, -s, z, s, c, z, z, z, o)
    else:
        raise ValueError("letter must be either X, Y or Z.")
    return torch.stack(flat, -1).reshape(angle.shape + (3, 3))


@fp32_region
def euler_angles_to_matrix(euler_angles: torch.Tensor, convention: str) -> torch.Tensor:
    """Euler angles (radians) [...,3] -> rotation matrices [...,3,3] (reference :315-340)."""
    if euler_angles.dim() == 0 or euler_angles.shape[-1] != 3:
        raise ValueError("Invalid input euler angles.")
    if len(convention) != 3:
        raise ValueError("Convention must have 3 letters.")
    if convention[1] in (convention[0], convention[2]):
        raise ValueError(f"Invalid convention {convention}.")
    for letter in convention:
        if letter not in ("X", "Y", "Z"):
            raise ValueError(f"Invalid letter {letter} in convention string.")
    m = [_axis_angle_rotation(c, e) for c, e in zip(convention, torch.unbind(euler_angles, -1))]
    return m[0] @ m[1] @ m[2]


def _estimate_bone_rotation(forward):
    """Rest frame of a bone: columns (right, up, forward) with right ~ +x (reference :251-270)."""
    forward = F.normalize(forward, p=2, dim=-1)
    right = torch.tensor([1.0, 0.0, 0.0], dtype=forward.dtype, device=forward.device).expand_as(forward)
    up = F.normalize(torch.cross(forward, right, dim=-1), p=2, dim=-1)
    right = torch.cross(up, forward, dim=-1)
    up = F.normalize(up, p=2, dim=-1)
    return torch.stack([right, up, forward], dim=-1)


def _prepare_transform_mtx(rotation=None, translation=None):
    """4x4 from a rotation and/or translation batch (reference :343-357)."""
    n = len(rotation) if rotation is not None else (len(translation) if translation is not None else 1)
    ref = rotation if rotation is not None else translation
    mtx = torch.eye(4, device=ref.device if ref is not None else None)[None].repeat(n, 1, 1)
    if rotation is not None:
        mtx[:, :3, :3] = rotation
    if translation is not None:
        mtx[:, :3, 3] = translation
    return mtx


def _invert_transform_mtx(mtx):
    """Inverse of a rigid 4x4 batch (reference :360-366)."""
    r, t = mtx[:, :3, :3], mtx[:, :3, 3]
    inv = torch.eye(4, device=mtx.device)[None].repeat(len(mtx), 1, 1)
    inv[:, :3, :3] = r.transpose(1, 2)
    inv[:, :3, 3] = -(r.transpose(1, 2) @ t.unsqueeze(-1)).squeeze(-1)
    return inv


# ------------------------------------------------------------------------------------------------ skinning
_chain_cache = {}


def _chain_index(kinematic_tree, device):
    """[K, D] long: for bone k (row k) the chain root -> ... -> k, front-padded with the identity slot K."""
    key = (repr(kinematic_tree), str(device))
    hit = _chain_cache.get(key)
    if hit is None:
        K = len(kinematic_tree)
        chains = {}
        for bone_id, _ in kinematic_tree:
            parents = [p for p, ch in kinematic_tree if bone_id in ch]  # listed root first (reference :392-395)
            chains[bone_id] = parents + [bone_id]
        assert sorted(chains) == list(range(K)), "kinematic_tree must list every bone exactly once"
        depth = max(len(c) for c in chains.values())
        idx = torch.full((K, depth), K, dtype=torch.long)
        for k, c in chains.items():
            idx[k, depth - len(c):] = torch.tensor(c)
        hit = idx.to(device)
        if len(_chain_cache) > 64:
            _chain_cache.clear()
        _chain_cache[key] = hit
    return hit


_chain32_cache = {}


def _chain_index32(kinematic_tree, device):
    """int32 [K,D] chain table for the HIP kernel: root -> leaf, front-padded with -1."""
    key = (repr(kinematic_tree), str(device))
    hit = _chain32_cache.get(key)
    if hit is None:
        idx = _chain_index(kinematic_tree, device)
        K = idx.shape[0]
        hit = torch.where(idx == K, torch.full_like(idx, -1), idx).to(torch.int32).contiguous()
        if len(_chain32_cache) > 64:
            _chain32_cache.clear()
        _chain32_cache[key] = hit
    return hit


@fp32_region
def bone_transforms_torch(bones, kinematic_tree, deform_params):
    """(torch formulation; skinning() itself uses the HIP kernel csrc/bones.hip, this one documents and cross-checks it.)
    World transform of every bone for every image: [B*F, K, 4, 4].

    M_k = L_root ... L_parent(k) L_k with L_i = Rest_i . Rot_i . Rest_i^-1 (reference :389-417), i.e. a rotation by
    the bone's Euler angles about its start joint expressed in its rest frame.  bones [1|B,1|F,K,2,3], deform_params [B,F,K,3].
    """
    B, Fr, K = deform_params.shape[:3]
    dev = deform_params.device
    if bones.shape[0] == 1 and bones.shape[1] == 1:
        bb = bones.reshape(1, K, 2, 3)  # shared skeleton: broadcast over the batch below
    else:
        bb = bones.expand(B, Fr, K, 2, 3).reshape(B * Fr, K, 2, 3)
    joint_b = bb[:, :, 0]
    rest_b = _estimate_bone_rotation((bb[:, :, 1] - bb[:, :, 0]).reshape(-1, 3)).reshape(bb.shape[0], K, 3, 3)
    rot = euler_angles_to_matrix(deform_params.reshape(B * Fr * K, 3), "XYZ").reshape(B * Fr, K, 3, 3)
    R = rest_b @ rot @ rest_b.transpose(-1, -2)
    t = joint_b - (R @ joint_b[..., None])[..., 0]
    L = torch.zeros(B * Fr, K + 1, 4, 4, dtype=rot.dtype, device=dev)
    L[:, :K, :3, :3] = R
    L[:, :K, :3, 3] = t
    L[:, :, 3, 3] = 1.0
    L[:, K, 0, 0] = L[:, K, 1, 1] = L[:, K, 2, 2] = 1.0  # identity slot for chain padding
    idx = _chain_index(kinematic_tree, dev)
    M = L[:, idx[:, 0]]
    for d in range(1, idx.shape[1]):
        M = M @ L[:, idx[:, d]]
    return M


class _LazyAux(dict):
    """aux dict whose 'vertices_to_bones' ([K,B,F,V], unused by any caller on the training path) and 'posed_bones' (read by the bone
    smoothness terms of the sequence models and by the visualisation only: AnimalModel.py:348-352,604-605) are computed on first access --
    differentiable like the reference's, from the transforms the skinning launch returned."""

    def __init__(self, make_weights, make_posed=None):
        super().__init__()
        self._lazy = {"vertices_to_bones": make_weights}
        if make_posed is not None:
            self._lazy["posed_bones"] = make_posed

    def __missing__(self, key):
        if key in self._lazy:
            self[key] = self._lazy[key]()
            return self[key]
        raise KeyError(key)

    def __contains__(self, key):
        return key in self._lazy or super().__contains__(key)

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default


FUSED_POSE = True  # chain composition inside the skinning launches (a3d_skin_pose_*); False = a3d_bone_transforms_* + a3d_skin_*


@fp32_region
def skinning(v_pos, bones_pred, kinematic_tree, deform_params, output_posed_bones=False, temperature=1):
    """Linear-blend skinning (reference :369-439).

    v_pos [1|B,1|F,V,3], bones_pred [1|B,1|F,K,2,3] (no gradient), deform_params [B,F,K,3] radians.
    Returns (verts [B,F,V,3], aux{bones_pred, vertices_to_bones, posed_bones}).  Gradients flow to ``v_pos``
    (through the affine maps; the weights see a detached copy, reference :377) and to ``deform_params``.
    """
    B, Fr = deform_params.shape[:2]
    K, V = bones_pred.shape[2], v_pos.shape[-2]
    chain32 = _chain_index32(kinematic_tree, deform_params.device)

    def flat(x, tail):
        if x.shape[0] == 1 and x.shape[1] == 1:
            return x.reshape(1, *tail)
        return x.expand(B, Fr, *tail).reshape(B * Fr, *tail)

    v_flat = flat(v_pos, (V, 3))
    bones_flat = flat(bones_pred.detach(), (K, 2, 3))
    if FUSED_POSE and ops.skin_pose_supported(K, chain32.shape[1]):
        # kinematic chain + blend in ONE launch each way (csrc/skin.hip): every work-group composes its image's transforms itself
        out, T = ops.skin_pose(v_flat, bones_flat, deform_params.reshape(B * Fr, K, 3), chain32, temperature)
        out = out.view(B, Fr, V, 3)
    else:
        # per-bone world transforms from the kinematic chain: one HIP launch (csrc/bones.hip), gradients reach deform_params
        T = ops.bone_transforms(bones_flat, deform_params.reshape(B * Fr, K, 3), chain32)  # [B*F,K,12]
        out = ops.skin(v_flat, bones_flat, T, temperature).view(B, Fr, V, 3)

    def weights():
        w = ops.skin_weights(v_flat.detach(), bones_flat, B * Fr, temperature)  # [K,Bw,V]
        if w.shape[1] == 1:
            return w.view(K, 1, 1, V)
        bx, fx = max(v_pos.shape[0], bones_pred.shape[0]), max(v_pos.shape[1], bones_pred.shape[1])
        return w.view(K, B, Fr, V)[:, :bx, :fx]  # broadcast dims hold duplicates

    def posed_bones():
        ends = bones_pred.detach().expand(B, Fr, K, 2, 3).reshape(B * Fr, K, 2, 3)
        T34 = T.view(B * Fr, K, 3, 4)
        posed = torch.einsum("nkij,nkej->nkei", T34[..., :3], ends) + T34[:, :, None, :, 3]
        return posed.view(B, Fr, K, 2, 3)

    aux = _LazyAux(weights, fp32_region(posed_bones) if output_posed_bones else None)
    aux["bones_pred"] = bones_pred
    return out, aux
