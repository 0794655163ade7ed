"""``use_python`` is accepted for signature parity (reference default False = its JIT CUDA plugin, which does not exist on this
platform); both values take the torch path below -- the one every config of the reference selects (render.py:72,278)."""
import torch
import torch.nn.functional as F

from ...._lib import fp32_region

NORMAL_THRESHOLD = 0.1  # reference renderutils/bsdf.py:13
HIP_XFM_POINTS = True  # xfm_points on the GPU as a3d_xfm_points_fwd / _bwd (False: the padded torch matmul, e.g. for double precision)


def _dot(x, y):
    return torch.sum(x * y, -1, keepdim=True)


def _hip_xfm_ok(points, matrix):
    if not (HIP_XFM_POINTS and points.is_cuda and matrix.is_cuda and points.dtype == torch.float32 and matrix.dtype == torch.float32):
        return False
    if points.dim() != 3 or matrix.dim() != 3 or points.shape[2] != 3 or tuple(matrix.shape[1:]) != (4, 4):
        return False
    b = max(points.shape[0], matrix.shape[0])
    return points.shape[0] in (1, b) and matrix.shape[0] in (1, b)


@fp32_region
def xfm_points(points, matrix, use_python=False):
    """[B|1,V,3] x [B,4,4] -> homogeneous [B,V,4] = [p,1] . M^T (reference ops.py:524-525).

    One padded batched matmul (rocBLAS); autograd reaches both the points and the matrix (camera pose).
    """
    if _hip_xfm_ok(points, matrix):
        from .... import ops  # one launch each way (csrc/xfm.hip) instead of pad + bmm, and two more bmm + a slice backward

        out = ops.xfm_points(points, matrix)
    else:
        out = torch.matmul(F.pad(points, pad=(0, 1), mode="constant", value=1.0), torch.transpose(matrix, 1, 2))
    if torch.is_anomaly_enabled():
        assert torch.all(torch.isfinite(out)), "Output of xfm_points contains inf or NaN"
    return out


@fp32_region
def xfm_vectors(vectors, matrix, use_python=False):
    """Direction transform (w = 0), reference ops.py:533-549."""
    out = torch.matmul(F.pad(vectors, pad=(0, 1), mode="constant", value=0.0), torch.transpose(matrix, 1, 2))[..., 0:3].contiguous()
    if torch.is_anomaly_enabled():
        assert torch.all(torch.isfinite(out)), "Output of xfm_vectors contains inf or NaN"
    return out


def prepare_shading_normal(pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm, two_sided_shading=True, opengl=True,
                           use_python=False):
    """Final shading normal (reference ops.py:194-227 -> bsdf.py:46-51): optional tangent-space perturbation,
    two-sided flip by the geometric normal, bend toward the geometric normal at grazing view angles.

    The hot path passes perturbed_nrm=None (render.py:71), for which the tangent frame cancels exactly and
    ``smooth_tng`` is not touched (so it may be None).
    """
    n = F.normalize(smooth_nrm, dim=-1)
    view = F.normalize(view_pos - pos, dim=-1)
    if perturbed_nrm is not None:
        t = F.normalize(smooth_tng, dim=-1)
        bt = F.normalize(torch.cross(t, n, dim=-1), dim=-1)
        sign = -1.0 if opengl else 1.0
        n = F.normalize(t * perturbed_nrm[..., 0:1] + sign * bt * perturbed_nrm[..., 1:2] + n * torch.clamp(perturbed_nrm[..., 2:3], min=0.0),
                        dim=-1)
    else:
        n = F.normalize(n, dim=-1)  # (0,0,1) perturbation: n*1, re-normalised (bsdf.py:38-44)
    g = geom_nrm
    if two_sided_shading:
        front = _dot(g, view) > 0
        n = torch.where(front, n, -n)
        g = torch.where(front, g, -g)
    t = torch.clamp(_dot(view, n) / NORMAL_THRESHOLD, min=0, max=1)
    out = torch.lerp(g, n, t)
    if torch.is_anomaly_enabled():
        assert torch.all(torch.isfinite(out)), "Output of prepare_shading_normal contains inf or NaN"
    return out
