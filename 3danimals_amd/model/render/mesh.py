"""Mesh container + make_mesh -- public API of /root/reference/model/render/mesh.py, HIP normals underneath.

Differences that do not change results on the hot path:
* vertex normals (auto_normals, reference :276-304) come from csrc/normals.hip (forward + backward);
* tangents (compute_tangents, reference :310-350) are LAZY: ``shade`` forces perturbed_nrm=None (render.py:71), for
  which value and gradient are independent of the tangent field, so ``v_tng`` is only built
  -- with the reference's torch expressions -- when something actually reads it ('tangent' render mode, export);
* ``v_tex`` is not copied B times: batch expansion uses views.
"""
from __future__ import annotations

import os
import weakref

import torch

from ... import ops
from . import util


LAZY_NORMALS = True  # auto_normals defers the kernel to the first read of Mesh.v_nrm (same values; False = at make_mesh time)
PAIR_NORMALS = True  # ... and a pending small mesh (<= PAIR_MAX_BATCH images: the canonical one) over the SAME triangle list rides in that launch
PAIR_MAX_BATCH = 2
RIDE_NORMALS = os.environ.get("A3D_RIDE_NORMALS", "1") != "0"  # ... and the whole pending pass rides in the rasteriser's triangle launch when the mesh is rendered first (render_mesh)
_pending_normals = {}  # (storage pointer, shape, version) of t_pos_idx -> [weakref to meshes whose normals are still pending]


def _tri_key(t):
    return (t.data_ptr(), tuple(t.shape), t._version, str(t.device))


def _register_pending(m):
    if PAIR_NORMALS and m._lazy_nrm is not None and m._v_nrm is None and m.t_pos_idx is not None:
        if len(_pending_normals) > 32:
            _pending_normals.clear()
        _pending_normals.setdefault(_tri_key(m.t_pos_idx), []).append(weakref.ref(m))


class Mesh:
    """Batched vertices sharing one topology: v_pos [B,V,3], t_pos_idx [1,F,3] int64 (reference mesh.py:21-175)."""

    def __init__(self, v_pos=None, t_pos_idx=None, v_nrm=None, t_nrm_idx=None, v_tex=None, t_tex_idx=None, v_tng=None, t_tng_idx=None,
                 material=None, base=None):
        self.v_pos = v_pos
        self._v_nrm = v_nrm
        self._lazy_nrm = None  # grad mode at make_mesh time while the normals are still to be computed (LAZY_NORMALS)
        self.v_tex = v_tex
        self._v_tng = v_tng
        self._lazy_tng = False
        self.t_pos_idx = t_pos_idx
        self.t_nrm_idx = t_nrm_idx
        self.t_tex_idx = t_tex_idx
        self._t_tng_idx = t_tng_idx
        self.material = material
        if base is not None:
            self.copy_none(base)

    # normals on demand: make_mesh is called for the canonical, the deformed and the posed mesh of every step (reference
    # BasePredictorBase.py / InstancePredictorBase.py) and only the rendered one is ever asked for its normals -------------------
    @property
    def v_nrm(self):
        if self._v_nrm is None and self._lazy_nrm is not None:
            partner = self._normals_partner()
            with torch.set_grad_enabled(self._lazy_nrm):
                if partner is None:
                    self._v_nrm = ops.vertex_normals(self.v_pos, self.t_pos_idx)
                else:  # the canonical mesh's normals (read by the regulariser every iteration, AnimalModel.py:317-328) from the same launch
                    self._v_nrm, partner._v_nrm = ops.vertex_normals_pair(self.v_pos, partner.v_pos, self.t_pos_idx)
                    partner._lazy_nrm = None
            self._lazy_nrm = None
            if torch.is_anomaly_enabled():
                assert torch.all(torch.isfinite(self._v_nrm))
        return self._v_nrm

    @v_nrm.setter
    def v_nrm(self, value):
        self._v_nrm, self._lazy_nrm = value, None

    # ... or inside the rasteriser's launch: render_mesh asks for the pending pass as a job, hands it to ops.rasterize and gives it back
    def normals_job(self):
        """The pending normals of this mesh (and of its pairing partner) as an ops.NormalsJob, or None when there is nothing pending."""
        if not RIDE_NORMALS or self._v_nrm is not None or self._lazy_nrm is None or self.t_pos_idx is None or not self.v_pos.is_cuda:
            return None
        if self.t_pos_idx.shape[-2] == 0:
            return None
        job = ops.NormalsJob(self.v_pos, None, self.t_pos_idx)
        job.partner = self._normals_partner()
        if job.partner is not None:
            job.v_b = ops.f32c(job.partner.v_pos.detach())
        return job

    def take_normals(self, job, graph_pos=None):
        """Adopt the results of a job the rasteriser ran (nothing happens if it did not: the lazy path stays armed).  ``graph_pos``: the
        same positions as another output of the clip transform's node (ops.xfm_points(alias=2)) -- the normals' gradient then arrives at
        that node's backward launch instead of at an accumulation kernel in front of it."""
        if not job.done or self._v_nrm is not None:
            return
        partner = job.partner
        v_self = self.v_pos if graph_pos is None or graph_pos.shape != self.v_pos.shape else graph_pos
        with torch.set_grad_enabled(self._lazy_nrm):
            nrm_a, nrm_b = ops.vertex_normals_attach(v_self, None if partner is None else partner.v_pos, job)
        self._v_nrm, self._lazy_nrm = nrm_a, None
        if partner is not None and partner._v_nrm is None and partner._lazy_nrm is not None:
            partner._v_nrm, partner._lazy_nrm = nrm_b, None

    def _normals_partner(self):
        """Another live mesh over the same triangle list whose normals are pending too and which is small (the one canonical mesh beside
        the B posed ones): its normals are computed in this mesh's launch instead of a launch of their own later (one image = pure
        launch latency).  Speculative for at most PAIR_MAX_BATCH images; the deformed meshes (B images, never read) are not picked up."""
        if not PAIR_NORMALS or self.t_pos_idx is None or not self.v_pos.is_cuda:
            return None
        entries = _pending_normals.get(_tri_key(self.t_pos_idx))
        if not entries:
            return None
        found = None
        alive = []
        for ref in entries:
            m = ref()
            if m is None or m._lazy_nrm is None or m._v_nrm is not None:
                continue
            alive.append(ref)
            if (found is None and m is not self and m._lazy_nrm == self._lazy_nrm and m.v_pos.shape[0] <= PAIR_MAX_BATCH
                    and m.v_pos.shape[1] == self.v_pos.shape[1] and m.v_pos.device == self.v_pos.device and m.t_nrm_idx is m.t_pos_idx):
                found = m
        _pending_normals[_tri_key(self.t_pos_idx)] = alive
        return found

    # tangents on demand ----------------------------------------------------------------------------
    @property
    def v_tng(self):
        if self._v_tng is None and self._lazy_tng:
            self._v_tng = _tangents(self)
        return self._v_tng

    @v_tng.setter
    def v_tng(self, value):
        self._v_tng = value

    @property
    def t_tng_idx(self):
        if self._t_tng_idx is None and self._lazy_tng:
            return self.t_nrm_idx
        return self._t_tng_idx

    @t_tng_idx.setter
    def t_tng_idx(self, value):
        self._t_tng_idx = value

    def __len__(self):
        return len(self.v_pos)

    def copy_none(self, other):
        for name in ("v_pos", "t_pos_idx", "t_nrm_idx", "v_tex", "t_tex_idx", "material"):
            if getattr(self, name) is None:
                setattr(self, name, getattr(other, name))
        if self._v_nrm is None and self._lazy_nrm is None:
            if other._lazy_nrm is not None and self.v_pos is not other.v_pos:
                self._v_nrm = other.v_nrm  # the normals belong to the other mesh's vertices: compute them there
            else:
                self._v_nrm, self._lazy_nrm = other._v_nrm, other._lazy_nrm
                _register_pending(self)  # (the mesh make_mesh returns is a copy of the one auto_normals marked)
        if self._v_tng is None:
            self._v_tng, self._lazy_tng = other._v_tng, other._lazy_tng
        if self._t_tng_idx is None:
            self._t_tng_idx = other._t_tng_idx

    def clone(self):
        out = Mesh(base=self)
        for name in ("v_pos", "t_pos_idx", "v_nrm", "t_nrm_idx", "v_tex", "t_tex_idx", "_v_tng", "_t_tng_idx"):
            t = getattr(out, name)
            if t is not None:
                setattr(out, name, t.clone().detach())
        return out

    def detach(self):
        return self.clone()

    def _sub(self, verts, uvs):
        return make_mesh(verts, self.t_pos_idx, uvs, self.t_tex_idx, self.material)

    def extend(self, N: int):
        """Each mesh of the batch repeated N times (reference :92-108)."""
        return self._sub(self.v_pos.repeat(N, 1, 1), _expand_uv(self.v_tex, len(self.v_pos) * N))

    def deform(self, deformation):
        """v_pos + deformation [B,V,3] -> new Mesh (reference :110-122)."""
        assert deformation.shape[1] == self.v_pos.shape[1] and deformation.shape[2] == 3
        verts = self.v_pos + deformation
        return self._sub(verts, _expand_uv(self.v_tex, len(verts)))

    def get_m_to_n(self, m: int, n: int):
        return self._sub(self.v_pos[m:n, ...], self.v_tex[m:n, ...])

    def first_n(self, n: int):
        return self.get_m_to_n(0, n)

    def get_n(self, n: int):
        return self.get_m_to_n(n, n + 1)


def _expand_uv(v_tex, batch):
    if v_tex.shape[0] == batch:
        return v_tex
    if v_tex.shape[0] == 1:
        return v_tex.expand(batch, -1, -1)  # view: the reference materialises B copies of a ~50 MB atlas here (:122)
    return v_tex.repeat(batch // v_tex.shape[0], 1, 1)


def load_mesh(filename, mtl_override=None):
    raise NotImplementedError("OBJ import is outside the reconstruct-and-render hot path")


def aabb(mesh):
    return torch.min(mesh.v_pos, dim=0).values, torch.max(mesh.v_pos, dim=0).values


def _sorted_edges(attr_idx):
    idx = attr_idx[0]
    e = torch.cat((idx[:, [0, 1]], idx[:, [1, 2]], idx[:, [2, 0]]), dim=-1).view(-1, 2)
    swapped = e[:, 0] > e[:, 1]
    return torch.where(swapped[:, None], e.flip(1), e), swapped


def compute_edges(attr_idx, return_inverse=False):
    """Unique (min,max) edges of a triangle index list (reference :196-214)."""
    with torch.no_grad():
        e, _ = _sorted_edges(attr_idx)
        return torch.unique(e, dim=0, return_inverse=return_inverse)


def compute_edge_to_face_mapping(attr_idx, return_inverse=False):
    """[E,2] the two faces on either side of each unique edge (reference :219-250)."""
    with torch.no_grad():
        e, swapped = _sorted_edges(attr_idx)
        uniq, inv = torch.unique(e, dim=0, return_inverse=True)
        tris = torch.arange(attr_idx.shape[1], device=e.device).repeat_interleave(3)
        out = torch.zeros((uniq.shape[0], 2), dtype=torch.int64, device=e.device)
        out[inv[~swapped], 0] = tris[~swapped]
        out[inv[swapped], 1] = tris[swapped]
        return out


def unit_size(mesh):
    with torch.no_grad():
        vmin, vmax = aabb(mesh)
        scale = 2 / torch.max(vmax - vmin).item()
        return Mesh((mesh.v_pos - (vmax + vmin) / 2) * scale, base=mesh)


def center_by_reference(base_mesh, ref_aabb, scale):
    center = (ref_aabb[0] + ref_aabb[1]) * 0.5
    scale = scale / torch.max(ref_aabb[1] - ref_aabb[0]).item()
    return Mesh((base_mesh.v_pos - center[None, ...]) * scale, base=base_mesh)


def auto_normals(imesh):
    """Smooth vertex normals (reference :276-304) -- one HIP scatter pass + one normalise pass, see csrc/normals.hip."""
    if LAZY_NORMALS:
        out = Mesh(t_nrm_idx=imesh.t_pos_idx, base=imesh)
        out._v_nrm, out._lazy_nrm = None, torch.is_grad_enabled()
        _register_pending(out)
        return out
    v_nrm = ops.vertex_normals(imesh.v_pos, imesh.t_pos_idx)
    if torch.is_anomaly_enabled():
        assert torch.all(torch.isfinite(v_nrm))
    return Mesh(v_nrm=v_nrm, t_nrm_idx=imesh.t_pos_idx, base=imesh)


def _tangents(imesh):
    """Per-vertex tangents from the uv atlas, reference :310-350 (torch; off the training path)."""
    faces, uv_idx = imesh.t_pos_idx[0], imesh.t_tex_idx[0]
    pos = [imesh.v_pos[:, faces[:, i]] for i in range(3)]
    tex = [imesh.v_tex[:, uv_idx[:, i]] for i in range(3)]
    uve1, uve2 = tex[1] - tex[0], tex[2] - tex[0]
    pe1, pe2 = pos[1] - pos[0], pos[2] - pos[0]
    nom = pe1 * uve2[..., 1:2] - pe2 * uve1[..., 1:2]
    denom = uve1[..., 0:1] * uve2[..., 1:2] - uve1[..., 1:2] * uve2[..., 0:1]
    tang = nom / torch.where(denom > 0.0, torch.clamp(denom, min=1e-6), torch.clamp(denom, max=-1e-6))
    tsum = torch.zeros_like(imesh.v_nrm)
    cnt = torch.zeros_like(imesh.v_nrm)
    nidx = imesh.t_nrm_idx[0]
    for i in range(3):
        tsum = tsum.index_add(1, nidx[:, i], tang)
        cnt = cnt.index_add(1, nidx[:, i], torch.ones_like(tang))
    t = util.safe_normalize(tsum / cnt)
    t = util.safe_normalize(t - util.dot(t, imesh.v_nrm) * imesh.v_nrm)
    if torch.is_anomaly_enabled():
        assert torch.all(torch.isfinite(t))
    return t


def compute_tangents(imesh):
    """Kept for API parity (reference :310-350); marks tangents as lazily computable."""
    out = Mesh(base=imesh)
    out._v_tng, out._lazy_tng, out._t_tng_idx = None, True, None
    return out


def make_mesh(verts, faces, uvs, uv_idx, material):
    """verts [B,V,3], faces [1,F,3], uvs [B,Nuv,2], uv_idx [1,F,3] -> Mesh with normals (and lazy tangents) (reference :355-375)."""
    assert len(verts.shape) == 3 and len(faces.shape) == 3 and len(uvs.shape) == 3 and len(uv_idx.shape) == 3, "All components must be batched."
    assert faces.shape[0] == 1 and uv_idx.shape[0] == 1, "Every mesh must share the same edge connectivity."
    assert verts.shape[0] == uvs.shape[0], "Batch size must be consistent."
    ret = Mesh(verts, faces, v_tex=uvs, t_tex_idx=uv_idx, material=material)
    ret = auto_normals(ret)
    return compute_tangents(ret)
