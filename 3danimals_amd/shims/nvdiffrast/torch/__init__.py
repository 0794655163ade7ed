"""``nvdiffrast.torch``-compatible operator API on top of the HIP kernels.

Put ``3danimals_amd/shims`` on ``sys.path`` and the reference's unchanged callers --
``dr.RasterizeGLContext()`` (AnimalModel.py:235-236), ``dr.DepthPeeler`` / ``dr.interpolate`` / ``dr.antialias``
(render.py:24,264-267,292-294), ``dr.rasterize`` (render.py:351, visualize_results.py:225-229) -- run on MI355X
without OpenGL or CUDA.  Instanced and range mode, ``grad_db``, ``diff_attrs`` and ``pos_gradient_boost`` are covered; ``dr.texture`` has
no call site on the reconstruct-and-render path (only EnvironmentLight, Texture2D mips and image_grad use it, all dead code in every
config): its 2-D bilinear / nearest tap is provided in torch, mip-mapped and cube-map sampling raise.
"""
import importlib

import torch

_ops = importlib.import_module("3danimals_amd.ops")


class _Context:
    """Opaque rasteriser handle.  nvdiffrast's owns an EGL/GL (or CUDA) context; the HIP rasteriser is stateless."""

    def __init__(self, output_db=True, mode="automatic", device=None):
        self.output_db = output_db
        self.device = device

    def set_context(self):
        pass

    def release_context(self):
        pass


class RasterizeGLContext(_Context):
    pass


class RasterizeCudaContext(_Context):
    def __init__(self, device=None):
        super().__init__(output_db=True, device=device)


class _LazyRastDb:
    """rast_db computed on first use.  The reference's render.py receives it from every rasterize call and discards it (render.py:24
    passes rast_db=None), so the ~30 torch launches over [B,H,W,3,4] gathers are only paid by a caller that actually looks at it:
    any torch function applied to the object, an attribute (.shape, .dtype ...) or an index materialises the tensor."""

    def __init__(self, make):
        self._make, self._value = make, None

    def materialize(self):
        if self._value is None:
            self._value, self._make = self._make(), None
        return self._value

    def __getattr__(self, name):
        return getattr(self.materialize(), name)

    def __getitem__(self, index):
        return self.materialize()[index]

    def __len__(self):
        return len(self.materialize())

    def __setitem__(self, index, value):
        self.materialize()[index] = value

    def __iter__(self):
        return iter(self.materialize())

    def __bool__(self):
        return bool(self.materialize())

    def __repr__(self):
        return repr(self.materialize()) if self._value is not None else "<rast_db (not computed yet)>"

    def __neg__(self):
        return -self.materialize()

    def __pos__(self):
        return self.materialize()

    def __abs__(self):
        return abs(self.materialize())

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        unwrap = lambda a: a.materialize() if isinstance(a, _LazyRastDb) else a
        args = tuple(type(a)(unwrap(x) for x in a) if isinstance(a, (list, tuple)) else unwrap(a) for a in args)
        kwargs = {k: unwrap(v) for k, v in (kwargs or {}).items()}
        return func(*args, **kwargs)


def _lazy_binary(name):
    def op(self, other):
        other = other.materialize() if isinstance(other, _LazyRastDb) else other
        return getattr(self.materialize(), name)(other)

    op.__name__ = name
    return op


# arithmetic and comparisons act on the tensor itself (``rast_db * 2``, ``1 - rast_db``, ``rast_db > 0`` ...): the stand-in is meant for
# unchanged nvdiffrast callers, not only for the reference's own, which hand the object straight to scale_img / interpolate
for _n in ("add", "sub", "mul", "truediv", "floordiv", "mod", "pow", "matmul", "and", "or", "xor"):
    setattr(_LazyRastDb, f"__{_n}__", _lazy_binary(f"__{_n}__"))
    setattr(_LazyRastDb, f"__r{_n}__", _lazy_binary(f"__r{_n}__"))
for _n in ("lt", "le", "gt", "ge", "eq", "ne"):
    setattr(_LazyRastDb, f"__{_n}__", _lazy_binary(f"__{_n}__"))
_LazyRastDb.__hash__ = object.__hash__  # (defining __eq__ would otherwise make the object unhashable)


def _rast_db(pos, tri, rast, grad_db):
    if grad_db:
        return _LazyRastDb(lambda: _ops.rasterize_db(pos, tri, rast))
    return _LazyRastDb(lambda: _ops.rasterize_db(pos.detach(), tri, rast.detach()))


def _check(pos, tri, resolution, ranges=None):
    if not (torch.is_tensor(pos) and torch.is_tensor(tri)):
        raise RuntimeError("pos and tri must be tensors")
    if ranges is None:
        if pos.dim() != 3 or pos.shape[-1] != 4:
            raise RuntimeError("instanced mode: pos must have shape [minibatch, num_vertices, 4]")
    else:  # range mode: one shared vertex array, image b renders the triangles ranges[b] = (first, count)
        if pos.dim() != 2 or pos.shape[-1] != 4:
            raise RuntimeError("range mode: pos must have shape [num_vertices, 4]")
        if not torch.is_tensor(ranges) or ranges.dim() != 2 or ranges.shape[-1] != 2 or ranges.dtype != torch.int32 or ranges.is_cuda:
            raise RuntimeError("range mode: ranges must be a CPU int32 tensor of shape [minibatch, 2]")
    if tri.dim() != 2 or tri.shape[-1] != 3:
        raise RuntimeError("tri must have shape [num_triangles, 3]")
    if len(resolution) != 2:
        raise RuntimeError("resolution must be [height, width]")


def _rasterize_ranges(pos, tri, resolution, ranges, prev=None):
    """Range mode (nvdiffrast: pos [V,4], ranges [B,2] = (first triangle, count) per image): one rasterisation per image over its slice
    of the triangle list, ids re-based to the full list.  No config of the reference uses it (render.py:292-294 is instanced); it
    exists so that the stand-in covers the operator's whole signature.  Gradients reach the shared ``pos`` from every image."""
    layers = []
    for b, (first, count) in enumerate(ranges.tolist()):
        if count <= 0:
            layers.append(torch.zeros((1, int(resolution[0]), int(resolution[1]), 4), dtype=torch.float32, device=pos.device))
            continue
        r = _ops.rasterize(pos[None], tri[first:first + count].contiguous(), resolution, prev=None if prev is None else prev[b:b + 1])
        ids = r[..., 3:]
        layers.append(torch.cat((r[..., :3], torch.where(ids > 0, ids + float(first), ids)), dim=-1))
    return torch.cat(layers, dim=0)


def rasterize(glctx, pos, tri, resolution, ranges=None, grad_db=True):
    """-> (rast [B,H,W,4] = (u, v, z/w, triangle_id+1), rast_db [B,H,W,4] = (du/dX, du/dY, dv/dX, dv/dY)).  The image-space
    derivatives are analytic (ops.rasterize_db, torch ops) and computed on first use: no caller on the training path consumes them
    (render.py:24)."""
    _check(pos, tri, resolution, ranges)
    if ranges is not None:
        rast = _rasterize_ranges(pos, tri, resolution, ranges)
        return rast, _rast_db(pos[None], tri, rast, grad_db)
    rast = _ops.rasterize(pos, tri, resolution)
    return rast, _rast_db(pos, tri, rast, grad_db)


class DepthPeeler:
    def __init__(self, glctx, pos, tri, resolution, ranges=None, grad_db=True):
        _check(pos, tri, resolution, ranges)
        self.pos, self.tri, self.resolution, self.grad_db, self.ranges = pos, tri, resolution, grad_db, ranges
        self.layer, self._prev = 0, None

    def __enter__(self):
        return self

    def __exit__(self, *args):
        return False

    def rasterize_next_layer(self):
        """Layer 0 = rasterize(); layer n = the nearest surface strictly behind layer n-1 (depth peeling)."""
        if self.ranges is not None:
            prev = self._prev
            if prev is not None:  # the previous layer's ids are indices into the full list: back to each image's slice for the peel
                first = self.ranges[:, 0].to(prev.device).float().view(-1, 1, 1, 1)
                ids = prev[..., 3:]
                prev = torch.cat((prev[..., :3], torch.where(ids > 0, ids - first, ids)), dim=-1)
            rast = _rasterize_ranges(self.pos, self.tri, self.resolution, self.ranges, prev=prev)
            pos_db = self.pos[None]
        else:
            rast = _ops.rasterize(self.pos, self.tri, self.resolution, prev=self._prev)
            pos_db = self.pos
        self.layer += 1
        self._prev = rast.detach()
        return rast, _rast_db(pos_db, self.tri, rast, self.grad_db)


def interpolate(attr, rast, tri, rast_db=None, diff_attrs=None):
    """-> (out [B,H,W,C], out_da).  With ``rast_db`` and ``diff_attrs`` ('all' or a list of attribute indices) out_da
    [B,H,W,2*len(diff_attrs)] holds (dA/dX, dA/dY) per selected attribute: dA/dX = du/dX (A0 - A2) + dv/dX (A1 - A2) (torch ops)."""
    if attr.dim() == 2:  # range mode: one shared attribute array
        attr = attr[None]
    out = _ops.interpolate(attr, rast, tri)
    if rast_db is None or diff_attrs is None:
        return out, torch.empty(0, device=rast.device)
    if isinstance(rast_db, _LazyRastDb):
        rast_db = rast_db.materialize()
    return out, _ops.interpolate_da(attr, rast, tri, rast_db, diff_attrs)


class _ScaleGradient(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.scale, None


def antialias(color, rast, pos, tri, topology_hash=None, pos_gradient_boost=1.0):
    """``pos_gradient_boost``: the gradient that reaches ``pos`` through the silhouette blends is multiplied by it (nvdiffrast's knob; 1.0
    everywhere in the reference).  ``pos`` [B,V,4], or [V,4] (range mode).  ``topology_hash`` is accepted and ignored: the topology is
    cached per triangle list."""
    if pos_gradient_boost != 1.0:
        pos = _ScaleGradient.apply(pos, float(pos_gradient_boost))
    return _ops.antialias(color, rast, pos if pos.dim() == 3 else pos[None], tri)


def antialias_construct_topology_hash(tri):
    return _ops.aa_topology(_ops.tri_int32(tri), int(tri.max().item()) + 1)


def texture(tex, uv, uv_da=None, mip_level_bias=None, mip=None, filter_mode="auto", boundary_mode="wrap", max_mip_level=None):
    """dr.texture for 2-D textures WITHOUT mip-mapping: tex [1|B,Th,Tw,C], uv [B,H,W,2] in texture units (0..1, texel centres at
    (i + 0.5) / size) -> [B,H,W,C]; filter_mode 'nearest' | 'linear' ('auto' = 'linear' when no uv_da / mip_level_bias is given),
    boundary_mode 'wrap' | 'clamp' | 'zero'.  Plain torch (differentiable w.r.t. tex and uv): no config of the reference reaches
    dr.texture from the reconstruct-and-render path (the texture is a coordinate MLP, render.py:53-57); this serves the off-path callers
    that only need a bilinear tap -- texture2d_mip's backward (texture.py:32), image_grad's tap (regularizer.py:24), the FG look-up table
    (light.py:118).  Mip-mapped and cube-map sampling (Texture2D.sample with uv_da, EnvironmentLight) raise: out of scope (SURVEY.md
    section 2 row 12)."""
    if filter_mode == "auto":
        filter_mode = "linear-mipmap-linear" if (uv_da is not None or mip_level_bias is not None) else "linear"
    if filter_mode not in ("nearest", "linear"):
        raise NotImplementedError(f"dr.texture filter_mode {filter_mode!r}: mip-mapped sampling is not on the reconstruct-and-render path")
    if boundary_mode == "cube" or tex.dim() != 4 or uv.shape[-1] != 2:
        raise NotImplementedError("dr.texture: cube maps are not on the reconstruct-and-render path")
    if boundary_mode not in ("wrap", "clamp", "zero"):
        raise ValueError(f"dr.texture: unknown boundary_mode {boundary_mode!r}")
    B = uv.shape[0]
    assert tex.shape[0] in (1, B), "texture batch must be 1 or the uv batch"
    Th, Tw, C = tex.shape[1:]
    x, y = uv[..., 0] * Tw - 0.5, uv[..., 1] * Th - 0.5  # continuous texel coordinates (texel i covers [i, i + 1) in texel units)
    bidx = torch.arange(B, device=uv.device).view(B, *([1] * (uv.dim() - 2))).expand(uv.shape[:-1]) if tex.shape[0] == B else torch.zeros_like(x, dtype=torch.long)

    def tap(ix, iy):
        if boundary_mode == "wrap":
            valid, ix, iy = None, ix % Tw, iy % Th
        else:
            valid = ((ix >= 0) & (ix < Tw) & (iy >= 0) & (iy < Th)) if boundary_mode == "zero" else None
            ix, iy = ix.clamp(0, Tw - 1), iy.clamp(0, Th - 1)
        v = tex[bidx, iy, ix]
        return v if valid is None else v * valid[..., None].to(v.dtype)

    if filter_mode == "nearest":
        return tap(torch.floor(x + 0.5).long(), torch.floor(y + 0.5).long())
    x0, y0 = torch.floor(x), torch.floor(y)
    fx, fy = (x - x0)[..., None], (y - y0)[..., None]
    x0, y0 = x0.long(), y0.long()
    top = tap(x0, y0) * (1 - fx) + tap(x0 + 1, y0) * fx
    bot = tap(x0, y0 + 1) * (1 - fx) + tap(x0 + 1, y0 + 1) * fx
    return top * (1 - fy) + bot * fy


def get_log_level():
    return 1


def set_log_level(level):
    pass
