"""torch.autograd.Function wrappers over the C ABI (include/a3d.h).

Every Function allocates its outputs with torch (so PyTorch's caching allocator owns all device memory),
launches on the current torch stream and returns real gradients for every differentiable input -- DDP runs
with find_unused_parameters=False in the reference.  Nothing here runs on CPU.
"""
from __future__ import annotations

import ctypes
import math
import os
from collections import OrderedDict

import torch

from . import _lib
from ._lib import call, f32c, f32h, ptr, require_device, stream


# ---------------------------------------------------------------------------------------------- index caches
class _IdentityCache:
    """Small LRU keyed by tensor identity (storage pointer, shape, version).

    The keyed tensor is kept alive inside the entry, so its storage cannot be recycled while the entry exists.
    """

    def __init__(self, maxsize=8):
        self.maxsize = maxsize
        self.data = OrderedDict()

    @staticmethod
    def key(t):
        return (t.data_ptr(), tuple(t.shape), t.dtype, t._version)

    def get(self, t, make):
        k = self.key(t)
        hit = self.data.get(k)
        if hit is not None:
            self.data.move_to_end(k)
            return hit[1]
        val = make(t)
        self.data[k] = (t, val)
        while len(self.data) > self.maxsize:
            self.data.popitem(last=False)
        return val

    def put(self, t, val):
        self.data[self.key(t)] = (t, val)
        self.data.move_to_end(self.key(t))
        while len(self.data) > self.maxsize:
            self.data.popitem(last=False)
        return val

    def peek(self, t):
        k = self.key(t)
        hit = self.data.get(k)
        if hit is None:
            return None
        self.data.move_to_end(k)
        return hit[1]

    def take(self, t):
        """peek + remove (None when absent)."""
        hit = self.data.pop(self.key(t), None)
        return None if hit is None else hit[1]


_tri32_cache = _IdentityCache()
_topo_cache = _IdentityCache()


def tri_int32(tri: torch.Tensor) -> torch.Tensor:
    """[F,3] (or [1,F,3]) index tensor -> contiguous int32 [F,3]; converted once per topology and cached
    (the reference converts with .int() at every dr.* call, render.py:182-209,292)."""
    if tri.dim() == 3:
        tri = tri[0]
    if tri.dtype == torch.int32 and tri.is_contiguous():
        return tri
    return _tri32_cache.get(tri, lambda t: t.to(torch.int32).contiguous())


class AATopology:
    """opp[F,3] for one triangle list (nvdiffrast's topology hash), built by a3d_aa_topology."""

    def __init__(self, tri32: torch.Tensor, num_vertices: int, build: bool = True, lists=None):
        """``lists`` (a VertexFaceAdjacency of the same triangle list) instead of the table: a3d_aa_analyze then looks the opposite
        vertices it needs up in the vertex -> face lists (the DMTet extraction's topology builds no hash and no table)."""
        require_device(tri32, what="aa_topology")
        F = tri32.shape[0]
        self.tri, self.lists = tri32, lists
        self.opp = torch.empty((F, 3), dtype=torch.int32, device=tri32.device) if lists is None else None
        if F > 0 and build:
            nbytes = _lib.lib().a3d_aa_hash_bytes(F)
            scratch = torch.empty(nbytes, dtype=torch.uint8, device=tri32.device)
            call("a3d_aa_topology", ptr(tri32), F, int(num_vertices), ptr(scratch), ptr(self.opp), stream())


def opposite_vertices_from_lists(adjacency) -> torch.Tensor:
    """opp[F,3] from the vertex -> face lists (a3d_aa_topology_from_lists): the table a3d_aa_topology builds with its hash."""
    tri32 = adjacency.tri
    opp = torch.empty((tri32.shape[0], 3), dtype=torch.int32, device=tri32.device)
    call("a3d_aa_topology_from_lists", ptr(tri32), tri32.shape[0], ptr(adjacency.off), ptr(adjacency.adj), ptr(opp), adjacency.stride, stream())
    return opp


def aa_topology(tri32: torch.Tensor, num_vertices: int) -> AATopology:
    hit = _topo_cache.peek(tri32)
    return hit if hit is not None else mesh_topology(tri32, num_vertices)[1]


_topology_scratch = {}


def mesh_topology(tri32: torch.Tensor, num_vertices: int):
    """(VertexFaceAdjacency, AATopology) of one triangle list, built together by a3d_mesh_topology (5 launches instead of 9) and
    entered into both caches: the normals, the G-buffer backward and the antialiasing of every mesh that shares the list hit them."""
    require_device(tri32, what="mesh_topology")
    F, V = tri32.shape[0], int(num_vertices)
    adj = VertexFaceAdjacency(tri32, V, build=False)
    topo = AATopology(tri32, V, build=False)
    # cursor + hash are kept per (device, stream, hash size): the last launch of a call re-arms them, so only the first use pays the init
    nbytes = _lib.lib().a3d_aa_hash_bytes(max(F, 1))
    key = (tri32.device, stream(), nbytes)
    kept = _topology_scratch.pop(key, None)
    clean = kept is not None and kept[0].shape[0] >= V
    if clean:
        cursor, scratch = kept
    else:
        cursor = torch.zeros(max(V, nbytes // 24), dtype=torch.int32, device=tri32.device)  # zero beyond V too: later calls of this F class may have more vertices (V <= 3F < slots)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=tri32.device)
    call("a3d_mesh_topology", ptr(tri32), V, F, ptr(adj.off), ptr(adj.adj), ptr(cursor), ptr(scratch), ptr(topo.opp), int(clean), stream())
    if len(_topology_scratch) >= 4:
        _topology_scratch.clear()
    if F > 0 or clean:  # only after a completed call whose last launch re-armed it (F == 0 launches nothing that touches the hash: a fresh,
        _topology_scratch[key] = (cursor, scratch)  # never initialised buffer must not be handed to the next call as clean)
    _adj_cache.put(tri32, adj)
    _topo_cache.put(tri32, topo)
    return adj, topo


# ---------------------------------------------------------------------------------------------- DMTet
DMTET_TOPOLOGY = True  # build the mesh topology inside the extraction (emit + one finalize launch) instead of a3d_mesh_topology on first use
_topo_sets = {}


def _topology_sets(dev, F, V):
    """Two alternating valence-count arrays (int32 [vcap], zero) per (device, stream, size class): an extraction's emit launch counts into
    one and its finalize launch zeroes the other, so only the first use of a size class pays an initialisation."""
    vcap = 1 << max(10, int(V - 1).bit_length())
    key = (dev, stream(), vcap)
    st = _topo_sets.get(key)
    if st is None:
        if len(_topo_sets) >= 4:
            _topo_sets.clear()
        st = _topo_sets[key] = dict(sets=[torch.zeros(vcap, dtype=torch.int32, device=dev) for _ in range(2)], cur=0, vcap=vcap, key=key)
    return st


DMTET_CULL_MIN_VERTS = 1 << 17
DMTET_EMIT_LISTS = os.environ.get("A3D_EMIT_LISTS", "1") != "0"  # the emit launch writes the vertex -> face lists itself (no finalize launch)
DMTET_EMIT_LISTS_MAX_STRIDE = 32
DMTET_SPECULATIVE_EMIT = os.environ.get("A3D_SPECULATIVE_EMIT", "1") != "0"  # the emit launch enqueued before the counts read-back (sizes guessed from the last extraction)
_dm_vertex_scratch = {}
_dm_grad_buffers = _IdentityCache(maxsize=2)  # vert_edge of an extraction -> its cleared SDF gradient buffer


def _dmtet_count(sdf_c, pos_c, grid, scratch, counts, vscratch, vclean, counters):
    """The count call of an extraction, by the pass the grid takes (TetGridTopology.count_pass): 'plain' / 'culled' = a3d_dmtet_count
    without / with the word groups of the file's own row order, 'ordered' = a3d_dmtet_count_ordered (any numbering).  -> the pass."""
    Ne, Nt, Nv = grid.edges32.shape[0], grid.tets32.shape[0], sdf_c.shape[0]
    which = grid.count_pass(Nv, pos_c) if hasattr(grid, "count_pass") else "plain"
    n_clear = 0 if counters is None else counters.shape[0]
    if which == "ordered":
        order = grid.spatial_order(pos_c)[0]
        call("a3d_dmtet_count_ordered", ptr(sdf_c), Nv, Ne, Nt, ctypes.addressof(order), ptr(scratch), ptr(counts), ptr(vscratch), int(vclean),
             ptr(counters), n_clear, stream())
    else:
        groups = grid.word_groups() if which == "culled" else None
        call("a3d_dmtet_count", ptr(sdf_c), ptr(grid.edges32), ptr(grid.tets32), Ne, Nt, ptr(scratch), ptr(counts), ptr(vscratch), int(vclean), Nv,
             ptr(groups[0]) if groups else None, ptr(groups[1]) if groups else None, groups[0].shape[1] if groups else 0, ptr(counters), n_clear,
             stream())
    grid._last_count_pass = which
    return which


def dmtet_count_only(pos, sdf, grid, surface_vertices=False):
    """The count call alone (measurement: tools/bench_dmtet.py); -> counts int32 [6] on the device.  With ``surface_vertices`` the flagged
    vertex plane is left dirty (no emit clears it), so every call pays its memset like a first call does."""
    pos_c, sdf_c = f32c(pos.detach()), f32c(sdf.detach()).reshape(-1)
    dev, Nv = pos_c.device, pos_c.shape[0]
    scratch = torch.empty(_lib.lib().a3d_dmtet_scratch_bytes(grid.edges32.shape[0], grid.tets32.shape[0]), dtype=torch.uint8, device=dev)
    vscratch = torch.empty(_lib.lib().a3d_dmtet_vertex_scratch_bytes(Nv), dtype=torch.uint8, device=dev) if surface_vertices else None
    counts = torch.empty(6, dtype=torch.int32, device=dev)
    _dmtet_count(sdf_c, pos_c, grid, scratch, counts, vscratch, False, None)
    return counts


def dmtet_extract(pos, sdf, grid, surface_vertices=False, for_backward=False, surface_points=0):
    """Topology + vertex placement, no autograd: (verts [V,3], faces int64 [F,3], uv_idx int64 [F,3], vert_edge int32 [V]).
    ``surface_vertices``: also the sorted int64 list of the grid vertices at the ends of sign-crossing edges (their count rides in the
    same read-back as V, n1, n2: no torch.nonzero, no second host synchronisation).
    ``for_backward``: a dmtet_verts(...) on this extraction will be differentiated -- its dense SDF gradient buffer is allocated now and
    cleared by the emit launch (picked up by dmtet_verts through the returned vert_edge tensor).
    ``surface_points`` = bucket > 0 (with surface_vertices): a sixth result, pos[idx] as a block of round_up(len(idx), bucket) rows with
    zero rows behind, written by the emit launch itself (a3d_dmtet_emit_opts.surf_pts)."""
    assert not surface_points or surface_vertices
    require_device(pos, sdf, grid.edges32, what="dmtet")
    pos_c, sdf_c = f32c(pos.detach()), f32c(sdf.detach()).reshape(-1)
    Ne, Nt, Nv = grid.edges32.shape[0], grid.tets32.shape[0], pos_c.shape[0]
    assert sdf_c.shape[0] == Nv, "sdf must have one value per grid vertex"
    dev = pos_c.device
    scratch = torch.empty(_lib.lib().a3d_dmtet_scratch_bytes(Ne, Nt), dtype=torch.uint8, device=dev)
    # the vertex bit plane is kept per (device, stream, grid size): the emit leaves it cleared, so only its first use pays the memset
    vscratch, vkey, vclean = None, (dev, stream(), Nv), False
    if surface_vertices:
        vscratch = _dm_vertex_scratch.pop(vkey, None)
        vclean = vscratch is not None
        if vscratch is None:
            vscratch = torch.empty(_lib.lib().a3d_dmtet_vertex_scratch_bytes(Nv), dtype=torch.uint8, device=dev)
    counts = torch.empty(6, dtype=torch.int32, device=dev)
    # (which count pass runs is decided by the grid: static tables let it skip the words off the surface.  Three launches (sign plane,
    # culled count, scan) against two: R = 64 (2.7e5 vertices) 14 vs 18 us back to back and equal inside the step, R = 128 32 vs 88 us;
    # small grids keep the plain pass)
    # mesh topology inside the extraction (DMTET_TOPOLOGY).  Preferred form: the emit launch writes the int32 triangle list AND the vertex ->
    # face lists themselves, every vertex owning ``stride`` slots (the grid bounds the valence: grid.face_list_stride()); the valence
    # counters it appends through are zeroed by the count call -- sized by a guess at V (the last extraction on this grid), since V is
    # only known after the read-back.  No launch of its own; the normals / antialiasing find the lists in the caches.
    stride = grid.face_list_stride() if (DMTET_TOPOLOGY and DMTET_EMIT_LISTS and hasattr(grid, "face_list_stride")) else 0
    counters = None
    if 0 < stride <= DMTET_EMIT_LISTS_MAX_STRIDE:
        guess = getattr(grid, "_last_surface_vertices", 0)
        counters = torch.empty(max(1024, -(-int(1.25 * guess) // 1024) * 1024), dtype=torch.int32, device=dev)
    which = _dmtet_count(sdf_c, pos_c, grid, scratch, counts, vscratch, vclean, counters)

    def alloc(Vn, Fn, n_surf_n):
        rows = -(-n_surf_n // surface_points) * surface_points if surface_points else 0
        return (torch.empty((Vn, 3), dtype=torch.float32, device=dev), torch.empty((Vn,), dtype=torch.int32, device=dev),
                torch.empty((Fn, 3), dtype=torch.int64, device=dev), torch.empty((Fn, 3), dtype=torch.int64, device=dev),
                torch.empty((n_surf_n,), dtype=torch.int64, device=dev) if surface_vertices else None,
                torch.empty((rows, 3), dtype=torch.float32, device=dev) if surface_points else None)

    g_sdf = torch.empty((Nv,), dtype=torch.float32, device=dev) if for_backward else None

    def emit(Vn, n1n, n2n, bufs, n_surf_n, tri32_t, cnt_t, adj_t, stride_n, listed, dev_counts):
        """a3d_dmtet_emit, or after the ordered count pass (surface items spread evenly over the planes) a3d_dmtet_emit_sparse."""
        use_lists = which != "ordered" and listed[0] >= 0 and listed[1] >= 0
        opts = _lib.DmtetEmitOpts(size=ctypes.sizeof(_lib.DmtetEmitOpts), Nv=Nv, vertex_scratch=ptr(vscratch), surf_idx=ptr(bufs[4]),
                                  g_sdf_to_clear=ptr(g_sdf), tri32=ptr(tri32_t), topo_count=ptr(cnt_t), topo_adj=ptr(adj_t),
                                  device_counts=ptr(dev_counts), n_surf=n_surf_n if surface_vertices else 0, topo_stride=stride_n,
                                  use_block_lists=int(use_lists), n_edge_blocks_listed=listed[0] if use_lists else 0,
                                  n_tet_blocks_listed=listed[1] if use_lists else 0, surf_bucket=int(surface_points), surf_pts=ptr(bufs[5]))
        call("a3d_dmtet_emit_sparse" if which == "ordered" else "a3d_dmtet_emit", ptr(pos_c), ptr(sdf_c), ptr(grid.edges32), ptr(grid.tet2edge32), Ne, Nt,
             ptr(scratch), Vn, n1n, n2n, ptr(bufs[0]), ptr(bufs[1]), ptr(bufs[2]), ptr(bufs[3]), ctypes.addressof(opts), stream())

    # SPECULATIVE emit: with the numbers of the previous extraction on this grid as a guess (+25 %), the emit launch is enqueued BEFORE
    # the host reads the counts -- the kernel takes the true sizes from the device, the GPU does not idle across the read-back, and the
    # host's wait overlaps the launch.  If anything outgrew its capacity the launch left every buffer untouched and the exact path
    # below runs as if nothing had happened.
    spec = None
    last = getattr(grid, "_last_counts", None)
    if DMTET_SPECULATIVE_EMIT and last is not None and which in ("culled", "ordered") and counters is not None and stride > 0:
        cap = lambda n, unit: max(unit, -(-int(1.25 * n) // unit) * unit)
        items = _lib.lib().a3d_dmtet_block_items()
        nbe, nbt = -(-Ne // items), -(-Nt // items)
        capV, capF, capS = min(cap(last[0], 256), counters.shape[0]), cap(last[1] + 2 * last[2], 256), cap(last[3], max(256, int(surface_points)))
        capE, capT = min(cap(last[4], 16), nbe), min(cap(last[5], 16), nbt)
        if last[0] > 0 and last[1] + last[2] > 0 and (last[4] >= 0 or which == "ordered") and capV * stride < 2 ** 31:
            bufs = alloc(capV, capF, capS)
            tri32_b = torch.empty((capF, 3), dtype=torch.int32, device=dev)
            adj_b = torch.empty(capV * stride, dtype=torch.int32, device=dev)
            emit(capV, capF, 0, bufs, capS, tri32_b, counters, adj_b, stride, (capE, capT), counts)
            spec = (capV, capF, capS, capE, capT, bufs, tri32_b, adj_b)
    # the one host sync of DMTet (the reference syncs here too, dmtet.py:110); listed_*: how many non-empty blocks the culled count
    # pass listed for the emit launch (-1: none listed)
    V, n1, n2, n_surf, listed_e, listed_t = _lib.read_back(counts).tolist()  # (+ any deferred device-side check: same transfer)
    F = n1 + 2 * n2
    grid._last_counts = (V, n1, n2, n_surf, listed_e, listed_t)
    if counters is not None:
        grid._last_surface_vertices = V
    if spec is not None and V > 0 and F > 0 and V <= spec[0] and F <= spec[1] and (not surface_vertices or n_surf <= spec[2]) \
            and (which == "ordered" or (0 <= listed_e <= spec[3] and listed_t <= spec[4])):
        bufs, tri32_b, adj_b = spec[5], spec[6], spec[7]
        verts, vert_edge, faces, uv_idx = bufs[0][:V], bufs[1][:V], bufs[2][:F], bufs[3][:F]
        idx = bufs[4][:n_surf] if surface_vertices else None
        pts = bufs[5][:-(-n_surf // surface_points) * surface_points] if surface_points else None
        tri32 = tri32_b[:F]
        adj = VertexFaceAdjacency(tri32, V, build=False, lists=(counters, adj_b, stride))
        _adj_cache.put(tri32, adj)
        _topo_cache.put(tri32, AATopology(tri32, V, build=False, lists=adj))
        _tri32_cache.put(faces, tri32)
    else:
        verts, vert_edge, faces, uv_idx, idx, pts = alloc(V, F, n_surf)
        emit_lists = counters is not None and F > 0 and 0 < V <= counters.shape[0] and V * stride < 2 ** 31
        # fallback (the guess at V was too small, or a grid with very many tets around an edge): the emit launch only counts the valences
        # and ONE more launch (a3d_mesh_topology_finalize) scans them and fills CSR lists -- instead of the conversion kernel + the four
        # launches of a3d_mesh_topology on first use
        topo = None
        if not emit_lists and DMTET_TOPOLOGY and F > 0 and 0 < V <= _lib.lib().a3d_mesh_topology_finalize_max_vertices():
            topo = _topology_sets(dev, F, V)
        tri32 = cur = nxt = lists_adj = None
        if topo is not None:
            cur, nxt = topo["sets"][topo["cur"]], topo["sets"][1 - topo["cur"]]
        if emit_lists:
            cur, lists_adj = counters, torch.empty(V * stride, dtype=torch.int32, device=dev)
        if cur is not None:
            tri32 = torch.empty((F, 3), dtype=torch.int32, device=dev)
        try:
            emit(V, n1, n2, (verts, vert_edge, faces, uv_idx, idx, pts), n_surf, tri32, cur, lists_adj, stride if emit_lists else 0, (listed_e, listed_t), None)
            adj = None
            if emit_lists:
                adj = VertexFaceAdjacency(tri32, V, build=False, lists=(counters, lists_adj, stride))
            elif topo is not None:
                adj = VertexFaceAdjacency(tri32, V, build=False)
                adj.sorted = False
                call("a3d_mesh_topology_finalize", ptr(tri32), V, F, ptr(cur), ptr(adj.off), ptr(adj.adj), ptr(nxt), topo["vcap"], stream())
                topo["cur"] ^= 1
            if adj is not None:
                _adj_cache.put(tri32, adj)
                _topo_cache.put(tri32, AATopology(tri32, V, build=False, lists=adj))
                _tri32_cache.put(faces, tri32)
        except Exception:
            if topo is not None:  # a half-used scratch pair must not be taken for clean
                _topo_sets.pop(topo["key"], None)
            raise
    if g_sdf is not None:
        _dm_grad_buffers.put(vert_edge, g_sdf)
    if not surface_vertices:
        return verts, faces, uv_idx, vert_edge
    if len(_dm_vertex_scratch) >= 4:
        _dm_vertex_scratch.clear()
    _dm_vertex_scratch[vkey] = vscratch  # only after a completed count + emit pair (a failed call leaves the buffer out of the cache)
    if surface_points:
        return verts, faces, uv_idx, vert_edge, idx, pts
    return verts, faces, uv_idx, vert_edge, idx


class _DMTetVerts(torch.autograd.Function):
    """Attaches the gradient of already-extracted surface vertices to (pos, sdf): forward returns ``verts0`` as is."""

    @staticmethod
    def forward(ctx, pos, sdf, verts0, vert_edge, grid):
        pos_c, sdf_c = f32c(pos), f32c(sdf).reshape(-1)
        ctx.save_for_backward(pos_c, sdf_c, vert_edge)
        ctx.grid, ctx.sdf_shape = grid, sdf.shape
        ctx.g_sdf = _dm_grad_buffers.take(vert_edge)  # cleared by the emit launch of dmtet_extract(for_backward=True), or None
        return verts0.detach()  # (a fresh tensor over the extraction's own buffer, which nobody else holds: no copy launch)

    @staticmethod
    def backward(ctx, g_verts):
        pos_c, sdf_c, vert_edge = ctx.saved_tensors
        Nv, V = pos_c.shape[0], vert_edge.shape[0]
        g_sdf, ctx.g_sdf = ctx.g_sdf, None  # serves ONE backward
        clear = g_sdf is not None and g_sdf.shape[0] == Nv
        if not clear:
            g_sdf = torch.empty_like(sdf_c)
        g_pos = torch.empty_like(pos_c) if ctx.needs_input_grad[0] else None
        g = f32c(g_verts) if V > 0 else None
        call("a3d_dmtet_bwd", ptr(g), ptr(pos_c), ptr(sdf_c), ptr(ctx.grid.edges32), ptr(vert_edge), V, Nv, ptr(g_pos), ptr(g_sdf), int(clear), stream())
        return g_pos, g_sdf.reshape(ctx.sdf_shape), None, None, None


def gather_rows_padded(src, idx, rows):
    """[rows,C] = src[idx] followed by zero rows (rows >= len(idx)); src [N,C] float32, idx int64 -- one launch, no autograd."""
    require_device(src, idx, what="gather_rows_padded")
    src = f32c(src)
    C = src.shape[1] if src.dim() == 2 else 1
    out = torch.empty((rows, C), dtype=torch.float32, device=src.device)
    call("a3d_dmtet_gather_rows", ptr(src), ptr(idx), idx.shape[0], rows, C, ptr(out), stream())
    return out


class _SurfaceSdf(torch.autograd.Function):
    """sdf0 with the graph of ``sdf_sub`` attached at the rows ``idx``: forward returns sdf0's values (the reference of this trick,
    sdf0.index_add(0, idx, sdf_sub - sdf_sub.detach()), adds exact zeros), backward gathers g[idx] into sdf_sub's (padded) shape."""

    @staticmethod
    def forward(ctx, sdf0, idx, sdf_sub):
        ctx.save_for_backward(idx)
        ctx.sub_shape = sdf_sub.shape
        return sdf0.detach()

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return None, None, gather_rows_padded(g.reshape(g.shape[0], -1), idx, ctx.sub_shape[0]).view(ctx.sub_shape)


def surface_sdf(sdf0, idx, sdf_sub):
    """sdf0 [Nv,1] (no graph) whose rows idx [n] carry the graph of sdf_sub [>= n,1] (its first n rows; the rest is padding)."""
    return _SurfaceSdf.apply(sdf0, idx, sdf_sub)


def dmtet_verts(pos, sdf, verts0, vert_edge, grid):
    return _DMTetVerts.apply(pos, sdf, verts0, vert_edge, grid)


def dmtet(pos, sdf, grid):
    """(verts [V,3] differentiable w.r.t. sdf/pos, faces int64 [F,3], uv_idx int64 [F,3])."""
    verts0, faces, uv_idx, vert_edge = dmtet_extract(pos, sdf, grid)
    if torch.is_grad_enabled() and (pos.requires_grad or sdf.requires_grad):
        return _DMTetVerts.apply(pos, sdf, verts0, vert_edge, grid), faces, uv_idx
    return verts0, faces, uv_idx


# ---------------------------------------------------------------------------------------------- bone transforms
class _BoneTransforms(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bones, angles, chain):
        require_device(bones, angles, chain, what="bone_transforms")
        bones, angles = f32c(bones), f32c(angles)
        N, K = angles.shape[0], angles.shape[1]
        assert bones.shape[0] in (1, N) and bones.shape[1] == K and chain.shape[0] == K and chain.dtype == torch.int32
        M = torch.empty((N, K, 12), dtype=torch.float32, device=angles.device)
        call("a3d_bone_transforms_fwd", ptr(bones), bones.shape[0], ptr(angles), ptr(chain), N, K, chain.shape[1], ptr(M), stream())
        ctx.save_for_backward(bones, angles, chain)
        return M

    @staticmethod
    def backward(ctx, g_M):
        bones, angles, chain = ctx.saved_tensors
        N, K = angles.shape[0], angles.shape[1]
        g_angles = torch.empty_like(angles)
        call("a3d_bone_transforms_bwd", ptr(f32h(g_M)), ptr(bones), bones.shape[0], ptr(angles), ptr(chain), N, K, chain.shape[1], ptr(g_angles),
             stream())
        return None, g_angles, None


def bone_transforms(bones, angles, chain):
    """bones [1|N,K,2,3] (no grad), angles [N,K,3], chain int32 [K,D] (root -> leaf, -1 padded) -> M [N,K,12]."""
    return _BoneTransforms.apply(bones, angles, chain)


_linspace_cache = {}


def _linspace01(n):
    """torch.linspace(0, 1, n) as Python floats (float32 values, computed once per n on the CPU: what the reference's tensors hold)."""
    if n not in _linspace_cache:
        _linspace_cache[n] = [float(v) for v in torch.linspace(0.0, 1.0, n, dtype=torch.float32)]
    return _linspace_cache[n]


def estimate_bones_device(seq_shape, n_body_bones, n_leg_bones, body_mode_y_plus, y_threshold, attach):
    """a3d_estimate_bones: seq_shape [B,F,V,3] (float32, on the GPU) -> (bones [B,F,K,2,3], nearest int32 [2], ok int32 [1]) -- one launch,
    no host synchronisation.  ``attach``: the four legs' body joints (negative = found on the device, see include/a3d.h); ``y_threshold``
    None or Fauna's bone_y_threshold."""
    require_device(seq_shape, what="estimate_bones")
    pos = f32c(seq_shape)
    B, Fr, V = pos.shape[:3]
    K = n_body_bones + 4 * n_leg_bones
    dev = pos.device
    bones = torch.empty((B, Fr, K, 2, 3), dtype=torch.float32, device=dev)
    nearest = torch.empty(2, dtype=torch.int32, device=dev)
    ok = torch.empty(1, dtype=torch.int32, device=dev)
    work = torch.empty(3 * B * Fr * V, dtype=torch.float32, device=dev)
    a = _lib.EstimateBonesArgs(size=ctypes.sizeof(_lib.EstimateBonesArgs), N=B * Fr, pos=ptr(pos), bones=ptr(bones), nearest=ptr(nearest), ok=ptr(ok), V=V,
                               workspace=ptr(work),
                               n_body=n_body_bones, n_leg=n_leg_bones, body_mode_y_plus=int(bool(body_mode_y_plus)),
                               use_y_threshold=int(y_threshold is not None), y_threshold=float(y_threshold or 0.0))
    for i, v in enumerate(attach):
        a.attach[i] = int(v)
    for i, v in enumerate(_linspace01(math.ceil((n_body_bones + 1) / 2))):
        a.blend[i] = v
    for i, v in enumerate(_linspace01(n_leg_bones + 1)):
        a.ramp[i] = v
    call("a3d_estimate_bones", ctypes.addressof(a), stream())
    return bones, nearest, ok


def estimate_bones_device_ok(seq_shape, n_body_bones, n_leg_bones, n_legs):
    """Whether a3d_estimate_bones takes this call (else: the torch restatement)."""
    return (seq_shape.is_cuda and seq_shape.dtype == torch.float32 and seq_shape.dim() == 4 and seq_shape.shape[3] == 3
            and seq_shape.shape[0] * seq_shape.shape[1] <= 32 and seq_shape.shape[0] * seq_shape.shape[1] * seq_shape.shape[2] <= (1 << 22)
            and seq_shape.shape[2] > 0 and n_body_bones % 2 == 0 and 2 <= n_body_bones <= 32 and 0 <= n_leg_bones <= 8 and (n_leg_bones == 0 or n_legs == 4))


# ---------------------------------------------------------------------------------------------- skinning
class _Skin(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, bones, T, temperature):
        require_device(v, bones, T, what="skinning")
        v, bones, T = f32c(v), f32c(bones), f32c(T)
        B, K = T.shape[0], T.shape[1]
        V = v.shape[1]
        assert v.shape[0] in (1, B) and bones.shape[0] in (1, B) and bones.shape[1] == K and T.shape[2] == 12
        out = torch.empty((B, V, 3), dtype=torch.float32, device=v.device)
        # the backward accumulates g_T with atomics: allocated now and cleared by the forward launch when a backward can follow
        g_T = torch.empty_like(T) if ctx.needs_input_grad[2] else None
        call("a3d_skin_fwd", ptr(v), v.shape[0], ptr(bones), bones.shape[0], ptr(T), B, V, K, float(temperature), ptr(out), None, ptr(g_T), stream())
        ctx.save_for_backward(v, bones, T)
        ctx.g_T = g_T
        ctx.temperature = float(temperature)
        return out

    @staticmethod
    def backward(ctx, g_out):
        v, bones, T = ctx.saved_tensors
        B, K, V = T.shape[0], T.shape[1], v.shape[1]
        g_v = torch.empty((B, V, 3), dtype=torch.float32, device=v.device) if ctx.needs_input_grad[0] else None
        g_T, ctx.g_T = ctx.g_T, None  # the cleared buffer serves ONE backward (it is handed out as the gradient)
        clear = g_T is not None
        if g_T is None:
            g_T = torch.empty_like(T)
        call("a3d_skin_bwd", ptr(f32h(g_out)), ptr(v), v.shape[0], ptr(bones), bones.shape[0], ptr(T), B, V, K, ctx.temperature, ptr(g_v),
             ptr(g_T), int(clear), stream())
        if g_v is not None and v.shape[0] == 1 and B > 1:
            g_v = g_v.sum(0, keepdim=True)  # shared canonical mesh: per-image partials, reduced here (no 16-way atomic contention)
        return g_v, None, g_T, None


def skin(v, bones, T, temperature):
    """v [1|B,V,3], bones [1|B,K,2,3] (detached), T [B,K,12] -> [B,V,3]."""
    return _Skin.apply(v, bones, T, temperature)


class _SkinPose(torch.autograd.Function):
    """skinning() as one launch each way (csrc/skin.hip, POSE kernels): (posed vertices [B,V,3], transforms T [B,K,12])."""

    @staticmethod
    def forward(ctx, v, bones, angles, chain, temperature):
        require_device(v, bones, angles, chain, what="skin_pose")
        v, bones, angles = f32c(v), f32c(bones), f32c(angles)
        B, K, V, D = angles.shape[0], angles.shape[1], v.shape[1], chain.shape[1]
        assert v.shape[0] in (1, B) and bones.shape[0] in (1, B) and bones.shape[1] == K and chain.shape[0] == K and chain.dtype == torch.int32
        out = torch.empty((B, V, 3), dtype=torch.float32, device=v.device)
        T = torch.empty((B, K, 12), dtype=torch.float32, device=v.device)
        # the backward accumulates the angle gradients with atomics: its buffer is allocated now and cleared by the forward launch
        g_angles = torch.empty_like(angles) if (ctx.needs_input_grad[0] or ctx.needs_input_grad[2]) else None
        products = None if g_angles is None else torch.empty((B, _lib.lib().a3d_skin_pose_products_floats(K, D)), dtype=torch.float32, device=v.device)
        call("a3d_skin_pose_fwd", ptr(v), v.shape[0], ptr(bones), bones.shape[0], ptr(angles), ptr(chain), B, V, K, D, float(temperature),
             ptr(out), ptr(T), ptr(products), ptr(g_angles), stream())
        ctx.save_for_backward(v, bones, angles, chain, T, products)
        ctx.g_angles = g_angles
        ctx.temperature = float(temperature)
        ctx.set_materialize_grads(False)
        return out, T

    @staticmethod
    def backward(ctx, g_out, g_T_ext):
        v, bones, angles, chain, T, products = ctx.saved_tensors
        B, K, V, D = angles.shape[0], angles.shape[1], v.shape[1], chain.shape[1]
        if g_out is None:  # only the transforms were used downstream
            g_out = torch.zeros((B, V, 3), dtype=torch.float32, device=v.device)
        g_v = torch.empty((B, V, 3), dtype=torch.float32, device=v.device) if ctx.needs_input_grad[0] else None
        g_angles, ctx.g_angles = ctx.g_angles, None  # the cleared buffer serves ONE backward
        clear = g_angles is not None
        if g_angles is None:
            g_angles = torch.empty_like(angles)
        call("a3d_skin_pose_bwd", ptr(f32h(g_out)), ptr(v), v.shape[0], ptr(bones), bones.shape[0], ptr(T), ptr(products), ptr(angles), ptr(chain), B, V, K, D,
             ctx.temperature, ptr(g_v), ptr(None if g_T_ext is None else f32h(g_T_ext)), ptr(g_angles), int(clear), stream())
        if g_v is not None and v.shape[0] == 1 and B > 1:
            g_v = g_v.sum(0, keepdim=True)
        return g_v, None, g_angles, None, None


def skin_pose_supported(K, D):
    return K <= _lib.lib().a3d_skin_pose_max_bones() and D <= 8


def skin_pose(v, bones, angles, chain, temperature):
    """v [1|B,V,3], bones [1|B,K,2,3] (detached), angles [B,K,3], chain int32 [K,D] -> (posed [B,V,3], T [B,K,12])."""
    return _SkinPose.apply(v, bones, angles, chain, temperature)


def skin_weights(v, bones, B, temperature):
    """aux['vertices_to_bones'] on demand: [K, Bw, V]."""
    require_device(v, bones, what="skin_weights")
    v, bones = f32c(v), f32c(bones)
    K, V = bones.shape[1], v.shape[1]
    Bw = max(v.shape[0], bones.shape[0])
    T = torch.zeros((Bw, K, 12), dtype=torch.float32, device=v.device)
    out = torch.empty((Bw, V, 3), dtype=torch.float32, device=v.device)
    w = torch.empty((K, Bw, V), dtype=torch.float32, device=v.device)
    call("a3d_skin_fwd", ptr(v), v.shape[0], ptr(bones), bones.shape[0], ptr(T), Bw, V, K, float(temperature), ptr(out), ptr(w), None, stream())
    return w


# ---------------------------------------------------------------------------------------------- normals
class VertexFaceAdjacency:
    """Vertex -> incident (corner, face) entries of one triangle list: CSR, built by a3d_normals_adjacency / a3d_mesh_topology[_finalize],
    or -- ``lists`` -- fixed stride, as the DMTet emit launch writes them."""

    def __init__(self, tri32: torch.Tensor, num_vertices: int, build: bool = True, lists=None):
        require_device(tri32, what="normals_adjacency")
        F, V = tri32.shape[0], int(num_vertices)
        self.tri, self.num_vertices = tri32, V
        self.sorted = True  # lists stored in ascending key order (False for the lists a3d_mesh_topology_finalize / the DMTet emit leave)
        self.stride = 0  # 0: CSR (off[V+1]); S > 0: list v = adj[v*S .. v*S + off[v]) -- written by the DMTet emit launch (include/a3d.h: lists_stride)
        if lists is not None:  # (list lengths, slots, stride) as a3d_dmtet_emit wrote them
            self.off, self.adj, self.stride = lists
            self.sorted = False
            return
        self.off = torch.empty(V + 1, dtype=torch.int32, device=tri32.device)
        self.adj = torch.empty(max(3 * F, 1), dtype=torch.int32, device=tri32.device)
        if build:
            cursor = torch.empty(V, dtype=torch.int32, device=tri32.device)
            call("a3d_normals_adjacency", ptr(tri32), V, F, ptr(self.off), ptr(self.adj), ptr(cursor), stream())


_adj_cache = _IdentityCache()


def vertex_face_adjacency(tri32: torch.Tensor, num_vertices: int) -> VertexFaceAdjacency:
    adj = _adj_cache.peek(tri32)
    if adj is None:
        adj = mesh_topology(tri32, num_vertices)[0]
    if adj.num_vertices != num_vertices:  # same triangle list used with another vertex count: rebuild, do not trust the cache
        adj = VertexFaceAdjacency(tri32, num_vertices)
    return adj


NORMALS_FACES_FIRST = os.environ.get("A3D_NORMALS_FACES_FIRST", "1") != "0"  # a3d_normals_bwd: every face's adjoint once + a sum per vertex


def _normals_bwd_call(g, acc, v, tri32, adjacency, B, V, F):
    """a3d_normals_bwd for one vertex array -> g_v.  Faces first (a [B,F,9] scratch) unless switched off: same bits, ~half the gathers."""
    g_v = torch.empty_like(v)
    faces_first = NORMALS_FACES_FIRST and F > 0
    face_scratch = torch.empty((B, F, 9), dtype=torch.float32, device=v.device) if faces_first else None
    scratch = None if faces_first else torch.empty_like(v)
    call("a3d_normals_bwd", ptr(g), g.stride(1), ptr(acc), ptr(v), ptr(tri32), ptr(adjacency.off), ptr(adjacency.adj), B, V, F, ptr(scratch),
         ptr(g_v), adjacency.stride, ptr(face_scratch), stream(), tag=f"[B{B}]")
    return g_v


class _Normals(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, tri32, adjacency):
        require_device(v, tri32, what="vertex_normals")
        v = f32c(v)
        B, V, F = v.shape[0], v.shape[1], tri32.shape[0]
        acc = torch.empty_like(v)
        nrm = torch.empty_like(v)
        call("a3d_normals_fwd", ptr(v), ptr(tri32), ptr(adjacency.off), ptr(adjacency.adj), B, V, F, ptr(acc), ptr(nrm), adjacency.stride,
             stream(), tag=f"[B{B}]")
        ctx.save_for_backward(v, acc, tri32)
        ctx.adjacency = adjacency
        return nrm

    @staticmethod
    def backward(ctx, g_nrm):
        v, acc, tri32 = ctx.saved_tensors
        B, V, F = v.shape[0], v.shape[1], tri32.shape[0]
        if g_nrm.dtype != torch.float32 or g_nrm.stride(2) != 1 or g_nrm.stride(0) != V * g_nrm.stride(1):  # rows must be evenly strided
            g_nrm = f32c(g_nrm)
        return _normals_bwd_call(g_nrm, acc, v, tri32, ctx.adjacency, B, V, F), None, None


class _NormalsPair(torch.autograd.Function):
    """Vertex normals of two vertex arrays over one triangle list from ONE launch: (nrm_a [Ba,V,3], nrm_b [Bb,V,3])."""

    @staticmethod
    def forward(ctx, v_a, v_b, tri32, adjacency):
        require_device(v_a, v_b, tri32, what="vertex_normals_pair")
        v_a, v_b = f32c(v_a), f32c(v_b)
        V, F = v_a.shape[1], tri32.shape[0]
        assert v_b.shape[1] == V
        acc_a, nrm_a, acc_b, nrm_b = torch.empty_like(v_a), torch.empty_like(v_a), torch.empty_like(v_b), torch.empty_like(v_b)
        call("a3d_normals_fwd_pair", ptr(v_a), v_a.shape[0], ptr(v_b), v_b.shape[0], ptr(tri32), ptr(adjacency.off), ptr(adjacency.adj), V, F,
             ptr(acc_a), ptr(nrm_a), ptr(acc_b), ptr(nrm_b), adjacency.stride, stream(), tag=f"[B{v_a.shape[0]}+B{v_b.shape[0]}]")
        ctx.save_for_backward(v_a, acc_a, v_b, acc_b, tri32)
        ctx.adjacency = adjacency
        ctx.set_materialize_grads(False)
        return nrm_a, nrm_b

    @staticmethod
    def backward(ctx, g_a, g_b):
        v_a, acc_a, v_b, acc_b, tri32 = ctx.saved_tensors
        F = tri32.shape[0]
        out = []
        for v, acc, g in ((v_a, acc_a, g_a), (v_b, acc_b, g_b)):
            if g is None or v is None:
                out.append(None)
                continue
            B, V = v.shape[0], v.shape[1]
            if g.dtype != torch.float32 or g.stride(2) != 1 or g.stride(0) != V * g.stride(1):
                g = f32c(g)
            out.append(_normals_bwd_call(g, acc, v, tri32, ctx.adjacency, B, V, F))
        return out[0], out[1], None, None


class NormalsJob:
    """The forward pass of vertex_normals_pair(v_a, v_b, tri) as a side job of the rasteriser's triangle launch: hand it to
    rasterize(normals_job=...), which fills acc / nrm and sets ``done``; vertex_normals_attach then puts the results on the autograd
    graph.  ``v_b`` may be None."""

    def __init__(self, v_a, v_b, tri):
        self.tri32 = tri_int32(tri)
        self.v_a = f32c(v_a.detach())
        self.v_b = None if v_b is None else f32c(v_b.detach())
        assert self.v_b is None or self.v_b.shape[1] == self.v_a.shape[1]
        self.adjacency = vertex_face_adjacency(self.tri32, self.v_a.shape[1])
        self.acc_a = self.nrm_a = self.acc_b = self.nrm_b = None
        self.done = False


class _NormalsAttach(torch.autograd.Function):
    """(nrm_a, nrm_b) of a finished NormalsJob as functions of (v_a, v_b): forward hands out the job's buffers, backward is
    _NormalsPair's."""

    @staticmethod
    def forward(ctx, v_a, v_b, job):
        assert job.done
        ctx.save_for_backward(job.v_a, job.acc_a, job.v_b, job.acc_b, job.tri32)
        ctx.adjacency = job.adjacency
        ctx.set_materialize_grads(False)
        return job.nrm_a, job.nrm_b

    @staticmethod
    def backward(ctx, g_a, g_b):
        return _NormalsPair.backward(ctx, g_a, g_b)[:2] + (None,)


def vertex_normals_attach(v_a, v_b, job):
    """(normals of v_a, normals of v_b or None) computed by ``job`` inside a rasteriser launch, differentiable like vertex_normals_pair."""
    return _NormalsAttach.apply(v_a, v_b, job)


def vertex_normals_pair(v_a, v_b, tri):
    """(normals of v_a, normals of v_b): two meshes that share ``tri``, one launch."""
    tri32 = tri_int32(tri)
    return _NormalsPair.apply(v_a, v_b, tri32, vertex_face_adjacency(tri32, v_a.shape[1]))


def vertex_normals(v, tri):
    """Area-weighted, normalised vertex normals [B,V,3] (auto_normals); the adjacency is built once per triangle list."""
    tri32 = tri_int32(tri)
    return _Normals.apply(v, tri32, vertex_face_adjacency(tri32, v.shape[1]))


# ---------------------------------------------------------------------------------------------- clip-space transform
class _XfmPoints(torch.autograd.Function):
    """xfm_points(points, matrix, use_python=True) (renderutils/ops.py:515-531): [points, 1] . matrix^T as one launch each way
    (csrc/xfm.hip) instead of pad + bmm (+ two bmm and a slice backward)."""

    @staticmethod
    def forward(ctx, points, matrix, alias=False):
        require_device(points, matrix, what="xfm_points")
        points, matrix = f32c(points), f32c(matrix)
        Bp, V, Bm = points.shape[0], points.shape[1], matrix.shape[0]
        B = max(Bp, Bm)
        assert points.shape[2] == 3 and matrix.shape[1:] == (4, 4) and Bp in (1, B) and Bm in (1, B)
        out = torch.empty((B, V, 4), dtype=torch.float32, device=points.device)
        g_M = torch.empty_like(matrix) if ctx.needs_input_grad[1] else None  # (accumulated by the backward with atomics: cleared here)
        call("a3d_xfm_points_fwd", ptr(points), Bp, ptr(matrix), Bm, B, V, ptr(out), ptr(g_M), stream())
        ctx.save_for_backward(points, matrix)
        ctx.g_M = g_M
        ctx.set_materialize_grads(False)
        # ``alias``: the points once more as a second output (no copy) for their OTHER consumer in render_mesh (the G-buffer's position
        # attribute): both gradients then arrive at this node and are summed inside its one backward launch, instead of by an accumulation
        # kernel of the autograd engine in front of it.  alias = 2: ... and a third output for their third consumer, the vertex normals.
        # alias = 3: ... and the CLIP positions a second time, for the consumer beside the rasteriser / G-buffer: the antialiasing.
        if alias and Bp == B:
            return out, points.detach(), (points.detach() if int(alias) >= 2 else None), (out.detach() if int(alias) >= 3 else None)
        return out, None, None, None

    @staticmethod
    def backward(ctx, g, g_alias=None, g_alias2=None, g_second=None):
        points, matrix = ctx.saved_tensors
        Bp, V, Bm = points.shape[0], points.shape[1], matrix.shape[0]
        B = max(Bp, Bm)
        if g_alias is None and g_alias2 is not None:
            g_alias, g_alias2 = g_alias2, None
        if g is None and g_second is not None:
            g, g_second = g_second, None
        if g is None:
            if g_alias is None:
                return None, None, None
            g = torch.zeros((B, V, 4), dtype=torch.float32, device=points.device)
        # (the gradient may be the clip columns of the G-buffer's 16-float gradient rows: read in place, with its vertex stride)
        if g.dtype != torch.float32 or g.stride(2) != 1 or g.stride(1) < 4 or g.stride(0) != V * g.stride(1):
            g = f32c(g)
        if g_alias is not None and (g_alias.dtype != torch.float32 or g_alias.stride(2) != 1 or g_alias.stride(1) < 3 or g_alias.stride(0) != V * g_alias.stride(1)):
            g_alias = f32c(g_alias)
        if g_second is not None and (g_second.dtype != torch.float32 or g_second.stride(2) != 1 or g_second.stride(1) < 4 or g_second.stride(0) != V * g_second.stride(1)):
            g_second = f32c(g_second)
        if g_alias2 is not None and (g_alias2.dtype != torch.float32 or g_alias2.stride(2) != 1 or g_alias2.stride(1) < 3 or g_alias2.stride(0) != V * g_alias2.stride(1)):
            g_alias2 = f32c(g_alias2)
        g_p = torch.empty((B, V, 3), dtype=torch.float32, device=points.device) if ctx.needs_input_grad[0] else None
        g_M, ctx.g_M = ctx.g_M, None  # the cleared buffer serves ONE backward
        clear = g_M is not None
        if g_M is None and ctx.needs_input_grad[1]:
            g_M = torch.empty_like(matrix)
        if g_alias is not None and g_p is None:
            g_p = torch.empty((B, V, 3), dtype=torch.float32, device=points.device)
        call("a3d_xfm_points_bwd", ptr(g), g.stride(1), ptr(points), Bp, ptr(matrix), Bm, B, V, ptr(g_p), ptr(g_M), int(clear), ptr(g_alias),
             0 if g_alias is None else g_alias.stride(1), ptr(g_alias2), 0 if g_alias2 is None else g_alias2.stride(1), ptr(g_second),
             0 if g_second is None else g_second.stride(1), stream())
        if g_p is not None and Bp == 1 and B > 1:
            g_p = g_p.sum(0, keepdim=True)
        return g_p, g_M, None


def xfm_points(points, matrix, alias=False):
    """points [1|B,V,3], matrix [1|B,4,4] -> [B,V,4] homogeneous clip-space positions.  ``alias``: -> (clip, the points as a second output
    of the same node or None); alias = 2: -> (clip, second output, third output); alias = 3: -> (..., clip once more): see _XfmPoints."""
    out, again, third, out_again = _XfmPoints.apply(points, matrix, alias)
    if int(alias) >= 3:
        return out, again, third, out_again
    if int(alias) == 2:
        return out, again, third
    return (out, again) if alias else out


class _FlowDelta(torch.autograd.Function):
    """delta_xy of render_mesh (render.py:281-288) from the clip positions: one launch each way (csrc/xfm.hip)."""

    @staticmethod
    def forward(ctx, clip, num_frames):
        require_device(clip, what="flow_delta")
        clip = f32c(clip)
        N, V = clip.shape[0], clip.shape[1]
        delta = torch.empty((N, V, 2), dtype=torch.float32, device=clip.device)
        call("a3d_flow_delta_fwd", ptr(clip), N, int(num_frames), V, ptr(delta), stream())
        ctx.save_for_backward(clip)
        ctx.num_frames = int(num_frames)
        return delta

    @staticmethod
    def backward(ctx, g):
        (clip,) = ctx.saved_tensors
        N, V = clip.shape[0], clip.shape[1]
        g_clip = torch.empty_like(clip)
        # (the gradient may be two columns of the G-buffer backward's 16-float rows: read in place, with its vertex stride)
        if g.dtype != torch.float32 or g.stride(2) != 1 or g.stride(1) < 2 or g.stride(0) != V * g.stride(1):
            g = f32c(g)
        call("a3d_flow_delta_bwd", ptr(g), g.stride(1), ptr(clip), N, ctx.num_frames, V, ptr(g_clip), stream())
        return g_clip, None


def flow_delta(clip, num_frames):
    """clip [B*F,V,4] -> [B*F,V,2]: ndc of the next frame minus ndc of this one, zeros for a sequence's last frame."""
    return _FlowDelta.apply(clip, num_frames)


# ---------------------------------------------------------------------------------------------- covered pixels
_cover_counts = _IdentityCache(maxsize=2)  # raster buffer -> block counts of its covered-pixel list
_aa_prepared = _IdentityCache(maxsize=2)  # raster buffer -> (key of its clip tensor, screen positions, zeroed counters)


def _cover_counted(rast, tile):
    """(scratch, length of the list): block counts + group sums left by the rasteriser's resolve (same launch), otherwise a counting pass
    of our own; the ONE read-back: the group sums (a few KB; unused words are zero), added up on the host."""
    B, H, W = rast.shape[:3]
    ensure_resolved(rast)
    scratch = _cover_counts.peek(rast) if tile == 8 else None
    if scratch is None:
        scratch = torch.empty(_lib.lib().a3d_cover_scratch_bytes(B, H, W) // 4, dtype=torch.int32, device=rast.device)
        call("a3d_cover_count", ptr(rast), B, H, W, tile, ptr(scratch), stream())
    nb = _lib.lib().a3d_cover_blocks(B, H, W)
    tail = _lib.read_back(scratch[nb:nb + _lib.lib().a3d_cover_groups(B, H, W) * _lib.lib().a3d_cover_group_stride()])  # THE read-back: one sum per group of 64 blocks, one per 64-byte line; words 1, 2 of the first line: the binned rasteriser's status
    if tail.shape[0] > 2 and int(tail[1]) > 0:
        _rast_bins_grow((rast.device, B, H, W), int(tail[1]), int(tail[2]))
    return scratch, int(tail[::_lib.lib().a3d_cover_group_stride()].sum())


def covered_pixels(rast, tile=8, return_inverse=False):
    """int64 [P] flat indices of the pixels with rast[...,3] > 0, image-major, 8x8-tile order inside an image (``tile=8``; falls back
    to row-major when H or W is not a multiple of 8).  One 8-byte read-back (P) between the count and the emit launches.
    ``return_inverse``: also int32 [B*H*W], the list entry of every pixel (-1 = uncovered), written by the same launch."""
    require_device(rast, what="covered_pixels")
    rast = f32c(rast.detach())
    B, H, W = rast.shape[:3]
    if tile != 8 or H % 8 or W % 8:
        tile = 0
    dev = rast.device
    scratch, total = _cover_counted(rast, tile)
    pix = torch.empty(total, dtype=torch.int64, device=dev)
    inv = torch.empty(B * H * W, dtype=torch.int32, device=dev) if return_inverse else None
    if pix.shape[0] or return_inverse:
        call("a3d_cover_emit", ptr(rast), B, H, W, tile, ptr(scratch), ptr(pix), ptr(inv), stream())
    return (pix, inv) if return_inverse else pix


# ---------------------------------------------------------------------------------------------- rasterise
_rast_keys = {}
# (round 5) rasterize(defer_resolve=True): only the triangle launch runs; the raster buffer is WRITTEN by the covered_gbuffer call that
# follows (a3d_rast_resolve_gbuffer_fwd: resolve + covered-pixel list + G-buffer rows in one launch, the list offsets by decoupled
# look-back) -- or, for any other reader, by ensure_resolved (the stand-alone resolve).  The list's length is not known when that launch
# is enqueued: its rows are allocated for the previous frame's length + 25 % (per device and frame size) and the exact two-launch path
# re-runs when the frame outgrew them.
DEFER_RESOLVE = os.environ.get("A3D_DEFER_RESOLVE", "1") != "0"


class _PendingResolves(_IdentityCache):
    """raster buffer -> what its resolve needs (clip, triangle list, key buffer).  An entry that is pushed out before anybody resolved
    it is RESOLVED on the way out (stand-alone launch): a raster buffer handed to a reader must never be left as uninitialised texels
    because a third deferred call came before its consumer did."""

    def put(self, t, val):
        self.data[self.key(t)] = (t, val)
        self.data.move_to_end(self.key(t))
        while len(self.data) > self.maxsize:
            _, (rast_old, pend) = self.data.popitem(last=False)
            _run_standalone_resolve(rast_old, pend)
            resolve_events["evicted"] = resolve_events.get("evicted", 0) + 1
        return val


def _run_standalone_resolve(rast, pend):
    B, H, W = rast.shape[:3]
    clip, tri32 = pend["clip"], pend["tri32"]
    call("a3d_rast_resolve", ptr(clip), clip.shape[0], ptr(tri32), B, clip.shape[1], tri32.shape[0], H, W, ptr(pend["rast"]), ptr(pend["keys"]),
         ptr(pend["cover"]), stream())
    _rast_keys[pend["key"]] = pend["keys"]  # (re-armed by the resolve)
    resolve_events["standalone"] += 1


_pending_resolve = _PendingResolves(maxsize=2)
_cover_last_len = {}  # (device, B, H, W) -> length of the last covered-pixel list
resolve_events = dict(fused=0, outgrown=0, standalone=0)
_debug_force_lookback_timeout = False  # tests: take the recovery path of a timed-out look-back


_dispatch_probe = {}  # device -> dict(ok, workgroups, stride, ms): the look-back's dispatch-order property as probed on that device


def dispatch_order_ok(device, n_workgroups=16384, stride=64):
    """Probe (once per device and process) that work-groups are dispatched in the order of their linear index -- what the fused resolve's
    look-back rests on (a3d_dispatch_order_probe) -- and remember the answer; rasterize defers its resolve only where it holds."""
    key = torch.device(device)
    hit = _dispatch_probe.get(key)
    if hit is None:
        import time

        with torch.cuda.device(key):
            scratch = torch.empty(n_workgroups + 1, dtype=torch.int32, device=key)
            torch.cuda.synchronize(key)
            t0 = time.perf_counter()
            call("a3d_dispatch_order_probe", n_workgroups, stride, ptr(scratch), stream())
            status = int(scratch[n_workgroups].item())
            ms = (time.perf_counter() - t0) * 1e3
        hit = _dispatch_probe[key] = dict(ok=status == 2, status=status, workgroups=n_workgroups, stride=stride, ms=round(ms, 3))
        if not hit["ok"]:
            import warnings

            warnings.warn(f"a3d_dispatch_order_probe on {key}: status {status} (2 = in order): the rasteriser's resolve will not be deferred on this device")
    return hit["ok"]


def ensure_resolved(rast):
    """Run the stand-alone resolve launch now if ``rast`` came out of rasterize(defer_resolve=True) and nothing has resolved it yet."""
    pend = _pending_resolve.take(rast)
    if pend is not None:
        _run_standalone_resolve(rast, pend)


def drop_pending_resolve(rast):
    """A deferred raster buffer whose consumer failed before it ran: resolve it now (keys re-armed, nothing left pinned)."""
    ensure_resolved(rast)

# the binned path (a3d_rast_opts.bins: per-tile triangle lists + a fine pass, no memory-side atomics): its scratch is kept per (device,
# stream, frame) like the key buffer (the fine pass leaves the tile counts at zero), the capacity of a tile list per (device, frame):
# 1024 entries to start with, 4 x the largest count ever reported above half the capacity (the covered-pixel read-back carries the
# report).  An overflowing tile is still rasterised exactly -- by the slow route inside the fine pass -- so the capacity is a matter of
# speed only; memory is not a constraint (1024 entries: 67 MB at B = 16, 256 x 256).
# OFF by default (round 5, measured inside the bench step on one box): bin launch 18.3 us + fine pass 36-60 us against 23.6 + 9.0 us for
# the triangle-parallel atomics + resolve.  Same fragment tests in total, but dealt out by SCREEN position: under a head-on camera
# nearly all of the mesh's 12k triangles fall into a dozen 256-pixel blocks (1400 list entries each, six chunks of dependent gathers
# in ONE work-group) and the launch ends when that work-group does; the atomic form deals the same work out by triangle and does not
# care where on screen it lands ("Rasteriser: the binned form").
RASTER_BINNED = os.environ.get("A3D_RASTER_BINNED", "0") == "1"
RASTER_BIN_CAP0 = 1024
_rast_bins = {}
_rast_bin_caps = {}
rast_bin_events = dict(grown=0, overflowed_blocks=0)


def _rast_bins_grow(key, max_count, overflowed):
    cap = _rast_bin_caps.get(key, RASTER_BIN_CAP0)
    want = 1 << max(4, int(4 * max_count - 1).bit_length())
    rast_bin_events["overflowed_blocks"] += overflowed
    if want > cap and want <= (1 << 16):
        _rast_bin_caps[key] = want
        rast_bin_events["grown"] += 1


class _Rasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, clip, tri32, B, H, W, prev, job, defer=False):
        require_device(clip, tri32, what="rasterize")
        clip = f32c(clip)
        V, F = clip.shape[1], tri32.shape[0]
        rast = torch.empty((B, H, W, 4), dtype=torch.float32, device=clip.device)
        # the (depth, id) key buffer is kept per (device, stream, size): the resolve leaves it armed, so only its first use pays the clear
        key = (clip.device, stream(), B, H, W)
        if prev is not None:
            prev = f32c(prev.detach())
            assert prev.shape == (B, H, W, 4)
        # the covered-pixel list's block counts ride along with the resolve when the list's tile order applies (covered_pixels picks them up)
        cover = None
        if H % 8 == 0 and W % 8 == 0 and (H * W) % 256 == 0 and F > 0:
            cover = torch.empty(_lib.lib().a3d_cover_scratch_bytes(B, H, W) // 4, dtype=torch.int32, device=clip.device)
        # binned path (per-tile triangle lists, no key buffer) when the frame qualifies; otherwise the atomic path and its key buffer
        bins = None
        if RASTER_BINNED and cover is not None and prev is None:
            cap = _rast_bin_caps.get((clip.device, B, H, W), RASTER_BIN_CAP0)
            bins = _rast_bins.pop(key, None)
            if bins is None or bins[1] != cap:
                nbytes = _lib.lib().a3d_rast_bins_bytes(B, H, W, cap)
                bins = (torch.empty(nbytes, dtype=torch.uint8, device=clip.device), cap, False) if nbytes else None
        scratch, clean = None, False
        if bins is None:
            scratch = _rast_keys.pop(key, None)
            clean = scratch is not None
            if scratch is None:
                scratch = torch.empty(_lib.lib().a3d_rast_scratch_bytes(B, H, W), dtype=torch.uint8, device=clip.device)
        # ... and so does what the silhouette analysis of this frame needs first (pixel-space vertex positions, zeroed append counters):
        # AAAnalysis picks them up when it is built for this raster buffer and this clip tensor
        aa_screen = aa_count = None
        if F > 0 and prev is None:
            aa_screen = torch.empty((clip.shape[0], V, 2), dtype=torch.float32, device=clip.device)
            aa_count = torch.empty((_lib.lib().a3d_aa_shards(),), dtype=torch.int32, device=clip.device)
        # ... and the opposite-vertex table, when this triangle list came with vertex -> face lists only (DMTet extraction): filled by
        # further extra work-groups of the same launch, picked up by the silhouette analysis through the topology cache
        topo = _topo_cache.peek(tri32) if F > 0 else None
        lists = getattr(topo, "lists", None) if (topo is not None and topo.opp is None) else None
        opp = torch.empty((F, 3), dtype=torch.int32, device=clip.device) if lists is not None else None
        # ... and a pending vertex-normals pass over the same triangle list (NormalsJob: the mesh being rasterised + the canonical one)
        if job is not None and not (F > 0 and job.tri32.data_ptr() == tri32.data_ptr() and job.v_a.shape[1] == V and job.v_a.device == clip.device):
            job = None  # (stays not done: the caller computes the normals in a launch of their own)
        opts = _lib.RastOpts(size=ctypes.sizeof(_lib.RastOpts), prev_rast=ptr(prev), cover_scratch=ptr(cover), aa_screen=ptr(aa_screen),
                             aa_count=ptr(aa_count), topo_off=ptr(None if lists is None else lists.off), topo_adj=ptr(None if lists is None else lists.adj),
                             topo_opp=ptr(opp))
        stride = lists.stride if lists is not None else (job.adjacency.stride if job is not None else 0)
        if job is not None and lists is not None and lists is not job.adjacency and lists.stride != job.adjacency.stride:
            job = None  # (one layout per launch)
        if job is not None:
            job.acc_a, job.nrm_a = torch.empty_like(job.v_a), torch.empty_like(job.v_a)
            if job.v_b is not None:
                job.acc_b, job.nrm_b = torch.empty_like(job.v_b), torch.empty_like(job.v_b)
            opts.normals_v_a, opts.normals_B_a, opts.normals_v_b = ptr(job.v_a), job.v_a.shape[0], ptr(job.v_b)
            opts.normals_B_b = 0 if job.v_b is None else job.v_b.shape[0]
            opts.normals_off, opts.normals_adj = ptr(job.adjacency.off), ptr(job.adjacency.adj)
            opts.normals_acc_a, opts.normals_a, opts.normals_acc_b, opts.normals_b = ptr(job.acc_a), ptr(job.nrm_a), ptr(job.acc_b), ptr(job.nrm_b)
        opts.lists_stride = stride
        if bins is not None:
            opts.bins, opts.bin_cap, opts.bins_clean = ptr(bins[0]), bins[1], int(bins[2])
        defer = bool(defer) and DEFER_RESOLVE and cover is not None and bins is None and prev is None and F > 0
        defer = defer and not torch.cuda.is_current_stream_capturing() and dispatch_order_ok(clip.device)  # (probed once per device)
        opts.defer_resolve = int(defer)
        call("a3d_rast_fwd", ptr(clip), clip.shape[0], ptr(tri32), B, V, F, H, W, ptr(rast), ptr(scratch), int(clean), ctypes.addressof(opts), stream(),
             tag=("" if job is None else f"[N{opts.normals_B_a}+{opts.normals_B_b}]") + ("[defer]" if defer else ""))
        if bins is not None:
            if len(_rast_bins) >= 4:
                _rast_bins.clear()
            _rast_bins[key] = (bins[0], bins[1], bool(bins[2]) or F > 0)  # (after a successful call its fine pass left every tile count at zero; F == 0 launches nothing)
        if job is not None:
            job.done = True
        if opp is not None:
            topo.opp = opp
        if cover is not None:
            _cover_counts.put(rast.detach(), cover)  # keyed on a detached alias: the entry must not pin this iteration's autograd graph
        if aa_screen is not None:
            # (the clip tensor rides in the entry so that its address cannot be recycled while the entry can still match)
            _aa_prepared.put(rast.detach(), (_IdentityCache.key(clip), aa_screen, aa_count, clip.detach()))
        if len(_rast_keys) >= 4:
            _rast_keys.clear()
        if defer:  # the keys are full until a resolve has consumed them: they travel with the raster buffer, not back into the cache
            _pending_resolve.put(rast.detach(), dict(clip=clip, tri32=tri32, keys=scratch, key=key, rast=rast, cover=cover))
        elif scratch is not None and (F > 0 or clean):  # only after a successful call whose resolve re-armed the keys (F == 0 returns before touching
            _rast_keys[key] = scratch  # them: a fresh torch.empty buffer must not come back as "clean"; a failed call leaves the buffer out too)
        ctx.save_for_backward(clip, tri32, rast)
        return rast

    @staticmethod
    def backward(ctx, g_rast):
        clip, tri32, rast = ctx.saved_tensors
        ensure_resolved(rast)
        B, H, W = rast.shape[:3]
        g_clip = torch.empty_like(clip)
        call("a3d_rast_bwd", ptr(f32h(g_rast)), ptr(rast), ptr(clip), clip.shape[0], ptr(tri32), B, clip.shape[1], tri32.shape[0], H, W,
             ptr(g_clip), stream())
        return g_clip, None, None, None, None, None, None, None


def rasterize(clip, tri, resolution, batch=None, prev=None, normals_job=None, defer_resolve=False):
    """clip [B|1,V,4] -> rast [B,H,W,4] = (u, v, z/w, triangle_id+1); differentiable through (u,v).  ``prev`` = the previous
    depth layer (DepthPeeler.rasterize_next_layer for layer n > 0): the nearest surface strictly behind it is returned.
    ``normals_job``: a NormalsJob over the same triangle list, run as extra work-groups of the triangle launch.
    ``defer_resolve``: the caller promises that covered_gbuffer(..., rast, ...) comes next (render_mesh's fused path): the texels are then
    written by THAT launch (see DEFER_RESOLVE); any other reader must call ensure_resolved(rast) first (covered_pixels does)."""
    if clip.dim() == 2:
        clip = clip[None]
    B = clip.shape[0] if batch is None else batch
    return _Rasterize.apply(clip, tri_int32(tri), B, int(resolution[0]), int(resolution[1]), prev, normals_job, defer_resolve)


def rasterize_db(clip, tri, rast):
    """rast_db [B,H,W,4] = (du/dX, du/dY, dv/dX, dv/dY) per pixel: the image-space derivatives of the perspective-correct
    barycentrics of the stored triangle, from the definition u = a0/s, a_i = q_j x q_k, q_i = p_i.xy - f p_i.w (torch ops, so they are
    differentiable w.r.t. clip like nvdiffrast's; nothing on the training path consumes them: render.py:24 passes rast_db=None)."""
    B, H, W, _ = rast.shape
    ensure_resolved(rast)
    ids = rast[..., 3].long() - 1
    hit = ids >= 0
    t = tri.long()[ids.clamp(min=0)]
    P = clip.expand(B, -1, -1)[torch.arange(B, device=rast.device)[:, None, None, None].expand_as(t), t]  # [B,H,W,3,4]
    fx = ((torch.arange(W, dtype=torch.float32, device=rast.device) + 0.5) * (2.0 / W) - 1.0)[None, None, :, None]
    fy = ((torch.arange(H, dtype=torch.float32, device=rast.device) + 0.5) * (2.0 / H) - 1.0)[None, :, None, None]
    x, y, w = P[..., 0], P[..., 1], P[..., 3]
    qx, qy = x - fx * w, y - fy * w
    a = [qx[..., 1] * qy[..., 2] - qy[..., 1] * qx[..., 2], qx[..., 2] * qy[..., 0] - qy[..., 2] * qx[..., 0],
         qx[..., 0] * qy[..., 1] - qy[..., 0] * qx[..., 1]]
    # d a_i / d fx = -w_j qy_k + qy_j w_k ;  d a_i / d fy = -qx_j w_k + w_j qx_k   (j, k) = (i+1, i+2)
    dax = [-w[..., (i + 1) % 3] * qy[..., (i + 2) % 3] + qy[..., (i + 1) % 3] * w[..., (i + 2) % 3] for i in range(3)]
    day = [-qx[..., (i + 1) % 3] * w[..., (i + 2) % 3] + w[..., (i + 1) % 3] * qx[..., (i + 2) % 3] for i in range(3)]
    s_ = a[0] + a[1] + a[2]
    s_ = torch.where(hit, s_, torch.ones_like(s_))
    sx, sy = dax[0] + dax[1] + dax[2], day[0] + day[1] + day[2]
    d = lambda ai, dai, ds, scale: (dai * s_ - ai * ds) / (s_ * s_) * scale
    db = torch.stack([d(a[0], dax[0], sx, 2.0 / W), d(a[0], day[0], sy, 2.0 / H), d(a[1], dax[1], sx, 2.0 / W), d(a[1], day[1], sy, 2.0 / H)], -1)
    return torch.where(hit[..., None], db, torch.zeros_like(db))


# ---------------------------------------------------------------------------------------------- interpolate
class _Interpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, attr, rast, tri32):
        require_device(attr, rast, tri32, what="interpolate")
        ensure_resolved(rast)
        attr, rast = f32c(attr), f32c(rast)
        B, H, W = rast.shape[:3]
        V, C = attr.shape[1], attr.shape[2]
        out = torch.empty((B, H, W, C), dtype=torch.float32, device=rast.device)
        call("a3d_interp_fwd", ptr(attr), attr.shape[0], C, ptr(rast), ptr(tri32), B, V, tri32.shape[0], H, W, ptr(out), stream(),
             tag=f"[C{C}]")
        ctx.save_for_backward(attr, rast, tri32)
        return out

    @staticmethod
    def backward(ctx, g_out):
        attr, rast, tri32 = ctx.saved_tensors
        B, H, W = rast.shape[:3]
        V, C = attr.shape[1], attr.shape[2]
        g_attr = torch.empty_like(attr) if ctx.needs_input_grad[0] else None
        g_rast = torch.empty_like(rast)
        call("a3d_interp_bwd", ptr(f32h(g_out)), ptr(attr), attr.shape[0], C, ptr(rast), ptr(tri32), B, V, tri32.shape[0], H, W, ptr(g_attr),
             ptr(g_rast), stream(), tag=f"[C{C}]")
        return g_attr, (g_rast if ctx.needs_input_grad[1] else None), None


def interpolate(attr, rast, tri):
    """attr [B|1,V,C] (or [V,C]) -> [B,H,W,C]."""
    if attr.dim() == 2:
        attr = attr[None]
    return _Interpolate.apply(attr, rast, tri_int32(tri))


def interpolate_da(attr, rast, tri, rast_db, diff_attrs="all"):
    """out_da [B,H,W,2*S] = (dA/dX, dA/dY) per selected attribute (dr.interpolate's second output with ``rast_db`` / ``diff_attrs``):
    dA/dX = du/dX (A0 - A2) + dv/dX (A1 - A2).  torch ops (differentiable); nothing on the training path consumes it (render.py:24)."""
    a = attr if attr.dim() == 3 else attr[None]
    sel = list(range(a.shape[-1])) if isinstance(diff_attrs, str) and diff_attrs == "all" else list(diff_attrs)
    B = rast.shape[0]
    ids = rast[..., 3].long() - 1
    hit = (ids >= 0)[..., None]
    t = tri.long()[ids.clamp(min=0)]
    A = a.expand(B, -1, -1)[torch.arange(B, device=rast.device)[:, None, None, None].expand_as(t), t][..., sel]  # [B,H,W,3,S]
    d0, d1 = A[..., 0, :] - A[..., 2, :], A[..., 1, :] - A[..., 2, :]
    dx = rast_db[..., 0:1] * d0 + rast_db[..., 2:3] * d1
    dy = rast_db[..., 1:2] * d0 + rast_db[..., 3:4] * d1
    da = torch.stack([dx, dy], -1).reshape(*dx.shape[:-1], -1)
    return torch.where(hit, da, torch.zeros_like(da))


# ---------------------------------------------------------------------------------------------- fused G-buffer
GBUFFER_GRAD_COLS = 16  # A3D_GBUFFER_GRAD_COLS of include/a3d.h


def _round_up(n, m):
    return -(-int(n) // int(m)) * int(m) if m else int(n)


class _GBuffer(torch.autograd.Function):
    """-> (rows [P,12], extra rows [P,E] | None, tex_in [Pp,3] | None, img [Pp] | None, pix [P], inv [B*H*W] | None).

    ``bucket`` > 0 (round 6): the launch also writes what the texture / feature fields take from the G-buffer in the form they take it
    (a3d_gb_aux): ``tex_in`` = the canonical positions as dense rows, padded with zero rows to a multiple of ``bucket``, and ``img`` =
    the point -> image index, padded with the last image.  The fields' input gradient then comes back as dense rows too (``g_tex`` of
    the backward): no column slice of the 12-wide rows, no pad, no padded gradient of either -- no torch kernel between this path and
    model/networks in either direction."""

    @staticmethod
    def forward(ctx, clip, v_pos, v_nrm, prior, rast, tri32, pix, extra, bucket=0):
        require_device(clip, v_pos, v_nrm, prior, rast, tri32, pix, extra, what="gbuffer")
        clip, v_pos, v_nrm, prior, rast = f32c(clip), f32c(v_pos), f32c(v_nrm), f32c(prior), f32c(rast)
        B, H, W = rast.shape[:3]
        V = v_pos.shape[1]
        assert clip.shape[:2] == (B, V) and v_pos.shape[0] == B and v_nrm.shape == v_pos.shape and prior.shape[0] in (1, B)
        listed = pix is None  # no list given: build it in the same launch (a3d_cover_gbuffer_fwd) and return it with the rows
        inv = cover_scratch = None
        dev = rast.device
        E = 0
        if extra is not None:
            extra = f32c(extra)
            E = extra.shape[2]
            assert extra.shape[:2] == (B, V) and 1 <= E <= 3
        # when a backward will follow, its gradient rows (one 64-byte row per (image, vertex), see backward) are allocated now and
        # cleared by the forward launch: one memset less on the backward path
        needs_grad = any(ctx.needs_input_grad)  # (forward runs with grad mode off: this is what says whether a backward can follow)
        rows = torch.empty((B, V, GBUFFER_GRAD_COLS), dtype=torch.float32, device=dev) if needs_grad else None
        out = extra_out = tex_in = img = None
        want_aux = ctx_wants_aux(bucket)  # (-1 / None: the rows alone)
        bucket = int(bucket) if want_aux else 0

        def aux_for(n_rows):
            """(struct or None, tex buffer, img buffer) with ``n_rows`` rows behind them."""
            if not want_aux:
                return None, None, None
            t = torch.empty((n_rows, 3), dtype=torch.float32, device=dev)
            im = torch.empty((n_rows,), dtype=torch.int64, device=dev)
            return _lib.GbAux(size=ctypes.sizeof(_lib.GbAux), tex_out=ptr(t), img_out=ptr(im), rows=n_rows, pad_to=bucket), t, im

        cap_key = (dev, B, H, W)
        pend = _pending_resolve.peek(rast) if listed else None
        if pend is not None and cap_key in _cover_last_len:
            # the raster buffer is still keys (rasterize(defer_resolve=True)): resolve, list and rows from ONE launch, the rows allocated for
            # the previous frame's list length + 25 %; the read-back of the true length comes after the launch is enqueued
            _pending_resolve.take(rast)
            cap = max(1024, -(-int(1.25 * _cover_last_len[cap_key] + 1) // 1024) * 1024)
            clip_r, tri_r = pend["clip"], pend["tri32"]
            cover_scratch = _cover_counts.peek(rast)
            pix_c = torch.empty(cap, dtype=torch.int64, device=dev)
            inv = torch.empty(B * H * W, dtype=torch.int32, device=dev)
            out_c = torch.empty((cap, 12), dtype=torch.float32, device=dev)
            extra_c = torch.empty((cap, E), dtype=torch.float32, device=dev) if extra is not None else None
            aux, tex_c, img_c = aux_for(_round_up(cap, bucket))
            call("a3d_rast_resolve_gbuffer_fwd", ptr(clip_r), clip_r.shape[0], ptr(tri_r), B, V, tri_r.shape[0], H, W, ptr(rast), ptr(pend["keys"]),
                 ptr(cover_scratch), cap, ptr(pix_c), ptr(inv), ptr(v_pos), ptr(v_nrm), ptr(prior), prior.shape[0], ptr(out_c), ptr(extra), E,
                 ptr(extra_c), ptr(rows), None if aux is None else ctypes.addressof(aux), stream())
            _rast_keys[pend["key"]] = pend["keys"]  # (every key the launch consumed is re-armed)
            nb = _lib.lib().a3d_cover_blocks(B, H, W)
            tail = _lib.read_back(cover_scratch[nb:nb + _lib.lib().a3d_cover_groups(B, H, W) * _lib.lib().a3d_cover_group_stride()])
            P = int(tail[::_lib.lib().a3d_cover_group_stride()].sum())
            resolve_events["fused"] += 1
            if int(tail[3]) != 0 or _debug_force_lookback_timeout:
                # a look-back ran out of its spin budget (never observed; it would mean work-groups were not dispatched in the order of their
                # index): offsets and sums of this launch are not to be trusted -- but the TEXELS are complete (they do not depend on the
                # look-up).  Count from the texels, take the two-launch path's second half, and stop deferring in this process.
                import warnings

                globals()["DEFER_RESOLVE"] = False
                resolve_events["timeouts"] = resolve_events.get("timeouts", 0) + 1
                warnings.warn("a3d_rast_resolve_gbuffer_fwd: look-back timed out; recovered through a3d_cover_count + a3d_cover_gbuffer_fwd, "
                              "deferred resolve switched off for this process")
                cover_scratch = torch.empty(_lib.lib().a3d_cover_scratch_bytes(B, H, W) // 4, dtype=torch.int32, device=dev)
                call("a3d_cover_count", ptr(rast), B, H, W, 8, ptr(cover_scratch), stream())
                _cover_counts.put(rast, cover_scratch)
                tail = _lib.read_back(cover_scratch[nb:nb + _lib.lib().a3d_cover_groups(B, H, W) * _lib.lib().a3d_cover_group_stride()])
                P, cap = int(tail[::_lib.lib().a3d_cover_group_stride()].sum()), -1
            _cover_last_len[cap_key] = P
            if P <= cap:
                pix, out, extra_out = pix_c[:P], out_c[:P], (extra_c[:P] if extra_c is not None else None)
                if want_aux:
                    tex_in, img = tex_c[:_round_up(P, bucket)], img_c[:_round_up(P, bucket)]
            else:  # outgrown: texels, block counts and sums are complete -- the exact list + rows through the two-launch path's second half
                resolve_events["outgrown"] += 1
                pix = torch.empty(P, dtype=torch.int64, device=dev)
        elif listed:
            cover_scratch, P = _cover_counted(rast, 8)  # (resolves a deferred buffer first)
            _cover_last_len[cap_key] = P
            pix = torch.empty(P, dtype=torch.int64, device=dev)
            inv = torch.empty(B * H * W, dtype=torch.int32, device=dev)
        P = pix.shape[0]
        assert pix.dtype == torch.int64 and pix.is_contiguous()
        fused_done = out is not None
        if out is None:
            out = torch.empty((P, 12), dtype=torch.float32, device=dev)
            if extra is not None:
                extra_out = torch.empty((P, E), dtype=torch.float32, device=dev)
        if fused_done:
            pass
        elif listed:
            aux, tex_in, img = aux_for(_round_up(P, bucket))
            call("a3d_cover_gbuffer_fwd", ptr(rast), ptr(tri32), B, V, tri32.shape[0], H, W, ptr(cover_scratch), P, ptr(pix), ptr(inv), ptr(v_pos),
                 ptr(v_nrm), ptr(prior), prior.shape[0], ptr(out), ptr(extra), E, ptr(extra_out), ptr(rows),
                 None if aux is None else ctypes.addressof(aux), stream())
        else:
            aux, tex_in, img = aux_for(_round_up(P, bucket))
            call("a3d_gbuffer_fwd", ptr(rast), ptr(tri32), ptr(pix), P, ptr(v_pos), ptr(v_nrm), ptr(prior), prior.shape[0], B, V, tri32.shape[0], H, W,
                 ptr(out), ptr(extra), E, ptr(extra_out), ptr(rows), None if aux is None else ctypes.addressof(aux), stream())
        ctx.save_for_backward(clip, v_pos, v_nrm, prior, rast, tri32, pix, extra)
        ctx.rows = rows
        ctx.set_materialize_grads(False)
        nondiff = [t for t in (img, pix if listed else None, inv) if t is not None]
        if nondiff:
            ctx.mark_non_differentiable(*nondiff)
        return out, extra_out, tex_in, img, (pix if listed else None), inv

    @staticmethod
    def backward(ctx, g_out, g_extra_out=None, g_tex=None, *_):
        clip, v_pos, v_nrm, prior, rast, tri32, pix, extra = ctx.saved_tensors
        B, H, W = rast.shape[:3]
        V, P = v_pos.shape[1], pix.shape[0]
        want_prior, want_clip = ctx.needs_input_grad[3], ctx.needs_input_grad[0]
        E = 0 if extra is None else extra.shape[2]
        if extra is not None and g_extra_out is None:
            g_extra_out = torch.zeros((P, E), dtype=torch.float32, device=rast.device)
        if g_out is None:  # (only the fields' input had a gradient)
            g_out = torch.zeros((P, 12), dtype=torch.float32, device=rast.device)
        # one 64-byte gradient row per (image, vertex): the kernel's atomics of a vertex are then one line request; the gradients are
        # strided views of the rows (a3d.h: v_pos 0..2 | v_nrm 3..5 | canonical 6..8 | extra 9..11 | clip 12..15)
        rows, ctx.rows = ctx.rows, None  # the rows the forward cleared serve ONE backward (their views are handed out as gradients)
        clear = rows is not None
        if rows is None:
            rows = torch.empty((B, V, GBUFFER_GRAD_COLS), dtype=torch.float32, device=rast.device)
        call("a3d_gbuffer_bwd", ptr(f32h(g_out)), ptr(rast), ptr(tri32), ptr(pix), P, ptr(v_pos), ptr(v_nrm), ptr(prior), prior.shape[0],
             ptr(clip) if want_clip else None, B, V, tri32.shape[0], H, W, ptr(rows), int(clear), int(want_prior), ptr(extra), E,
             None if extra is None else ptr(f32h(g_extra_out)), None if g_tex is None else ptr(f32h(g_tex)), stream())
        g_vpos, g_vnrm = rows[..., 0:3], rows[..., 3:6]
        g_clip = rows[..., 12:16] if want_clip else None
        g_prior = None
        if want_prior:
            if prior.shape[0] == 1:  # shared canonical mesh: the per-image partials summed by one small launch
                g_prior = torch.empty((1, V, 3), dtype=torch.float32, device=rast.device)
                call("a3d_gbuffer_prior_grad", ptr(rows), B, V, ptr(g_prior), stream())
            else:
                g_prior = rows[..., 6:9]
        g_extra = rows[..., 9:9 + E] if (extra is not None and ctx.needs_input_grad[7]) else None
        return g_clip, g_vpos, g_vnrm, g_prior, None, None, None, g_extra, None


def ctx_wants_aux(bucket):
    return bucket is not None and int(bucket) >= 0


def covered_gbuffer(clip, v_pos, v_nrm, prior_v_pos, rast, tri, extra=None, field_inputs=None):
    """covered_pixels(rast, return_inverse=True) + gbuffer(...) as ONE launch: (gb [P,12][, extra [P,E]], pix [P], inv [B*H*W]).
    H and W must be multiples of 8 (the tile-ordered list).
    ``field_inputs`` = bucket size (0 = no padding): also (tex_in [Pp,3], img [Pp]) -- see _GBuffer -- appended to the result."""
    assert rast.shape[1] % 8 == 0 and rast.shape[2] % 8 == 0
    gb, ex, tex_in, img, pix, inv = _GBuffer.apply(clip, v_pos, v_nrm, prior_v_pos, f32c(rast.detach()), tri_int32(tri), None, extra,
                                                   -1 if field_inputs is None else int(field_inputs))
    res = (gb, pix, inv) if extra is None else (gb, ex, pix, inv)
    return res if field_inputs is None else res + (tex_in, img)


def gbuffer(clip, v_pos, v_nrm, prior_v_pos, rast, tri, pix, extra=None):
    """[P,12] = world position | face normal | smooth normal | canonical position at the covered pixels ``pix``; with ``extra`` [B,V,E<=3]
    also that attribute interpolated to [P,E] (returns a pair).

    Differentiable w.r.t. v_pos, v_nrm, prior_v_pos, extra and -- through the barycentrics -- clip (x, y, w); pass ``rast.detach()``
    semantics are implied: the gradient to ``clip`` is produced here, not through ``rast``.
    """
    gb, ex = _GBuffer.apply(clip, v_pos, v_nrm, prior_v_pos, rast.detach(), tri_int32(tri), pix, extra, -1)[:2]
    return gb if extra is None else (gb, ex)


# ---------------------------------------------------------------------------------------------- per-point shading
class _ShadePoints(torch.autograd.Function):
    """(shading normal [P,3], shading [P,1], shaded [P,3]) from the G-buffer rows, the camera/light rows and kd."""

    @staticmethod
    def forward(ctx, gb, par, kd, two_sided, img):
        require_device(gb, par, img, what="shade_points")
        gb, par = f32c(gb), f32c(par)
        P, ncol = gb.shape[0], par.shape[1]
        assert gb.shape == (P, 12) and ncol in (12, 17) and (kd is not None) == (ncol == 17)
        assert par.shape[0] == P if img is None else (img.shape == (P,) and img.dtype == torch.int64 and img.is_contiguous())
        _assert_sorted(img)
        nrm = torch.empty((P, 3), dtype=torch.float32, device=gb.device)
        shading = shaded = None
        kd_stride = 0
        if kd is not None:
            assert kd.shape == (P, 3) and kd.dtype == torch.float32
            if kd.stride(1) != 1:
                kd = kd.contiguous()
            kd_stride = kd.stride(0)
            shading = torch.empty((P, 1), dtype=torch.float32, device=gb.device)
            shaded = torch.empty((P, 3), dtype=torch.float32, device=gb.device)
        # per-image rows: their gradient is accumulated by the backward with atomics -- allocated now, cleared by the forward launch
        g_par = torch.empty_like(par) if (img is not None and ctx.needs_input_grad[1]) else None
        call("a3d_shade_fwd", ptr(gb), ptr(par), ncol, ptr(img), ptr(kd), kd_stride, P, int(two_sided), ptr(nrm), ptr(shading), ptr(shaded),
             ptr(g_par), par.shape[0], stream())
        ctx.save_for_backward(gb, par, kd, img)
        ctx.g_par = g_par
        ctx.two_sided, ctx.kd_stride = int(two_sided), kd_stride
        ctx.set_materialize_grads(False)  # (an output nobody differentiated arrives as None = a NULL pointer, not as a tensor of zeros filled for it)
        if kd is None:
            return nrm
        return nrm, shading, shaded

    @staticmethod
    def backward(ctx, g_nrm, g_shading=None, g_shaded=None):
        gb, par, kd, img = ctx.saved_tensors
        P, ncol = gb.shape[0], par.shape[1]
        g_gb = torch.empty_like(gb)
        g_par, ctx.g_par = ctx.g_par, None  # (cleared by the forward launch; serves one backward)
        clear = g_par is not None
        if g_par is None:
            g_par = torch.empty_like(par)
        g_kd = torch.empty((P, 3), dtype=torch.float32, device=gb.device) if kd is not None else None
        opt = lambda t: None if t is None else f32h(t)
        call("a3d_shade_bwd", ptr(opt(g_nrm)), ptr(opt(g_shading)), ptr(opt(g_shaded)), ptr(gb), ptr(par), ncol, ptr(img), par.shape[0], ptr(kd),
             ctx.kd_stride, P, ctx.two_sided, ptr(g_gb), ptr(g_par), ptr(g_kd), int(clear), stream())
        return g_gb, g_par, g_kd, None, None


def shade_points(gb, par, kd=None, two_sided=True, img=None):
    """Shading normal, Lambert shading and shaded colour at the covered pixels (csrc/shade.hip).

    gb [P,12] from :func:`gbuffer`; par rows (w2c rotation 9, view position 3[, light direction 3, ambient, diffuse]): one per point
    [P,12|17], or -- with ``img`` [P] (point -> image, int64, non-decreasing) -- one per image [B,12|17], in which case their gradient is
    reduced per image inside the backward kernel; kd [P,3] (any row stride).  Returns nrm, or (nrm, shading [P,1], shaded [P,3]) when a
    light is given."""
    return _ShadePoints.apply(gb, par, kd, two_sided, img)


KEEP_SHADED_COLOUR = os.environ.get("A3D_KEEP_SHADED", "1") != "0"  # the compositor keeps the colours it computes on the fly for its blend launch and its backward


class ShadeRecipe:
    """The shaded colour of a fused render, NOT computed yet: what it is computed from.  Handed to shade_composite_antialias the colour of
    every covered pixel is computed inside the compositor's launches (a3d_ca_shade) and the whole backward -- compositor gather, shading
    adjoint -- is ONE autograd node that writes the gradients of the G-buffer rows, of the texture field's output rows and of the camera
    / light tensors where their consumers read them (round 6; a3d_shade_fwd / a3d_shade_bwd and the [B,17] table, its cat and the slices
    around them are not part of that path).  Any other reader calls ``materialize()``: the classic ops.shade_points on the same inputs.

    gb [P,12]; w2c [1|B,4,4] (or [.,3,3]); view_pos [1|B,3]; light [1|B,5] = DirectionalLight.forward(feat) (direction 3, ambient,
    diffuse); all_tex [>= P, >= 3]: the texture field's output rows, kd = columns 0..2 (rows past P: padding, never read)."""

    def __init__(self, gb, w2c, view_pos, light, all_tex, two_sided, img):
        self.gb, self.w2c, self.view_pos, self.light, self.all_tex, self.two_sided, self.img = gb, w2c, view_pos, light, all_tex, bool(two_sided), img
        self.outs = None

    def per_image(self):
        """The [B,17] table of a3d_shade_fwd (rotation 9 | view 3 | light 5) from the same tensors (torch ops: the unfused path)."""
        b = max(self.w2c.shape[0], self.view_pos.shape[0], self.light.shape[0])
        return torch.cat([self.w2c[:, :3, :3].reshape(-1, 9).expand(b, 9), self.view_pos.reshape(-1, 3).expand(b, 3), self.light.expand(b, 5)], dim=-1)

    def materialize(self):
        """(shading normal [P,3], shading [P,1], shaded colour [P,3]) through the stand-alone launch."""
        if self.outs is None:
            P = self.gb.shape[0]
            self.outs = shade_points(self.gb, self.per_image(), self.all_tex[:P, :3], self.two_sided, img=self.img[:P])
        return self.outs


# ---------------------------------------------------------------------------------------------- per-image rows <-> points
class _RowsPerPoint(torch.autograd.Function):
    """t[img] for a per-image tensor t [B,C] and a point -> image map img [P]; backward = per-image segment sums (csrc/segsum.hip)
    instead of index_add's P x C float atomics onto B rows."""

    @staticmethod
    def forward(ctx, t, img):
        ctx.save_for_backward(img)
        ctx.b = t.shape[0]
        return t.index_select(0, img)

    @staticmethod
    def backward(ctx, g):
        (img,) = ctx.saved_tensors
        g = f32c(g)
        P, C = g.shape[0], int(math.prod(g.shape[1:]))
        out = torch.empty((ctx.b,) + tuple(g.shape[1:]), dtype=torch.float32, device=g.device)
        call("a3d_rows_segsum", ptr(g), ptr(img), P, C, ctx.b, ptr(out), stream(), tag=f"[C{C}]")
        return out, None


CHECK_SORTED_INDEX = False  # debugging aid: device-side assert (no host sync) that a point -> image index is non-decreasing


def _assert_sorted(img):
    """The per-image segment sums (csrc/segsum.hip) treat a 128-row block whose first and last rows belong to the same image as
    single-image: the index must be non-decreasing (the render path's is: pixel lists are image-major, padding rows repeat an image)."""
    if CHECK_SORTED_INDEX and img is not None and img.numel() > 1:
        torch._assert_async((img[1:] >= img[:-1]).all(), "point -> image index must be non-decreasing")


def rows_per_point(t, img):
    """t [B,...] -> [P,...] rows gathered by img [P] (int64, contiguous, NON-DECREASING: see _assert_sorted)."""
    require_device(t, img, what="rows_per_point")
    _assert_sorted(img)
    return _RowsPerPoint.apply(t, img) if (t.requires_grad and torch.is_grad_enabled()) else t.index_select(0, img)


class _RowsAddReLU(torch.autograd.Function):
    """relu(y + rows[img]) computed in place on y (the output of a GEMM over the point list)."""

    @staticmethod
    def forward(ctx, y, rows, img):
        require_device(y, rows, img, what="rows_add_relu")
        assert y.is_contiguous() and y.dtype == torch.float32 and y.dim() == 2 and rows.shape[1] == y.shape[1] and img.shape[0] == y.shape[0]
        rows = f32c(rows)
        call("a3d_rows_add_relu_fwd", ptr(y), ptr(rows), ptr(img), y.shape[0], y.shape[1], rows.shape[0], stream(), tag=f"[C{y.shape[1]}]")
        ctx.mark_dirty(y)
        ctx.save_for_backward(y, img)
        ctx.b = rows.shape[0]
        return y

    @staticmethod
    def backward(ctx, g):
        y, img = ctx.saved_tensors
        g = f32c(g)
        g_pre = torch.empty_like(g)
        g_rows = torch.empty((ctx.b, y.shape[1]), dtype=torch.float32, device=g.device)
        call("a3d_rows_add_relu_bwd", ptr(g), ptr(y), ptr(img), y.shape[0], y.shape[1], ctx.b, ptr(g_pre), ptr(g_rows), stream(),
             tag=f"[C{y.shape[1]}]")
        return g_pre, g_rows, None


def rows_add_relu_raw_(y, rows, img):
    """In place, no autograd: y[p] = relu(y[p] + rows[img[p]]) (used inside hostnets' stack Function)."""
    call("a3d_rows_add_relu_fwd", ptr(y), ptr(f32h(rows)), ptr(img), y.shape[0], y.shape[1], rows.shape[0], stream(), tag=f"[C{y.shape[1]}]")
    return y


def rows_segsum_raw(g, img, b):
    """[B,C] per-image sums of g [P,C] (no autograd); img non-decreasing."""
    _assert_sorted(img)
    g = f32c(g)
    out = torch.empty((b, g.shape[1]), dtype=torch.float32, device=g.device)
    call("a3d_rows_segsum", ptr(g), ptr(img), g.shape[0], g.shape[1], b, ptr(out), stream(), tag=f"[C{g.shape[1]}]")
    return out


def rows_add_relu_(y, rows, img):
    """In place: y[p] = relu(y[p] + rows[img[p]]);  y [P,C] fresh GEMM output, rows [B,C], img int64 [P], non-decreasing."""
    _assert_sorted(img)
    return _RowsAddReLU.apply(y, rows, img)


# ---------------------------------------------------------------------------------------------- antialias
class AAAnalysis:
    """Silhouette-crossing work list for one (rast, clip, topology); shared by every colour buffer.
    ``defer``: do not launch now -- the first composite_antialias over this analysis runs it inside its own first launch (possible when
    the rasteriser prepared the screen positions and counters for this clip); every other consumer calls ensure() first."""

    def __init__(self, rast, clip, topo: AATopology, defer=False):
        require_device(rast, clip, what="antialias")
        ensure_resolved(rast)  # (one dictionary miss when nothing is pending -- the render path resolves in its G-buffer launch before this)
        self.rast, self.clip, self.topo = f32c(rast.detach()), f32c(clip.detach()), topo
        B, H, W = rast.shape[:3]
        self.B, self.H, self.W = B, H, W
        shards = _lib.lib().a3d_aa_shards()  # the work list is kept in segments, each with its own append counter
        self.capacity = _lib.lib().a3d_aa_capacity(B, H, W)
        dev = rast.device
        self.work = torch.empty((self.capacity, 4), dtype=torch.int32, device=dev)
        prep = _aa_prepared.take(self.rast)  # (one analysis per prepared pair: the counters are consumed)
        self.prepared = prep is not None and prep[0] == _IdentityCache.key(self.clip)
        if self.prepared:
            self.screen, self.count = prep[1], prep[2]
        else:
            self.count = torch.empty((shards,), dtype=torch.int32, device=dev)
            self.screen = torch.empty((self.clip.shape[0], self.clip.shape[1], 2), dtype=torch.float32, device=dev)
        self.pending = True
        if not (defer and self.prepared and topo.tri.shape[0] > 0):
            self.ensure()

    def ride_args(self):
        """The a3d_aa_ride struct that makes a3d_composite_aa_fwd / a3d_mask_aa_fwd run the pending analysis in their first launch, or None.  The caller marks the
        analysis done (``pending = False``) once that call has SUCCEEDED: a refused call must not leave an unfilled work list behind
        that later consumers would take for finished.  Batches the riding form cannot address (B > 65535) run stand-alone now."""
        if self.pending and self.B > 65535:
            self.ensure()
        if not self.pending:
            return None
        topo, lists = self.topo, getattr(self.topo, "lists", None)
        return _lib.AaRide(size=ctypes.sizeof(_lib.AaRide), clip_batch=self.clip.shape[0], rast=ptr(self.rast), screen=ptr(self.screen),
                           tri=ptr(topo.tri), opp=ptr(topo.opp), off=ptr(None if lists is None else lists.off),
                           adj=ptr(None if lists is None else lists.adj), V=self.clip.shape[1], F=topo.tri.shape[0],
                           lists_stride=0 if lists is None else lists.stride)

    def ensure(self):
        """Run the analysis now if it has not run yet (stand-alone launch)."""
        if not self.pending:
            return self
        self.pending = False
        topo, lists = self.topo, getattr(self.topo, "lists", None)
        call("a3d_aa_analyze", ptr(self.rast), ptr(self.clip), self.clip.shape[0], ptr(topo.tri), ptr(topo.opp), self.B, self.clip.shape[1],
             topo.tri.shape[0], self.H, self.W, ptr(self.screen), ptr(self.work), self.capacity, ptr(self.count), int(self.prepared),
             ptr(None if lists is None else lists.off), ptr(None if lists is None else lists.adj), 0 if lists is None else lists.stride, stream())
        return self


class _Antialias(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, clip, analysis):
        require_device(color, what="antialias")
        color = f32c(color)
        a = analysis.ensure()
        B, H, W, C = color.shape
        assert (B, H, W) == (a.B, a.H, a.W)
        out = torch.empty_like(color)
        call("a3d_aa_fwd", ptr(color), C, ptr(a.work), ptr(a.count), a.capacity, B, H, W, ptr(out), stream(), tag=f"[C{C}]")
        ctx.save_for_backward(color)
        ctx.analysis = a
        return out

    @staticmethod
    def backward(ctx, g_out):
        (color,) = ctx.saved_tensors
        a = ctx.analysis
        B, H, W, C = color.shape
        g_color = torch.empty_like(color)
        g_clip = torch.empty_like(a.clip)
        call("a3d_aa_bwd", ptr(f32h(g_out)), ptr(color), C, ptr(a.work), ptr(a.count), a.capacity, ptr(a.clip), a.clip.shape[0], ptr(a.topo.tri),
             B, a.clip.shape[1], a.topo.tri.shape[0], H, W, ptr(g_color), ptr(g_clip), stream(), tag=f"[C{C}]")
        return g_color, g_clip, None


def _image_gradient_in_place(g, H, W):
    """(tensor, floats between two pixels, channels) of an NHWC image gradient, read where it is when its pixels are evenly strided (a
    channel slice of a wider image: the losses write the gradient of 'dino_pred' into the 17-channel layout of the image it was sliced
    from), else a contiguous copy."""
    ok = g.dtype == torch.float32 and g.dim() == 4 and (g.shape[3] == 1 or g.stride(3) == 1) and g.stride(2) >= g.shape[3]
    ok = ok and (g.shape[1] == 1 or g.stride(1) == W * g.stride(2)) and (g.shape[0] == 1 or g.stride(0) == H * W * g.stride(2))
    if not ok:
        g = f32c(g)
    return g, g.stride(2), g.shape[3]


class _CompositeAntialias(torch.autograd.Function):
    """One or two buffers (vals2 None = one) against the same pixel list and crossing records, in the same launches.

    ``keep`` / ``keep2``: leading channels of the composited image that are materialised and handed out (None = all C + 1): render_mesh
    returns 'dino_pred' and 'flow' without their alpha channel (render.py:320-331) -- here that channel is never written, the image comes
    out contiguous in the channels the caller keeps, and its gradient arrives without a SliceBackward (a zero fill and a strided copy of
    the whole image) in front of it.
    Shading recipe (``gb`` .. ``all_tex`` given, ``vals`` None): the first buffer's colour is kd * shading computed inside the launches,
    and the backward of this node runs the shading adjoint too (a3d_shade_bwd_rows): g_gb [P,12], the gradient of the texture field's
    output rows [rows,T] and of w2c / view_pos / light in their own layouts.
    ``vals`` / ``vals2`` / ``all_tex`` may have MORE rows than the list (the fields' padded point list): the extra rows are never read and
    their gradient rows are written as zeros by the same launches."""

    @staticmethod
    def forward(ctx, vals, vals2, clip, pix, inv, bg, bg2, analysis, keep, keep2, gb, w2c, view, light, all_tex, two_sided):
        require_device(vals, vals2, pix, inv, gb, all_tex, what="composite_antialias")
        a = analysis
        P = pix.shape[0]
        assert pix.dtype == torch.int64 and inv.shape == (a.B * a.H * a.W,) and inv.dtype == torch.int32
        shade = gb is not None
        dev = pix.device

        def prep(v, g, k):
            if v is None:
                return None, None, 0, None
            v = f32c(v)
            assert v.shape[0] >= P
            C = v.shape[1]
            if g is not None:
                g = f32c(g)
                assert g.shape[1:3] == (a.H, a.W) and g.shape[3] <= C + 1 and g.shape[0] in (1, a.B)
            k = C + 1 if k is None else int(k)
            assert 1 <= k <= C + 1
            return v, g, C, torch.empty((a.B, a.H, a.W, k), dtype=torch.float32, device=dev)  # (only the channels handed out are materialised)

        par_struct = sh = g_par = None
        if shade:
            assert vals is None and light is not None
            ctx.view_shape = view.shape
            gb, w2c, view, light, all_tex = f32c(gb), f32c(w2c), f32c(view.reshape(-1, 3)), f32c(light), f32c(all_tex)
            assert gb.shape == (P, 12) and all_tex.shape[0] >= P and all_tex.shape[1] >= 3 and w2c.shape[1] == w2c.shape[2] and w2c.shape[1] in (3, 4)
            assert all(t.shape[0] in (1, a.B) for t in (w2c, view, light)) and light.shape[1] == 5
            bg = None if bg is None else f32c(bg)
            assert bg is None or (bg.shape[1:3] == (a.H, a.W) and bg.shape[3] <= 4 and bg.shape[0] in (1, a.B))
            if bg is not None and bg.shape[3] == 4 and bg.data_ptr() % 16:  # (an offset view: the compose kernel reads 4-channel texels as 16-byte loads)
                bg = bg.clone()
            C, out = 3, torch.empty((a.B, a.H, a.W, 4), dtype=torch.float32, device=dev)
            # gradients of the per-image tensors: accumulated by the backward with atomics into ONE buffer (cleared by the forward's
            # first launch), handed out as views in the tensors' own shapes
            if any(ctx.needs_input_grad[11:14]):
                g_par = torch.empty(w2c.numel() + view.numel() + light.numel(), dtype=torch.float32, device=dev)
            par_struct = _shade_params(w2c, view, light)
            # (the colours the compose launch computes are kept, 12 B per point: the blend launch and the backward read them as value rows)
            shaded = torch.empty((P, 3), dtype=torch.float32, device=dev) if KEEP_SHADED_COLOUR else None
            sh = _lib.CaShade(size=ctypes.sizeof(_lib.CaShade), kd_stride=all_tex.stride(0), gb=ptr(gb), par=None, kd=ptr(all_tex), clear=ptr(g_par),
                              n_clear=0 if g_par is None else g_par.numel(), two_sided=int(two_sided), params=ctypes.addressof(par_struct),
                              shaded_out=ptr(shaded))
        else:
            vals, bg, C, out = prep(vals, bg, keep)
        vals2, bg2, C2, out2 = prep(vals2, bg2, keep2)
        ctag = lambda c, o: f"C{c + 1}" if o.shape[3] == c + 1 else f"C{c + 1}>{o.shape[3]}"  # (C17>16: a 17-channel composite of which 16 channels are materialised)
        tag = f"[{ctag(C, out)}]" if vals2 is None else f"[{ctag(C, out)}+{ctag(C2, out2)}]"
        ride = a.ride_args()  # a deferred analysis runs inside the first launch of this call
        buf = lambda v, c, g, o: _lib.CaBuffer(size=ctypes.sizeof(_lib.CaBuffer), C=c, vals=ptr(v), bg=ptr(g), out=ptr(o), bg_batch=0 if g is None else g.shape[0],
                                               bg_channels=0 if g is None else g.shape[3], out_channels=o.shape[3])
        first, second = buf(None if shade else vals, C, bg, out), (buf(vals2, C2, bg2, out2) if vals2 is not None else None)
        call("a3d_composite_aa_fwd", ctypes.addressof(first), None if second is None else ctypes.addressof(second), ptr(inv), ptr(a.work), ptr(a.count),
             a.capacity, a.B, a.H, a.W, None if ride is None else ctypes.addressof(ride), None if sh is None else ctypes.addressof(sh), stream(),
             tag=tag + ("[+shade]" if shade else "") + ("[+analysis]" if ride is not None else ""))
        if ride is not None:
            a.pending = False  # (only now: the call above raises on a refused argument)
        ctx.save_for_backward(shaded if shade else vals, vals2, pix, inv, bg, bg2, gb, w2c, view, light, all_tex)
        ctx.analysis, ctx.tag, ctx.g_par, ctx.two_sided, ctx.shade = a, tag, g_par, int(bool(two_sided)), shade
        ctx.set_materialize_grads(False)
        return out, out2

    @staticmethod
    def backward(ctx, g_out, g_out2=None):
        vals, vals2, pix, inv, bg, bg2, gb, w2c, view, light, all_tex = ctx.saved_tensors
        a, shade = ctx.analysis, ctx.shade
        P = pix.shape[0]
        dev = pix.device
        C = 3 if shade else vals.shape[1]
        two = vals2 is not None
        C2 = vals2.shape[1] if two else 0
        if g_out is None:  # (an output nobody differentiated: zero gradient)
            g_out = torch.zeros((a.B, a.H, a.W, C + 1), dtype=torch.float32, device=dev)
        if two and g_out2 is None:
            g_out2 = torch.zeros((a.B, a.H, a.W, C2 + 1), dtype=torch.float32, device=dev)
        g_out, gs, gc = _image_gradient_in_place(g_out, a.H, a.W)
        if two:
            g_out2, gs2, gc2 = _image_gradient_in_place(g_out2, a.H, a.W)
        rows = P if shade else vals.shape[0]
        g_vals = torch.empty((rows, C), dtype=torch.float32, device=dev)  # (shade: the gradient of the shaded colour, consumed below)
        g_vals2 = torch.empty_like(vals2) if two else None
        g_clip = torch.empty_like(a.clip)
        par_struct = sh = None
        if shade:
            par_struct = _shade_params(w2c, view, light)
            if vals is None:  # (no kept colours: the backward's kernels re-derive them from the recipe)
                sh = _lib.CaShade(size=ctypes.sizeof(_lib.CaShade), kd_stride=all_tex.stride(0), gb=ptr(gb), par=None, kd=ptr(all_tex), clear=None, n_clear=0,
                                  two_sided=ctx.two_sided, params=ctypes.addressof(par_struct))
        buf = lambda v, c, g, go, gst, gch, gv, nrows: _lib.CaBuffer(size=ctypes.sizeof(_lib.CaBuffer), C=c, vals=ptr(v), bg=ptr(g), g_out=ptr(go), g_vals=ptr(gv),
                                                                     bg_batch=0 if g is None else g.shape[0], bg_channels=0 if g is None else g.shape[3],
                                                                     g_stride=gst, g_channels=gch, vals_rows=nrows)
        first = buf(vals, C, bg, g_out, gs, gc, g_vals, rows)  # (shade: ``vals`` = the colours the forward kept, or None)
        second = buf(vals2, C2, bg2, g_out2, gs2, gc2, g_vals2, vals2.shape[0]) if two else None
        call("a3d_composite_aa_bwd", ctypes.addressof(first), None if second is None else ctypes.addressof(second), ptr(pix), P, ptr(inv), ptr(a.work),
             ptr(a.count), a.capacity, ptr(a.clip), a.clip.shape[0], ptr(a.topo.tri), a.B, a.clip.shape[1], a.topo.tri.shape[0], a.H, a.W, ptr(g_clip),
             None if sh is None else ctypes.addressof(sh), stream(), tag=ctx.tag)
        if not shade:
            return (g_vals, g_vals2, g_clip) + (None,) * 13
        # the shading adjoint in the same node: g_vals -> G-buffer rows, texture rows (kd columns; the other columns and the padding rows
        # zero), camera / light tensors -- each in the layout its consumer reads
        g_par, ctx.g_par = ctx.g_par, None  # (cleared by the forward's first launch; serves ONE backward)
        if g_par is None:
            g_par = torch.zeros(w2c.numel() + view.numel() + light.numel(), dtype=torch.float32, device=dev)
        n0, n1 = w2c.numel(), w2c.numel() + view.numel()
        g_w2c, g_view, g_light = g_par[:n0].view(w2c.shape), g_par[n0:n1].view(view.shape), g_par[n1:].view(light.shape)
        g_struct = _shade_params(g_w2c, g_view, g_light)
        g_gb = torch.empty_like(gb)
        g_tex = torch.empty_like(all_tex)
        call("a3d_shade_bwd_rows", ptr(g_vals), ptr(gb), ctypes.addressof(par_struct), ctypes.addressof(g_struct), ptr(pix), a.H * a.W, ptr(all_tex),
             all_tex.stride(0), P, ctx.two_sided, ptr(g_gb), ptr(g_tex), all_tex.shape[1], all_tex.shape[0], stream())
        return (None, g_vals2, g_clip) + (None,) * 7 + (g_gb, g_w2c, g_view.view(ctx.view_shape), g_light, g_tex, None)


def _shade_params(w2c, view, light):
    """a3d_shade_params over the tensors themselves: w2c [1|B,4,4] / [1|B,3,3], view [1|B,3], light [1|B,5] (contiguous float32)."""
    n = w2c.shape[1]
    return _lib.ShadeParams(size=ctypes.sizeof(_lib.ShadeParams), rot_row_stride=n, rot=ptr(w2c), view=ptr(view), light=ptr(light),
                            rot_image_stride=0 if w2c.shape[0] == 1 else n * n, view_image_stride=0 if view.shape[0] == 1 else 3,
                            light_image_stride=0 if light.shape[0] == 1 else 5)


def composite_antialias(vals, pix, inv, background, clip, analysis, vals2=None, background2=None, keep=None, keep2=None):
    """antialias(lerp(background, [vals, 1], coverage)) for a buffer given as rows ``vals`` [>= P,C] at the covered pixels ``pix`` (``inv`` =
    the pixel -> row map of covered_pixels(return_inverse=True)): [B,H,W,C+1].  ``background`` [1|B,H,W,<= C+1] (missing trailing channels
    = 0: the reference's 3-channel background as it is) or None (zeros); it gets no gradient (callers with a differentiable background
    composite with torch and call antialias).  With ``vals2`` (and ``background2``) a second buffer over the same pixels is composited
    and antialiased by the same launches: returns the pair of images.  ``keep`` / ``keep2``: see _CompositeAntialias."""
    assert background is None or not background.requires_grad
    assert background2 is None or not background2.requires_grad
    if clip.dim() == 2:
        clip = clip[None]
    o1, o2 = _CompositeAntialias.apply(vals, vals2, clip, pix, inv, background, background2, analysis, keep, keep2, None, None, None, None, None, True)
    return o1 if vals2 is None else (o1, o2)


def shade_composite_antialias(recipe, pix, inv, background, clip, analysis, vals2=None, background2=None, keep2=None):
    """composite_antialias for the shaded colour of ``recipe`` (a ShadeRecipe) as first buffer: the colour is computed inside the launches,
    and one backward node runs compositor gather + shading adjoint (see _CompositeAntialias)."""
    assert background is None or not background.requires_grad
    assert background2 is None or not background2.requires_grad
    if clip.dim() == 2:
        clip = clip[None]
    r = recipe
    o1, o2 = _CompositeAntialias.apply(None, vals2, clip, pix, inv, background, background2, analysis, None, keep2, r.gb, r.w2c, r.view_pos, r.light,
                                       r.all_tex, r.two_sided)
    return o1 if vals2 is None else (o1, o2)


def antialias(color, rast, clip, tri, analysis=None):
    """dr.antialias semantics; pass ``analysis`` (AAAnalysis) to share the geometry pass between buffers."""
    if clip.dim() == 2:
        clip = clip[None]
    if analysis is None:
        tri32 = tri_int32(tri)
        analysis = AAAnalysis(rast, clip, aa_topology(tri32, clip.shape[1]))
    return _Antialias.apply(color, clip, analysis)


# ---------------------------------------------------------------------------------------------- reconstruction losses
class _MaskAntialias(torch.autograd.Function):
    """lerp(bg, [1, .., 1], coverage) + antialias for a texture-less, light-less render (csrc/antialias.hip: a3d_mask_aa_*): the coverage
    comes from the raster texels, only the silhouette is differentiated (g_clip)."""

    @staticmethod
    def forward(ctx, rast, clip, bg, analysis, C):
        require_device(rast, clip, what="mask_antialias")
        a = analysis
        rast_c = f32c(rast.detach())
        if bg is not None:
            bg = f32c(bg)
            assert bg.shape[1:] == (a.H, a.W, C + 1) and bg.shape[0] in (1, a.B)
        out = torch.empty((a.B, a.H, a.W, C + 1), dtype=torch.float32, device=rast.device)
        ride = a.ride_args()
        call("a3d_mask_aa_fwd", ptr(rast_c), C, ptr(bg), 0 if bg is None else bg.shape[0], ptr(out), ptr(a.work), ptr(a.count), a.capacity, a.B, a.H,
             a.W, None if ride is None else ctypes.addressof(ride), stream(), tag=f"[C{C + 1}]" + ("[+analysis]" if ride is not None else ""))
        if ride is not None:
            a.pending = False
        ctx.save_for_backward(rast_c, bg)
        ctx.analysis, ctx.C = a, C
        return out

    @staticmethod
    def backward(ctx, g_out):
        rast_c, bg = ctx.saved_tensors
        a, C = ctx.analysis, ctx.C
        g_clip = torch.empty_like(a.clip)
        # the image usually went on as permute(0, 3, 1, 2) + a channel slice (Fauna.py:166-173): the engine then hands back a channels-FIRST
        # tensor seen through the inverse permute -- read where it is (its contiguous copy was 10 us of the Fauna step)
        first = g_out.dtype == torch.float32 and g_out.dim() == 4 and not g_out.is_contiguous() and g_out.permute(0, 3, 1, 2).is_contiguous()
        if not first:
            g_out = f32h(g_out)
        call("a3d_mask_aa_bwd", ptr(g_out), ptr(rast_c), C, ptr(bg), 0 if bg is None else bg.shape[0], ptr(a.work), ptr(a.count), a.capacity,
             ptr(a.clip), a.clip.shape[0], ptr(a.topo.tri), a.B, a.clip.shape[1], a.topo.tri.shape[0], a.H, a.W, ptr(g_clip), int(first), stream(),
             tag=f"[C{C + 1}]")
        return None, g_clip, None, None, None


def mask_antialias(rast, clip, background, analysis, channels=3):
    """[B,H,W,channels+1]: ones where ``rast`` is covered, ``background`` ([1|B,H,W,channels+1] or None = zeros) elsewhere, antialiased
    with ``analysis`` (an AAAnalysis of the same rast / clip; may still be pending: it then runs inside this op's first launch)."""
    assert analysis.rast.data_ptr() == f32c(rast.detach()).data_ptr(), "the analysis must belong to this raster buffer"
    return _MaskAntialias.apply(rast.detach(), clip, background, analysis, int(channels))  # (coverage carries no gradient)


class _ReconLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, shaded, dino, image_gt, dino_gt, mask_gt, dt0, dt1, valid):
        require_device(shaded, image_gt, mask_gt, dt0, valid, what="reconstruction_losses")
        B, H, W = shaded.shape[:3]
        D = 0 if dino is None else dino.shape[3]
        assert shaded.shape == (B, H, W, 4) and shaded.is_contiguous() and image_gt.shape == (B, 3, H, W)
        ds = D if dino is None else dino.stride(2)  # floats between two pixels (D + 1: the renderer's 17-channel image read in place)
        assert dt0.shape == (B, H, W) and dt0.stride(2) == 1 and dt0.stride(1) == W
        assert dt1 is None or (dt1.shape == dt0.shape and dt1.stride() == dt0.stride())
        image_gt, mask_gt, valid = f32c(image_gt), f32c(mask_gt), f32c(valid)
        if D:
            assert dino.stride(3) == 1 and ds >= D and dino.stride(1) == W * ds and dino.stride(0) == H * W * ds and dino_gt.shape == (B, D, H, W)
            dino_gt = f32c(dino_gt)
        loss = torch.empty((B, _lib.lib().a3d_recon_losses_columns()), dtype=torch.float32, device=shaded.device)
        scratch = torch.empty(_lib.lib().a3d_recon_losses_scratch_bytes(B, H, W), dtype=torch.uint8, device=shaded.device)
        both = torch.empty(_lib.lib().a3d_recon_losses_mask_bytes(B, H, W), dtype=torch.uint8, device=shaded.device)
        call("a3d_recon_losses_fwd", ptr(shaded), ptr(dino), D, ds, ptr(image_gt), ptr(dino_gt), ptr(mask_gt), ptr(dt0), ptr(dt1), dt0.stride(0), ptr(valid),
             B, H, W, ptr(scratch), ptr(both), ptr(loss), stream())
        ctx.save_for_backward(shaded, dino, image_gt, dino_gt, mask_gt, dt0, dt1, valid, both)
        ctx.mark_non_differentiable(both)
        ctx.set_materialize_grads(False)  # (no zero tensor filled for the mask output's "gradient")
        return loss, both

    @staticmethod
    def backward(ctx, g_loss, _g_both=None):
        shaded, dino, image_gt, dino_gt, mask_gt, dt0, dt1, valid, both = ctx.saved_tensors
        B, H, W = shaded.shape[:3]
        D = 0 if dino is None else dino.shape[3]
        if g_loss is None:
            return (None,) * 8
        g_shaded = torch.empty_like(shaded)
        # the feature gradient in the layout of the image the features were sliced from (stride D + 1: the alpha slot stays unwritten, it
        # has no gradient): the compositor's backward reads it where it is, no zero-padded copy in between
        ds = dino.stride(2) if D else 0
        g_dino = torch.empty((B, H, W, ds), dtype=torch.float32, device=shaded.device)[..., :D] if D else None
        call("a3d_recon_losses_bwd", ptr(f32h(g_loss)), ptr(shaded), ptr(dino), D, ds, ds, ptr(image_gt), ptr(dino_gt), ptr(mask_gt), ptr(dt0), ptr(dt1),
             dt0.stride(0), ptr(valid), B, H, W, ptr(both), ptr(g_shaded), ptr(g_dino), stream())
        return g_shaded, g_dino, None, None, None, None, None, None


def reconstruction_losses(shaded_nchw, dino_nchw, image_gt, dino_gt, mask_gt, mask_dt, mask_valid, return_mask=False):
    """Per-frame [N,5] = (mask, mask_inv_dt, rgb, dino, mask_dt) losses of compute_reconstruction_losses (AnimalModel.py:260-307,
    background_mode 'none'; N = images x frames) from render_mesh's outputs: ``shaded_nchw`` [N,4,H,W] and ``dino_nchw`` [N,D,H,W] (or
    None) are the NCHW views render_mesh returns (NHWC in memory -- read in place, no copy); targets in the dataset's NCHW layout;
    mask_dt [N,2,H,W].  ``return_mask=True`` also returns the eroded common mask (uint8 [N*H*W]) that :func:`flow_loss` needs."""
    shaded = shaded_nchw.permute(0, 2, 3, 1)
    shaded = shaded if shaded.is_contiguous() else shaded.contiguous()
    dino = None
    if dino_nchw is not None:
        dino = dino_nchw.permute(0, 2, 3, 1)
        H, W, S = dino.shape[1], dino.shape[2], dino.stride(2)
        # (render_mesh's 'dino_pred' = the first 16 channels of its 17-channel image: evenly strided pixels are read in place)
        if not (dino.dtype == torch.float32 and dino.stride(3) == 1 and S >= dino.shape[3] and dino.stride(1) == W * S and dino.stride(0) == H * W * S):
            dino = dino.contiguous()
    loss, both = _ReconLosses.apply(shaded, dino, image_gt, dino_gt, mask_gt, mask_dt[:, 0], mask_dt[:, 1], mask_valid)
    return (loss, both) if return_mask else loss


class _FlowLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flow, flow_gt, both, B, F):
        require_device(flow, flow_gt, both, what="flow_loss")
        N, H, W = flow.shape[:3]
        assert N == B * F and F >= 2 and flow.shape[3] == 2 and flow.stride(3) == 1 and flow.stride(1) == W * flow.stride(2) \
            and flow.stride(0) == H * flow.stride(1), "flow must be the renderer's NHWC buffer (channel slice allowed)"
        flow_gt = f32c(flow_gt)
        assert flow_gt.shape == (B, F - 1, 2, H, W) and both.numel() == N * H * W
        loss = torch.empty((B, F - 1), dtype=torch.float32, device=flow.device)
        scale = torch.empty_like(loss)
        scratch = torch.empty(_lib.lib().a3d_flow_loss_scratch_bytes(B, F, H, W), dtype=torch.uint8, device=flow.device)
        call("a3d_flow_loss_fwd", ptr(flow), flow.stride(2), ptr(flow_gt), ptr(both), B, F, H, W, ptr(scratch), ptr(loss), ptr(scale), stream())
        ctx.save_for_backward(flow, flow_gt, both, scale)
        ctx.dims = (B, F, H, W)
        return loss

    @staticmethod
    def backward(ctx, g_loss):
        flow, flow_gt, both, scale = ctx.saved_tensors
        B, F, H, W = ctx.dims
        g_flow = torch.empty((B * F, H, W, 2), dtype=torch.float32, device=flow.device)
        call("a3d_flow_loss_bwd", ptr(f32h(g_loss)), ptr(scale), ptr(flow), flow.stride(2), ptr(flow_gt), ptr(both), B, F, H, W, ptr(g_flow), stream())
        return g_flow, None, None, None, None


def flow_loss(flow_nchw, flow_gt, both, batch, num_frames):
    """[B,F-1] flow loss between consecutive frames (AnimalModel.py:285-298) from render_mesh's 'flow' output [B*F,2,H,W] (the NCHW
    view of the renderer's NHWC buffer, read in place), flow_gt [B,F-1,2,H,W] and the eroded common mask of
    :func:`reconstruction_losses` (``return_mask=True``)."""
    return _FlowLoss.apply(flow_nchw.permute(0, 2, 3, 1), flow_gt, both, int(batch), int(num_frames))


# ---------------------------------------------------------------------------------------------- harmonic embedding
class _HarmonicEmbed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, freq, symmetrize, ones):
        require_device(x, freq, what="harmonic_embed")
        x, freq = f32c(x), f32c(freq)
        P, n = x.shape[0], freq.shape[0]
        assert x.shape == (P, 3)
        out = torch.empty((P, 3 + 6 * n + int(ones)), dtype=torch.float32, device=x.device)
        call("a3d_harmonic_embed_fwd", ptr(x), ptr(freq), n, int(symmetrize), int(ones), P, ptr(out), stream())
        ctx.save_for_backward(x, freq)
        ctx.cfg = (int(symmetrize), int(ones))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, freq = ctx.saved_tensors
        g_x = torch.empty_like(x)
        call("a3d_harmonic_embed_bwd", ptr(f32h(g)), ptr(x), ptr(freq), freq.shape[0], ctx.cfg[0], ctx.cfg[1], x.shape[0], ptr(g_x), stream())
        return g_x, None, None, None


def harmonic_embed(x, freq, symmetrize=False, ones=False):
    """[P,3] -> [P, 3 + 6n (+1)] = [x (|x_0| if symmetrize), sin(x_c f_k), cos(x_c f_k), (1)] (csrc/embed.hip).  First-order
    differentiable only (double backward raises): the SDF regulariser, which differentiates the field twice, takes the torch path."""
    return _HarmonicEmbed.apply(x, freq, symmetrize, ones)


# ---------------------------------------------------------------------------------------------- fp32 MFMA GEMM + ReLU adjoint
def gemm_nn_relumask(a, b, x=None):
    """(a [M,K] @ b [K,256]) * (x [M,256] > 0)  (x None: plain product) -- csrc/gemm.hip, fp32 MFMA.  No autograd: it is the
    input-gradient GEMM inside hostnets' stack backward."""
    require_device(a, b, what="gemm_nn_relumask")
    a, b = f32c(a), f32c(b)
    M, K = a.shape
    assert b.shape == (K, 256) and K % 32 == 0 and (x is None or (x.shape == (M, 256) and x.is_contiguous() and x.dtype == torch.float32))
    c = torch.empty((M, 256), dtype=torch.float32, device=a.device)
    call("a3d_gemm_nn_relumask", ptr(a), ptr(b), ptr(x), M, 256, K, ptr(c), stream())
    return c


# ---------------------------------------------------------------------------------------------- mixed precision
def _amp_wrap_functions():
    """Every autograd.Function of this module runs its forward with autocast OFF on float32 copies of half-precision inputs, and its
    backward in the same state (torch.amp.custom_fwd / custom_bwd): the kernels are float32, so are their outputs and the gradients
    they return -- autograd casts a gradient back to the dtype of a bf16 / fp16 input.  What the reference does by hand with .float()
    at the nvdiffrast boundary (render.py:265,292) under Trainer.py:208-218's autocast."""
    from torch.amp import custom_bwd, custom_fwd

    for obj in list(globals().values()):
        if isinstance(obj, type) and issubclass(obj, torch.autograd.Function) and obj is not torch.autograd.Function and obj.__module__ == __name__:
            obj.forward = staticmethod(custom_fwd(obj.forward, device_type="cuda", cast_inputs=torch.float32))
            obj.backward = staticmethod(custom_bwd(obj.backward, device_type="cuda"))


_amp_wrap_functions()

if _lib.GUARD:  # A3D_GUARD in the environment: guarded allocations from the first call on (see _lib.set_guard)
    _lib.set_guard(_lib.GUARD)
