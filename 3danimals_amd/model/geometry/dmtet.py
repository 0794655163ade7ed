"""DMTet geometry -- same public API as /root/reference/model/geometry/dmtet.py, HIP underneath.

* ``DMTet.__call__(pos_nx3, sdf_n, tet_fx4) -> (verts, faces, uvs, uv_idx)`` (reference dmtet.py:104-155):
  ``faces`` / ``uv_idx`` int64 and bit-identical to the reference, ``verts`` differentiable w.r.t. ``sdf_n``
  and ``pos_nx3``.  The sort + ``torch.unique`` + ~15 boolean-mask compactions of the reference become two
  HIP phases over the grid's static sorted edge list (csrc/dmtet.hip), with one 16-byte read-back.
* ``DMTetGeometry`` (reference dmtet.py:175-310) keeps its attributes (``verts``, ``indices``, ``all_edges``,
  ``current_sdf``, ``mesh_verts``, ``mlp``) and methods; the SDF MLP stays a PyTorch module.
* ``uvs`` depend only on the number of tets (dmtet.py:69-84): built once per grid and cached instead of
  regenerating ~50 MB per call.
"""
from __future__ import annotations

import ctypes
import os
import warnings

import numpy as np
import torch

from ... import ops, tetgrid
from ..render import mesh

try:  # overlaid on the reference tree: use its (unchanged) networks
    from model.networks import CoordMLP, CoordMLP_Mod  # type: ignore

    NETWORKS = "model.networks (reference tree)"
except ImportError:  # stand-alone: same classes / state_dict layout from hostnets (any other error in the overlay must surface)
    from ...hostnets import CoordMLP, CoordMLP_Mod

    NETWORKS = "3danimals_amd.hostnets (stand-alone)"


SURFACE_BUCKET = 1024  # row padding of the surface-adjacent SDF re-evaluation (0 = off)
SURFACE_POINTS_IN_EMIT = os.environ.get("A3D_SURFACE_POINTS_IN_EMIT", "1") != "0"  # the rows of that block written by the DMTet emit launch


class TetGridTopology:
    """Static per-grid device buffers the kernels stream: tets/edges/tet2edge as int32."""

    def __init__(self, indices: torch.Tensor, positions: torch.Tensor = None):
        """``positions`` [Nv,3] (optional; any scale): where the grid's vertices are -- only ever used to choose an order of evaluation
        (spatial_order), never for a result.  Without them the first extraction's own ``pos_nx3`` is taken."""
        idx = indices.long()
        dev = idx.device
        nv = int(idx.max().item()) + 1
        slots = torch.as_tensor(tetgrid.TET_EDGE_SLOTS, device=dev)
        pairs = idx[:, slots]  # [Nt,6,2]
        lo, hi = pairs.amin(-1).reshape(-1), pairs.amax(-1).reshape(-1)
        uniq, inverse = torch.unique(lo * nv + hi, return_inverse=True)  # sorted == lexicographic (min,max)
        self.all_edges = torch.stack([uniq // nv, uniq % nv], -1)  # int64, == reference generate_edges (dmtet.py:283-288)
        self.edges32 = self.all_edges.to(torch.int32).contiguous()
        self.tet2edge32 = inverse.reshape(-1, 6).to(torch.int32).contiguous()
        self.tets32 = idx.to(torch.int32).contiguous()
        self.num_verts = nv
        self._uvs = None
        self._word_groups = None
        self._face_list_stride = None
        self._positions = positions.detach() if positions is not None else None
        self._order = None

    def face_list_stride(self) -> int:
        """Slots per surface vertex that hold its whole vertex -> face list whatever the SDF: a surface vertex sits on a grid edge, every
        tet around that edge contributes at most two triangles at it, so 2 x (the largest number of tets around one edge) bounds the
        valence (12 on the Kuhn grids); rounded up to a multiple of 8.  Static per grid."""
        if self._face_list_stride is None:
            most = int(torch.bincount(self.tet2edge32.reshape(-1).long()).max())
            self._face_list_stride = -(-2 * most // 8) * 8
        return self._face_list_stride

    WORD_GROUPS = True  # False: the count pass streams every index row (a3d_dmtet_count without the cull)
    WORD_GROUPS_MAX_DENSE = 0.25  # (fraction of words with more groups than slots: those are always read)

    def word_groups(self):
        """(edge_groups, tet_groups) int32 [blocks * words_per_block, slots] for the culled count pass, or None.

        Row w lists the distinct aligned 2^bits-vertex groups (vertex index >> bits) that the 64 consecutive index rows of word w touch,
        padded by repetition; -1 in every slot = more than ``slots`` groups (the kernel then always reads the word's rows).  A property
        of the grid alone, built once.  None when more than half of the words are of that kind (a grid numbered without regard to
        space): the plain count pass is then the faster one."""
        if self._word_groups is None:
            if not self.WORD_GROUPS:
                return None
            from ... import _lib

            lib = _lib.lib()
            bits, block = lib.a3d_dmtet_word_group_bits(), lib.a3d_dmtet_block_items()
            self._word_groups = False
            # 8 slots (32 B per word) where that holds most words -- grids numbered along their rows --, else 16 (64 B: spatially
            # coherent files whose words touch more groups, e.g. a BCC lattice in its generator's order); a file whose row order
            # ignores space fails both and takes the ordered pass (spatial_order)
            for slots in (lib.a3d_dmtet_word_group_slots(), 16):
                e, t = _word_groups(self.edges32, slots, bits, block), _word_groups(self.tets32, slots, bits, block)
                dense = (int((e[:, 0] < 0).sum()) + int((t[:, 0] < 0).sum())) / float(e.shape[0] + t.shape[0])
                if dense <= self.WORD_GROUPS_MAX_DENSE:
                    self._word_groups = (e, t)
                    break
        return self._word_groups or None

    SPATIAL_ORDER = True  # False: a grid numbered without regard to space takes the plain count pass (a3d_dmtet_count, no tables)
    ORDER_SLOTS = 16  # group ids per word of the ranked lists (8 or 16; 16: one 64-byte line per word, 92 % of a scrambled BCC lattice's words fit)
    ORDER_MAX_DENSE = 0.5  # above this fraction of words with more groups than slots nothing is gained over the streaming pass

    def spatial_order(self, positions: torch.Tensor = None):
        """The static tables of a3d_dmtet_count_ordered (include/a3d.h: a3d_dmtet_order) for a grid whose file numbering ignores space --
        the reference's Quartet grids (dmtet.py:214-226) -- or None (switched off, or most words of the ranked lists need more than
        ORDER_SLOTS groups: then nothing is gained over the plain pass).  Vertices ranked along a Morton curve through ``positions``
        (16 bits per axis); edge rows rewritten in ranks and sorted by (lower, higher) rank; tet rows rewritten in ranks with their corner
        order kept and sorted by (lowest, second lowest) rank; the word groups of those two lists.  Built once per grid; the positions
        decide how much of the grid a later call can skip, never what it returns."""
        if self._order is None:
            pos = positions if positions is not None else self._positions
            if not self.SPATIAL_ORDER or pos is None:
                return None
            from ... import _lib

            lib = _lib.lib()
            bits, block = lib.a3d_dmtet_word_group_bits(), lib.a3d_dmtet_block_items()
            nv, dev = self.num_verts, self.tets32.device
            p = pos.detach().to(dev, torch.float64).reshape(-1, 3)[:nv]
            lo = p.amin(0)
            q = ((p - lo) / (p.amax(0) - lo).max().clamp(min=1e-30) * 65535.0).round().long().clamp(0, 65535)
            key = _interleave3(q[:, 0]) | (_interleave3(q[:, 1]) << 1) | (_interleave3(q[:, 2]) << 2)
            vertex_of_rank = torch.argsort(key, stable=True)
            rank = torch.empty_like(vertex_of_rank)
            rank[vertex_of_rank] = torch.arange(nv, device=dev)
            er = rank[self.edges32.long()]
            er = torch.stack([er.amin(1), er.amax(1)], 1)
            e_row = torch.argsort(er[:, 0] * nv + er[:, 1], stable=True)
            tr = rank[self.tets32.long()]
            two = tr.sort(1).values[:, :2]
            t_row = torch.argsort(two[:, 0] * nv + two[:, 1], stable=True)
            edges_ranked, tets_ranked = er[e_row].to(torch.int32).contiguous(), tr[t_row].to(torch.int32).contiguous()
            eg, tg = _word_groups(edges_ranked, self.ORDER_SLOTS, bits, block), _word_groups(tets_ranked, self.ORDER_SLOTS, bits, block)
            dense = (int((eg[:, 0] < 0).sum()) + int((tg[:, 0] < 0).sum())) / float(eg.shape[0] + tg.shape[0])
            if dense > self.ORDER_MAX_DENSE:
                self._order = False
            else:
                tensors = dict(vertex_of_rank=vertex_of_rank.to(torch.int32).contiguous(), edges_ranked=edges_ranked,
                               edge_of_row=e_row.to(torch.int32).contiguous(), tets_ranked=tets_ranked,
                               tet_of_row=t_row.to(torch.int32).contiguous(), edge_groups=eg, tet_groups=tg)
                struct = _lib.DmtetOrder(size=ctypes.sizeof(_lib.DmtetOrder), group_slots=self.ORDER_SLOTS,
                                         **{k: v.data_ptr() for k, v in tensors.items()})
                self._order = (struct, tensors)  # (the struct holds raw pointers: the tensors live as long as it does)
        return self._order or None

    def count_pass(self, num_verts: int = None, positions: torch.Tensor = None) -> str:
        """Which count pass an extraction on this grid takes: 'plain' (a3d_dmtet_count streaming every index row: small grids, or
        nothing better available), 'culled' (a3d_dmtet_count with the word groups of the file's own row order: grids numbered along
        space) or 'ordered' (a3d_dmtet_count_ordered: any numbering)."""
        from ... import ops

        nv = self.num_verts if num_verts is None else num_verts
        if nv < ops.DMTET_CULL_MIN_VERTS:
            return "plain"
        if self.word_groups() is not None:
            return "culled"
        return "ordered" if self.spatial_order(positions) is not None else "plain"

    def words_read(self, sdf: torch.Tensor):
        """Diagnostic (measurement only): (edge words read, edge words, tet words read, tet words) of the culled count pass for this SDF,
        by the kernel's own rule -- a word is skipped when all its groups read 0x0000 or all 0xffff in the sign plane -- or None."""
        from ... import _lib

        groups, rows = self.word_groups(), (self.edges32, self.tets32)
        inside = sdf.detach().reshape(-1) > 0
        if groups is None:
            if self._order:  # the ordered pass: the same rule over the ranked lists and the sign plane in rank order
                t = self._order[1]
                groups, rows, inside = (t["edge_groups"], t["tet_groups"]), (t["edges_ranked"], t["tets_ranked"]), inside[t["vertex_of_rank"].long()]
            else:
                return None
        bits = _lib.lib().a3d_dmtet_word_group_bits()
        size = 1 << bits
        pad = (-inside.shape[0]) % size
        field = torch.cat([inside, inside.new_zeros(pad)]).reshape(-1, size)  # (the plane's padding bits are zeros)
        state = torch.where(field.all(1), 2, torch.where(field.any(1), 1, 0))  # 0: all outside, 2: all inside, 1: mixed
        out = []
        for rows32, tab in zip(rows, groups):
            t = tab.long()
            st = torch.where(t >= 0, state[t.clamp(min=0)], torch.ones_like(t))
            skip = (st == 0).all(1) | (st == 2).all(1)
            n_words = -(-rows32.shape[0] // 64)
            out += [int((~skip[:n_words]).sum()), n_words]
        return tuple(out)

    def uvs(self) -> torch.Tensor:
        """Per-tet uv quads [4N^2,2] (reference map_uv, dmtet.py:69-84); same torch expressions, cached."""
        if self._uvs is None:
            dev = self.tets32.device
            n = tetgrid.uv_grid_size(self.tets32.shape[0])
            lin = torch.linspace(0, 1 - (1 / n), n, dtype=torch.float32, device=dev)
            ty, tx = torch.meshgrid(lin, lin, indexing="ij")
            pad = 0.9 / n
            self._uvs = torch.stack([tx, ty, tx + pad, ty, tx + pad, ty + pad, tx, ty + pad], dim=-1).view(-1, 2)
        return self._uvs


def _interleave3(x: torch.Tensor) -> torch.Tensor:
    """Bits of x (int64, < 2^21) spread to every third position."""
    x = x & 0x1FFFFF
    x = (x | (x << 32)) & 0x1F00000000FFFF
    x = (x | (x << 16)) & 0x1F0000FF0000FF
    x = (x | (x << 8)) & 0x100F00F00F00F00F
    x = (x | (x << 4)) & 0x10C30C30C30C30C3
    x = (x | (x << 2)) & 0x1249249249249249
    return x  # (the y / z copies are shifted by the caller)


def _word_groups(rows32: torch.Tensor, slots: int, bits: int, block: int, chunk_words: int = 1 << 16) -> torch.Tensor:
    """See TetGridTopology.word_groups.  ``rows32`` int32 [N, C] (edges: C = 2, tets: C = 4)."""
    n, c = rows32.shape
    words_per_block = block // 64
    nw = -(-n // 64)
    nwp = -(-n // block) * words_per_block
    out = torch.full((nwp, slots), -1, dtype=torch.int32, device=rows32.device)
    for w0 in range(0, nw, chunk_words):  # bounded temporaries: [chunk, 64 C] sorted copies
        w1 = min(w0 + chunk_words, nw)
        g = rows32[w0 * 64:min(w1 * 64, n)] >> bits
        if g.shape[0] < (w1 - w0) * 64:  # the last word of the grid: padded with its own last row
            g = torch.cat([g, g[-1:].expand((w1 - w0) * 64 - g.shape[0], c)], 0)
        g = g.reshape(w1 - w0, 64 * c).sort(dim=1).values
        first = torch.ones_like(g, dtype=torch.bool)
        first[:, 1:] = g[:, 1:] != g[:, :-1]
        count = first.sum(1)
        ok = count <= slots
        rank = first.cumsum(1) - 1
        sel = first & ok[:, None]
        row = torch.arange(w1 - w0, device=g.device)[:, None].expand_as(g)
        blk = torch.full((w1 - w0, slots), -1, dtype=torch.int32, device=g.device)
        blk[row[sel], rank[sel]] = g[sel]
        fill = (torch.arange(slots, device=g.device)[None, :] >= count[:, None]) & ok[:, None]  # unused slots repeat the first group
        blk = torch.where(fill, blk[:, :1].expand_as(blk), blk)
        out[w0:w1] = blk
    return out.contiguous()


class DMTet:
    """Marching tetrahedra; batch size 1, like the reference (dmtet.py:20)."""

    def __init__(self, device=None):
        self.device = device if device is not None else "cuda"
        self._topo = None
        self._topo_key = None

    def to(self, device):
        self.device = device
        return self

    def topology(self, tet_fx4: torch.Tensor) -> TetGridTopology:
        key = (tet_fx4.data_ptr(), tuple(tet_fx4.shape), tet_fx4._version, str(tet_fx4.device))
        if self._topo_key != key:
            self._topo, self._topo_key, self._tets_ref = TetGridTopology(tet_fx4), key, tet_fx4
        return self._topo

    def sort_edges(self, edges_ex2):
        """(min, max) order per edge (reference dmtet.py:59-67; kept for API parity -- the HIP path never needs it)."""
        with torch.no_grad():
            order = (edges_ex2[:, 0] > edges_ex2[:, 1]).long().unsqueeze(dim=1)
            a = torch.gather(input=edges_ex2, index=order, dim=1)
            b = torch.gather(input=edges_ex2, index=1 - order, dim=1)
        return torch.stack([a, b], -1)

    def map_uv(self, faces, face_gidx, max_idx):
        """(uvs, uv_idx) of the reference's per-tet texture atlas (dmtet.py:69-98): an N x N grid of quads, N = ceil(sqrt((max_idx+1)//2)),
        tet t owns quad t, its first / second triangle use corners (0,1,2) / (0,2,3).  Kept for API parity: __call__ takes the same
        values from the per-grid cache (TetGridTopology.uvs) and from the emit kernel."""
        n = int(np.ceil(np.sqrt((max_idx + 1) // 2)))
        lin = torch.linspace(0, 1 - (1 / n), n, dtype=torch.float32, device=faces.device)
        ty, tx = torch.meshgrid(lin, lin, indexing="ij")
        pad = 0.9 / n
        uvs = torch.stack([tx, ty, tx + pad, ty, tx + pad, ty + pad, tx, ty + pad], dim=-1).view(-1, 2)
        quad = torch.div(face_gidx, 2, rounding_mode="trunc")
        second = face_gidx % 2
        uv_idx = torch.stack((quad * 4, quad * 4 + second + 1, quad * 4 + second + 2), dim=-1).view(-1, 3)
        return uvs, uv_idx

    def __call__(self, pos_nx3, sdf_n, tet_fx4, topology: TetGridTopology = None):
        topo = topology if topology is not None else self.topology(tet_fx4)
        verts, faces, uv_idx = ops.dmtet(pos_nx3, sdf_n, topo)
        return verts, faces, topo.uvs(), uv_idx


def sdf_bce_reg_loss(sdf, all_edges):
    """reference dmtet.py:161-169 (torch; off by default: sdf_bce_reg_loss_weight 0)."""
    pair = sdf[all_edges.reshape(-1)].reshape(-1, 2)
    mask = torch.sign(pair[..., 0]) != torch.sign(pair[..., 1])
    pair = pair[mask]
    bce = torch.nn.functional.binary_cross_entropy_with_logits
    return bce(pair[..., 0], (pair[..., 1] > 0).float()) + bce(pair[..., 1], (pair[..., 0] > 0).float())


GRAPH_SDF_GRADIENT = True  # replay the eikonal regulariser's launch-bound chain from HIP graphs (DMTetGeometry._graphed_sdf_gradient)


def _single_process():
    """Lazy graph capture inside a training step is allowed in single-process runs only: under multi-process DDP it would happen
    while RCCL's watchdog threads exist, a combination that cannot be exercised on the single-GPU boxes this round was developed
    on.  Multi-process runs capture up front (DMTetGeometry.capture_sdf_gradient_graph, before init_process_group) and only replay."""
    import torch.distributed as dist

    return not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)


def _sdf_gradient_callable(geometry, names):
    """(pts [N,3], *field parameters) -> d sdf / d pts [N,3], differentiable w.r.t. the parameters: the function the HIP graphs
    capture.  The parameters come in as explicit arguments (and are swapped into the field for the call) so that the captured
    backward ends at plain graph inputs rather than at the Parameters' own AccumulateGrad nodes, which live on the training stream."""
    from torch.nn.utils import stateless

    def fn(pts, *weights):
        pts = pts.detach().requires_grad_(True)
        with stateless._reparametrize_module(geometry.mlp, dict(zip(names, weights))):
            y = geometry.get_sdf(pts=pts)
        return torch.autograd.grad([y], pts, grad_outputs=torch.ones_like(y), create_graph=True, retain_graph=True, only_inputs=True)[0]

    return fn


class DMTetGeometry(torch.nn.Module):
    def __init__(self, grid_res, spatial_scale, num_layers=None, hidden_size=None, embedder_freq=None, embed_concat_pts=True,
                 init_sdf=None, jitter_grid=0.0, symmetrize=False, condition_choice=None, device="cuda", tets_dir="data/tets", tet_grid=None,
                 surface_only_backward=True, **kwargs):
        super().__init__()
        self.grid_res = grid_res
        self.marching_tets = DMTet()
        self._sdf_gradient_graphs = {}
        self.grid_scale = spatial_scale
        self.init_sdf = init_sdf
        self.jitter_grid = jitter_grid
        self.symmetrize = symmetrize
        self._device = device
        self._tets_dir = tets_dir
        self.surface_only_backward = surface_only_backward
        self._tet_grid = tet_grid  # optional (vertices [Nv,3] in (-0.5,0.5), indices [Nt,4]) instead of an npz file
        self.load_tets(self.grid_res, self.grid_scale)
        embedder_scalar = 2 * np.pi / self.grid_scale * 0.9  # (-0.5 s, 0.5 s) -> (-pi, pi) * 0.9  (dmtet.py:186)
        common = dict(dropout=0, activation=None, min_max=None, n_harmonic_functions=embedder_freq, embedder_scalar=embedder_scalar,
                      embed_concat_pts=embed_concat_pts)
        if condition_choice == "mod":  # pan-category model: SDF conditioned on the batch's class embedding (dmtet.py:187-189)
            self.mlp = CoordMLP_Mod(3, 1, num_layers, nf=hidden_size, condition_dim=128, **common)
        else:
            self.mlp = CoordMLP(3, 1, num_layers, nf=hidden_size, **common)

    # ---- grid -------------------------------------------------------------------------------------
    def load_tets(self, grid_res=None, scale=None):
        """reference dmtet.py:214-226.  Reads data/tets/{res}_tets.npz when present; the reference downloads those
        files (Quartet grids) and they cannot be fetched here, so a Kuhn grid of res/2 cells per axis -- about the
        same vertex count as the Quartet '{res}' grid -- is generated instead, with a warning."""
        self.grid_res = grid_res if grid_res is not None else self.grid_res
        self.grid_scale = scale if scale is not None else self.grid_scale
        path = os.path.join(self._tets_dir, f"{self.grid_res}_tets.npz")
        if self._tet_grid is not None:
            vertices, indices = self._tet_grid
        elif os.path.exists(path):
            vertices, indices = tetgrid.load_tets_npz(path)
        else:
            warnings.warn(f"{path} not found: generating a Kuhn tet grid with {max(self.grid_res // 2, 1)} cells per axis")
            vertices, indices = tetgrid.kuhn_grid(max(self.grid_res // 2, 1))
        self.verts = torch.tensor(vertices, dtype=torch.float32, device=self._device) * self.grid_scale
        self.indices = torch.tensor(indices, dtype=torch.long, device=self._device)
        self.generate_edges()

    def generate_edges(self):
        with torch.no_grad():
            self.topology = TetGridTopology(self.indices, positions=self.verts)
            self.all_edges = self.topology.all_edges

    @torch.no_grad()
    def getAABB(self):
        return torch.min(self.verts, dim=0).values, torch.max(self.verts, dim=0).values

    # ---- SDF field (PyTorch, unchanged semantics: reference dmtet.py:228-281) --------------------
    def get_sdf(self, pts=None, total_iter=0, feats=None):
        if pts is None:
            pts = self.verts
        if self.symmetrize:
            pts = torch.cat([pts[..., :1].abs(), pts[..., 1:]], -1)
        if feats is not None:
            # the reference repeats the embedding for every point (dmtet.py:231-233); the weight-modulated field only ever reads the
            # first row (Linear_Mod, MLPs.py:233-235), so one row gives the same values without a [Nv,128] x [128,256] style GEMM
            feats = feats.unsqueeze(0) if hasattr(self.mlp, "style_mlp") else feats.unsqueeze(0).repeat(pts.shape[0], 1)
        sdf = self.mlp(pts, feat=feats)
        if self.init_sdf is None:
            pass
        elif type(self.init_sdf) in [float, int]:
            sdf = sdf + self.init_sdf
        elif self.init_sdf == "sphere":
            sdf = sdf + (self.grid_scale * 0.25 - pts.norm(dim=-1, keepdim=True))
        elif self.init_sdf == "ellipsoid":
            xs, ys, zs = pts.unbind(-1)
            sdf = sdf + (self.grid_scale * 0.15 - torch.stack([xs, ys, zs / 2], -1).norm(dim=-1, keepdim=True))
        else:
            raise NotImplementedError
        return sdf

    def get_sdf_gradient(self, feats=None):
        num_samples = 5000
        pts = (torch.rand(num_samples, 3, device=self.verts.device) - 0.5) * self.grid_scale
        mv = self.mesh_verts.detach() + (torch.rand_like(self.mesh_verts) - 0.5) * 0.1 * self.grid_scale
        mv = mv[torch.randperm(len(mv), device=mv.device)[:5000]]
        pts = torch.cat([pts, mv], 0)
        if (GRAPH_SDF_GRADIENT and feats is None and pts.is_cuda and torch.is_grad_enabled() and self._graph_safe(pts.shape[0])
                and ((pts.shape[0], pts.device) in self._sdf_gradient_graphs or _single_process())
                and any(p.requires_grad for p in self.mlp.parameters())):
            try:
                return self._graphed_sdf_gradient(pts)
            except RuntimeError as err:  # capture refused (e.g. a field that synchronises inside forward): eager from now on
                warnings.warn(f"SDF-gradient graph capture failed ({err}); using the eager path")
                self._graph_capture_failed = True
        pts = pts.requires_grad_(True)
        y = self.get_sdf(pts=pts, feats=feats)
        try:
            return torch.autograd.grad([y], pts, grad_outputs=torch.ones_like(y), create_graph=True, retain_graph=True, only_inputs=True)[0]
        except RuntimeError:  # validation runs under no_grad
            return torch.zeros_like(pts)

    def _graph_safe(self, num_points):
        """Replay from HIP graphs only where that is safe and pays: a field whose embedding keeps its frequency table on the device
        (the reference's HarmonicEmbedding copies it host -> device inside every forward, HarmonicEmbedding.py:41 -- a blocking
        copy that cannot be captured), the full-size sample (5000 + 5000 points: a surface with fewer vertices changes the count
        from step to step, and every new count would re-capture and pin another graph memory pool), no earlier capture failure."""
        from ... import hostnets

        embedder = getattr(self.mlp, "embedder", None)
        known = any(k[0] == num_points for k in self._sdf_gradient_graphs)  # captured up front by capture_sdf_gradient_graph()
        return (not getattr(self, "_graph_capture_failed", False) and not hostnets.REFERENCE_FORMULATION and (num_points == 10000 or known)
                and (embedder is None or hasattr(embedder, "_frequencies")))

    def capture_sdf_gradient_graph(self, num_points=None):
        """Capture the regulariser's HIP graphs now (e.g. before torch.distributed is initialised: a multi-process run only replays
        graphs that already exist, it never captures inside a step).  ``num_points`` defaults to what get_sdf_gradient will sample
        for the current mesh (5000 random + up to 5000 surface points).  No-op off the GPU."""
        if num_points is None:
            mv = getattr(self, "mesh_verts", None)
            num_points = 5000 + (min(5000, len(mv)) if mv is not None else 5000)
        if (GRAPH_SDF_GRADIENT and self.verts.is_cuda and self._graph_safe(10000) and any(p.requires_grad for p in self.mlp.parameters())
                and not hasattr(self.mlp, "style_mlp")):  # (an explicit capture may ask for any point count; the conditioned field runs eagerly)
            with torch.enable_grad():
                self._graphed_sdf_gradient((torch.rand(num_points, 3, device=self.verts.device) - 0.5) * self.grid_scale)

    def _graphed_sdf_gradient(self, pts):
        """d sdf / d pts with its (double) backward replayed from two HIP graphs.

        The regulariser differentiates a 5-layer MLP twice over 1e4 points: ~300 launches of 5-15 us of GPU work each, i.e. the
        chain is bound by launch latency, not by the GPU.  The point count is static, so the forward (field + autograd.grad) and the
        backward (gradient of that w.r.t. the MLP parameters) are captured once per point count by torch.cuda.make_graphed_callables
        and replayed as two graph launches.  Same kernels, same order, same values.
        """
        key = (pts.shape[0], pts.device)
        names = [n for n, _ in self.mlp.named_parameters()]
        params = [p for _, p in self.mlp.named_parameters()]
        graphed = self._sdf_gradient_graphs.get(key)
        if graphed is None:
            sample = (pts.detach().clone(),) + tuple(p.detach().clone().requires_grad_(p.requires_grad) for p in params)
            while len(self._sdf_gradient_graphs) >= 2:  # bounded: each entry pins a private graph memory pool
                self._sdf_gradient_graphs.pop(next(iter(self._sdf_gradient_graphs)))
            graphed = self._sdf_gradient_graphs[key] = torch.cuda.make_graphed_callables(_sdf_gradient_callable(self, names), sample,
                                                                                          allow_unused_input=True)
        return graphed(pts.detach(), *params)

    def get_sdf_reg_loss(self, feats=None):
        return {"sdf_bce_reg_loss": sdf_bce_reg_loss(self.current_sdf, self.all_edges).mean(),
                "sdf_gradient_reg_loss": ((self.get_sdf_gradient(feats=feats).norm(dim=-1) - 1) ** 2).mean()}

    def _get_mesh_surface_backward(self, pos, total_iter, feats):
        """Same values and gradients as the plain path, ~2/3 less MLP work.

        Only grid vertices at the ends of sign-crossing edges (a few thousand of the ~3e5) ever receive a gradient from the
        mesh (dmtet.py:123-131); every other SDF value only decides a sign.  So the field is evaluated on the whole grid WITHOUT
        a graph, the surface is extracted, and the graph is built by re-evaluating the field on the surface-adjacent vertices
        alone.  The re-evaluated values enter as (x - x.detach()), i.e. exactly zero in forward, so the vertex positions are
        bit-identical to the single-pass result while autograd sees the dependency.
        """
        with torch.no_grad():
            sdf0 = self.get_sdf(pos, total_iter=total_iter, feats=feats)
        # idx = the grid vertices at the ends of crossing edges, sorted and unique; their count arrives with the DMTet counts (the
        # mask + torch.nonzero this replaces was a second host synchronisation per step)
        fused = pos.is_cuda and pos.dtype == torch.float32 and sdf0.dim() == 2 and sdf0.shape[1] == 1
        if fused and SURFACE_BUCKET and SURFACE_POINTS_IN_EMIT and pos.is_contiguous():
            # (round 6) the emit launch itself leaves the positions as the bucket-padded block (zero rows behind, so that the MLP's GEMM
            # shapes repeat from step to step), and one node splices the re-evaluated values in: forward no launch at all, backward one gather
            verts0, faces, uv_idx, vert_edge, idx, pts = ops.dmtet_extract(pos, sdf0, self.topology, surface_vertices=True, for_backward=True,
                                                                           surface_points=SURFACE_BUCKET)
            sdf_sub = self.get_sdf(pts, total_iter=total_iter, feats=feats)
            self.current_sdf = ops.surface_sdf(sdf0, idx, sdf_sub)
            return ops.dmtet_verts(pos, self.current_sdf, verts0, vert_edge, self.topology), faces, uv_idx
        verts0, faces, uv_idx, vert_edge, idx = ops.dmtet_extract(pos, sdf0, self.topology, surface_vertices=True, for_backward=True)
        n_pad = (-idx.shape[0]) % SURFACE_BUCKET if SURFACE_BUCKET else 0
        if fused:
            pts = ops.gather_rows_padded(pos, idx, idx.shape[0] + n_pad)
            sdf_sub = self.get_sdf(pts, total_iter=total_iter, feats=feats)
            self.current_sdf = ops.surface_sdf(sdf0, idx, sdf_sub)
        else:
            pts = pos[idx]
            if n_pad:
                pts = torch.nn.functional.pad(pts, (0, 0, 0, n_pad))
            sdf_sub = self.get_sdf(pts, total_iter=total_iter, feats=feats)[: idx.shape[0]]
            self.current_sdf = sdf0.index_add(0, idx, sdf_sub - sdf_sub.detach())
        return ops.dmtet_verts(pos, self.current_sdf, verts0, vert_edge, self.topology), faces, uv_idx

    # ---- mesh extraction (reference dmtet.py:294-310) ---------------------------------------------
    def getMesh(self, material=None, total_iter=0, jitter_grid=True, feats=None):
        v_deformed = self.verts
        if jitter_grid and self.jitter_grid > 0:
            # U(-1, 1) * jitter_grid * grid_scale (dmtet.py:297-299) drawn in ONE launch instead of rand, * 2, - 1, * scale
            amp = float(self.jitter_grid) * float(self.grid_scale)
            jitter = torch.empty(1, device=v_deformed.device).uniform_(-amp, amp)
            v_deformed = v_deformed + jitter
        self.current_pos = v_deformed  # (extra attribute: the jittered grid this mesh was extracted on)
        if self.surface_only_backward and torch.is_grad_enabled() and any(p.requires_grad for p in self.mlp.parameters()):
            verts, faces, uv_idx = self._get_mesh_surface_backward(v_deformed, total_iter, feats)
            uvs = self.topology.uvs()
        else:
            self.current_sdf = self.get_sdf(v_deformed, total_iter=total_iter, feats=feats)
            verts, faces, uvs, uv_idx = self.marching_tets(v_deformed, self.current_sdf, self.indices, topology=self.topology)
        self.mesh_verts = verts
        return mesh.make_mesh(verts[None], faces[None], uvs[None], uv_idx[None], material)
