"""Tetrahedral background grids for DMTet.

The reference loads Quartet-generated grids from ``data/tets/{res}_tets.npz``
(keys ``vertices`` float [Nv,3] in (-0.5,0.5) and ``indices`` int [Nt,4];
``/root/reference/model/geometry/dmtet.py:214-226``,
``/root/reference/data/tets/generate_tets.py:31-47``).  Those files are
downloaded by the reference and cannot be fetched here, so this module also
provides a generator for a Kuhn (Freudenthal) 6-tet-per-cube grid that is
written in the very same npz format, plus the *static topology* every DMTet
call on a given grid re-uses:

* ``edges``    int32 [Ne,2]  lexicographically sorted unique (min,max) edges --
  the same list the reference builds in ``generate_edges``
  (``dmtet.py:283-288``);
* ``tet2edge`` int32 [Nt,6]  row index into ``edges`` for each of the six tet
  edges in the reference's slot order (v0v1, v0v2, v0v3, v1v2, v1v3, v2v3;
  ``dmtet.py:46``).

Because ``edges`` is sorted, the reference's vertex numbering (rank among the
``torch.unique``-sorted crossing edges, ``dmtet.py:115-121``) equals an
exclusive prefix sum of the crossing flag over ``edges`` -- which is what the
HIP kernels compute.
"""
from __future__ import annotations

import itertools
import os

import numpy as np

# slot order of the six tet edges (reference dmtet.py:46)
TET_EDGE_SLOTS = np.array([[0, 1], [0, 2], [0, 3], [1, 2], [1, 3], [2, 3]], dtype=np.int64)


def kuhn_grid(res: int):
    """Kuhn triangulation of the cube [-0.5,0.5]^3 split into res^3 cells.

    Returns (vertices float32 [(res+1)^3,3], indices int64 [6 res^3,4]).
    Vertex (i,j,k) has index (i*(res+1)+j)*(res+1)+k and position
    (i,j,k)/res-0.5.  All tets share one orientation, chosen so that the marching-tets table
    (dmtet.py:26-43) winds the surface with normals pointing from inside (sdf>0) to outside.
    """
    n = res + 1
    ax = np.arange(n, dtype=np.float64) / res - 0.5
    vx, vy, vz = np.meshgrid(ax, ax, ax, indexing="ij")
    vertices = np.stack([vx, vy, vz], -1).reshape(-1, 3).astype(np.float32)

    ci, cj, ck = np.meshgrid(np.arange(res), np.arange(res), np.arange(res), indexing="ij")
    base = np.stack([ci, cj, ck], -1).reshape(-1, 3)  # [res^3,3]

    def vid(p):
        return (p[:, 0] * n + p[:, 1]) * n + p[:, 2]

    tets = []
    for perm in itertools.permutations(range(3)):
        # walk 000 -> 111 adding one axis at a time in the order given by perm
        p = base.copy()
        chain = [vid(p)]
        for a in perm:
            p = p.copy()
            p[:, a] += 1
            chain.append(vid(p))
        t = np.stack(chain, -1)  # [res^3,4]
        # parity of the permutation decides orientation; give every tet the same (negative) one
        sign = np.linalg.det(np.eye(3)[list(perm)])
        if sign > 0:
            t = t[:, [0, 1, 3, 2]]
        tets.append(t)
    # interleave so the six tets of a cell are consecutive
    indices = np.stack(tets, 1).reshape(-1, 4).astype(np.int64)
    return vertices, indices


def bcc_grid(res: int, seed=None):
    """Body-centred-cubic tetrahedral grid of the cube [-0.5,0.5]^3 -- the lattice family Quartet (the generator of the reference's
    ``{res}_tets.npz`` files, data/tets/generate_tets.py:14-25) builds its meshes from: vertices = the (res+1)^3 cell corners + the
    res^3 cell centres; every interior cell face contributes four tets (the two centres on either side + one edge of the face).
    With ``seed`` the vertex numbering is a random permutation, the tet rows are shuffled and the four indices of every row are
    permuted -- the arbitrary numbering a file produced by an external mesher has.  -> (vertices float32 [Nv,3], indices int64 [Nt,4])."""
    n = res + 1
    ax = np.arange(n, dtype=np.float64) / res - 0.5
    corners = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    cx = (np.arange(res, dtype=np.float64) + 0.5) / res - 0.5
    centres = np.stack(np.meshgrid(cx, cx, cx, indexing="ij"), -1).reshape(-1, 3)
    vertices = np.concatenate([corners, centres]).astype(np.float32)
    # corner (i,j,k) -> (i n + j) n + k, centre (i,j,k) -> n^3 + (i res + j) res + k.  Rows in the order axis, i, j, k, ring edge
    # (vectorised over (i,j,k): the "256" class is 1.3e7 tets)
    gi, gj, gk = [a.reshape(-1) for a in np.meshgrid(np.arange(res - 1), np.arange(res), np.arange(res), indexing="ij")]
    per_axis = []
    for axis in range(3):
        u, v = [a for a in range(3) if a != axis]

        def cell(di, du, dv):  # integer coordinates with the face's (axis, u, v) components put in place
            q = np.zeros((gi.size, 3), dtype=np.int64)
            q[:, axis], q[:, u], q[:, v] = gi + di, gj + du, gk + dv
            return q

        lo, hi = cell(0, 0, 0), cell(1, 0, 0)  # the face between cell i and cell i + 1 along ``axis``
        c0 = n ** 3 + (lo[:, 0] * res + lo[:, 1]) * res + lo[:, 2]
        c1 = n ** 3 + (hi[:, 0] * res + hi[:, 1]) * res + hi[:, 2]
        ring = []
        for du, dv in ((0, 0), (1, 0), (1, 1), (0, 1)):  # the face's four corners in cyclic order
            q = cell(1, du, dv)
            ring.append((q[:, 0] * n + q[:, 1]) * n + q[:, 2])
        per_axis.append(np.stack([np.stack([c0, c1, ring[e], ring[(e + 1) % 4]], -1) for e in range(4)], 1).reshape(-1, 4))
    indices = np.concatenate(per_axis).astype(np.int64)
    if seed is not None:
        vertices, indices = scramble(vertices, indices, seed)
    return vertices, indices


def delaunay_grid(num_points: int, seed: int = 0):
    """Delaunay tetrahedralisation of ``num_points`` random points in [-0.5,0.5]^3 (scipy / Qhull), scrambled like an external
    mesher's output: an irregular grid with varying vertex valence, the opposite extreme of the Kuhn grid's one repeated cell."""
    from scipy.spatial import Delaunay

    rng = np.random.default_rng(seed)
    pts = rng.uniform(-0.5, 0.5, size=(num_points, 3))
    tets = Delaunay(pts).simplices.astype(np.int64)
    vol = np.abs(np.einsum("ij,ij->i", np.cross(pts[tets[:, 1]] - pts[tets[:, 0]], pts[tets[:, 2]] - pts[tets[:, 0]]), pts[tets[:, 3]] - pts[tets[:, 0]]))
    tets = tets[vol > 1e-12]  # Qhull may emit flat slivers on the hull
    return scramble(pts.astype(np.float32), tets, seed + 1)


def scramble(vertices: np.ndarray, indices: np.ndarray, seed: int, renumber: bool = True, shuffle_rows: bool = True):
    """Renumber the vertices by a random permutation, shuffle the tet rows and permute the four indices inside every row.
    (``renumber`` / ``shuffle_rows`` False: that step left out -- the same random draws either way -- to tell their effects apart.)"""
    rng = np.random.default_rng(seed)
    perm = rng.permutation(vertices.shape[0])  # new id of old vertex v = perm[v]
    if not renumber:
        perm = np.arange(vertices.shape[0])
    out_v = np.empty_like(vertices)
    out_v[perm] = vertices
    idx = perm[indices]
    rows = rng.permutation(idx.shape[0])
    if shuffle_rows:
        idx = idx[rows]
    order = np.argsort(rng.random(idx.shape), axis=1)
    return out_v, np.take_along_axis(idx, order, axis=1).astype(np.int64)


def named_grid(name: str):
    """'kuhnR' / 'bccR' / 'delaunayN', with a trailing 's' for a scrambled numbering (seed 7): (vertices, indices, cells per axis).
    bcc51s ~ the vertex / tet counts of the reference's "128" Quartet grid (2.7e5 / 1.6e6), bcc102s ~ its "256" grid (2.2e6 / 1.3e7),
    in the arbitrary numbering such a file has."""
    import re

    m = re.fullmatch(r"(kuhn|bcc|delaunay)(\d+)([svt]?)", name)
    if not m:
        raise ValueError(f"unknown grid {name!r} (kuhnR, bccR, delaunayN, optional trailing s = scrambled numbering; v / t = only the "
                         "vertex numbering / only the tet rows scrambled)")
    kind, res, scr = m.group(1), int(m.group(2)), m.group(3)
    if kind == "delaunay":
        return (*delaunay_grid(res, seed=7), max(2, round(res ** (1 / 3))))
    p, t = kuhn_grid(res) if kind == "kuhn" else bcc_grid(res)
    if scr:
        p, t = scramble(p, t, 7, renumber=scr in "sv", shuffle_rows=scr in "st")
    return p, t, res


def save_tets_npz(path: str, vertices: np.ndarray, indices: np.ndarray) -> None:
    """Write a grid in the reference's ``{res}_tets.npz`` format."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.savez_compressed(path, vertices=vertices, indices=indices)


def load_tets_npz(path: str):
    z = np.load(path)
    return np.asarray(z["vertices"], dtype=np.float32), np.asarray(z["indices"], dtype=np.int64)


def build_topology(indices: np.ndarray):
    """Static per-grid topology: (edges int32 [Ne,2], tet2edge int32 [Nt,6]).

    ``edges`` reproduces ``torch.unique(sort(indices[:, slots]), dim=0)`` of the
    reference (``dmtet.py:283-288``): ascending lexicographic (min,max).
    """
    indices = np.asarray(indices, dtype=np.int64)
    nt = indices.shape[0]
    pairs = indices[:, TET_EDGE_SLOTS]  # [Nt,6,2]
    lo = pairs.min(-1).reshape(-1)
    hi = pairs.max(-1).reshape(-1)
    nv = int(indices.max()) + 1 if nt else 0
    key = lo * nv + hi
    uniq, inverse = np.unique(key, return_inverse=True)
    edges = np.stack([uniq // nv, uniq % nv], -1).astype(np.int32)
    tet2edge = inverse.reshape(nt, 6).astype(np.int32)
    return edges, tet2edge


def uv_grid_size(num_tets: int) -> int:
    """N of the reference's per-tet uv atlas: ceil(sqrt((2*Nt+1)//2)) (dmtet.py:70,153)."""
    return int(np.ceil(np.sqrt((2 * num_tets + 1) // 2)))
