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
