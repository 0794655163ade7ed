"""GPU parity tests: the HIP path (through the C ABI) against the golden vectors and the CPU oracle.

Run on the MI355X box:  python -m pytest tests -m gpu -x -q
Bars: index buffers and triangle ids bit-exact; floating point within the tolerance written at each assert
(north_star: rendered buffers within 1e-4 absolute).
"""
import glob
import importlib
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden, kuhn, seeded

pytestmark = pytest.mark.gpu

DMTET_CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "dmtet_*_r[0-9]*.npz")))  # Kuhn grids (regenerated from res); the irregular-grid fixtures have their own tests


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    return importlib.import_module("3danimals_amd.ops")


@pytest.fixture(scope="module")
def mods():
    m = lambda n: importlib.import_module("3danimals_amd." + n)
    return dict(dmtet=m("model.geometry.dmtet"), skinning=m("model.geometry.skinning"), mesh=m("model.render.mesh"),
                render=m("model.render.render"), light=m("model.render.light"), nets=m("hostnets"), synthetic=m("synthetic"))


def quadruped_mesh(res=16, leg_radius=0.3):
    from oracle import dmtet_ref

    synthetic = importlib.import_module("3danimals_amd.synthetic")
    pos, tets = kuhn(res)
    verts, faces, _, _ = dmtet_ref.marching_tets(pos, synthetic.quadruped_sdf(pos, leg_radius=leg_radius), tets)
    return verts, faces


# ------------------------------------------------------------------------------------------------ DMTet
@pytest.mark.parametrize("name", DMTET_CASES)
def test_dmtet_matches_reference_golden(name, dev, mods):
    g = golden(name)
    pos, tets = kuhn(int(g["res"]))
    sdf = torch.from_numpy(g["sdf"]).to(dev).requires_grad_(True)
    dm = mods["dmtet"].DMTet()
    verts, faces, uvs, uv_idx = dm(pos.to(dev), sdf[:, None], tets.to(dev))
    assert faces.dtype == torch.int64 and uv_idx.dtype == torch.int64
    assert np.array_equal(faces.cpu().numpy(), g["faces"])  # bit-exact
    assert np.array_equal(uv_idx.cpu().numpy(), g["uv_idx"])
    assert np.array_equal(verts.detach().cpu().numpy(), g["verts"])  # same rounding sequence as the reference
    assert tuple(uvs.shape) == tuple(g["uvs_shape"])
    assert np.array_equal(uvs[:64].cpu().numpy(), g["uvs_head"]) and np.array_equal(uvs[-64:].cpu().numpy(), g["uvs_tail"])
    if verts.numel():
        wgt = seeded(verts.shape, int(g["grad_wgt_seed"]), -1, 1).to(dev)
        (gs,) = torch.autograd.grad((verts * wgt).sum(), sdf)
        np.testing.assert_allclose(gs.cpu().numpy(), g["grad_sdf"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", ["dmtet_bcc10_sphere.npz", "dmtet_bcc10_random.npz", "dmtet_delaunay3k_sphere.npz", "dmtet_delaunay3k_random.npz"])
def test_dmtet_on_irregular_grids_matches_reference_golden(name, tmp_path, dev, mods):
    """The grid class the reference really trains on (Quartet files, dmtet.py:214-226): arbitrary vertex numbering, unordered rows,
    mixed orientation.  DMTet.__call__ on the HIP path against the reference's own outputs, bit for bit, plus d/dsdf; and the same grid
    written as a ``{res}_tets.npz`` file and loaded through DMTetGeometry.load_tets gives the same extraction."""
    g = golden(name)
    pos, tets = torch.from_numpy(g["pos"]).to(dev), torch.from_numpy(g["tets"]).long().to(dev)
    sdf = torch.from_numpy(g["sdf"]).to(dev).requires_grad_(True)
    dm = mods["dmtet"].DMTet()
    verts, faces, uvs, uv_idx = dm(pos, sdf[:, None], tets)
    assert np.array_equal(faces.cpu().numpy(), g["faces"]) and np.array_equal(uv_idx.cpu().numpy(), g["uv_idx"])
    assert np.array_equal(verts.detach().cpu().numpy(), g["verts"]) and tuple(uvs.shape) == tuple(g["uvs_shape"])
    (gs,) = torch.autograd.grad((verts * seeded(verts.shape, int(g["grad_wgt_seed"]), -1, 1).to(dev)).sum(), sdf)
    np.testing.assert_allclose(gs.cpu().numpy(), g["grad_sdf"], rtol=1e-4, atol=1e-5)
    # through the file format and the geometry class
    a3d_pkg = importlib.import_module("3danimals_amd")
    a3d_pkg.tetgrid.save_tets_npz(str(tmp_path / "77_tets.npz"), g["pos"] / 7.0, g["tets"].astype(np.int64))
    geo = mods["dmtet"].DMTetGeometry(77, 7.0, num_layers=2, hidden_size=16, embedder_freq=2, device=dev, tets_dir=str(tmp_path)).to(dev)
    assert torch.equal(geo.indices, tets) and torch.allclose(geo.verts, pos, atol=1e-6)
    v2, f2, _, u2 = geo.marching_tets(geo.verts, sdf.detach()[:, None], geo.indices, topology=geo.topology)
    if torch.equal(geo.verts, pos):  # (x / 7) * 7 is not always x in float32: vertices only compare when the round trip is exact
        assert torch.equal(v2, verts.detach())
    assert torch.equal(f2, faces) and torch.equal(u2, uv_idx)


def _extract_all(ops, pos, sdf, topo):
    v, f, u, ve, idx = ops.dmtet_extract(pos, sdf, topo, surface_vertices=True)
    return [x.cpu() for x in (v, f, u, ve, idx)]


@pytest.mark.parametrize("grid", ["kuhn7", "kuhn24", "kuhn40", "kuhn64", "bcc", "delaunay", "kuhn20s", "bcc14g"])
def test_culled_dmtet_count_equals_the_plain_one(grid, dev, ops, mods, monkeypatch):
    """a3d_dmtet_count with the per-grid word groups (sign-plane pre-pass, words whose vertex groups all lie on one side of the surface
    are never read) against the same entry without them: every output of the extraction bit for bit -- smooth surfaces (most words
    skipped), noise (none skipped), and the cases a wrong skip would lose: ONE inside vertex, at group and grid boundaries; one
    outside vertex in a full grid; nothing inside; everything inside."""
    a3d_pkg = importlib.import_module("3danimals_amd")
    monkeypatch.setattr(ops, "DMTET_CULL_MIN_VERTS", 0)  # (small grids normally keep the plain pass)
    if grid == "kuhn20s":  # a Kuhn grid in a random numbering
        p, t = a3d_pkg.tetgrid.scramble(*a3d_pkg.tetgrid.kuhn_grid(20), 11)
        pos, tets = torch.from_numpy(p), torch.from_numpy(t).long()
    elif grid.startswith("kuhn"):
        pos, tets = kuhn(int(grid[4:]))
    elif grid == "bcc14g":  # a BCC lattice in its GENERATOR's order: spatially coherent rows that touch 10-13 vertex groups -- 16 slots
        p, t = a3d_pkg.tetgrid.bcc_grid(14)
        pos, tets = torch.from_numpy(p), torch.from_numpy(t).long()
    elif grid == "bcc":
        p, t = a3d_pkg.tetgrid.bcc_grid(9, seed=2)
        pos, tets = torch.from_numpy(p), torch.from_numpy(t).long()
    else:
        p, t = a3d_pkg.tetgrid.delaunay_grid(900, seed=4)
        pos, tets = torch.from_numpy(p), torch.from_numpy(t).long()
    pos = pos.to(dev)
    T = mods["dmtet"].TetGridTopology
    # culled: what the grid picks by itself (its own row order where that is spatial, else the ranked lists); ordered: the ranked lists
    # forced (a3d_dmtet_count_ordered, also on the Kuhn grids); plain: no tables at all
    culled, ordered, plain = T(tets.to(dev)), T(tets.to(dev), positions=pos), T(tets.to(dev))
    plain.WORD_GROUPS = plain.SPATIAL_ORDER = ordered.WORD_GROUPS = False
    assert plain.word_groups() is None and plain.count_pass(positions=pos) == "plain" and ordered.count_pass() == "ordered"
    spatial = (grid.startswith("kuhn") and not grid.endswith("s")) or grid == "bcc14g"
    assert culled.count_pass(positions=pos) == ("culled" if spatial else "ordered")
    if spatial:
        assert culled.word_groups()[0].shape[1] == (16 if grid == "bcc14g" else 8)
    eo, to = (ordered.spatial_order()[1][k] for k in ("edge_groups", "tet_groups"))
    assert (eo[:-16, 0] >= 0).float().mean() > 0.9 and (to[:-16, 0] >= 0).float().mean() > 0.9  # 16 slots hold (nearly) every word of the ranked lists
    if spatial and grid != "bcc14g":
        e, t = culled.word_groups()  # the cull is live on these grids: (nearly) every word that holds rows has <= 8 groups
        assert (e[:-16, 0] >= 0).float().mean() > 0.95 and (t[:-16, 0] >= 0).float().mean() > 0.95
    Nv = pos.shape[0]
    g = torch.Generator().manual_seed(Nv)
    centre = pos.mean(0)
    sdfs = {"sphere": 0.3 * (pos.amax(0) - pos.amin(0)).min() - (pos - centre).norm(dim=1),
            "quadruped": mods["synthetic"].quadruped_sdf(pos.cpu(), 0.2, noise=0.01, seed=3).to(dev) if grid.startswith("kuhn") else None,
            "noise": torch.randn(Nv, generator=g).to(dev), "none": -torch.ones(Nv, device=dev), "all": torch.ones(Nv, device=dev),
            "zeros": torch.zeros(Nv, device=dev)}
    for v in (0, 15, 16, 17, Nv // 2, Nv - 17, Nv - 16, Nv - 1):
        one = -torch.ones(Nv, device=dev)
        one[v] = 1.0
        sdfs[f"one_in_{v}"] = one
        sdfs[f"one_out_{v}"] = -one
    # the 8-slot form of the ordered pass (its other template instance; words with more than 8 groups -- most, on these grids -- are read)
    ordered8 = T(tets.to(dev), positions=pos)
    ordered8.WORD_GROUPS, ordered8.ORDER_SLOTS, ordered8.ORDER_MAX_DENSE = False, 8, 1.0
    assert ordered8.count_pass() == "ordered" and ordered8.spatial_order()[0].group_slots == 8
    for name, sdf in sdfs.items():
        if sdf is None:
            continue
        a, b, c = _extract_all(ops, pos, sdf, culled), _extract_all(ops, pos, sdf, plain), _extract_all(ops, pos, sdf, ordered)
        for x, z in zip(a, _extract_all(ops, pos, sdf, ordered8)):
            assert x.shape == z.shape and torch.equal(x, z), (grid, name, "ordered, 8 slots")
        assert (culled._last_count_pass, plain._last_count_pass, ordered._last_count_pass) == (culled.count_pass(), "plain", "ordered")
        for x, y, z in zip(a, b, c):
            assert x.shape == y.shape and torch.equal(x, y), (grid, name)
            assert x.shape == z.shape and torch.equal(x, z), (grid, name, "ordered")
        if name.startswith("one_in") and bool((tets == int(name.rsplit("_", 1)[1])).any()):  # (the cube's corners belong to no tet of a BCC lattice)
            assert a[0].shape[0] > 0, (grid, name)


@pytest.mark.parametrize("res,kind", [(24, "random"), (32, "quadruped"), (64, "quadruped")])
def test_dmtet_matches_oracle_larger(res, kind, dev, mods):
    from oracle import dmtet_ref

    pos, tets = kuhn(res)
    g = torch.Generator().manual_seed(res)
    sdf = torch.randn(pos.shape[0], generator=g) if kind == "random" else mods["synthetic"].quadruped_sdf(pos, 0.2, noise=0.01, seed=res)
    rv, rf, _, ru = dmtet_ref.marching_tets(pos, sdf, tets)
    sdf_d = sdf.to(dev).requires_grad_(True)
    pos_d = pos.to(dev).requires_grad_(True)
    verts, faces, _, uv_idx = mods["dmtet"].DMTet()(pos_d, sdf_d, tets.to(dev))
    assert np.array_equal(faces.cpu().numpy(), rf.numpy()) and np.array_equal(uv_idx.cpu().numpy(), ru.numpy())
    assert np.array_equal(verts.detach().cpu().numpy(), rv.numpy())
    # gradients w.r.t. sdf AND pos against autograd through the oracle
    sdf_c, pos_c = sdf.clone().requires_grad_(True), pos.clone().requires_grad_(True)
    rv2, _, _, _ = dmtet_ref.marching_tets(pos_c, sdf_c, tets)
    wgt = seeded(rv.shape, 5, -1, 1)
    gs_ref, gp_ref = torch.autograd.grad((rv2 * wgt).sum(), [sdf_c, pos_c])
    gs, gp = torch.autograd.grad((verts * wgt.to(dev)).sum(), [sdf_d, pos_d])
    np.testing.assert_allclose(gs.cpu().numpy(), gs_ref.numpy(), rtol=2e-4, atol=2e-4 * float(gs_ref.abs().max()))
    np.testing.assert_allclose(gp.cpu().numpy(), gp_ref.numpy(), rtol=2e-4, atol=1e-5)


@pytest.mark.parametrize("grid,kind,want", [("bcc51s", "quadruped", "ordered"), ("bcc51s", "noise", "ordered"), ("bcc102s", "quadruped", "ordered"),
                                            ("kuhn128", "quadruped", "culled")])
def test_dmtet_on_the_reference_grid_classes_at_real_size_matches_the_oracle(grid, kind, want, dev, mods):
    """DMTet.__call__ against oracle/dmtet_ref (sort + unique, the reference's own formulation: dmtet.py:104-155) on grids of the size and
    KIND the reference trains on -- data/tets/{128,256}_tets.npz are Quartet BCC-derived meshes in an external mesher's numbering
    (dmtet.py:214-226, generate_tets.py:31-47): bcc51s = 2.7e5 vertices / 1.6e6 tets, bcc102s = 2.2e6 / 1.3e7, randomly numbered,
    rows shuffled, row entries permuted -- and on the Kuhn R = 128 grid: faces / uv_idx / verts bit for bit, d/dsdf and d/dpos against
    autograd through the oracle; the count pass that ran is asserted ('ordered' = a3d_dmtet_count_ordered + a3d_dmtet_emit_sparse)."""
    from oracle import dmtet_ref

    tg = importlib.import_module("3danimals_amd.tetgrid")
    p, t, _ = tg.named_grid(grid)
    pos, tets = torch.from_numpy(p), torch.from_numpy(t).long()
    scale = 7.0 / float((pos.amax(0) - pos.amin(0)).max())
    if kind == "quadruped":
        sdf = mods["synthetic"].quadruped_sdf(pos * scale, 0.2, noise=0.002, seed=3)
    else:  # white noise: half the edges cross (nothing to cull; V ~ 9e5, F ~ 2e6)
        sdf = torch.randn(pos.shape[0], generator=torch.Generator().manual_seed(11))
    rv, rf, _, ru = dmtet_ref.marching_tets(pos, sdf, tets)
    assert rf.shape[0] > 8000
    sdf_d, pos_d = sdf.to(dev).requires_grad_(True), pos.to(dev).requires_grad_(True)
    dm = mods["dmtet"].DMTet()
    tets_d = tets.to(dev)
    for it in range(2):  # (the second extraction takes the speculative emit: sizes guessed from the first)
        verts, faces, _, uv_idx = dm(pos_d, sdf_d, tets_d)
        assert dm.topology(tets_d)._last_count_pass == want
        assert np.array_equal(faces.cpu().numpy(), rf.numpy()) and np.array_equal(uv_idx.cpu().numpy(), ru.numpy()), it
        assert np.array_equal(verts.detach().cpu().numpy(), rv.numpy()), it
    sdf_c, pos_c = sdf.clone().requires_grad_(True), pos.clone().requires_grad_(True)
    rv2, _, _, _ = dmtet_ref.marching_tets(pos_c, sdf_c, tets)
    wgt = seeded(rv.shape, 5, -1, 1)
    gs_ref, gp_ref = torch.autograd.grad((rv2 * wgt).sum(), [sdf_c, pos_c])
    gs, gp = torch.autograd.grad((verts * wgt.to(dev)).sum(), [sdf_d, pos_d])
    np.testing.assert_allclose(gs.cpu().numpy(), gs_ref.numpy(), rtol=2e-4, atol=2e-4 * float(gs_ref.abs().max()))
    np.testing.assert_allclose(gp.cpu().numpy(), gp_ref.numpy(), rtol=2e-4, atol=1e-5)


@pytest.mark.parametrize("res,kind", [(12, "random"), (32, "quadruped"), (64, "quadruped"), (16, "empty")])
def test_dmtet_surface_adjacent_vertices_ride_in_the_count(res, kind, dev, mods, ops):
    """The sorted list of grid vertices at the ends of crossing edges (what DMTetGeometry re-evaluates the SDF network on) against its
    plain definition, and the other outputs unchanged by asking for it."""
    pos, tets = kuhn(res)
    g = torch.Generator().manual_seed(res + 1)
    if kind == "random":
        sdf = torch.randn(pos.shape[0], generator=g)
    elif kind == "empty":
        sdf = torch.ones(pos.shape[0])
    else:
        sdf = mods["synthetic"].quadruped_sdf(pos, 0.2, noise=0.01, seed=res)
    topo = mods["dmtet"].TetGridTopology(tets.to(dev))
    pos_d, sdf_d = pos.to(dev), sdf.to(dev)
    v0, f0, u0, e0 = ops.dmtet_extract(pos_d, sdf_d, topo)
    v1, f1, u1, e1, idx = ops.dmtet_extract(pos_d, sdf_d, topo, surface_vertices=True)
    assert torch.equal(v0, v1) and torch.equal(f0, f1) and torch.equal(u0, u1) and torch.equal(e0, e1)
    want = torch.unique(topo.edges32[e0.long()].reshape(-1).long())  # sorted
    assert idx.dtype == torch.int64 and torch.equal(idx, want)
    assert (kind == "empty") == (idx.numel() == 0)


def test_dmtet_full_size_properties(dev, mods):
    """BASELINE-size grid (Kuhn R=64): size-independent invariants + run-to-run determinism."""
    pos, tets = kuhn(64)
    sdf = mods["synthetic"].quadruped_sdf(pos, 0.2, noise=0.01, seed=1).to(dev)
    dm = mods["dmtet"].DMTet()
    pos_d, tets_d = pos.to(dev), tets.to(dev)
    v1, f1, _, u1 = dm(pos_d, sdf, tets_d)
    v2, f2, _, u2 = dm(pos_d, sdf, tets_d)
    assert torch.equal(v1, v2) and torch.equal(f1, f2) and torch.equal(u1, u2)
    V, F = v1.shape[0], f1.shape[0]
    assert F > 10000 and int(f1.min()) == 0 and int(f1.max()) == V - 1
    assert len(torch.unique(f1)) == V  # every vertex referenced
    # closed 2-manifold: each undirected edge shared by exactly two faces, traversed once in each direction
    e = torch.cat([f1[:, [0, 1]], f1[:, [1, 2]], f1[:, [2, 0]]], 0)
    key = e[:, 0] * V + e[:, 1]
    assert len(torch.unique(key)) == key.numel()  # no directed edge twice -> consistent winding
    und = torch.minimum(e[:, 0], e[:, 1]) * V + torch.maximum(e[:, 0], e[:, 1])
    _, cnt = torch.unique(und, return_counts=True)
    assert bool((cnt == 2).all())
    # uv_idx rows: (4t, 4t+k+1, 4t+k+2)
    assert bool(((u1[:, 0] % 4) == 0).all()) and bool(((u1[:, 2] - u1[:, 1]) == 1).all())
    # vertices sit strictly inside the grid's bounding box
    assert float(v1.abs().max()) <= 3.5


# ------------------------------------------------------------------------------------------------ normals / mesh
@pytest.mark.parametrize("B", [1, 4])
def test_normals_match_reference_golden(B, dev, mods):
    g = golden(f"mesh_b{B}.npz")
    v = torch.from_numpy(g["v_pos"]).to(dev).requires_grad_(True)
    faces, uv_idx = torch.from_numpy(g["faces"]).to(dev), torch.from_numpy(g["uv_idx"]).to(dev)
    uvs = mods["dmtet"].TetGridTopology(kuhn(8)[1].to(dev)).uvs()
    m = mods["mesh"].make_mesh(v, faces[None], uvs[None].expand(B, -1, -1), uv_idx[None], None)
    np.testing.assert_allclose(m.v_nrm.detach().cpu().numpy(), g["v_nrm"], atol=2e-6)
    np.testing.assert_allclose(m.v_tng.detach().cpu().numpy(), g["v_tng"], atol=5e-5)  # lazy torch path
    wgt = seeded(m.v_nrm.shape, int(g["grad_wgt_seed"]), -1, 1).to(dev)
    (gv,) = torch.autograd.grad((m.v_nrm * wgt).sum(), v)
    np.testing.assert_allclose(gv.cpu().numpy(), g["grad_v"], rtol=1e-3, atol=2e-5)


def test_normals_adjacency_and_bit_reproducibility(dev, ops):
    """CSR vertex->(corner, face) lists in the reference's scatter_add_ order (mesh.py:291-293); no atomics -> identical bits per run."""
    g = golden("mesh_b4.npz")
    faces = torch.from_numpy(g["faces"]).to(dev)
    tri32 = ops.tri_int32(faces)
    V, F = g["v_pos"].shape[1], faces.shape[0]
    adj = ops.VertexFaceAdjacency(tri32, V)
    off, lst = adj.off.cpu().numpy(), adj.adj.cpu().numpy()[: 3 * F]
    flat = g["faces"].T.reshape(-1)  # corner-major: entry key = c*F + f
    order = np.argsort(flat, kind="stable")
    assert np.array_equal(off, np.concatenate([[0], np.cumsum(np.bincount(flat, minlength=V))]))
    assert np.array_equal(lst, order.astype(np.int32))
    v = torch.from_numpy(g["v_pos"]).to(dev).requires_grad_(True)
    runs = []
    for _ in range(3):
        n = ops.vertex_normals(v, faces)
        (gv,) = torch.autograd.grad((n * n.roll(1, -1)).sum() + n[..., 0].sum(), v)
        runs.append((n.detach().clone(), gv.clone()))
    assert all(torch.equal(runs[0][0], r[0]) and torch.equal(runs[0][1], r[1]) for r in runs[1:])
    exact = float((runs[0][0].cpu() == torch.from_numpy(g["v_nrm"])).float().mean())
    print(f"normals bit-identical to the reference CPU golden on {exact:.4%} of elements")


def test_canonical_mesh_normals_ride_in_the_posed_meshes_launch(dev, ops, mods):
    """make_mesh for the canonical mesh (B = 1) and for B posed meshes over the same triangle list: the first read of the posed meshes'
    normals computes the canonical mesh's too (one launch, a3d_normals_fwd_pair) -- same bits as separate launches, gradients reach both
    vertex arrays, a larger pending mesh (the deformed ones, never read) is not picked up, and nothing is shared across triangle lists."""
    M = mods["mesh"]
    L = importlib.import_module("3danimals_amd._lib")
    verts, faces = quadruped_mesh(16, 0.3)
    tri = faces.to(dev)[None]
    uv, uvi = torch.zeros(1, 4, 2, device=dev), torch.zeros(1, faces.shape[0], 3, dtype=torch.int64, device=dev)
    B = 5
    v1 = verts[None].to(dev).requires_grad_(True)
    vB = (verts[None] + 0.03 * seeded((B, *verts.shape), 3, -1, 1)).to(dev).requires_grad_(True)
    vD = (verts[None] + 0.01 * seeded((B, *verts.shape), 4, -1, 1)).to(dev)
    prior = M.make_mesh(v1, tri, uv, uvi, None)
    deformed = M.make_mesh(vD, tri, uv.expand(B, -1, -1), uvi, None)
    posed = M.make_mesh(vB, tri, uv.expand(B, -1, -1), uvi, None)
    with L.KernelTimer() as timer:
        nB = posed.v_nrm
        n1 = prior.v_nrm  # already there
    names = [n for n in timer.summary() if n.startswith("a3d_normals")]  # (+ a3d_mesh_topology: this list did not come from the DMTet emit)
    assert names == [f"a3d_normals_fwd_pair[B{B}+B1]"], names
    assert deformed._v_nrm is None and deformed._lazy_nrm is not None  # not touched
    ref1, refB = ops.vertex_normals(v1.detach(), tri), ops.vertex_normals(vB.detach(), tri)
    assert torch.equal(n1, ref1) and torch.equal(nB, refB)
    w1, wB = seeded(tuple(n1.shape), 5, -1, 1).to(dev), seeded(tuple(nB.shape), 6, -1, 1).to(dev)
    g1, gB = torch.autograd.grad((n1 * w1).sum() + (nB * wB).sum(), [v1, vB])
    v1r, vBr = v1.detach().clone().requires_grad_(True), vB.detach().clone().requires_grad_(True)
    r1, rB = torch.autograd.grad((ops.vertex_normals(v1r, tri) * w1).sum() + (ops.vertex_normals(vBr, tri) * wB).sum(), [v1r, vBr])
    assert torch.equal(g1, r1) and torch.equal(gB, rB)
    # only one of the pair used downstream: the other's gradient is simply absent
    prior2, posed2 = M.make_mesh(v1, tri, uv, uvi, None), M.make_mesh(vB, tri, uv.expand(B, -1, -1), uvi, None)
    (gB2,) = torch.autograd.grad((posed2.v_nrm * wB).sum(), [vB])
    assert torch.equal(gB2, rB) and prior2._v_nrm is not None
    # another triangle list: no pairing
    other = M.make_mesh(v1, tri.clone(), uv, uvi, None)
    posed3 = M.make_mesh(vB, tri, uv.expand(B, -1, -1), uvi, None)
    _ = posed3.v_nrm
    assert other._v_nrm is None


def test_normals_isolated_vertex_and_empty(dev, ops):
    g = golden("mesh_isolated.npz")
    nrm = ops.vertex_normals(torch.from_numpy(g["v_pos"]).to(dev), torch.from_numpy(g["faces"]).to(dev))
    np.testing.assert_allclose(nrm.cpu().numpy(), g["v_nrm"], atol=2e-6)
    assert np.array_equal(nrm[0, -1].cpu().numpy(), [0.0, 0.0, 1.0])
    none = ops.vertex_normals(torch.rand(2, 5, 3, device=dev), torch.zeros(0, 3, dtype=torch.int64, device=dev))
    assert np.array_equal(none.cpu().numpy(), np.tile([0.0, 0.0, 1.0], (2, 5, 1)))


# ------------------------------------------------------------------------------------------------ skinning
@pytest.mark.parametrize("tag", ["b1f1_t1", "b3f2_t005", "b2f2_inst"])
def test_skinning_matches_reference_golden(tag, dev, mods):
    g = golden(f"skinning_{tag}.npz")
    chain = eval(str(g["chain"]))
    v = torch.from_numpy(g["v_in"]).to(dev).requires_grad_(True)
    ang = torch.from_numpy(g["angles"]).to(dev).requires_grad_(True)
    out, aux = mods["skinning"].skinning(v, torch.from_numpy(g["bones"]).to(dev), chain, ang, output_posed_bones=True,
                                         temperature=float(g["temperature"]))
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["out"], atol=5e-6)
    np.testing.assert_allclose(aux["posed_bones"].detach().cpu().numpy(), g["posed_bones"], atol=5e-6)
    w = aux["vertices_to_bones"]
    assert tuple(w.shape) == tuple(g["weights"].shape)
    np.testing.assert_allclose(w.cpu().numpy(), g["weights"], atol=2e-6)
    wgt, wgt_b = seeded(out.shape, 77, -1, 1).to(dev), seeded(aux["posed_bones"].shape, 78, -1, 1).to(dev)
    gv, ga = torch.autograd.grad((out * wgt).sum() + (aux["posed_bones"] * wgt_b).sum(), [v, ang])
    np.testing.assert_allclose(gv.cpu().numpy(), g["grad_v"], rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(ga.cpu().numpy(), g["grad_angles"], rtol=1e-3, atol=1e-3)


def test_skinning_identity_and_oracle_larger(dev, mods):
    from oracle import skinning_ref

    verts, _ = quadruped_mesh(24, 0.25)
    sk = mods["skinning"]
    bones, tree, _ = sk.estimate_bones(verts[None, None], n_body_bones=8, n_legs=4, n_leg_bones=3, body_bones_mode="z_minmax_y+")
    B = 5
    zero = torch.zeros(B, 1, 20, 3, device=dev)
    out, _ = sk.skinning(verts[None, None].to(dev), bones.to(dev), tree, zero, temperature=0.05)
    np.testing.assert_allclose(out.cpu().numpy(), verts[None, None].expand(B, 1, -1, -1).numpy(), atol=2e-6)  # zero angles -> identity
    ang = seeded((B, 1, 20, 3), 9, -0.6, 0.6)
    ref, _ = skinning_ref.skinning(verts[None, None], bones, tree, ang, 0.05)
    out, _ = sk.skinning(verts[None, None].to(dev), bones.to(dev), tree, ang.to(dev), temperature=0.05)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), atol=1e-5)


@pytest.mark.parametrize("B,F,shared", [(16, 1, True), (3, 4, False), (1, 1, True)])
def test_fused_pose_skinning_equals_the_two_launch_path(B, F, shared, dev, mods):
    """a3d_skin_pose_* (chain composition inside the skinning launches; the chain adjoint as the tail of the work-group that finishes an
    image last) against a3d_bone_transforms_* + a3d_skin_*: same posed vertices and transforms, same gradients to the rest vertices and
    the angles -- with the loss on the vertices, on the posed bones only (the transforms' direct gradient), on both, and for a second
    backward through the same graph (the cleared g_T / ticket buffer serves one backward; the next one takes the memset path)."""
    sk = mods["skinning"]
    verts, _ = quadruped_mesh(24, 0.25)
    bones, tree, _ = sk.estimate_bones(verts[None, None], n_body_bones=8, n_legs=4, n_leg_bones=3, body_bones_mode="z_minmax_y+")
    V = verts.shape[0]
    rest = verts[None, None] if shared else verts[None, None] + 0.02 * seeded((B, F, V, 3), 3, -1, 1)
    ang0 = seeded((B, F, 20, 3), 9, -0.6, 0.6)
    wv, wb = seeded((B, F, V, 3), 5, -1, 1).to(dev), seeded((B, F, 20, 2, 3), 6, -1, 1).to(dev)

    def run(fused, which):
        sk.FUSED_POSE = fused
        try:
            v, a = rest.to(dev).requires_grad_(True), ang0.to(dev).requires_grad_(True)
            out, aux = sk.skinning(v, bones.to(dev), tree, a, output_posed_bones=True, temperature=0.05)
            loss = ((out * wv).sum() if which != "bones" else 0) + ((aux["posed_bones"] * wb).sum() if which != "verts" else 0)
            g1 = torch.autograd.grad(loss, [v, a], retain_graph=True, allow_unused=True)
            g2 = torch.autograd.grad(loss, [v, a], allow_unused=True)
            return out.detach(), aux["posed_bones"].detach(), g1, g2
        finally:
            sk.FUSED_POSE = True

    for which in ("verts", "bones", "both"):
        o_f, b_f, g1_f, g2_f = run(True, which)
        o_s, b_s, g1_s, _ = run(False, which)
        np.testing.assert_allclose(o_f.cpu().numpy(), o_s.cpu().numpy(), atol=2e-6)
        np.testing.assert_allclose(b_f.cpu().numpy(), b_s.cpu().numpy(), atol=2e-6)
        for gf, gs, g2, name in zip(g1_f, g1_s, g2_f, ("rest", "angles")):
            if gs is None:
                assert gf is None or float(gf.abs().max()) == 0.0, (which, name)
                continue
            scale = float(gs.abs().max())
            assert scale > 0, (which, name)
            np.testing.assert_allclose(gf.cpu().numpy(), gs.cpu().numpy(), rtol=1e-4, atol=2e-5 * scale, err_msg=f"{which}/{name}")
            np.testing.assert_allclose(g2.cpu().numpy(), gf.cpu().numpy(), rtol=1e-5, atol=2e-6 * scale, err_msg=f"{which}/{name} second backward")


@pytest.mark.parametrize("K", [7, 20, 28, 40])
def test_skin_blend_for_any_skeleton_size_vs_torch(K, dev, ops):
    """a3d_skin_fwd / a3d_skin_bwd (the blend alone: transforms given) against plain torch fp32 for skeletons below, at and above the
    reference's 20 bones -- the kernels come in three sizes (<= 20, <= 32, <= 64 bones), the backward's transform gradient is a matrix
    product over 16-bone tiles (fp32 MFMA), and only K = 20 is exercised by the reference's configurations.  Weights = softmax over the
    bones of -distance(point, bone segment) / temperature on a DETACHED copy of the vertices (skinning.py:377)."""
    B, V, temp = 3, 777, 0.07
    v = seeded((B, V, 3), 31, -1, 1).to(dev).requires_grad_(True)
    bones = seeded((B, K, 2, 3), 32, -1, 1).to(dev)
    T = (torch.eye(3, 4).reshape(1, 1, 12) + 0.3 * seeded((B, K, 12), 33, -1, 1)).to(dev).requires_grad_(True)
    wgt = seeded((B, V, 3), 34, -1, 1).to(dev)

    def ref(v, T):
        a, d = bones[:, :, 0][:, :, None], (bones[:, :, 1] - bones[:, :, 0])[:, :, None]  # [B,K,1,3]
        r = v.detach()[:, None] - a  # [B,K,V,3]
        t = ((r * d).sum(-1) * (1.0 / (d * d).sum(-1).clamp_min(1e-6))).clamp(0, 1)
        s_ = t[..., None] * d - r
        w = torch.softmax(-torch.sqrt((s_ * s_).sum(-1) + 1e-6) / temp, dim=1)  # [B,K,V]
        R = T.reshape(B, K, 3, 4)
        posed = torch.einsum("bkij,bvj->bkvi", R[..., :3], v) + R[..., 3][:, :, None]
        return (w[..., None] * posed).sum(1)

    out = ops.skin(v, bones, T, temp)
    want = ref(v, T)
    assert float((out - want).abs().max()) <= 2e-5, float((out - want).abs().max())
    g = torch.autograd.grad((out * wgt).sum(), [v, T])
    gw = torch.autograd.grad((want * wgt).sum(), [v, T])
    for x, y, name in zip(g, gw, ("vertices", "transforms")):
        scale = float(y.abs().max())
        assert scale > 0 and float((x - y).abs().max()) <= 2e-5 * scale, (name, float((x - y).abs().max()), scale)


# ------------------------------------------------------------------------------------------------ rasterise
def _scene(B, res=16, seed=1):
    from oracle import render_ref

    synthetic = importlib.import_module("3danimals_amd.synthetic")
    verts, faces = quadruped_mesh(res, 0.3)
    mvp, w2c, campos = synthetic.random_cameras(B, seed=seed)
    clip = render_ref.xfm_points(verts[None].expand(B, -1, -1), mvp)
    return verts, faces, clip.contiguous(), (mvp, w2c, campos)


@pytest.mark.parametrize("B,H,W", [(1, 64, 64), (2, 50, 70), (4, 256, 256), (16, 256, 256)])
def test_rasterize_bit_exact_vs_oracle(B, H, W, dev, ops):
    """All three tile configurations (16x16, 32x32, 64x64 tiles) incl. a resolution that is no tile multiple."""
    from oracle import raster_ref

    _, faces, clip, _ = _scene(B)
    ref = raster_ref.rasterize(clip, faces.int(), (H, W))
    out = ops.rasterize(clip.to(dev), faces.to(dev), (H, W)).cpu()
    assert np.array_equal(out[..., 3].numpy(), ref[..., 3].numpy())  # triangle ids bit-exact
    assert np.array_equal(out.numpy(), ref.numpy())  # u, v, z/w: identical rounding sequence
    assert float((out[..., 3] > 0).float().mean()) > 0.05


def test_depth_peeling_bit_exact_vs_oracle(dev, ops):
    """DepthPeeler layers 1 and 2 of a closed mesh (front surface, then what lies behind it) against oracle/raster_ref, ids and values."""
    from oracle import raster_ref

    B, H, W = 2, 96, 80
    _, faces, clip, _ = _scene(B, seed=4)
    tri = faces.int()
    prev_o = prev_h = None
    for layer in range(3):
        prev_o = raster_ref.rasterize(clip, tri, (H, W), prev=prev_o)
        prev_h = ops.rasterize(clip.to(dev), tri.to(dev), (H, W), prev=prev_h)
        assert torch.equal(prev_h.cpu()[..., 3], prev_o[..., 3]), layer
        np.testing.assert_allclose(prev_h.cpu().numpy(), prev_o.numpy(), atol=1e-6)
        if layer:
            assert 0 < int((prev_o[..., 3] > 0).sum()) <= cover
        cover = int((prev_o[..., 3] > 0).sum())


def test_rasterize_edge_cases(dev, ops):
    from oracle import raster_ref

    H = W = 32
    pos = torch.tensor([[[-1.5, -1.5, 0.2, 1.0], [1.5, -1.5, 0.2, 1.0], [0.0, 1.5, 0.2, 1.0],  # 0: covers most of the screen
                         [-0.5, -0.5, -0.5, 1.0], [0.5, -0.5, -0.5, 1.0], [0.0, 0.5, -0.5, 1.0],  # 1: nearer, smaller
                         [-0.2, -0.2, 0.0, -1.0], [0.2, -0.2, 0.0, -1.0], [0.0, 0.2, 0.0, -1.0],  # 2: behind the eye (all w<0)
                         [-0.3, 0.1, 0.0, 1.0], [0.3, 0.1, 0.0, 1.0], [0.0, 0.1, 0.0, -0.5],  # 3: straddles w=0
                         [0.1, 0.1, 0.0, 1.0], [0.1, 0.1, 0.0, 1.0], [0.1, 0.1, 0.0, 1.0],  # 4: degenerate
                         [-0.5, -0.5, 2.0, 1.0], [0.5, -0.5, 2.0, 1.0], [0.0, 0.5, 2.0, 1.0]]])  # 5: beyond the far plane
    tri = torch.arange(18, dtype=torch.int32).view(6, 3)
    ref = raster_ref.rasterize(pos, tri, (H, W))
    out = ops.rasterize(pos.to(dev), tri.to(dev), (H, W)).cpu()
    assert np.array_equal(out.numpy(), ref.numpy())
    ids = set(np.unique(out[..., 3].numpy()).astype(int))
    assert {1, 2}.issubset(ids) and 3 not in ids and 5 not in ids and 6 not in ids
    # exact barycentrics of a screen-aligned triangle at a known pixel: triangle 1, pixel centre (x=16.5,y=16.5)/32 -> ndc 0.03125
    u, v = float(out[0, 16, 16, 0]), float(out[0, 16, 16, 1])
    fx = fy = 0.03125
    a0 = (0.5 - fx) * (0.5 - fy) - (-0.5 - fy) * (0.0 - fx)
    a1 = (0.0 - fx) * (-0.5 - fy) - (0.5 - fy) * (-0.5 - fx)
    a2 = (-0.5 - fx) * (-0.5 - fy) - (-0.5 - fy) * (0.5 - fx)
    assert abs(u - a0 / (a0 + a1 + a2)) < 1e-6 and abs(v - a1 / (a0 + a1 + a2)) < 1e-6 and int(out[0, 16, 16, 3]) == 2
    # empty mesh -> all zeros
    empty = ops.rasterize(pos.to(dev), torch.zeros(0, 3, dtype=torch.int32, device=dev), (H, W))
    assert float(empty.abs().max()) == 0.0


def test_rasterize_watertight_and_deterministic(dev, ops):
    """No pin-holes along shared edges of a closed mesh; identical output run to run (order-independent depth test)."""
    from oracle import raster_ref

    _, faces, clip, _ = _scene(4, res=24)
    a = ops.rasterize(clip.to(dev), faces.to(dev), (256, 256))
    b = ops.rasterize(clip.to(dev), faces.to(dev), (256, 256))
    assert torch.equal(a, b)
    cover = (a[..., 3] > 0).float()[:, None]
    # a hole = uncovered pixel whose 4 neighbours are all covered
    nb = torch.nn.functional.conv2d(cover, torch.tensor([[[[0, 1, 0], [1, 0, 1], [0, 1, 0.0]]]], device=dev), padding=1)
    assert int(((cover == 0) & (nb == 4)).sum()) == 0


def test_rasterize_backward_vs_oracle(dev, ops):
    from oracle import raster_ref

    _, faces, clip, _ = _scene(2)
    H = W = 64
    clip_d = clip.to(dev).requires_grad_(True)
    rast = ops.rasterize(clip_d, faces.to(dev), (H, W))
    wgt = seeded((2, H, W, 2), 3, -1, 1)
    (g,) = torch.autograd.grad((rast[..., :2] * wgt.to(dev)).sum(), clip_d)
    clip_c = clip.clone().requires_grad_(True)
    uv = raster_ref.barycentrics(clip_c, faces.int(), rast.detach().cpu())
    (g_ref,) = torch.autograd.grad((uv * wgt).sum(), clip_c)
    scale = float(g_ref.abs().max())
    np.testing.assert_allclose(g.cpu().numpy(), g_ref.numpy(), rtol=1e-3, atol=2e-4 * scale)
    assert float(g[..., 2].abs().max()) == 0.0  # z carries no gradient


# ------------------------------------------------------------------------------------------------ interpolate
@pytest.mark.parametrize("C,shared", [(3, False), (3, True), (2, False), (17, False)])
def test_interpolate_fwd_bwd_vs_oracle(C, shared, dev, ops):
    from oracle import raster_ref

    B, H, W = 3, 64, 64
    verts, faces, clip, _ = _scene(B)
    rast = raster_ref.rasterize(clip, faces.int(), (H, W))
    attr = seeded((1 if shared else B, verts.shape[0], C), 4, -1, 1)
    a_c, r_c = attr.clone().requires_grad_(True), rast.clone().requires_grad_(True)
    a_d, r_d = attr.to(dev).requires_grad_(True), rast.to(dev).requires_grad_(True)
    ref = raster_ref.interpolate(a_c, r_c, faces.int())
    out = ops.interpolate(a_d, r_d, faces.to(dev))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), atol=1e-6)
    wgt = seeded(ref.shape, 6, -1, 1)
    ga_ref, gr_ref = torch.autograd.grad((ref * wgt).sum(), [a_c, r_c])
    ga, gr = torch.autograd.grad((out * wgt.to(dev)).sum(), [a_d, r_d])
    np.testing.assert_allclose(ga.cpu().numpy(), ga_ref.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(gr.cpu().numpy()[..., :2], gr_ref.numpy()[..., :2], rtol=1e-4, atol=1e-5)
    assert float(gr[..., 2:].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------ antialias
def test_aa_topology_matches_oracle(dev, ops):
    from oracle import raster_ref

    _, faces = quadruped_mesh(16)
    topo = ops.aa_topology(ops.tri_int32(faces.to(dev)), int(faces.max()) + 1)
    assert np.array_equal(topo.opp.cpu().numpy(), raster_ref.edge_opposites(faces.numpy()))
    # open mesh (drop some faces -> boundary edges), a degenerate face and a non-manifold fan
    cut = torch.cat([faces[: len(faces) // 2], torch.tensor([[0, 0, 1], [0, 1, 2], [1, 0, 3], [0, 1, 4]])])
    topo = ops.AATopology(cut.to(dev).int().contiguous(), int(cut.max()) + 1)
    assert np.array_equal(topo.opp.cpu().numpy(), raster_ref.edge_opposites(cut.numpy()))
    # the same table from the vertex -> face lists (what the silhouette analysis walks when it is given lists instead of a table):
    # degenerate faces, boundary edges and the non-manifold fan included
    # (round 4) ... and the same meshes wound inconsistently: a random half of the faces flipped -- every face still finds its one
    # neighbour across a manifold edge, through the hash and through the lists
    flip = torch.rand(faces.shape[0], generator=torch.Generator().manual_seed(2)) < 0.5
    flipped = torch.where(flip[:, None], faces[:, [0, 2, 1]], faces)
    assert (raster_ref.edge_opposites(flipped.numpy()) >= 0).all()
    topo = ops.AATopology(flipped.to(dev).int().contiguous(), int(flipped.max()) + 1)
    assert np.array_equal(topo.opp.cpu().numpy(), raster_ref.edge_opposites(flipped.numpy()))
    for tri in (faces, cut, flipped, torch.cat([flipped[: len(faces) // 2], cut[-4:]])):
        tri32 = tri.to(dev).int().contiguous()
        lists = ops.VertexFaceAdjacency(tri32, int(tri.max()) + 1)
        assert np.array_equal(ops.opposite_vertices_from_lists(lists).cpu().numpy(), raster_ref.edge_opposites(tri.numpy()))
        adj, topo2 = ops.mesh_topology(tri32.clone(), int(tri.max()) + 1)  # (both tables in one entry point: the same hash)
        assert np.array_equal(topo2.opp.cpu().numpy(), raster_ref.edge_opposites(tri.numpy()))


def test_antialias_analysis_from_lists_equals_the_table_based_one(dev, ops):
    """a3d_aa_analyze with opp = NULL + the vertex -> face lists against the table-based analysis: the same silhouette records (as a
    set: the append order is not defined), hence the same antialiased image up to the blend order."""
    B, H, W, C = 3, 96, 96, 4
    _, faces, clip, _ = _scene(B, seed=5)
    tri32 = faces.to(dev).int().contiguous()
    V = clip.shape[1]
    clip_d = clip.to(dev)
    rast = ops.rasterize(clip_d, tri32, (H, W))
    lists, table = ops.mesh_topology(tri32, V)
    a_table = ops.AAAnalysis(rast, clip_d, table)
    a_lists = ops.AAAnalysis(rast, clip_d, ops.AATopology(tri32, V, build=False, lists=lists))

    def records(a):
        shards = a.count.shape[0]
        seg = a.capacity // shards
        rows = [a.work[s * seg: s * seg + int(n)] for s, n in enumerate(a.count.cpu().tolist())]
        r = torch.cat(rows).cpu().numpy()
        return r[np.lexsort(r.T[::-1])]

    ra, rb = records(a_table), records(a_lists)
    assert ra.shape[0] > 500 and np.array_equal(ra, rb)
    col = torch.lerp(seeded((B, H, W, C), 1, 0, 0.2).to(dev), seeded((B, H, W, C), 2, 0.5, 1.0).to(dev), (rast[..., 3:] > 0).float())
    o1 = ops.antialias(col, rast, clip_d, tri32, analysis=a_table)
    o2 = ops.antialias(col, rast, clip_d, tri32, analysis=a_lists)
    assert float((o1 - o2).abs().max()) <= 2.4e-7


@pytest.mark.parametrize("C", [4, 17])
def test_antialias_fwd_bwd_vs_oracle(C, dev, ops):
    from oracle import raster_ref

    B, H, W = 2, 64, 64
    _, faces, clip, _ = _scene(B)
    rast = raster_ref.rasterize(clip, faces.int(), (H, W))
    cover = (rast[..., 3:] > 0).float()
    color = torch.lerp(seeded((B, H, W, C), 1, 0, 0.2), seeded((B, H, W, C), 2, 0.5, 1.0), cover)
    c_c, p_c = color.clone().requires_grad_(True), clip.clone().requires_grad_(True)
    ref = raster_ref.antialias(c_c, rast, p_c, faces.int())
    c_d, p_d = color.to(dev).requires_grad_(True), clip.to(dev).requires_grad_(True)
    out = ops.antialias(c_d, rast.to(dev), p_d, faces.to(dev))
    assert float((ref - color).abs().max()) > 0.05  # the silhouette really was blended
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), atol=2e-6)
    wgt = seeded(ref.shape, 8, -1, 1)
    gc_ref, gp_ref = torch.autograd.grad((ref * wgt).sum(), [c_c, p_c])
    gc, gp = torch.autograd.grad((out * wgt.to(dev)).sum(), [c_d, p_d])
    np.testing.assert_allclose(gc.cpu().numpy(), gc_ref.numpy(), atol=2e-6)
    scale = float(gp_ref.abs().max())
    assert scale > 0
    np.testing.assert_allclose(gp.cpu().numpy(), gp_ref.numpy(), rtol=1e-3, atol=2e-4 * scale)


@pytest.mark.parametrize("fused", [False, True])
def test_antialias_run_to_run_difference_is_bounded_by_the_blend_order(fused, dev, ops):
    """The silhouette crossings are appended to 256 work-list segments with atomics, so the order in which a pixel's (at most four:
    left / right / up / down pair) blend terms are added changes from run to run -- the only non-determinism of the antialiasing ops.
    Stated bound: every output value within 2 ulp of 1.0 (2.4e-7) of any other run, only silhouette pixels differ at all, and the two
    gradients agree to float-atomic order."""
    B, H, W, C = 4, 128, 128, 4
    _, faces, clip, _ = _scene(B, seed=9)
    tri = faces.to(dev)
    clip_d = clip.to(dev)
    rast = ops.rasterize(clip_d, tri, (H, W))
    wgt = seeded((B, H, W, C), 8, -1, 1).to(dev)

    def run():
        p = clip_d.clone().requires_grad_(True)
        if fused:
            pix, inv = ops.covered_pixels(rast, return_inverse=True)
            vals = seeded((pix.shape[0], C - 1), 3, 0.3, 1.0).to(dev).requires_grad_(True)
            analysis = ops.AAAnalysis(rast, p, ops.aa_topology(ops.tri_int32(tri), clip.shape[1]))
            out = ops.composite_antialias(vals, pix, inv, None, p, analysis)
            leaf = vals
        else:
            cover = (rast[..., 3:] > 0).float()
            leaf = torch.lerp(seeded((B, H, W, C), 1, 0, 0.2).to(dev), seeded((B, H, W, C), 2, 0.5, 1.0).to(dev), cover).requires_grad_(True)
            out = ops.antialias(leaf, rast, p, tri)
        gl, gp = torch.autograd.grad((out * wgt).sum(), [leaf, p])
        return out.detach(), gl, gp

    runs = [run() for _ in range(4)]
    o0, gl0, gp0 = runs[0]
    for o, gl, gp in runs[1:]:
        d = (o - o0).abs()
        assert float(d.max()) <= 2.4e-7, float(d.max())
        assert int((d > 0).any(-1).sum()) <= 0.05 * B * H * W  # silhouette pixels only
        assert float((gl - gl0).abs().max()) <= 5e-7 * float(gl0.abs().max())
        assert float((gp - gp0).abs().max()) <= 2e-5 * float(gp0.abs().max())


@pytest.mark.parametrize("two", [False, True])
def test_silhouette_analysis_riding_in_the_compositor_launch_leaves_the_same_records(two, dev, ops):
    """AAAnalysis(defer=True): the analysis runs as extra work-groups of the compositor's first launch instead of a launch of its own.
    Same crossing records (as a set: the append order is the atomics'), same images up to the blend order, and nothing is launched
    before the compositor call."""
    _lib = importlib.import_module("3danimals_amd._lib")
    B, H, W = 3, 96, 96
    _, faces, clip, _ = _scene(B, seed=11)
    tri, clip_d = faces.to(dev), clip.to(dev)
    topo = ops.aa_topology(ops.tri_int32(tri), clip.shape[1])

    def records(a):
        cnt = a.count.cpu().numpy()
        seg = a.capacity // cnt.shape[0]
        w = a.work.cpu().numpy()
        rows = np.concatenate([w[i * seg:i * seg + min(int(c), seg)] for i, c in enumerate(cnt)])
        return rows[np.lexsort(rows.T[::-1])]

    def run(defer):
        rast = ops.rasterize(clip_d, tri, (H, W))  # (prepares the screen positions and the counters for this clip)
        pix, inv = ops.covered_pixels(rast, return_inverse=True)
        vals = seeded((pix.shape[0], 3), 3, 0.3, 1.0).to(dev)
        vals2 = seeded((pix.shape[0], 16), 4, 0.0, 1.0).to(dev) if two else None
        with _lib.KernelTimer() as timer:
            a = ops.AAAnalysis(rast, clip_d, topo, defer=defer)
            assert a.pending == defer and (not defer or not timer.records)
            out = ops.composite_antialias(vals, pix, inv, None, clip_d, a, vals2=vals2)
        assert not a.pending
        names = set(timer.summary())
        assert ("a3d_aa_analyze" in names) != defer and any(n.endswith("[+analysis]") for n in names) == defer, names
        return records(a), (out if two else (out,))

    rec_ride, out_ride = run(True)
    rec_alone, out_alone = run(False)
    assert rec_ride.shape[0] > 100 and np.array_equal(rec_ride, rec_alone)
    for x, y in zip(out_ride, out_alone):
        assert float((x - y).abs().max()) <= 2.4e-7
    # a second compositor call over the same analysis finds it done
    rast = ops.rasterize(clip_d, tri, (H, W))
    pix, inv = ops.covered_pixels(rast, return_inverse=True)
    a = ops.AAAnalysis(rast, clip_d, topo, defer=True)
    v = seeded((pix.shape[0], 3), 3, 0.3, 1.0).to(dev)
    o1 = ops.composite_antialias(v, pix, inv, None, clip_d, a)
    o2 = ops.composite_antialias(v, pix, inv, None, clip_d, a)
    assert float((o1 - o2).abs().max()) <= 2.4e-7


def test_antialias_known_answer_vertical_edge(dev, ops):
    """A surface covering x < k+0.3: pixel k (centre k+0.5, uncovered) takes 0.3 of its covered left neighbour."""
    H = W = 16
    k = 8
    xe = (k + 0.3) / W * 2 - 1
    pos = torch.tensor([[[-3.0, -3.0, 0.0, 1.0], [xe, -3.0, 0.0, 1.0], [xe, 3.0, 0.0, 1.0], [-3.0, 3.0, 0.0, 1.0]]])
    tri = torch.tensor([[0, 1, 2], [0, 2, 3]], dtype=torch.int32)
    rast = ops.rasterize(pos.to(dev), tri.to(dev), (H, W))
    cover = (rast[..., 3:] > 0).float()
    assert float(cover[0, 5, k - 1]) == 1.0 and float(cover[0, 5, k]) == 0.0
    out = ops.antialias(cover.contiguous(), rast, pos.to(dev), tri.to(dev)).cpu()
    np.testing.assert_allclose(out[0, 2:-2, k, 0].numpy(), 0.3, atol=1e-5)
    np.testing.assert_allclose(out[0, 2:-2, k - 1, 0].numpy(), 1.0, atol=1e-6)
    np.testing.assert_allclose(out[0, 2:-2, k + 1, 0].numpy(), 0.0, atol=1e-6)


# ------------------------------------------------------------------------------------------------ render_mesh end to end
def _nets(mods, dev, feat_dim=16):
    torch.manual_seed(0)
    N, L = mods["nets"], mods["light"]
    tex = N.CoordMLP(3, 9, 3, nf=32, activation="sigmoid", min_max=torch.tensor([[0.0, 1.0]] * 9), n_harmonic_functions=4, extra_feat_dim=feat_dim,
                     symmetrize=True)
    dino = N.CoordMLP(3, 16, 3, nf=32, activation="sigmoid", min_max=torch.tensor([[0.0, 1.0]] * 16), n_harmonic_functions=4)
    lgt = L.DirectionalLight(feat_dim, 3, 32, intensity_min_max=torch.tensor([[0.0, 1.0], [0.5, 1.0]]))
    return tex, dino, lgt


@pytest.mark.parametrize("modes,with_nets", [(["shaded", "dino_pred"], True), (["geo_normal", "kd", "shading", "normal", "depth"], True),
                                             (["shaded"], False)])
def test_render_mesh_matches_oracle(modes, with_nets, dev, mods):
    import copy

    from oracle import mesh_ref, render_ref

    B, H, W = 2, 64, 64
    verts, faces, _, (mvp, w2c, campos) = _scene(B, seed=3)
    posed = verts[None] + 0.05 * seeded((B, *verts.shape), 11, -1, 1)
    tex, dino, lgt = _nets(mods, dev)
    feat = seeded((B, 16), 12, -1, 1)
    bg = seeded((B, H, W, 3), 13, 0, 1)
    nrm = mesh_ref.vertex_normals(posed, faces)
    with torch.no_grad():
        ref = render_ref.render_mesh(posed, faces, nrm, mvp, w2c, campos, tex if with_nets else None, lgt if with_nets else None, (H, W),
                                     background=bg, feat=feat if with_nets else None, render_modes=modes, prior_v_pos=verts[None],
                                     dino_net=dino if with_nets else None)
    M = mods["mesh"]
    uvs = torch.zeros(1, 4, 2, device=dev)
    uvi = torch.zeros(1, faces.shape[0], 3, dtype=torch.int64, device=dev)
    shape = M.make_mesh(posed.to(dev), faces[None].to(dev), uvs.expand(B, -1, -1), uvi, None)
    prior = M.make_mesh(verts[None].to(dev), faces[None].to(dev), uvs, uvi, None)
    tex_d, dino_d, lgt_d = (copy.deepcopy(m).to(dev) for m in (tex, dino, lgt))
    with torch.no_grad():
        out = mods["render"].render_mesh(None, shape, mvp.to(dev), w2c.to(dev), campos.to(dev), tex_d if with_nets else None,
                                         lgt_d if with_nets else None, (H, W), background=bg.to(dev), bsdf="diffuse",
                                         feat=feat.to(dev) if with_nets else None, render_modes=modes, prior_mesh=prior,
                                         dino_net=dino_d if with_nets else None)
    assert len(out) == len(modes)
    for m, o, r in zip(modes, out, ref):
        if r is None:
            assert o is None, m
            continue
        assert tuple(o.shape) == tuple(r.shape), m
        np.testing.assert_allclose(o.cpu().numpy(), r.numpy(), atol=1e-4, err_msg=m)  # north_star tolerance


def test_render_mesh_flow_and_empty_mesh_assert(dev, mods):
    from oracle import mesh_ref, render_ref

    B, H, W = 4, 32, 32
    verts, faces, _, (mvp, w2c, campos) = _scene(B, seed=5)
    posed = verts[None] + 0.1 * seeded((B, *verts.shape), 21, -1, 1)
    nrm = mesh_ref.vertex_normals(posed, faces)
    ref = render_ref.render_mesh(posed, faces, nrm, mvp, w2c, campos, None, None, (H, W), render_modes=["shaded", "flow"], num_frames=2)
    M = mods["mesh"]
    uvs = torch.zeros(1, 4, 2, device=dev)
    uvi = torch.zeros(1, faces.shape[0], 3, dtype=torch.int64, device=dev)
    shape = M.make_mesh(posed.to(dev), faces[None].to(dev), uvs.expand(B, -1, -1), uvi, None)
    out = mods["render"].render_mesh(None, shape, mvp.to(dev), w2c.to(dev), campos.to(dev), None, None, (H, W), bsdf="diffuse",
                                     render_modes=["shaded", "flow"], num_frames=2)
    for o, r in zip(out, ref):
        np.testing.assert_allclose(o.cpu().numpy(), r.numpy(), atol=1e-4)
    # gradient of the flow image w.r.t. the posed vertices: fused path (flow through the modular interpolate / rasteriser backward,
    # everything else through the fused G-buffer) against torch autograd through the oracle
    wgt = seeded(tuple(ref[1].shape), 33, -1, 1)
    pc = posed.clone().requires_grad_(True)
    ref2 = render_ref.render_mesh(pc, faces, mesh_ref.vertex_normals(pc, faces), mvp, w2c, campos, None, None, (H, W), render_modes=["shaded", "flow"],
                                  num_frames=2)
    (g_ref,) = torch.autograd.grad((ref2[1] * wgt).sum(), pc)
    pd = posed.to(dev).requires_grad_(True)
    shape2 = M.make_mesh(pd, faces[None].to(dev), uvs.expand(B, -1, -1), uvi, None)
    out2 = mods["render"].render_mesh(None, shape2, mvp.to(dev), w2c.to(dev), campos.to(dev), None, None, (H, W), bsdf="diffuse",
                                      render_modes=["shaded", "flow"], num_frames=2)
    (g_hip,) = torch.autograd.grad((out2[1] * wgt.to(dev)).sum(), pd)
    assert float((g_hip.cpu() - g_ref).abs().max()) <= 2e-3 * float(g_ref.abs().max())
    empty = M.Mesh(posed.to(dev), torch.zeros(1, 0, 3, dtype=torch.int64, device=dev))
    with pytest.raises(AssertionError, match="empty training triangle mesh"):
        mods["render"].render_mesh(None, empty, mvp.to(dev), w2c.to(dev), campos.to(dev), None, None, (H, W), render_modes=["shaded"])


def test_allocator_cache_is_given_back_when_new_point_counts_have_stranded_it(dev, mods, monkeypatch):
    """Shading only the covered pixels makes the activation sizes follow the silhouette; every outgrown size strands its cached blocks
    (tools/mem_growth.py: 208 GB reserved after 1200 steps of a growing shape).  render._trim_allocator_cache: a point count no earlier
    step had + a cache above ALLOCATOR_TRIM_RATIO x the peak in use -> the cache goes back to the driver; a known count, a small cache
    or ratio 0 -> nothing happens."""
    R = mods["render"]
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats(dev)
    monkeypatch.setattr(R, "_point_counts_seen", set())
    monkeypatch.setattr(R, "_in_use_peak", {})
    base = torch.cuda.memory_reserved(dev)
    for k in range(24):  # 24 blocks of growing size, each freed again: ~5 GB cached, < 1 GB ever in use at once
        del_me = torch.empty((200 + 8 * k) << 20, dtype=torch.uint8, device=dev)
        del del_me
    stranded = torch.cuda.memory_reserved(dev) - base
    assert stranded > 4 << 30
    monkeypatch.setattr(R, "ALLOCATOR_TRIM_RATIO", 0.0)
    assert not R._trim_allocator_cache(8192, dev) and torch.cuda.memory_reserved(dev) - base == stranded
    monkeypatch.setattr(R, "ALLOCATOR_TRIM_RATIO", 3.0)
    assert R._trim_allocator_cache(8192, dev)
    assert torch.cuda.memory_reserved(dev) - base < 1 << 30
    keep = torch.empty(5 << 30, dtype=torch.uint8, device=dev)  # in use, not cache: nothing to give back
    assert not R._trim_allocator_cache(16384, dev)
    del keep
    assert not R._trim_allocator_cache(16384, dev)  # (a count seen before strands nothing new: not even looked at)
    # the peak in use is tracked by the module itself: a caller that resets torch's statistics (every epoch, say) does not make the
    # 5 GB that now sit in the cache -- exactly what the process needed a moment ago -- look like something to give back
    torch.cuda.reset_peak_memory_stats(dev)
    assert torch.cuda.memory_reserved(dev) - base >= 5 << 30 and not R._trim_allocator_cache(24576, dev)
    torch.cuda.empty_cache()
    # round 5, ALLOCATOR_TRIM_MODE 'step_end': the render only records the wish; the training loop gives the cache back between two steps
    monkeypatch.setattr(R, "ALLOCATOR_TRIM_MODE", "step_end")
    monkeypatch.setattr(R, "_in_use_peak", {})
    monkeypatch.setattr(R, "_trim_wanted", set())
    torch.cuda.reset_peak_memory_stats(dev)
    base = torch.cuda.memory_reserved(dev)
    for k in range(24):
        del_me = torch.empty((200 + 8 * k) << 20, dtype=torch.uint8, device=dev)
        del del_me
    stranded = torch.cuda.memory_reserved(dev) - base
    assert stranded > 4 << 30 and not R._trim_allocator_cache(32768, dev) and torch.cuda.memory_reserved(dev) - base == stranded
    assert R.allocator_trim_at_step_end(dev) and torch.cuda.memory_reserved(dev) - base < 1 << 30
    assert not R.allocator_trim_at_step_end(dev)  # (the wish is consumed)


def test_render_mesh_with_nothing_on_screen(dev, mods):
    """A mesh pushed entirely out of the frustum: no covered pixel (P = 0) through the fused path -- rasterise, covered-pixel list,
    G-buffer, shading, fused compositor -- forward returns the background with alpha 0 and the backward runs to zero gradients."""
    B, H, W = 2, 32, 32
    verts, faces, _, (mvp, w2c, campos) = _scene(B, seed=5)
    M = mods["mesh"]
    posed = (verts[None].expand(B, -1, -1) + torch.tensor([50.0, 0.0, 0.0])).to(dev).requires_grad_(True)
    uvs = torch.zeros(1, 4, 2, device=dev)
    uvi = torch.zeros(1, faces.shape[0], 3, dtype=torch.int64, device=dev)
    shape = M.make_mesh(posed, faces[None].to(dev), uvs.expand(B, -1, -1), uvi, None)
    bg = torch.rand(B, H, W, 3, generator=torch.Generator().manual_seed(3)).to(dev)
    out = mods["render"].render_mesh(None, shape, mvp.to(dev), w2c.to(dev), campos.to(dev), None, None, (H, W), bsdf="diffuse",
                                     background=bg, render_modes=["shaded", "flow"], num_frames=2)
    shaded, flow = out
    assert shaded.shape == (B, 4, H, W) and torch.equal(shaded[:, :3], bg.permute(0, 3, 1, 2)) and float(shaded[:, 3].detach().abs().max()) == 0.0
    assert float(flow.abs().max()) == 0.0
    (g,) = torch.autograd.grad(shaded.sum() + flow.sum(), posed, allow_unused=True)
    assert g is None or float(g.abs().max()) == 0.0


def test_backward_twice_with_retain_graph_gives_the_same_gradients(dev, mods, ops):
    """The G-buffer, skinning, shading and DMTet-vertex backwards accumulate into buffers that their forward launch cleared; those
    serve one backward.  A second backward through the same graph (retain_graph) takes the memset path: same gradients."""
    B, H, W = 2, 48, 48
    verts, faces, _, (mvp, w2c, campos) = _scene(B, seed=13)
    tri = faces.to(dev)
    ru = importlib.import_module("3danimals_amd.model.render.renderutils")
    v = (verts[None] + 0.02 * seeded((B, *verts.shape), 51, -1, 1)).to(dev).requires_grad_(True)
    K = 4
    bones = seeded((1, K, 2, 3), 52, -0.4, 0.4).to(dev)
    T = (torch.eye(3, 4).reshape(1, 1, 12).repeat(B, K, 1) + 0.01 * seeded((B, K, 12), 53, -1, 1)).to(dev).requires_grad_(True)
    posed = ops.skin(v, bones, T, 0.05)
    nrm = ops.vertex_normals(posed, tri)
    clip = ru.xfm_points(posed, mvp.to(dev))
    rast = ops.rasterize(clip, tri, (H, W))
    pix = ops.covered_pixels(rast)
    gb = ops.gbuffer(clip, posed, nrm, verts[None].to(dev), rast, tri, pix)
    img = torch.div(pix, H * W, rounding_mode="floor")
    par = torch.cat((w2c.to(dev)[:, :3, :3].reshape(B, 9), campos.to(dev).reshape(B, 3), seeded((B, 5), 54, 0.1, 1.0).to(dev)), -1).requires_grad_(True)
    kd = seeded((pix.shape[0], 3), 55, 0, 1).to(dev)
    n_s, shading, shaded = ops.shade_points(gb, par, kd, True, img=img)
    loss = (shaded * seeded(tuple(shaded.shape), 56, -1, 1).to(dev)).sum() + (n_s * seeded(tuple(n_s.shape), 57, -1, 1).to(dev)).sum()
    g1 = torch.autograd.grad(loss, (v, T, par), retain_graph=True)
    g2 = torch.autograd.grad(loss, (v, T, par))
    for a, b_, name in zip(g1, g2, ("v", "T", "par")):
        scale = float(a.abs().max())
        assert scale > 0, name
        torch.testing.assert_close(a, b_, atol=2e-5 * scale, rtol=1e-4, msg=name)  # float atomics: order only


def test_full_step_against_oracle_and_grads_finite(dev):
    """One fwd+bwd step of the synthetic training scene; every stage re-done by the oracle from the same numbers."""
    from oracle import check

    pipeline = importlib.import_module("3danimals_amd.pipeline")
    scene = pipeline.SyntheticScene(grid_res=16, batch=3, resolution=(64, 64), device=dev, seed=0, net_width=32, net_layers=3, feat_dim=16,
                                    embedder_freq=4)
    out = scene.step(backward=True, optimizer_step=False)
    rep = check.compare_step(scene, out)
    assert rep["faces_equal"], rep
    # posed normals: a near-degenerate vertex (incident face normals nearly cancel) amplifies the 1e-6 skinning difference when
    # normalising -- measured 2.6e-4 on one vertex; the rendered buffers below are the bar that counts
    assert rep["max_abs_vert_err"] == 0.0 and rep["max_abs_skin_err"] < 1e-5 and rep["max_abs_posed_normal_err"] < 2e-3, rep
    assert rep["max_abs_image_err"] < 1e-4, rep  # renderer on the same posed vertices, outside pixels whose owner flipped
    assert rep["frac_pixels_owner_flip"] < 1e-3 and rep["frac_pixels_gt_1e-4_end_to_end"] < 2e-3, rep  # a silhouette pixel or two
    assert 0.02 < rep["coverage"] < 0.9, rep
    for name, leaf in [("mvp", scene.mvp), ("campos", scene.campos), ("feat", scene.feat), ("arti", scene.arti)]:
        assert leaf.grad is not None and bool(torch.isfinite(leaf.grad).all()) and float(leaf.grad.abs().max()) > 0, name
    for n, p in scene.named_parameters():  # DDP runs with find_unused_parameters=False: every parameter must get a gradient
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), n


def test_ops_refuse_cpu_tensors(ops):
    A3DError = importlib.import_module("3danimals_amd._lib").A3DError
    with pytest.raises(A3DError, match="no CPU fallback"):
        ops.vertex_normals(torch.rand(1, 4, 3), torch.zeros(1, 3, dtype=torch.int64))


GRAD_RTOL, GRAD_ATOL_RMS = 1e-3, 1e-3


def _gradients_close(pairs, rtol=GRAD_RTOL, atol_rms=GRAD_ATOL_RMS):
    """Every (name, hip, oracle64) pair: |hip - oracle| <= rtol*|oracle| + atol_rms*RMS(oracle) PER ENTRY, against the float64 oracle
    (its own rounding is then out of the picture).  All offenders are reported together."""
    bad = []
    for name, a, b in pairs:
        assert a is not None, name
        a, b = a.detach().cpu().double(), b.double()
        rms = float(b.pow(2).mean().sqrt())
        if rms == 0:  # e.g. the camera position at 32x32: it only enters through the shading normal's bend, which may not trigger
            if float(a.abs().max()) >= 1e-6:
                bad.append((name, "oracle is zero", float(a.abs().max())))
            continue
        worst = float(((a - b).abs() / (rtol * b.abs() + atol_rms * rms)).max())
        if not worst <= 1.0:
            bad.append((name, round(worst, 2), f"max|d|={float((a - b).abs().max()):.3e} rms={rms:.3e}"))
    assert not bad, bad


def test_full_step_loss_and_gradients_vs_oracle_step(dev):
    """fwd+bwd of the whole path (HIP) against torch-CPU autograd through the oracle, from identical weights and inputs."""
    from oracle import step_ref

    pipeline = importlib.import_module("3danimals_amd.pipeline")
    scene = pipeline.SyntheticScene(grid_res=16, batch=2, resolution=(64, 64), device=dev, seed=3, net_width=32, net_layers=3, feat_dim=16,
                                    embedder_freq=4)
    out = scene.step(backward=True, optimizer_step=False, sdf_reg=False)
    ref = step_ref.cpu_step(step_ref.snapshot(scene), backward=True, dtype=torch.float64)
    assert ref["num_faces"] == scene.last["prior"].t_pos_idx.shape[1]
    np.testing.assert_allclose(float(out["loss"]), float(ref["loss"]), rtol=2e-5)
    np.testing.assert_allclose(out["shaded"].detach().cpu().numpy(), ref["shaded"].numpy(), atol=1e-4)
    pairs = [(k, getattr(scene, k).grad, ref["grads"][k]) for k in ("arti", "feat", "mvp", "campos", "w2c")]
    for name, mod in (("sdf_mlp", scene.netShape.mlp), ("tex", scene.netTexture), ("dino", scene.netDINO), ("lgt", scene.netLight)):
        pairs += [(f"{name}.{pn}", p.grad, ref["grads"][f"{name}.{pn}"]) for pn, p in mod.named_parameters()]
    _gradients_close(pairs)


def _owner_flip_mask(scene, ref):
    """[N,H,W] bool: pixels (and their antialiasing neighbours) whose owning triangle differs between the HIP rasteriser's buffer and
    the oracle's, both fed the same posed vertices -- the clip transform is a GPU matmul on one side and a CPU matmul on the other,
    and a last-bit difference can hand an edge pixel to a neighbouring triangle (oracle/check.py does the same)."""
    from oracle import raster_ref, render_ref

    n = ref["posed"].shape[0]
    clip = render_ref.xfm_points(ref["posed"].float(), scene.mvp.detach().cpu()[:n]).contiguous()
    rast_o = raster_ref.rasterize(clip, ref["faces"].int(), scene.resolution)
    flip = rast_o[..., 3] != scene.last["rast"].cpu()[:n, ..., 3]
    assert float(flip.float().mean()) < 2e-3
    return torch.nn.functional.max_pool2d(flip.float()[:, None], 3, 1, 1)[:, 0] > 0


@pytest.mark.parametrize("workload,kw", [("magicpony", dict(deform=True)), ("fauna", {}), ("ponymation", dict(num_frames=3)),
                                         ("ponymation", dict(num_frames=8, batch=1, resolution=(32, 32)))])
def test_workload_steps_vs_oracle_step(workload, kw, dev):
    """BASELINE configs 3 (with the instance deformation), 4 (train_fauna per rank: conditioned SDF, bones re-estimated inside the
    step with bone_y_threshold 0.4, second random-view mask render) and 5 (T-frame sequences, [B,F] skinning, 'flow' mode): one
    fwd+bwd HIP step against torch-CPU autograd through the oracle from identical weights and inputs."""
    from oracle import step_ref

    pipeline = importlib.import_module("3danimals_amd.pipeline")
    kw = dict(kw)
    batch, resolution = kw.pop("batch", 2), kw.pop("resolution", (64, 64))
    scene = pipeline.SyntheticScene(grid_res=16, batch=batch, resolution=resolution, device=dev, seed=5, net_width=32, net_layers=3, feat_dim=16,
                                    embedder_freq=4, workload=workload, **kw)
    out = scene.step(backward=True, optimizer_step=False, sdf_reg=False)
    ref = step_ref.cpu_step(step_ref.snapshot(scene), backward=True, dtype=torch.float64)
    assert torch.equal(ref["faces"], scene.last["prior"].t_pos_idx[0].cpu())  # index buffers bit-exact
    keep = ~_owner_flip_mask(scene, ref)
    for name in ("shaded", "dino_pred", "flow"):
        if name in ref:
            # END-TO-END figure against the float64 chain (oracle skinning -> normals -> render); the renderer's own 1e-4 bar on
            # identical stage inputs is test_full_step_against_oracle_and_grads_finite / test_full_size_workloads_stagewise_parity
            err = ((out[name].detach().cpu() - ref[name]).abs() * keep[:, None]).max()
            assert float(err) < 2e-4, (name, float(err))
    if workload == "ponymation":
        assert "flow" in ref and out["flow"].shape[1] == 2 and scene.frames == batch * kw["num_frames"]
    if workload == "fauna":  # the second render has its own cameras; its mask is 0/1 + antialiased edges
        d = (out["mask_random"].detach().cpu() - ref["mask_random"]).abs()
        assert float((d > 1e-4).float().mean()) < 2e-3
        assert scene.bone_aux is not None and scene.class_emb.grad is not None and float(scene.class_emb.grad.abs().max()) > 0
    np.testing.assert_allclose(float(out["loss"]), float(ref["loss"]), rtol=5e-5)
    for k, v in ref["losses"].items():
        np.testing.assert_allclose(out["losses"][k].detach().cpu().numpy(), v.numpy(), rtol=5e-4, atol=1e-6, err_msg=k)
    pairs = [(k, getattr(scene, k).grad, ref["grads"][k]) for k in ("arti", "feat", "mvp", "campos", "w2c")]
    if workload == "fauna":
        pairs.append(("class_emb", scene.class_emb.grad, ref["grads"]["class_emb"]))
    nets = [("sdf_mlp", scene.netShape.mlp), ("tex", scene.netTexture), ("dino", scene.netDINO), ("lgt", scene.netLight)]
    if scene.deform:
        nets.append(("deform", scene.netDeform))
    for name, mod in nets:
        pairs += [(f"{name}.{pn}", p.grad, ref["grads"][f"{name}.{pn}"]) for pn, p in mod.named_parameters()]
    _gradients_close(pairs)


@pytest.mark.parametrize("workload,kw,n", [
    ("magicpony", dict(deform=True), 16), ("fauna", {}, 16), ("ponymation", dict(num_frames=8, batch=8), 64),
    # round 4: the reference's real grid class at its real size -- a BCC lattice (Quartet's family) of the "128" class in a random
    # numbering (dmtet.py:214-226, data/tets/generate_tets.py:31-47): the ordered count pass + the sparse emit inside the step
    ("magicpony", dict(deform=True, grid="bcc51s"), 16),
    # BASELINE configs[1] (test_magicpony_horse.yaml:14: forward only, batch 8) at both grid classes, and the training step at the
    # "256" class (config/model/magicpony.yaml:31-33: grid_res 256 ~ Kuhn R = 128); 2-image oracle sample at R = 128, as bench.py does
    ("magicpony", dict(deform=True, batch=8, forward_only=True), 8),
    ("magicpony", dict(deform=True, batch=8, forward_only=True, grid_res=128), 2),
    ("magicpony", dict(deform=True, grid_res=128), 2),
    # round 5: the trained-like mesh (pipeline.synthetic_spikes: what 600 optimiser steps turn the quadruped into; magicpony.yaml:31-33 trains
    # for 1e5) -- big pixel boxes through the rasteriser's tile stage, long silhouettes through the antialiasing
    ("magicpony", dict(deform=True, mesh="spiky"), 16),
], ids=["magicpony", "fauna", "ponymation", "magicpony-bcc51s", "magicpony-fwd-b8", "magicpony-fwd-b8-grid128", "magicpony-grid128", "magicpony-spiky"])
def test_full_size_workloads_stagewise_parity(workload, kw, n, dev):
    """BASELINE configs 2 / 3 / 4 / 5 at FULL size (batch 16 resp. 8 sequences x 8 frames, 256x256, Kuhn R=64 grid, the networks at the
    reference's sizes), fixed weights (no optimiser step before the check): every stage of the step re-done by the CPU oracle from the
    HIP output of the stage before it (oracle/check.py) -- index buffers bit-exact, triangle ids bit-exact on the same clip-space
    vertices, every rendered buffer ('shaded', 'dino_pred', 'flow', the random-view mask) within 1e-4 absolute with NO pixel excluded."""
    from oracle import check

    pipeline = importlib.import_module("3danimals_amd.pipeline")
    kw = dict(kw)
    forward_only = kw.pop("forward_only", False)
    scene = pipeline.SyntheticScene(grid_res=kw.pop("grid_res", 64), batch=kw.pop("batch", 16), resolution=(256, 256), device=dev, seed=0,
                                    workload=workload, **kw)
    want_pass = {"bcc51s": "ordered"}.get(kw.get("grid"), "culled")
    out = scene.step(backward=not forward_only, optimizer_step=False)
    assert scene.netShape.topology._last_count_pass == want_pass
    # (round 5: the headline workload also through the oracle's OWN chain from the SDF values on -- its skinning, its clip matmul, its CPU
    # networks -- on four images: what bench.py's parity.end_to_end and smoke() bound, now in the driver-run suite at full size)
    e2e = workload == "magicpony" and kw == dict(deform=True)
    rep = check.compare_step(scene, out, n_images=4 if e2e else n, end_to_end=e2e)
    if e2e:
        # north_star's bar on the oracle's OWN chain too (round 6; measured on five boxes: 0 .. 7.6e-6 of the frame owner-flipped, 0 .. 3 of the
        # 262144 pixels above 1e-4): the pixels both chains give to the same triangle are all but 1e-4 of the frame, and on them every rendered
        # buffer is within 1e-4 except for at most 1e-5 of the pixels (<= 2 of them).  Those are silhouette pixels, of two kinds: the
        # antialiasing weight is 0.5 - (distance of the edge in pixels) times a colour contrast of O(1), and one ulp of a clip coordinate
        # after two float32 chains is 1e-4 of a pixel (four boxes: worst value 3.8e-5 .. 1.2e-4); and the antialiasing takes the edges of
        # whichever of the two pixels' triangles is nearer in depth, so a tie broken the other way blends -- or does not -- by up to 0.5 (one
        # box: one pixel, 0.447 in alpha).  Their NUMBER is the bound; their size cannot exceed the blend weight.  (Until round 5: 2e-3 / 5e-2.)
        assert rep["frac_pixels_owner_flip"] < 1e-4, (rep["frac_pixels_owner_flip"], rep["end_to_end"])
        assert rep["frac_pixels_gt_1e-4_end_to_end"] <= 1e-5 and rep["max_abs_image_err_end_to_end"] <= 0.5 + 1e-4, rep["end_to_end"]
    assert rep["faces_equal"] and rep["num_faces"] > 8000, rep
    assert rep["max_abs_vert_err"] == 0.0 and rep["max_abs_skin_err"] < 5e-6, rep
    # vertex normals: within 2e-5 of the float32 oracle, or -- on the BCC surface, whose sliver triangles make some sums ill-conditioned
    # (1e-4 between ANY two float32 summation orders) -- as close to the float64 result as the oracle's own float32 is
    n64 = rep["posed_normal_err_vs_float64"]
    assert rep["max_abs_posed_normal_err"] < 2e-5 or (kw.get("grid") and n64["hip"] <= 1.5 * n64["oracle_float32"] + 1e-5), (rep["max_abs_posed_normal_err"], n64)
    assert rep["raster_ids_equal"] and rep["raster"]["max_abs_err"] <= 2e-6, rep["raster"]
    assert rep["gbuffer"]["max_abs_err"] < 1e-5, rep["gbuffer"]
    assert set(rep["images"]) >= {"shaded", "dino_pred"} | ({"flow"} if workload == "ponymation" else set()) | ({"mask_random"} if workload == "fauna" else set())
    assert rep["max_abs_image_err"] < 1e-4 and check.passes(rep), rep["images"]
    assert rep["max_abs_image_err"] < 2e-5, rep["images"]  # measured: <= 1e-5; a regression shows long before the 1e-4 bar


def test_training_step_hot_path_calls_are_pinned(dev):
    """The structure DESIGN.md section 4 describes, pinned: a steady-state magicpony training step at the bench size calls exactly these
    16 hot-path entry points (7 forward, 9 backward; the row gather of the surface-adjacent grid vertices only backward -- ABI 404: the DMTet emit
    leaves the forward rows itself), once each -- no topology launch (the DMTet emit writes the lists), no normals
    launch (they ride in the rasteriser's), no analysis launch (it rides in the compositor's), no shading launch forward (round 4: the
    compositor computes the colour of a covered pixel itself) -- and the forward-only step 7.  Round 6: the clip transform is a call of
    the path (it was a torch bmm each way that nobody counted), and the shading adjoint is launched by the compositor's backward node
    (a3d_shade_bwd_rows: gradients written where the fields / the camera tensors read them)."""
    _lib = importlib.import_module("3danimals_amd._lib")
    pipeline = importlib.import_module("3danimals_amd.pipeline")
    scene = pipeline.SyntheticScene(grid_res=64, batch=16, resolution=(256, 256), device=dev, seed=0, workload="magicpony", deform=True)
    scene.step(backward=True, optimizer_step=True)  # (the first extraction on a grid has no guess at V: topology through the fallback)
    hot = ("a3d_dmtet_", "a3d_skin_", "a3d_bone_", "a3d_normals_", "a3d_mesh_topology", "a3d_rast_", "a3d_interp_", "a3d_cover_",
           "a3d_gbuffer_", "a3d_shade_", "a3d_aa_", "a3d_composite_aa_", "a3d_xfm_")

    def calls(backward):
        with _lib.KernelTimer() as timer:
            scene.step(backward=backward, optimizer_step=backward)
        return {n: c for n, (c, _) in timer.summary().items() if n.startswith(hot)}

    train = calls(True)
    # (round 5: the rasteriser's resolve, the covered-pixel list and the G-buffer rows are ONE launch)
    assert train == {"a3d_dmtet_count": 1, "a3d_dmtet_emit": 1, "a3d_skin_pose_fwd": 1, "a3d_xfm_points_fwd": 1, "a3d_rast_fwd[N16+1][defer]": 1,
                     "a3d_rast_resolve_gbuffer_fwd": 1, "a3d_composite_aa_fwd[C4+C17>16][+shade][+analysis]": 1, "a3d_composite_aa_bwd[C4+C17>16]": 1,
                     "a3d_shade_bwd_rows": 1, "a3d_gbuffer_bwd": 1, "a3d_xfm_points_bwd": 1, "a3d_normals_bwd[B16]": 1, "a3d_skin_pose_bwd": 1,
                     "a3d_dmtet_bwd": 1, "a3d_dmtet_gather_rows": 1, "a3d_gbuffer_prior_grad": 1}, train
    with torch.no_grad():
        fwd = calls(False)
    # (forward only: no graph is wanted, the field is evaluated on the whole grid once -- no surface re-evaluation, no row gather)
    assert sorted(fwd) == sorted(k for k in train if "bwd" not in k and k not in ("a3d_dmtet_gather_rows", "a3d_gbuffer_prior_grad")), fwd


@pytest.mark.parametrize("res,H,W", [(16, 64, 64), (8, 160, 128), (16, 256, 256)])
def test_fused_gbuffer_matches_generic_path_and_gradients(res, H, W, dev, mods, ops):
    """csrc/gbuffer.hip (one kernel forward; gather backward with the rasteriser backward folded in) against the modular
    rasterise/interpolate kernels.  (8, 160, 128): triangles of hundreds of pixels (long same-triangle runs in the pixel list)."""
    B = 3
    verts, faces, _, (mvp, w2c, campos) = _scene(B, res=res, seed=7)
    posed = (verts[None] + 0.05 * seeded((B, *verts.shape), 31, -1, 1)).to(dev)
    tri = faces.to(dev)
    R = mods["render"]
    ru = importlib.import_module("3danimals_amd.model.render.renderutils")

    def run(fused):
        v = posed.clone().requires_grad_(True)
        pv = verts[None].to(dev).clone().requires_grad_(True)
        m = mvp.to(dev).clone().requires_grad_(True)
        nrm = ops.vertex_normals(v, tri)
        clip = ru.xfm_points(v, m)
        rast = ops.rasterize(clip, tri, (H, W))
        pix = torch.nonzero(rast[..., 3].reshape(-1) > 0).squeeze(1)
        if fused:
            gb = ops.gbuffer(clip, v, nrm, pv, rast, tri, pix)
        else:
            fn = R.util.safe_normalize(torch.cross(v[:, tri[:, 1]] - v[:, tri[:, 0]], v[:, tri[:, 2]] - v[:, tri[:, 0]], dim=-1))
            parts = [ops.interpolate(v, rast, tri), ops.interpolate(fn, rast, R._face_index_buffer(tri)), ops.interpolate(nrm, rast, tri),
                     ops.interpolate(pv, rast, tri)]
            gb = torch.cat([t.reshape(-1, 3).index_select(0, pix) for t in parts], -1)
        wgt = seeded(gb.shape, 41, -1, 1).to(dev)
        gv, gp, gm = torch.autograd.grad((gb * wgt).sum(), [v, pv, m])
        return gb.detach(), gv, gp, gm

    a, b = run(True), run(False)
    np.testing.assert_allclose(a[0].cpu().numpy(), b[0].cpu().numpy(), atol=2e-6)
    for x, y, name in zip(a[1:], b[1:], ("v_pos", "prior", "mvp")):
        scale = float(y.abs().max())
        assert scale > 0, name
        np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), rtol=1e-3, atol=2e-4 * scale, err_msg=name)
    c = run(True)  # run to run: only the order of the float atomics differs
    for x, y in zip(a[1:], c[1:]):
        np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), rtol=1e-5, atol=1e-6 * float(y.abs().max()))
    # the product's own pixel list (tile order: neighbouring entries on one triangle are merged before the LDS table) gives the
    # same gradients as the ad-hoc row-major list
    v = posed.clone().requires_grad_(True)
    nrm = ops.vertex_normals(v, tri)
    clip = ru.xfm_points(v, mvp.to(dev))
    rast = ops.rasterize(clip, tri, (H, W))
    pix = ops.covered_pixels(rast)
    gb = ops.gbuffer(clip, v, nrm, verts[None].to(dev), rast, tri, pix)
    order = torch.argsort(pix)
    wgt = seeded((pix.shape[0], 12), 41, -1, 1).to(dev)
    (gv,) = torch.autograd.grad((gb[order] * wgt).sum(), [v])
    np.testing.assert_allclose(gv.cpu().numpy(), a[1].cpu().numpy(), rtol=1e-4, atol=1e-5 * float(a[1].abs().max()))


@pytest.mark.parametrize("E,hw,from_rasteriser", [(0, (64, 64), True), (2, (96, 80), True), (0, (48, 40), True), (3, (64, 64), False)])
def test_cover_list_and_gbuffer_in_one_launch_equal_the_two_launches(E, hw, from_rasteriser, dev, ops, mods):
    """a3d_cover_gbuffer_fwd (list + pixel -> entry map + G-buffer rows [+ extra attribute] from one launch) against a3d_cover_emit followed
    by a3d_gbuffer_fwd: identical list, map and rows (bit for bit), identical gradients -- for buffers whose block counts came from the
    rasteriser's resolve (H*W a multiple of 256), from a3d_cover_count (48x40), and for a buffer that did not come from the rasteriser."""
    B, (H, W) = 3, hw
    verts, faces, _, (mvp, w2c, campos) = _scene(B, res=16, seed=7)
    ru = importlib.import_module("3danimals_amd.model.render.renderutils")
    tri = faces.to(dev)
    posed = (verts[None] + 0.05 * seeded((B, *verts.shape), 31, -1, 1)).to(dev)
    extra0 = seeded((B, verts.shape[0], E), 12, -1, 1).to(dev) if E else None

    def run(fused):
        v = posed.clone().requires_grad_(True)
        pv = verts[None].to(dev).clone().requires_grad_(True)
        m = mvp.to(dev).clone().requires_grad_(True)
        ex = extra0.clone().requires_grad_(True) if E else None
        nrm = ops.vertex_normals(v, tri)
        clip = ru.xfm_points(v, m)
        rast = ops.rasterize(clip, tri, (H, W))
        if not from_rasteriser:
            rast = rast.detach().clone()  # a buffer the rasteriser's caches do not know
        if fused:
            res = ops.covered_gbuffer(clip, v, nrm, pv, rast, tri, extra=ex)
            gb, fl, pix, inv = res if E else (res[0], None, res[1], res[2])
        else:
            pix, inv = ops.covered_pixels(rast, return_inverse=True)
            res = ops.gbuffer(clip, v, nrm, pv, rast, tri, pix, extra=ex)
            gb, fl = res if E else (res, None)
        loss = (gb * seeded(tuple(gb.shape), 41, -1, 1).to(dev)).sum() + ((fl * seeded(tuple(fl.shape), 42, -1, 1).to(dev)).sum() if E else 0)
        grads = torch.autograd.grad(loss, [v, pv, m] + ([ex] if E else []))
        return gb.detach(), None if fl is None else fl.detach(), pix, inv, grads

    a, b = run(True), run(False)
    assert a[2].shape[0] > 100 and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    assert torch.equal(a[0], b[0]) and (not E or torch.equal(a[1], b[1]))
    for x, y in zip(a[4], b[4]):
        scale = float(y.abs().max())
        assert scale > 0
        np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), rtol=1e-5, atol=1e-6 * scale)  # same kernels: atomic order only


def test_fused_gbuffer_backward_table_overflow_falls_back_to_global_atomics(dev, mods, ops):
    """One tiny triangle per pixel, no shared vertices: 256 consecutive list entries reference 768 distinct vertices, more than the 512
    slots of the work-group's LDS table, so part of every block takes the global-atomic fallback of gb_bwd_kernel.  Gradients must still
    equal the modular rasterise/interpolate path."""
    H = W = 32
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    cx, cy = ((xs + 0.5) * 2 / W - 1).reshape(-1), ((ys + 0.5) * 2 / H - 1).reshape(-1)
    r = 0.6 / W  # the triangle holds its own pixel centre only
    offs = torch.tensor([[-1.0, -0.7], [1.0, -0.7], [0.0, 1.1]]) * r
    xy = torch.stack([cx, cy], -1)[:, None, :] + offs[None]  # [HW,3,2]
    V = H * W * 3
    depth = seeded((H * W, 3, 1), 3, 0.1, 0.6)
    verts = torch.cat([xy, depth], -1).reshape(1, V, 3)
    tri = torch.arange(V, dtype=torch.int64).reshape(-1, 3)
    mvp = torch.eye(4)[None]
    R = mods["render"]
    ru = importlib.import_module("3danimals_amd.model.render.renderutils")

    def run(fused):
        v = verts.clone().to(dev).requires_grad_(True)
        pv = (verts * 0.5 + 0.1).to(dev).requires_grad_(True)
        nrm = torch.nn.functional.normalize(seeded((1, V, 3), 8, -1, 1), dim=-1).to(dev).requires_grad_(True)
        clip = ru.xfm_points(v, mvp.to(dev))
        rast = ops.rasterize(clip, tri.to(dev), (H, W))
        assert int((rast[..., 3] > 0).sum()) == H * W  # every pixel owned by its own triangle
        pix = ops.covered_pixels(rast)
        t = tri.to(dev)
        if fused:
            gb = ops.gbuffer(clip, v, nrm, pv, rast, t, pix)
        else:
            fn = R.util.safe_normalize(torch.cross(v[:, t[:, 1]] - v[:, t[:, 0]], v[:, t[:, 2]] - v[:, t[:, 0]], dim=-1))
            parts = [ops.interpolate(v, rast, t), ops.interpolate(fn, rast, R._face_index_buffer(t)), ops.interpolate(nrm, rast, t),
                     ops.interpolate(pv, rast, t)]
            gb = torch.cat([p.reshape(-1, 3).index_select(0, pix) for p in parts], -1)
        wgt = seeded((pix.shape[0], 12), 41, -1, 1).to(dev)
        return (gb.detach(),) + torch.autograd.grad((gb * wgt).sum(), [v, nrm, pv])

    a, b = run(True), run(False)
    np.testing.assert_allclose(a[0].cpu().numpy(), b[0].cpu().numpy(), atol=2e-6)
    for x, y, name in zip(a[1:], b[1:], ("v_pos", "v_nrm", "prior")):
        scale = float(y.abs().max())
        assert scale > 0, name
        np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), rtol=1e-3, atol=2e-4 * scale, err_msg=name)



@pytest.mark.parametrize("with_partner", [True, False])
def test_normals_riding_in_the_rasteriser_launch_equal_the_stand_alone_pass(with_partner, dev, ops, mods):
    """ops.rasterize(normals_job=...) runs normals.hip's forward pass as extra work-groups of the triangle launch (batches of 4 index
    rows instead of 8, for the registers): same normals and same raster buffer bit for bit, same gradients; a job the launch cannot take
    (another triangle list) is left not done."""
    verts, faces = quadruped_mesh(20)
    g = torch.Generator().manual_seed(5)
    B = 3
    v_a = (verts[None] + 0.01 * torch.randn(B, *verts.shape, generator=g)).to(dev).requires_grad_(True)
    v_b = verts[None].clone().to(dev).requires_grad_(True) if with_partner else None
    tri = faces.to(dev)
    mvp = mods["synthetic"].random_cameras(B, seed=2)
    mvp = (mvp[0] if isinstance(mvp, (tuple, list)) else mvp).to(dev)
    clip = torch.cat([v_a.detach(), torch.ones_like(v_a[..., :1])], -1) @ mvp.transpose(1, 2)
    ref_rast = ops.rasterize(clip, tri, (64, 64))
    if with_partner:
        ref_a, ref_b = ops.vertex_normals_pair(v_a, v_b, tri)
    else:
        ref_a, ref_b = ops.vertex_normals(v_a, tri), None
    job = ops.NormalsJob(v_a, v_b, tri)
    rast = ops.rasterize(clip, tri, (64, 64), normals_job=job)
    assert job.done and torch.equal(rast, ref_rast)
    n_a, n_b = ops.vertex_normals_attach(v_a, v_b, job)
    assert torch.equal(n_a, ref_a) and (n_b is None) == (ref_b is None) and (n_b is None or torch.equal(n_b, ref_b))
    w_a, w_b = torch.randn(n_a.shape, generator=g).to(dev), torch.randn(1, *verts.shape, generator=g).to(dev)
    loss = lambda a, b: (a * w_a).sum() + (0 if b is None else (b * w_b).sum())
    ins = [v_a] + ([v_b] if with_partner else [])
    got = torch.autograd.grad(loss(n_a, n_b), ins)
    want = torch.autograd.grad(loss(ref_a, ref_b), ins)
    for x, y in zip(got, want):
        assert torch.equal(x, y)
    other = ops.NormalsJob(v_a, v_b, tri.flip(0).contiguous())  # a different triangle list than the one being rasterised
    ops.rasterize(clip, tri, (64, 64), normals_job=other)
    assert not other.done


def test_render_mesh_takes_the_pending_normals_into_the_rasteriser_launch(dev, ops, mods):
    """make_mesh leaves auto_normals pending; render_mesh hands them to the rasteriser: no a3d_normals_fwd* call, the raster call carries
    the job (and the canonical mesh's normals with it), images and vertex gradients equal those of the path with the stand-alone launch."""
    _lib = importlib.import_module("3danimals_amd._lib")
    M, render = mods["mesh"], mods["render"]
    B, H, W = 2, 32, 32
    verts, faces, _, (mvp, w2c, campos) = _scene(B, seed=5)
    uvs = torch.zeros(1, 4, 2, device=dev)
    uvi = torch.zeros(1, faces.shape[0], 3, dtype=torch.int64, device=dev)
    tri = faces[None].to(dev)

    def run(ride):
        M.RIDE_NORMALS = ride
        try:
            posed = (verts[None] + 0.1 * seeded((B, *verts.shape), 21, -1, 1)).to(dev).requires_grad_(True)
            prior = M.make_mesh(verts[None].to(dev), tri, uvs, uvi, None)
            shape = M.make_mesh(posed, tri, uvs.expand(B, -1, -1), uvi, None)
            with _lib.KernelTimer() as timer:
                out = render.render_mesh(None, shape, mvp.to(dev), w2c.to(dev), campos.to(dev), None, None, (H, W), bsdf="diffuse",
                                         render_modes=["shaded", "geo_normal"], prior_mesh=prior)
            (g,) = torch.autograd.grad(sum((o * seeded(tuple(o.shape), 7 + i, -1, 1).to(dev)).sum() for i, o in enumerate(out)), posed)
            return out, g, set(timer.summary()), prior
        finally:
            M.RIDE_NORMALS = True

    out_ride, g_ride, names_ride, prior = run(True)
    out_alone, g_alone, names_alone, _ = run(False)
    assert not any(n.startswith("a3d_normals_fwd") for n in names_ride), names_ride
    assert any(n.startswith(f"a3d_rast_fwd[N{B}+1]") for n in names_ride), names_ride  # (+ "[defer]": the resolve rides in the G-buffer launch)
    assert any(n.startswith("a3d_normals_fwd") for n in names_alone) and any(n in ("a3d_rast_fwd", "a3d_rast_fwd[defer]") for n in names_alone), names_alone
    assert prior._v_nrm is not None  # the canonical mesh's normals came out of the same launch
    # (the antialiased image: pixels that take two blends add them in the order the atomics land, 1 ulp run to run -- DESIGN.md section 2,
    # test_antialias_run_to_run_difference_is_rounding_only; the un-antialiased geo_normal image is bit-equal)
    assert float((out_ride[0] - out_alone[0]).abs().max()) <= 2.4e-7 and torch.equal(out_ride[1], out_alone[1])
    # (the backward scatters with float atomics: equal up to their order, run to run)
    np.testing.assert_allclose(g_ride.cpu().numpy(), g_alone.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(g_alone.abs().max()))


@pytest.mark.parametrize("seed,B,hw,res,zoom", [(0, 1, (64, 64), 12, 1.0), (1, 3, (40, 72), 16, 1.0), (2, 2, (128, 96), 20, 1.0), (3, 5, (24, 24), 12, 1.0),
                                                (4, 2, (30, 50), 16, 3.5), (5, 4, (256, 256), 28, 1.6), (6, 1, (8, 8), 8, 1.0),
                                                (7, 3, (96, 96), -20, 1.0), (8, 2, (256, 256), -28, 1.6)])
def test_every_launch_folding_switch_off_gives_the_same_frames_and_gradients(seed, B, hw, res, zoom, dev, ops, mods, monkeypatch):
    """SDF -> DMTet -> make_mesh -> render_mesh -> loss -> backward, three consecutive frames of a moving surface, with everything this
    round folded into other launches switched ON (culled count, lists from the emit launch, speculative emit, normals in the rasteriser
    launch, covered-pixel list with the G-buffer, the analysis in the compositor, fused compositor) against everything OFF (the modular
    entry points one by one).  Same meshes bit for bit, same images up to the antialiasing's blend order, same gradients on the SDF and
    the pose offsets up to the order of the float atomics -- whatever the batch, the image shape (incl. no multiple of 8, and a surface
    that reaches far outside the frustum: zoom) and the grid.  res < 0 (round 4): the Kuhn grid of -res cells in a RANDOM numbering (rows
    shuffled, row entries permuted: a randomly wound surface) -- the ordered count pass + the sparse emit against the streaming pass."""
    M, R, D = mods["mesh"], mods["render"], mods["dmtet"]
    H, W = hw
    pos, tets = kuhn(abs(res))
    if res < 0:
        tg = importlib.import_module("3danimals_amd.tetgrid")
        p_np, t_np = tg.scramble(pos.numpy(), tets.numpy(), 3 + seed)
        pos, tets = torch.from_numpy(p_np), torch.from_numpy(t_np).long()
    pos_d = pos.to(dev)
    centre, ext = pos.mean(0), float((pos.amax(0) - pos.amin(0)).max())
    synthetic = importlib.import_module("3danimals_amd.synthetic")
    mvp, w2c, campos = (t.to(dev) for t in synthetic.random_cameras(B, seed=seed))
    switches = [(ops, "DMTET_CULL_MIN_VERTS", 0, 1 << 30), (ops, "DMTET_EMIT_LISTS", True, False), (ops, "DMTET_SPECULATIVE_EMIT", True, False),
                (ops, "DMTET_TOPOLOGY", True, False), (M, "RIDE_NORMALS", True, False), (R, "FUSED_COVER_GBUFFER", True, False),
                (R, "DEFER_ANALYSIS", True, False), (R, "FUSED_COMPOSITE", True, False), (R, "SHADE_IN_COMPOSITOR", True, False),
                (R, "FUSED_MASK_RENDER", True, False), (R, "DEFER_RESOLVE", True, False), (R, "ALIAS_POSITIONS", True, False)]

    def run(on):
        """on: True / False = every switch on / off; a tuple of booleans = one setting per switch (mixed)."""
        for i, (mod, name, a, b) in enumerate(switches):
            monkeypatch.setattr(mod, name, a if (on if isinstance(on, bool) else on[i]) else b)
        grid = D.TetGridTopology(tets.to(dev), positions=pos_d)
        frames = []
        for t in range(3):
            sdf = ((0.30 + 0.04 * t) * ext - (pos - centre).norm(dim=-1) + 0.03 * ext * seeded((pos.shape[0],), 50 + seed, -1, 1)).to(dev).requires_grad_(True)
            verts, faces, uv_idx = ops.dmtet(pos_d, sdf, grid)
            V = verts.shape[0]
            offs = (0.02 * ext * seeded((B, V, 3), 60 + seed + t, -1, 1)).to(dev).requires_grad_(True)
            posed = (verts[None] - centre.to(dev)) * (2.0 * zoom / ext) + offs
            uvs = torch.zeros(1, 4, 2, device=dev)
            uvi = torch.zeros(1, faces.shape[0], 3, dtype=torch.int64, device=dev)
            prior = M.make_mesh(((verts[None] - centre.to(dev)) * (2.0 * zoom / ext)), faces[None], uvs, uvi, None)
            shape = M.make_mesh(posed, faces[None], uvs.expand(B, -1, -1), uvi, None)
            out = R.render_mesh(None, shape, mvp, w2c, campos, None, None, (H, W), bsdf="diffuse", render_modes=["shaded", "geo_normal"],
                                prior_mesh=prior)
            loss = sum((o * seeded(tuple(o.shape), 70 + i, -1, 1).to(dev)).sum() for i, o in enumerate(out))
            g_sdf, g_offs = torch.autograd.grad(loss, [sdf, offs])
            frames.append((verts.detach(), faces, [o.detach() for o in out], g_sdf, g_offs))
        if isinstance(on, bool):
            assert grid._last_count_pass == (("ordered" if res < 0 else "culled") if on else "plain")
        return frames

    a, b = run(True), run(False)
    # (round 5) ... and MIXED settings: the switches are independent knobs of a deployment (A3D_* environment variables), so two
    # random subsets per case go through the same comparison -- a switch that only works next to another one would show here
    rng = np.random.RandomState(100 + seed)
    # (round 6: every case, and the clip transform's aliases -- ALIAS_POSITIONS -- are one of the knobs)
    mixed = [run(tuple(bool(v) for v in rng.randint(0, 2, len(switches)))) for _ in range(2)]
    for other in [a] + mixed:
      for t, (fa, fb) in enumerate(zip(other, b)):
          assert fa[1].shape[0] > 50 and torch.equal(fa[0], fb[0]) and torch.equal(fa[1], fb[1]), t
          for x, y in zip(fa[2], fb[2]):
              assert float((x - y).abs().max()) <= 5e-7, (t, float((x - y).abs().max()))
          for x, y in zip(fa[3:], fb[3:]):
              scale = float(y.abs().max())
              assert scale > 0
              np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), rtol=2e-4, atol=2e-5 * scale)


@pytest.mark.parametrize("numbering", ["spatial", "random"])
def test_speculative_dmtet_emit_equals_the_exact_one(numbering, dev, ops, mods, monkeypatch):
    """ops.dmtet_extract enqueues the emit launch BEFORE the host has read the counts, with buffers and grid sized by the previous
    extraction on the grid + 25 %; the kernel takes the true sizes from the device.  A sequence of surfaces that shrink, grow slowly,
    outgrow the guess (the launch must then leave everything untouched and the exact path take over), vanish and come back: every output
    equal to the exact path's, bit for bit, and the mesh topology usable either way."""
    monkeypatch.setattr(ops, "DMTET_CULL_MIN_VERTS", 0)
    pos, tets = kuhn(40)
    if numbering == "random":  # the ordered count pass + a3d_dmtet_emit_sparse: its speculative form has no block capacities
        tg = importlib.import_module("3danimals_amd.tetgrid")
        p_np, t_np = tg.scramble(pos.numpy(), tets.numpy(), 17)
        pos, tets = torch.from_numpy(p_np), torch.from_numpy(t_np).long()
    pos_d = pos.to(dev)
    T = mods["dmtet"].TetGridTopology
    spec_topo, exact_topo = T(tets.to(dev), positions=pos_d), T(tets.to(dev), positions=pos_d)
    radii = (2.0, 1.9, 2.1, 1.2, 3.1, 3.0, -1.0, 2.5, 2.45)  # (-1: no surface at all)
    took = []
    for trial, r in enumerate(radii):
        sdf = (r - pos.norm(dim=-1) + 0.05 * seeded((pos.shape[0],), 90 + trial, -1, 1)).to(dev)
        monkeypatch.setattr(ops, "DMTET_SPECULATIVE_EMIT", True)
        a = ops.dmtet_extract(pos_d, sdf, spec_topo, surface_vertices=True, surface_points=512)
        tri_a = ops._tri32_cache.peek(a[1]) if a[1].shape[0] else None
        monkeypatch.setattr(ops, "DMTET_SPECULATIVE_EMIT", False)
        b = ops.dmtet_extract(pos_d, sdf, exact_topo, surface_vertices=True, surface_points=512)
        for x, y in zip(a, b):
            assert x.shape == y.shape and torch.equal(x, y), (trial, r)
        # (round 6) the surface rows the emit launch leaves for the SDF re-evaluation: pos[idx], zero rows up to the bucket
        idx, pts = a[4], a[5]
        assert pts.shape == (-(-idx.shape[0] // 512) * 512, 3) and torch.equal(pts[: idx.shape[0]], pos_d[idx])
        assert pts.shape[0] == idx.shape[0] or float(pts[idx.shape[0]:].abs().max()) == 0.0
        assert torch.equal(pts, ops.gather_rows_padded(pos_d, idx, pts.shape[0]))
        took.append(a[0].shape[0] > 0 and a[0].untyped_storage().nbytes() > a[0].numel() * 4)  # a view into a larger buffer = speculative
        if a[1].shape[0]:
            n_a = ops.vertex_normals(a[0][None], a[1])
            n_b = ops.vertex_normals(b[0][None], b[1])
            assert torch.equal(n_a, n_b) and tri_a is not None and torch.equal(tri_a.long(), a[1])
    # first extraction: no guess; 1, 2, 3: within the guess; 4: outgrown; 5: fits again; 6: empty; 7: the guess was an empty mesh; 8: fits
    assert took == [False, True, True, True, False, True, False, False, True], took


@pytest.mark.parametrize("grid", ["kuhn20", "bcc", "delaunay"])
def test_emitted_vertex_face_lists_never_overflow_their_stride_even_on_noise(grid, dev, ops, mods):
    """The fixed-stride lists of the DMTet emit rest on a bound: a surface vertex sits on a grid edge, every tet around that edge gives
    at most two triangles at it, so valence <= 2 x (most tets around one edge) = TetGridTopology.face_list_stride() (rounded up).  White
    noise as the SDF is the worst case (every tet a surface tet): no list is longer than the stride, every list complete, the normals
    through them equal those through the stand-alone CSR lists bit for bit."""
    a3d_pkg = importlib.import_module("3danimals_amd")
    if grid.startswith("kuhn"):
        pos, tets = kuhn(int(grid[4:]))
    elif grid == "bcc":
        p, t = a3d_pkg.tetgrid.bcc_grid(9, seed=2)
        pos, tets = torch.from_numpy(p), torch.from_numpy(t).long()
    else:
        p, t = a3d_pkg.tetgrid.delaunay_grid(1500, seed=4)
        pos, tets = torch.from_numpy(p), torch.from_numpy(t).long()
    topo = mods["dmtet"].TetGridTopology(tets.to(dev))
    stride = topo.face_list_stride()
    most = int(torch.bincount(topo.tet2edge32.reshape(-1).long()).max())
    assert stride % 8 == 0 and 2 * most <= stride < 2 * most + 8
    pos_d = pos.to(dev)
    seen = 0
    for trial in range(3):
        sdf = torch.randn(pos.shape[0], generator=torch.Generator().manual_seed(70 + trial)).to(dev)
        verts, faces, _, _ = ops.dmtet_extract(pos_d, sdf, topo)  # (the first extraction on a grid has no guess at V: CSR fallback)
        V, F = verts.shape[0], faces.shape[0]
        tri32 = ops.tri_int32(faces)
        adj = ops._adj_cache.peek(tri32)
        ref_adj, _ = ops.mesh_topology(tri32.clone(), V)
        valence = torch.diff(ref_adj.off.long())
        assert int(valence.max()) <= 2 * most
        if adj is None or not adj.stride:
            continue
        seen += 1
        assert adj.stride == stride and torch.equal(adj.off[:V].long(), valence)
        vv = verts[None].clone().requires_grad_(True)
        n1, n2 = ops._Normals.apply(vv, tri32, adj), ops._Normals.apply(vv, tri32, ref_adj)
        assert torch.equal(n1, n2)
        w = seeded((1, V, 3), 9, -1, 1).to(dev)
        assert torch.equal(torch.autograd.grad((n1 * w).sum(), vv)[0], torch.autograd.grad((n2 * w).sum(), vv)[0])
    assert seen >= 1 or stride > ops.DMTET_EMIT_LISTS_MAX_STRIDE


@pytest.mark.parametrize("grid", ["kuhn24", "kuhn64", "delaunay3k"])
def test_topology_built_inside_the_dmtet_extraction_equals_the_stand_alone_one(grid, dev, ops, mods):
    """ops.dmtet_extract leaves the mesh topology in the caches (emit launch: int32 list + the vertex -> face lists, stride slots per
    vertex; or, when its guess at V was too small, valence counts + ONE finalize launch: offsets + unsorted CSR lists) -- against
    a3d_mesh_topology on the same list: list lengths / offsets bit for bit, every vertex's list the same SET, the
    opposite-vertex table a walk over the lists gives == the hash's, and vertex normals / their gradient the same BITS through either
    and through lists stored in reverse (the kernels order the keys in registers).  Several extractions in a row with different SDFs:
    the two count arrays alternate and zero each other."""
    from oracle import dmtet_ref

    a3d_pkg = importlib.import_module("3danimals_amd")
    if grid.startswith("kuhn"):
        pos, tets = kuhn(int(grid[4:]))
    else:
        v, t = a3d_pkg.tetgrid.delaunay_grid(3000, seed=5)
        pos, tets = torch.from_numpy(v) * 7.0, torch.from_numpy(t)
    topo = mods["dmtet"].TetGridTopology(tets.to(dev))
    pos_d = pos.to(dev)
    ops._topo_sets.clear()
    layouts = set()
    for trial, radius in enumerate((2.2, 1.4, 2.9, 2.25)):
        sdf = (radius - pos.norm(dim=-1) + 0.1 * seeded((pos.shape[0],), 40 + trial, -1, 1)).to(dev)
        verts, faces, _, _ = ops.dmtet_extract(pos_d, sdf, topo)
        V, F = verts.shape[0], faces.shape[0]
        assert F > 100
        tri32 = ops.tri_int32(faces)
        assert tri32.dtype == torch.int32 and torch.equal(tri32.long(), faces)  # written by the emit launch, found through the cache
        adj, aat = ops._adj_cache.peek(tri32), ops._topo_cache.peek(tri32)
        if V > importlib.import_module("3danimals_amd._lib").lib().a3d_mesh_topology_finalize_max_vertices():
            assert adj is None and aat is None  # too many vertices for the LDS scan: the stand-alone path builds it on first use
            continue
        assert adj is not None and aat is not None and not adj.sorted and aat.opp is None and aat.lists is adj
        ref_adj, ref_aat = ops.mesh_topology(tri32.clone(), V)
        layouts.add(adj.stride)
        ref_off = ref_adj.off.cpu().numpy()
        b = ref_adj.adj[: 3 * F].cpu().numpy()
        assert np.array_equal(b, np.sort(b)[np.argsort(np.argsort(b))])
        seg = np.repeat(np.arange(V), np.diff(ref_off))
        if adj.stride:  # written by the emit launch itself: list v = adj[v * stride ..], off[v] = its length; the grid bounds the valence
            assert adj.stride == topo.face_list_stride() and adj.stride >= int(np.diff(ref_off).max())
            assert np.array_equal(adj.off[:V].cpu().numpy(), np.diff(ref_off))
            slots = adj.adj[: V * adj.stride].cpu().numpy().reshape(V, adj.stride)  # (a speculative emit sized the buffer by a guess at V)
            a = np.concatenate([slots[v, : ref_off[v + 1] - ref_off[v]] for v in range(V)])
        else:  # valence counts from the emit launch, offsets + lists from ONE finalize launch (CSR)
            assert torch.equal(adj.off, ref_adj.off)
            a = adj.adj[: 3 * F].cpu().numpy()
        assert np.array_equal(a[np.lexsort((a, seg))], b[np.lexsort((b, seg))])  # every vertex's list: the same set
        # no hash and no opposite-vertex table on this path: the silhouette analysis walks the lists -- the walk gives the hash's table
        assert torch.equal(ops.opposite_vertices_from_lists(adj), ref_aat.opp) and torch.equal(ops.opposite_vertices_from_lists(ref_adj), ref_aat.opp)
        vv = (verts[None] + 0.03 * seeded((3, V, 3), 7, -1, 1).to(dev)).requires_grad_(True)
        n1 = ops._Normals.apply(vv, tri32, adj)
        n2 = ops._Normals.apply(vv, tri32, ref_adj)
        shuffled = ops.VertexFaceAdjacency(tri32, V, build=False)  # the sorted lists, every list reversed in storage
        shuffled.off, shuffled.sorted = ref_adj.off, False
        start = torch.repeat_interleave(ref_adj.off[:-1].long(), torch.diff(ref_adj.off.long()))
        end = torch.repeat_interleave(ref_adj.off[1:].long(), torch.diff(ref_adj.off.long()))
        shuffled.adj = ref_adj.adj[: 3 * F][(start + end - 1 - torch.arange(3 * F, device=dev)).long()].contiguous()
        assert torch.equal(n1, n2) and torch.equal(n1, ops._Normals.apply(vv, tri32, shuffled))
        w = seeded((3, V, 3), 8, -1, 1).to(dev)
        assert torch.equal(torch.autograd.grad((n1 * w).sum(), vv)[0], torch.autograd.grad((n2 * w).sum(), vv)[0])
        if grid == "kuhn24" and trial == 0:  # and against the reference's own index buffers through the oracle
            _, f_ref, _, _ = dmtet_ref.marching_tets(pos, sdf.cpu(), tets)
            assert torch.equal(f_ref, faces.cpu())
    # the first extraction on a grid has no guess at V for the counters (finalize launch, CSR); trial 2 outgrows the guess of trial 1
    assert len(layouts) == 2 or grid == "delaunay3k", layouts
    assert len(ops._topo_sets) >= 1


def test_mesh_topology_fused_entry_point_equals_the_two_separate_ones(dev, ops):
    """a3d_mesh_topology against a3d_normals_adjacency + a3d_aa_topology, bit for bit (incl. an empty list and isolated vertices)."""
    verts, faces = quadruped_mesh(16, 0.3)
    for tri, V in ((faces, verts.shape[0]), (faces[:37], verts.shape[0] + 5), (faces[:0], 4)):
        tri32 = tri.to(dev).to(torch.int32).contiguous()
        a1, t1 = ops.VertexFaceAdjacency(tri32, V), ops.AATopology(tri32, V)
        a2, t2 = ops.mesh_topology(tri32.clone(), V)
        F = tri32.shape[0]
        assert torch.equal(a1.off, a2.off) and torch.equal(a1.adj[: 3 * F], a2.adj[: 3 * F]) and torch.equal(t1.opp, t2.opp)
    # and the caches hand the fused result to both consumers
    tri32 = faces.to(dev).to(torch.int32).contiguous()
    adj = ops.vertex_face_adjacency(tri32, verts.shape[0])
    assert ops.aa_topology(tri32, verts.shape[0]) is ops._topo_cache.peek(tri32) and adj is ops._adj_cache.peek(tri32)


def test_ddp_wrapped_step_single_rank_nccl(dev):
    """bench.py's N>1 code path (DistributedDataParallel over RCCL) exercised with a 1-rank process group: the wrapped
    step must give exactly the gradients of the plain step (custom Functions return grads for every parameter)."""
    import socket

    import torch.distributed as dist

    pipeline = importlib.import_module("3danimals_amd.pipeline")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        kw = dict(grid_res=16, batch=2, resolution=(64, 64), device=dev, seed=5, net_width=32, net_layers=3, feat_dim=16, embedder_freq=4, jitter_grid=0.0)
        plain, wrapped = pipeline.SyntheticScene(**kw), pipeline.SyntheticScene(**kw)
        ddp = torch.nn.parallel.DistributedDataParallel(wrapped, device_ids=[dev.index], broadcast_buffers=False, gradient_as_bucket_view=True)
        a = plain.step(backward=True, optimizer_step=False, sdf_reg=False)
        b = wrapped.step(backward=True, optimizer_step=False, sdf_reg=False, module=ddp)
        assert float(a["loss"]) == pytest.approx(float(b["loss"]), rel=1e-5)
        for (n, p), (_, q) in zip(plain.named_parameters(), wrapped.named_parameters()):
            assert q.grad is not None, n
            scale = float(p.grad.abs().max()) + 1e-12
            assert float((p.grad - q.grad).abs().max()) <= 2e-3 * scale, n  # atomics reorder float sums run to run
    finally:
        dist.destroy_process_group()


def test_surface_only_sdf_backward_is_output_identical(dev, mods):
    """DMTetGeometry evaluates the SDF MLP with a graph only on surface-adjacent grid vertices: same mesh, same gradients."""
    a3d = importlib.import_module("3danimals_amd")
    grid = a3d.tetgrid.kuhn_grid(16)

    def run(flag):
        torch.manual_seed(0)
        geo = mods["dmtet"].DMTetGeometry(32, 7.0, num_layers=3, hidden_size=32, embedder_freq=4, init_sdf="ellipsoid", jitter_grid=0.0,
                                          symmetrize=True, device=dev, tet_grid=grid, surface_only_backward=flag).to(dev)
        m = geo.getMesh(jitter_grid=False)
        wgt = seeded(m.v_pos.shape, 3, -1, 1).to(dev)
        loss = (m.v_pos * wgt).sum() + (m.v_nrm * wgt).sum()
        loss.backward()
        return m, [p.grad.clone() for p in geo.mlp.parameters()], geo.current_sdf.detach()

    (m1, g1, s1), (m0, g0, s0) = run(True), run(False)
    assert torch.equal(m1.v_pos, m0.v_pos) and torch.equal(m1.t_pos_idx, m0.t_pos_idx) and torch.equal(s1, s0)
    for a, b in zip(g1, g0):
        scale = float(b.abs().max())
        assert scale > 0 and float((a - b).abs().max()) <= 2e-3 * scale
    # (round 6) the rows of the re-evaluation written by the DMTet emit launch (dmtet.SURFACE_POINTS_IN_EMIT) against the gather launch:
    # the same rows, so the same mesh and the same gradients up to the order of the backward's float atomics
    _lib = importlib.import_module("3danimals_amd._lib")
    D = mods["dmtet"]
    assert D.SURFACE_POINTS_IN_EMIT
    with _lib.KernelTimer() as t_on:
        run(True)
    D.SURFACE_POINTS_IN_EMIT = False
    try:
        with _lib.KernelTimer() as t_off:
            m2, g2, s2 = run(True)
    finally:
        D.SURFACE_POINTS_IN_EMIT = True
    count = lambda t: sum(c for n, (c, _) in t.summary().items() if n.startswith("a3d_dmtet_gather_rows"))
    assert count(t_on) == 1 and count(t_off) == 2  # (backward only | forward and backward)
    assert torch.equal(m2.v_pos, m1.v_pos) and torch.equal(s2, s1)
    for a, b in zip(g2, g1):
        assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max())


def test_nvdiffrast_shim_end_to_end(dev):
    """The reference's own call pattern (render.py:292-294, 24, 264-267) through the nvdiffrast.torch stand-in."""
    import sys

    from oracle import raster_ref

    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3danimals_amd", "shims")
    sys.path.insert(0, shim)
    try:
        dr = importlib.import_module("nvdiffrast.torch")
        ops_mod = importlib.import_module("3danimals_amd.ops")
        B, H, W = 2, 48, 48
        verts, faces, clip, _ = _scene(B)
        ctx = dr.RasterizeGLContext()
        pos, tri = clip.to(dev).requires_grad_(True), faces.to(dev).int()
        with dr.DepthPeeler(ctx, pos.float(), tri, [H, W]) as peeler:
            rast, db = peeler.rasterize_next_layer()
        rast2, db2 = dr.rasterize(ctx, pos, tri, [H, W])
        assert db._value is None  # rast_db is computed on first use only (the reference discards it: render.py:24)
        assert torch.equal(rast, rast2) and db.shape == rast.shape
        dense = ops_mod.rasterize_db(pos, tri, rast)
        assert torch.equal(db.materialize(), dense) and torch.equal(torch.nn.functional.avg_pool2d(db2.permute(0, 3, 1, 2), 1).permute(0, 2, 3, 1), dense)
        _, da = dr.interpolate(verts[None].to(dev).contiguous(), rast, tri, rast_db=db2, diff_attrs="all")
        assert da.shape == (B, H, W, 6) and bool(torch.isfinite(da).all()) and float(da.abs().max()) > 0
        assert np.array_equal(rast.detach().cpu().numpy(), raster_ref.rasterize(clip, faces.int(), (H, W)).numpy())
        attr = verts[None].to(dev)
        out, _ = dr.interpolate(attr.contiguous(), rast, tri, rast_db=None, diff_attrs=None)
        np.testing.assert_allclose(out.detach().cpu().numpy(), raster_ref.interpolate(verts[None], rast.detach().cpu(), faces.int()).numpy(), atol=1e-6)
        col = torch.lerp(torch.zeros(B, H, W, 4, device=dev), torch.ones(B, H, W, 4, device=dev), (rast[..., 3:] > 0).float())
        aa = dr.antialias(col.contiguous().float(), rast.float(), pos.float(), tri)
        ref = raster_ref.antialias(col.cpu(), rast.detach().cpu(), clip, faces.int())
        np.testing.assert_allclose(aa.detach().cpu().numpy(), ref.numpy(), atol=2e-6)
        aa.sum().backward()
        assert pos.grad is not None and float(pos.grad.abs().max()) > 0
    finally:
        sys.path.remove(shim)


def test_nvdiffrast_shim_range_mode_and_gradient_boost(dev):
    """The rest of the operator signature: range mode (one shared vertex array, image b renders tri[first : first + count], ids index
    the full list) for rasterize / DepthPeeler / interpolate / antialias against the oracle image by image, and pos_gradient_boost
    (the gradient to pos through the silhouette blends multiplied, nothing else changed)."""
    import sys

    from oracle import raster_ref

    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3danimals_amd", "shims")
    sys.path.insert(0, shim)
    try:
        dr = importlib.import_module("nvdiffrast.torch")
        H, W = 48, 56
        verts, faces, clip, _ = _scene(1)
        F = faces.shape[0]
        ranges = torch.tensor([[0, F // 2], [F // 2, F - F // 2], [5, 0]], dtype=torch.int32)
        pos = clip[0].to(dev).requires_grad_(True)  # [V,4]
        tri = faces.to(dev).int()
        ctx = dr.RasterizeGLContext()
        rast, _ = dr.rasterize(ctx, pos, tri, [H, W], ranges=ranges)
        assert rast.shape == (3, H, W, 4) and float(rast[2].abs().max()) == 0.0
        for b, (first, count) in enumerate(ranges.tolist()[:2]):
            ref = raster_ref.rasterize(clip, faces[first:first + count].int(), (H, W))[0]
            ref[..., 3] = torch.where(ref[..., 3] > 0, ref[..., 3] + first, ref[..., 3])
            assert np.array_equal(rast[b].detach().cpu().numpy(), ref.numpy()), b
        with dr.DepthPeeler(ctx, pos, tri, [H, W], ranges=ranges) as peeler:
            l0, _ = peeler.rasterize_next_layer()
            l1, _ = peeler.rasterize_next_layer()
        assert torch.equal(l0, rast)
        both = (l0[..., 3] > 0) & (l1[..., 3] > 0)
        assert int(both.sum()) > 50 and bool((l1[..., 2][both] >= l0[..., 2][both]).all()) and bool((l1[..., 3][both] != l0[..., 3][both]).all())
        out, _ = dr.interpolate(verts.to(dev), rast, tri)  # attr [V,3]
        assert out.shape == (3, H, W, 3) and float(out[2].abs().max()) == 0.0 and float(out[:2].abs().max()) > 0
        col = torch.lerp(torch.zeros(3, H, W, 3, device=dev), torch.ones(3, H, W, 3, device=dev), (rast[..., 3:] > 0).float()).contiguous()
        aa = dr.antialias(col, rast, pos, tri)
        (g1,) = torch.autograd.grad(aa.sum(), pos)
        aa3 = dr.antialias(col, rast, pos, tri, pos_gradient_boost=3.0)
        (g3,) = torch.autograd.grad(aa3.sum(), pos)
        # (two analyses of one frame list their records in the order their waves' atomics landed: a pixel that takes two blends may sum them either way)
        assert float((aa - aa3).abs().max()) <= 2.4e-7 and float(g1.abs().max()) > 0
        torch.testing.assert_close(g3, 3.0 * g1, rtol=1e-5, atol=1e-6 * float(g1.abs().max()))
        with pytest.raises(RuntimeError, match="range mode"):
            dr.rasterize(ctx, pos, tri, [H, W], ranges=ranges.to(dev))
    finally:
        sys.path.remove(shim)


def test_render_mesh_spp2_runs_and_matches_generic_semantics(dev, mods):
    """spp > 1 takes the generic (dense, torch-resampled) path like the reference; output shapes and ranges are sane."""
    B, H, W = 2, 32, 32
    verts, faces, _, (mvp, w2c, campos) = _scene(B, seed=9)
    M = mods["mesh"]
    uvs = torch.zeros(1, 4, 2, device=dev)
    uvi = torch.zeros(1, faces.shape[0], 3, dtype=torch.int64, device=dev)
    shape = M.make_mesh(verts[None].expand(B, -1, -1).contiguous().to(dev), faces[None].to(dev), uvs.expand(B, -1, -1), uvi, None)
    out1 = mods["render"].render_mesh(None, shape, mvp.to(dev), w2c.to(dev), campos.to(dev), None, None, (H, W), spp=1, msaa=True,
                                      bsdf="diffuse", render_modes=["shaded"])[0]
    out2 = mods["render"].render_mesh(None, shape, mvp.to(dev), w2c.to(dev), campos.to(dev), None, None, (H, W), spp=2, msaa=True,
                                      bsdf="diffuse", render_modes=["shaded"])[0]
    assert out1.shape == out2.shape == (B, 4, H, W)
    assert float(out2.min()) >= 0.0 and float(out2.max()) <= 1.0 + 1e-5
    # supersampled coverage agrees with the 1-spp coverage away from the silhouette
    assert float((out1[:, 3] - out2[:, 3]).abs().mean()) < 0.05


def test_dmtet_geometry_loads_reference_npz_and_sequence_skinning(tmp_path, dev, mods):
    """DMTetGeometry reads data/tets/{res}_tets.npz in the reference's format; Ponymation-style [B,F] skinning + flow render."""
    a3d = importlib.import_module("3danimals_amd")
    v, t = a3d.tetgrid.kuhn_grid(12)
    a3d.tetgrid.save_tets_npz(str(tmp_path / "24_tets.npz"), v, t)
    geo = mods["dmtet"].DMTetGeometry(24, 7.0, num_layers=3, hidden_size=32, embedder_freq=4, init_sdf="ellipsoid", jitter_grid=0.05,
                                      symmetrize=True, device=dev, tets_dir=str(tmp_path)).to(dev)
    assert geo.verts.shape == (13**3, 3) and geo.indices.dtype == torch.int64 and geo.all_edges.shape[1] == 2
    prior = geo.getMesh(jitter_grid=True)
    assert prior.v_pos.shape[0] == 1 and prior.t_pos_idx.shape[1] > 100 and prior.v_tex.shape[1] == 4 * a3d.tetgrid.uv_grid_size(t.shape[0]) ** 2
    loss = geo.get_sdf_reg_loss()
    assert set(loss) == {"sdf_bce_reg_loss", "sdf_gradient_reg_loss"} and all(bool(torch.isfinite(x)) for x in loss.values())
    sk = mods["skinning"]
    bones, tree, aux = sk.estimate_bones(prior.v_pos[None].detach(), n_body_bones=8, n_legs=0, n_leg_bones=0, body_bones_mode="z_minmax")
    Bq, Fr = 2, 3
    ang = seeded((Bq, Fr, 8, 3), 2, -0.3, 0.3).to(dev).requires_grad_(True)
    verts, aux = sk.skinning(prior.v_pos[None], bones, tree, ang, output_posed_bones=True, temperature=0.05)
    assert verts.shape == (Bq, Fr, prior.v_pos.shape[1], 3) and aux["posed_bones"].shape == (Bq, Fr, 8, 2, 3)
    assert aux["vertices_to_bones"].shape == (8, 1, 1, prior.v_pos.shape[1])
    shape = mods["mesh"].make_mesh(verts.view(Bq * Fr, -1, 3), prior.t_pos_idx, prior.v_tex.expand(Bq * Fr, -1, -1), prior.t_tex_idx, None)
    mvp, w2c, campos = mods["synthetic"].random_cameras(Bq * Fr, seed=4)
    shaded, flow = mods["render"].render_mesh(None, shape, mvp.to(dev), w2c.to(dev), campos.to(dev), None, None, (32, 32), bsdf="diffuse",
                                              render_modes=["shaded", "flow"], num_frames=Fr, prior_mesh=prior)
    assert shaded.shape == (Bq * Fr, 4, 32, 32) and flow.shape == (Bq * Fr, 2, 32, 32)
    (shaded.sum() + flow.sum()).backward()
    assert ang.grad is not None and bool(torch.isfinite(ang.grad).all())
    assert all(p.grad is not None for p in geo.mlp.parameters())


@pytest.mark.parametrize("tag", ["b1f1_t1", "b3f2_t005", "b2f2_inst"])
def test_bone_transforms_kernel_vs_torch_chain(tag, dev, mods):
    """csrc/bones.hip (one launch) against the level-batched torch composition that is itself pinned to the reference goldens."""
    ops = importlib.import_module("3danimals_amd.ops")
    sk = mods["skinning"]
    g = golden(f"skinning_{tag}.npz")
    chain = eval(str(g["chain"]))
    bones = torch.from_numpy(g["bones"]).to(dev)
    ang = torch.from_numpy(g["angles"]).to(dev)
    B, Fr, K = ang.shape[:3]
    a1, a2 = ang.clone().requires_grad_(True), ang.clone().requires_grad_(True)
    ref = sk.bone_transforms_torch(bones, chain, a1)[:, :, :3, :].reshape(B * Fr, K, 12)
    out = ops.bone_transforms(bones.reshape(-1, K, 2, 3), a2.reshape(B * Fr, K, 3), sk._chain_index32(chain, dev))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), atol=3e-6)
    wgt = seeded(ref.shape, 17, -1, 1).to(dev)
    (g1,) = torch.autograd.grad((ref * wgt).sum(), a1)
    (g2,) = torch.autograd.grad((out * wgt).sum(), a2)
    np.testing.assert_allclose(g2.cpu().numpy(), g1.cpu().numpy(), rtol=1e-4, atol=1e-4)


def test_max_size_grid_and_raster_properties(dev, mods, ops):
    """Kuhn R=128 (the reference's grid_res=256 regime: 2.1M grid vertices, 12.6M tets) and a 512x512 render of its mesh."""
    from oracle import raster_ref, render_ref

    pos, tets = kuhn(128)
    sdf = mods["synthetic"].quadruped_sdf(pos, 0.2).to(dev)
    dm = mods["dmtet"].DMTet()
    pos_d, tets_d = pos.to(dev), tets.to(dev)
    v1, f1, _, u1 = dm(pos_d, sdf, tets_d)
    v2, f2, _, u2 = dm(pos_d, sdf, tets_d)
    assert torch.equal(v1, v2) and torch.equal(f1, f2) and torch.equal(u1, u2)  # deterministic
    V, F = v1.shape[0], f1.shape[0]
    assert F > 40000 and int(f1.max()) == V - 1 and len(torch.unique(f1)) == V
    e = torch.cat([f1[:, [0, 1]], f1[:, [1, 2]], f1[:, [2, 0]]], 0)
    und = torch.minimum(e[:, 0], e[:, 1]) * V + torch.maximum(e[:, 0], e[:, 1])
    _, cnt = torch.unique(und, return_counts=True)
    assert bool((cnt == 2).all())  # closed manifold
    # 512x512 render of the 47k-face mesh: bit-exact ids against the oracle, deterministic, watertight
    mvp, _, _ = mods["synthetic"].random_cameras(2, seed=8)
    clip = render_ref.xfm_points(v1.cpu()[None].expand(2, -1, -1), mvp).contiguous()
    ref = raster_ref.rasterize(clip, f1.cpu().int(), (512, 512))
    a = ops.rasterize(clip.to(dev), f1, (512, 512))
    b = ops.rasterize(clip.to(dev), f1, (512, 512))
    assert torch.equal(a, b) and np.array_equal(a.cpu().numpy(), ref.numpy())
    cover = (a[..., 3] > 0).float()[:, None]
    nb = torch.nn.functional.conv2d(cover, torch.tensor([[[[0, 1, 0], [1, 0, 1], [0, 1, 0.0]]]], device=dev), padding=1)
    assert int(((cover == 0) & (nb == 4)).sum()) == 0


@pytest.mark.parametrize("tag,nets,kw", [("a", True, {}), ("b", True, {}), ("c", False, dict(num_frames=2)), ("d", False, dict(two_sided_shading=False)),
                                         ("e", True, dict(num_layers=2))])
def test_render_mesh_matches_reference_render_mesh_golden(tag, nets, kw, dev, mods):
    """G7: product render_mesh against buffers produced by the REFERENCE's render_mesh (oracle operators as nvdiffrast)."""
    import copy

    from test_oracle_golden import _load_nets

    g = golden("render_mesh_e2e.npz")
    tex, dino, lgt = (copy.deepcopy(m).to(dev) for m in _load_nets(None, g))
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    M = mods["mesh"]
    faces = t("faces")
    B = g["v_pos"].shape[0]
    uvs = torch.zeros(1, 4, 2, device=dev)
    uvi = torch.zeros(1, faces.shape[0], 3, dtype=torch.int64, device=dev)
    shape = M.make_mesh(t("v_pos"), faces[None], uvs.expand(B, -1, -1), uvi, None)
    prior = M.make_mesh(t("prior_v_pos")[None], faces[None], uvs, uvi, None)
    modes = str(g[f"{tag}_modes"]).split(",")
    with torch.no_grad():
        kw = dict(kw)
        outs = mods["render"].render_mesh(None, shape, t("mvp"), t("w2c"), t("campos"), tex if nets else None, lgt if nets else None, (32, 32), spp=1,
                                          num_layers=kw.pop("num_layers", 1), msaa=True, background=t("background"), bsdf="diffuse",
                                          feat=t("feat") if nets else None, render_modes=modes, prior_mesh=prior, dino_net=dino if nets else None, **kw)
    for m, o in zip(modes, outs):
        assert tuple(o.shape) == tuple(g[f"{tag}_{m}"].shape), m
        np.testing.assert_allclose(o.cpu().numpy(), g[f"{tag}_{m}"], atol=1e-4, err_msg=m)
    with pytest.raises(KeyError):
        mods["render"].render_mesh(None, shape, t("mvp"), t("w2c"), t("campos"), None, None, (32, 32), bsdf="diffuse", render_modes=["shaded", "bogus"])


@pytest.mark.gpu
def test_config1_geometry_path_matches_oracle():
    """BASELINE config 1 (DMTet R=32 + LBS, no raster): HIP path vs CPU oracle on the same inputs, forward and backward."""
    from oracle import geometry_ref

    pipeline = importlib.import_module("3danimals_amd.pipeline")
    dmtet_mod = importlib.import_module("3danimals_amd.model.geometry.dmtet")
    inp = geometry_ref.make_inputs(res=32, batch=4, seed=0)
    ref = geometry_ref.cpu_step(inp)
    gin = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp.items()}
    out = pipeline.geometry_config1_step(gin, dmtet_mod.TetGridTopology(gin["tets"]))
    assert (out["V"], out["F"]) == (ref["V"], ref["F"]) == (2174, 4344)
    assert abs(float(out["loss"]) - ref["loss"]) < 1e-5
    assert torch.allclose(out["grad_arti"].cpu(), ref["grad_arti"], atol=2e-6, rtol=1e-4)
    # d(vertex)/d(sdf) ~ 1/(s_a - s_b)^2 spans orders of magnitude on a noisy field: compare relative to the gradient's scale
    gs, rs = out["grad_sdf"].cpu(), ref["grad_sdf"]
    # (a handful of sliver faces of the noisy field have ill-conditioned normal gradients: fp32 cancellation in the cross product
    #  differs with operation order, and the order of dm_bwd's float atomics varies from run to run -- measured 6 of 1144 entries
    #  beyond 1e-6, worst 5.4e-5 ... 1.1e-4 against a gradient scale of 8.9e-2)
    d, scale = (gs - rs).abs(), float(rs.abs().max())
    assert float(d.max()) <= 3e-3 * scale
    assert float((d > 1e-5 * scale).float().mean()) < 0.01 * float((rs != 0).float().mean())
    assert torch.equal(gs != 0, rs != 0)


@pytest.mark.parametrize("shape,tile", [((3, 64, 64), 8), ((2, 24, 40), 8), ((2, 30, 20), 8), ((1, 16, 16), 0), ((2, 8, 8), 8)])
def test_covered_pixels_compaction_is_exact(shape, tile, dev, ops):
    """a3d_cover_count/emit against the plain torch expression of the same list (integer work: bit-exact)."""
    b, h, w = shape
    g = torch.Generator().manual_seed(h * w + b)
    for density in (0.0, 0.03, 0.5, 1.0):
        tri_id = torch.where(torch.rand(b, h, w, generator=g) < density, torch.randint(1, 1000, (b, h, w), generator=g), 0).float()
        rast = torch.cat((torch.rand(b, h, w, 3, generator=g), tri_id[..., None]), -1).to(dev)
        pix = ops.covered_pixels(rast, tile=tile)
        cover = rast[..., 3] > 0
        if tile == 8 and h % 8 == 0 and w % 8 == 0:
            flat = torch.arange(b * h * w, device=dev).view(b, h // 8, 8, w // 8, 8).permute(0, 1, 3, 2, 4).reshape(-1)
            want = flat[cover.view(b, h // 8, 8, w // 8, 8).permute(0, 1, 3, 2, 4).reshape(-1)]
        else:
            want = torch.nonzero(cover.reshape(-1)).squeeze(1)
        assert pix.dtype == torch.int64 and torch.equal(pix, want), (shape, tile, density)
        pix2, inv = ops.covered_pixels(rast, tile=tile, return_inverse=True)  # the pixel -> entry map written by the same launch
        want_inv = torch.full((b * h * w,), -1, dtype=torch.int32, device=dev)
        want_inv[want] = torch.arange(want.shape[0], dtype=torch.int32, device=dev)
        assert torch.equal(pix2, want) and inv.dtype == torch.int32 and torch.equal(inv, want_inv), (shape, tile, density)


@pytest.mark.parametrize("hw", [(64, 64), (32, 96), (48, 40), (256, 256)])
def test_covered_pixels_of_a_rasterised_buffer_use_the_resolve_counts(hw, dev, ops):
    """For buffers that come out of ops.rasterize the list's block counts are written by the rasteriser's resolve (tile order, H*W a
    multiple of 256) instead of a3d_cover_count's own pass: the list must equal the plain torch expression either way."""
    H, W = hw
    B = 3
    _, faces, clip, _ = _scene(B, seed=11)
    rast = ops.rasterize(clip.to(dev), faces.to(dev), (H, W))
    counted = ops._cover_counts.peek(rast) is not None
    assert counted == ((H * W) % 256 == 0)
    pix, inv = ops.covered_pixels(rast, return_inverse=True)
    cover = rast[..., 3] > 0
    flat = torch.arange(B * H * W, device=dev).view(B, H // 8, 8, W // 8, 8).permute(0, 1, 3, 2, 4).reshape(-1)
    want = flat[cover.view(B, H // 8, 8, W // 8, 8).permute(0, 1, 3, 2, 4).reshape(-1)]
    assert want.shape[0] > 50 and torch.equal(pix, want)
    want_inv = torch.full((B * H * W,), -1, dtype=torch.int32, device=dev)
    want_inv[want] = torch.arange(want.shape[0], dtype=torch.int32, device=dev)
    assert torch.equal(inv, want_inv)
    # a SECOND list of the same, unmodified buffer: the resolve's block counts are only read by the emit, so they serve again
    assert (ops._cover_counts.peek(rast) is not None) == counted
    pix2, inv2 = ops.covered_pixels(rast, return_inverse=True)
    assert torch.equal(pix2, want) and torch.equal(inv2, want_inv)
    # an in-place edit of the buffer invalidates the cached counts (the version counter is part of the cache key)
    rast[0, : H // 2] = 0.0
    assert ops._cover_counts.peek(rast) is None
    cover = rast[..., 3] > 0
    want = flat[cover.view(B, H // 8, 8, W // 8, 8).permute(0, 1, 3, 2, 4).reshape(-1)]
    assert torch.equal(ops.covered_pixels(rast), want)


def test_empty_triangle_list_does_not_poison_the_kept_scratch_buffers(dev, ops):
    """F == 0 launches nothing that touches the rasteriser's key buffer / the topology hash; the fresh (uninitialised) scratch of such a
    call must not be kept as 'clean' for the next call of the same size (DMTet can legitimately emit an empty mesh)."""
    from oracle import raster_ref

    B, H, W = 2, 40, 56  # a size no other test uses: the first call of this size allocates fresh scratch
    _, faces, clip, _ = _scene(B, seed=21)
    ops._rast_keys.clear()
    ops._topology_scratch.clear()
    poison = torch.full((1 << 22,), 0x3C, dtype=torch.uint8, device=dev)  # make it likely that recycled allocations hold non-0xFF garbage
    del poison
    empty = faces[:0].to(dev).int().contiguous()
    r0 = ops.rasterize(clip.to(dev), empty, (H, W))
    assert float(r0.abs().max()) == 0.0
    r1 = ops.rasterize(clip.to(dev), faces.to(dev), (H, W))
    assert np.array_equal(r1.cpu().numpy(), raster_ref.rasterize(clip, faces.int(), (H, W)).numpy())
    V = int(faces.max()) + 1
    few = faces[:7].to(dev).int().contiguous()  # same hash-size class as F == 0
    ops.mesh_topology(empty, V)
    a, t = ops.mesh_topology(few, V)
    assert np.array_equal(t.opp.cpu().numpy(), raster_ref.edge_opposites(few.cpu().numpy()))


@pytest.mark.parametrize("E,res,hw", [(2, 16, (64, 64)), (3, 16, (96, 80)), (1, 40, (64, 64))])
def test_gbuffer_extra_attribute_equals_modular_interpolate(E, res, hw, dev, ops, mods):
    """The optional per-vertex attribute of the fused G-buffer (the sequence models' 2-D motion) against dr.interpolate + the
    rasteriser's backward on the same pixels: values and the gradients to the attribute and -- through the barycentrics -- to the camera.
    res = 40 gives sub-pixel triangles (the large work-group tables of the backward)."""
    H, W = hw
    B = 3
    verts, faces, _, (mvp, w2c, campos) = _scene(B, res=res, seed=9)
    posed = (verts[None] + 0.03 * seeded((B, *verts.shape), 33, -1, 1)).to(dev)
    tri = faces.to(dev)
    ru = importlib.import_module("3danimals_amd.model.render.renderutils")
    attr0 = seeded((B, verts.shape[0], E), 35, -1, 1).to(dev)

    def run(fused):
        v = posed.clone().requires_grad_(True)
        a = attr0.clone().requires_grad_(True)
        m = mvp.to(dev).clone().requires_grad_(True)
        nrm = ops.vertex_normals(v, tri)
        clip = ru.xfm_points(v, m)
        rast = ops.rasterize(clip, tri, (H, W))
        pix = ops.covered_pixels(rast)
        if fused:
            gb, out = ops.gbuffer(clip, v, nrm, verts[None].to(dev), rast, tri, pix, extra=a)
        else:
            gb = ops.gbuffer(clip, v, nrm, verts[None].to(dev), rast, tri, pix)
            out = ops.interpolate(a, rast, tri).reshape(-1, E).index_select(0, pix)
        wgt, wgb = seeded(out.shape, 43, -1, 1).to(dev), seeded(gb.shape, 45, -1, 1).to(dev)
        ga, gv, gm = torch.autograd.grad((out * wgt).sum() + 0.1 * (gb * wgb).sum(), [a, v, m])
        return out.detach(), gb.detach(), ga, gv, gm

    f, u = run(True), run(False)
    assert f[0].shape[0] > 500
    np.testing.assert_allclose(f[0].cpu().numpy(), u[0].cpu().numpy(), atol=2e-6)
    assert torch.equal(f[1], u[1])
    for x, y, name in zip(f[2:], u[2:], ("attr", "v_pos", "mvp")):
        scale = float(y.abs().max())
        assert scale > 0, name
        np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), rtol=1e-3, atol=2e-4 * scale, err_msg=name)


@pytest.mark.parametrize("C,bg_mode,hw", [(3, "per_image", (64, 64)), (3, "shared", (48, 40)), (16, None, (64, 64)), (2, None, (50, 70)), (1, "per_image", (32, 32))])
def test_composite_antialias_equals_composite_then_antialias(C, bg_mode, hw, dev, ops):
    """The fused compositor (one pass over the image, gradient straight to the point rows) against the two steps it replaces:
    background fill + scatter of the rows (render.py:261-262 with coverage 0/1) and ops.antialias (itself checked against the oracle)."""
    B, (H, W) = 3, hw
    _, faces, clip, _ = _scene(B, seed=7)
    clip, tri = clip.to(dev).requires_grad_(True), faces.to(dev)
    rast = ops.rasterize(clip.detach(), tri, (H, W))
    pix, inv = ops.covered_pixels(rast, return_inverse=True)
    P = pix.shape[0]
    assert P > 100
    g = torch.Generator().manual_seed(C * 100 + H)
    vals = torch.rand(P, C, generator=g).to(dev).requires_grad_(True)
    bg = None
    if bg_mode is not None:
        bg = torch.rand(B if bg_mode == "per_image" else 1, H, W, C + 1, generator=g).to(dev)
    g_out = torch.randn(B, H, W, C + 1, generator=g).to(dev)
    analysis = ops.AAAnalysis(rast, clip.detach(), ops.aa_topology(ops.tri_int32(tri), clip.shape[1]))

    fused = ops.composite_antialias(vals, pix, inv, bg, clip, analysis)
    gv_f, gc_f = torch.autograd.grad(fused, (vals, clip), g_out)

    base = bg.expand(B, -1, -1, -1).clone() if bg is not None else torch.zeros(B, H, W, C + 1, device=dev)
    base = base.view(-1, C + 1).index_copy(0, pix, torch.cat((vals, torch.ones_like(vals[:, :1])), -1)).view(B, H, W, C + 1)
    two_step = ops.antialias(base, rast, clip, tri, analysis=analysis)
    gv_t, gc_t = torch.autograd.grad(two_step, (vals, clip), g_out)

    assert float((fused.detach() - base.detach()).abs().max()) > 1e-3  # the silhouette was really blended
    torch.testing.assert_close(fused, two_step, atol=2e-6, rtol=0)  # same blends; pixels with two crossings add in any order
    torch.testing.assert_close(gv_f, gv_t, atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(gc_f, gc_t, atol=1e-4 * float(gc_t.abs().max()), rtol=1e-4)


@pytest.mark.parametrize("light,two_sided", [(True, True), (True, False), (False, True)])
def test_shade_points_kernel_matches_oracle(light, two_sided, dev, ops):
    """csrc/shade.hip against oracle/render_ref (pinned on the reference's shade goldens): values and all input gradients."""
    from oracle import mesh_ref, render_ref

    P = 5000
    g = torch.Generator().manual_seed(11 + light + 2 * two_sided)
    r = lambda *s: torch.rand(*s, generator=g) * 2 - 1
    gb = torch.cat((r(P, 3), torch.nn.functional.normalize(r(P, 3), dim=-1), r(P, 3), r(P, 3)), -1)
    gb[:7, 6:9] = 0.0  # zero smooth normals: the eps branches of normalize
    par = torch.cat((r(P, 9), r(P, 3) * 3, torch.nn.functional.normalize(r(P, 3), dim=-1), torch.rand(P, 2, generator=g)), -1)[:, : 17 if light else 12]
    tex = torch.rand(P, 9, generator=g)
    w_n, w_s, w_c = r(P, 3), r(P, 1), r(P, 3)

    def run(device, fused):
        gb_, par_, tex_ = (t.clone().to(device).requires_grad_(True) for t in (gb, par, tex))
        kd = tex_[:, :3]
        if fused:
            out = ops.shade_points(gb_, par_, kd if light else None, two_sided)
            nrm, shading, shaded = out if light else (out, None, None)
        else:
            nrm = render_ref.shading_normal(gb_[:, 0:3], par_[:, 9:12], gb_[:, 6:9], gb_[:, 3:6], two_sided)
            shading = shaded = None
            if light:
                cam = mesh_ref.safe_normalize((par_[:, 0:9].reshape(-1, 3, 3) * nrm[:, None, :]).sum(-1))
                shading = par_[:, 15:16] + par_[:, 16:17] * torch.clamp((par_[:, 12:15] * cam).sum(-1, keepdim=True), min=0.0)
                shaded = shading * kd
        loss = (nrm * w_n.to(device)).sum()
        if light:
            loss = loss + (shading * w_s.to(device)).sum() + (shaded * w_c.to(device)).sum()
        loss.backward()
        outs = [nrm] + ([shading, shaded] if light else [])
        return [t.detach().cpu() for t in outs], [t.grad.cpu() if t.grad is not None else None for t in (gb_, par_, tex_)]

    (vals, grads), (vals_ref, grads_ref) = run(dev, True), run("cpu", False)
    for a, b_ in zip(vals, vals_ref):
        assert torch.allclose(a, b_, atol=2e-6, rtol=1e-5), float((a - b_).abs().max())
    for a, b_, name in zip(grads, grads_ref, ("gb", "par", "tex")):
        if b_ is None:
            assert a is None or float(a.abs().max()) == 0.0
            continue
        # gradients through normalize() of near-zero vectors are huge and ill-conditioned: compare relative to each row's scale
        scale = b_.abs().amax(dim=1, keepdim=True).clamp(min=1.0)
        assert float(((a - b_).abs() / scale).max()) < 2e-4, (name, float(((a - b_).abs() / scale).max()))


@pytest.mark.parametrize("light,P,B", [(True, 5000, 7), (False, 3000, 4), (True, 700, 16), (True, 257, 1), (True, 20, 16), (True, 900, 64),
                                       (False, 3, 40)])  # the last three: B*ncol far above the forward launch's thread count
def test_shade_points_per_image_rows_equal_per_point_rows(light, P, B, dev, ops):
    """The indexed form (camera / light rows per image + point -> image index, gradient reduced per image inside the kernel) against the
    per-point form fed with the gathered rows (itself checked against the oracle above): same values, row gradient = per-image sums.
    Work-groups inside one image, straddling two, and images without points are all present."""
    g = torch.Generator().manual_seed(P + B)
    r = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).to(dev)
    ncol = 17 if light else 12
    gb = torch.cat((r(P, 3), torch.nn.functional.normalize(r(P, 3), dim=-1), r(P, 3), r(P, 3)), -1).requires_grad_(True)
    rows = torch.cat((r(B, 9), r(B, 3) * 3, torch.nn.functional.normalize(r(B, 3), dim=-1), torch.rand(B, 2, generator=g).to(dev)), -1)[:, :ncol]
    img = torch.sort(torch.randint(0, B, (P,), generator=g))[0].to(dev)
    if B > 2:
        img[img == 1] = 2  # an image without points
    tex = torch.rand(P, 9, generator=g).to(dev).requires_grad_(True)
    kd = tex[:, :3] if light else None
    w = [r(P, 3), r(P, 1), r(P, 3)]

    def run(indexed):
        rows_ = rows.clone().requires_grad_(True)
        out = ops.shade_points(gb, rows_, kd, True, img=img) if indexed else ops.shade_points(gb, rows_.index_select(0, img), kd, True)
        outs = out if light else (out,)
        loss = sum((o * wi).sum() for o, wi in zip(outs, w))
        grads = torch.autograd.grad(loss, (gb, rows_) + ((tex,) if light else ()))
        return [o.detach() for o in outs], grads

    (v_i, g_i), (v_p, g_p) = run(True), run(False)
    for a, b_ in zip(v_i, v_p):
        assert torch.equal(a, b_)
    assert torch.equal(g_i[0], g_p[0]) and (not light or torch.equal(g_i[2], g_p[2]))
    scale = float(g_p[1].abs().max())
    torch.testing.assert_close(g_i[1], g_p[1], atol=2e-5 * scale, rtol=1e-4)  # summation order differs
    if B > 2:
        assert float(g_i[1][1].abs().max()) == 0.0


@pytest.mark.parametrize("Ca,Cb,hw", [(3, 16, (64, 64)), (16, 3, (48, 40)), (2, 1, (50, 70))])
def test_composite_antialias_two_buffers_in_one_call(Ca, Cb, hw, dev, ops):
    """Two buffers over the same pixel list through the same launches against two single-buffer calls: values, both row gradients
    and the summed vertex gradient; one output left undifferentiated must behave as a zero gradient."""
    B, (H, W) = 3, hw
    _, faces, clip, _ = _scene(B, seed=7)
    clip, tri = clip.to(dev).requires_grad_(True), faces.to(dev)
    rast = ops.rasterize(clip.detach(), tri, (H, W))
    pix, inv = ops.covered_pixels(rast, return_inverse=True)
    P = pix.shape[0]
    g = torch.Generator().manual_seed(Ca * 10 + Cb)
    va = torch.rand(P, Ca, generator=g).to(dev).requires_grad_(True)
    vb = torch.rand(P, Cb, generator=g).to(dev).requires_grad_(True)
    bga = torch.rand(B, H, W, Ca + 1, generator=g).to(dev)
    ga, gb = torch.randn(B, H, W, Ca + 1, generator=g).to(dev), torch.randn(B, H, W, Cb + 1, generator=g).to(dev)
    analysis = ops.AAAnalysis(rast, clip.detach(), ops.aa_topology(ops.tri_int32(tri), clip.shape[1]))

    oa, ob = ops.composite_antialias(va, pix, inv, bga, clip, analysis, vals2=vb, background2=None)
    sa = ops.composite_antialias(va, pix, inv, bga, clip, analysis)
    sb = ops.composite_antialias(vb, pix, inv, None, clip, analysis)
    torch.testing.assert_close(oa, sa, atol=2e-6, rtol=0)
    torch.testing.assert_close(ob, sb, atol=2e-6, rtol=0)
    g2 = torch.autograd.grad((oa * ga).sum() + (ob * gb).sum(), (va, vb, clip))
    g1 = torch.autograd.grad((sa * ga).sum() + (sb * gb).sum(), (va, vb, clip))
    torch.testing.assert_close(g2[0], g1[0], atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(g2[1], g1[1], atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(g2[2], g1[2], atol=1e-4 * float(g1[2].abs().max()), rtol=1e-4)
    # only the second output used
    oa, ob = ops.composite_antialias(va, pix, inv, bga, clip, analysis, vals2=vb)
    h2 = torch.autograd.grad((ob * gb).sum(), (va, vb, clip), allow_unused=True)
    h1 = torch.autograd.grad((ops.composite_antialias(vb, pix, inv, None, clip, analysis) * gb).sum(), (vb, clip))
    assert h2[0] is None or float(h2[0].abs().max()) == 0.0
    torch.testing.assert_close(h2[1], h1[0], atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(h2[2], h1[1], atol=1e-4 * float(h1[1].abs().max()), rtol=1e-4)


def test_rows_add_relu_and_indexed_feature_field(dev, ops):
    """a3d_rows_add_relu_fwd/bwd against torch, and CoordMLP's per-image feature path (HIP add+ReLU, split-K weight gradient)
    against the reference formulation (feature concatenated per point) on the GPU."""
    hostnets = importlib.import_module("3danimals_amd.hostnets")
    g = torch.Generator().manual_seed(5)
    P, C, B = 70000, 64, 5
    img = torch.randint(0, B, (P,), generator=g).sort().values.to(dev)
    img[-100:] = 0  # padded tail rows map to image 0, unsorted (render.POINT_BUCKET padding)
    y0 = torch.randn(P, C, generator=g).to(dev).requires_grad_(True)
    rows = torch.randn(B, C, generator=g).to(dev).requires_grad_(True)
    w = torch.randn(P, C, generator=g).to(dev)
    out = ops.rows_add_relu_(y0 * 1.0, rows, img)
    ga = torch.autograd.grad((out * w).sum(), [y0, rows])
    ref = torch.relu(y0 + rows[img])
    gb = torch.autograd.grad((ref * w).sum(), [y0, rows])
    assert torch.equal(out, ref) and torch.equal(ga[0], gb[0])
    assert torch.allclose(ga[1], gb[1], rtol=1e-4, atol=1e-3)

    torch.manual_seed(0)
    net = hostnets.CoordMLP(3, 9, 5, nf=256, n_harmonic_functions=10, extra_feat_dim=256, min_max=torch.tensor([[0.0, 1.0]] * 9),
                            activation="sigmoid", symmetrize=True).to(dev)
    P = hostnets.SPLITK_MIN_ROWS + 8192
    x = (torch.rand(P, 3, generator=g) * 2 - 1).to(dev)
    feat = torch.randn(B, 256, generator=g).to(dev).requires_grad_(True)
    idx = torch.randint(0, B, (P,), generator=g).sort().values.to(dev)
    a = net.sample(x, feat=feat, feat_index=idx)
    ga = torch.autograd.grad(a.square().sum(), [feat] + list(net.parameters()))
    b_ = net.sample(x, feat=feat[idx])
    gb = torch.autograd.grad(b_.square().sum(), [feat] + list(net.parameters()))
    assert torch.allclose(a, b_, atol=2e-6)
    for u, v in zip(ga, gb):
        assert float((u - v).abs().max()) <= 1e-4 * float(v.abs().max()) + 1e-6


def test_ddp_wrapper_over_rccl_gives_the_same_gradients(dev):
    """The scene under DistributedDataParallel (backend nccl = RCCL, world size 1 on this box): custom autograd Functions, in-place
    HIP ops and the split-K weight gradients behind DDP's hooks give the gradients of the bare module; no parameter is unused."""
    import socket
    import torch.distributed as dist

    pipeline = importlib.import_module("3danimals_amd.pipeline")
    kw = dict(grid_res=16, batch=2, resolution=(64, 64), device=dev, seed=3, net_width=32, net_layers=3, feat_dim=16, embedder_freq=4)
    bare = pipeline.SyntheticScene(**kw)
    torch.manual_seed(7)  # the step jitters the grid with the global generator
    bare.step(optimizer_step=False)
    want = {n: p.grad.clone() for n, p in bare.named_parameters()}
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device(dev))
    try:
        scene = pipeline.SyntheticScene(**kw)
        module = torch.nn.parallel.DistributedDataParallel(scene, device_ids=[torch.device(dev).index or 0], broadcast_buffers=False,
                                                           gradient_as_bucket_view=True)
        for _ in range(2):  # twice: the second backward runs with DDP's rebuilt buckets
            torch.manual_seed(7)
            scene.step(module=module, optimizer_step=False)
        for n, p in scene.named_parameters():
            assert p.grad is not None, n
            assert float((p.grad - want[n]).abs().max()) <= 1e-5 * float(want[n].abs().max()) + 1e-7, n
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("with_dino,hw", [(True, (64, 64)), (False, (40, 24)), (True, (17, 33))])
def test_fused_reconstruction_losses_match_torch_formulation(with_dino, hw, dev, ops):
    """csrc/losses.hip against the five torch expressions of compute_reconstruction_losses (AnimalModel.py:260-307), values and
    both image gradients; the eroded 'both' mask is integer logic and must agree exactly."""
    B, (H, W), D = 3, hw, 16
    g = torch.Generator().manual_seed(H * W + with_dino)
    shaded_nhwc = torch.rand(B, H, W, 4, generator=g)
    shaded_nhwc[..., 3] = (torch.rand(B, H, W, generator=g) > 0.4).float() * torch.rand(B, H, W, generator=g).clamp(min=0.2)
    shaded_nhwc[:, 5:12, 3:15, 3] = 1.0
    dino_nhwc = torch.rand(B, H, W, D, generator=g)
    image_gt, dino_gt = torch.rand(B, 3, H, W, generator=g), torch.rand(B, D, H, W, generator=g)
    mask_gt = (torch.rand(B, H, W, generator=g) > 0.3).float()
    mask_gt[:, 4:13, 2:16] = 1.0
    mask_dt, valid = torch.rand(B, 2, H, W, generator=g), (torch.rand(B, H, W, generator=g) > 0.1).float()
    w = torch.rand(B, 5, generator=g) + 0.5

    def torch_losses(shaded, dino):
        image_pred, mask_pred = shaded[:, :3], shaded[:, 3]
        out = [((mask_pred * valid_d - mask_gt_d) ** 2).flatten(1).mean(1), ((1 - mask_pred) * mask_dt_d[:, 0]).flatten(1).mean(1)]
        both = ((mask_pred * valid_d > 0.0).float() * mask_gt_d).detach()
        both = (torch.nn.functional.avg_pool2d(both.unsqueeze(1), 3, stride=1, padding=1).squeeze(1) > 0.99).float()
        out.append(((image_pred - image_gt_d).abs() * both.unsqueeze(1)).flatten(1).mean(1))
        out.append((((dino - dino_gt_d) ** 2) * both.unsqueeze(1)).flatten(1).mean(1) if dino is not None else torch.zeros_like(out[0]))
        out.append((mask_pred * mask_dt_d[:, 1]).flatten(1).mean(1))  # mask_dt_loss (AnimalModel.py:268)
        return torch.stack(out, 1), both

    image_gt_d, dino_gt_d, mask_gt_d, mask_dt_d, valid_d = (t.to(dev) for t in (image_gt, dino_gt, mask_gt, mask_dt, valid))
    res = []
    for fused in (True, False):
        s = shaded_nhwc.clone().to(dev).requires_grad_(True)
        d = dino_nhwc.clone().to(dev).requires_grad_(True) if with_dino else None
        s_nchw, d_nchw = s.permute(0, 3, 1, 2), (d.permute(0, 3, 1, 2) if with_dino else None)
        if fused:
            loss = ops.reconstruction_losses(s_nchw, d_nchw, image_gt_d, dino_gt_d if with_dino else None, mask_gt_d, mask_dt_d, valid_d)
        else:
            loss, both = torch_losses(s_nchw, d_nchw)
            assert 0 < float(both.mean()) < 1  # the eroded mask is neither empty nor full
        (loss * w.to(dev)).sum().backward()
        res.append((loss.detach(), s.grad, d.grad if with_dino else None))
    (la, gsa, gda), (lb, gsb, gdb) = res
    assert torch.allclose(la, lb, rtol=2e-5, atol=1e-7), (la, lb)
    assert torch.allclose(gsa, gsb, rtol=1e-5, atol=1e-9)
    if with_dino:
        assert torch.allclose(gda, gdb, rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("tag", ["b3f1", "b2f4_flow"])
def test_fused_losses_match_reference_compute_reconstruction_losses_golden(tag, dev, ops):
    """G9: csrc/losses.hip (all six terms: mask, mask_dt, mask_inv_dt, rgb, flow, dino; F = 1 and F = 4) against the outputs of the
    REFERENCE's AnimalModel.compute_reconstruction_losses (imported in the build container, tests/golden/make_golden.py)."""
    g = golden("recon_losses.npz")
    t = lambda k: torch.from_numpy(g[f"{tag}_in_{k}"]).to(dev)
    image_pred, mask_pred = t("image_pred"), t("mask_pred")
    B, F, _, H, W = image_pred.shape
    N = B * F
    # the renderer's layout: NHWC buffers, handed over as NCHW views
    shaded = torch.cat([image_pred.view(N, 3, H, W), mask_pred.view(N, 1, H, W)], 1).permute(0, 2, 3, 1).contiguous().requires_grad_(True)
    dino = t("dino_pred").view(N, 16, H, W).permute(0, 2, 3, 1).contiguous().requires_grad_(True)
    loss, both = ops.reconstruction_losses(shaded.permute(0, 3, 1, 2), dino.permute(0, 3, 1, 2), t("image_gt").view(N, 3, H, W),
                                           t("dino_gt").view(N, 16, H, W), t("mask_gt").view(N, H, W), t("mask_dt").view(N, 2, H, W),
                                           t("mask_valid").view(N, H, W), return_mask=True)
    cols = dict(mask_loss=0, mask_inv_dt_loss=1, rgb_loss=2, dino_feat_im_loss=3, mask_dt_loss=4)
    for name, c in cols.items():
        np.testing.assert_allclose(loss[:, c].detach().cpu().view(B, F).numpy(), g[f"{tag}_out_{name}"], rtol=2e-5, atol=1e-7, err_msg=name)
    if f"{tag}_in_flow_pred" in g.files:
        flow3 = torch.cat([t("flow_pred"), torch.zeros(B, 1, 2, H, W, device=dev)], 1).view(N, 2, H, W)  # last frame: no pair
        buf = torch.cat([flow3.permute(0, 2, 3, 1), torch.ones(N, H, W, 1, device=dev)], -1).contiguous().requires_grad_(True)  # [N,H,W,3] as rendered
        fl = ops.flow_loss(buf[..., :2].permute(0, 3, 1, 2), t("flow_gt"), both, B, F)
        ref = g[f"{tag}_out_flow_loss"]
        assert (ref == 0).any() and (ref > 0).any()  # the fixture exercises the large-flow rule both ways
        np.testing.assert_allclose(fl.detach().cpu().numpy(), ref, rtol=2e-5, atol=1e-8)
        fl.sum().backward()
        assert float(buf.grad[..., 2].abs().max()) == 0 and float(buf.grad.view(B, F, H, W, 3)[:, -1].abs().max()) == 0
        assert float(buf.grad.abs().max()) > 0


def test_animal_model_render_golden(dev, mods):
    """G8: what the sole training-time caller, AnimalModel.render (AnimalModel.py:217-258), returns in the reference
    -- background image by mode, render_mesh(spp=1, num_layers=1, msaa=True) -- against the product's render_mesh called the same way."""
    import copy

    from test_oracle_golden import _load_nets

    g7, g8 = golden("render_mesh_e2e.npz"), golden("animal_model_render.npz")
    assert bool(g8["context_created"])
    tex, dino, lgt = (copy.deepcopy(m).to(dev) for m in _load_nets(None, g7))
    t = lambda k: torch.from_numpy(g7[k]).to(dev)
    M, faces = mods["mesh"], t("faces")
    B = g7["v_pos"].shape[0]
    uvs = torch.zeros(1, 4, 2, device=dev)
    uvi = torch.zeros(1, faces.shape[0], 3, dtype=torch.int64, device=dev)
    shape = M.make_mesh(t("v_pos"), faces[None], uvs.expand(B, -1, -1), uvi, None)
    prior = M.make_mesh(t("prior_v_pos")[None], faces[None], uvs, uvi, None)
    for tag, bg in (("none", torch.zeros), ("white", torch.ones)):
        bg_image = bg((B, 32, 32, 3), device=dev)  # AnimalModel.py:226-229
        with torch.no_grad():
            outs = mods["render"].render_mesh(None, shape, mtx_in=t("mvp"), w2c=t("w2c"), view_pos=t("campos"), material=tex, lgt=lgt,
                                              resolution=(32, 32), spp=1, num_layers=1, msaa=True, background=bg_image, bsdf="diffuse",
                                              feat=t("feat"), render_modes=["shaded", "dino_pred"], prior_mesh=prior, two_sided_shading=True,
                                              dino_net=dino, num_frames=None, class_vector=None)
        np.testing.assert_allclose(outs[0].cpu().numpy(), g8[f"{tag}_shaded"], atol=1e-4, err_msg=tag)
        np.testing.assert_allclose(outs[1].cpu().numpy(), g8[f"{tag}_dino_pred"], atol=1e-4, err_msg=tag)


@pytest.mark.parametrize("bone_y_threshold", [None, 0.4])
def test_estimate_bones_on_device_matches_golden_without_host_sync(bone_y_threshold, dev, mods):
    """with the kinematic chain cached (every iteration after the first; Fauna recomputes the bones per iteration) the
    bone estimate runs on the device without a single host synchronisation, and equals the CPU result."""
    sk = mods["skinning"]
    verts, _, _, _ = _scene(1, seed=5)
    seq = (verts[None, None] + 0.02 * seeded((3, 2, *verts.shape), 9, -1, 1))
    kw = dict(n_body_bones=8, n_legs=4, n_leg_bones=3, body_bones_mode="z_minmax_y+", bone_y_threshold=bone_y_threshold)
    bones_cpu, chain, aux = sk.estimate_bones(seq, compute_kinematic_chain=True, **kw)
    seq_d = seq.to(dev)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        bones_dev = sk.estimate_bones(seq_d, compute_kinematic_chain=False, aux=aux, **kw)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert torch.allclose(bones_dev.cpu(), bones_cpu, atol=1e-6)


@pytest.mark.parametrize("mode,thr,shape", [("z_minmax_y+", None, (1, 1)), ("z_minmax_y+", 0.4, (1, 1)), ("z_minmax", None, (3, 2)), ("z_minmax_y+", 0.4, (2, 3)),
                                            ("z_minmax_y+", None, (1, 1, 0))])
def test_estimate_bones_kernel_equals_the_torch_restatement(mode, thr, shape, dev, mods, monkeypatch):
    """a3d_estimate_bones (round 6: the whole heuristic skeleton as ONE launch of one work-group -- centroid, spine ends, quantiles by radix
    select, the four feet, joints, bones) against the torch restatement it replaces (~245 launches; itself pinned by the reference's
    goldens): bones within 1e-6, the same kinematic chain and attachment joints, chain rebuilt and chain cached, several instances at once
    (the quantiles run over ALL values of the call, as the reference's tensor.quantile() does), with and without legs."""
    sk = mods["skinning"]
    _lib = importlib.import_module("3danimals_amd._lib")
    n_leg = 0 if len(shape) == 3 else 3
    B, Fr = shape[0], shape[1]
    verts, _, _, _ = _scene(1, seed=5)
    seq = (verts[None, None] + 0.02 * seeded((B, Fr, *verts.shape), 9, -1, 1)).to(dev)
    kw = dict(n_body_bones=8, n_legs=4, n_leg_bones=n_leg, body_bones_mode=mode, bone_y_threshold=thr)

    def run(device_kernel):
        monkeypatch.setattr(sk, "DEVICE_ESTIMATE_BONES", device_kernel)
        with _lib.KernelTimer() as timer:
            bones, chain, aux = sk.estimate_bones(seq.clone(), compute_kinematic_chain=True, **kw)
            again = sk.estimate_bones(seq.clone(), compute_kinematic_chain=False, aux=aux, **kw)
        return bones, chain, aux, again, timer.summary()

    b_k, chain_k, aux_k, again_k, calls_k = run(True)
    b_t, chain_t, aux_t, again_t, calls_t = run(False)
    assert calls_k.get("a3d_estimate_bones", (0,))[0] == 2 and "a3d_estimate_bones" not in calls_t
    assert b_k.shape == b_t.shape == (B, Fr, 8 + 4 * n_leg, 2, 3)
    assert float((b_k - b_t).abs().max()) <= 1e-6 and float((again_k - again_t).abs().max()) <= 1e-6 and float((again_k - b_k).abs().max()) == 0.0
    assert repr(chain_k) == repr(chain_t) and aux_k["bones_to_joints"] == aux_t["bones_to_joints"]
    if n_leg:
        assert [l["body_bone_idx"] for l in aux_k["legs"]] == [l["body_bone_idx"] for l in aux_t["legs"]]
    _lib.poll_deferred()  # (every quadrant holds a vertex: nothing pending raises)


def test_estimate_bones_kernel_reports_an_empty_quadrant_through_the_deferred_check(dev, mods):
    """A shape without a vertex in one leg quadrant: the kernel's ``ok`` word travels as a deferred check and raises the reference's message at
    the next poll / read-back (the reference drops into pdb there, skinning.py:183)."""
    sk = mods["skinning"]
    _lib = importlib.import_module("3danimals_amd._lib")
    verts, _, _, _ = _scene(1, seed=5)
    v = verts.clone()
    v[:, 0] = v[:, 0].abs() * 0.01 + 0.5  # everything on the +x side: the two -x quadrants are empty
    _lib.poll_deferred()
    bones = sk.estimate_bones(v[None, None].to(dev), n_body_bones=8, n_legs=4, n_leg_bones=3, body_bones_mode="z_minmax_y+",
                              compute_kinematic_chain=True, legs_to_body_joint_indices=[2, 7, 7, 2])[0]
    assert torch.isfinite(bones).all()
    with pytest.raises(_lib.A3DError, match="no vertex in a leg quadrant"):
        _lib.poll_deferred()


@pytest.mark.parametrize("tag,kw", [("default", dict(attach_legs_to_body=True)), ("fauna", dict(attach_legs_to_body=True, bone_y_threshold=0.4)),
                                    ("fixed", dict(attach_legs_to_body=True, legs_to_body_joint_indices=[2, 7, 7, 2]))])
def test_estimate_bones_on_device_against_reference_golden(tag, kw, dev, mods):
    """estimate_bones run ON THE GPU with the kinematic chain rebuilt (what Fauna does every iteration) against the reference's own
    outputs (bones_quadruped_r16.npz): bones, chain, attachment joints.  Rebuilding the chain costs exactly one read-back (the two
    attachment joints in one transfer; none when they are prescribed) -- counted through torch's sync-debug warnings."""
    import warnings

    sk = mods["skinning"]
    g = golden("bones_quadruped_r16.npz")
    shape = torch.from_numpy(g["verts"])[None, None].to(dev)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("warn")
    try:
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            bones, chain, aux = sk.estimate_bones(shape.clone(), n_body_bones=8, n_legs=4, n_leg_bones=3, body_bones_mode="z_minmax_y+",
                                                  compute_kinematic_chain=True, **{k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()})
    finally:
        torch.cuda.set_sync_debug_mode("default")
    syncs = [w for w in caught if "synchroniz" in str(w.message).lower() and "prototype feature" not in str(w.message)]
    assert len(syncs) <= (0 if "legs_to_body_joint_indices" in kw else 1), [str(w.message)[:120] for w in syncs]
    np.testing.assert_allclose(bones.cpu().numpy(), g[f"{tag}_bones"], atol=1e-6)
    assert repr(chain) == str(g[f"{tag}_chain"])
    assert [l["body_bone_idx"] for l in aux["legs"]] == list(g[f"{tag}_leg_body_idx"])


def test_bench_two_ranks_over_rccl_when_two_gpus_are_present():
    """The first N > 1 RCCL execution should not be the driver's scaling run: bench.py --gpus 2 under torch.distributed.run, exactly as the
    driver launches it.  Skips on the single-GPU development boxes."""
    import json
    import subprocess
    import sys

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--grid-res", "32", "--resolution", "128", "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak" and line["config"]["global_batch"] == 32


@pytest.mark.parametrize("launcher", ["plain", "torchrun"])
def test_bench_two_ranks_on_one_gpu_through_both_command_shapes(launcher):
    """`python bench.py --gpus 2 ...` (bench.py spawns its own ranks, the way `accelerate launch --multi_gpu run.py` does for the reference,
    README.md:49-52) and the same under torch.distributed.run: two ranks, both on cuda:0, DDP over gloo (two RCCL ranks cannot share a
    device) -- the whole N > 1 path of bench.py (scene -> graph capture -> process group -> DDP -> barrier-bracketed timing -> max over
    ranks -> gathered per-rank covered pixels -> ONE json line, last on stdout) on the single-GPU box."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tail = [os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "2", "--warmup", "1", "--grid-res", "16",
            "--resolution", "64", "--batch", "2", "--no-cpu-baseline", "--no-tuned-gemms"]
    cmd = [sys.executable] + tail if launcher == "plain" else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                                              "--master-addr", "127.0.0.1", "--master-port", "29541"] + tail
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = p.stdout.strip().splitlines()
    line = json.loads(lines[-1])
    if launcher == "plain":
        assert len(lines) == 1 and "self-launch" in line["launcher"]  # everything else the ranks print goes to stderr
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["backend"] == "gloo" and line["value"] > 0 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 4 and len(line["covered_pixels_per_rank"]) == 2 and min(line["covered_pixels_per_rank"]) > 0
    assert line["roofline"] is not None and line["roofline"]["in_scope"]["us_per_step"] > 0
    # (round 4) the same launch also times every rank on its own poses and the Fauna per-rank step, each through its own DDP wrapper
    legs = line["extra_legs"]
    assert set(legs) == {"per_rank_poses", "fauna"} and all(v["value"] > 0 and len(v["ms_per_step_per_rank"]) == 2 for v in legs.values())
    assert len(set(legs["per_rank_poses"]["covered_pixels_per_rank"])) == 2 and min(legs["fauna"]["covered_pixels_per_rank"]) > 0
    assert legs["fauna"]["workload"] == "fauna" and len(line["ms_per_step_per_rank"]) == 2 and line["dmtet_pass"] in ("plain", "culled", "ordered")


def test_bench_eight_ranks_on_one_gpu_report_their_host_side_figures():
    """Multi-GPU readiness without the node: bench.py --gpus 8, eight processes on cuda:0, DDP over gloo -- the launch the
    driver uses at N = 8, with the only scaling risk SURVEY 8e names (eight Python processes issuing their launches and stalling at their
    read-backs side by side) exercised for real.  The line carries per-rank step times and, per rank, the blocking host synchronisations
    of a step and the GPU time of its own kernels against its wall time."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--backend", "gloo", "--share-gpu", "--steps", "3", "--warmup", "1", "--grid-res", "32",
           "--resolution", "128", "--batch", "2", "--no-cpu-baseline", "--no-tuned-gemms", "--no-extra-legs", "--no-fingerprint"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and line["rccl_ranks"] == 8 and line["backend"] == "gloo" and line["value"] > 0 and line["config"]["global_batch"] == 16
    assert len(line["ms_per_step_per_rank"]) == 8 and len(line["covered_pixels_per_rank"]) == 8 and min(line["covered_pixels_per_rank"]) > 0
    diag = line["rank_diagnostics"]["per_rank"]
    assert sorted(d["rank"] for d in diag) == list(range(8))
    # the magicpony step reads back twice (DMTet counts, covered-pixel sums) on every rank, wherever it runs
    assert all(2 <= d["host_syncs_per_step"] <= 4 and d["own_kernels_ms_per_step"] > 0 and 0.0 <= d["host_bound_frac"] <= 1.0 for d in diag), diag
    assert "no N > 1 RCCL scaling curve" in line["scaling_curve_note"]


def test_render_uv_bake_and_material_export(tmp_path, dev, mods):
    """f4 output side: render_uv (render.py:342-360) -- rasterise the uv atlas, interpolate the positions, sample the texture field --
    against the oracle operators on the same mesh, then write_obj + save_mtl with the MLP material (the test_* configs' export)."""
    from PIL import Image

    from oracle import raster_ref

    pipeline = importlib.import_module("3danimals_amd.pipeline")
    scene = pipeline.SyntheticScene(grid_res=8, batch=2, resolution=(32, 32), device=dev, seed=2, net_width=32, net_layers=3, feat_dim=16,
                                    embedder_freq=4)
    with torch.no_grad():
        scene.step(backward=False)
        shape = scene.last["shape"]
        res = (64, 64)
        mask, kd, ks, nrm = mods["render"].render_uv(None, shape, res, scene.netTexture, feat=scene.feat)
    assert mask.shape == (2, 64, 64, 1) and kd.shape == (2, 64, 64, 3) and ks.shape == kd.shape and nrm.shape == kd.shape
    # oracle: the same three steps with the CPU operators and the same field
    import copy

    tex = copy.deepcopy(scene.netTexture).cpu()
    uv = shape.v_tex.cpu() * 2 - 1
    uv4 = torch.cat((uv, torch.zeros_like(uv[..., :1]), torch.ones_like(uv[..., :1])), -1)
    rast = raster_ref.rasterize(uv4.contiguous(), shape.t_tex_idx[0].cpu().int(), res)
    gb_pos = raster_ref.interpolate(shape.v_pos.cpu(), rast, shape.t_pos_idx[0].cpu().int())
    with torch.no_grad():
        all_tex = tex.sample(gb_pos, feat=scene.feat.detach().cpu())
    assert torch.equal(mask.cpu()[..., 0] > 0, rast[..., 3] > 0) and 0.0 < float(mask.mean()) < 1.0
    cov = (rast[..., 3] > 0)[..., None]
    np.testing.assert_allclose((kd.cpu() * cov).numpy(), (all_tex[..., :3] * cov).numpy(), atol=1e-4)
    np.testing.assert_allclose((ks.cpu() * cov).numpy(), (all_tex[..., 3:6] * cov).numpy(), atol=1e-4)
    # export: OBJ + MTL + the three baked maps
    shape.material = {"bsdf": "diffuse", "kd_ks_normal": scene.netTexture}
    mods["render"]  # (render module imported)
    obj = importlib.import_module("3danimals_amd.model.render.obj")
    obj.write_obj(str(tmp_path), "horse_mesh", shape, 1, save_material=True, feat=scene.feat[1:2], resolution=[32, 32])
    mtl = open(tmp_path / "horse_mesh.mtl").read()
    assert "newmtl defaultMat" in mtl and "bsdf   diffuse" in mtl and "map_Kd horse_texture_kd.png" in mtl and "bump horse_texture_n.png" in mtl
    img = np.asarray(Image.open(tmp_path / "horse_texture_kd.png"))
    assert img.shape == (32, 32, 3) and img.dtype == np.uint8 and img.max() > 0


def test_graphed_sdf_gradient_equals_eager(dev):
    """DMTetGeometry._graphed_sdf_gradient (forward + double backward replayed from HIP graphs) against the eager autograd path on
    the same points: same kernels in the same order, so the values and the parameter gradients are bit-identical."""
    pipeline = importlib.import_module("3danimals_amd.pipeline")
    scene = pipeline.SyntheticScene(grid_res=16, batch=1, resolution=(32, 32), device=dev, seed=1, net_width=32, net_layers=3, feat_dim=16,
                                    embedder_freq=4)
    geo = scene.netShape
    pts = (seeded((2000, 3), 77, -2, 2)).to(dev)
    params = list(geo.mlp.parameters())

    def eager(p):
        p = p.clone().requires_grad_(True)
        y = geo.get_sdf(pts=p)
        return torch.autograd.grad([y], p, grad_outputs=torch.ones_like(y), create_graph=True)[0]

    loss = lambda g: ((g.norm(dim=-1) - 1) ** 2).mean()
    ge = eager(pts)
    grads_e = torch.autograd.grad(loss(ge), params, allow_unused=True)
    for _ in range(2):  # capture, then a pure replay
        gg = geo._graphed_sdf_gradient(pts)
        grads_g = torch.autograd.grad(loss(gg), params, allow_unused=True)
        assert torch.equal(ge, gg)
        for a, b_ in zip(grads_e, grads_g):
            assert (a is None and (b_ is None or float(b_.abs().max()) == 0.0)) or torch.equal(a, b_)
    with torch.no_grad():  # parameters change between replays (the optimizer step): the graph must read the current ones
        for p in params:
            p.mul_(1.01)
    assert torch.equal(eager(pts), geo._graphed_sdf_gradient(pts))


@pytest.mark.parametrize("n,symmetrize,ones", [(10, True, True), (8, False, True), (4, True, False)])
def test_harmonic_embed_kernel_matches_reference_formulation(n, symmetrize, ones, dev, ops):
    """csrc/embed.hip against the torch expressions of HarmonicEmbedding.py:33-44 + MLPs.py:73-83: forward bit-identical (same
    libm, no contraction), input gradient to fp32 rounding."""
    hostnets = importlib.import_module("3danimals_amd.hostnets")
    emb = hostnets.HarmonicEmbedding(n, 2 * np.pi / 7 * 0.9)
    P = 5000
    x = (seeded((P, 3), 3 + n, -3.5, 3.5)).to(dev)
    x[:5, 0] = 0.0  # |x| is not differentiable at 0: torch.abs gives 0 there
    w = seeded((P, 3 + 6 * n + int(ones)), 17, -1, 1).to(dev)
    xa = x.clone().requires_grad_(True)
    got = ops.harmonic_embed(xa, emb._frequencies(x.device), symmetrize=symmetrize, ones=ones)
    (ga,) = torch.autograd.grad((got * w).sum(), xa)
    xb = x.clone().requires_grad_(True)
    xs = torch.cat([xb[..., :1].abs(), xb[..., 1:]], -1) if symmetrize else xb
    want = torch.cat([xs, emb(xs)] + ([torch.ones(P, 1, device=dev)] if ones else []), -1)
    (gb,) = torch.autograd.grad((want * w).sum(), xb)
    assert torch.equal(got, want)
    assert float((ga - gb).abs().max()) <= 2e-6 * float(gb.abs().max())  # gradients reach 2^(n-1) * scalar * sum|w|: compare to scale


def test_two_process_ddp_on_one_gpu_replays_captured_graphs(tmp_path):
    """Two data-parallel processes (gloo; both on the one GPU of the box) in bench.py's order -- scene, HIP-graph capture, then
    init_process_group, DDP: the steps replay the captured graphs (no capture inside a step), every rank ends with the same
    all-reduced gradients and the same weights, and the ranks' losses differ (different targets)."""
    import socket
    import subprocess
    import sys

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ddp_gpu_worker.py")
    outs = [str(tmp_path / f"r{r}.json") for r in range(2)]
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(port), outs[r]], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    logs = [p.communicate(timeout=600)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\\n".join(l[-2000:] for l in logs)
    import json

    a, b_ = (json.load(open(o)) for o in outs)
    assert a["n_graphs"] == b_["n_graphs"] == 1 and a["graphs_after"] == b_["graphs_after"] == 1
    assert all(np.isfinite(v) for v in a["losses"] + b_["losses"]) and a["losses"] != b_["losses"]
    for n in a["grad_sums"]:
        assert abs(a["grad_sums"][n] - b_["grad_sums"][n]) <= 1e-6 * max(1.0, abs(a["grad_sums"][n])), n
        assert abs(a["weight_sums"][n] - b_["weight_sums"][n]) <= 1e-9 * max(1.0, abs(a["weight_sums"][n])), n


@pytest.mark.parametrize("M", [4096, 128 * 37 + 32, 32 * 77 + 9])
def test_mfma_gemm_with_relu_adjoint_epilogue(M, dev, ops):
    """csrc/gemm.hip (fp32 MFMA, hand-written): (A . B) * (X > 0) against torch mm + threshold_backward; exact fp32 products, so
    only the summation order differs (1e-6 relative), and the zero pattern must be identical."""
    g = torch.Generator().manual_seed(M)
    a = torch.randn(M, 256, generator=g).to(dev)
    b = (torch.randn(256, 256, generator=g) * 0.05).to(dev)
    x = torch.randn(M, 256, generator=g).to(dev)
    ref = a.mm(b)
    plain = ops.gemm_nn_relumask(a, b, None)
    assert float((plain - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    masked = ops.gemm_nn_relumask(a, b, x)
    want = torch.ops.aten.threshold_backward(ref, x, 0)
    assert torch.equal(masked != 0, want != 0)
    assert float((masked - want).abs().max()) <= 2e-5 * float(ref.abs().max())
    # other K take the block-tiled kernel (K % 32 == 0; the mask needs K >= 256)
    for K, with_mask in ((64, False), (512, True)):
        a2 = torch.randn(M, K, generator=g).to(dev)
        b2 = (torch.randn(K, 256, generator=g) * 0.05).to(dev)
        ref2 = a2.mm(b2)
        got = ops.gemm_nn_relumask(a2, b2, x if with_mask else None)
        want2 = torch.ops.aten.threshold_backward(ref2, x, 0) if with_mask else ref2
        assert float((got - want2).abs().max()) <= 2e-5 * float(ref2.abs().max()), K
        assert torch.equal(got != 0, want2 != 0)


@pytest.mark.parametrize("with_feat", [False, True])
def test_field_stack_matches_layer_by_layer_path(with_feat, dev):
    """hostnets._FieldStack (one autograd node, MFMA input-gradient GEMMs with the ReLU adjoint fused) against the same network
    evaluated layer by layer (USE_FIELD_STACK off): outputs, input gradient and every parameter gradient."""
    hostnets = importlib.import_module("3danimals_amd.hostnets")
    torch.manual_seed(1)
    net = hostnets.CoordMLP(3, 16, 5, nf=256, n_harmonic_functions=8, extra_feat_dim=256 if with_feat else 0, activation="sigmoid",
                            min_max=torch.tensor([[0.0, 1.0]] * 16), symmetrize=not with_feat).to(dev)
    g = torch.Generator().manual_seed(3)
    P, B = hostnets.SPLITK_MIN_ROWS + 8192, 4
    x0 = (torch.rand(P, 3, generator=g) * 2 - 1).to(dev)
    feat0 = torch.randn(B, 256, generator=g).to(dev)
    idx = torch.randint(0, B, (P,), generator=g).sort().values.to(dev)
    w = torch.rand(P, 16, generator=g).to(dev)

    def run(use_stack):
        hostnets.USE_FIELD_STACK = use_stack
        try:
            x = x0.clone().requires_grad_(True)
            feat = feat0.clone().requires_grad_(True) if with_feat else None
            out = net.sample(x, feat=feat, feat_index=idx) if with_feat else net.sample(x)
            leaves = [x] + ([feat] if with_feat else []) + list(net.parameters())
            return out.detach(), torch.autograd.grad((out * w).sum(), leaves)
        finally:
            hostnets.USE_FIELD_STACK = True

    (oa, ga), (ob, gb) = run(True), run(False)
    assert torch.allclose(oa, ob, atol=2e-6)
    for u, v in zip(ga, gb):
        assert float((u - v).abs().max()) <= 1e-4 * float(v.abs().max()) + 1e-7


# ------------------------------------------------------------------------------------------------ mixed precision (round 4)
class _StubField(torch.nn.Module):
    """Stands for a texture / DINO / light network evaluated under the caller's autocast: a deterministic elementwise function of its
    inputs whose values are bf16 numbers -- returned AS bf16 when autocast is on (what an nn.Linear gives there), as the same numbers in
    float32 otherwise.  So both runs hand the path identical values, in the dtype the reference's call pattern produces."""

    def __init__(self, channels):
        super().__init__()
        self.coef = torch.nn.Parameter(torch.linspace(0.3, 2.9, channels * 3).reshape(channels, 3))
        self.dtypes_seen = []

    def _out(self, y):
        y = y.bfloat16()
        self.dtypes_seen.append(torch.is_autocast_enabled())
        return y if torch.is_autocast_enabled() else y.float()

    def sample(self, x, feat=None, feat_index=None):
        y = 0.5 + 0.5 * torch.sin((x.float()[:, None, :] * self.coef[None]).sum(-1))
        if feat is not None:
            y = y * (0.75 + 0.25 * torch.tanh(feat.float().mean(-1, keepdim=True)))
        return self._out(y)

    def forward(self, feat):  # as a light: [B,5] = direction(3), ambient, diffuse
        f = feat.float().mean(-1, keepdim=True)
        d = torch.nn.functional.normalize(torch.cat([torch.sin(f), torch.cos(f), 1 + 0 * f], -1), dim=-1)
        return self._out(torch.cat([d, 0.3 + 0 * f, 0.6 + 0.1 * torch.tanh(f)], -1))


@pytest.mark.parametrize("amp_dtype", [torch.bfloat16, torch.float16])
def test_path_inside_autocast_returns_the_float32_results(amp_dtype, dev, mods):
    """The reference's mixed-precision call pattern (Trainer.py:208-218: mixed_precision bf16 / fp16; AnimalModel.py:382-444: netBase /
    netInstance / render / losses inside torch.autocast; .float() only at the nvdiffrast boundary, render.py:265,292; GradScaler-scaled
    loss, AnimalModel.py:192-204): estimate_bones, skinning, make_mesh (+ normals) and render_mesh (+ backward) called inside
    torch.autocast with half-precision image features and half-precision network outputs arriving at shade -- every output float32 and
    equal to the run without autocast on the same numbers; gradients finite, in the dtype of their leaves, equal to the float32 run's
    up to the rounding of the half-precision leaves' gradients."""
    B, H, W = 2, 64, 64
    verts, faces, _, (mvp, w2c, campos) = _scene(B, seed=3)
    sk, M, R = mods["skinning"], mods["mesh"], mods["render"]
    bones, tree, _ = sk.estimate_bones(verts[None, None], n_body_bones=8, n_legs=4, n_leg_bones=3, body_bones_mode="z_minmax_y+")
    uvs = torch.zeros(1, 4, 2, device=dev)
    uvi = torch.zeros(1, faces.shape[0], 3, dtype=torch.int64, device=dev)
    bg = seeded((B, H, W, 3), 13, 0, 1).to(dev)
    feat32 = seeded((B, 16), 12, -1, 1).to(amp_dtype).float()  # numbers that exist in the half-precision format

    def run(amp):
        tex, dino, lgt = _StubField(9).to(dev), _StubField(16).to(dev), _StubField(5).to(dev)
        rest = verts[None, None].to(dev).clone().requires_grad_(True)
        ang = seeded((B, 1, 20, 3), 9, -0.4, 0.4).to(dev).requires_grad_(True)
        cam = mvp.to(dev).clone().requires_grad_(True)
        feat = (feat32.to(dev).to(amp_dtype) if amp else feat32.to(dev)).requires_grad_(True)
        with torch.autocast("cuda", dtype=amp_dtype, enabled=amp):
            bones_d = sk.estimate_bones(rest.detach(), n_body_bones=8, n_legs=4, n_leg_bones=3, body_bones_mode="z_minmax_y+")[0]
            posed, aux = sk.skinning(rest, bones_d, tree, ang, output_posed_bones=True, temperature=0.05)
            shape = M.make_mesh(posed.view(B, -1, 3), faces[None].to(dev), uvs.expand(B, -1, -1), uvi, None)
            prior = M.make_mesh(rest.view(1, -1, 3), faces[None].to(dev), uvs, uvi, None)
            out = R.render_mesh(None, shape, cam, w2c.to(dev), campos.to(dev), tex, lgt, (H, W), background=bg, bsdf="diffuse", feat=feat,
                                render_modes=["shaded", "dino_pred", "geo_normal"], prior_mesh=prior, dino_net=dino)
            loss = sum((o.float() * seeded(tuple(o.shape), 40 + i, -1, 1).to(dev)).sum() for i, o in enumerate(out)) + posed.sum() + aux["posed_bones"].sum()
            loss = loss + shape.v_nrm.sum()
        scale = 1024.0 if amp else 1.0  # (a GradScaler multiplies the loss before backward and divides the gradients afterwards)
        grads = torch.autograd.grad(loss * scale, [rest, ang, cam, feat] + list(tex.parameters()) + list(dino.parameters()))
        assert all(tex.dtypes_seen) == amp and len(tex.dtypes_seen) > 0
        return bones_d, posed, aux["posed_bones"], shape.v_nrm, out, [g / scale for g in grads], feat

    ref, amp = run(False), run(True)
    for a, b in zip(ref[:4], amp[:4]):
        assert b.dtype == torch.float32 and torch.allclose(a, b, atol=1e-6), float((a - b).abs().max())
    for name, a, b in zip(("shaded", "dino_pred", "geo_normal"), ref[4], amp[4]):
        assert b.dtype == torch.float32, name
        assert float((a - b).abs().max()) < 1e-5, (name, float((a - b).abs().max()))
    assert amp[6].dtype == amp_dtype and amp[5][3].dtype == amp_dtype  # the half-precision leaf gets a half-precision gradient
    for i, (a, b) in enumerate(zip(ref[5], amp[5])):
        assert torch.isfinite(b.float()).all()
        tol = (2e-2 if amp_dtype == torch.bfloat16 else 4e-3) if i == 3 else 2e-4  # (the feature's gradient is rounded to its dtype)
        assert float((a.float() - b.float()).abs().max()) <= tol * float(a.abs().max()) + 1e-6, (i, float((a.float() - b.float()).abs().max()), float(a.abs().max()))


@pytest.mark.parametrize("H,W,kind", [(256, 256, "slivers"), (512, 384, "slivers"), (256, 256, "eye_plane"), (96, 1024, "slivers")])
def test_rasterize_large_boxes_through_the_tile_stage_bit_exact_vs_oracle(H, W, kind, dev, ops):
    """Boxes above 512 pixels are not walked pixel by pixel: their 8x8 tiles are tested against the three (linear) edge functions first and
    only the tiles the triangle can touch are (csrc/raster.hip, round 4).  A tile wrongly rejected would lose covered pixels: random
    long slivers (boxes of 1e3 .. 1e5 pixels for areas of a few hundred), near-degenerate ones, triangles with unequal w, triangles
    that straddle the eye plane (box = the whole frame), mixed with small ones, over several images -- triangle ids and (u, v, z/w)
    bit-exact against oracle/raster_ref.c, which walks every pixel of every box."""
    from oracle import raster_ref

    g = torch.Generator().manual_seed(H + W + len(kind))
    B, n_big, n_small = 3, 60, 400
    centre = torch.rand(B, n_big, 1, 2, generator=g) * 2.4 - 1.2
    direction = torch.nn.functional.normalize(torch.randn(B, n_big, 1, 2, generator=g), dim=-1)
    normal = torch.stack([-direction[..., 1], direction[..., 0]], -1)
    length = 0.2 + 1.8 * torch.rand(B, n_big, 1, 1, generator=g)
    width = 10 ** (-3.5 + 2.5 * torch.rand(B, n_big, 1, 1, generator=g))  # 3e-4 .. 1e-1 of the frame
    t = torch.tensor([[-1.0, 0.0], [1.0, -1.0], [0.3, 1.0]]).reshape(1, 1, 3, 2)
    xy_big = centre + direction * length * t[..., :1] + normal * width * t[..., 1:]
    xy_small = (torch.rand(B, n_small, 1, 2, generator=g) * 2 - 1) + 0.03 * torch.randn(B, n_small, 3, 2, generator=g)
    xy = torch.cat([xy_big, xy_small], 1)
    n = n_big + n_small
    z = torch.rand(B, n, 3, 1, generator=g) * 1.6 - 0.8
    w = 0.4 + 2.0 * torch.rand(B, n, 3, 1, generator=g)
    if kind == "eye_plane":
        w[:, :6, 0] *= -1.0  # six triangles per image with one vertex behind the eye: every pixel of the frame is a candidate
    clip = torch.cat([xy * w, z * w, w], -1).reshape(B, 3 * n, 4).contiguous()
    tri = torch.arange(3 * n, dtype=torch.int32).reshape(n, 3)
    tri = tri[torch.randperm(n, generator=g)].contiguous()
    ref = raster_ref.rasterize(clip, tri, (H, W))
    out = ops.rasterize(clip.to(dev), tri.to(dev), (H, W)).cpu()
    assert np.array_equal(out[..., 3].numpy(), ref[..., 3].numpy())
    assert np.array_equal(out.numpy(), ref.numpy())
    big_ids = set((torch.nonzero((tri[:, 0] < 3 * n_big))[:, 0] + 1).tolist())
    seen = set(out[..., 3].unique().int().tolist())
    assert len(seen & big_ids) > n_big // 4 and float((out[..., 3] > 0).float().mean()) > 0.02  # the slivers really are on screen


@pytest.mark.parametrize("with_bg", [False, True])
def test_texture_less_render_through_the_mask_compositor_equals_the_general_path(with_bg, dev, mods, monkeypatch):
    """render_mesh(material = None, lgt = None, render_modes = ['shaded']) -- Fauna's random-view mask render (Fauna.py:111-173) -- takes
    its coverage straight from the raster texels (a3d_mask_aa_*: no covered-pixel list, no G-buffer, no shading launch, no host read-back of
    the list length) instead of compositing [P,3] rows of ones: same image, same gradient to the vertices, and the calls it makes."""
    _lib = importlib.import_module("3danimals_amd._lib")
    B, H, W = 3, 96, 96
    verts, faces, _, (mvp, w2c, campos) = _scene(B, seed=9)
    M, R = mods["mesh"], mods["render"]
    uvs = torch.zeros(1, 4, 2, device=dev)
    uvi = torch.zeros(1, faces.shape[0], 3, dtype=torch.int64, device=dev)
    bg = seeded((B, H, W, 3), 13, 0, 1).to(dev) if with_bg else None
    wgt = seeded((B, 4, H, W), 17, -1, 1).to(dev)

    def run(fast):
        monkeypatch.setattr(R, "FUSED_MASK_RENDER", fast)
        posed = (verts[None] + 0.05 * seeded((B, *verts.shape), 11, -1, 1)).to(dev).requires_grad_(True)
        shape = M.make_mesh(posed, faces[None].to(dev), uvs.expand(B, -1, -1), uvi, None)
        _ = shape.v_nrm  # (normals computed before the render, as in the step: this is the posed meshes' SECOND render)
        with _lib.KernelTimer() as timer:
            out = R.render_mesh(None, shape, mvp.to(dev), w2c.to(dev), campos.to(dev), None, None, (H, W), background=bg, bsdf="diffuse",
                                render_modes=["shaded"], two_sided_shading=False)[0]
            (g,) = torch.autograd.grad((out * wgt).sum(), posed, retain_graph=True)
        # Fauna's use of it (Fauna.py:166-173): the alpha channel alone, clamped -- the engine hands the op a channels-FIRST zero-padded
        # gradient seen through the inverse permute, which a3d_mask_aa_bwd reads in place (g_channels_first)
        (g_alpha,) = torch.autograd.grad((out[:, 3:].clamp(0, 1) * wgt[:, 3:]).sum(), posed)
        return out.detach(), g, sorted(timer.summary()), g_alpha

    out_f, g_f, calls_f, ga_f = run(True)
    out_g, g_g, calls_g, ga_g = run(False)
    assert float(ga_g.abs().max()) > 0 and float((ga_f - ga_g).abs().max()) <= 1e-4 * float(ga_g.abs().max())
    assert out_f.shape == (B, 4, H, W) and float((out_f - out_g).abs().max()) < 1e-6
    assert float((out_f[:, 3] > 0).float().mean()) > 0.05 and float(((out_f[:, 3] > 0.01) & (out_f[:, 3] < 0.99)).float().mean()) > 1e-3
    assert float(g_g.abs().max()) > 0 and float((g_f - g_g).abs().max()) <= 1e-4 * float(g_g.abs().max())
    assert calls_f == ["a3d_mask_aa_bwd[C4]", "a3d_mask_aa_fwd[C4][+analysis]", "a3d_rast_fwd", "a3d_xfm_points_bwd", "a3d_xfm_points_fwd"], calls_f
    assert any(c.startswith(("a3d_cover_gbuffer_fwd", "a3d_rast_resolve_gbuffer_fwd")) for c in calls_g) and any(c.startswith("a3d_composite_aa_fwd") for c in calls_g)


@pytest.mark.parametrize("Bp,Bm", [(4, 4), (1, 4), (4, 1)])
def test_xfm_points_kernel_matches_the_float64_matmul_and_its_gradients(Bp, Bm, dev, ops):
    """a3d_xfm_points_fwd / _bwd (round 6: one launch each way for xfm_points' pad + bmm, renderutils/ops.py:515-531) against the
    definition in float64: values, d/d points, d/d matrix -- shared points, a shared matrix, and the gradient arriving as the clip
    columns of 16-float rows (the G-buffer backward's layout, read in place)."""
    V, B = 777, max(Bp, Bm)
    pts = seeded((Bp, V, 3), 3, -1, 1).to(dev).requires_grad_(True)
    mtx = (torch.eye(4)[None] + 0.3 * seeded((Bm, 4, 4), 4, -1, 1)).to(dev).requires_grad_(True)
    out = ops.xfm_points(pts, mtx)
    p64, m64 = pts.detach().double().requires_grad_(True), mtx.detach().double().requires_grad_(True)
    ref = torch.matmul(torch.nn.functional.pad(p64, (0, 1), value=1.0), m64.transpose(1, 2))
    assert out.shape == (B, V, 4) and float((out.double() - ref).abs().max()) < 2e-6
    rows = seeded((B, V, 16), 5, -1, 1).to(dev)
    for g in (rows[..., 12:16], rows[..., 12:16].contiguous()):  # strided (read in place) and contiguous
        gp, gm = torch.autograd.grad(out, [pts, mtx], g, retain_graph=True)
        rp, rm = torch.autograd.grad(ref, [p64, m64], g.double(), retain_graph=True)
        assert float((gp.double() - rp).abs().max()) < 1e-5 * max(1.0, float(rp.abs().max()))
        assert float((gm.double() - rm).abs().max()) < 1e-4 * max(1.0, float(rm.abs().max()))


def test_xfm_points_aliases_sum_their_gradients_inside_the_backward_launch(dev, ops):
    """ops.xfm_points(alias=True / 2): the points come back as a second (and third) output of the clip transform's node for their other
    consumers in render_mesh -- the G-buffer's position attribute (gradient = columns of 16-float rows, read in place) and the vertex
    normals (contiguous) -- and every gradient is added inside a3d_xfm_points_bwd: the same sum as autograd's accumulation of three
    separate consumers, in float64; any subset of the three may be missing."""
    B, V = 3, 515
    pts = seeded((B, V, 3), 3, -1, 1).to(dev).requires_grad_(True)
    mtx = (torch.eye(4)[None] + 0.3 * seeded((B, 4, 4), 4, -1, 1)).to(dev).requires_grad_(True)
    rows = seeded((B, V, 16), 5, -1, 1).to(dev)
    g_n = seeded((B, V, 3), 6, -1, 1).to(dev)
    p64, m64 = pts.detach().double().requires_grad_(True), mtx.detach().double().requires_grad_(True)
    ref = torch.matmul(torch.nn.functional.pad(p64, (0, 1), value=1.0), m64.transpose(1, 2))
    for use in ((1, 1, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1), (0, 0, 1), (0, 1, 0)):
        clip, again, third = ops.xfm_points(pts, mtx, alias=2)
        assert again.data_ptr() == pts.data_ptr() == third.data_ptr() and again is not third
        loss = 0.0
        loss64 = 0.0
        if use[0]:
            loss, loss64 = loss + (clip * rows[..., 12:16]).sum(), loss64 + (ref * rows[..., 12:16].double()).sum()
        if use[1]:
            loss, loss64 = loss + (again * rows[..., 0:3]).sum(), loss64 + (p64 * rows[..., 0:3].double()).sum()
        if use[2]:
            loss, loss64 = loss + (third * g_n).sum(), loss64 + (p64 * g_n.double()).sum()
        (gp,) = torch.autograd.grad(loss, [pts])
        (rp,) = torch.autograd.grad(loss64, [p64], retain_graph=True)
        assert float((gp.double() - rp).abs().max()) < 1e-5 * max(1.0, float(rp.abs().max())), use
    clip, again = ops.xfm_points(pts, mtx, alias=True)  # (the two-output form is unchanged)
    assert again.data_ptr() == pts.data_ptr()
    # alias = 3: the clip positions a second time (the antialiasing's operand): both clip gradients summed before the transform, also d/d matrix
    g2 = seeded((B, V, 4), 7, -1, 1).to(dev)
    for use in ((1, 1), (0, 1), (1, 0)):
        clip, again, third, clip2 = ops.xfm_points(pts, mtx, alias=3)
        assert clip2.data_ptr() == clip.data_ptr() and clip2 is not clip
        loss = (again * rows[..., 0:3]).sum() + (third * g_n).sum()
        loss64 = (p64 * rows[..., 0:3].double()).sum() + (p64 * g_n.double()).sum()
        if use[0]:
            loss, loss64 = loss + (clip * rows[..., 12:16]).sum(), loss64 + (ref * rows[..., 12:16].double()).sum()
        if use[1]:
            loss, loss64 = loss + (clip2 * g2).sum(), loss64 + (ref * g2.double()).sum()
        gp, gm = torch.autograd.grad(loss, [pts, mtx])
        rp, rm = torch.autograd.grad(loss64, [p64, m64], retain_graph=True)
        assert float((gp.double() - rp).abs().max()) < 1e-5 * max(1.0, float(rp.abs().max())), use
        assert float((gm.double() - rm).abs().max()) < 1e-4 * max(1.0, float(rm.abs().max())), use


def test_render_mesh_flow_mode_with_the_fused_motion_equals_the_torch_expression(dev, mods):
    """render.FUSED_FLOW_DELTA: render_mesh(render_modes=[..., 'flow']) with the per-vertex motion from a3d_flow_delta_* against the same
    call with the reference's torch expression (render.py:281-288): the same 'flow' image (the same float32 operations) and the same
    gradient on the posed vertices up to the order of the backward's float atomics."""
    M, R = mods["mesh"], mods["render"]
    Bs, Fr, H, W = 2, 3, 48, 48
    N = Bs * Fr
    verts, faces, _, (mvp, w2c, campos) = _scene(N, seed=9)
    uvs = torch.zeros(1, 4, 2, device=dev)
    uvi = torch.zeros(1, faces.shape[0], 3, dtype=torch.int64, device=dev)
    tri = faces[None].to(dev)

    def run(fused):
        R.FUSED_FLOW_DELTA = fused
        try:
            posed = (verts[None] + 0.05 * seeded((N, *verts.shape), 33, -1, 1)).to(dev).requires_grad_(True)
            prior = M.make_mesh(verts[None].to(dev), tri, uvs, uvi, None)
            shape = M.make_mesh(posed, tri, uvs.expand(N, -1, -1), uvi, None)
            out = R.render_mesh(None, shape, mvp.to(dev), w2c.to(dev), campos.to(dev), None, None, (H, W), bsdf="diffuse",
                                render_modes=["shaded", "flow"], prior_mesh=prior, num_frames=Fr)
            (g,) = torch.autograd.grad(sum((o * seeded(tuple(o.shape), 40 + i, -1, 1).to(dev)).sum() for i, o in enumerate(out)), posed)
            return out, g
        finally:
            R.FUSED_FLOW_DELTA = True

    (o1, g1), (o0, g0) = run(True), run(False)
    assert o1[1].shape == o0[1].shape and o1[1].shape[1] == 2 and float(o0[1].detach().abs().max()) > 0
    assert float((o1[1] - o0[1]).abs().max()) <= 2.4e-7  # (antialiased: two blends on a pixel add in the order the atomics land)
    assert float((o1[0] - o0[0]).abs().max()) <= 2.4e-7
    np.testing.assert_allclose(g1.cpu().numpy(), g0.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(g0.abs().max()))


@pytest.mark.parametrize("B,F", [(3, 4), (5, 2), (1, 8)])
def test_flow_delta_equals_the_reference_expression_and_its_gradients(B, F, dev, ops):
    """ops.flow_delta (a3d_flow_delta_fwd / _bwd): the per-vertex motion to the next frame of render.py:281-288 -- ndc = clip.xy / clip.w,
    ndc[f+1] - ndc[f], zeros for the last frame of a sequence -- against the torch expression the reference writes: the same values
    (the same float32 divisions and subtraction: bit for bit) and the expression's own gradient in float64."""
    V, N = 333, B * F
    clip = seeded((N, V, 4), 11, -1, 1)
    clip[..., 3] = 5.0 + clip[..., 3]  # (w of a vertex in front of the camera)
    clip = clip.to(dev).requires_grad_(True)

    def expression(c, frames):
        ndc = c[..., :2] / c[..., -1:]
        ndc = ndc.view(-1, frames, *ndc.shape[1:])
        d = ndc[:, 1:] - ndc[:, :-1]
        return torch.cat([d, torch.zeros_like(d[:, :1])], dim=1).view(-1, *ndc.shape[2:])

    got = ops.flow_delta(clip, F)
    ref = expression(clip, F)
    assert got.shape == ref.shape == (N, V, 2) and torch.equal(got, ref)
    rows = seeded((N, V, 16), 12, -1, 1).to(dev)
    w = rows[..., 9:11]  # (strided, as the G-buffer backward hands the attribute's gradient out: read in place)
    (g,) = torch.autograd.grad(got, [clip], grad_outputs=w, retain_graph=True)
    (g_c,) = torch.autograd.grad(got, [clip], grad_outputs=w.contiguous())
    assert torch.equal(g, g_c)
    c64 = clip.detach().double().requires_grad_(True)
    (r,) = torch.autograd.grad((expression(c64, F) * w.double()).sum(), [c64])
    assert g.shape == clip.shape and float(g[..., 2].abs().max()) == 0.0
    assert float((g.double() - r).abs().max()) < 1e-6 * max(1.0, float(r.abs().max()))


def test_compositor_hands_out_kept_channels_reads_strided_gradients_and_short_backgrounds(dev, ops, mods):
    """Round 6 plumbing of a3d_composite_aa_*: ``keep`` materialises only the leading channels a mode returns (dino_pred / flow without
    alpha) -- same values as the slice of the full image, same gradients as through the slice; value rows may carry padding rows behind
    the list (their gradient rows come back zero); a 3-channel background reads as the 4-channel one with zero alpha; an image gradient
    that is a channel slice of a wider buffer is read in place."""
    B, H, W = 2, 64, 64
    verts, faces, _, (mvp, w2c, campos) = _scene(B, seed=4)
    ru = importlib.import_module("3danimals_amd.model.render.renderutils")
    clip = ru.xfm_points((verts[None] + 0.05 * seeded((B, *verts.shape), 8, -1, 1)).to(dev), mvp.to(dev)).contiguous()
    tri = faces.to(dev)
    rast = ops.rasterize(clip, tri, (H, W)).detach()
    pix, inv = ops.covered_pixels(rast, return_inverse=True)
    P = pix.shape[0]
    pad = 37
    vals = seeded((P + pad, 5), 9, 0, 1).to(dev).requires_grad_(True)  # (padding rows behind the list)
    bg3 = seeded((B, H, W, 3), 10, 0, 1).to(dev)
    bg4 = torch.cat((bg3, torch.zeros_like(bg3[..., :1])), -1)
    col = seeded((P, 3), 11, 0, 1).to(dev).requires_grad_(True)
    clip_g = clip.clone().requires_grad_(True)

    def analysis():
        return ops.AAAnalysis(rast, clip_g, ops.aa_topology(ops.tri_int32(tri), clip.shape[1]))

    full_c, full_v = ops.composite_antialias(col, pix, inv, bg4, clip_g, analysis(), vals2=vals[:P])
    kept_c, kept_v = ops.composite_antialias(col, pix, inv, bg3, clip_g, analysis(), vals2=vals, keep2=5)
    assert kept_v.shape == (B, H, W, 5) and kept_v.is_contiguous() and full_v.shape == (B, H, W, 6)
    assert float((kept_v - full_v[..., :5]).abs().max()) <= 2.4e-7 and float((kept_c - full_c).abs().max()) <= 2.4e-7
    wide = seeded((B, H, W, 9), 12, -1, 1).to(dev)
    gc = seeded((B, H, W, 4), 13, -1, 1).to(dev)
    ga = torch.autograd.grad([full_c, full_v[..., :5]], [col, vals, clip_g], [gc, wide[..., 2:7].contiguous()])
    gb = torch.autograd.grad([kept_c, kept_v], [col, vals, clip_g], [gc, wide[..., 2:7]])  # (a channel slice of a wider buffer: read in place)
    assert float(gb[1][P:].abs().max()) == 0.0  # padding rows: zero gradient rows
    for x, y in zip(ga, gb):
        assert float((x - y).abs().max()) <= 1e-5 * max(1.0, float(x.abs().max()))


def test_reconstruction_losses_read_a_strided_feature_image_in_place(dev, ops):
    """a3d_recon_losses_* with dino_stride = D + 1 (the 17-channel image read in place, its gradient written into the same layout) equal
    the contiguous form: losses bit for bit, gradients bit for bit."""
    B, H, W, D = 2, 48, 40, 16
    sh = seeded((B, H, W, 4), 1, 0, 1).to(dev)
    wide = seeded((B, H, W, D + 1), 2, 0, 1).to(dev)
    img, dgt = seeded((B, 3, H, W), 3, 0, 1).to(dev), seeded((B, D, H, W), 4, 0, 1).to(dev)
    mask = (seeded((B, H, W), 5, 0, 1) > 0.4).float().to(dev)
    dt = seeded((B, 2, H, W), 6, 0, 1).to(dev)
    valid = torch.ones(B, H, W, device=dev)
    outs = []
    for strided in (True, False):
        s_ = sh.clone().requires_grad_(True)
        w_ = wide.clone().requires_grad_(True)
        d_nhwc = w_[..., :D] if strided else w_[..., :D].contiguous()
        loss = ops.reconstruction_losses(s_.permute(0, 3, 1, 2), d_nhwc.permute(0, 3, 1, 2), img, dgt, mask, dt, valid)
        gs, gw = torch.autograd.grad((loss * seeded(tuple(loss.shape), 7, 0.5, 1.5).to(dev)).sum(), [s_, w_])
        outs.append((loss.detach(), gs, gw))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert float(outs[0][2][..., D].abs().max()) == 0.0 and float(outs[0][2][..., :D].abs().max()) > 0  # (nothing flows to the alpha slot)


def test_surface_sdf_splice_and_padded_row_gather(dev, ops):
    """ops.gather_rows_padded / ops.surface_sdf (a3d_dmtet_gather_rows): what DMTetGeometry._get_mesh_surface_backward used torch's index,
    pad, slice and index_add for -- same values, same gradient."""
    Nv, n, rows = 5000, 700, 1024
    pos = seeded((Nv, 3), 1, -1, 1).to(dev)
    idx = torch.sort(torch.randperm(Nv, generator=torch.Generator().manual_seed(2))[:n])[0].to(dev)
    got = ops.gather_rows_padded(pos, idx, rows)
    assert got.shape == (rows, 3) and torch.equal(got[:n], pos[idx]) and float(got[n:].abs().max()) == 0.0
    sdf0 = seeded((Nv, 1), 3, -1, 1).to(dev)
    sub = seeded((rows, 1), 4, -1, 1).to(dev).requires_grad_(True)
    w = seeded((Nv, 1), 5, -1, 1).to(dev)
    cur = ops.surface_sdf(sdf0, idx, sub)
    ref = sdf0.index_add(0, idx, sub[:n] - sub[:n].detach())
    assert torch.equal(cur, ref)
    (g,) = torch.autograd.grad((cur * w).sum(), sub)
    (gr,) = torch.autograd.grad((ref * w).sum(), sub)
    assert torch.equal(g, gr)


def test_shading_inside_the_compositor_equals_the_separate_launch(dev, mods, monkeypatch):
    """render.SHADE_IN_COMPOSITOR: when only the composited colour reads what a3d_shade_fwd computes (the training modes), the launch is
    skipped and the compositor computes kd * shading per covered pixel itself (a3d_ca_shade: one shared device function, same bits).
    Same images, same gradients to vertices, camera, light and texture parameters; one hot-path call less; and a mode that needs the
    launch's other outputs ('normal', 'shading') still gets them."""
    import copy

    _lib = importlib.import_module("3danimals_amd._lib")
    B, H, W = 2, 64, 64
    verts, faces, _, (mvp, w2c, campos) = _scene(B, seed=3)
    M, R = mods["mesh"], mods["render"]
    tex0, dino0, lgt0 = _nets(mods, dev)
    feat = seeded((B, 16), 12, -1, 1).to(dev)
    bg = seeded((B, H, W, 3), 13, 0, 1).to(dev)
    uvs = torch.zeros(1, 4, 2, device=dev)
    uvi = torch.zeros(1, faces.shape[0], 3, dtype=torch.int64, device=dev)

    def run(on, modes):
        monkeypatch.setattr(R, "SHADE_IN_COMPOSITOR", on)
        tex, dino, lgt = (copy.deepcopy(m).to(dev) for m in (tex0, dino0, lgt0))
        posed = (verts[None] + 0.05 * seeded((B, *verts.shape), 11, -1, 1)).to(dev).requires_grad_(True)
        cam = w2c.to(dev).clone().requires_grad_(True)
        shape = M.make_mesh(posed, faces[None].to(dev), uvs.expand(B, -1, -1), uvi, None)
        prior = M.make_mesh(verts[None].to(dev), faces[None].to(dev), uvs, uvi, None)
        with _lib.KernelTimer() as timer:
            out = R.render_mesh(None, shape, mvp.to(dev), cam, campos.to(dev), tex, lgt, (H, W), background=bg, bsdf="diffuse", feat=feat,
                                render_modes=modes, prior_mesh=prior, dino_net=dino)
            loss = sum((o * seeded(tuple(o.shape), 40 + i, -1, 1).to(dev)).sum() for i, o in enumerate(out))
            grads = torch.autograd.grad(loss, [posed, cam] + list(tex.parameters()) + list(lgt.parameters()))
        return [o.detach() for o in out], grads, sorted(timer.summary())

    for modes in (["shaded", "dino_pred"], ["dino_pred", "shaded"], ["shaded", "normal", "shading"]):
        o_on, g_on, calls_on = run(True, modes)
        o_off, g_off, calls_off = run(False, modes)
        for a, b in zip(o_on, o_off):
            assert float((a - b).abs().max()) < 1e-6, modes
        for a, b in zip(g_on, g_off):
            assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()) + 1e-7, modes
        deferred = modes == ["shaded", "dino_pred"]  # (only the FIRST buffer of a compositor call, and only the training modes)
        assert any("[+shade]" in c for c in calls_on) == deferred, (modes, calls_on)
        assert ("a3d_shade_fwd" in calls_on) == (not deferred) and "a3d_shade_fwd" in calls_off


@pytest.mark.parametrize("numbering", ["spatial", "random"])
def test_normals_backward_faces_first_is_bit_identical_to_the_gather_form(numbering, dev, ops, mods, monkeypatch):
    """a3d_normals_bwd with face_scratch (every face's corner adjoints once, then a per-vertex sum in ascending key order) against the
    gather form (every vertex re-derives its incident faces' adjoints): the same float values added in the same order -- equal bits --
    on a spatially numbered marching-tets surface and on the randomly numbered, randomly wound one a scrambled grid file gives; CSR and
    fixed-stride lists; valences above eight (a fan)."""
    from oracle import dmtet_ref

    tg = importlib.import_module("3danimals_amd.tetgrid")
    pos, tets = kuhn(14)
    if numbering == "random":
        p_np, t_np = tg.scramble(pos.numpy(), tets.numpy(), 4)
        pos, tets = torch.from_numpy(p_np), torch.from_numpy(t_np).long()
    ext = float((pos.amax(0) - pos.amin(0)).max())
    sdf = 0.31 * ext - (pos - pos.mean(0)).norm(dim=1) + 0.02 * ext * seeded((pos.shape[0],), 5, -1, 1)
    verts, faces, _, _ = dmtet_ref.marching_tets(pos, sdf, tets)
    assert verts.shape[0] > 500
    fan = torch.tensor([[0, k, k + 1] for k in range(1, 14)])  # vertex 0 gains 13 more faces: valence > 8
    meshes = [(verts, faces), (verts, torch.cat([faces, fan]))]
    B = 3
    for v0, f in meshes:
        v = (v0[None] + 0.01 * seeded((B, *v0.shape), 8, -1, 1)).to(dev)
        g = seeded((B, v0.shape[0], 3), 9, -1, 1).to(dev)
        outs = []
        for faces_first in (True, False):
            monkeypatch.setattr(ops, "NORMALS_FACES_FIRST", faces_first)
            vv = v.clone().requires_grad_(True)
            n = ops.vertex_normals(vv, f.to(dev))
            (gv,) = torch.autograd.grad((n * g).sum(), vv)
            outs.append(gv)
        assert float(outs[0].abs().max()) > 0 and torch.equal(outs[0], outs[1])
    # ... and through the fixed-stride lists a DMTet extraction leaves (the training path)
    grid = mods["dmtet"].TetGridTopology(tets.to(dev), positions=pos.to(dev))
    for _ in range(2):
        ve, fa, _ = ops.dmtet(pos.to(dev), sdf.to(dev), grid)
    outs = []
    for faces_first in (True, False):
        monkeypatch.setattr(ops, "NORMALS_FACES_FIRST", faces_first)
        vv = (ve[None].expand(B, -1, -1) + 0.0).clone().requires_grad_(True)
        n = ops.vertex_normals(vv, fa)
        (gv,) = torch.autograd.grad((n * g[:, : ve.shape[0]]).sum(), vv)
        outs.append(gv)
    assert torch.equal(outs[0], outs[1])


# ------------------------------------------------------------------------------------------------ guard mode (round 5)
@pytest.fixture
def guard():
    """Switches A3D_GUARD on for one test (3danimals_amd/_lib.py: canaries around every buffer ops.py hands to the library, compared after
    every entry point; level 2 also poisons the payload) and off again afterwards."""
    L = importlib.import_module("3danimals_amd._lib")
    yield L.set_guard
    L.set_guard(0)


def test_guard_mode_reports_a_one_element_overrun_at_the_call_that_did_it(dev, ops, guard):
    """The detector itself: a library call told that its output holds one texel more than it does writes 16 bytes past the end; guard
    mode raises at that call, names the buffer and the side, and repairs the canary."""
    L = importlib.import_module("3danimals_amd._lib")
    guard(1)
    B, H, W = 1, 16, 16
    clip = torch.tensor([[[-0.5, -0.5, 0.5, 1.0], [0.5, -0.5, 0.5, 1.0], [0.0, 0.5, 0.5, 1.0]]], device=dev)
    tri = torch.tensor([[0, 1, 2]], dtype=torch.int32, device=dev)
    rast = ops.rasterize(clip, tri, (H, W))  # a correct call: no complaint
    assert int((rast[..., 3] > 0).sum()) > 0 and L.guard_stats["checks"] > 0
    small = ops.torch.empty((B, H - 1, W, 4), dtype=torch.float32, device=dev)  # one row short of what the call is told
    keys = ops.torch.empty(L.lib().a3d_rast_scratch_bytes(B, H, W), dtype=torch.uint8, device=dev)
    with pytest.raises(L.A3DError, match="AFTER the buffer"):
        L.call("a3d_rast_fwd", L.ptr(clip), 1, L.ptr(tri), B, 3, 1, H, W, L.ptr(small), L.ptr(keys), 0, None, L.stream())
    L.guard_check("after the repair")  # the canaries were restored: the next check is clean


@pytest.mark.parametrize("workload,kw", [("magicpony", dict(deform=True)), ("fauna", {}), ("magicpony", dict(deform=True, mesh="spiky")),
                                         ("ponymation", dict(num_frames=4, batch=4))], ids=["magicpony", "fauna", "magicpony-spiky", "ponymation"])
def test_guard_mode_finds_no_overrun_in_300_training_steps(workload, kw, dev, guard):
    """the full-size training steps (B = 16, 256x256, Kuhn R = 64; Fauna = the step that once aborted with an HSA
    hardware exception) for 300 optimiser steps each with every library buffer between canaries, checked after every entry point -- the
    mesh drifts, so the data-dependent capacities (speculative DMTet emit, covered-pixel buckets, silhouette record list, G-buffer
    backward tables) move through many values.  (300 steps x ~60 calls x one device synchronisation: about a minute per workload.)"""
    L = importlib.import_module("3danimals_amd._lib")
    pipeline = importlib.import_module("3danimals_amd.pipeline")
    kw = dict(kw)
    scene = pipeline.SyntheticScene(grid_res=64, batch=kw.pop("batch", 16), resolution=(256, 256), device=dev, seed=0, workload=workload, **kw)
    steps = 300 if workload != "ponymation" else 60
    guard(1)
    before = dict(L.guard_stats)
    for i in range(steps):
        out = scene.step(backward=True)
        if i % 50 == 0:
            assert bool(torch.isfinite(out["loss"])), i
    assert L.guard_stats["checks"] - before["checks"] > 30 * steps and L.guard_stats["buffers_checked"] > before["buffers_checked"]


@pytest.mark.parametrize("workload,kw", [("magicpony", dict(deform=True)), ("fauna", {})])
def test_poisoned_buffers_change_nothing(workload, kw, dev, guard):
    """Guard level 2: every buffer the library is handed starts as NaN / 0x7f7f7f7f instead of whatever the allocator left.  A kernel
    that read a slot nobody wrote (the tail of a speculative capacity, a list entry past its count, a scratch word a skipped launch
    never zeroed) would turn the images NaN, or gather far out of range -- deterministically.  Same images as the unguarded step."""
    pipeline = importlib.import_module("3danimals_amd.pipeline")

    def run(level):
        guard(level)
        torch.manual_seed(7)
        scene = pipeline.SyntheticScene(grid_res=32, batch=5, resolution=(136, 200), device=dev, seed=0, workload=workload, net_width=64, **kw)
        outs = []
        for _ in range(4):
            out = scene.step(backward=True, optimizer_step=False)
            grads = [p.grad.detach().clone() for p in scene.netShape.parameters() if p.grad is not None][:2] + [scene.arti.grad.detach().clone()]
            outs.append((out["shaded"].detach().clone(), out["dino_pred"].detach().clone(), out["loss"].detach().clone(), grads))
        return outs

    a, b = run(2), run(0)
    for (sa, da, la, ga), (sb, db, lb, gb_) in zip(a, b):
        assert bool(torch.isfinite(sa).all()) and bool(torch.isfinite(da).all())
        if workload == "magicpony":  # (Fauna draws its random views from the device generator: the frames differ run to run by construction)
            assert float((sa - sb).abs().max()) < 1e-5 and float((da - db).abs().max()) < 1e-5
            assert abs(float(la) - float(lb)) <= 1e-4 * abs(float(lb))
        for x, y in zip(ga, gb_):
            assert bool(torch.isfinite(x).all())
            if workload == "magicpony":
                np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), rtol=2e-3, atol=2e-4 * float(y.abs().max()))


def test_guard_mode_shape_fuzz(dev, ops, mods, guard):
    """Random shapes through the whole path under guard level 2: B in 1..17, any H / W (odd ones included: the row-major list, the
    general compositor path), grids of several sizes and numberings, surfaces that leave the frustum, nothing on screen (P = 0), an empty
    surface (F = 0), and the capacities that depend on the previous call forced too small (the speculative DMTet emit's buffers, the
    valence counters sized by the last vertex count) -- the exact re-run must take over without a byte out of place."""
    L = importlib.import_module("3danimals_amd._lib")
    M, R, D = mods["mesh"], mods["render"], mods["dmtet"]
    tg = importlib.import_module("3danimals_amd.tetgrid")
    synthetic = importlib.import_module("3danimals_amd.synthetic")
    guard(2)
    rng = np.random.RandomState(5)
    for case in range(28):
        B = int(rng.randint(1, 18))
        H, W = (int(rng.randint(1, 26)) * 8, int(rng.randint(1, 26)) * 8) if case % 3 else (int(rng.randint(5, 150)), int(rng.randint(5, 150)))
        res = int(rng.choice([6, 9, 12, 17, 24]))
        zoom = float(rng.choice([0.0, 0.6, 1.0, 1.0, 2.5, 6.0]))  # 0.0: pushed off screen (P = 0)
        pos, tets = kuhn(res)
        if case % 4 == 1:
            p_np, t_np = tg.scramble(pos.numpy(), tets.numpy(), case)
            pos, tets = torch.from_numpy(p_np), torch.from_numpy(t_np).long()
        pos_d = pos.to(dev)
        grid = D.TetGridTopology(tets.to(dev), positions=pos_d)
        centre, ext = pos.mean(0), float((pos.amax(0) - pos.amin(0)).max())
        mvp, w2c, campos = (t.to(dev) for t in synthetic.random_cameras(B, seed=case))
        for t in range(3):
            radius = (0.30 + 0.05 * t) * ext
            sdf = (radius - (pos - centre).norm(dim=-1) + 0.04 * ext * seeded((pos.shape[0],), 90 + case, -1, 1)).to(dev).requires_grad_(True)
            if case == 9 and t == 1:
                sdf = (-1.0 - 0 * sdf.detach()).requires_grad_(True)  # nothing inside: V = F = 0
            if t == 2 and getattr(grid, "_last_counts", None) is not None:  # the previous extraction's sizes, falsified: capacities too small
                grid._last_counts = tuple(max(1, c // 3) if c > 0 else c for c in grid._last_counts)
                grid._last_surface_vertices = max(1, getattr(grid, "_last_surface_vertices", 3) // 3)
            verts, faces, uv_idx = ops.dmtet(pos_d, sdf, grid)
            if faces.shape[0] == 0:
                assert verts.shape[0] == 0
                continue
            V = verts.shape[0]
            offs = (0.02 * ext * seeded((B, V, 3), 60 + case + t, -1, 1)).to(dev).requires_grad_(True)
            scale = 2.0 * (zoom if zoom > 0 else 1.0) / ext
            posed = (verts[None] - centre.to(dev)) * scale + offs + (torch.tensor([0.0, 80.0, 0.0], device=dev) if zoom == 0 else 0.0)  # (far above every camera's frustum)
            uvs, uvi = torch.zeros(1, 4, 2, device=dev), torch.zeros(1, faces.shape[0], 3, dtype=torch.int64, device=dev)
            prior = M.make_mesh((verts[None] - centre.to(dev)) * scale, faces[None], uvs, uvi, None)
            shape = M.make_mesh(posed, faces[None], uvs.expand(B, -1, -1), uvi, None)
            modes = ["shaded", "geo_normal"] if case % 2 else ["shaded"]
            out = R.render_mesh(None, shape, mvp, w2c, campos, None, None, (H, W), bsdf="diffuse", render_modes=modes, prior_mesh=prior)
            loss = sum((o * seeded(tuple(o.shape), 70 + i, -1, 1).to(dev)).sum() for i, o in enumerate(out))
            g_sdf, g_offs = torch.autograd.grad(loss, [sdf, offs], allow_unused=True)
            for o in out:
                assert bool(torch.isfinite(o).all()), (case, t, B, H, W, res, zoom)
            assert g_sdf is None or bool(torch.isfinite(g_sdf).all()), (case, t)
            assert g_offs is None or bool(torch.isfinite(g_offs).all()), (case, t)
            if zoom == 0:
                assert float(out[0][:, 3].abs().max()) == 0.0  # nothing on screen
    assert L.guard_stats["checks"] > 500


# ------------------------------------------------------------------------------------------------ binned rasteriser (round 5)
@pytest.mark.parametrize("case", ["mesh-b16", "mesh-b3-512x384", "slivers", "eye_plane", "overflow", "spiky-step"])
def test_binned_rasteriser_equals_the_atomic_path_bit_for_bit(case, dev, ops, monkeypatch):
    """a3d_rast_fwd's two forms -- per-tile triangle lists + a fine pass with the depth test in LDS (a3d_rast_opts.bins) against
    triangle-parallel 64-bit atomicMin + resolve -- leave the same texels bit for bit (ids, u, v, z/w) and the same covered-pixel block
    counts: the marching-tets mesh of the bench, long slivers and eye-plane straddlers (boxes of many tiles: the tile stage appends),
    tile lists forced to overflow their capacity (16 entries: every block of the object takes the exact all-triangles route and
    reports it), and the trained-like mesh of a full step.  Both against oracle/raster_ref.c where it finishes in seconds."""
    from oracle import raster_ref

    L = importlib.import_module("3danimals_amd._lib")
    oracle_check = True
    if case in ("mesh-b16", "mesh-b3-512x384", "overflow"):
        B, (H, W) = (16, (256, 256)) if case != "mesh-b3-512x384" else (3, (512, 384))
        _, faces, clip, _ = _scene(B, res=32 if B == 16 else 16)
        tri = faces.int()
    elif case == "spiky-step":
        pipeline = importlib.import_module("3danimals_amd.pipeline")
        scene = pipeline.SyntheticScene(grid_res=64, batch=16, resolution=(256, 256), device=dev, seed=0, workload="magicpony", deform=True, mesh="spiky",
                                        net_width=32, net_layers=3, feat_dim=16, embedder_freq=4)
        scene.step(backward=False)
        clip, tri, (H, W), B = scene.last["points"]["clip"].detach().cpu().contiguous(), scene.last["prior"].t_pos_idx[0].int().cpu(), (256, 256), 16
        oracle_check = False  # (2e5 triangle-image pairs with 9e3-pixel boxes: minutes on the CPU; the atomic path is itself pinned on slivers)
    else:
        H, W, B = 256, 256, 3
        g = torch.Generator().manual_seed(11 + len(case))
        n_big, n_small = 60, 400
        centre = torch.rand(B, n_big, 1, 2, generator=g) * 2.4 - 1.2
        direction = torch.nn.functional.normalize(torch.randn(B, n_big, 1, 2, generator=g), dim=-1)
        normal = torch.stack([-direction[..., 1], direction[..., 0]], -1)
        length = 0.2 + 1.8 * torch.rand(B, n_big, 1, 1, generator=g)
        width = 10 ** (-3.5 + 2.5 * torch.rand(B, n_big, 1, 1, generator=g))
        t = torch.tensor([[-1.0, 0.0], [1.0, -1.0], [0.3, 1.0]]).reshape(1, 1, 3, 2)
        xy = torch.cat([centre + direction * length * t[..., :1] + normal * width * t[..., 1:],
                        (torch.rand(B, n_small, 1, 2, generator=g) * 2 - 1) + 0.03 * torch.randn(B, n_small, 3, 2, generator=g)], 1)
        n = n_big + n_small
        z, w = torch.rand(B, n, 3, 1, generator=g) * 1.6 - 0.8, 0.4 + 2.0 * torch.rand(B, n, 3, 1, generator=g)
        if case == "eye_plane":
            w[:, :6, 0] *= -1.0
        clip = torch.cat([xy * w, z * w, w], -1).reshape(B, 3 * n, 4).contiguous()
        tri = torch.arange(3 * n, dtype=torch.int32).reshape(n, 3)[torch.randperm(n, generator=g)].contiguous()
    clip_d, tri_d = clip.to(dev), tri.to(dev)
    nb = L.lib().a3d_cover_blocks(B, H, W)

    def run(binned):
        monkeypatch.setattr(ops, "RASTER_BINNED", binned)
        ops._rast_bins.clear()
        ops._rast_bin_caps.clear()
        ops._rast_bin_caps[(dev, B, H, W)] = 16 if case == "overflow" else 256
        outs = []
        for _ in range(2):  # twice: the second call finds its scratch re-armed by the first (bins_clean / scratch_is_clean)
            rast = ops.rasterize(clip_d, tri_d, (H, W))
            cover = ops._cover_counts.peek(rast.detach())
            outs.append((rast.cpu(), cover[:nb].cpu(), cover[nb:nb + L.lib().a3d_cover_groups(B, H, W) * 16].cpu()))
        return outs

    a, b = run(True), run(False)
    for (ra, ca, ta), (rb, cb, tb) in zip(a, b):
        assert np.array_equal(ra[..., 3].numpy(), rb[..., 3].numpy())  # triangle ids
        assert np.array_equal(ra.numpy(), rb.numpy())  # u, v, z/w: the same operations in the same order
        assert np.array_equal(ca.numpy(), cb.numpy()) and np.array_equal(ta[::16].numpy(), tb[::16].numpy())  # covered-pixel block counts, group sums
    assert float((a[0][0][..., 3] > 0).float().mean()) > 0.02
    status = a[0][2]
    if case == "overflow":
        assert int(status[1]) > 16 and int(status[2]) > 50  # reported: the largest tile count, the blocks that took the exact route
    elif case != "spiky-step":  # (spikes through one tile: lists of several hundred entries are the point of that case)
        assert int(status[2]) == 0
    if oracle_check:
        ref = raster_ref.rasterize(clip, tri, (H, W))
        assert np.array_equal(a[0][0].numpy(), ref.numpy())


# ------------------------------------------------------------------------------------------------ ponymation stage 2 as configured (round 5)
def test_ponymation_stage2_without_rendering_at_full_size_stagewise(dev):
    """config/train_ponymation_horse_stage2.yaml:16-27 as it is written: 20 sequences x 10 frames, enable_render false -- DMTet, the instance
    deformation, [B,F] skinning (skinning.py:369-439) and the vertex normals make_mesh computes for all 200 meshes, no rasteriser.  Every
    stage re-done by the CPU oracle from the HIP output of the stage before (two sequences = 20 meshes on the CPU)."""
    from oracle import check

    pipeline = importlib.import_module("3danimals_amd.pipeline")
    scene = pipeline.SyntheticScene(grid_res=64, batch=20, num_frames=10, resolution=(256, 256), device=dev, seed=0, workload="ponymation", deform=True,
                                    render=False)
    out = scene.step(backward=True, optimizer_step=False)
    assert scene.frames == 200 and out["posed"].shape[0] == 200 and "rast" not in scene.last
    rep = check.compare_step(scene, out, n_images=20)
    assert rep["faces_equal"] and rep["geometry_only"] and rep["num_faces"] > 8000
    assert rep["max_abs_vert_err"] == 0.0 and rep["max_abs_prior_normal_err"] < 2e-5, rep
    assert rep["max_abs_skin_err"] < 5e-6 and rep["max_abs_deform_add_err"] == 0.0, rep
    assert rep["max_abs_posed_normal_err"] < 2e-5, rep
    assert scene.arti.grad is not None and float(scene.arti.grad.abs().max()) > 0 and bool(torch.isfinite(scene.arti.grad).all())


def test_ponymation_stage2_without_rendering_gradients_vs_float64_oracle(dev):
    """The same step small (2 sequences x 3 frames, grid 16): losses and every leaf / parameter gradient -- through the normals backward,
    the [B,F] skinning backward, the deformation network and DMTet -- against float64 autograd through the oracle."""
    from oracle import step_ref

    pipeline = importlib.import_module("3danimals_amd.pipeline")
    scene = pipeline.SyntheticScene(grid_res=16, batch=2, num_frames=3, resolution=(64, 64), device=dev, seed=5, net_width=32, net_layers=3, feat_dim=16,
                                    embedder_freq=4, workload="ponymation", deform=True, render=False)
    out = scene.step(backward=True, optimizer_step=False, sdf_reg=False)
    ref = step_ref.cpu_step(step_ref.snapshot(scene), backward=True, dtype=torch.float64)
    assert torch.equal(ref["faces"], scene.last["prior"].t_pos_idx[0].cpu())
    # (float32 against the float64 chain: the normals of the coarse grid's sliver fans are conditioned no better than ~1e-4, see compare_step)
    assert float((out["posed"].detach().cpu() - ref["posed"]).abs().max()) < 1e-5 and float((out["normals"].detach().cpu() - ref["normals"]).abs().max()) < 3e-4
    np.testing.assert_allclose(float(out["loss"]), float(ref["loss"]), rtol=5e-5, atol=1e-7)
    for k, v in ref["losses"].items():
        np.testing.assert_allclose(float(out["losses"][k]), float(v), rtol=5e-4, atol=1e-6, err_msg=k)
    pairs = [(k, getattr(scene, k).grad, ref["grads"][k]) for k in ("arti", "feat")]
    for name, mod in (("sdf_mlp", scene.netShape.mlp), ("deform", scene.netDeform)):
        pairs += [(f"{name}.{pn}", p.grad, ref["grads"][f"{name}.{pn}"]) for pn, p in mod.named_parameters()]
    _gradients_close(pairs)


def test_empty_leg_quadrant_on_the_gpu_is_a_python_exception_at_the_next_read_back(dev, mods):
    """estimate_bones on a vertex cloud with an empty leg quadrant (the reference drops into pdb, skinning.py:183).  On the GPU the check
    must cost no synchronisation and must not be a device-side assert (which ends a ROCm process as an anonymous 'HSA hardware exception'):
    with a cached kinematic chain the call returns, and the NEXT read-back the path performs raises with the original message; when the
    chain is rebuilt the call's own read-back raises."""
    L = importlib.import_module("3danimals_amd._lib")
    S = mods["skinning"]
    verts, _ = quadruped_mesh(16, 0.3)
    full = verts[None, None].to(dev)
    kw = dict(n_body_bones=8, n_legs=4, n_leg_bones=3, body_bones_mode="z_minmax_y+")
    bones, chain, aux = S.estimate_bones(full, compute_kinematic_chain=True, **kw)
    L.poll_deferred()  # all four quadrants populated: nothing pending raises
    half = full.clone()
    half[..., 0] = half[..., 0].abs() + 0.05  # every vertex at x > 0: the two -x quadrants are empty
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        out = S.estimate_bones(half, compute_kinematic_chain=False, aux=aux, **kw)  # cached chain: no host synchronisation, no exception yet
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert out.shape == bones.shape and bool(torch.isfinite(out).all())
    with pytest.raises(L.A3DError, match="no vertex in a leg quadrant"):
        L.read_back(torch.zeros(3, dtype=torch.int32, device=dev))  # what the DMTet counts / covered-pixel sums read-back does
    L.poll_deferred()  # consumed
    with pytest.raises(RuntimeError, match="no vertex in a leg quadrant"):
        S.estimate_bones(half, compute_kinematic_chain=True, **kw)
    del L._deferred[:]


# ------------------------------------------------------------------------------------------------ resolve + list + rows in one launch (round 5)
@pytest.mark.parametrize("B,hw,E", [(16, (256, 256), 0), (3, (64, 96), 2), (5, (512, 384), 0), (1, (16, 16), 0)])
def test_resolve_cover_gbuffer_in_one_launch_equals_the_two_launch_path(B, hw, E, dev, ops, mods):
    """rasterize(defer_resolve=True) + covered_gbuffer (a3d_rast_resolve_gbuffer_fwd: the list offsets by decoupled look-back) against
    rasterize + covered_gbuffer (a3d_rast_fwd's resolve, then a3d_cover_gbuffer_fwd): the same texels, list, pixel -> entry map, rows (bit
    for bit) and gradients, over several frames of a moving mesh -- the first frame has no list length to allocate by (stand-alone
    resolve), one frame is made to OUTGROW the capacity the frame before left (exact re-run), frames with nothing on screen in between."""
    H, W = hw
    _, faces, clip0, _ = _scene(B, res=32 if B == 16 else 16)
    tri = faces.to(dev)
    V = clip0.shape[1]
    g = torch.Generator().manual_seed(3)
    v_pos = torch.randn(B, V, 3, generator=g).to(dev)
    v_nrm = torch.nn.functional.normalize(torch.randn(B, V, 3, generator=g), dim=-1).to(dev)
    prior = torch.randn(1, V, 3, generator=g).to(dev)
    extra = torch.randn(B, V, E, generator=g).to(dev) if E else None
    before = dict(ops.resolve_events)
    ops._cover_last_len.clear()
    scales = [1.0, 1.03, 0.4, 1.6, 1e-3 if B > 1 else 1.0, 1.0]  # 0.4 -> 1.6: the list grows ~16x, far beyond + 25 %; 1e-3: (nearly) nothing covered

    def frame(scale, defer):
        clip = clip0.clone()
        clip[..., :2] *= scale
        clip = clip.to(dev).requires_grad_(True)
        vp = v_pos.clone().requires_grad_(True)
        rast = ops.rasterize(clip, tri, (H, W), defer_resolve=defer)
        res = ops.covered_gbuffer(clip, vp, v_nrm, prior, rast, tri, extra=extra)
        gb, pix, inv = res[0], res[-2], res[-1]
        wgt = seeded((gb.shape[0], 12), 5, -1, 1).to(dev)
        loss = (gb * wgt).sum() + (0 if extra is None else res[1].sum())
        grads = torch.autograd.grad(loss, [clip, vp]) if gb.shape[0] else (torch.zeros_like(clip), torch.zeros_like(vp))
        return rast.detach().clone(), gb.detach().clone(), pix.clone(), inv.clone(), (res[1].detach().clone() if extra is not None else None), grads

    for scale in scales:
        a, b_ = frame(scale, True), frame(scale, False)
        assert torch.equal(a[0], b_[0]) and torch.equal(a[2], b_[2]) and torch.equal(a[3], b_[3]), scale  # texels, list, map
        assert torch.equal(a[1], b_[1]) and (extra is None or torch.equal(a[4], b_[4])), scale  # rows
        for x, y in zip(a[5], b_[5]):
            np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), rtol=2e-4, atol=2e-5 * max(float(y.abs().max()), 1e-12))
    ev = {k: ops.resolve_events[k] - before[k] for k in before}
    assert ev["fused"] >= 4 and ev["standalone"] >= 1 and (ev["outgrown"] >= 1 or B < 16), ev  # (small frames sit below the 1024-row minimum)
    # a look-back that ran out of its spin budget (never observed; forced here): the texels do not depend on the look-up, so the frame is
    # recovered by counting from them + the two-launch path's second half, with a warning, and the process stops deferring
    ops._debug_force_lookback_timeout = True
    try:
        with pytest.warns(UserWarning, match="look-back timed out"):
            a = frame(1.0, True)
        assert ops.DEFER_RESOLVE is False
    finally:
        ops._debug_force_lookback_timeout, ops.DEFER_RESOLVE = False, True
    b_ = frame(1.0, False)
    assert torch.equal(a[0], b_[0]) and torch.equal(a[1], b_[1]) and torch.equal(a[2], b_[2]) and torch.equal(a[3], b_[3])


@pytest.mark.parametrize("case", [(1, 3, (40, 72), 16, 1.0), (5, 4, (256, 256), 28, 1.6), (7, 3, (96, 96), -20, 1.0)])
def test_switch_matrix_under_guard_mode(case, dev, ops, mods, monkeypatch, guard):
    """The A/B matrix of every launch-folding switch (all on, all off, two mixed settings) once more with every library buffer between
    canaries and poisoned (guard level 2): the modular entry points behind the switches -- a3d_cover_count / _emit, a3d_gbuffer_fwd,
    a3d_aa_analyze, a3d_aa_fwd / _bwd, a3d_mesh_topology, the streaming DMTet count -- get the same scrutiny as the fused path."""
    L = importlib.import_module("3danimals_amd._lib")
    guard(2)
    before = L.guard_stats["checks"]
    test_every_launch_folding_switch_off_gives_the_same_frames_and_gradients(*case, dev, ops, mods, monkeypatch)
    assert L.guard_stats["checks"] - before > 100
