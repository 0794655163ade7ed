"""CPU tests: ABI surface, host logic of the mirror modules (torch, any device), oracle known-answer tests."""
import ctypes
import importlib
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, golden, kuhn, seeded
from oracle import dmtet_ref, mesh_ref, raster_ref, skinning_ref


# ------------------------------------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "a3d.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(a3d_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    L = importlib.import_module("3danimals_amd._lib")
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    lib = L.lib()  # loads here without a GPU (links libamdhip64 only)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.a3d_version() == L.ABI_VERSION == 404
    assert isinstance(lib.a3d_last_error(), bytes)
    assert lib.a3d_dmtet_scratch_bytes(238688, 196608) >= 4 * (234 + 2 * 192)
    assert lib.a3d_aa_hash_bytes(1000) >= 16 * 6000


def _header_prototypes():
    """name -> (return kind, [parameter kinds]) parsed from include/a3d.h; kinds: ptr, int, int64, float, size_t, char_p."""
    header = open(os.path.join(ROOT, "include", "a3d.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    header = re.sub(r"//[^\n]*", "", header)
    header = re.sub(r"^\s*#[^\n]*", "", header, flags=re.M)

    def kind(decl):
        decl = decl.strip()
        if "*" in decl or re.search(r"\ba3d_stream_t\b", decl):
            return "char_p" if re.match(r"const\s+char\s*\*$", decl) else "ptr"
        base = re.sub(r"\b(const|unsigned)\b", "", decl).split()
        base = base[0] if base else ""
        return {"int": "int", "int32_t": "int", "int64_t": "int64", "float": "float", "size_t": "size_t"}.get(base, "?" + decl)

    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_\s\*]*?)\b(a3d_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", header):
        ret, name, params = m.group(1), m.group(2), m.group(3).strip()
        ret = re.sub(r"\b(A3D_API|extern|\"C\")\b", "", ret).strip()
        plist = [] if params in ("", "void") else [kind(re.sub(r"\b[A-Za-z_][A-Za-z0-9_]*\s*$", "", p.strip()) if not p.strip().endswith("*") else p) for p in params.split(",")]
        protos[name] = (kind(ret), plist)
    return protos


def test_header_prototypes_match_the_ctypes_signatures_in_arity_and_kind():
    """include/a3d.h against _lib.SIGNATURES parameter by parameter: count, and pointer / int / int64 / float / size_t kind.  With ctypes an
    argument inserted on one side only is silent stack garbage; this makes it a CPU-suite failure."""
    L = importlib.import_module("3danimals_amd._lib")
    protos = _header_prototypes()
    assert set(protos) == set(L.SIGNATURES), set(protos) ^ set(L.SIGNATURES)
    ck = {ctypes.c_void_p: "ptr", ctypes.c_int: "int", ctypes.c_int64: "int64", ctypes.c_float: "float", ctypes.c_size_t: "size_t",
          ctypes.c_char_p: "char_p"}
    bad = []
    for name, (res, args) in L.SIGNATURES.items():
        want = (ck[res], [ck[a] for a in args])
        if protos[name] != want:
            bad.append((name, "header", protos[name], "ctypes", want))
    assert not bad, bad
    assert not any(k.startswith("?") for r, ps in protos.values() for k in [r] + ps), "unparsed parameter kind"
    # the check bites: an extra int in one place only, or an int where the header has a pointer, is reported
    res, args = L.SIGNATURES["a3d_rast_fwd"]
    assert protos["a3d_rast_fwd"] != (ck[res], [ck[a] for a in args] + ["int"])
    assert protos["a3d_rast_fwd"][1][0] == "ptr" and protos["a3d_rast_fwd"][1][1] == "int"
    assert sum(len(ps) for _, ps in protos.values()) > 400


def test_product_library_carries_no_phase_stamps():
    """The in-kernel phase stamps (csrc/a3d_common.h: A3D_STAMP, tools/kernel_phases.py) live in a second library built with -DA3D_PROFILE;
    the library the package loads exports none of their setters -- and the instrumented twin, when it has been built, exports one per
    instrumented translation unit on top of the same ABI."""
    L = importlib.import_module("3danimals_amd._lib")
    lib = L.lib()
    tus = ("dmtet", "skin", "raster", "gbuffer", "shade", "normals", "antialias")
    assert not any(hasattr(lib, f"a3d_profile_set_{t}") for t in tus)
    prof = os.path.join(ROOT, "3danimals_amd", "lib", "liba3d_hip_prof.so")
    if os.path.exists(prof) and os.path.getmtime(prof) >= os.path.getmtime(L.LIB_PATH):  # (a twin older than the library is a stale build: says nothing)
        twin = ctypes.CDLL(prof)
        assert all(hasattr(twin, f"a3d_profile_set_{t}") for t in tus)
        assert all(hasattr(twin, name) for name in L.SIGNATURES)


def test_phase_tool_registry_matches_the_stamps_in_the_sources():
    """tools/kernel_phases.py names every kernel it can stamp by (translation unit, kernel id, {slot: label}); the ids and slots are
    whatever A3D_STAMP(id, slot) the sources carry -- an edit on one side only would print phases that do not exist."""
    src = open(os.path.join(ROOT, "tools", "kernel_phases.py")).read()
    table = src[src.index("KERNELS = {"):src.index("MAX_WG")]
    entries = re.findall(r'"(\w+)": \("(\w+)", (\d+), \{(.*?)\}\)', table, flags=re.S)
    assert len(entries) >= 15
    for name, tu, kid, labels in entries:
        code = open(os.path.join(ROOT, "3danimals_amd", "csrc", tu + ".hip")).read()
        assert f"A3D_PROFILE_TU({tu})" in code, tu
        stamped = {int(m) for m in re.findall(r"A3D_STAMP\(%s, (\d)\)" % kid, code)}
        assert 0 in stamped, (name, "no start stamp")
        for slot in re.findall(r"(\d): \"", labels):
            assert int(slot) in stamped, (name, tu, kid, slot)


def test_missing_library_fails_loudly(monkeypatch):
    L = importlib.import_module("3danimals_amd._lib")
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/liba3d_hip.so")
    with pytest.raises(L.A3DError, match="no CPU fallback"):
        L.lib()


def test_ops_refuse_cpu_tensors():
    ops = importlib.import_module("3danimals_amd.ops")
    A3DError = importlib.import_module("3danimals_amd._lib").A3DError
    with pytest.raises(A3DError, match="no CPU fallback"):
        ops.rasterize(torch.rand(1, 3, 4), torch.zeros(1, 3, dtype=torch.int32), (8, 8))
    with pytest.raises(A3DError, match="no CPU fallback"):
        ops.skin(torch.rand(1, 4, 3), torch.rand(1, 2, 2, 3), torch.rand(1, 2, 12), 1.0)


def test_product_never_imports_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "3danimals_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(base, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "/root/reference" in re.sub(r'""".*?"""', "", src, flags=re.S).replace("#", "\n#").split("\n#")[0]:
                    bad.append(f)
    assert not bad, bad


# ------------------------------------------------------------------------------------------------ tet grids
@pytest.mark.parametrize("res", [4, 8, 32])
def test_kuhn_grid_counts_and_topology(res, a3d):
    v, t = a3d.tetgrid.kuhn_grid(res)
    assert v.shape == ((res + 1) ** 3, 3) and t.shape == (6 * res**3, 4)
    edges, t2e = a3d.tetgrid.build_topology(t)
    assert edges.shape[0] == 3 * res * (res + 1) ** 2 + 3 * res**2 * (res + 1) + res**3
    assert np.all(edges[:, 0] < edges[:, 1])
    key = edges[:, 0].astype(np.int64) * v.shape[0] + edges[:, 1]
    assert np.all(np.diff(key) > 0)  # lexicographically sorted, unique
    pairs = t[:, a3d.tetgrid.TET_EDGE_SLOTS]
    assert np.array_equal(edges[t2e], np.stack([pairs.min(-1), pairs.max(-1)], -1))
    p = v[t]
    vol = np.einsum("ti,ti->t", np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]), p[:, 3] - p[:, 0])
    assert np.all(vol < 0) and abs(abs(vol).sum() / 6 - 1.0) < 1e-5  # one orientation, tiles the unit cube


def test_tets_npz_roundtrip_reference_format(tmp_path, a3d):
    v, t = a3d.tetgrid.kuhn_grid(4)
    p = str(tmp_path / "8_tets.npz")
    a3d.tetgrid.save_tets_npz(p, v, t)
    z = np.load(p)
    assert set(z.files) == {"vertices", "indices"}  # reference data/tets/generate_tets.py:47
    v2, t2 = a3d.tetgrid.load_tets_npz(p)
    assert np.array_equal(v, v2) and np.array_equal(t, t2)


def test_grid_topology_torch_equals_numpy(a3d):
    dm = importlib.import_module("3danimals_amd.model.geometry.dmtet")
    _, t = a3d.tetgrid.kuhn_grid(6)
    topo = dm.TetGridTopology(torch.from_numpy(t))
    edges, t2e = a3d.tetgrid.build_topology(t)
    assert np.array_equal(topo.edges32.numpy(), edges) and np.array_equal(topo.tet2edge32.numpy(), t2e)
    uvs_ref, _ = dmtet_ref.uv_atlas(t.shape[0])
    assert torch.equal(topo.uvs(), uvs_ref)


IRREGULAR_DMTET = ["dmtet_bcc10_sphere.npz", "dmtet_bcc10_random.npz", "dmtet_delaunay3k_sphere.npz", "dmtet_delaunay3k_random.npz"]


@pytest.mark.parametrize("name", IRREGULAR_DMTET)
def test_dmtet_oracle_matches_reference_on_irregular_grids(name, a3d):
    """The reference's DMTet run on grids with arbitrary numbering (a scrambled BCC lattice -- Quartet's family -- and a scrambled
    Delaunay tetrahedralisation): the oracle and the product's static edge topology must reproduce faces / uv_idx / vertices bit for
    bit; the grid generators must reproduce the grid the fixture holds."""
    g = golden(name)
    pos, tets = torch.from_numpy(g["pos"]), torch.from_numpy(g["tets"]).long()
    gen = a3d.tetgrid.bcc_grid(10, seed=3) if "bcc10" in name else a3d.tetgrid.delaunay_grid(3000, seed=5)
    assert np.array_equal(gen[1], g["tets"]) and np.array_equal((torch.from_numpy(gen[0]) * 7.0).numpy(), g["pos"])
    assert not np.array_equal(np.sort(g["tets"], 1), g["tets"]) and not np.array_equal(g["tets"][np.lexsort(g["tets"].T[::-1])], g["tets"])
    sdf = torch.from_numpy(g["sdf"]).requires_grad_(True)
    verts, faces, _, uv_idx = dmtet_ref.marching_tets(pos, sdf, tets)
    assert np.array_equal(faces.numpy(), g["faces"]) and np.array_equal(uv_idx.numpy(), g["uv_idx"])
    assert np.array_equal(verts.detach().numpy(), g["verts"])
    (gs,) = torch.autograd.grad((verts * seeded(verts.shape, int(g["grad_wgt_seed"]), -1, 1)).sum(), sdf)
    np.testing.assert_allclose(gs.numpy(), g["grad_sdf"], rtol=1e-5, atol=1e-6)
    # the static topology the kernels stream: sorted unique (min, max) edges == the reference's generate_edges (dmtet.py:283-288)
    dm = importlib.import_module("3danimals_amd.model.geometry.dmtet")
    topo = dm.TetGridTopology(tets)
    e, t2e = a3d.tetgrid.build_topology(g["tets"])
    assert np.array_equal(topo.edges32.numpy(), e) and np.array_equal(topo.tet2edge32.numpy(), t2e)
    pairs = np.sort(g["tets"].astype(np.int64)[:, a3d.tetgrid.TET_EDGE_SLOTS], -1).reshape(-1, 2)
    assert np.array_equal(np.unique(pairs, axis=0), e)


# ------------------------------------------------------------------------------------------------ oracle known answers
def test_dmtet_table_symmetry_and_plane():
    # complementary cases use the same edges with opposite winding (dmtet.py:26-43)
    T, N = dmtet_ref.TRIANGLE_TABLE, dmtet_ref.NUM_TRIANGLES
    for c in range(16):
        assert N[c] == N[15 - c]
        assert set(T[c][T[c] >= 0]) == set(T[15 - c][T[15 - c] >= 0])
    # exact plane SDF: every vertex on the plane, face count closed-form for an axis-aligned cut between grid layers
    pos, tets = kuhn(4, scale=1.0)
    sdf = 0.1 - pos[:, 2]
    verts, faces, _, _ = dmtet_ref.marching_tets(pos, sdf, tets)
    assert torch.allclose(verts[:, 2], torch.full_like(verts[:, 2], 0.1), atol=1e-6)
    assert faces.shape[0] > 0
    area = torch.cross(verts[faces[:, 1]] - verts[faces[:, 0]], verts[faces[:, 2]] - verts[faces[:, 0]], dim=-1).norm(dim=-1).sum() / 2
    assert abs(float(area) - 1.0) < 1e-5  # the plane section of the unit cube


def test_oracle_rasterizer_known_answers():
    H = W = 8
    pos = torch.tensor([[[-1.0, -1.0, 0.5, 1.0], [1.0, -1.0, 0.5, 1.0], [-1.0, 1.0, 0.5, 1.0], [1.0, 1.0, 0.5, 1.0]]])
    tri = torch.tensor([[0, 1, 2], [2, 1, 3]], dtype=torch.int32)
    rast = raster_ref.rasterize(pos, tri, (H, W))
    assert float((rast[..., 3] > 0).float().mean()) == 1.0  # two triangles tile the screen, no gap on the shared diagonal
    ids = rast[0, :, :, 3]
    assert set(np.unique(ids.numpy())) == {1.0, 2.0}
    assert torch.allclose(rast[..., 2], torch.full_like(rast[..., 2], 0.5))
    # barycentrics of pixel (0,0): centre ndc (-0.875,-0.875) inside triangle 0: u weights v0
    u, v = float(rast[0, 0, 0, 0]), float(rast[0, 0, 0, 1])
    assert abs(u - 0.875) < 1e-6 and abs(v - 0.0625) < 1e-6
    # row 0 is y=-1 (OpenGL bottom-up)
    low = torch.tensor([[[-1.0, -1.0, 0.0, 1.0], [1.0, -1.0, 0.0, 1.0], [0.0, -0.5, 0.0, 1.0]]])
    r2 = raster_ref.rasterize(low, torch.tensor([[0, 1, 2]], dtype=torch.int32), (H, W))
    assert float(r2[0, 0, :, 3].sum()) > 0 and float(r2[0, -1, :, 3].sum()) == 0
    # nearer triangle wins; equal depth -> lower id
    two = torch.cat([pos[:, :3], pos[:, :3] * torch.tensor([1, 1, 0.2, 1.0])], 1)
    r3 = raster_ref.rasterize(two, torch.tensor([[0, 1, 2], [3, 4, 5]], dtype=torch.int32), (H, W))
    assert set(np.unique(r3[..., 3].numpy())) == {0.0, 2.0}
    same = torch.cat([pos[:, :3], pos[:, :3]], 1)
    r4 = raster_ref.rasterize(same, torch.tensor([[3, 4, 5], [0, 1, 2]], dtype=torch.int32), (H, W))
    assert set(np.unique(r4[..., 3].numpy())) == {0.0, 1.0}


def test_oracle_antialias_known_answer():
    H = W = 16
    k = 8
    xe = (k + 0.3) / W * 2 - 1
    pos = torch.tensor([[[-3.0, -3.0, 0.0, 1.0], [xe, -3.0, 0.0, 1.0], [xe, 3.0, 0.0, 1.0], [-3.0, 3.0, 0.0, 1.0]]], requires_grad=True)
    tri = torch.tensor([[0, 1, 2], [0, 2, 3]], dtype=torch.int32)
    rast = raster_ref.rasterize(pos, tri, (H, W))
    cover = (rast[..., 3:] > 0).float()
    out = raster_ref.antialias(cover, rast, pos, tri)
    np.testing.assert_allclose(out[0, 2:-2, k, 0].detach().numpy(), 0.3, atol=1e-5)
    np.testing.assert_allclose(out[0, 2:-2, k - 1, 0].detach().numpy(), 1.0, atol=1e-6)
    # moving the edge right by d(ndc) raises the blended coverage by d * W/2 per row
    (g,) = torch.autograd.grad(out[0, 4, k, 0], pos)
    assert abs(float(g[0, 1, 0] + g[0, 2, 0]) - W / 2) < 1e-3
    # interior id changes (the shared diagonal) are not silhouettes: nothing else moves
    assert float((out - cover).abs()[0, :, : k - 1].max()) == 0.0


def test_oracle_edge_opposites_closed_and_open():
    tri = np.array([[0, 1, 2], [0, 2, 3], [0, 3, 1], [1, 3, 2]])  # tetrahedron surface
    opp = raster_ref.edge_opposites(tri)
    assert (opp >= 0).all()
    assert opp[0, 0] == 3  # edge (1,2) of face 0 -> face [1,3,2] -> opposite vertex 3
    assert (raster_ref.edge_opposites(tri[:1]) == -1).all()


# ------------------------------------------------------------------------------------------------ host logic (torch, CPU)
def _chain_equal(a, b):
    return repr(a) == repr(b)


@pytest.mark.parametrize("tag,kw", [("default", dict(attach_legs_to_body=True)), ("noattach", dict(attach_legs_to_body=False)),
                                    ("fixed", dict(attach_legs_to_body=True, legs_to_body_joint_indices=[2, 7, 7, 2])),
                                    ("fauna", dict(attach_legs_to_body=True, bone_y_threshold=0.4))])
def test_estimate_bones_matches_reference_golden(tag, kw):
    sk = importlib.import_module("3danimals_amd.model.geometry.skinning")
    g = golden("bones_quadruped_r16.npz")
    shape = torch.from_numpy(g["verts"])[None, None]
    bones, chain, aux = sk.estimate_bones(shape.clone(), n_body_bones=8, n_legs=4, n_leg_bones=3, body_bones_mode="z_minmax_y+",
                                          compute_kinematic_chain=True, **kw)
    np.testing.assert_allclose(bones.numpy(), g[f"{tag}_bones"], atol=1e-6)
    assert repr(chain) == str(g[f"{tag}_chain"])
    assert np.array_equal(np.array(aux["bones_to_joints"]), g[f"{tag}_bones_to_joints"])
    assert [l["body_bone_idx"] for l in aux["legs"]] == list(g[f"{tag}_leg_body_idx"])
    kw2 = {k: v for k, v in kw.items() if k != "attach_legs_to_body"}
    cached = sk.estimate_bones(shape.clone() + 0.01, n_body_bones=8, n_legs=4, n_leg_bones=3, body_bones_mode="z_minmax_y+",
                               compute_kinematic_chain=False, aux=aux, **kw2)
    np.testing.assert_allclose(cached.numpy(), g[f"{tag}_bones_cached"], atol=1e-6)


def test_estimate_bones_no_legs_and_empty_quadrant():
    sk = importlib.import_module("3danimals_amd.model.geometry.skinning")
    g = golden("bones_quadruped_r16.npz")
    shape = torch.from_numpy(g["verts"])[None, None]
    bones, chain, _ = sk.estimate_bones(shape.clone(), n_body_bones=4, n_legs=4, n_leg_bones=0, body_bones_mode="z_minmax")
    np.testing.assert_allclose(bones.numpy(), g["nolegs_bones"], atol=1e-6)
    assert repr(chain) == str(g["nolegs_chain"])
    half = shape[:, :, shape[0, 0, :, 0] > 0.0]  # no vertex on the -x side
    with pytest.raises(RuntimeError, match="leg quadrant"):
        sk.estimate_bones(half, n_body_bones=8, n_legs=4, n_leg_bones=3, body_bones_mode="z_minmax_y+")


@pytest.mark.parametrize("tag", ["b1f1_t1", "b3f2_t005", "b2f2_inst"])
def test_bone_transforms_and_posed_bones_match_reference(tag):
    """The level-batched chain composition (torch, differentiable) against the reference's link-by-link loop."""
    sk = importlib.import_module("3danimals_amd.model.geometry.skinning")
    g = golden(f"skinning_{tag}.npz")
    chain = eval(str(g["chain"]))
    bones = torch.from_numpy(g["bones"])
    ang = torch.from_numpy(g["angles"]).requires_grad_(True)
    M = sk.bone_transforms_torch(bones, chain, ang)
    ref = skinning_ref.bone_transforms(bones, chain, ang.detach())
    for k in range(20):
        np.testing.assert_allclose(M[:, k].detach().numpy(), ref[k].expand(M.shape[0], 4, 4).numpy(), atol=2e-6)
    B, Fr = ang.shape[:2]
    ends = bones.expand(B, Fr, 20, 2, 3).reshape(B * Fr, 20, 2, 3)
    posed = (torch.einsum("nkij,nkej->nkei", M[:, :, :3, :3], ends) + M[:, :, None, :3, 3]).view(B, Fr, 20, 2, 3)
    np.testing.assert_allclose(posed.detach().numpy(), g["posed_bones"], atol=5e-6)
    # zero angles -> identity transforms
    M0 = sk.bone_transforms_torch(bones, chain, torch.zeros_like(ang))
    np.testing.assert_allclose(M0.numpy(), np.broadcast_to(np.eye(4, dtype=np.float32), M0.shape), atol=1e-6)


def test_euler_and_rigid_helpers_match_reference():
    sk = importlib.import_module("3danimals_amd.model.geometry.skinning")
    g = golden("xfm.npz")
    np.testing.assert_allclose(sk.euler_angles_to_matrix(torch.from_numpy(g["euler"]), "XYZ").numpy(), g["euler_mat"], atol=1e-6)
    with pytest.raises(ValueError):
        sk.euler_angles_to_matrix(torch.zeros(2, 3), "XXY")
    R = sk.euler_angles_to_matrix(torch.from_numpy(g["euler"]), "ZYX")
    m = sk._prepare_transform_mtx(rotation=R, translation=torch.ones(5, 3))
    eye = sk._invert_transform_mtx(m) @ m
    np.testing.assert_allclose(eye.numpy(), np.broadcast_to(np.eye(4, dtype=np.float32), eye.shape), atol=1e-6)


def test_render_helpers_match_reference_golden():
    ru = importlib.import_module("3danimals_amd.model.render.renderutils")
    util = importlib.import_module("3danimals_amd.model.render.util")
    gu = importlib.import_module("3danimals_amd.model.geometry.util")
    g = golden("xfm.npz")
    np.testing.assert_allclose(ru.xfm_points(torch.from_numpy(g["pts"]), torch.from_numpy(g["mvp"])).numpy(), g["clip"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(util.perspective(25 / 180 * np.pi, 1, 0.1, 1000.0).numpy(), g["proj"], rtol=1e-6)
    gs = golden("skinning_b1f1_t1.npz")
    bones, v = torch.from_numpy(gs["bones"]), torch.from_numpy(gs["v_in"])
    d = torch.stack([gu.line_segment_distance(bones[:, :, k, 0], bones[:, :, k, 1], v) for k in range(20)])
    np.testing.assert_allclose(torch.softmax(-d / 1.0, 0).numpy(), gs["weights"], atol=1e-6)


def test_shade_product_matches_reference_golden_on_cpu():
    """render.shade is plain torch around the caller's MLPs -> runs on CPU; pinned against the imported reference."""
    render = importlib.import_module("3danimals_amd.model.render.render")
    from test_oracle_golden import _load_nets

    g = golden("shade.npz")
    tex, dino, lgt = _load_nets(None, g)
    gb = {k: torch.from_numpy(g[f"gb_{k}"]) for k in ("pos", "geo", "nrm", "tng", "tex")}
    feat, w2c, campos = (torch.from_numpy(g[k]) for k in ("feat", "w2c", "campos"))
    modes = ["shaded", "dino_pred", "kd", "normal", "geo_normal", "shading", "depth"]
    with torch.no_grad():
        buf = render.shade(gb["pos"], gb["geo"], gb["nrm"], gb["tng"], gb["tex"], w2c, campos[:, None, None, :], lgt, tex, "diffuse", feat=feat,
                           render_modes=modes, two_sided_shading=True, dino_net=dino)
    for m in modes:
        np.testing.assert_allclose(buf[m].numpy(), g[f"out_{m}"], atol=2e-6, err_msg=m)


def test_mesh_container_api(a3d):
    M = importlib.import_module("3danimals_amd.model.render.mesh")
    v = torch.rand(2, 5, 3)
    f = torch.tensor([[[0, 1, 2], [2, 3, 4]]])
    m = M.Mesh(v, f, v_nrm=torch.rand(2, 5, 3), t_nrm_idx=f, v_tex=torch.rand(2, 6, 2), t_tex_idx=f)
    assert len(m) == 2 and m.v_tng is None
    c = m.clone()
    assert c.v_pos is not m.v_pos and torch.equal(c.v_pos, m.v_pos)
    e = M.compute_edges(f)
    assert e.shape == (6, 2) and bool((e[:, 0] < e[:, 1]).all())
    assert M.compute_edge_to_face_mapping(f).shape == (6, 2)
    lo, hi = M.aabb(M.Mesh(v[0]))
    assert bool((lo <= hi).all())
    with pytest.raises(AssertionError, match="share the same edge connectivity"):
        M.make_mesh(v, f.repeat(2, 1, 1), torch.rand(2, 6, 2), f, None)


def test_nvdiffrast_shim_surface():
    shim = os.path.join(ROOT, "3danimals_amd", "shims")
    sys.path.insert(0, shim)
    try:
        dr = importlib.import_module("nvdiffrast.torch")
        ctx = dr.RasterizeGLContext()
        assert isinstance(dr.RasterizeCudaContext(), type(ctx).__mro__[1])
        for name in ("rasterize", "interpolate", "antialias", "DepthPeeler", "texture"):
            assert hasattr(dr, name)
        with pytest.raises(NotImplementedError):  # (cube maps / mip-mapping: off the path; the 2-D bilinear tap is provided)
            dr.texture(torch.zeros(6, 2, 2, 1), torch.zeros(1, 1, 1, 3), boundary_mode="cube")
        with pytest.raises(RuntimeError, match="num_vertices, 4"):
            dr.rasterize(ctx, torch.rand(1, 3, 3), torch.zeros(1, 3, dtype=torch.int32), [8, 8])
    finally:
        sys.path.remove(shim)


def test_write_obj_format(tmp_path):
    """The exported text matches what the reference's per-element writes produce (obj.py:128-177)."""
    M = importlib.import_module("3danimals_amd.model.render.mesh")
    obj = importlib.import_module("3danimals_amd.model.render.obj")
    v = torch.tensor([[[0.0, 0.5, 1.0], [1.0, 0.0, 0.25], [0.0, 1.0, 0.125]]])
    f = torch.tensor([[[0, 1, 2]]])
    m = M.Mesh(v, f, v_nrm=torch.tensor([[[0.0, 0.0, 1.0]] * 3]), t_nrm_idx=f, v_tex=torch.tensor([[[0.0, 0.0], [1.0, 0.0], [0.0, 0.25]]]), t_tex_idx=f)
    obj.write_obj(str(tmp_path), "m", m, 0, save_material=True)
    text = open(tmp_path / "m.obj").read().splitlines()
    assert text[0] == "mtllib m.mtl" and text[1] == "g default"
    assert text[2] == "v 0.0 0.5 1.0 " and text[5] == "vt 0.0 1.0 " and text[7] == "vt 0.0 0.75 "
    assert text[8] == "vn 0.0 0.0 1.0"
    assert text[-1] == "f  1/1/1 2/2/2 3/3/3" and "usemtl defaultMat" in text


def test_write_obj_matches_the_reference_writer_golden(tmp_path):
    """G10: byte-for-byte the text the REFERENCE's write_obj produced for the same mesh (obj.py:128-177; generated by importing the
    reference in the build container), with and without texture coordinates, second and first batch entry."""
    from types import SimpleNamespace

    obj = importlib.import_module("3danimals_amd.model.render.obj")
    g = golden("write_obj.npz")
    t = lambda k: torch.from_numpy(g[k])
    mesh = SimpleNamespace(v_pos=t("v_pos"), v_nrm=t("v_nrm"), v_tex=t("v_tex"), t_pos_idx=t("t_pos_idx"), t_nrm_idx=t("t_pos_idx"),
                           t_tex_idx=t("t_tex_idx"), material=None)
    obj.write_obj(str(tmp_path), "animal", mesh, 1, save_material=True)
    assert open(tmp_path / "animal.obj").read() == str(g["obj_idx1_material"])
    obj.write_obj(str(tmp_path), "animal2", mesh, 0, save_material=False)
    assert open(tmp_path / "animal2.obj").read() == str(g["obj_idx0_nomaterial"])


def test_save_mtl_without_material(tmp_path):
    obj = importlib.import_module("3danimals_amd.model.render.obj")
    obj.save_mtl(str(tmp_path / "a_b.mtl"), None)
    assert open(tmp_path / "a_b.mtl").read().split() == "newmtl defaultMat Kd 1 1 1 Ks 0 0 0 Ka 0 0 0 Tf 1 1 1 Ni 1 Ns 0".split()


def test_coordmlp_per_image_feature_path_equals_concatenation():
    """CoordMLP.sample(x, feat=[B,C], feat_index=[P]) == the reference formulation with the per-point feature concatenated
    (networks/MLPs.py:84-90), values and gradients (same sum, other association -> fp32 rounding)."""
    import importlib

    hostnets = importlib.import_module("3danimals_amd.hostnets")
    torch.manual_seed(0)
    net = hostnets.CoordMLP(3, 9, 4, nf=32, n_harmonic_functions=4, extra_feat_dim=16, min_max=torch.tensor([[0.0, 1.0]] * 9),
                            activation="sigmoid", symmetrize=True)
    x = torch.rand(500, 3) * 2 - 1
    feat = torch.randn(4, 16, requires_grad=True)
    idx = torch.randint(0, 4, (500,)).sort().values
    a = net.sample(x, feat=feat, feat_index=idx)
    ga = torch.autograd.grad(a.square().sum(), [feat] + list(net.parameters()))
    b = net.sample(x, feat=feat[idx])
    gb = torch.autograd.grad(b.square().sum(), [feat] + list(net.parameters()))
    assert torch.allclose(a, b, atol=1e-6)
    for u, v in zip(ga, gb):
        assert torch.allclose(u, v, atol=1e-5, rtol=1e-4)


def test_oracle_step_handles_every_workload():
    """oracle/step_ref.cpu_step on CPU for the three workloads of pipeline.SyntheticScene (BASELINE configs 3-5): the instance
    deformation with its regularisers, the conditioned SDF + second random-view render of train_fauna, and T-frame sequences with
    the flow loss.  (The GPU tests compare the HIP step with exactly this function.)"""
    import math

    from oracle import step_ref

    a3d = importlib.import_module("3danimals_amd")
    nets = importlib.import_module("3danimals_amd.hostnets")

    def base(n=2):
        return step_ref.synthetic_state(grid_res=8, n=n, resolution=(32, 32), net_width=16, net_layers=3, feat_dim=8, embedder_freq=3)

    st = base()
    st["deform"] = nets.CoordMLP(3, 3, 3, nf=16, n_harmonic_functions=3, embedder_scalar=1.0, extra_feat_dim=8, symmetrize=True)
    r = step_ref.cpu_step(st)
    assert {"arti_reg", "deform_reg"} <= set(r["losses"]) and any(k.startswith("deform.") and v is not None for k, v in r["grads"].items())

    st = base()
    st["workload"] = "fauna"
    st["sdf_mlp"] = nets.CoordMLP_Mod(3, 1, 3, nf=16, n_harmonic_functions=3, embedder_scalar=2 * math.pi / 7 * 0.9)
    st["class_emb"] = 0.1 * torch.randn(128, generator=torch.Generator().manual_seed(0))
    mvp, w2c, campos = a3d.synthetic.random_cameras(2, seed=9)
    st["random_view"] = dict(mvp=mvp, w2c=w2c, campos=campos)
    r = step_ref.cpu_step(st)
    assert "mask_random" in r["losses"] and float(r["grads"]["class_emb"].abs().max()) > 0 and r["mask_random"].shape == (2, 1, 32, 32)

    st = base(n=4)
    st.update(workload="ponymation", nb=2, num_frames=2)
    st["arti"] = st["arti"].view(2, 2, 20, 3)
    st["flow_gt"] = 0.05 * torch.randn(2, 1, 2, 32, 32, generator=torch.Generator().manual_seed(1))
    r = step_ref.cpu_step(st)
    assert r["flow"].shape == (4, 2, 32, 32) and r["losses"]["flow"].shape == (2, 1) and bool(torch.isfinite(r["loss"]))


def test_weight_modulated_field_matches_reference_state_dict_layout():
    """hostnets.CoordMLP_Mod keeps the parameter names of the reference class (MLPs.py:104-247) so Fauna checkpoints load."""
    nets = importlib.import_module("3danimals_amd.hostnets")
    m = nets.CoordMLP_Mod(3, 1, 5, nf=32, n_harmonic_functions=4)
    names = set(m.state_dict())
    assert {"in_layer.weight", "in_layer.bias", "mlp.linear_0.weight", "mlp.linear_4.weight", "style_mlp.network.0.weight",
            "style_mlp.network.2.weight"} <= names
    x = torch.randn(7, 3)
    f = torch.randn(128)
    assert torch.allclose(m(x, feat=f[None]), m(x, feat=f[None].repeat(7, 1)), atol=1e-6)  # only the first style row is read


@pytest.mark.parametrize("grid", ["kuhn12", "kuhn9", "bcc5", "delaunay"])
def test_word_groups_cover_every_vertex_of_their_rows(grid):
    """The static tables of the culled DMTet count (TetGridTopology.word_groups): row w must name the 16-vertex group of EVERY vertex
    that the 64 index rows of word w touch -- a group missing from a row could let the kernel skip a word that holds a crossing -- or
    be marked dense; rows past the list are dense; the table is padded to whole 1024-item blocks."""
    import importlib

    tg = importlib.import_module("3danimals_amd.tetgrid")
    dm = importlib.import_module("3danimals_amd.model.geometry.dmtet")
    if grid.startswith("kuhn"):
        _, tets = tg.kuhn_grid(int(grid[4:]))
    elif grid == "bcc5":
        _, tets = tg.bcc_grid(5, seed=3)
    else:
        _, tets = tg.delaunay_grid(300, seed=1)
    edges, _ = tg.build_topology(tets)
    slots, bits, block = 8, 4, 1024
    some_sparse = False
    for rows in (edges, np.asarray(tets, dtype=np.int32)):
        t = dm._word_groups(torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int32)), slots, bits, block, chunk_words=7).numpy()
        n = rows.shape[0]
        assert t.shape == (-(-n // block) * (block // 64), slots)
        for w in range(t.shape[0]):
            if 64 * w >= n:
                assert (t[w] == -1).all()
                continue
            need = set((rows[64 * w:64 * w + 64] >> bits).reshape(-1).tolist())
            if (t[w] == -1).all():
                assert len(need) > slots
            else:
                assert set(t[w].tolist()) == need
                some_sparse = True
    assert some_sparse or not grid.startswith("kuhn")


def test_bench_credits_ride_along_work_to_the_call_that_carries_it():
    """bench.algorithmic_bytes: a call that carries another pass as extra work-groups ([N..] = vertex normals in the rasteriser's launch,
    [+analysis] = the silhouette analysis in the compositor's) is credited both passes' bytes -- the sum of the two stand-alone figures --
    and the culled DMTet count only what it still moves."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    d = dict(B=16, V=6000, F=12000, H=256, W=256, Nv=274625, Ne=1872064, Nt=1572864, K=20, P=200000)
    ab = bench.algorithmic_bytes
    assert ab("a3d_rast_fwd[N16+1]", d) == ab("a3d_rast_fwd", d) + ab("a3d_normals_fwd_pair[B16+B1]", d)
    assert ab("a3d_composite_aa_fwd[C4+C17][+analysis]", d) == ab("a3d_composite_aa_fwd[C4+C17]", d) + ab("a3d_aa_analyze", d)
    assert ab("a3d_composite_aa_fwd[C4+C17]", d) == ab("a3d_composite_aa_fwd[C4]", d) + ab("a3d_composite_aa_fwd[C17]", d)
    streamed = ab("a3d_dmtet_count", d)
    culled = ab("a3d_dmtet_count", {**d, "dm_words_read": (1130, 29251, 980, 24576)})
    assert streamed > 42e6 and 5e6 < culled < 7e6


def test_bench_credit_table_is_frozen_against_the_survey_formulas():
    """SURVEY.md section 8(d) defines the algorithmic bytes of every stage; bench.algorithmic_bytes must credit exactly those -- inputs read
    once, outputs written once, index data shared over the batch once; NO staging: not the per-face intermediate of the normals backward,
    not the chain products the skinning forward leaves for its backward, not the 32 bytes the memory side makes of a 4-byte atomic.  The
    table is restated here formula by formula, so that a credit cannot move between rounds without this test moving with it (VERDICT r4
    weak 8: a3d_normals_bwd went 19.7 -> 35.7 MB with a kernel rewrite)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod_frozen", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    B, V, F, H, W, Nv, Ne, Nt, K, P = 16, 5928, 11852, 256, 256, 274625, 1872064, 1572864, 20, 199956
    HW = H * W
    nblk = 1100
    d = dict(B=B, V=V, F=F, H=H, W=W, Nv=Nv, Ne=Ne, Nt=Nt, K=K, P=P, skin_v_batch=B, cover_blocks=nblk)
    planes = Ne // 8 + Nt // 2 + Ne // 16
    want = {
        # 8d "DMTet per call", streaming form: sdf + both index arrays in, bit planes out (the culled form: test above)
        "a3d_dmtet_count": 4 * Nv + 8 * Ne + 16 * Nt + planes,
        "a3d_dmtet_emit": planes + 56 * V + 72 * F + 12 * F + (12 * F + 4 * V) + 4 * Nv,
        "a3d_dmtet_bwd": 12 * V + 4 * V + 8 * V + 8 * V + 24 * V + 8 * V,
        # 8d "Skinning fwd per image: 12 V (1/B or 1) + 12 V", + the bones / angles / transforms of the fused chain; bwd: g_out, v, g_v
        "a3d_skin_pose_fwd": 12 * B * V + 12 * B * V + B * K * (12 + 48 + 24),
        "a3d_skin_pose_bwd": 12 * B * V + 12 * B * V + 12 * B * V + 48 * B * K + B * K * (12 + 48 + 12 + 24),
        # 8d "Normals fwd per image: 12 F idx (once per step) + 36 F gathers + 12 V write; bwd same order"
        "a3d_normals_fwd[B16]": 4 * V + 24 * F + B * (36 * F + 24 * V),
        "a3d_normals_bwd[B16]": 4 * V + 24 * F + B * (36 * F + 36 * V),
        # 8d "Rasterise fwd per image: 16 V + 12 F + 16 HW"
        "a3d_rast_fwd": B * (16 * V + 16 * HW) + 12 * F,
        "a3d_cover_gbuffer_fwd": 16 * 256 * nblk + 4 * (B * HW // 256) + 8 * P + 4 * B * HW + 48 * P,
        "a3d_gbuffer_bwd": P * (8 + 16 + 48) + B * V * (36 + 16),
        "a3d_shade_bwd": P * (48 + 8 + 12 + 28 + 48 + 12),
        "a3d_composite_aa_fwd[C4]": 4 * B * HW + 4 * P * 3 + 4 * 4 * B * HW,
        "a3d_composite_aa_fwd[C17]": 4 * B * HW + 4 * P * 16 + 4 * 17 * B * HW,
        "a3d_composite_aa_fwd[C17>16]": 4 * B * HW + 4 * P * 16 + 4 * 16 * B * HW,  # (round 6: the feature image without the alpha channel nobody keeps)
        "a3d_shade_bwd_rows": P * (12 + 48 + 8 + 12 + 48) + 36 * (-(-P // 8192) * 8192),
        "a3d_xfm_points_fwd": B * V * 28 + 64 * B,
        "a3d_xfm_points_bwd": B * V * 40 + 128 * B,
        "a3d_flow_delta_fwd": B * V * 24,
        "a3d_flow_delta_bwd": B * V * 40,
        "a3d_composite_aa_bwd[C4]": 8 * P + 8 * P * 3 + 16 * B * V,
        "a3d_aa_analyze": B * 16 * HW + B * 16 * V,
    }
    for name, bytes_ in want.items():
        assert bench.algorithmic_bytes(name, d) == bytes_, (name, bench.algorithmic_bytes(name, d), bytes_)
    # the figures the round-4 review recomputed by hand (MB at the bench mesh): nothing credited beyond them
    assert 9e6 < want["a3d_normals_bwd[B16]"] < 20e6 and want["a3d_skin_pose_fwd"] < 2.4e6 and want["a3d_skin_pose_bwd"] < 3.5e6


def test_product_library_ignores_the_experiment_knob(monkeypatch):
    """The A3D_EXP measurement knobs (tools/kernel_lab.py) are compiled into the experiment / profile builds only (csrc/build.py --exp /
    --profile): the library the package loads never reads the environment, so no setting of a measurement can change what it computes."""
    src = open(os.path.join(ROOT, "3danimals_amd", "csrc", "common.hip")).read()
    body = src.split("int a3d_exp(void) {")[1].split("\n}")[0]
    assert "#if defined(A3D_EXPERIMENT) || defined(A3D_PROFILE)" in body and body.count("getenv") == 1 and body.index("getenv") < body.index("#else")
    import glob

    for path in glob.glob(os.path.join(ROOT, "3danimals_amd", "csrc", "*.hip")):  # every getenv sits behind the experiment macro
        text = open(path).read()
        for m in re.finditer(r"getenv\(", text):
            before = text[: m.start()]
            assert before.rfind("#if") > max(before.rfind("#endif"), before.rfind("#else")) and "A3D_EXPERIMENT" in before[before.rfind("#if"):], os.path.basename(path)
    import subprocess

    lib = os.path.join(ROOT, "3danimals_amd", "lib", "liba3d_hip.so")
    syms = subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True).stdout
    assert "getenv" not in syms, "the product library must not read the environment"


# ------------------------------------------------------------------------------------------------ DMTet count in any numbering
@pytest.mark.parametrize("grid", ["bcc6s", "delaunay", "kuhn9s", "kuhn9"])
def test_spatial_order_tables_reproduce_the_planes_of_the_file_order(grid):
    """The static tables of a3d_dmtet_count_ordered (TetGridTopology.spatial_order), replayed on the CPU by the kernel's own rule: signs
    looked up by rank, words culled by their groups, crossing / case bits scattered to the row each ranked row came from -- must give
    the crossing flags and marching-tets cases of the file's own `edges` / `tets` order for every SDF, and a culled word must hold
    no crossing.  Also: the tables are permutations, the corner order of a tet row survives the ranking."""
    import importlib

    tg = importlib.import_module("3danimals_amd.tetgrid")
    dm = importlib.import_module("3danimals_amd.model.geometry.dmtet")
    if grid == "bcc6s":
        pos, tets = tg.bcc_grid(6, seed=3)
    elif grid == "delaunay":
        pos, tets = tg.delaunay_grid(400, seed=1)
    else:
        pos, tets = tg.kuhn_grid(9)
        if grid.endswith("s"):
            pos, tets = tg.scramble(pos, tets, 5)
    topo = dm.TetGridTopology(torch.from_numpy(tets), positions=torch.from_numpy(pos))
    struct, t = topo.spatial_order()
    nv, ne, nt = topo.num_verts, topo.edges32.shape[0], topo.tets32.shape[0]
    assert struct.size == ctypes.sizeof(struct) and struct.group_slots == 16 and struct.edge_groups == t["edge_groups"].data_ptr()
    vor, eor, tor = t["vertex_of_rank"].long(), t["edge_of_row"].long(), t["tet_of_row"].long()
    for perm, n in ((vor, nv), (eor, ne), (tor, nt)):
        assert torch.equal(perm.sort().values, torch.arange(n))
    assert torch.equal(vor[t["tets_ranked"].long()], topo.tets32.long()[tor])  # corner order kept
    assert torch.equal(vor[t["edges_ranked"].long()].sort(1).values, topo.edges32.long()[eor])
    # neighbouring ranks are neighbouring in space: the point of the ranking (mean step along the curve << the grid's extent)
    p = torch.from_numpy(pos)[vor]
    assert (p[1:] - p[:-1]).norm(dim=1).mean() < 0.25 * float((p.amax(0) - p.amin(0)).max())
    g = torch.Generator().manual_seed(7)
    centre = torch.from_numpy(pos).mean(0)
    sdfs = [0.3 - (torch.from_numpy(pos) - centre).norm(dim=1), torch.randn(nv, generator=g), -torch.ones(nv), torch.ones(nv)]
    one = -torch.ones(nv)
    one[nv // 3] = 1.0
    sdfs += [one, -one]
    culled_any = False
    for sdf in sdfs:
        inside = sdf > 0
        sign_r = inside[vor]  # the sign plane, in rank order
        pad = (-nv) % 16
        field = torch.cat([sign_r, torch.zeros(pad, dtype=torch.bool)]).reshape(-1, 16)
        state = torch.where(field.all(1), 2, torch.where(field.any(1), 1, 0))
        for rows, of_row, groups, want in (
                (t["edges_ranked"].long(), eor, t["edge_groups"].long(), (inside[topo.edges32.long()[:, 0]] != inside[topo.edges32.long()[:, 1]]).long()),
                (t["tets_ranked"].long(), tor, t["tet_groups"].long(), (inside[topo.tets32.long()] * torch.tensor([1, 2, 4, 8])).sum(1))):
            n = rows.shape[0]
            st = torch.where(groups >= 0, state[groups.clamp(min=0)], torch.ones_like(groups))
            skip = ((st == 0).all(1) | (st == 2).all(1))[: -(-n // 64)]
            read = (~skip).repeat_interleave(64)[:n]
            s = sign_r[rows]
            val = (s[:, 0] != s[:, 1]).long() if rows.shape[1] == 2 else (s * torch.tensor([1, 2, 4, 8])).sum(1)
            if rows.shape[1] == 4:
                val = torch.where(val == 15, 0, val)  # (all four inside: no bit is set, the emit reads it as case 0 -- no triangle either way)
                want = torch.where(want == 15, 0, want)
            got = torch.zeros(n, dtype=torch.long)
            got[of_row[read]] = val[read]
            assert torch.equal(got, want), grid
            culled_any |= bool(skip.any())
    assert culled_any


@pytest.mark.parametrize("struct,cls", [("a3d_dmtet_order", "DmtetOrder"), ("a3d_dmtet_emit_opts", "DmtetEmitOpts"), ("a3d_rast_opts", "RastOpts"), ("a3d_aa_ride", "AaRide"), ("a3d_ca_shade", "CaShade"), ("a3d_ca_buffer", "CaBuffer"), ("a3d_shade_params", "ShadeParams"), ("a3d_gb_aux", "GbAux")])
def test_abi_structs_match_the_header_field_for_field(struct, cls):
    """The option structs of include/a3d.h against their ctypes mirrors: names, order, pointer / int32 / uint32 kind; `size` first."""
    L = importlib.import_module("3danimals_amd._lib")
    header = open(os.path.join(ROOT, "include", "a3d.h")).read()
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), header, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = [(re.sub(r"\s+", " ", d.strip()).rsplit(" ", 1)) for d in body.split(";") if d.strip()]
    kinds = {"uint32_t": ctypes.c_uint32, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64}
    want = [(name.lstrip("*"), ctypes.c_void_p if "*" in decl + name else kinds[decl]) for decl, name in fields]
    assert [(n, t) for n, t in getattr(L, cls)._fields_] == want and want[0] == ("size", ctypes.c_uint32)


def test_oracle_edge_opposites_do_not_depend_on_the_winding():
    """On a closed 2-manifold every edge has exactly one other face, whichever way the two traverse it: flipping the winding of any
    subset of the faces must leave the opposite VERTICES unchanged (as sets per face: the flip permutes a face's own corners)."""
    from oracle import dmtet_ref

    pos, tets = kuhn(6)
    sdf = 0.3 - pos.norm(dim=1)
    _, faces, _, _ = dmtet_ref.marching_tets(pos, sdf, tets)
    tri = faces.numpy()
    opp = raster_ref.edge_opposites(tri)
    assert (opp >= 0).all()
    rng = np.random.default_rng(0)
    flip = rng.random(tri.shape[0]) < 0.5
    tri_f = np.where(flip[:, None], tri[:, [0, 2, 1]], tri)
    opp_f = raster_ref.edge_opposites(tri_f)
    assert (opp_f >= 0).all()
    back = np.where(flip[:, None], opp_f[:, [0, 2, 1]], opp_f)  # corner i of the flipped face = corner (0, 2, 1)[i] of the original
    assert np.array_equal(back, opp)


def test_lazy_rast_db_behaves_like_the_tensor_it_stands_for():
    """The nvdiffrast stand-in returns rast_db as an object that computes the tensor on first use (the reference discards it,
    render.py:24): arithmetic, comparisons, indexing (both ways), iteration and torch functions must all act on the tensor."""
    shim = importlib.import_module("3danimals_amd.shims.nvdiffrast.torch")
    calls = []

    def make():
        calls.append(1)
        return torch.arange(12.0).reshape(3, 4)

    lazy = shim._LazyRastDb(make)
    assert not calls and "not computed" in repr(lazy)
    t = torch.arange(12.0).reshape(3, 4)
    assert torch.equal(lazy * 2, t * 2) and len(calls) == 1
    assert torch.equal(2 * lazy, t * 2) and torch.equal(1 - lazy, 1 - t) and torch.equal(-lazy, -t) and torch.equal(lazy / 2, t / 2)
    assert torch.equal(lazy > 5, t > 5) and torch.equal(lazy == t, torch.ones(3, 4, dtype=torch.bool))
    assert torch.equal(torch.cat([lazy, lazy]), torch.cat([t, t])) and lazy.shape == (3, 4) and len(lazy) == 3
    lazy[0, 0] = 7.0
    assert float(lazy[0, 0]) == 7.0 and [r.shape for r in lazy] == [torch.Size([4])] * 3 and len(calls) == 1
    assert torch.equal(lazy + shim._LazyRastDb(lambda: torch.ones(3, 4)), lazy.materialize() + 1)


def test_shim_texture_bilinear_tap_known_answers():
    """The nvdiffrast stand-in's dr.texture (2-D, no mip-mapping: the off-path callers of the reference that only need a bilinear tap):
    a texel centre returns the texel, the midpoint of four texels their mean, 'clamp' / 'wrap' / 'zero' at the border, 'nearest', gradients
    to texture and uv from autograd, and the unsupported modes refuse loudly."""
    dr = importlib.import_module("3danimals_amd.shims.nvdiffrast.torch")
    tex = torch.arange(2 * 3 * 4 * 2, dtype=torch.float64).reshape(2, 3, 4, 2)  # [B=2, Th=3, Tw=4, C=2]
    centre = lambda i, j: torch.tensor([(j + 0.5) / 4, (i + 0.5) / 3], dtype=torch.float64)
    uv = torch.stack([torch.stack([centre(1, 2), (centre(1, 2) + centre(2, 3)) / 2]), torch.stack([centre(0, 0), torch.tensor([-0.3, 0.5], dtype=torch.float64)])])[:, None]
    out = dr.texture(tex, uv, filter_mode="linear", boundary_mode="clamp")
    assert out.shape == (2, 1, 2, 2)
    assert torch.allclose(out[0, 0, 0], tex[0, 1, 2]) and torch.allclose(out[0, 0, 1], (tex[0, 1, 2] + tex[0, 1, 3] + tex[0, 2, 2] + tex[0, 2, 3]) / 4)
    assert torch.allclose(out[1, 0, 0], tex[1, 0, 0]) and torch.allclose(out[1, 0, 1], tex[1, 1, 0])  # clamped to column 0
    wrapped = dr.texture(tex, uv, filter_mode="linear", boundary_mode="wrap")[1, 0, 1]
    # u = -0.3 -> x = -1.7: texels -2 and -1 (= columns 2 and 3 wrapped) with weights 0.7 / 0.3
    assert torch.allclose(wrapped, 0.7 * tex[1, 1, 2] + 0.3 * tex[1, 1, 3])
    assert torch.allclose(dr.texture(tex, uv, filter_mode="linear", boundary_mode="zero")[1, 0, 1], torch.zeros(2, dtype=torch.float64))
    assert torch.allclose(dr.texture(tex, uv, filter_mode="nearest", boundary_mode="clamp")[0, 0, 1], tex[0, 2, 3])  # (.5 rounds up)
    shared = dr.texture(tex[:1], uv, filter_mode="linear", boundary_mode="clamp")
    assert torch.allclose(shared[1, 0, 0], tex[0, 0, 0])
    t, u = tex.clone().requires_grad_(True), uv.clone().requires_grad_(True)
    dr.texture(t, u, filter_mode="linear", boundary_mode="clamp").sum().backward()
    assert float(t.grad.sum()) == pytest.approx(2 * 2 * 2) and u.grad.abs().sum() > 0
    for kw in (dict(filter_mode="linear-mipmap-linear"), dict(boundary_mode="cube"), dict(uv_da=uv)):
        with pytest.raises(NotImplementedError):
            dr.texture(tex, uv, **kw)


def test_option_structs_of_an_older_header_are_refused_before_anything_is_launched():
    """Every option struct of the ABI starts with its own size; an entry point that is handed a struct SHORTER than the one it was built
    with (a caller compiled against an older header) must refuse with A3D_EINVAL and say so -- before it touches a pointer or the GPU
    (this runs without one: the pointers below are never dereferenced)."""
    L = importlib.import_module("3danimals_amd._lib")
    lib = L.lib()
    fake = 0x1000  # non-NULL, never dereferenced
    short = lambda cls: cls(size=ctypes.sizeof(cls) - 4)
    cases = {
        "a3d_rast_fwd": lambda o: lib.a3d_rast_fwd(fake, 1, fake, 1, 3, 1, 8, 8, fake, fake, 0, ctypes.addressof(o), None),
        "a3d_dmtet_emit": lambda o: lib.a3d_dmtet_emit(fake, fake, fake, fake, 6, 1, fake, 1, 1, 0, fake, fake, fake, fake, ctypes.addressof(o), None),
        "a3d_dmtet_emit_sparse": lambda o: lib.a3d_dmtet_emit_sparse(fake, fake, fake, fake, 6, 1, fake, 1, 1, 0, fake, fake, fake, fake, ctypes.addressof(o), None),
        "a3d_dmtet_count_ordered": lambda o: lib.a3d_dmtet_count_ordered(fake, 4, 6, 1, ctypes.addressof(o), fake, fake, None, 0, None, 0, None),
        "a3d_composite_aa_fwd": lambda o: lib.a3d_composite_aa_fwd(ctypes.addressof(good_buf), None, fake, fake, fake, 4096, 1, 8, 8, ctypes.addressof(o),
                                                                   None, None),
        "a3d_mask_aa_fwd": lambda o: lib.a3d_mask_aa_fwd(fake, 3, None, 0, fake, fake, fake, 4096, 1, 8, 8, ctypes.addressof(o), None),
    }
    good_buf = L.CaBuffer(size=ctypes.sizeof(L.CaBuffer), C=3, vals=fake, out=fake)
    structs = {"a3d_rast_fwd": L.RastOpts, "a3d_dmtet_emit": L.DmtetEmitOpts, "a3d_dmtet_emit_sparse": L.DmtetEmitOpts,
               "a3d_dmtet_count_ordered": L.DmtetOrder, "a3d_composite_aa_fwd": L.AaRide, "a3d_mask_aa_fwd": L.AaRide}
    for name, fn in cases.items():
        o = short(structs[name])
        rc = fn(o)
        assert rc != 0, name
        msg = lib.a3d_last_error().decode()
        assert "size" in msg and "invalid argument" in msg, (name, msg)  # (names the entry point or the helper that checks its struct)
    # the shading struct of the compositor (checked after the riding analysis, which is absent here)
    sh = short(L.CaShade)
    no_vals = L.CaBuffer(size=ctypes.sizeof(L.CaBuffer), C=3, out=fake)
    assert lib.a3d_composite_aa_fwd(ctypes.addressof(no_vals), None, fake, fake, fake, 4096, 1, 8, 8, None, ctypes.addressof(sh), None) != 0
    assert "size" in lib.a3d_last_error().decode()
    short_buf = short(L.CaBuffer)
    assert lib.a3d_composite_aa_fwd(ctypes.addressof(short_buf), None, fake, fake, fake, 4096, 1, 8, 8, None, None, None) != 0
    assert "size" in lib.a3d_last_error().decode()
