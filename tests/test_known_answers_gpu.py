"""Derived known-answer tests for rasterise / interpolate / antialias (forward AND gradient) that do not call oracle/raster_ref.

nvdiffrast is absent and un-pinned ("parity unpinned"), so the oracle for these three operators is this project's
own restatement of the published semantics.  The tests below pin the kernels against closed forms instead -- identities and analytic
values that follow from the operator definitions -- so that an error shared by kernel and oracle cannot hide.
Everything goes through the C ABI (3danimals_amd.ops).  Float64 torch expressions written out in the tests are the closed forms.
"""
import importlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    return importlib.import_module("3danimals_amd.ops")


def _centres(H, W):
    """NDC coordinates of the pixel centres: column j <-> x = (j + 0.5) * 2 / W - 1, row i <-> y = (i + 0.5) * 2 / H - 1."""
    ys = (torch.arange(H, dtype=torch.float64) + 0.5) * 2 / H - 1
    xs = (torch.arange(W, dtype=torch.float64) + 0.5) * 2 / W - 1
    return torch.meshgrid(ys, xs, indexing="ij")


# ------------------------------------------------------------------------------------------------ rasterise
def test_rasterize_perspective_correct_barycentrics_unequal_w(dev, ops):
    """(u, v, 1-u-v) are PERSPECTIVE-correct: sum_i b_i * (x_i - fx * w_i) = 0 and likewise for y at every covered pixel, and
    z/w = sum b_i z_i / sum b_i w_i.  With w = (1, 3, 0.6) they differ from the screen-space (affine) weights by > 0.1."""
    H = W = 64
    pos = torch.tensor([[[-0.8, -0.7, 0.1, 1.0], [2.4, -0.3, 0.9, 3.0], [-0.06, 0.54, -0.12, 0.6]]])  # NDC (-.8,-.7) (.8,-.1) (-.1,.9)
    tri = torch.tensor([[0, 1, 2]], dtype=torch.int32)
    rast = ops.rasterize(pos.to(dev), tri.to(dev), (H, W)).cpu().double()[0]
    cov = rast[..., 3] > 0
    assert 300 < int(cov.sum()) < 1500
    fy, fx = _centres(H, W)
    p = pos[0].double()
    u, v = rast[..., 0], rast[..., 1]
    b = torch.stack([u, v, 1 - u - v], -1)
    rx = (b * (p[:, 0] - fx[..., None] * p[:, 3])).sum(-1)
    ry = (b * (p[:, 1] - fy[..., None] * p[:, 3])).sum(-1)
    assert float(rx[cov].abs().max()) < 2e-6 and float(ry[cov].abs().max()) < 2e-6
    zw = (b * p[:, 2]).sum(-1) / (b * p[:, 3]).sum(-1)
    assert float((zw - rast[..., 2])[cov].abs().max()) < 2e-6
    assert float(b[cov].min()) >= 0 and float(b[cov].max()) <= 1
    # the affine weights of the projected triangle are something else
    s = p[:, :2] / p[:, 3:]
    den = (s[1, 1] - s[2, 1]) * (s[0, 0] - s[2, 0]) + (s[2, 0] - s[1, 0]) * (s[0, 1] - s[2, 1])
    lam0 = ((s[1, 1] - s[2, 1]) * (fx - s[2, 0]) + (s[2, 0] - s[1, 0]) * (fy - s[2, 1])) / den
    assert float((lam0 - u)[cov].abs().max()) > 0.1
    # coverage = the pixel centres inside the projected triangle (none of them within 1e-6 of an edge here)
    lam1 = ((s[2, 1] - s[0, 1]) * (fx - s[2, 0]) + (s[0, 0] - s[2, 0]) * (fy - s[2, 1])) / den
    inside = (lam0 > 0) & (lam1 > 0) & (1 - lam0 - lam1 > 0)
    assert torch.equal(inside, cov)


def test_rasterize_image_orientation_row0_is_ndc_y_minus_one(dev, ops):
    """Row 0 is NDC y = -1 and column 0 is NDC x = -1 (the OpenGL convention nvdiffrast returns; the reference's projection flips y,
    util.py:192, so that images come out upright)."""
    H, W = 32, 48
    tri = torch.tensor([[0, 1, 2], [0, 2, 3]], dtype=torch.int32)
    low = torch.tensor([[[-1.0, -1.0, 0, 1], [1.0, -1.0, 0, 1], [1.0, -0.5, 0, 1], [-1.0, -0.5, 0, 1]]])  # the band y in [-1, -0.5]
    r = ops.rasterize(low.to(dev), tri.to(dev), (H, W)).cpu()[0, ..., 3] > 0
    assert bool(r[: H // 4].all()) and not bool(r[H // 4:].any())
    left = torch.tensor([[[-1.0, -1.0, 0, 1], [-0.5, -1.0, 0, 1], [-0.5, 1.0, 0, 1], [-1.0, 1.0, 0, 1]]])  # the band x in [-1, -0.5]
    r = ops.rasterize(left.to(dev), tri.to(dev), (H, W)).cpu()[0, ..., 3] > 0
    assert bool(r[:, : W // 4].all()) and not bool(r[:, W // 4:].any())


def test_rasterize_depth_test_and_depth_tie_goes_to_lower_id(dev, ops):
    H = W = 16
    quad = lambda z: [[-0.9, -0.9, z, 1.0], [0.9, -0.9, z, 1.0], [0.9, 0.9, z, 1.0], [-0.9, 0.9, z, 1.0]]
    tri = torch.tensor([[0, 1, 2], [0, 2, 3], [4, 5, 6], [4, 6, 7]], dtype=torch.int32)
    for z_first, z_second, winners in ((0.5, -0.5, {3, 4}), (-0.5, 0.5, {1, 2}), (0.25, 0.25, {1, 2})):  # nearer wins; a tie -> lower id
        pos = torch.tensor([quad(z_first) + quad(z_second)])
        r = ops.rasterize(pos.to(dev), tri.to(dev), (H, W)).cpu()[0]
        ids = set(r[..., 3][r[..., 3] > 0].int().tolist())
        assert ids == winners, (z_first, z_second, ids)
        assert torch.allclose(r[..., 2][r[..., 3] > 0], torch.tensor(min(z_first, z_second)))


@pytest.mark.parametrize("axis", ["x", "y"])
def test_rasterize_shared_edge_through_pixel_centres_is_owned_exactly_once(axis, dev, ops):
    """Two quads share an edge that runs exactly through a line of pixel centres: every such pixel belongs to exactly one of the two
    (no hole, no double hit) and the same side owns the whole line; swapping the triangle order does not change the owner."""
    H = W = 32
    k = 12
    e = (k + 0.5) * 2 / W - 1  # exactly representable
    if axis == "x":
        a = [[-1.0, -1.0, 0, 1], [e, -1.0, 0, 1], [e, 1.0, 0, 1], [-1.0, 1.0, 0, 1]]
        b = [[e, -1.0, 0, 1], [1.0, -1.0, 0, 1], [1.0, 1.0, 0, 1], [e, 1.0, 0, 1]]
    else:
        a = [[-1.0, -1.0, 0, 1], [1.0, -1.0, 0, 1], [1.0, e, 0, 1], [-1.0, e, 0, 1]]
        b = [[-1.0, e, 0, 1], [1.0, e, 0, 1], [1.0, 1.0, 0, 1], [-1.0, 1.0, 0, 1]]
    tri = torch.tensor([[0, 1, 2], [0, 2, 3], [4, 5, 6], [4, 6, 7]], dtype=torch.int32)
    owners = []
    for order in ((a, b), (b, a)):
        r = ops.rasterize(torch.tensor([order[0] + order[1]]).to(dev), tri.to(dev), (H, W)).cpu()[0, ..., 3]
        assert bool((r > 0).all())  # watertight
        line = r[:, k] if axis == "x" else r[k, :]
        first_quad = line <= 2
        assert bool(first_quad.all()) or bool((~first_quad).all())
        owners.append(bool(first_quad.all()) == (order[0] is a))  # True: quad `a` owns the line
    assert owners[0] == owners[1]


def test_rasterize_backward_matches_float64_autograd_of_the_definition(dev, ops):
    """d(u, v)/d(clip) against float64 autograd of the DEFINITION u = a0/(a0+a1+a2), a_i = q_j x q_k, q_i = p_i.xy - f * p_i.w
    (Appendix A), evaluated on the pixels the kernel covered -- unequal w, so x, y AND w gradients are exercised."""
    H = W = 48
    pos = torch.tensor([[[-0.7, -0.8, 0.2, 1.0], [1.6, -0.4, 0.5, 2.0], [0.1, 0.63, 0.1, 0.7]]])
    tri = torch.tensor([[0, 1, 2]], dtype=torch.int32)
    p_gpu = pos.clone().to(dev).requires_grad_(True)
    rast = ops.rasterize(p_gpu, tri.to(dev), (H, W))
    g = torch.rand(1, H, W, 4, generator=torch.Generator().manual_seed(3)) - 0.5
    (rast[..., :2] * g[..., :2].to(dev)).sum().backward()
    cov = rast.detach().cpu()[0, ..., 3] > 0
    fy, fx = _centres(H, W)
    p = pos[0].detach().double().clone().requires_grad_(True)
    q = p[:, None, None, :2] - torch.stack([fx, fy], -1)[None] * p[:, None, None, 3:]
    cross = lambda a, b: a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]
    a0, a1, a2 = cross(q[1], q[2]), cross(q[2], q[0]), cross(q[0], q[1])
    s = a0 + a1 + a2
    ((a0 / s * g[0, ..., 0].double() + a1 / s * g[0, ..., 1].double()) * cov).sum().backward()
    got, want = p_gpu.grad.cpu()[0].double(), p.grad
    assert float(want.abs().max()) > 1
    np.testing.assert_allclose(got[:, [0, 1, 3]].numpy(), want[:, [0, 1, 3]].numpy(), rtol=2e-4, atol=2e-4 * float(want.abs().max()))
    assert float(got[:, 2].abs().max()) == 0  # z does not enter the barycentrics


# ------------------------------------------------------------------------------------------------ interpolate
def test_interpolate_reproduces_an_affine_function_and_its_broadcast_gradient(dev, ops):
    """Attribute = an affine function of NDC position -> the interpolated value at every covered pixel is that function of the pixel
    centre (w = 1).  The attribute is [1,V,C], broadcast over B = 3 images with different coverage; its gradient for L = sum(out * g)
    is sum over images and pixels of g * barycentric weight -- written out in float64 from the 2-D geometry."""
    H = W = 40
    B = 3
    xy = torch.tensor([[-0.8, -0.6], [0.9, -0.2], [-0.3, 0.85], [0.7, 0.8]], dtype=torch.float64)
    shift = torch.tensor([[0.0, 0.0], [0.05, -0.1], [-0.12, 0.07]], dtype=torch.float64)
    pos = torch.zeros(B, 4, 4, dtype=torch.float64)
    pos[..., :2] = xy[None] + shift[:, None]
    pos[..., 3] = 1
    tri = torch.tensor([[0, 1, 2], [1, 3, 2]], dtype=torch.int32)
    A = torch.tensor([[2.0, -1.0, 0.5], [0.3, 0.7, -1.1]], dtype=torch.float64)  # attribute c = A[c] . (x, y, 1) of the UNSHIFTED vertex
    attr = (torch.cat([xy, torch.ones(4, 1, dtype=torch.float64)], -1) @ A.T)[None].float()  # [1,4,2]
    rast = ops.rasterize(pos.float().to(dev), tri.to(dev), (H, W))
    a_gpu = attr.clone().to(dev).requires_grad_(True)
    out = ops.interpolate(a_gpu, rast, tri.to(dev))
    g = torch.rand(B, H, W, 2, generator=torch.Generator().manual_seed(5)) - 0.5
    (out * g.to(dev)).sum().backward()
    fy, fx = _centres(H, W)
    ids = rast.detach().cpu()[..., 3].long()
    want_grad = torch.zeros(4, 2, dtype=torch.float64)
    for b in range(B):
        cov = ids[b] > 0
        # forward: the affine function at the pixel centre, in the vertex frame of image b (vertices moved by shift[b])
        expect = torch.stack([A[c, 0] * (fx - shift[b, 0]) + A[c, 1] * (fy - shift[b, 1]) + A[c, 2] for c in range(2)], -1)
        assert float((out.detach().cpu()[b].double() - expect)[cov].abs().max()) < 2e-5
        assert float(out.detach().cpu()[b][~cov].abs().max()) == 0
        for t in range(2):
            m = ids[b] == t + 1
            v = pos[b, tri[t].long(), :2]
            den = (v[1, 1] - v[2, 1]) * (v[0, 0] - v[2, 0]) + (v[2, 0] - v[1, 0]) * (v[0, 1] - v[2, 1])
            l0 = ((v[1, 1] - v[2, 1]) * (fx - v[2, 0]) + (v[2, 0] - v[1, 0]) * (fy - v[2, 1])) / den
            l1 = ((v[2, 1] - v[0, 1]) * (fx - v[2, 0]) + (v[0, 0] - v[2, 0]) * (fy - v[2, 1])) / den
            for lam, vi in ((l0, tri[t, 0]), (l1, tri[t, 1]), (1 - l0 - l1, tri[t, 2])):
                want_grad[vi] += ((lam * m)[..., None] * g[b].double()).sum((0, 1))
    np.testing.assert_allclose(a_gpu.grad.cpu()[0].double().numpy(), want_grad.numpy(), rtol=1e-4, atol=1e-4)
    # barycentrics sum to one: the total gradient of a constant-1 upstream is the number of covered pixels
    a2 = attr.clone().to(dev).requires_grad_(True)
    ops.interpolate(a2, rast, tri.to(dev)).sum().backward()
    total = 2 * int((ids > 0).sum())  # C = 2 channels
    assert abs(float(a2.grad.double().sum()) - total) < 2e-4 * total


# ------------------------------------------------------------------------------------------------ antialias
def _edge_scene(axis, k, frac, H, W, fold=None):
    """A surface covering everything on the low side of coordinate (k + frac) pixels along ``axis`` (two triangles, vertices far
    outside the view).  ``fold`` adds a third triangle on the silhouette edge: 'back' = folded back under the surface (deeper),
    'forward' = continuing the surface past the edge."""
    e = (k + frac) / (W if axis == "x" else H) * 2 - 1
    verts = [[-3.0, -3.0, 0.0, 1.0], [e, -3.0, 0.0, 1.0], [e, 3.0, 0.0, 1.0], [-3.0, 3.0, 0.0, 1.0]]
    tris = [[0, 1, 2], [0, 2, 3]]
    if fold == "back":
        verts.append([-2.0, 0.0, 0.5, 1.0])  # behind the surface, same side of the edge
        tris.append([1, 4, 2])
    elif fold == "forward":
        verts.append([3.0, 0.0, 0.0, 1.0])  # the surface goes on: the edge is interior
        tris.append([1, 4, 2])
    pos = torch.tensor([verts])
    if axis == "y":
        pos = pos[..., [1, 0, 2, 3]].contiguous()
    return pos, torch.tensor(tris, dtype=torch.int32)


@pytest.mark.parametrize("axis,frac", [("x", 0.3), ("y", 0.3), ("x", 0.8), ("y", 0.65)])
def test_antialias_straight_silhouette_known_answer(axis, frac, dev, ops):
    """Silhouette at k + frac pixels.  frac < 0.5: pixel k (uncovered) takes ``frac`` of its covered neighbour k-1; frac > 0.5: pixel k
    IS covered and gives 1 - frac of itself away to the background colour of pixel k+1 -- i.e. it keeps ``frac`` coverage."""
    H = W = 16
    k = 8
    pos, tri = _edge_scene(axis, k, frac, H, W)
    rast = ops.rasterize(pos.to(dev), tri.to(dev), (H, W))
    cover = (rast[..., 3:] > 0).float()
    out = ops.antialias(cover.contiguous(), rast, pos.to(dev), tri.to(dev)).cpu()[0, ..., 0]
    line = (lambda j: out[2:-2, j]) if axis == "x" else (lambda j: out[j, 2:-2])
    np.testing.assert_allclose(line(k).numpy(), frac, atol=1e-5)
    np.testing.assert_allclose(line(k - 1).numpy(), 1.0, atol=1e-6)
    np.testing.assert_allclose(line(k + 1).numpy(), 0.0, atol=1e-6)


def test_antialias_fold_is_a_silhouette_and_an_interior_edge_is_not(dev, ops):
    """The silhouette test of a NON-boundary edge: with a second triangle folded back under the surface the edge still antialiases
    exactly like a boundary edge; with the second triangle continuing the surface it is interior and nothing is blended."""
    H = W = 16
    k = 8
    pos, tri = _edge_scene("x", k, 0.3, H, W, fold="back")
    rast = ops.rasterize(pos.to(dev), tri.to(dev), (H, W))
    assert set(rast[0, :, :k, 3].int().unique().tolist()) <= {1, 2}  # the folded triangle is hidden behind the surface
    cover = (rast[..., 3:] > 0).float()
    out = ops.antialias(cover.contiguous(), rast, pos.to(dev), tri.to(dev)).cpu()[0, ..., 0]
    np.testing.assert_allclose(out[2:-2, k].numpy(), 0.3, atol=1e-5)
    pos, tri = _edge_scene("x", k, 0.3, H, W, fold="forward")
    rast = ops.rasterize(pos.to(dev), tri.to(dev), (H, W))
    colour = torch.where(rast[..., 3:] > 2.5, 0.25, 1.0) * (rast[..., 3:] > 0)  # the continuation has another colour
    out = ops.antialias(colour.contiguous(), rast, pos.to(dev), tri.to(dev)).cpu()[0, ..., 0]
    np.testing.assert_allclose(out[4:-4, k - 1].numpy(), 1.0, atol=1e-6)  # id discontinuity, but no silhouette: untouched
    np.testing.assert_allclose(out[4:-4, k].numpy(), 0.25, atol=1e-6)


def test_antialias_gradients_closed_form_vertical_edge(dev, ops):
    """out[row, k] = frac * c_in + (1 - frac) * c_bg with frac = W/2 * (x_edge(row) + 1) - k, x_edge linear between the two edge
    vertices.  Hence d sum_rows(out[:, k]) / d x_v = W/2 * (c_in - c_bg) * sum_rows t_v(row) (t_v = the vertex's interpolation weight
    along the edge), d/d colour of the covered neighbour = frac, d/d colour of pixel k itself = 1 - frac."""
    H = W = 16
    k, frac = 8, 0.3
    pos, tri = _edge_scene("x", k, frac, H, W)
    p = pos.clone().to(dev).requires_grad_(True)
    rast = ops.rasterize(p, tri.to(dev), (H, W)).detach()
    c_in, c_bg = 0.9, 0.2
    colour = (torch.where(rast[..., 3:] > 0, c_in, c_bg)).contiguous().clone().requires_grad_(True)
    out = ops.antialias(colour, rast, p, tri.to(dev))
    out[0, :, k, 0].sum().backward()
    gc = colour.grad.cpu()[0, ..., 0]
    np.testing.assert_allclose(gc[:, k - 1].numpy(), frac, atol=1e-5)
    np.testing.assert_allclose(gc[:, k].numpy(), 1 - frac, atol=1e-5)
    assert float(gc[:, :k - 1].abs().max()) == 0 and float(gc[:, k + 1:].abs().max()) == 0
    ys = (torch.arange(H, dtype=torch.float64) + 0.5) * 2 / H - 1
    t2 = (ys + 3) / 6  # weight of vertex 2 (y = +3) at each row; vertex 1 (y = -3) has 1 - t2
    want1, want2 = W / 2 * (c_in - c_bg) * float((1 - t2).sum()), W / 2 * (c_in - c_bg) * float(t2.sum())
    g = p.grad.cpu()[0]
    np.testing.assert_allclose([float(g[1, 0]), float(g[2, 0])], [want1, want2], rtol=1e-4)
    assert float(g[0].abs().max()) == 0 and float(g[3].abs().max()) == 0  # the far vertices do not move the silhouette
    # moving a vertex ALONG the edge direction does not change a vertical edge's crossing
    assert abs(float(g[1, 1])) < 1e-4 * want1 and abs(float(g[2, 1])) < 1e-4 * want2


@pytest.mark.parametrize("axis,frac", [("x", 0.3), ("y", 0.65)])
def test_composite_antialias_known_answer_and_closed_form_gradients(axis, frac, dev, ops):
    """The fused compositor on the straight-silhouette scene, from the definition alone.  Rows [c_in] at the covered pixels over a
    background [c_bg, 0]: the crossing pixel k ends up with coverage ``frac`` in the alpha channel and frac*c_in + (1-frac)*c_bg in the
    colour channel (whichever side of the pixel centre the edge lies on); d out[.., k] / d (row of the covered neighbour) = frac for
    frac < 0.5 -- where pixel k is background -- and d out[.., k] / d (its own row) = frac for frac > 0.5; the colour channel's vertex
    gradient is W/2 (or H/2) * (c_in - c_bg) * the vertex's interpolation weight summed over the rows (columns)."""
    H = W = 16
    k = 8
    c_in, c_bg = 0.9, 0.2
    pos, tri = _edge_scene(axis, k, frac, H, W)
    p = pos.clone().to(dev).requires_grad_(True)
    rast = ops.rasterize(p.detach(), tri.to(dev), (H, W))
    pix, inv = ops.covered_pixels(rast, return_inverse=True)
    vals = torch.full((pix.shape[0], 1), c_in, device=dev).requires_grad_(True)
    bg = torch.zeros(1, H, W, 2, device=dev)
    bg[..., 0] = c_bg
    analysis = ops.AAAnalysis(rast, p.detach(), ops.aa_topology(ops.tri_int32(tri.to(dev)), p.shape[1]))
    out = ops.composite_antialias(vals, pix, inv, bg, p, analysis)  # [1,H,W,2]
    o = out.detach().cpu()[0]
    line = (lambda t, j: t[2:-2, j]) if axis == "x" else (lambda t, j: t[j, 2:-2])
    np.testing.assert_allclose(line(o[..., 1], k).numpy(), frac, atol=1e-5)  # coverage
    np.testing.assert_allclose(line(o[..., 0], k).numpy(), frac * c_in + (1 - frac) * c_bg, atol=1e-5)
    np.testing.assert_allclose(line(o[..., 1], k - 1).numpy(), 1.0, atol=1e-6)
    np.testing.assert_allclose(line(o[..., 0], k + 1).numpy(), c_bg, atol=1e-6)
    # gradients of the colour channel summed along the crossing line
    sel = out[0, :, k, 0] if axis == "x" else out[0, k, :, 0]
    gv, gp = torch.autograd.grad(sel.sum(), (vals, p))
    dense = torch.zeros(H * W, device=dev).index_copy(0, pix, gv[:, 0]).view(H, W).cpu()
    src = k - 1 if frac < 0.5 else k  # the covered pixel whose row feeds pixel k's colour
    np.testing.assert_allclose(line(dense, src).numpy(), frac, atol=1e-5)
    zeroed = dense.clone()
    if axis == "x":
        zeroed[:, src] = 0
    else:
        zeroed[src, :] = 0
    assert float(zeroed.abs().max()) == 0
    n = H if axis == "x" else W
    ys = (torch.arange(n, dtype=torch.float64) + 0.5) * 2 / n - 1
    t2 = (ys + 3) / 6  # weight of edge vertex 2 (at +3 along the edge) at each row / column; vertex 1 (at -3) has 1 - t2
    half = (W if axis == "x" else H) / 2
    want1, want2 = half * (c_in - c_bg) * float((1 - t2).sum()), half * (c_in - c_bg) * float(t2.sum())
    g = gp.cpu()[0]
    comp = 0 if axis == "x" else 1
    np.testing.assert_allclose(sorted([abs(float(g[1, comp])), abs(float(g[2, comp]))]), sorted([want1, want2]), rtol=1e-4)
    assert float(g[0].abs().max()) == 0 and float(g[3].abs().max()) == 0  # the far vertices do not move the silhouette


# ------------------------------------------------------------------------------------------------ depth peeling, pixel differentials
def test_depth_peeling_layers_known_answer(dev, ops):
    """Three stacked quads: layer 0 sees the nearest, layer 1 the middle one, layer 2 the farthest, layer 3 nothing; pixels the
    previous layer left empty stay empty; coincident duplicates come out one per layer in id order."""
    H = W = 16
    quad = lambda z, r=0.9: [[-r, -r, z, 1.0], [r, -r, z, 1.0], [r, r, z, 1.0], [-r, r, z, 1.0]]
    tri = torch.tensor([[0, 1, 2], [0, 2, 3], [4, 5, 6], [4, 6, 7], [8, 9, 10], [8, 10, 11]], dtype=torch.int32).to(dev)
    pos = torch.tensor([quad(0.2) + quad(-0.6, 0.5) + quad(0.7)]).to(dev)  # the nearest quad (ids 3, 4) is the small one
    layers, prev = [], None
    for _ in range(4):
        prev = ops.rasterize(pos, tri, (H, W), prev=prev)
        layers.append(prev.cpu()[0])
    ids = [set(l[..., 3].int().unique().tolist()) for l in layers]
    assert ids[0] == {0, 1, 2, 3, 4} and ids[3] == {0}  # (0: the border pixels outside the 0.9 quads)
    inner = layers[0][..., 3] >= 3  # where the small near quad is
    ring = (layers[0][..., 3] >= 1) & ~inner
    empty = layers[0][..., 3] == 0
    assert set(layers[1][..., 3][inner].int().tolist()) == {1, 2} and set(layers[2][..., 3][inner].int().tolist()) == {5, 6}
    assert set(layers[1][..., 3][ring].int().tolist()) == {5, 6} and float(layers[2][..., 3][ring].abs().max()) == 0
    assert all(float(l[empty].abs().max()) == 0 for l in layers)  # what layer 0 left empty stays empty
    assert torch.allclose(layers[1][..., 2][inner], torch.tensor(0.2)) and torch.allclose(layers[2][..., 2][inner], torch.tensor(0.7))
    dup = torch.tensor([quad(0.3) + quad(0.3) + quad(0.3)]).to(dev)  # three coincident surfaces
    prev, seen = None, []
    for _ in range(3):
        prev = ops.rasterize(dup, tri, (H, W), prev=prev)
        seen.append(set(prev[..., 3].int().unique().tolist()))
    assert [x - {0} for x in seen] == [{1, 2}, {3, 4}, {5, 6}]


def test_rast_db_equals_finite_differences_for_an_affine_triangle(dev, ops):
    """w = 1: the barycentrics are affine in the pixel coordinates, so rast_db (du/dX, du/dY, dv/dX, dv/dY) must equal the difference
    of (u, v) between neighbouring pixels of the same triangle; with unequal w it must match float64 autograd of the definition."""
    H = W = 32
    pos = torch.tensor([[[-0.8, -0.7, 0.1, 1.0], [0.9, -0.4, 0.3, 1.0], [-0.1, 0.85, 0.2, 1.0]]]).to(dev)
    tri = torch.tensor([[0, 1, 2]], dtype=torch.int32).to(dev)
    rast = ops.rasterize(pos, tri, (H, W))
    db = ops.rasterize_db(pos, tri, rast).cpu()[0]
    r = rast.cpu()[0]
    cov = r[..., 3] > 0
    both_x = cov[:, :-1] & cov[:, 1:]
    both_y = cov[:-1, :] & cov[1:, :]
    np.testing.assert_allclose(db[:, :-1, 0][both_x].numpy(), (r[:, 1:, 0] - r[:, :-1, 0])[both_x].numpy(), atol=2e-6)
    np.testing.assert_allclose(db[:, :-1, 2][both_x].numpy(), (r[:, 1:, 1] - r[:, :-1, 1])[both_x].numpy(), atol=2e-6)
    np.testing.assert_allclose(db[:-1, :, 1][both_y].numpy(), (r[1:, :, 0] - r[:-1, :, 0])[both_y].numpy(), atol=2e-6)
    np.testing.assert_allclose(db[:-1, :, 3][both_y].numpy(), (r[1:, :, 1] - r[:-1, :, 1])[both_y].numpy(), atol=2e-6)
    assert float(db[~cov].abs().max()) == 0
    # perspective case: autograd of u(fx, fy) in float64
    posp = torch.tensor([[[-0.7, -0.8, 0.2, 1.0], [1.6, -0.4, 0.5, 2.0], [0.1, 0.63, 0.1, 0.7]]])
    rast = ops.rasterize(posp.to(dev), tri, (H, W))
    db = ops.rasterize_db(posp.to(dev), tri, rast).cpu()[0].double()
    cov = rast.cpu()[0, ..., 3] > 0
    fy, fx = _centres(H, W)
    fx, fy = fx.clone().requires_grad_(True), fy.clone().requires_grad_(True)
    p = posp[0].double()
    q = p[:, None, None, :2] - torch.stack([fx, fy], -1)[None] * p[:, None, None, 3:]
    cross = lambda a, b: a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]
    a0, a1, a2 = cross(q[1], q[2]), cross(q[2], q[0]), cross(q[0], q[1])
    s = a0 + a1 + a2
    for comp, (num, cx, cy) in enumerate(((a0, 0, 1), (a1, 2, 3))):
        gx, gy = torch.autograd.grad((num / s).sum(), [fx, fy], retain_graph=True)
        np.testing.assert_allclose(db[..., cx][cov].numpy(), (gx * 2 / W)[cov].numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(db[..., cy][cov].numpy(), (gy * 2 / H)[cov].numpy(), rtol=1e-4, atol=1e-6)


# ------------------------------------------------------------------------------------------------ antialias on oblique geometry
def _halfplane_scene(theta_deg, cx, cy, H, W, transpose):
    """Everything on the low-x side of the line through pixel-space point (cx, cy) at ``theta_deg`` from the x axis is covered (two
    triangles; the edge's end points and the far corners lie well outside the view).  ``transpose`` swaps x and y."""
    t = np.deg2rad(theta_deg)
    d = np.array([np.cos(t), np.sin(t)])
    L = 4.0 * max(H, W)
    a, b = np.array([cx, cy]) - L * d, np.array([cx, cy]) + L * d  # edge end points (pixels)
    far = np.array([-L, 0.0])
    px = np.stack([a, b, b + far, a + far])  # counter-clockwise or clockwise: the rasteriser takes both windings
    ndc = px / np.array([W / 2.0, H / 2.0]) - 1.0
    pos = torch.tensor(np.concatenate([ndc, np.zeros((4, 1)), np.ones((4, 1))], 1), dtype=torch.float32)[None]
    if transpose:
        pos = pos[..., [1, 0, 2, 3]].contiguous()
    return pos, torch.tensor([[0, 1, 2], [0, 2, 3]], dtype=torch.int32)


@pytest.mark.parametrize("theta,transpose", [(60.0, False), (60.0, True), (75.0, False), (50.0, True), (120.0, False)])
def test_antialias_oblique_silhouette_matches_the_per_row_crossing_distance(theta, transpose, dev, ops):
    """A straight silhouette at 60 / 75 / 50 / 120 degrees to the pixel rows (transposed: 30 / 40 degrees).  nvdiffrast's rule, from the
    published algorithm: for each pixel PAIR straddling the silhouette along the axis the edge is steeper against, the edge crosses the
    line joining the two pixel centres at distance dc from the covered pixel's centre; the covered pixel keeps min(1, 0.5 + dc), the
    uncovered one receives max(0, dc - 0.5).  Pairs along the OTHER axis see an edge shallower than 45 degrees and are skipped.  Closed
    form per row: x_c(row) = cx + (row + 0.5 - cy) / tan(theta); everything else stays 1 / 0.  No oracle involved."""
    H = W = 32
    cx, cy = 15.37, 16.21
    pos, tri = _halfplane_scene(theta, cx, cy, H, W, transpose)
    rast = ops.rasterize(pos.to(dev), tri.to(dev), (H, W))
    cover = (rast[..., 3:] > 0).float()
    out = ops.antialias(cover.contiguous(), rast, pos.to(dev), tri.to(dev)).cpu()[0, ..., 0].double()
    if transpose:
        out = out.t()
    rows = torch.arange(H, dtype=torch.float64) + 0.5
    xc = cx + (rows - cy) / np.tan(np.deg2rad(theta))
    want = torch.zeros(H, W, dtype=torch.float64)
    checked = 0
    for r in range(H):
        x = float(xc[r])
        if not (2.0 < x < W - 2.0):
            want[r] = out[r]  # the silhouette leaves the view on this row: not checked
            continue
        k = int(np.floor(x - 0.5))
        dc = x - (k + 0.5)
        want[r, :k] = 1.0
        want[r, k] = min(1.0, 0.5 + dc)
        want[r, k + 1] = max(0.0, dc - 0.5)
        checked += 1
    assert checked >= 12
    np.testing.assert_allclose(out.numpy(), want.numpy(), atol=2e-5)
    # and each checked row's total coverage is the covered length of its centre line
    sel = (xc > 2.0) & (xc < W - 2.0)
    np.testing.assert_allclose(out.sum(1)[sel].numpy(), xc[sel].numpy(), atol=5e-5)


def test_antialias_coverage_integral_equals_projected_area_and_its_vertex_gradient(dev, ops):
    """Sum of the antialiased alpha of one random OBLIQUE triangle over an empty background against the triangle's projected area in
    pixels, and d(sum)/d(vertex) against d(area)/d(vertex) (closed form of the shoelace formula) -- the property the silhouette
    gradients of the mask / flow losses rely on (render.py:264-267 is their only source, AnimalModel.py:265-269), checked on 40
    arbitrary orientations without any restatement of the operator.  Bounds: the per-pair rule integrates each row (column) of a steep
    (shallow) edge exactly; what is left are the O(1) pixels at the three corners and at steep/shallow hand-overs: |sum - area| <= 1.5
    px^2 (plain coverage without antialiasing: up to 10 px^2 on these triangles), gradient within 10 % in norm (median within 4 %)."""
    H = W = 64
    g = torch.Generator().manual_seed(0)
    tri = torch.tensor([[0, 1, 2]], dtype=torch.int32).to(dev)
    errs, raw_errs, grad_rel = [], [], []
    for _ in range(40):
        while True:
            p = torch.rand(3, 2, generator=g) * 1.6 - 0.8
            pd = p.double().clone().requires_grad_(True)
            px = (pd + 1) * torch.tensor([W / 2, H / 2], dtype=torch.float64)
            area = 0.5 * ((px[1, 0] - px[0, 0]) * (px[2, 1] - px[0, 1]) - (px[2, 0] - px[0, 0]) * (px[1, 1] - px[0, 1]))
            if abs(float(area.detach())) > 150:
                break
        area.abs().backward()
        pos = torch.cat([p, torch.zeros(3, 1), torch.ones(3, 1)], -1)[None].to(dev).requires_grad_(True)
        rast = ops.rasterize(pos, tri, (H, W)).detach()
        cover = (rast[..., 3:] > 0).float()
        total = ops.antialias(cover.contiguous(), rast, pos, tri).sum()
        total.backward()
        gs = pos.grad[0, :, :2].cpu().double()
        errs.append(abs(float(total.detach()) - abs(float(area.detach()))))
        raw_errs.append(abs(float(cover.sum()) - abs(float(area.detach()))))
        grad_rel.append(float((gs - pd.grad).norm() / pd.grad.norm()))
        assert float(pos.grad[0, :, 2].abs().max()) == 0.0  # nothing to z (w receives the x/w, y/w share)
    assert max(errs) <= 1.5 and float(np.mean(errs)) <= 0.6, (max(errs), np.mean(errs))
    assert max(raw_errs) > 4.0  # the un-antialiased coverage is visibly worse: the property is not vacuous
    assert max(grad_rel) <= 0.10 and float(np.median(grad_rel)) <= 0.04, (max(grad_rel), np.median(grad_rel))


# ------------------------------------------------------------------------------------------------ round 4: perspective and winding
def _with_w(pos, w):
    """The same NDC positions through other homogeneous coordinates: every clip-space row scaled by its own w_i > 0."""
    return pos * torch.as_tensor(w, dtype=pos.dtype).reshape(1, -1, 1)


@pytest.mark.parametrize("theta,transpose", [(90.0, False), (60.0, False), (75.0, True), (120.0, False)])
def test_antialias_silhouette_under_perspective_equals_the_affine_one_and_w_gradient_identity(theta, transpose, dev, ops):
    """The operator works on x/w, y/w (render.py:264-267 hands it clip-space positions): a straight silhouette whose vertices carry
    w_i in {0.5, 1, 2, 0.7} (same NDC line) must antialias exactly like the w == 1 scene, whose closed form the tests above pin.  And
    because the output depends on (x_i, y_i, w_i) only through x_i/w_i, y_i/w_i, its gradients obey, vertex by vertex,
        d/dx_i = (1/w_i) d/d(x_i/w_i),  d/dw_i = -(x_i/w_i^2) d/d(x_i/w_i) - (y_i/w_i^2) d/d(y_i/w_i)
    with d/d(x_i/w_i) = the w == 1 scene's own position gradient.  Through the C ABI; no oracle."""
    H = W = 32
    pos1, tri = _halfplane_scene(theta, 15.37, 16.21, H, W, transpose)
    w = torch.tensor([0.5, 1.0, 2.0, 0.7])
    g = torch.Generator().manual_seed(3)
    weight = torch.rand(1, H, W, 2, generator=g).to(dev)  # an arbitrary linear functional of the output
    outs, grads = [], []
    for pos in (pos1, _with_w(pos1, w)):
        p = pos.clone().to(dev).requires_grad_(True)
        rast = ops.rasterize(p, tri.to(dev), (H, W)).detach()
        colour = torch.cat([(rast[..., 3:] > 0).float(), 0.2 + 0.6 * (rast[..., 3:] > 0).float()], -1).contiguous()
        out = ops.antialias(colour, rast, p, tri.to(dev))
        (out * weight).sum().backward()
        outs.append(out.detach().cpu())
        grads.append(p.grad.detach().cpu()[0].double())
    np.testing.assert_allclose(outs[1].numpy(), outs[0].numpy(), atol=2e-6)
    assert float((outs[0][..., 0] - (outs[0][..., 0] > 0.5).float()).abs().max()) > 0.2  # the silhouette really was blended
    g1, gw = grads  # g1 = d/d(ndc) (w == 1), gw = clip-space gradient under perspective
    wd, pw = w.double(), _with_w(pos1, w)[0].double()
    assert float(g1[:, :2].abs().max()) > 1.0
    scale = float(g1[:, :2].abs().max())
    np.testing.assert_allclose((gw[:, :2] * wd[:, None]).numpy(), g1[:, :2].numpy(), atol=2e-4 * scale)
    want_w = -(pw[:, 0] / wd ** 2) * g1[:, 0] - (pw[:, 1] / wd ** 2) * g1[:, 1]
    np.testing.assert_allclose(gw[:, 3].numpy(), want_w.numpy(), atol=2e-4 * max(scale, float(want_w.abs().max())))
    assert float(gw[:, 2].abs().max()) == 0.0  # nothing to z
    # the w == 1 scene's w gradient is the same identity at w = 1 (the share the closed-form tests leave unchecked)
    np.testing.assert_allclose(g1[:, 3].numpy(), (-(pos1[0, :, 0].double()) * g1[:, 0] - pos1[0, :, 1].double() * g1[:, 1]).numpy(),
                               atol=2e-4 * scale)


def test_antialias_coverage_integral_under_a_random_projective_map(dev, ops):
    """The coverage-integral property again, with every vertex given its own w in (0.4, 2.5): sum of the antialiased alpha = projected
    area (the NDC triangle is unchanged) and the clip-space gradients are the area's, chain-ruled through x/w, y/w -- d(area)/d(x_c) =
    (1/w) d(area)/d(ndc_x), d(area)/dw = -(x_c/w^2) d(area)/d(ndc_x) - (y_c/w^2) d(area)/d(ndc_y), from float64 autograd of the
    shoelace formula.  Same bounds as the affine test (1.5 px^2; gradient within 10 % in norm)."""
    H = W = 64
    g = torch.Generator().manual_seed(1)
    tri = torch.tensor([[0, 1, 2]], dtype=torch.int32).to(dev)
    errs, grad_rel = [], []
    for _ in range(25):
        while True:
            p = torch.rand(3, 2, generator=g) * 1.6 - 0.8
            w = 0.4 + 2.1 * torch.rand(3, generator=g)
            clip64 = torch.cat([p.double() * w.double()[:, None], torch.zeros(3, 1, dtype=torch.float64), w.double()[:, None]], -1).requires_grad_(True)
            ndc = clip64[:, :2] / clip64[:, 3:]
            px = (ndc + 1) * torch.tensor([W / 2, H / 2], dtype=torch.float64)
            area = 0.5 * ((px[1, 0] - px[0, 0]) * (px[2, 1] - px[0, 1]) - (px[2, 0] - px[0, 0]) * (px[1, 1] - px[0, 1]))
            if abs(float(area.detach())) > 150:
                break
        area.abs().backward()
        pos = clip64.detach().float()[None].to(dev).requires_grad_(True)
        rast = ops.rasterize(pos, tri, (H, W)).detach()
        cover = (rast[..., 3:] > 0).float()
        total = ops.antialias(cover.contiguous(), rast, pos, tri).sum()
        total.backward()
        gs = pos.grad[0].cpu().double()
        errs.append(abs(float(total.detach()) - abs(float(area.detach()))))
        grad_rel.append(float((gs[:, [0, 1, 3]] - clip64.grad[:, [0, 1, 3]]).norm() / clip64.grad[:, [0, 1, 3]].norm()))
    assert max(errs) <= 1.5 and float(np.mean(errs)) <= 0.6, (max(errs), np.mean(errs))
    assert max(grad_rel) <= 0.10 and float(np.median(grad_rel)) <= 0.04, (max(grad_rel), np.median(grad_rel))


def _uv_sphere(n_lat=10, n_lon=16):
    """A closed, consistently wound triangle mesh (clip-space positions [1,V,4] at w = 1, int32 triangles [F,3])."""
    v = [[0.0, 0.0, 1.0]]
    for i in range(1, n_lat):
        t = np.pi * i / n_lat
        v += [[np.sin(t) * np.cos(2 * np.pi * j / n_lon), np.sin(t) * np.sin(2 * np.pi * j / n_lon), np.cos(t)] for j in range(n_lon)]
    v.append([0.0, 0.0, -1.0])
    ring = lambda i, j: 1 + (i - 1) * n_lon + j % n_lon
    f = [[0, ring(1, j), ring(1, j + 1)] for j in range(n_lon)]
    for i in range(1, n_lat - 1):
        for j in range(n_lon):
            f += [[ring(i, j), ring(i + 1, j), ring(i + 1, j + 1)], [ring(i, j), ring(i + 1, j + 1), ring(i, j + 1)]]
    last = len(v) - 1
    f += [[last, ring(n_lat - 1, j + 1), ring(n_lat - 1, j)] for j in range(n_lon)]
    v = torch.tensor(v, dtype=torch.float32)
    rot = torch.tensor([[0.8, 0.36, -0.48], [0.0, 0.8, 0.6], [0.6, -0.48, 0.64]])  # (no pole on the view axis)
    v = v @ rot.T
    pos = torch.cat([0.7 * v[:, :2], 0.3 * v[:, 2:] , torch.ones(v.shape[0], 1)], -1)[None]
    return pos, torch.tensor(f, dtype=torch.int32)


def test_antialias_does_not_depend_on_the_winding_of_the_triangles(dev, ops):
    """Whether an edge is a silhouette is geometry, not bookkeeping: a closed surface whose triangles are wound inconsistently (marching
    tets over a grid file whose tets are not uniformly oriented gives exactly that) must antialias like the consistently wound one.
    nvdiffrast keeps the opposite vertices of an UNDIRECTED edge (two of them, whatever the traversal directions).  Until round 3 the
    topology here kept one face per traversal direction, and of two faces that traverse a shared edge the same way one did not find
    the other: its interior edges were blended as if they were boundaries (10x the silhouette records on a randomly wound mesh)."""
    H = W = 96
    pos, tri = _uv_sphere()
    g = torch.Generator().manual_seed(5)
    colour_of_face = torch.rand(tri.shape[0] + 1, 3, generator=g)
    colour_of_face[0] = 0.0

    def run(t):
        t = t.contiguous().to(dev)
        rast = ops.rasterize(pos.to(dev), t, (H, W))
        colour = colour_of_face.to(dev)[rast[..., 3].long()].contiguous()  # flat colour per triangle: every id change is a colour change
        return rast.cpu(), colour.cpu(), ops.antialias(colour, rast, pos.to(dev), t).cpu()

    rast0, colour0, out0 = run(tri)
    flip = torch.rand(tri.shape[0], generator=g) < 0.5
    tri_f = torch.where(flip[:, None], tri[:, [0, 2, 1]], tri)
    rast1, colour1, out1 = run(tri_f)
    assert torch.equal(rast0[..., 3], rast1[..., 3]) and torch.equal(colour0, colour1)  # same coverage, same owners
    touched0 = (out0 - colour0).abs().amax(-1) > 0
    assert 50 < int(touched0.sum()) < 900  # only the outline (a ~64-pixel-wide disc) is blended ...
    inside = (rast0[..., 3] > 0)
    ys, xs = torch.nonzero(inside[0], as_tuple=True)
    cy, cx = ys.float().mean(), xs.float().mean()
    r = ((ys - cy) ** 2 + (xs - cx) ** 2).sqrt()
    core = torch.zeros_like(inside[0])
    core[ys[r < 0.8 * r.max()], xs[r < 0.8 * r.max()]] = True
    assert not bool((touched0[0] & core).any())  # ... nothing inside, although every pixel pair there sees two triangle ids
    np.testing.assert_allclose(out1.numpy(), out0.numpy(), atol=1e-6)
