"""One training step of the reconstruct-and-render hot path on synthetic inputs (bench / smoke / tests).

Mirrors what ``AnimalModel.forward`` does around the replaced modules for MagicPony in the articulated regime
(/root/reference/model/models/AnimalModel.py:356-515, call stack in SURVEY.md section 3.1):

    netBase      : DMTetGeometry.getMesh  (SDF MLP on every grid vertex -> DMTet -> make_mesh)       [HIP + torch MLP]
    netInstance  : skinning(prior verts, bones, kinematic tree, articulation angles) -> make_mesh    [HIP]
                   (the ViT encoder, pose / articulation networks are the reference's unchanged predictors and are
                    NOT part of the path: their outputs -- image feature, camera, angles -- are synthetic leaves that
                    require grad, so the backward of the path is complete)
    render       : render_mesh(['shaded','dino_pred']) with the texture / DINO / light MLPs                [HIP + torch MLPs]
    losses       : mask L2, mask inverse-distance-transform, masked RGB L1, masked DINO L2 (AnimalModel.py:260-307)
                   + SDF eikonal regulariser (dmtet.py:256-281)
    backward + Adam step on the MLP parameters.

Network sizes follow config/model/magicpony.yaml (SDF 5x256 f8, texture 8x256 f10 + 256-d feature, DINO 5x256 f8 -> 16,
light 5x256) unless overridden for small tests.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import hostnets, synthetic, tetgrid
from .model.geometry import dmtet as dmtet_mod
from .model.geometry import skinning as skinning_mod
from .model.render import light as light_mod
from .model.render import mesh as mesh_mod
from .model.render import render as render_mod

LOSS_WEIGHTS = dict(mask=10.0, mask_inv_dt=100.0, rgb=1.0, dino=10.0, sdf_gradient=0.01)  # config/model/magicpony.yaml:128-140


class _SyntheticGeometry(dmtet_mod.DMTetGeometry):
    """DMTetGeometry whose SDF is (small MLP output) + a fixed horse-like prior instead of the reference's ellipsoid
    init, so that all four leg quadrants exist for bone estimation.  Everything else is the product class."""

    def __init__(self, *args, leg_radius=0.2, mlp_gain=0.05, **kwargs):
        super().__init__(*args, init_sdf=None, **kwargs)
        self._leg_radius, self._mlp_gain = leg_radius, mlp_gain

    def get_sdf(self, pts=None, total_iter=0, feats=None):
        pts = self.verts if pts is None else pts
        prior = synthetic_quadruped_device(pts, self._leg_radius)
        return super().get_sdf(pts, total_iter=total_iter, feats=feats) * self._mlp_gain + prior[..., None]


_QUADRUPED_CONSTANTS = {}


def _quadruped_constants(device, leg_radius):
    """Capsule end points / radii of synthetic.quadruped_sdf as device tensors, uploaded once (a per-call new_tensor is a pageable
    host->device copy that blocks the host until the stream drains)."""
    key = (device, float(leg_radius))
    if key not in _QUADRUPED_CONSTANTS:
        caps = [((0, 0.7, 1.0), (0, 1.3, 1.55), 0.27), ((0, 1.3, 1.55), (0, 1.25, 1.95), 0.22)]
        for sx in (-1, 1):
            for sz in (-1, 1):
                caps.append(((0.3 * sx, 0.3, 0.85 * sz), (0.33 * sx, -1.15, 0.9 * sz), leg_radius))
        t = lambda v: torch.tensor(v, dtype=torch.float32, device=device)
        _QUADRUPED_CONSTANTS[key] = dict(a=t([c[0] for c in caps]), b=t([c[1] for c in caps]), r=t([c[2] for c in caps]),
                                         centre=t([0.0, 0.45, 0.0]), radii=t([0.5, 0.55, 1.25]))
    return _QUADRUPED_CONSTANTS[key]


def synthetic_quadruped_device(pts, leg_radius):
    """synthetic.quadruped_sdf evaluated on the tensor's own device."""
    c = _quadruped_constants(pts.device, leg_radius)
    body = (1.0 - ((pts - c["centre"]) / c["radii"]).norm(dim=-1)) * 0.5
    ab = c["b"] - c["a"]  # [6,3]
    rel = pts[..., None, :] - c["a"]  # [...,6,3]
    t = ((rel * ab).sum(-1) / (ab * ab).sum(-1)).clamp(0, 1)
    capsules = c["r"] - (rel - t[..., None] * ab).norm(dim=-1)  # [...,6]
    return torch.maximum(body, capsules.amax(-1))


FUSED_LOSSES = True  # reconstruction losses as one HIP kernel each way (csrc/losses.hip) instead of ~45 torch launches


class SyntheticScene(torch.nn.Module):
    def __init__(self, grid_res=64, batch=16, resolution=(256, 256), device="cuda", seed=0, net_width=256, net_layers=None, feat_dim=256,
                 embedder_freq=None, spatial_scale=7.0, temperature=0.05, jitter_grid=0.05, leg_radius=None, lr=1e-4, data_seed=None):
        """``seed`` fixes the networks, cameras and poses; ``data_seed`` (default: ``seed``) the image features and the target images --
        data-parallel ranks share the former (equal work per GPU: the same number of covered pixels) and differ in the latter."""
        super().__init__()
        data_seed = seed if data_seed is None else data_seed
        self.batch, self.resolution, self.temperature = batch, tuple(resolution), temperature
        self.last = {}
        dev = torch.device(device)
        self.dev = dev
        torch.manual_seed(seed)
        layers = dict(sdf=5, texture=8, dino=5, light=5)
        if net_layers is not None:
            layers = {k: net_layers for k in layers}
        freq = dict(sdf=8, texture=10, dino=8)
        if embedder_freq is not None:
            freq = {k: embedder_freq for k in freq}
        if leg_radius is None:  # keep the legs a few cells thick on coarse grids
            leg_radius = max(0.2, 1.6 * spatial_scale / grid_res)
        scalar = 2 * math.pi / spatial_scale * 0.9
        grid = tetgrid.kuhn_grid(grid_res)
        self.netShape = _SyntheticGeometry(grid_res, spatial_scale, num_layers=layers["sdf"], hidden_size=net_width, embedder_freq=freq["sdf"],
                                           jitter_grid=jitter_grid, symmetrize=True, device=dev, tet_grid=grid, leg_radius=leg_radius)
        self.netTexture = hostnets.CoordMLP(3, 9, layers["texture"], nf=net_width, activation="sigmoid", min_max=torch.tensor([[0.0, 1.0]] * 9),
                                            n_harmonic_functions=freq["texture"], embedder_scalar=scalar, extra_feat_dim=feat_dim, symmetrize=True)
        self.netDINO = hostnets.CoordMLP(3, 16, layers["dino"], nf=net_width, activation="sigmoid", min_max=torch.tensor([[0.0, 1.0]] * 16),
                                         n_harmonic_functions=freq["dino"], embedder_scalar=scalar)
        self.netLight = light_mod.DirectionalLight(feat_dim, layers["light"], net_width, intensity_min_max=torch.tensor([[0.0, 1.0], [0.5, 1.0]]))
        self.to(dev)
        self.optimizer = torch.optim.Adam(self.parameters(), lr=lr)

        # ---- synthetic stand-ins for the (unchanged) predictors' outputs: leaves that require grad
        B, (H, W) = batch, self.resolution
        mvp, w2c, campos = synthetic.random_cameras(B, seed=seed + 1)
        self.mvp = mvp.to(dev).requires_grad_(True)
        self.w2c = w2c.to(dev).requires_grad_(True)
        self.campos = campos.to(dev).requires_grad_(True)
        self.feat = torch.randn(B, feat_dim, generator=torch.Generator().manual_seed(data_seed + 2)).to(dev).requires_grad_(True)
        self.arti = synthetic.seeded((B, 1, 20, 3), seed + 3, -0.25, 0.25).to(dev).requires_grad_(True)
        # ---- bones once per "epoch" from the un-jittered prior (InstancePredictorBase.py:316-335)
        with torch.no_grad():
            prior = self.netShape.getMesh(jitter_grid=False)
            self.bones, self.kinematic_tree, self.bone_aux = skinning_mod.estimate_bones(
                prior.v_pos[None].detach(), n_body_bones=8, n_legs=4, n_leg_bones=3, body_bones_mode="z_minmax_y+", compute_kinematic_chain=True,
                attach_legs_to_body=True)
        # ---- targets shaped like ImageDataset batches (model/dataset/ImageDataset.py:57-90)
        g = torch.Generator().manual_seed(data_seed + 4)
        self.image_gt = torch.rand(B, 3, H, W, generator=g).to(dev)
        self.dino_gt = torch.rand(B, 16, H, W, generator=g).to(dev)
        self.background = torch.zeros(B, H, W, 3, device=dev)
        with torch.no_grad():  # mask of the same animal under a perturbed articulation, + its distance transforms
            arti0 = synthetic.seeded((B, 1, 20, 3), seed + 5, -0.25, 0.25).to(dev)
            mask = self.forward_render(arti0, prior=prior, modes=["shaded"], with_nets=False)[0][:, 3]
            self.mask_gt = (mask > 0.5).float()
            self.mask_dt = _distance_transforms(self.mask_gt).to(dev)
        self.mask_valid = torch.ones(B, H, W, device=dev)

    # ------------------------------------------------------------------------------------------------
    def forward_render(self, arti, prior=None, modes=("shaded", "dino_pred"), with_nets=True, jitter=False):
        B = self.batch
        if prior is None:
            prior = self.netShape.getMesh(jitter_grid=jitter)
        verts, aux = skinning_mod.skinning(prior.v_pos[None], self.bones, self.kinematic_tree, arti, output_posed_bones=True,
                                           temperature=self.temperature)
        verts = verts.view(B, *verts.shape[2:])
        shape = mesh_mod.make_mesh(verts, prior.t_pos_idx, prior.v_tex.expand(B, -1, -1), prior.t_tex_idx, None)
        self.last.update(prior=prior, shape=shape, posed_bones=aux["posed_bones"])
        out = render_mod.render_mesh(None, shape, self.mvp, self.w2c, self.campos, self.netTexture if with_nets else None,
                                      self.netLight if with_nets else None, self.resolution, background=self.background, bsdf="diffuse",
                                      feat=self.feat if with_nets else None, render_modes=list(modes), prior_mesh=prior,
                                      dino_net=self.netDINO if with_nets else None)
        self.last["rast"] = render_mod.LAST_RAST[0]
        return out

    def losses(self, shaded, dino_pred):
        """compute_reconstruction_losses (AnimalModel.py:260-307), F=1, background_mode 'none'."""
        if FUSED_LOSSES and shaded.is_cuda:
            from . import ops

            per_image = ops.reconstruction_losses(shaded, dino_pred, self.image_gt, self.dino_gt, self.mask_gt, self.mask_dt, self.mask_valid)
            return {k: per_image[:, i] for i, k in enumerate(("mask", "mask_inv_dt", "rgb", "dino"))}
        return self.losses_torch(shaded, dino_pred)

    def losses_torch(self, shaded, dino_pred):
        """The same four terms as plain torch expressions (the reference's formulation)."""
        image_pred, mask_pred = shaded[:, :3], shaded[:, 3]
        out = {}
        out["mask"] = ((mask_pred * self.mask_valid - self.mask_gt) ** 2).flatten(1).mean(1)
        out["mask_inv_dt"] = ((1 - mask_pred) * self.mask_dt[:, 0]).flatten(1).mean(1)
        both = ((mask_pred * self.mask_valid > 0.0).float() * self.mask_gt).detach()
        both = (torch.nn.functional.avg_pool2d(both.unsqueeze(1), 3, stride=1, padding=1).squeeze(1) > 0.99).float()
        out["rgb"] = ((image_pred - self.image_gt).abs() * both.unsqueeze(1)).flatten(1).mean(1)
        out["dino"] = (((dino_pred - self.dino_gt) ** 2) * both.unsqueeze(1)).flatten(1).mean(1)
        return out

    def forward(self, jitter=True, sdf_reg=True):
        """Forward of one iteration -> dict(shaded, dino_pred, loss, losses).  (DDP wraps this module: its backward hooks
        all-reduce the MLP gradients over RCCL while the HIP backward kernels are still running.)"""
        shaded, dino_pred = self.forward_render(self.arti, jitter=jitter)
        parts = self.losses(shaded, dino_pred)
        total = sum(LOSS_WEIGHTS[k] * v.mean() for k, v in parts.items())
        if sdf_reg and torch.is_grad_enabled():
            eikonal = ((self.netShape.get_sdf_gradient().norm(dim=-1) - 1) ** 2).mean()  # dmtet.py:278-281
            total = total + LOSS_WEIGHTS["sdf_gradient"] * eikonal
        return dict(shaded=shaded, dino_pred=dino_pred, loss=total, losses=parts)

    def step(self, backward=True, optimizer_step=None, sdf_reg=True, module=None):
        """One iteration (forward, backward, Adam).  ``module`` = the DDP wrapper of this scene when data-parallel."""
        optimizer_step = backward if optimizer_step is None else optimizer_step
        with torch.set_grad_enabled(backward):
            out = (module if module is not None else self)(jitter=backward, sdf_reg=sdf_reg)
        if backward:
            self.optimizer.zero_grad(set_to_none=True)
            for leaf in (self.mvp, self.w2c, self.campos, self.feat, self.arti):
                leaf.grad = None
            out["loss"].backward()
            if optimizer_step:
                self.optimizer.step()
        return out


def _distance_transforms(mask: torch.Tensor) -> torch.Tensor:
    """[B,H,W] {0,1} -> [B,2,H,W]: Euclidean distance to the mask and to its complement, normalised by the image size
    (what ImageDataset stores as mask_dt; scipy EDT since cv2 is absent)."""
    from scipy.ndimage import distance_transform_edt

    m = mask.detach().cpu().numpy() > 0.5
    out = np.zeros((m.shape[0], 2, *m.shape[1:]), dtype=np.float32)
    for b in range(m.shape[0]):
        out[b, 0] = distance_transform_edt(~m[b]) / max(m.shape[1:])
        out[b, 1] = distance_transform_edt(m[b]) / max(m.shape[1:])
    return torch.from_numpy(out)


def geometry_config1_step(inp, topology, backward=True):
    """BASELINE config 1 on the HIP path (same inputs / same loss as oracle/geometry_ref.cpu_step): DMTet -> normals -> bones ->
    skinning -> normals -> backward.  ``inp`` tensors on the GPU, ``topology`` = dmtet.TetGridTopology of the grid."""
    from . import ops

    sdf = inp["sdf"].clone().requires_grad_(backward)
    arti = inp["arti"].clone().requires_grad_(backward)
    verts, faces, uv_idx = ops.dmtet(inp["pos"], sdf, topology)
    uvs = topology.uvs()
    prior = mesh_mod.make_mesh(verts[None], faces[None], uvs[None], uv_idx[None], None)
    bones, tree, _ = skinning_mod.estimate_bones(verts[None, None].detach(), n_body_bones=8, n_legs=4, n_leg_bones=3, body_bones_mode="z_minmax_y+")
    posed, _ = skinning_mod.skinning(verts[None, None], bones, tree, arti, temperature=inp["temperature"])
    posed = posed.view(inp["batch"], -1, 3)
    shape = mesh_mod.make_mesh(posed, prior.t_pos_idx, prior.v_tex.expand(inp["batch"], -1, -1), prior.t_tex_idx, None)
    loss = (posed ** 2).mean() + (shape.v_nrm[..., 1]).mean() + (prior.v_nrm[..., 2]).mean()
    if backward:
        loss.backward()
    return dict(loss=loss, V=int(verts.shape[0]), F=int(faces.shape[0]), grad_sdf=sdf.grad, grad_arti=arti.grad)
