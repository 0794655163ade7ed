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


# A "trained-like" mesh the driver can see (bench.py --mesh spiky): what the synthetic training drifts into after a few hundred optimiser
# steps is the quadruped with a percent of its vertices pulled out into thin spikes (profiles/r04_long_run_diag.txt, step 600: mean
# pixel box 36-46, a few boxes of 9e3-1.7e4 pixels, 3.1-3.6e5 covered pixels where the fresh mesh has 14 / 129 / 2.0e5; tools/spiky_diag.py:
# 36 / 9.5e3 / 2.75e5 for these parameters, 6 % of the boxes above 64 pixels) -- and the rasteriser's and
# the compositor backward's cost follow depth complexity and silhouette length.  The spikes are a fixed, smooth displacement field of
# the CANONICAL position (sparse peaks of a product of sines, pushed radially away from the body axis), added to the rest vertices where
# the instance deformation is added (InstancePredictorBase.py:306-313): an input generator like the quadruped SDF, shared with the oracle.
# (two scales: narrow peaks = the few spikes of thousands of pixels; broad, low bumps = the tenth of the triangles stretched above 64)
SPIKES = dict(omega=8.0, tau=0.94, length=3.0, gamma=1.0, phase=(0.3, 1.1, 2.0), omega2=3.0, tau2=0.3, length2=1.0, phase2=(1.7, 0.2, 0.9),
              centre=(0.0, 0.45, 0.0))


_SPIKE_CONSTANTS = {}


def _spike_constant(pts, values):
    """A small constant as a tensor of pts' device / dtype, uploaded once (a per-call new_tensor is a pageable host -> device copy that
    blocks the host until the stream drains: three of them made the trained-like step five read-backs long instead of two)."""
    key = (pts.device, pts.dtype, tuple(values))
    if key not in _SPIKE_CONSTANTS:
        if len(_SPIKE_CONSTANTS) > 64:
            _SPIKE_CONSTANTS.clear()
        _SPIKE_CONSTANTS[key] = torch.tensor(values, dtype=pts.dtype, device=pts.device)
    return _SPIKE_CONSTANTS[key]


def synthetic_spikes(pts, params=None):
    """[...,3] canonical positions -> [...,3] displacement (zero off the peaks).  Pure torch, any device / dtype."""
    q = dict(SPIKES, **(params or {}))
    amp = 0.0
    for om, tau, length, ph in ((q["omega"], q["tau"], q["length"], q["phase"]), (q["omega2"], q["tau2"], q["length2"], q["phase2"])):
        g = torch.sin(om * pts + _spike_constant(pts, ph)).prod(-1).abs()
        amp = amp + length * ((g - tau) / (1.0 - tau)).clamp(min=0.0) ** q["gamma"]
    d = pts - _spike_constant(pts, q["centre"])
    return amp[..., None] * d / d.norm(dim=-1, keepdim=True).clamp(min=1e-6)


FUSED_LOSSES = True  # reconstruction losses as one HIP kernel each way (csrc/losses.hip) instead of ~45 torch launches

WORKLOADS = ("magicpony", "fauna", "ponymation")
REG_WEIGHTS = dict(arti_reg=0.1, deform_reg=10.0, prior_normal_reg=0.0, mask_random=0.1, flow=1.0)  # magicpony.yaml:136-140; fauna / ponymation


class SyntheticScene(torch.nn.Module):
    """One training iteration of the hot path on synthetic inputs.  ``workload`` selects what surrounds it:

    magicpony  -- train_magicpony_horse, BASELINE configs[2]: one category-level SDF, bones once per epoch, one render per iteration.
                  ``deform=True`` adds the instance deformation of the post-90k regime (InstancePredictorBase.py:306-313: a coordinate
                  MLP on the prior vertices -> Mesh.deform -> a THIRD make_mesh per iteration over the B deformed meshes) and the
                  articulation / deformation regularisers (AnimalModel.py:309-328).
    fauna      -- train_fauna per rank, configs[3]: the SDF is a weight-modulated field conditioned on the batch's 128-d class
                  embedding (CoordMLP_Mod, dmtet.py:187-189), bones and kinematic chain are re-estimated EVERY iteration with
                  bone_y_threshold 0.4 (InstancePredictorFauna.py:79-100), and a second, texture-less one-sided render of the posed
                  meshes from random azimuths feeds the mask discriminator (Fauna.py:111-173; the discriminator itself is
                  model/networks and is replaced by a fixed quadratic on the mask).
    ponymation -- train_ponymation stage 2 with rendering, configs[4]: B sequences x ``num_frames`` frames, [B,F] skinning of the shared
                  prior, B*F meshes and frames rendered with the 'flow' mode next to 'shaded' / 'dino_pred' (render.py:281-288), flow
                  loss between consecutive frames (AnimalModel.py:285-298).
    """

    def __init__(self, grid_res=64, batch=16, resolution=(256, 256), device="cuda", seed=0, net_width=256, net_layers=None, feat_dim=256,
                 embedder_freq=None, spatial_scale=7.0, temperature=0.05, jitter_grid=0.05, leg_radius=None, lr=1e-4, data_seed=None,
                 workload="magicpony", num_frames=1, deform=False, pose_seed=0, grid=None, mesh="quadruped", spikes=None, render=True,
                 mesh_loss=True):
        """``seed`` fixes the networks, cameras and poses; ``data_seed`` (default: ``seed``) the image features and the target images --
        data-parallel ranks share the former (equal work per GPU: the same number of covered pixels) and differ in the latter.
        ``grid``: a tetgrid.named_grid name ('bcc51s' = the reference's "128" Quartet class in a file's arbitrary numbering) instead of
        the Kuhn grid of ``grid_res`` cells.
        ``pose_seed`` != 0 draws other cameras / articulations with the SAME networks (per-rank poses: unequal covered-pixel counts).
        ``mesh`` = 'spiky': the trained-like mesh (synthetic_spikes; ``spikes`` overrides entries of SPIKES).
        ``render=False`` (ponymation): the step config/train_ponymation_horse_stage2.yaml really runs -- ``enable_render: false``,
        AnimalModel.py:404: DMTet, the instance deformation, [B,F] skinning and the make_mesh passes (whose vertex normals the reference
        computes eagerly, mesh.py:355-375) on B x F meshes, no rasteriser; the losses are the motion VAE's (teacher MSE on the angles, KL:
        Ponymation.py:65-85) + the articulation / deformation regularisers.  None of them reads the posed meshes, so the reference's
        backward never enters skinning or normals there; ``mesh_loss`` adds a fixed linear functional of the posed vertices and normals
        so that their backward kernels run at this size too (what stage 1 gets from the renderer)."""
        super().__init__()
        assert workload in WORKLOADS, workload
        assert mesh in ("quadruped", "spiky"), mesh
        self.mesh_kind, self.spike_params = mesh, (dict(SPIKES, **(spikes or {})) if mesh == "spiky" else None)
        assert render or workload == "ponymation", "only the sequence workload has a configuration without rendering"
        self.render, self.mesh_loss = bool(render), bool(mesh_loss)
        assert num_frames == 1 or workload == "ponymation"
        data_seed = seed if data_seed is None else data_seed
        self.workload, self.num_frames, self.deform = workload, int(num_frames), bool(deform)
        self.batch, self.resolution, self.temperature = batch, tuple(resolution), temperature
        self.last = {}
        dev = torch.device(device)
        self.dev = dev
        torch.manual_seed(seed)
        layers = dict(sdf=5, texture=8, dino=5, light=5, deform=5)
        if net_layers is not None:
            layers = {k: net_layers for k in layers}
        freq = dict(sdf=8, texture=10, dino=8, deform=10)
        if embedder_freq is not None:
            freq = {k: embedder_freq for k in freq}
        if grid is not None:
            *grid, grid_res = tetgrid.named_grid(grid)
            grid = tuple(grid)
        else:
            grid = tetgrid.kuhn_grid(grid_res)
        if leg_radius is None:  # keep the legs a few cells thick on coarse grids
            leg_radius = max(0.2, 1.6 * spatial_scale / grid_res)
        scalar = 2 * math.pi / spatial_scale * 0.9
        self.netShape = _SyntheticGeometry(grid_res, spatial_scale, num_layers=layers["sdf"], hidden_size=net_width, embedder_freq=freq["sdf"],
                                           jitter_grid=jitter_grid, symmetrize=True, device=dev, tet_grid=grid, leg_radius=leg_radius,
                                           condition_choice="mod" if workload == "fauna" else None)
        self.netTexture = hostnets.CoordMLP(3, 9, layers["texture"], nf=net_width, activation="sigmoid", min_max=torch.tensor([[0.0, 1.0]] * 9),
                                            n_harmonic_functions=freq["texture"], embedder_scalar=scalar, extra_feat_dim=feat_dim, symmetrize=True)
        self.netDINO = hostnets.CoordMLP(3, 16, layers["dino"], nf=net_width, activation="sigmoid", min_max=torch.tensor([[0.0, 1.0]] * 16),
                                         n_harmonic_functions=freq["dino"], embedder_scalar=scalar)
        self.netLight = light_mod.DirectionalLight(feat_dim, layers["light"], net_width, intensity_min_max=torch.tensor([[0.0, 1.0], [0.5, 1.0]]))
        if self.deform:  # cfg_deform (magicpony.yaml:89-95)
            self.netDeform = hostnets.CoordMLP(3, 3, layers["deform"], nf=net_width, n_harmonic_functions=freq["deform"], embedder_scalar=scalar,
                                               extra_feat_dim=feat_dim, symmetrize=True)
        self.to(dev)

        # ---- synthetic stand-ins for the (unchanged) predictors' outputs: leaves that require grad
        B, F, (H, W) = batch, self.num_frames, self.resolution
        N = B * F  # rendered frames per iteration
        self.frames = N
        mvp, w2c, campos = synthetic.random_cameras(N, seed=seed + 1 + 7919 * pose_seed)
        self.mvp = mvp.to(dev).requires_grad_(True)
        self.w2c = w2c.to(dev).requires_grad_(True)
        self.campos = campos.to(dev).requires_grad_(True)
        self.feat = torch.randn(N, feat_dim, generator=torch.Generator().manual_seed(data_seed + 2)).to(dev).requires_grad_(True)
        self.arti = synthetic.seeded((B, F, 20, 3), seed + 3 + 7919 * pose_seed, -0.25, 0.25).to(dev).requires_grad_(True)
        self.class_emb = None
        if workload == "fauna":  # the memory bank's batch embedding (BasePredictorBank.py:98-102)
            self.class_emb = (0.1 * torch.randn(128, generator=torch.Generator().manual_seed(seed + 6))).to(dev).requires_grad_(True)
        self.optimizer = torch.optim.Adam(self.parameters(), lr=lr)
        # ---- bones from the un-jittered prior (once per "epoch", InstancePredictorBase.py:316-335; Fauna redoes this every iteration)
        with torch.no_grad():
            prior = self.netShape.getMesh(jitter_grid=False, feats=self.class_emb)
            self._estimate_bones(prior)
        self._eye4, self._proj = torch.eye(4, device=dev), synthetic.perspective(25.0).to(dev)
        if not self.render:  # no image targets: the motion-VAE stage supervises angles, not pixels
            self.background = torch.zeros(1, H, W, 3, device=dev)
            self._arti_gt = synthetic.seeded((B, F, 20, 3), seed + 77, -0.25, 0.25).to(dev)
            g = torch.Generator().manual_seed(seed + 78)
            self._mesh_w = [torch.randn(3, generator=g).to(dev), torch.randn(3, generator=g).to(dev)]
            return
        # ---- targets shaped like ImageDataset batches (model/dataset/ImageDataset.py:57-90)
        g = torch.Generator().manual_seed(data_seed + 4)
        self.image_gt = torch.rand(N, 3, H, W, generator=g).to(dev)
        self.dino_gt = torch.rand(N, 16, H, W, generator=g).to(dev)
        self.flow_gt = (0.05 * torch.randn(B, max(F - 1, 1), 2, H, W, generator=g)).to(dev) if F > 1 else None
        self.background = torch.zeros(N, H, W, 3, device=dev)
        with torch.no_grad():  # mask of the same animal under a perturbed articulation, + its distance transforms
            arti0 = synthetic.seeded((B, F, 20, 3), seed + 5 + 7919 * pose_seed, -0.25, 0.25).to(dev)
            mask = self.forward_render(arti0, prior=prior, modes=["shaded"], with_nets=False)[0][:, 3]
            self.mask_gt = (mask > 0.5).float()
            self.mask_dt = _distance_transforms(self.mask_gt).to(dev)
        self.mask_valid = torch.ones(N, H, W, device=dev)
        self._eye4, self._proj = torch.eye(4, device=dev), synthetic.perspective(25.0).to(dev)

    # ------------------------------------------------------------------------------------------------
    def _estimate_bones(self, prior):
        kw = dict(n_body_bones=8, n_legs=4, n_leg_bones=3, body_bones_mode="z_minmax_y+", compute_kinematic_chain=True, attach_legs_to_body=True)
        if self.workload == "fauna":
            kw["bone_y_threshold"] = 0.4  # config/model/fauna.yaml
        self.bones, self.kinematic_tree, self.bone_aux = skinning_mod.estimate_bones(prior.v_pos[None].detach(), **kw)

    def forward_render(self, arti, prior=None, modes=("shaded", "dino_pred"), with_nets=True, jitter=False):
        B, F, N = self.batch, self.num_frames, self.frames
        if prior is None:
            prior = self.netShape.getMesh(jitter_grid=jitter, feats=self.class_emb)
            if self.workload == "fauna":  # "estimate bones every iteration for fauna" (InstancePredictorFauna.py:91-92)
                self._estimate_bones(prior)
        rest, deformation = prior.v_pos[None], None  # [1,1,V,3]
        spikes = None
        if self.spike_params is not None:
            with torch.no_grad():
                spikes = synthetic_spikes(prior.v_pos.detach(), self.spike_params)  # [1,V,3], a constant of this step's canonical mesh
            rest = rest + spikes[None]
        self.last["spikes"] = spikes
        if self.deform and with_nets:
            V = prior.v_pos.shape[1]
            if getattr(self, "_frame_of_vertex", None) is None or self._frame_of_vertex.shape[0] != N * V:
                self._frame_of_vertex = torch.arange(N, device=self.dev).repeat_interleave(V)
            # netDeform(verts, feat) * 0.1 (InstancePredictorBase.py:309-312): the feature enters as one row per frame + an index
            deformation = self.netDeform.sample(prior.v_pos.expand(N, -1, -1).reshape(N * V, 3), feat=self.feat,
                                                feat_index=self._frame_of_vertex).view(N, V, 3) * 0.1
            if spikes is not None:
                deformation = deformation + spikes
            deformed = prior.deform(deformation)  # make_mesh over the N deformed meshes (InstancePredictorBase.py:313)
            rest = deformed.v_pos.view(B, F, V, 3)
            self.last["deformed"] = deformed
        verts, aux = skinning_mod.skinning(rest, self.bones, self.kinematic_tree, arti, output_posed_bones=True, temperature=self.temperature)
        verts = verts.view(N, *verts.shape[2:])
        shape = mesh_mod.make_mesh(verts, prior.t_pos_idx, prior.v_tex.expand(N, -1, -1), prior.t_tex_idx, None)
        self.last.update(prior=prior, shape=shape, skinning_aux=aux, deformation=deformation)  # (aux["posed_bones"]: on first read, like the reference's consumers)
        if not self.render and with_nets:
            self.last.pop("rast", None)
            self.last.pop("points", None)
            return None
        out = render_mod.render_mesh(None, shape, self.mvp, self.w2c, self.campos, self.netTexture if with_nets else None,
                                      self.netLight if with_nets else None, self.resolution, background=self.background, bsdf="diffuse",
                                      feat=self.feat if with_nets else None, render_modes=list(modes), prior_mesh=prior,
                                      dino_net=self.netDINO if with_nets else None, num_frames=F)
        self.last["rast"], self.last["points"] = render_mod.LAST_RAST[0], render_mod.LAST_POINTS[0]
        return out

    def random_view_mask(self, shape, prior):
        """Fauna.get_random_view_mask (Fauna.py:111-173): the posed meshes seen from a random azimuth, no texture, no light,
        one-sided shading normal; the alpha channel is the mask the discriminator sees."""
        N, dev = self.frames, self.dev
        bins = 360
        deg = torch.randint(bins, [N], device=dev)  # on the device: no host round trip inside the step
        ang = (2 * math.pi / bins) * deg.float()
        c, s_, z, o = ang.cos(), ang.sin(), torch.zeros_like(ang), torch.ones_like(ang)
        rot = torch.stack([c, z, s_, z, z, o, z, z, -s_, z, c, z, z, z, z, o], -1).view(N, 4, 4)
        w2c = self._eye4.repeat(N, 1, 1)
        w2c[:, :3, 3] = self.w2c.detach()[:, :3, 3]  # "use the predicted transition"
        mvp = (self._proj @ w2c) @ rot
        campos = (rot[:, :3, :3].transpose(2, 1) @ (-w2c[:, :3, 3])[:, :, None])[:, :, 0]
        self.last["random_view"] = dict(mvp=mvp, w2c=w2c, campos=campos, deg=deg)
        out = render_mod.render_mesh(None, shape, mvp, w2c, campos, None, None, self.resolution, background=None, bsdf="diffuse", feat=None,
                                      render_modes=["shaded"], prior_mesh=prior, dino_net=None, two_sided_shading=False, num_frames=self.num_frames)
        self.last["random_view"].update(rast=render_mod.LAST_RAST[0], points=render_mod.LAST_POINTS[0])
        return out[0][:, 3:].clamp(0, 1)

    def losses(self, shaded, dino_pred):
        """compute_reconstruction_losses (AnimalModel.py:260-307), background_mode 'none' (per frame)."""
        if FUSED_LOSSES and shaded.is_cuda:
            from . import ops

            per_image, self._both = ops.reconstruction_losses(shaded, dino_pred, self.image_gt, self.dino_gt, self.mask_gt, self.mask_dt,
                                                               self.mask_valid, return_mask=True)
            return {k: per_image[:, i] for i, k in enumerate(("mask", "mask_inv_dt", "rgb", "dino"))}
        self._both = None
        return self.losses_torch(shaded, dino_pred)

    def losses_torch(self, shaded, dino_pred):
        """The same four terms as plain torch expressions (the reference's formulation)."""
        image_pred, mask_pred = shaded[:, :3], shaded[:, 3]
        out = {}
        out["mask"] = ((mask_pred * self.mask_valid - self.mask_gt) ** 2).flatten(1).mean(1)
        out["mask_inv_dt"] = ((1 - mask_pred) * self.mask_dt[:, 0]).flatten(1).mean(1)
        both = self.eroded_mask(mask_pred)
        out["rgb"] = ((image_pred - self.image_gt).abs() * both.unsqueeze(1)).flatten(1).mean(1)
        out["dino"] = (((dino_pred - self.dino_gt) ** 2) * both.unsqueeze(1)).flatten(1).mean(1)
        return out

    def eroded_mask(self, mask_pred):
        both = ((mask_pred * self.mask_valid > 0.0).float() * self.mask_gt).detach()
        return (torch.nn.functional.avg_pool2d(both.unsqueeze(1), 3, stride=1, padding=1).squeeze(1) > 0.99).float()

    def flow_loss(self, flow_pred, mask_pred):
        """AnimalModel.py:285-298: squared flow error on the eroded common mask between consecutive frames, frames whose target
        flow exceeds 0.5 anywhere on the mask dropped, normalised by the mask's pixel count.  flow_pred [B*F,2,H,W]."""
        B, F = self.batch, self.num_frames
        H, W = self.resolution
        if FUSED_LOSSES and flow_pred.is_cuda and getattr(self, "_both", None) is not None:
            from . import ops

            return ops.flow_loss(flow_pred, self.flow_gt, self._both, B, F)  # one HIP kernel each way (csrc/losses.hip)
        pred = flow_pred.reshape(B, F, 2, H, W)[:, :-1]
        both = self.eroded_mask(mask_pred).view(B, F, H, W)[:, :-1].unsqueeze(2).expand_as(self.flow_gt)
        large = ((self.flow_gt.abs() > 0.5).float() * both).reshape(B, F - 1, -1).sum(2) > 0
        err = (pred - self.flow_gt) ** 2 * both * (~large).float()[:, :, None, None, None]
        return err.reshape(B, F - 1, -1).sum(2) / both.reshape(B, F - 1, -1).sum(2).clamp(min=1)

    def forward(self, jitter=True, sdf_reg=True):
        """Forward of one iteration -> dict(shaded, dino_pred, loss, losses).  (DDP wraps this module: its backward hooks
        all-reduce the MLP gradients over RCCL while the HIP backward kernels are still running.)"""
        modes = ["shaded", "dino_pred"] + (["flow"] if self.workload == "ponymation" and self.num_frames > 1 else [])
        if not self.render:
            return self._forward_no_render(jitter, sdf_reg)
        rendered = self.forward_render(self.arti, jitter=jitter, modes=modes)
        shaded, dino_pred = rendered[0], rendered[1]
        parts = self.losses(shaded, dino_pred)
        total = sum(LOSS_WEIGHTS[k] * v.mean() for k, v in parts.items())
        out = dict(shaded=shaded, dino_pred=dino_pred)
        if len(rendered) > 2:
            out["flow"] = rendered[2]
            parts["flow"] = self.flow_loss(rendered[2], shaded[:, 3])
            total = total + REG_WEIGHTS["flow"] * parts["flow"].mean()
        if self.deform:  # R_art, R_def (AnimalModel.py:313-316)
            parts["arti_reg"] = (self.arti ** 2).mean()
            parts["deform_reg"] = (self.last["deformation"] ** 2).mean()
            total = total + REG_WEIGHTS["arti_reg"] * parts["arti_reg"] + REG_WEIGHTS["deform_reg"] * parts["deform_reg"]
        # prior surface-normal regulariser (AnimalModel.py:317-328): computed on EVERY iteration by the unchanged caller, weight 0 by
        # default (AnimalModel.py:58) and then left out of the total (AnimalModel.py:493-494).  It reads prior_shape.v_nrm, i.e. it is
        # what makes the prior mesh's normals pass run although nothing else on the training path looks at them.
        parts["prior_normal_reg"] = prior_normal_regulariser(self.last["prior"])
        if REG_WEIGHTS["prior_normal_reg"] > 0:
            total = total + REG_WEIGHTS["prior_normal_reg"] * parts["prior_normal_reg"]
        if self.workload == "fauna":
            mask_random = self.random_view_mask(self.last["shape"], self.last["prior"])
            out["mask_random"] = mask_random
            parts["mask_random"] = ((mask_random - 0.5) ** 2).flatten(1).mean(1)  # stand-in for the mask discriminator's generator loss
            total = total + REG_WEIGHTS["mask_random"] * parts["mask_random"].mean()
        if sdf_reg and torch.is_grad_enabled():
            eikonal = ((self.netShape.get_sdf_gradient(feats=self.class_emb).norm(dim=-1) - 1) ** 2).mean()  # dmtet.py:278-281
            total = total + LOSS_WEIGHTS["sdf_gradient"] * eikonal
        out.update(loss=total, losses=parts)
        return out

    def _forward_no_render(self, jitter, sdf_reg):
        """train_ponymation_horse_stage2 as configured (enable_render false): see __init__."""
        self.forward_render(self.arti, jitter=jitter)
        shape, prior = self.last["shape"], self.last["prior"]
        nrm, prior_nrm = shape.v_nrm, prior.v_nrm  # make_mesh's normals (mesh.py:355-375 computes them eagerly for every mesh)
        parts = {}
        parts["arti_recon"] = torch.nn.functional.mse_loss(self.arti, self._arti_gt)  # L_teacher (Ponymation.py:70-74)
        parts["arti_reg"] = (self.arti ** 2).mean()
        total = parts["arti_recon"] + REG_WEIGHTS["arti_reg"] * parts["arti_reg"]
        if self.last.get("deformation") is not None:
            parts["deform_reg"] = (self.last["deformation"] ** 2).mean()
            total = total + REG_WEIGHTS["deform_reg"] * parts["deform_reg"]
        parts["prior_normal_reg"] = prior_normal_regulariser(prior)
        if self.mesh_loss:  # a linear functional of the posed meshes: the backward of skinning / normals / DMTet at this size
            parts["mesh"] = (shape.v_pos * self._mesh_w[0]).mean() + (nrm * self._mesh_w[1]).mean()
            total = total + parts["mesh"]
        if sdf_reg and torch.is_grad_enabled():
            eikonal = ((self.netShape.get_sdf_gradient(feats=self.class_emb).norm(dim=-1) - 1) ** 2).mean()
            total = total + LOSS_WEIGHTS["sdf_gradient"] * eikonal
        return dict(loss=total, losses=parts, posed=shape.v_pos, normals=nrm, prior_normals=prior_nrm)

    def step(self, backward=True, optimizer_step=None, sdf_reg=True, module=None):
        """One iteration (forward, backward, Adam).  ``module`` = the DDP wrapper of this scene when data-parallel."""
        optimizer_step = backward if optimizer_step is None else optimizer_step
        with torch.set_grad_enabled(backward):
            out = (module if module is not None else self)(jitter=backward, sdf_reg=sdf_reg)
        if backward:
            self.optimizer.zero_grad(set_to_none=True)
            for leaf in (self.mvp, self.w2c, self.campos, self.feat, self.arti, self.class_emb):
                if leaf is not None:
                    leaf.grad = None
            out["loss"].backward()
            if optimizer_step:
                _lib_poll_deferred()  # a device-side check that failed in this step (empty leg quadrant) stops it BEFORE the weights move
                self.optimizer.step()
        if render_mod.ALLOCATOR_TRIM_MODE == "step_end":  # the allocator valve, between two steps instead of inside a forward (render.py)
            render_mod.allocator_trim_at_step_end(self.dev)
        return out


def _lib_poll_deferred():
    from . import _lib

    _lib.poll_deferred()


def prior_normal_regulariser(prior):
    """R_normal of compute_regularizers (AnimalModel.py:317-328): 1 - <n_a, n_b> over the vertex pairs (0,1) and (1,2) of every face,
    uniformly weighted (the reference overwrites its radial weights with ones, :326)."""
    idx = prior.t_nrm_idx[0]
    pairs = torch.cat([idx[:, 0:2], idx[:, 1:3]], dim=0)  # [2F,2]
    nrm = prior.v_nrm[0][pairs]  # [2F,2,3]
    return (1 - (nrm[:, 0] * nrm[:, 1]).sum(-1)).mean()


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
