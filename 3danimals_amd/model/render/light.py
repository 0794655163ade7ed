"""Directional light of the hot path -- /root/reference/model/render/light.py:169-193.

The environment-light / split-sum machinery of the reference file has no caller in any config (SURVEY.md
section 2 row 16) and needs dr.texture; it is out of scope and not provided.
"""
import torch
import torch.nn.functional as F

from . import util

try:
    from model.networks import MLP  # type: ignore  (overlaid on the reference tree)
except ImportError:  # stand-alone
    from ...hostnets import MLP

_STANDALONE_ONLY = ("EnvironmentLight",)  # placeholders for stand-alone use: never overlaid on the reference's real definitions


class DirectionalLight(torch.nn.Module):
    """MLP(feat) -> upper-hemisphere direction + ambient + diffuse intensity; Lambertian shading in camera space."""

    def __init__(self, mlp_in, mlp_layers, mlp_hidden_size, intensity_min_max=None):
        super().__init__()
        self.mlp = MLP(mlp_in, 4, mlp_layers, nf=mlp_hidden_size, activation="sigmoid")
        if intensity_min_max is not None:
            self.register_buffer("intensity_min_max", intensity_min_max)
        else:
            self.intensity_min_max = None

    def forward(self, feat):
        out = self.mlp(feat)
        direction = F.normalize(torch.cat([out[..., 0:1] * 2 - 1, torch.ones_like(out[..., :1]) * 0.5, out[..., 1:2] * 2 - 1], dim=-1), dim=-1)
        intensity = out[..., 2:]
        if self.intensity_min_max is not None:
            lo, hi = self.intensity_min_max[:, 0], self.intensity_min_max[:, 1]
            intensity = intensity * (hi - lo) + lo
        self.light_params = torch.cat([direction, intensity], -1)
        return self.light_params

    def shade(self, feat, kd, normal):
        p = self.forward(feat)
        light_dir, amb, diff = p[..., :3][:, None, None, :], p[..., 3:4][:, None, None, :], p[..., 4:5][:, None, None, :]
        shading = amb + diff * torch.clamp(util.dot(light_dir, normal), min=0.0)
        return shading * kd, shading


class EnvironmentLight:  # pragma: no cover - placeholder so isinstance checks in callers keep working
    def __init__(self, *a, **k):
        raise NotImplementedError("EnvironmentLight (split-sum, needs dr.texture) is outside the hot path; no config uses it")
