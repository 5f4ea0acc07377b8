"""Raw 86-channel head output -> Gaussian parameters (reference: common/gaussian_adapter.py:167-212,
common/gaussians.py:8-44, vicasplat.py:143-156)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Literal, Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn


@dataclass
class Gaussians:
    means: Tensor        # [*batch, 3]
    covariances: Tensor  # [*batch, 3, 3]
    scales: Tensor       # [*batch, 3]
    rotations: Tensor    # [*batch, 4] xyzw
    harmonics: Tensor    # [*batch, 3, d_sh]
    opacities: Tensor    # [*batch, 1]


@dataclass
class GaussianAdapterCfg:
    gaussian_scale_min: float
    gaussian_scale_max: float
    sh_degree: int = 0
    scale_act: Literal["bounded", "softplus", "exp"] = "softplus"


def quaternion_to_matrix(q: Tensor, eps: float = 1e-8) -> Tensor:
    i, j, k, r = q.unbind(-1)
    two_s = 2 / ((q * q).sum(-1) + eps)
    m = torch.stack([1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)], -1)
    return m.reshape(*q.shape[:-1], 3, 3)


def build_covariance(scale: Tensor, rotation_xyzw: Tensor) -> Tensor:
    rs = quaternion_to_matrix(rotation_xyzw) * scale[..., None, :]
    return rs @ rs.transpose(-1, -2)


class MyGaussianAdapter(nn.Module):
    def __init__(self, cfg: GaussianAdapterCfg):
        super().__init__()
        self.cfg = cfg
        mask = torch.ones((self.d_sh,), dtype=torch.float32)
        for degree in range(1, cfg.sh_degree + 1):
            mask[degree ** 2:(degree + 1) ** 2] = 0.1 * 0.25 ** degree
        self.register_buffer("sh_mask", mask, persistent=False)

    @property
    def d_sh(self) -> int:
        return (self.cfg.sh_degree + 1) ** 2

    @property
    def d_in(self) -> int:
        return 7 + 3 * self.d_sh

    def forward(self, raw_gaussians: Tensor, pdf_to_opacity_func: Optional[Callable] = None) -> Gaussians:
        xyz, opacity, scales, rotations = raw_gaussians[..., :11].split((3, 1, 3, 4), dim=-1)
        sh = raw_gaussians[..., 11:].unflatten(-1, (3, self.d_sh)) * self.sh_mask
        opacity = torch.sigmoid(opacity)
        if pdf_to_opacity_func is not None:
            opacity = pdf_to_opacity_func(opacity)
        act = self.cfg.scale_act
        if act == "bounded":
            scales = self.cfg.gaussian_scale_min + (self.cfg.gaussian_scale_max - self.cfg.gaussian_scale_min) * scales.sigmoid()
        elif act == "exp":
            scales = scales.exp().clamp_max(0.3)
        elif act == "softplus":
            scales = (0.001 * F.softplus(scales)).clamp_max(0.3)
        else:
            raise ValueError(act)
        rotations = F.normalize(rotations, dim=-1)
        return Gaussians(means=xyz, covariances=build_covariance(scales, rotations), harmonics=sh, opacities=opacity,
                         scales=scales, rotations=rotations)
