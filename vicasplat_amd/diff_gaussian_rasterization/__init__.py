"""Drop-in for the `diff_gaussian_rasterization` extension VicaSplat imports at
/root/reference/src/model/decoder/cuda_splatting.py:5-8 and calls at :207-235 (MonoGS fork with pose gradients,
requirements.txt:17).  Same names, argument meaning and error behaviour; backed by vs_raster_forward/_backward.

    settings = GaussianRasterizationSettings(image_height, image_width, tanfovx, tanfovy, bg, scale_modifier,
                                             viewmatrix, projmatrix, projmatrix_raw, sh_degree, campos, prefiltered, debug)
    image, radii, depth, opacity, n_touched = GaussianRasterizer(settings)(
        means3D=..., means2D=..., shs=..., colors_precomp=..., opacities=..., cov3D_precomp=..., theta=..., rho=...)

This single-view surface exists for source compatibility; the batched op in vicasplat_amd.raster is what the
decoder uses.
"""
from __future__ import annotations

from typing import NamedTuple, Optional

import torch
from torch import nn

from ..raster import rasterize


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    projmatrix_raw: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _cov6_from_scales_rotations(scales: torch.Tensor, rotations: torch.Tensor, scale_modifier: float) -> torch.Tensor:
    """computeCov3D of the upstream extension: rotations are (w,x,y,z), Sigma = R S S^T R^T, 6 unique entries."""
    r, x, y, z = rotations.unbind(-1)
    Rm = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                      2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                      2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    Ms = Rm * (scale_modifier * scales)[:, None, :]
    S = Ms @ Ms.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], -1)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, theta: Optional[torch.Tensor] = None, rho: Optional[torch.Tensor] = None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        if cov3D_precomp is None:
            cov3D_precomp = _cov6_from_scales_rotations(scales, rotations, float(rs.scale_modifier))
        dev = means3D.device
        tanfov = torch.tensor([[float(rs.tanfovx), float(rs.tanfovy)]], dtype=torch.float32, device=dev)
        color, radii, depth, opacity, n_touched = rasterize(
            means3D[None], cov3D_precomp[None], opacities.reshape(1, -1), rs.viewmatrix.reshape(1, 16),
            rs.projmatrix.reshape(1, 16), rs.campos.reshape(1, 3), tanfov, rs.bg.reshape(1, 3), rs.image_height,
            rs.image_width, shs=None if shs is None else shs[None],
            colors_precomp=None if colors_precomp is None else colors_precomp[None], sh_degree=rs.sh_degree,
            theta=theta, rho=rho, projmatrix_raw=rs.projmatrix_raw, count_touched=True)
        # means2D is upstream's "screen-space gradient holder"; VicaSplat never reads its .grad (cuda_splatting.py:201).
        return color[0], radii[0], depth, opacity, n_touched[0]
