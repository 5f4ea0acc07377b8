"""Camera helpers on the hot path (only what render_cuda needs).

`get_fov` mirrors /root/reference/src/geometry/projection.py:247-261: field of view from NORMALISED intrinsics as
the angle between the back-projected left/right (top/bottom) image-border midpoints.
"""
from __future__ import annotations

import torch
from torch import Tensor


def _unit_rays(K_inv: Tensor, uv1: tuple[float, float, float]) -> Tensor:
    # K_inv @ (u, v, 1) written out (3 multiply-adds per component in a fixed order): no vendor-BLAS GEMV launch for a 3x3 product
    r = K_inv[..., 0] * uv1[0] + K_inv[..., 1] * uv1[1] + K_inv[..., 2] * uv1[2]  # [b,3]
    return r / r.norm(dim=-1, keepdim=True)


def get_fov(intrinsics: Tensor) -> Tensor:
    """[b,3,3] normalised intrinsics -> [b,2] (fov_x, fov_y) in radians."""
    K_inv = torch.linalg.inv(intrinsics.float())
    cos_x = (_unit_rays(K_inv, (0.0, 0.5, 1.0)) * _unit_rays(K_inv, (1.0, 0.5, 1.0))).sum(-1)
    cos_y = (_unit_rays(K_inv, (0.5, 0.0, 1.0)) * _unit_rays(K_inv, (0.5, 1.0, 1.0))).sum(-1)
    return torch.stack((cos_x.acos(), cos_y.acos()), dim=-1)
