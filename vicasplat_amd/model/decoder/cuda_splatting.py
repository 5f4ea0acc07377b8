"""`render_cuda` call surface of /root/reference/src/model/decoder/cuda_splatting.py:148-239, MI355X-native.

Differences that do not change results: every camera of the call is rendered by ONE batched rasterizer launch
set (no Python loop over views, no tan_fov .item() host syncs, :199-238); Gaussians given per scene are shared by
the scene's cameras; the SH transpose copy (:182) and the triu covariance gather (:224) are folded into the
kernel's addressing.  `scale_invariant` is accepted and ignored, exactly like the reference (:170-178).
"""
from __future__ import annotations

from math import isqrt
from typing import Optional

import torch
from torch import Tensor

from ...geometry.projection import get_fov
from ...raster import rasterize


def get_projection_matrix(near: Tensor, far: Tensor, fov_x: Tensor, fov_y: Tensor) -> Tensor:
    """[b] each -> [b,4,4] perspective matrix with z mapped to (0,1) and no principal-point term
    (cuda_splatting.py:18-45: left=-right, bottom=-top, so the [0,2]/[1,2] entries vanish)."""
    tx, ty = (0.5 * fov_x).tan(), (0.5 * fov_y).tan()
    right, top = tx * near, ty * near
    m = torch.zeros((near.shape[0], 4, 4), dtype=torch.float32, device=near.device)
    m[:, 0, 0] = 2 * near / (right + right)
    m[:, 1, 1] = 2 * near / (top + top)
    m[:, 3, 2] = 1
    m[:, 2, 2] = far / (far - near)
    m[:, 2, 3] = -(far * near) / (far - near)
    return m


def camera_matrices(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor):
    """c2w [b,4,4], normalised K [b,3,3] -> (view^T [b,16], full^T [b,16], P^T [b,16], campos [b,3], tanfov [b,2]),
    stored the way the rasterizer expects (cuda_splatting.py:187-194)."""
    fov = get_fov(intrinsics)
    tanfov = (0.5 * fov).tan()
    proj_t = get_projection_matrix(near.float(), far.float(), fov[:, 0], fov[:, 1]).transpose(1, 2)
    view_t = torch.linalg.inv(extrinsics.float()).transpose(1, 2)
    # 4x4 product written out over k in a fixed order (288 tiny matrices: not worth a vendor-BLAS batched GEMM launch)
    full_t = sum(view_t[:, :, k, None] * proj_t[:, k, None, :] for k in range(4))
    b = extrinsics.shape[0]
    return (view_t.reshape(b, 16).contiguous(), full_t.reshape(b, 16).contiguous(), proj_t.reshape(b, 16).contiguous(),
            extrinsics[:, :3, 3].float().contiguous(), tanfov.contiguous())


def render_batched(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor, image_shape: tuple[int, int],
                   background_color: Tensor, means: Tensor, covariances: Tensor, harmonics: Tensor, opacities: Tensor,
                   cam_scene: Optional[Tensor], cam_rot_delta=None, cam_trans_delta=None, use_sh: bool = True,
                   sh_degree: Optional[int] = None):
    """Cameras [b,...]; Gaussians [S,g,...] in the ENCODER's layout (cov [S,g,3,3], harmonics [S,g,3,d_sh]);
    cam_scene [b] maps cameras to scenes.  Returns (images [b,3,h,w], depths [b,h,w])."""
    h, w = image_shape
    n = harmonics.shape[-1]
    degree = sh_degree or isqrt(n) - 1
    view_t, full_t, proj_t, campos, tanfov = camera_matrices(extrinsics, intrinsics, near, far)
    if use_sh:
        kw = dict(shs=harmonics, sh_rgb_major=True)
    else:
        kw = dict(colors_precomp=harmonics[..., 0].contiguous())
    color, _radii, depth, _opacity, _touched = rasterize(
        means, covariances, opacities, view_t, full_t, campos, tanfov, background_color, h, w, sh_degree=degree,
        cam_scene=cam_scene, theta=cam_rot_delta, rho=cam_trans_delta, projmatrix_raw=proj_t, **kw)
    return color, depth


def render_cuda(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor, image_shape: tuple[int, int],
                background_color: Tensor, gaussian_means: Tensor, gaussian_covariances: Tensor,
                gaussian_sh_coefficients: Tensor, gaussian_opacities: Tensor, scale_invariant: bool = True,
                cam_rot_delta: Tensor | None = None, cam_trans_delta: Tensor | None = None, use_sh: bool = True,
                sh_degree: Optional[int] = None) -> tuple[Tensor, Tensor]:
    """Same signature and return as the reference's render_cuda.  Gaussian tensors are either [b,g,...] (one set
    per camera) or [g,...] (shared by all b cameras, the demo path, demo.py:226)."""
    assert use_sh or gaussian_sh_coefficients.shape[-1] == 1
    b = extrinsics.shape[0]
    if gaussian_means.ndim == 2:
        cam_scene = torch.zeros(b, dtype=torch.int32, device=extrinsics.device)
        means, covs = gaussian_means[None], gaussian_covariances[None]
        shs, opac = gaussian_sh_coefficients[None], gaussian_opacities[None]
    else:
        cam_scene = None  # camera c -> Gaussian set c
        means, covs, shs, opac = gaussian_means, gaussian_covariances, gaussian_sh_coefficients, gaussian_opacities
    return render_batched(extrinsics, intrinsics, near, far, image_shape, background_color, means, covs, shs, opac,
                          cam_scene, cam_rot_delta, cam_trans_delta, use_sh, sh_degree)
