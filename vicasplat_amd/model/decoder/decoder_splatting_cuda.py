"""DecoderSplattingCUDA with the reference's forward signature (decoder_splatting_cuda.py:38-52), rendering all
b*v target cameras in one batched rasterizer call without replicating the Gaussians v times (:86-89)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Literal, Optional

import torch
from torch import Tensor

from ..types import Gaussians
from .cuda_splatting import render_batched
from .decoder import Decoder, DecoderOutput, DepthRenderingMode


@dataclass
class DecoderSplattingCUDACfg:
    name: Literal["splatting_cuda"]
    background_color: list[float]
    make_scale_invariant: bool
    use_gsplat: bool = False  # the gsplat path of the reference is never enabled by its configs (splatting_cuda.yaml:4)


class DecoderSplattingCUDA(Decoder[DecoderSplattingCUDACfg]):
    background_color: Tensor

    def __init__(self, cfg: DecoderSplattingCUDACfg) -> None:
        super().__init__(cfg)
        if cfg.use_gsplat:
            raise NotImplementedError("use_gsplat=True is out of scope: every reference experiment sets it to false")
        self.make_scale_invariant = cfg.make_scale_invariant
        self.register_buffer("background_color", torch.tensor(cfg.background_color, dtype=torch.float32), persistent=False)

    def forward(self, gaussians: Gaussians, extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor,
                image_shape: tuple[int, int], depth_mode: DepthRenderingMode | None = None,
                cam_rot_delta: Tensor | None = None, cam_trans_delta: Tensor | None = None, use_sh: bool = True,
                active_sh_degree: Optional[int] = None, return_dict: bool = True):
        b, v = extrinsics.shape[:2]
        means, covs, shs, opac = gaussians.means, gaussians.covariances, gaussians.harmonics, gaussians.opacities
        if means.ndim > 3:  # [b, view, h, w, ...] straight from the encoder
            means, covs, shs, opac = means.flatten(1, 3), covs.flatten(1, 3), shs.flatten(1, 3), opac.flatten(1)
        cam_scene = torch.arange(b, dtype=torch.int32, device=extrinsics.device).repeat_interleave(v)
        color, depth = render_batched(
            extrinsics.flatten(0, 1), intrinsics.flatten(0, 1), near.flatten(), far.flatten(), image_shape,
            self.background_color.expand(b * v, 3), means, covs, shs, opac, cam_scene,
            None if cam_rot_delta is None else cam_rot_delta.flatten(0, 1),
            None if cam_trans_delta is None else cam_trans_delta.flatten(0, 1), use_sh, active_sh_degree)
        color = color.unflatten(0, (b, v))
        depth = depth.unflatten(0, (b, v))
        if not return_dict:
            return color, depth
        return DecoderOutput(color, depth)
