"""DPT output heads of VicaSplat: `pts3d` regression head (downstream_head1) and Gaussian-parameter head
(gaussian_param_head).  Reference: heads/dpt_block.py:79-218,264-419, heads/dpt_head.py:21-119,
heads/dpt_gs_head.py:98-206, heads/postprocess.py:10-63.  Parameter names == the reference's state_dict keys.

SURVEY.md 8(f)-1 marks hand-written conv kernels as the NEXT row; this round the convolutions run on MIOpen
through PyTorch-ROCm in the compute dtype with channels-last activations (host code stays Python); the token
reshapes, the `exp` depth post-process and the head fusion are done here.  HIP device tensors only.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn


def _up2(x):
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)


class _RCU(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.conv1 = nn.Conv2d(c, c, 3, 1, 1, bias=True)
        self.conv2 = nn.Conv2d(c, c, 3, 1, 1, bias=True)

    def forward(self, x):
        return self.conv2(F.relu(self.conv1(F.relu(x)))) + x


class _Fusion(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.out_conv = nn.Conv2d(c, c, 1, bias=True)
        self.resConfUnit1 = _RCU(c)
        self.resConfUnit2 = _RCU(c)

    def forward(self, x, skip=None):
        if skip is not None:
            x = x + self.resConfUnit1(skip)
        return self.out_conv(_up2(self.resConfUnit2(x)))


class _Scratch(nn.Module):
    def __init__(self, layer_dims, feat):
        super().__init__()
        self.layer1_rn = nn.Conv2d(layer_dims[0], feat, 3, 1, 1, bias=False)
        self.layer2_rn = nn.Conv2d(layer_dims[1], feat, 3, 1, 1, bias=False)
        self.layer3_rn = nn.Conv2d(layer_dims[2], feat, 3, 1, 1, bias=False)
        self.layer4_rn = nn.Conv2d(layer_dims[3], feat, 3, 1, 1, bias=False)
        # same tensors under a second name, as in the reference (dpt_block.py:70-75)
        self.layer_rn = nn.ModuleList([self.layer1_rn, self.layer2_rn, self.layer3_rn, self.layer4_rn])
        self.refinenet1 = _Fusion(feat)
        self.refinenet2 = _Fusion(feat)
        self.refinenet3 = _Fusion(feat)
        self.refinenet4 = _Fusion(feat)


class _DPT(nn.Module):
    def __init__(self, dim_tokens, hooks, num_channels: int, head_type: str, feat: int = 256, layer_dims=(96, 192, 384, 768)):
        super().__init__()
        self.hooks, self.head_type = list(hooks), head_type
        ld = list(layer_dims)
        self.scratch = _Scratch(ld, feat)
        self.act_postprocess = nn.ModuleList([
            nn.Sequential(nn.Conv2d(dim_tokens[0], ld[0], 1), nn.ConvTranspose2d(ld[0], ld[0], 4, 4)),
            nn.Sequential(nn.Conv2d(dim_tokens[1], ld[1], 1), nn.ConvTranspose2d(ld[1], ld[1], 2, 2)),
            nn.Sequential(nn.Conv2d(dim_tokens[2], ld[2], 1)),
            nn.Sequential(nn.Conv2d(dim_tokens[3], ld[3], 1), nn.Conv2d(ld[3], ld[3], 3, 2, 1)),
        ])
        if head_type == "regression":
            self.head = nn.Sequential(nn.Conv2d(feat, feat // 2, 3, 1, 1), nn.Identity(), nn.Conv2d(feat // 2, feat // 2, 3, 1, 1),
                                      nn.ReLU(True), nn.Conv2d(feat // 2, num_channels, 1))
        else:  # gs_params
            self.head = nn.Sequential(nn.Conv2d(feat, feat, 3, padding=1, bias=False), nn.Identity(), nn.ReLU(True),
                                      nn.Dropout(0.1, False), nn.Conv2d(feat, num_channels, 1))
            self.input_merger = nn.Sequential(nn.Conv2d(3, feat, 7, 1, 3), nn.ReLU())

    def trunk(self, tokens, gh: int, gw: int):
        maps = []
        for idx, hk in enumerate(self.hooks):
            t = tokens[hk]  # [BT, n, C] 16-bit
            m = t.transpose(1, 2).reshape(t.shape[0], t.shape[2], gh, gw).contiguous(memory_format=torch.channels_last)
            maps.append(self.scratch.layer_rn[idx](self.act_postprocess[idx](m)))
        s = self.scratch
        p4 = s.refinenet4(maps[3])[:, :, :maps[2].shape[2], :maps[2].shape[3]]
        p3 = s.refinenet3(p4, maps[2])
        p2 = s.refinenet2(p3, maps[1])
        return s.refinenet1(p2, maps[0])


class PixelwiseTaskWithDPT(nn.Module):
    """`.dpt` holds the parameters (reference naming: <head>.dpt.<...>)."""

    def __init__(self, net, num_channels: int, head_type: str):
        super().__init__()
        L = net.dec_depth
        assert L > 9
        self.dpt = _DPT([net.enc_embed_dim] + [net.dec_embed_dim] * 3, [0, L * 2 // 4, L * 3 // 4, L], num_channels, head_type)
        self.head_type = head_type
        self.compute_dtype = torch.float16

    def _run(self, fn):
        if self.compute_dtype == torch.float32:
            return fn()
        with torch.autocast("cuda", dtype=self.compute_dtype):
            return fn()

    def forward_pts3d_raw(self, tokens, gh: int, gw: int) -> torch.Tensor:
        """-> [BT,3,H,W] head output in the compute dtype, channels-last, BEFORE the 'exp' post-process."""
        def fn():
            d = self.dpt
            x = d.head[0](d.trunk(tokens, gh, gw))
            x = d.head[2](_up2(x))
            return d.head[4](F.relu(x))
        return self._run(fn)[:, :3].contiguous(memory_format=torch.channels_last)

    def forward_pts3d(self, tokens, gh: int, gw: int) -> torch.Tensor:
        """-> [BT,H,W,3] f32 points; 'exp' depth mode (postprocess.py:46-56).  (distillation path only; the main
        path fuses this into the adapter kernel.)"""
        xyz = self.forward_pts3d_raw(tokens, gh, gw).float().permute(0, 2, 3, 1)
        dist = xyz.norm(dim=-1, keepdim=True)
        return xyz / dist.clip(min=1e-8) * torch.expm1(dist)

    def forward_gs(self, tokens, frames: torch.Tensor, gh: int, gw: int) -> torch.Tensor:
        """-> [BT,C,H,W] raw Gaussian parameters in the compute dtype, channels-last (dpt_gs_head.py:120-157)."""
        def fn():
            d = self.dpt
            x = _up2(d.trunk(tokens, gh, gw)) + d.input_merger(frames.contiguous(memory_format=torch.channels_last))
            return d.head[4](F.relu(d.head[0](x)))  # Dropout(0.1) is the identity at inference
        return self._run(fn).contiguous(memory_format=torch.channels_last)
