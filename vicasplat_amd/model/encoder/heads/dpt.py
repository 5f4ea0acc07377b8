"""DPT output heads of VicaSplat: `pts3d` regression head (downstream_head1) and Gaussian-parameter head
(gaussian_param_head).  Reference: heads/dpt_block.py:79-218,264-419, heads/dpt_head.py:21-119,
heads/dpt_gs_head.py:98-206, heads/postprocess.py:10-63.  Parameter names == the reference's state_dict keys.

SURVEY.md 8(f)-1: the heads are 42 % of the forward FLOPs.  All 3x3 / 1x1 / transposed convolutions and the
bilinear upsampling run on the hand-written HIP kernels (implicit-GEMM MFMA conv, GEMM, upsample) on NHWC 16-bit
activations; the 7x7 RGB stem conv is a window GEMM on the same main loop (`vs_conv7x7_rgb_nhwc`; im2col + GEMM on the f32 / split paths).  HIP device tensors only.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F      # F.pad / F.unfold of the exact-f32 class's im2col stem only -- no torch convolution / interpolation anywhere
from torch import nn

import os

from .... import ops

_PTS_UP_PACKED = os.environ.get("VS_PTS_UP_PACKED", "1") != "0"     # A/B switch: 0 = f32 upsampled map into the pts3d head's fused conv (round 3)
_STEM_STREAM = os.environ.get("VS_STEM_STREAM", "1") != "0"       # A/B switch: 0 = the tile route of the fused stem (round 4)
_STEM_UP_FUSED = os.environ.get("VS_STEM_UP_FUSED", "1") != "0"   # A/B switch: 0 = stem -> f32 map -> upsample-add kernel (round 3)


def _no_torch_forward(self, *a, **k):
    raise RuntimeError(f"{type(self).__name__} is a PARAMETER CONTAINER (the reference's state_dict names): the computation runs on the HIP "
                       "kernels driven by PixelwiseTaskWithDPT -- there is no PyTorch / MIOpen forward in vicasplat_amd")


class _RCU(nn.Module):      # ResidualConvUnit_custom (dpt_block.py:79-150): parameters only
    forward = _no_torch_forward

    def __init__(self, c: int):
        super().__init__()
        self.conv1 = nn.Conv2d(c, c, 3, 1, 1, bias=True)
        self.conv2 = nn.Conv2d(c, c, 3, 1, 1, bias=True)


class _Fusion(nn.Module):   # FeatureFusionBlock_custom (dpt_block.py:153-218): parameters only
    forward = _no_torch_forward

    def __init__(self, c: int):
        super().__init__()
        self.out_conv = nn.Conv2d(c, c, 1, bias=True)
        self.resConfUnit1 = _RCU(c)
        self.resConfUnit2 = _RCU(c)


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


class _DPT(nn.Module):      # DPTOutputAdapter / its GS variant (dpt_block.py:264-419, dpt_gs_head.py:98-157): parameters only
    forward = _no_torch_forward

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


def _pad_to(n: int, m: int = 64) -> int:
    return (n + m - 1) // m * m


class PixelwiseTaskWithDPT(nn.Module):
    """`.dpt` holds the parameters (reference naming: <head>.dpt.<...>).  The forward drives the hand-written HIP
    kernels on NHWC 16-bit activations:
      * every 3x3 convolution (layer_rn, the 16 ResidualConvUnit convs, head convs, the stride-2 reassemble conv) is the
        implicit-GEMM MFMA kernel `vs_conv3x3_nhwc` with pre-activation ReLU / bias / residual / ReLU fused;
      * every 1x1 convolution and both ConvTranspose2d(k == stride) layers are plain GEMMs (`vs_gemm_bias_act`) -- tokens
        [BT, 256, C] ARE an NHWC 16x16 map, so no reshape copy is needed on the way in;
      * bilinear x2 (align_corners=True) is `vs_upsample2x_nhwc` (with the image-feature add of the GS head fused).
    The 7x7 stem conv on the RGB image is `vs_conv7x7_rgb_nhwc` (window GEMM; im2col + GEMM on the f32 / split paths).
    Channel counts that are not multiples of 64 (96, 192) are zero-padded inside the packed weights."""

    def __init__(self, net, num_channels: int, head_type: str):
        super().__init__()
        L = net.dec_depth
        assert L > 9
        self.dpt = _DPT([net.enc_embed_dim] + [net.dec_embed_dim] * 3, [0, L * 2 // 4, L * 3 // 4, L], num_channels, head_type)
        self.head_type = head_type
        self.num_channels = num_channels
        self.compute_dtype = torch.float16
        self.split = False    # split operand class: f32 activations + ops.SplitWeight weights (VicaSplat.set_compute_dtype("split"))
        self._pk: dict = {}
        self._pk_key = None

    # ---- packed 16-bit weights (re-made when a parameter changes / moves / the compute dtype changes) ----
    def _packed(self):
        d = self.dpt
        w0 = d.scratch.layer1_rn.weight
        key = (self.compute_dtype, self.split, w0.device, sum(p._version for p in self.parameters()), id(w0))
        if key == self._pk_key:
            return self._pk
        dt = self.compute_dtype
        fin = (lambda t: ops.split_pack_weight(t)) if self.split else (lambda t: t)   # (split: pack the padded f32 GEMM weight)
        P = {}

        def lin(name, conv, n_pad=0, k_pad=0):  # 1x1 conv -> GEMM weight [N(+pad), K(+pad)], f32 bias [N(+pad)]
            w = conv.weight.detach().flatten(1)
            N, K = w.shape
            Np, Kp = max(n_pad, N), max(k_pad, K)
            wp = torch.zeros(Np, Kp, dtype=dt, device=w.device)
            wp[:N, :K] = w.to(dt)
            b = torch.zeros(Np, dtype=torch.float32, device=w.device)
            if conv.bias is not None:
                b[:N] = conv.bias.detach().float()
            P[name + ".w"], P[name + ".b"] = fin(wp), b

        def convT(name, ct, k_pad, co_pad):  # ConvTranspose2d(k == stride) -> GEMM weight [(i, j, co_pad), ci_pad]
            w = ct.weight.detach()  # [Cin, Cout, k, k]
            Cin, Cout, k, _ = w.shape
            wp = torch.zeros(k, k, co_pad, k_pad, dtype=dt, device=w.device)
            wp[:, :, :Cout, :Cin] = w.permute(2, 3, 1, 0).to(dt)
            b = torch.zeros(k, k, co_pad, dtype=torch.float32, device=w.device)
            b[:, :, :Cout] = ct.bias.detach().float()
            P[name + ".w"], P[name + ".b"] = fin(wp.reshape(k * k * co_pad, k_pad).contiguous()), b.reshape(-1).contiguous()

        def c3(name, conv, cin_pad=0):
            P[name + ".w"] = ops.pack_conv3x3_weight(conv.weight, "split" if self.split else dt, cin_pad)
            P[name + ".b"] = None if conv.bias is None else conv.bias.detach().float().contiguous()

        ap = d.act_postprocess
        c0, c1 = _pad_to(ap[0][0].out_channels), _pad_to(ap[1][0].out_channels)   # 96 -> 128, 192 -> 192? (192 % 64 == 0)
        lin("ap0.0", ap[0][0], n_pad=c0); convT("ap0.1", ap[0][1], k_pad=c0, co_pad=c0)
        lin("ap1.0", ap[1][0], n_pad=c1); convT("ap1.1", ap[1][1], k_pad=c1, co_pad=c1)
        lin("ap2.0", ap[2][0])
        lin("ap3.0", ap[3][0]); c3("ap3.1", ap[3][1])
        for i, cp in enumerate((c0, c1, 0, 0)):
            c3(f"rn{i}", d.scratch.layer_rn[i], cp)
        for r in (1, 2, 3, 4):
            f = getattr(d.scratch, f"refinenet{r}")
            for u in ("resConfUnit1", "resConfUnit2"):
                c3(f"rf{r}.{u}.c1", getattr(f, u).conv1); c3(f"rf{r}.{u}.c2", getattr(f, u).conv2)
            lin(f"rf{r}.out", f.out_conv)
        if self.head_type == "regression":
            c3("h0", d.head[0]); c3("h2", d.head[2]); lin("h4", d.head[4])
            lin("h4f", d.head[4], n_pad=4)                       # fused form: [4, 128] rows (row 3 zero), bias [4]
        else:
            c3("h0", d.head[0]); lin("h4", d.head[4])
            lin("h4f", d.head[4], n_pad=_pad_to(self.num_channels, 16))   # fused form: channels padded to a multiple of 16
        self._pk, self._pk_key = P, key
        return P

    def _stem_weights(self):
        P = self._packed()
        if "stem.w" not in P:
            c = self.dpt.input_merger[0]
            P["stem.w"] = ops.pack_conv7x7_rgb_weight(c.weight, self.compute_dtype)
            P["stem.b"] = c.bias.detach().float().contiguous()
        return P["stem.w"], P["stem.b"]

    def _stem_f32(self, frames: torch.Tensor) -> torch.Tensor:
        """7x7 stem of the reference-precision (f32) path: im2col rows (c, ky, kx order = weight.flatten(1)), K padded 147 -> 160,
        through the f32 MFMA GEMM; a few frames at a time (65 536 x 160 floats per frame)."""
        c = self.dpt.input_merger[0]
        N, _, H, W = frames.shape
        wk = F.pad(c.weight.detach().float().flatten(1), (0, 160 - 147)).contiguous()
        if self.split:
            P = self._packed()
            if "stem.ws" not in P:
                P["stem.ws"] = ops.split_pack_weight(wk)
            wk = P["stem.ws"]
        bias = c.bias.detach().float().contiguous()
        out = torch.empty(N, H, W, wk.shape[0], dtype=torch.float32, device=frames.device)
        step = 8
        for i in range(0, N, step):
            cols = F.pad(F.unfold(frames[i:i + step].float(), 7, padding=3).transpose(1, 2), (0, 160 - 147)).reshape(-1, 160).contiguous()
            ops.gemm(cols, wk, bias, out[i:i + step].view(-1, wk.shape[0]), ops.EPI_STORE16)
        return out

    @staticmethod
    def _gemm1x1(x, P, name, n_out=None):
        """x [..., K] NHWC 16-bit -> [..., N] via the GEMM kernel (1x1 convolution)."""
        w, b = P[name + ".w"], P[name + ".b"]
        lead = x.shape[:-1]
        a = x.reshape(-1, x.shape[-1])
        out = torch.empty(a.shape[0], w.shape[0], dtype=x.dtype, device=x.device)
        ops.gemm(a, w, b, out, ops.EPI_STORE16)
        return out.view(*lead, w.shape[0])

    @staticmethod
    def _rcu(x, P, name, extra=None):
        """ResidualConvUnit (dpt_block.py:79-142): x + conv2(relu(conv1(relu(x)))) (+ extra: a second residual added in the same epilogue)."""
        t = ops.conv3x3_nhwc(x, P[name + ".c1.w"], P[name + ".c1.b"], relu_in=True, relu_out=True)
        return ops.conv3x3_nhwc(t, P[name + ".c2.w"], P[name + ".c2.b"], residual=x, residual2=extra)

    def _fusion(self, P, r, x, skip=None, packed_out=False):
        if skip is not None:
            if self.split:      # x + rcu(skip): the add rides on the epilogue of the unit's last convolution
                x = self._rcu(skip, P, f"rf{r}.resConfUnit1", extra=x.contiguous())
            else:
                x = x + self._rcu(skip, P, f"rf{r}.resConfUnit1")
        # out_conv (1x1) commutes with the bilinear x2 (both linear, interpolation weights sum to 1): run it on the 4x
        # smaller map, then upsample (dpt_block.py:210-218 upsamples first)
        return ops.upsample2x_nhwc(self._gemm1x1(self._rcu(x, P, f"rf{r}.resConfUnit2"), P, f"rf{r}.out"), packed=packed_out)

    def _trunk(self, tokens, gh: int, gw: int, packed_out: bool = False):
        """tokens[hook] [BT, gh*gw, C] 16-bit -> path_1 [BT, 8gh, 8gw, 256] (dpt_head.py:35-62)."""
        P = self._packed()
        d = self.dpt
        dt = self.compute_dtype
        t = [tokens[h] for h in d.hooks]
        BT = t[0].shape[0]
        maps = [x.reshape(BT, gh, gw, x.shape[-1]).to(dt).contiguous() for x in t]
        # reassemble: 1x1 (+ ConvT k=s as a GEMM followed by a depth-to-space copy)
        def convT(x, name, k):
            y = self._gemm1x1(x, P, name)                         # [BT, gh, gw, k*k*Cp]
            Cp = y.shape[-1] // (k * k)
            return y.view(BT, gh, gw, k, k, Cp).permute(0, 1, 3, 2, 4, 5).reshape(BT, gh * k, gw * k, Cp)
        l0 = convT(self._gemm1x1(maps[0], P, "ap0.0"), "ap0.1", 4)
        l1 = convT(self._gemm1x1(maps[1], P, "ap1.0"), "ap1.1", 2)
        l2 = self._gemm1x1(maps[2], P, "ap2.0")
        l3 = ops.conv3x3_nhwc(self._gemm1x1(maps[3], P, "ap3.0"), P["ap3.1.w"], P["ap3.1.b"], stride=2)
        l0, l1, l2, l3 = [ops.conv3x3_nhwc(l.contiguous(), P[f"rn{i}.w"], None) for i, l in enumerate((l0, l1, l2, l3))]
        p4 = self._fusion(P, 4, l3)[:, :l2.shape[1], :l2.shape[2]].contiguous()
        p3 = self._fusion(P, 3, p4, l2)
        p2 = self._fusion(P, 2, p3, l1)
        # packed_out (split class, pts3d head): path_1 only feeds a 3x3 convolution -- the bilinear kernel writes it packed (hi, lo)
        return self._fusion(P, 1, p2, l0, packed_out=packed_out), P

    def forward_pts3d_raw(self, tokens, gh: int, gw: int) -> torch.Tensor:
        """-> [BT,C,H,W] view (channels-last memory, pixel stride 4) of the head output in the compute dtype, BEFORE the 'exp' post-process;
        C = 3 (xyz), or 4 with the confidence channel (predict_conf: the same fused kernels, whose fourth output column was zero padding)."""
        nc = self.num_channels
        pk = _PTS_UP_PACKED and self.split and self.dpt.head[0].out_channels == 128 and self.dpt.head[0].in_channels in (64, 128, 256)
        x, P = self._trunk(tokens, gh, gw, packed_out=pk and (tokens[self.dpt.hooks[0]].shape[0] * 64 * gh * gw) >= 224 * 256)
        x = ops.conv3x3_nhwc(x, P["h0.w"], P["h0.b"])
        npix = x.shape[0] * 4 * x.shape[1] * x.shape[2]
        if _PTS_UP_PACKED and self.split and npix % 256 == 0 and x.shape[-1] == 128 and self.dpt.head[2].out_channels == 128:
            # round 4: the upsampled 256^2 x 128 map is written PACKED (hi, lo) by the bilinear kernel (same bytes) and the fused conv3 -> ReLU ->
            # dot head reads it without converting (conv3x3_256x128_split_kernel<., ., A_PACKED>)
            if "h4f32.w" not in P:
                c4 = self.dpt.head[4]
                P["h4f32.w"] = torch.nn.functional.pad(c4.weight.detach().float().flatten(1), (0, 0, 0, 4 - c4.out_channels)).contiguous()
                P["h4f32.b"] = torch.nn.functional.pad(c4.bias.detach().float(), (0, 4 - c4.out_channels)).contiguous()
            xp = ops.upsample2x_nhwc(x, packed=True)
            y = ops.conv3x3_head1x1_nhwc(xp, P["h2.w"], P["h2.b"], P["h4f32.w"], P["h4f32.b"], nc)          # [BT,H,W,4] f32
            return y[..., :nc].permute(0, 3, 1, 2)
        x = ops.upsample2x_nhwc(x)
        npix = x.shape[0] * x.shape[1] * x.shape[2]
        if self.split and npix % 256 == 0 and x.shape[-1] == 128 and self.dpt.head[2].out_channels == 128:
            # split operands: the same fusion in f32 (conv3 -> ReLU -> three 128-long dot products per pixel), w2 stays f32
            if "h4f32.w" not in P:
                c4 = self.dpt.head[4]
                P["h4f32.w"] = torch.nn.functional.pad(c4.weight.detach().float().flatten(1), (0, 0, 0, 4 - c4.out_channels)).contiguous()
                P["h4f32.b"] = torch.nn.functional.pad(c4.bias.detach().float(), (0, 4 - c4.out_channels)).contiguous()
            y = ops.conv3x3_head1x1_nhwc(x, P["h2.w"], P["h2.b"], P["h4f32.w"], P["h4f32.b"], nc)          # [BT,H,W,4] f32
            return y[..., :nc].permute(0, 3, 1, 2)
        if self.compute_dtype != torch.float32 and npix % 256 == 0 and x.shape[-1] == 128:
            # conv3(128->128) -> ReLU -> conv1(128->3) in one kernel: the 128-channel activation at full resolution never reaches HBM
            y = ops.conv3x3_head1x1_nhwc(x, P["h2.w"], P["h2.b"], P["h4f.w"], P["h4f.b"], nc)          # [BT,H,W,4]
            return y[..., :nc].permute(0, 3, 1, 2)
        x = ops.conv3x3_nhwc(x, P["h2.w"], P["h2.b"], relu_out=True)
        y = self._gemm1x1(x, P, "h4f")                                                               # [BT,H,W,4]: rows >= nc are zero padding
        return y[..., :nc].permute(0, 3, 1, 2)

    def forward_pts3d(self, tokens, gh: int, gw: int) -> torch.Tensor:
        """-> [BT,H,W,3] f32 points; 'exp' depth mode (postprocess.py:46-56).  (distillation path only; the main
        path fuses this into the adapter kernel.)"""
        return self.postprocess_pts3d(self.forward_pts3d_raw(tokens, gh, gw))

    @staticmethod
    def postprocess_pts3d(raw: torch.Tensor) -> torch.Tensor:
        """[BT,>=3,H,W] raw head output -> [BT,H,W,3] f32 points, 'exp' depth mode: xyz / |xyz| * expm1(|xyz|) (postprocess.py:46-56)."""
        xyz = raw[:, :3].float().permute(0, 2, 3, 1)
        dist = xyz.norm(dim=-1, keepdim=True)
        return xyz / dist.clip(min=1e-8) * torch.expm1(dist)

    def forward_gs(self, tokens, frames: torch.Tensor, gh: int, gw: int) -> torch.Tensor:
        """-> [BT,C,H,W] view (channels-last memory) of the raw Gaussian parameters (dpt_gs_head.py:120-157)."""
        x, P = self._trunk(tokens, gh, gw)
        d = self.dpt
        dt = self.compute_dtype
        # 7x7 stem on the RGB image (dpt_gs_head.py:112-118): window GEMM on the zero-bordered NHWC frames, bias fused; its
        # ReLU is fused into the upsample-add kernel below
        fuse_gs = (self.split and self.num_channels <= 96 and (x.shape[0] * 4 * x.shape[1] * x.shape[2]) % 256 == 0
                   and self.dpt.head[0].out_channels == 256 and x.shape[-1] in (32, 64, 128, 256))
        if self.split and self.dpt.input_merger[0].out_channels % 256 == 0:
            P = self._packed()
            if "stem.ws7" not in P:
                c7 = self.dpt.input_merger[0]
                P["stem.ws7"], P["stem.b"] = ops.pack_conv7x7_rgb_weight(c7.weight, "split"), c7.bias.detach().float().contiguous()
            H_, W_ = frames.shape[-2], frames.shape[-1]
            if (_STEM_UP_FUSED and fuse_gs and x.shape[-1] == d.input_merger[0].out_channels and x.is_contiguous()
                    and (2 * x.shape[1], 2 * x.shape[2]) == (H_, W_)):
                # round 4: up2(trunk) + relu(stem) leaves the STEM kernel's epilogue in the packed form -- the f32 stem map (12.9 GB written and
                # read back per 24-scene step) and the stand-alone upsample-add launch are gone
                if _STEM_STREAM and W_ % 32 == 0 and d.input_merger[0].out_channels == 256:
                    # round 5: the streaming form (csrc/stem_stream.hip): the image as an LDS ring, taps by transpose reads, 9.4 -> see DESIGN
                    if "stem.w32" not in P:
                        c7 = d.input_merger[0]
                        P["stem.w32"] = c7.weight.detach().float().contiguous()
                        P["stem.e"] = ops.split_scale_exp(P["stem.w32"])
                    xp = ops.stem7x7_up_split_stream(ops.pad_rgb_nhwc(frames, torch.float32), P["stem.w32"], P["stem.b"], H_, W_, x, P["stem.e"])
                else:
                    xp = ops.conv7x7_rgb_nhwc(ops.pad_rgb_nhwc(frames, torch.float32), P["stem.ws7"], P["stem.b"], H_, W_, up_add=x)
                y = ops.conv3x3_head1x1_nhwc(xp, P["h0.w"], None, P["h4f.w"], P["h4f.b"], self.num_channels)   # [BT,H,W,96] f32
                return y[..., :self.num_channels].permute(0, 3, 1, 2)
            img = ops.conv7x7_rgb_nhwc(ops.pad_rgb_nhwc(frames, torch.float32), P["stem.ws7"], P["stem.b"], H_, W_)
        elif dt == torch.float32:
            img = self._stem_f32(frames)
        else:
            P7 = self._stem_weights()
            img = ops.conv7x7_rgb_nhwc(ops.pad_rgb_nhwc(frames, dt), P7[0], P7[1], frames.shape[-2], frames.shape[-1])
        if fuse_gs:
            # split operands: conv3(256->256) -> ReLU -> conv1(256->83) in one kernel, the second GEMM in four K-quarters on (hi, lo) images
            # of the tile in LDS (P["h4f.w"] is the packed [96, 256] weight in this class).  The upsample-add kernel writes the 256 x 256 x 256
            # operand in the packed (hi, lo) form (same bytes as f32): the convolution's main loop has no conversion left
            x = ops.upsample2x_nhwc(x, add=img, relu_add=True, packed=True)
            y = ops.conv3x3_head1x1_nhwc(x, P["h0.w"], None, P["h4f.w"], P["h4f.b"], self.num_channels)   # [BT,H,W,96] f32
            return y[..., :self.num_channels].permute(0, 3, 1, 2)
        x = ops.upsample2x_nhwc(x, add=img, relu_add=True)
        if (dt != torch.float32 and self.num_channels <= 96 and (x.shape[0] * x.shape[1] * x.shape[2]) % 256 == 0
                and self.dpt.head[0].out_channels == 256 and x.shape[-1] in (64, 128, 256, 512)):
            # conv3(256->256) -> ReLU -> conv1(256->83) in one kernel (dpt_block.py:335-343; Dropout(0.1) is the identity at inference)
            y = ops.conv3x3_head1x1_nhwc(x, P["h0.w"], None, P["h4f.w"], P["h4f.b"], self.num_channels)   # [BT,H,W,96]
            return y[..., :self.num_channels].permute(0, 3, 1, 2)
        x = ops.conv3x3_nhwc(x, P["h0.w"], None, relu_out=True)  # Dropout(0.1) is the identity at inference
        y = self._gemm1x1(x, P, "h4")
        return y.permute(0, 3, 1, 2)
