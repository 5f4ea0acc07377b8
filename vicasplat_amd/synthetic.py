"""Seeded synthetic inputs and deterministic random-init weights (no algorithm of the hot path lives here).

Shared by bench.py, __graft_entry__.smoke(), the golden-vector generator and the tests, so that the reference, the
CPU oracle and the HIP path all see bit-identical weights and images.  SURVEY.md 8(c)/8(d).
"""
from __future__ import annotations

import math

import torch


def golden_weights(shapes: dict, seed: int = 0, dtype=torch.float32) -> dict:
    """shapes: {state_dict key: shape}.  Sorted keys, one numpy PCG64 stream per key: >=2-D tensors get
    randn*sqrt(2/(fan_in+fan_out)), norm weights 1+0.02*randn, biases / tokens 0.02*randn (the pose head too,
    which the reference zero-initialises and would otherwise always predict identity)."""
    import zlib

    import numpy as np

    import re

    W = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        # the reference registers scratch.layer{n}_rn and scratch.layer_rn.{n-1} for the SAME tensor (Appendix C)
        canon = re.sub(r"scratch\.layer(\d)_rn\.", lambda m: f"scratch.layer_rn.{int(m.group(1)) - 1}.", k)
        rng = np.random.default_rng([seed, zlib.crc32(canon.encode())])
        r = rng.standard_normal(shp, dtype=np.float32)
        if len(shp) >= 2:
            recept = int(np.prod(shp[2:])) if len(shp) > 2 else 1
            fan_out, fan_in = shp[0] * recept, shp[1] * recept
            r *= math.sqrt(2.0 / (fan_in + fan_out))
        elif k.endswith("weight") and ("norm" in k):
            r = 1.0 + 0.02 * r
        else:
            r *= 0.02
        # Output-layer calibration so that random weights still yield a renderable scene (DESIGN.md "synthetic scene"):
        # the pts3d head's last 1x1 conv gets zero-mean rows and a small gain, the GS-parameter head a smaller gain so
        # SH colours / opacities / scales stay in their useful range.
        if k == "downstream_head1.dpt.head.4.weight":
            r = (r - r.mean(axis=1, keepdims=True)) * 0.15
        elif k == "gaussian_param_head.dpt.head.4.weight":
            r = r * 0.3
        W[k] = torch.from_numpy(np.ascontiguousarray(r)).to(dtype)
    _calibrate_scene(W, shapes)
    return W


# Measured by tools/calibrate_scene.py (CPU oracle, 8-view synthetic input, ViT-L 24+12, weights above WITHOUT this
# step): mean / std of the three output channels of downstream_head1.dpt.head.4 (pre-expm1).  The affine below maps them
# to mean (0, 0, 1.2) and std (0.5, 0.5, 0.15): after the 'exp' depth map that is a point cloud ~2.5-3 units in front of
# camera 0 whose projection covers camera 0's image with a centre-weighted density (x/z std ~0.4) -- close to what a
# trained model predicts (pixel-aligned points in frame-0 coordinates) and what keeps per-tile lists realistic.
_PTS3D_FULL_MEAN = (1.0063726902008057, -1.11910080909729, -1.4929273128509521)
_PTS3D_FULL_STD = (0.315048485994339, 0.30362898111343384, 0.37556469440460205)
_PTS3D_TARGET_MEAN = (0.0, 0.0, 1.2)
_PTS3D_TARGET_STD = (0.5, 0.5, 0.15)


def _calibrate_scene(W: dict, shapes: dict) -> None:
    wb = W.get("downstream_head1.dpt.head.4.bias")
    if wb is None:
        return
    if "backbone.enc_blocks.23.norm1.weight" not in shapes or "backbone.dec_blocks.11.norm1.weight" not in shapes:
        wb[2] += 1.2  # reduced-depth test configurations: only push the cloud in front of the camera
        return
    w = W["downstream_head1.dpt.head.4.weight"]
    for c in range(3):
        k = _PTS3D_TARGET_STD[c] / _PTS3D_FULL_STD[c]
        w[c] *= k
        wb[c] = _PTS3D_TARGET_MEAN[c] + k * (float(wb[c]) - _PTS3D_FULL_MEAN[c])


def synthetic_input(B: int, V: int, res: int = 256, seed: int = 0):
    """analytic sinusoid + noise images (then normalised), K = [[.9,0,.5],[0,.9,.5],[0,0,1]]."""
    g = torch.Generator().manual_seed(seed)
    ys, xs = torch.meshgrid(torch.arange(res, dtype=torch.float32), torch.arange(res, dtype=torch.float32), indexing="ij")
    U = torch.rand((B, V, 3, res, res), generator=g)
    img = torch.empty(B, V, 3, res, res)
    for v in range(V):
        for c in range(3):
            img[:, v, c] = 0.5 + 0.25 * torch.sin(2 * math.pi * (3 * xs + 5 * ys) / res + c + v) + 0.25 * (U[:, v, c] - 0.5)
    img = (img - 0.5) / 0.5
    K = torch.tensor([[0.9, 0, 0.5], [0, 0.9, 0.5], [0, 0, 1.0]]).expand(B, V, 3, 3).contiguous()
    return img, K


def conditioned_weights(shapes: dict, seed: int = 0, resid_gain: float = 0.1, scale_bias: float = 30.0, opacity_bias: float = -1.5,
                        dc_gain: float = 1.2, dtype=torch.float32) -> dict:
    """A CONDITIONED synthetic checkpoint: `golden_weights` with a handful of tensors rewritten so that the rendered
    image is a smooth function of the network's outputs -- on the plain random-init weights the scene is per-pixel noise in colour and
    depth, sub-pixel shifts of the Gaussians decorrelate the render and ANY TF32-class evaluation (the reference on its CUDA GPU
    included) scores 19-21 dB against an fp32 one, which makes the render PSNR uninformative for the 16-bit path.  Here:
      * every residual branch of the 36 transformer blocks (attn.proj, cross_attn.proj, mlp.fc2, mlp_cam.fc2) is scaled by `resid_gain`
        (a trained network's branches are small corrections of the stream, not equal partners);
      * the SH DC of a Gaussian is TIED TO ITS INPUT PIXEL: stem channels 0-2 / 3-5 carry relu(+rgb) / relu(-rgb) of the centre tap, the
        trunk and the 3x3 head conv pass those six channels through untouched, the final 1x1 conv forms DC_c = dc_gain * rgb_c; the
        higher SH bands keep a tenth of their random gain;
      * the Gaussians are a few pixels wide and translucent (scale bias -> sigma ~ 0.03 world units, opacity ~ 0.2), so that a pixel
        blends tens of them.
    Everything else -- architecture, tensor names, the kernels that run -- is untouched: it is a state_dict like any other."""
    W = golden_weights(shapes, seed, dtype)
    for k in W:
        if k.startswith("backbone.") and (k.endswith("attn.proj.weight") or k.endswith("attn.proj.bias") or k.endswith("mlp.fc2.weight")
                                         or k.endswith("mlp.fc2.bias") or k.endswith("mlp_cam.fc2.weight") or k.endswith("mlp_cam.fc2.bias")):
            W[k] = W[k] * resid_gain
    g = "gaussian_param_head.dpt."
    if g + "input_merger.0.weight" in W and W[g + "head.0.weight"].shape[0] >= 6:
        stem_w, stem_b = W[g + "input_merger.0.weight"], W[g + "input_merger.0.bias"]
        stem_w[:6] = 0.0
        stem_b[:6] = 0.0
        for c in range(3):
            stem_w[c, c, 3, 3] = 1.0
            stem_w[c + 3, c, 3, 3] = -1.0
        W[g + "scratch.refinenet1.out_conv.weight"][:6] = 0.0
        W[g + "scratch.refinenet1.out_conv.bias"][:6] = 0.0
        h0 = W[g + "head.0.weight"]                      # conv3 256 -> 256, no bias
        h0[:6] = 0.0
        h0[:, :6] *= 0.0                                 # the colour channels feed nothing but themselves
        for c in range(6):
            h0[c, c, 1, 1] = 1.0
        h4, b4 = W[g + "head.4.weight"], W[g + "head.4.bias"]   # [8 + 3 * d_sh, 256]: opacity | scale 3 | quat 4 | sh (rgb-major)
        d_sh = (h4.shape[0] - 8) // 3
        h4[:, :6] = 0.0
        for c in range(3):
            lo = 8 + c * d_sh
            h4[lo] = 0.0
            h4[lo, c], h4[lo, c + 3] = dc_gain, -dc_gain
            b4[lo] = 0.0
            h4[lo + 1:lo + d_sh] *= 0.1
        b4[0] = opacity_bias
        b4[1:4] = scale_bias
    # the damped residual branches change the statistics of the pts3d head's output: re-centre the cloud in front of camera 0 with the
    # constants measured for THIS checkpoint (tools/cond_calib.py: exact-f32 HIP path, 8-view smooth input), as _calibrate_scene does
    # for the plain weights
    if _COND_CALIB is not None and "backbone.enc_blocks.23.norm1.weight" in shapes and resid_gain == 0.1:
        w, wb = W["downstream_head1.dpt.head.4.weight"], W["downstream_head1.dpt.head.4.bias"]
        mean, std = _COND_CALIB
        for c in range(3):
            k = _PTS3D_TARGET_STD[c] / std[c]
            w[c] *= k
            wb[c] = _PTS3D_TARGET_MEAN[c] + k * (float(wb[c]) - mean[c])
    return W


_COND_CALIB = ((-1.2245568527050026, 0.3811309264168061, 1.6605721493175314), (0.21660378770669295, 0.36242989951717225, 0.05466144700395106))


def smooth_input(B: int, V: int, res: int = 256, seed: int = 0):
    """`synthetic_input` without the per-pixel noise term: the image the conditioned checkpoint is meant for."""
    ys, xs = torch.meshgrid(torch.arange(res, dtype=torch.float32), torch.arange(res, dtype=torch.float32), indexing="ij")
    img = torch.empty(B, V, 3, res, res)
    for b in range(B):
        for v in range(V):
            for c in range(3):
                img[b, v, c] = 0.5 + 0.4 * torch.sin(2 * math.pi * ((2 + (seed + b) % 3) * xs + (3 + c) * ys) / res + c + v)
    img = (img - 0.5) / 0.5
    K = torch.tensor([[0.9, 0, 0.5], [0, 0.9, 0.5], [0, 0, 1.0]]).expand(B, V, 3, 3).contiguous()
    return img, K
