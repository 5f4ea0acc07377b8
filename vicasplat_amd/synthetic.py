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
