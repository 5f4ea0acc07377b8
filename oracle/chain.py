"""Oracle chain: CPU encoder restatement -> CPU rasterizer restatement, and the image metrics that compare a render of
the product (HIP encoder -> HIP rasterizer) with it.

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline / psnr_vs_oracle legs,
never by the product package.  The encoder half is pinned against the real reference (tests/golden/encoder_*.npz); the
rasterizer half is PARITY UNPINNED (un-vendored dependency, see raster_ref.c).

Metric: compute_psnr of /root/reference/src/evaluation/metrics.py:21-29 restated in numpy (clip both images to [0,1],
-10 log10 of the mean squared difference over the whole image).
"""
from __future__ import annotations

import numpy as np
import torch

from . import encoder_ref as er
from . import raster_ref as rr


def psnr(a: np.ndarray, b: np.ndarray) -> float:
    d = np.clip(np.asarray(a, np.float64), 0, 1) - np.clip(np.asarray(b, np.float64), 0, 1)
    mse = float((d * d).mean())
    return float("inf") if mse == 0.0 else -10.0 * float(np.log10(mse))


def config1_targets(Vt: int = 4, step: float = 0.25):
    """identity + x translations j * 0.25 (bench.py's 12 targets: j * 0.05), K = 0.9 / 0.5, near 0.01, far 100."""
    E = np.tile(np.eye(4, dtype=np.float32), (Vt, 1, 1))
    E[:, 0, 3] = step * np.arange(Vt, dtype=np.float32)
    K = np.tile(np.array([[0.9, 0, 0.5], [0, 0.9, 0.5], [0, 0, 1]], np.float32), (Vt, 1, 1))
    return E, K, np.full(Vt, 0.01, np.float32), np.full(Vt, 100.0, np.float32)


def scene_from_gaussians(g: dict, E, K, near, far, scene: int = 0) -> dict:
    """encoder_ref.forward(...)['gaussians'] (torch, [B,V,H,W,...]) of one scene -> raster_ref scene dict."""
    f = lambda t, *s: np.ascontiguousarray(t[scene].reshape(-1, *s).detach().cpu().numpy().astype(np.float32))
    return dict(means=f(g["means"], 3), covariances=f(g["covariances"], 3, 3), harmonics=f(g["harmonics"], 3, 25),
                opacities=f(g["opacities"]), extrinsics=np.asarray(E, np.float32), intrinsics=np.asarray(K, np.float32),
                near=np.asarray(near, np.float32), far=np.asarray(far, np.float32))


_MEMO: dict = {}


def _fingerprint(W: dict, cfg: dict, image, intrinsics, E, K, near, far, res, dtype, bits):
    """Content key of one oracle_chain call: names, shapes and two float64 moments of every weight, the inputs' bytes, the cameras."""
    import hashlib
    h = hashlib.blake2b(digest_size=16)
    for k in sorted(W):
        v = W[k].detach()
        h.update(k.encode()); h.update(str(tuple(v.shape)).encode())
        d = v.double()
        h.update(np.float64(float(d.sum())).tobytes()); h.update(np.float64(float((d * d).sum())).tobytes())
    h.update(repr(sorted(cfg.items())).encode())
    for t in (image, intrinsics):
        h.update(np.ascontiguousarray(t.detach().cpu().double().numpy()).tobytes())
    for a in (E, K, near, far):
        h.update(np.ascontiguousarray(np.asarray(a, np.float64)).tobytes())
    h.update(repr((res, str(dtype), bits)).encode())
    return h.hexdigest()


def oracle_chain(W: dict, cfg: dict, image: torch.Tensor, intrinsics: torch.Tensor, E, K, near, far, res: int = 256,
                 dtype=torch.float32, operand_mantissa_bits=None):
    """Memoised by CONTENT within one process (several -m gpu tests hold different precision classes of the product to the SAME oracle run:
    the 8-view chain alone is ~40 s of host time per call; VS_ORACLE_NO_MEMO=1 switches the memo off).  Callers only read the results."""
    import os
    if os.environ.get("VS_ORACLE_NO_MEMO"):
        return _oracle_chain(W, cfg, image, intrinsics, E, K, near, far, res, dtype, operand_mantissa_bits)
    key = _fingerprint(W, cfg, image, intrinsics, E, K, near, far, res, dtype, operand_mantissa_bits)
    if key not in _MEMO:
        if len(_MEMO) >= 8:
            _MEMO.pop(next(iter(_MEMO)))
        _MEMO[key] = _oracle_chain(W, cfg, image, intrinsics, E, K, near, far, res, dtype, operand_mantissa_bits)
    return _MEMO[key]


def _oracle_chain(W: dict, cfg: dict, image: torch.Tensor, intrinsics: torch.Tensor, E, K, near, far, res: int = 256,
                  dtype=torch.float32, operand_mantissa_bits=None):
    """image [1,V,3,H,W], intrinsics [1,V,3,3] -> (encoder output dict, list of per-view rasterizer dicts).
    operand_mantissa_bits = 10 emulates the reference's CUDA precision (f32 storage, TF32 matmul / conv operands,
    encoder_ref.operand_rounding); the rasterizer is f32 either way."""
    if dtype == torch.float64:
        W = {k: v.double() for k, v in W.items()}
        image, intrinsics = image.double(), intrinsics.double()
    if operand_mantissa_bits is not None:
        with er.operand_rounding(operand_mantissa_bits):
            out = er.forward(W, cfg, image, intrinsics)
    else:
        out = er.forward(W, cfg, image, intrinsics)
    sc = scene_from_gaussians(out["gaussians"], E, K, near, far)
    return out, rr.render_views(sc, res=res), sc


def compare_renders(hip_color: np.ndarray, views: list, target_seed: int = 0) -> dict:
    """hip_color [Vt,3,H,W] vs the oracle views: per-view PSNR between the two renders, and |dPSNR| of both against one
    common random target image."""
    rng = np.random.default_rng(target_seed)
    tgt = rng.uniform(0, 1, hip_color.shape[1:]).astype(np.float32)
    ps, dps, mx = [], [], []
    for c, o in enumerate(views):
        ps.append(psnr(hip_color[c], o["color"]))
        dps.append(abs(psnr(hip_color[c], tgt) - psnr(o["color"], tgt)))
        mx.append(float(np.abs(hip_color[c] - o["color"]).max()))
    return dict(psnr_between=ps, dpsnr_common_target=dps, max_abs=mx)


def tile_assignment_diff(hip_radii: np.ndarray, hip_rect: np.ndarray, views: list) -> dict:
    """Integer tile state of the two chains: Gaussians whose visibility / tile rectangle differs, and the instance counts."""
    n_vis, n_rect, R_h, R_o = 0, 0, 0, 0
    for c, o in enumerate(views):
        vo, vh = o["radii"] > 0, hip_radii[c] > 0
        n_vis += int((vo != vh).sum())
        both = vo & vh
        n_rect += int((hip_rect[c][both].astype(np.int32) != o["rect"][both]).any(-1).sum())
        rh = hip_rect[c][vh].astype(np.int64)
        R_h += int(((rh[:, 2] - rh[:, 0]) * (rh[:, 3] - rh[:, 1])).sum())
        R_o += int(o["R"])
    P = views[0]["radii"].shape[0] * len(views)
    return dict(visibility_flips=n_vis, rect_changes=n_rect, instances_hip=R_h, instances_oracle=R_o, gaussian_views=P)
