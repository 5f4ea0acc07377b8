"""End-to-end render parity of the product chain (HIP encoder -> HIP rasterizer) against the oracle chain (CPU encoder
restatement, f32 and f64 -> C rasterizer restatement) on SURVEY 8(d) configs 1 / 2 / 4.  -m gpu.

What is reported and bounded per rendered view (BASELINE.json metric "render PSNR vs ref"):
  * PSNR between the two renders (compute_psnr, src/evaluation/metrics.py:21-29);
  * |dPSNR| of the two renders against one common target image;
  * the integer tile state: Gaussians whose visibility or 16x16-tile rectangle differs between the chains.

With IDENTICAL Gaussians the two rasterizers agree bit-exactly in the integer state and to |dPSNR| <= 1e-4 dB
(tests/test_raster_gpu.py).  Here the Gaussians differ by the encoder's operand rounding (16-bit operand path) or by f32
summation order only (compute dtype "f32x"), so the bounds are per precision mode and state what was measured.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import chain
from oracle import encoder_ref as er

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _model(dt):
    from vicasplat_amd.model.encoder import default_cfg, get_encoder
    shapes = json.load(open(os.path.join(G, "shapes_full.json")))
    m, _ = get_encoder(default_cfg())
    W = er.golden_weights(shapes, seed=0)
    m.load_state_dict(W, strict=True)
    m = m.cuda().eval().requires_grad_(False)     # inference: frozen weights -> the fused no-grad path of VicaSplat.forward
    m.set_compute_dtype(dt)
    return m, W


def _hip_chain(m, img, K, E, Kt, near, far):
    from vicasplat_amd.model.decoder.cuda_splatting import camera_matrices
    from vicasplat_amd.raster import forward_debug
    d = torch.device("cuda:0")
    out = m(dict(image=img.to(d), intrinsics=K.to(d)), compute_viewspace_depth=False)
    g = out["gaussians"]
    T = lambda a: torch.as_tensor(a, dtype=torch.float32, device=d)
    view_t, full_t, _proj_t, campos, tanfov = camera_matrices(T(E), T(Kt), T(near), T(far))
    Vt = E.shape[0]
    r = forward_debug(g.means.flatten(1, 3)[:1], g.covariances.flatten(1, 3)[:1], g.opacities.flatten(1)[:1], view_t, full_t, campos,
                      tanfov, torch.zeros(Vt, 3, device=d), 256, 256, shs=g.harmonics.flatten(1, 3)[:1], sh_degree=4,
                      sh_rgb_major=True, cam_scene=torch.zeros(Vt, dtype=torch.int32, device=d))
    torch.cuda.synchronize()
    return out, r


def _run(V, Vt, dt, oracle_dtype=torch.float32):
    m, W = _model(dt)
    img, K = er.synthetic_input(1, V, 256, 0)
    E, Kt, near, far = chain.config1_targets(Vt, 0.25 if Vt <= 4 else 0.05)
    out, r = _hip_chain(m, img, K, E, Kt, near, far)
    o_out, views, _sc = chain.oracle_chain(W, er.default_cfg(), img, K, E, Kt, near, far, dtype=oracle_dtype)
    cmp_ = chain.compare_renders(r["color"].cpu().numpy(), views)
    tiles = chain.tile_assignment_diff(r["radii"].cpu().numpy(), r["rect"].cpu().numpy(), views)
    pose = float((out["gaussian_camera_extrins"].cpu().double() - o_out["gaussian_camera_extrins"].double()).abs().max())
    cover = float((r["opacity"] > 0.5).float().mean())
    # the yardstick: the reference's OWN CUDA precision (f32 storage, TF32 operands -- backbone_vica.py:9) emulated on the oracle,
    # rendered by the same C rasterizer, against the same f32 / f64 oracle chain
    t_out, t_views, _ = chain.oracle_chain(W, er.default_cfg(), img, K, E, Kt, near, far, operand_mantissa_bits=10)
    ref_cmp = chain.compare_renders(np.stack([v["color"] for v in t_views]), views)
    ref_pose = float((t_out["gaussian_camera_extrins"].double() - o_out["gaussian_camera_extrins"].double()).abs().max())
    print(f"e2e V={V} Vt={Vt} {dt} vs oracle {oracle_dtype}: PSNR(hip, oracle) = {['%.2f' % p for p in cmp_['psnr_between']]} dB "
          f"[TF32-emulated reference vs the same oracle: {['%.2f' % p for p in ref_cmp['psnr_between']]} dB], "
          f"|dPSNR| vs common target = {['%.1e' % p for p in cmp_['dpsnr_common_target']]} [TF32-emulated: "
          f"{['%.1e' % p for p in ref_cmp['dpsnr_common_target']]}], tiles = {tiles}, max|d pose| = {pose:.2e} [TF32-emulated {ref_pose:.2e}], "
          f"coverage = {cover:.2f}")
    return cmp_, ref_cmp, tiles, pose, ref_pose, cover


def _bounds(cmp_, ref_cmp, tiles, pose, ref_pose, cover, max_flip_frac):
    """The 16-bit-operand path (f16: 10-bit mantissa, as TF32) must reproduce the f32 oracle's render at least as well as the
    reference's own TF32 CUDA path does (within ~2 dB: different rounding points give different noise draws of the same size)."""
    assert cover > 0.3, "the synthetic scene must actually be rendered (DESIGN.md 'synthetic scene')"
    # (each rounding scheme -- and each BLAS summation order of the CPU emulation itself -- is another draw of noise of the same size:
    #  the emulated reference scored 19.1-20.2 dB on one host and 19.5-21.2 dB on another for the same scene; the HIP f16 path 18.5-20.2)
    assert min(cmp_["psnr_between"]) >= min(ref_cmp["psnr_between"]) - 2.5, (cmp_, ref_cmp)
    assert float(np.mean(cmp_["psnr_between"])) >= float(np.mean(ref_cmp["psnr_between"])) - 2.0, (cmp_, ref_cmp)
    assert max(cmp_["dpsnr_common_target"]) <= max(2e-2, 3 * max(ref_cmp["dpsnr_common_target"])), (cmp_, ref_cmp)
    assert (tiles["visibility_flips"] + tiles["rect_changes"]) <= max_flip_frac * tiles["gaussian_views"], tiles
    assert abs(tiles["instances_hip"] - tiles["instances_oracle"]) <= 0.01 * tiles["instances_oracle"], tiles
    assert pose <= max(5e-3, 3 * ref_pose)


def test_config1_2view_4targets_f16_vs_oracle_f32():
    """Config 1 / 2: B=1, V=2, 131 072 Gaussians, Vt=4 (identity + x translations).  Measured (round 2): PSNR(HIP f16 chain, f32
    oracle chain) 18.8-20.2 dB; the TF32-emulated reference against the same oracle: 19.1-20.2 dB; f32 vs f64 oracle: 76-77 dB.
    The scene of a random-weight network is per-pixel noise in colour and depth, so sub-pixel shifts of the Gaussians decorrelate
    the render: no TF32-class implementation -- the reference on its CUDA GPU included -- is within 1e-4 dB of an f32 evaluation
    here; the f32-class path (compute dtype 'f32x') is the one held to the tight bound (test below)."""
    _bounds(*_run(2, 4, torch.float16), max_flip_frac=0.12)


def test_config1_2view_4targets_f16_vs_oracle_f64():
    _bounds(*_run(2, 4, torch.float16, torch.float64), max_flip_frac=0.12)


def test_config4_8view_12targets_f16_vs_oracle_f32():
    """Config 4 forward: one 8-view scene, 524 288 Gaussians, 12 target cameras."""
    _bounds(*_run(8, 12, torch.float16), max_flip_frac=0.16)


# ---- the CONDITIONED synthetic checkpoint (vicasplat_amd/synthetic.py conditioned_weights): residual branches damped,
# SH DC tied to the input pixel, Gaussians a few pixels wide and translucent.  On it the render is a smooth function of the network's outputs
# and the render PSNR separates the precision classes by tens of dB, so the 16-bit path gets ABSOLUTE bounds (on the plain random-init
# scene above every TF32-class evaluation, the reference's own included, decorrelates to 19-21 dB and only relative bounds are possible).
# Measured (round 3, HIP vs exact-f32 HIP / oracle): split 90-97 dB, f16 49-50 dB, the TF32-emulated oracle the same class, bf16 33 dB. ----
def _conditioned(V, Vt):
    from vicasplat_amd import synthetic
    shapes = json.load(open(os.path.join(G, "shapes_full.json")))
    W = synthetic.conditioned_weights(shapes, seed=0)
    img, K = synthetic.smooth_input(1, V, 256, 0)
    E, Kt, near, far = chain.config1_targets(Vt, 0.05)
    o_out, views, _ = chain.oracle_chain(W, er.default_cfg(), img, K, E, Kt, near, far)
    return W, img, K, (E, Kt, near, far), o_out, views


def _hip_on(W, dt, img, K, cams):
    from vicasplat_amd.model.encoder import default_cfg, get_encoder
    m, _ = get_encoder(default_cfg())
    m.load_state_dict(W, strict=True)
    m = m.cuda().eval().requires_grad_(False)     # inference: frozen weights -> the fused no-grad path of VicaSplat.forward
    m.set_compute_dtype(dt)
    return _hip_chain(m, img, K, *cams)


def test_conditioned_checkpoint_separates_the_precision_classes():
    W, img, K, cams, o_out, views = _conditioned(2, 3)
    ref_img = np.stack([v["color"] for v in views])
    assert 0.3 <= float(ref_img.mean()) <= 0.7 and float(ref_img.std()) >= 0.05, "the conditioned scene must be an actual image"
    # the yardstick on THIS checkpoint: the reference's CUDA precision (TF32 operands) emulated on the oracle
    t_out, t_views, _ = chain.oracle_chain(W, er.default_cfg(), img, K, *cams, operand_mantissa_bits=10)
    tf32 = chain.compare_renders(np.stack([v["color"] for v in t_views]), views)
    res = {}
    for dt in ("split", torch.float16, torch.bfloat16):
        out, r = _hip_on(W, dt, img, K, cams)
        c = chain.compare_renders(r["color"].cpu().numpy(), views)
        tiles = chain.tile_assignment_diff(r["radii"].cpu().numpy(), r["rect"].cpu().numpy(), views)
        pose = float((out["gaussian_camera_extrins"].cpu().double() - o_out["gaussian_camera_extrins"].double()).abs().max())
        res[str(dt)] = (c, tiles, pose)
        print(f"conditioned {dt}: PSNR vs oracle f32 chain {['%.1f' % p for p in c['psnr_between']]} dB, |dPSNR| {['%.1e' % p for p in c['dpsnr_common_target']]}, "
              f"tile changes {tiles['visibility_flips'] + tiles['rect_changes']} of {tiles['gaussian_views']}, pose {pose:.1e}")
    print(f"conditioned TF32-emulated oracle: PSNR {['%.1f' % p for p in tf32['psnr_between']]} dB, |dPSNR| {['%.1e' % p for p in tf32['dpsnr_common_target']]}")
    c, tiles, pose = res["split"]
    assert min(c["psnr_between"]) >= 80.0 and max(c["dpsnr_common_target"]) <= 1e-4 and pose <= 2e-5
    assert tiles["visibility_flips"] + tiles["rect_changes"] <= 2e-3 * tiles["gaussian_views"]
    c, tiles, pose = res[str(torch.float16)]
    assert min(c["psnr_between"]) >= 44.0, c                                    # absolute: measured 49-50 dB
    assert min(c["psnr_between"]) >= min(tf32["psnr_between"]) - 3.0, (c, tf32)   # and no worse than the reference's own precision class
    assert max(c["dpsnr_common_target"]) <= 5e-3 and pose <= 5e-3
    assert tiles["visibility_flips"] + tiles["rect_changes"] <= 0.05 * tiles["gaussian_views"], tiles
    c, tiles, pose = res[str(torch.bfloat16)]
    assert min(c["psnr_between"]) >= 28.0, c                                    # measured 33 dB: three mantissa bits fewer, ~ -17 dB
