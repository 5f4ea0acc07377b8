"""End-to-end parity of the MI355X encoder (hand-written HIP kernels end to end) against golden vectors produced by the REAL
reference on CPU in float64 (tests/golden/encoder_*.npz).  -m gpu.

Precision policy: GEMM/attention operands f16 with f32 accumulation and f32 residual stream -- the
mantissa the reference's TF32 matmuls keep.  Tolerances below are relative to the per-quantity max magnitude and
were set from the measured error with margin: pose / camera tokens <= 5e-3, raw Gaussian channels <= 3e-2
(36 residual blocks + 2 conv stacks amplify 16-bit operand rounding; the xyz channel is further stretched by expm1)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import encoder_ref as er

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
TINY = dict(enc_depth=2, dec_embed_dim=192, dec_num_heads=3)
LAT = slice(8, 256, 16)


def _model(kind, dt=torch.float16):
    from vicasplat_amd.model.encoder import default_cfg, get_encoder
    shapes = json.load(open(os.path.join(G, f"shapes_{kind}.json")))
    over = dict(TINY) if kind.startswith("tiny") else {}
    if kind == "tiny_noint":
        over["use_intrinsic_embedding"] = False
    m, _ = get_encoder(default_cfg(**over))
    W = er.golden_weights(shapes, seed=0)
    missing = m.load_state_dict(W, strict=True)
    m = m.cuda().eval().requires_grad_(False)     # inference: frozen weights -> the fused no-grad path of VicaSplat.forward
    m.set_compute_dtype(dt)
    return m


def _rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64).reshape(a.shape)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


# f16 bounds = 1.5 x the largest value measured over the five golden configurations (round 3, vs the reference's float64 run: pose 1.1e-3 /
# c2w 1.9e-3, raw and adapter quantities <= 5.2e-3, covariances <= 1.35e-2, per-block drift <= 4.0e-3): (pose, raw, block, covariances)
F16_TOL = (3e-3, 8e-3, 6e-3, 2e-2)
PROBE = [0, 1, 7, 100, 1000, 5000, 20000, 50000]      # flat indices of the goldens' per-block checksums (gen_encoder_golden.py)


def block_drift(z, probes):
    """Per-block distance to the reference's float64 block outputs (goldens' f64_blocks: mean, mean |x|, 8 probed elements of every
    encoder / decoder block's output stream), relative to the block's mean |x|: localises a drift to the block where it starts."""
    names = [str(n) for n in z["f64_block_names"]]
    worst = {}
    for n, row in zip(names, z["f64_blocks"]):
        t = probes[n].double().flatten()
        got = torch.cat([t.mean()[None], t.abs().mean()[None], t[torch.tensor([p % t.numel() for p in PROBE], device=t.device)]]).cpu().numpy()
        worst[n] = float(np.abs(got - row).max() / row[1])
    return worst


def _check(name, dt, tol_pose, tol_raw, tol_block=None, tol_cov=None):
    z = np.load(os.path.join(G, f"encoder_{name}.npz"))
    kind = "tiny_noint" if name.startswith("tiny_noint") else "tiny" if name.startswith("tiny") else "full"
    m = _model(kind, dt)
    B, V = int(z["cfg_B"]), int(z["cfg_V"])
    img, K = er.synthetic_input(B, V, 256, int(z["cfg_seed"]))
    probes = {}
    m.backbone._probe = lambda n, t: probes.__setitem__(n, t.detach().clone())
    out = m(dict(image=img.cuda(), intrinsics=K.cuda()), compute_viewspace_depth=False)
    m.backbone._probe = None
    torch.cuda.synchronize()
    if "f64_blocks" in z.files:
        drift = block_drift(z, probes)
        first_bad = next((n for n in sorted(drift) if tol_block is not None and drift[n] > tol_block), None)
        print(name, dt, "block drift: max %.2e at %s; enc23 %.2e, last dec img %.2e cam %.2e" % (
            max(drift.values()), max(drift, key=drift.get), drift.get("enc23", drift.get("enc01", 0.0)), drift[sorted(k for k in drift if k.endswith("_img"))[-1]],
            drift[sorted(k for k in drift if k.endswith("_cam"))[-1]]))
        assert first_bad is None, (first_bad, drift[first_bad], tol_block)
    errs = dict(pose=_rel(out["pred_extrins"].cpu(), z["f64_pred_extrins"]), c2w=_rel(out["gaussian_camera_extrins"].cpu(), z["f64_c2w"]))
    raw = out["raw_gaussians"][:, :, LAT, LAT].cpu().numpy()
    for nm, sl in (("xyz", slice(0, 3)), ("opacity", slice(3, 4)), ("scale", slice(4, 7)), ("quat", slice(7, 11)), ("sh", slice(11, 86))):
        errs[nm] = _rel(raw[..., sl], z["f64_raw"][..., sl])
    g = out["gaussians"]
    for k in ("means", "covariances", "harmonics", "opacities"):
        errs["g_" + k] = _rel(getattr(g, k)[:, :, LAT, LAT].cpu().numpy(), z[f"f64_{k}"])
    print(name, dt, {k: f"{v:.2e}" for k, v in errs.items()})
    assert errs["pose"] <= tol_pose and errs["c2w"] <= tol_pose, errs
    for k in ("xyz", "opacity", "scale", "quat", "sh", "g_means", "g_harmonics", "g_opacities"):
        assert errs[k] <= tol_raw, (k, errs)
    assert errs["g_covariances"] <= (tol_cov if tol_cov is not None else tol_raw), errs      # (R S)(R S)^T squares the scale error
    assert out["raw_gaussians"].shape == (B, V, 256, 256, 86) and g.covariances.shape == (B, V, 256, 256, 3, 3)
    if kind == "tiny_noint":
        e_fov = _rel(out["pred_intrins"].cpu(), z["f64_pred_intrins"])
        e_K = _rel(out["gaussian_camera_intrins"].cpu(), z["f64_intrins_3x3"])
        print("fov", e_fov, "K", e_K)
        assert e_fov <= tol_pose and e_K <= 10 * tol_pose, (e_fov, e_K)      # K = 0.5 / tan(fov / 2) amplifies a small fov's error
    else:
        assert out["pred_intrins"] is None and out["gaussian_camera_intrins"] is None
    return errs


def test_state_dict_is_the_reference_abi():
    from vicasplat_amd.model.encoder import default_cfg, get_encoder
    shapes = json.load(open(os.path.join(G, "shapes_full.json")))
    m, vis = get_encoder(default_cfg())
    sd = m.state_dict()
    assert vis is None and set(sd) == set(shapes) and all(tuple(sd[k].shape) == tuple(shapes[k]) for k in sd)
    assert m.backbone.config.use_intrinsic_embedding is True


@pytest.mark.parametrize("name", ["tiny_v2", "tiny_v3"])
def test_encoder_tiny_matches_reference(name):
    _check(name, torch.float16, *F16_TOL)


def test_encoder_without_intrinsic_embedding_matches_reference():
    """use_intrinsic_embedding=false (the released *_no_intrin checkpoints, README.md:51-53): 256 tokens per frame, global scope for
    camera token 0, fov head; goldens from the real reference built with that flag."""
    _check("tiny_noint_v3", torch.float16, *F16_TOL)


def test_encoder_full_vitl_2view_matches_reference():
    _check("full_v2", torch.float16, *F16_TOL)


def test_encoder_full_vitl_8view_matches_reference():
    """BASELINE.json bench configuration: 8 context views, ViT-L, 524 288 Gaussians."""
    _check("full_v8", torch.float16, *F16_TOL)


def test_encoder_full_vitl_2view_batch16_matches_reference():
    """BASELINE.json config 2 at the configuration's batch size (re10k_2view: 16 scenes per step): the golden scene rides in slot 5 of
    a 16-scene batch (15 other seeds around it) and must match the reference's float64 outputs at the same bounds as alone -- the
    batched launches (other GEMM tile routes, 32 frames per attention launch) may not leak between scenes."""
    z = np.load(os.path.join(G, "encoder_full_v2.npz"))
    assert int(z["cfg_B"]) == 1 and int(z["cfg_V"]) == 2
    m = _model("full")
    img0, K0 = er.synthetic_input(1, 2, 256, int(z["cfg_seed"]))
    img, K = er.synthetic_input(16, 2, 256, 1234)
    img[5], K[5] = img0[0], K0[0]
    out = m(dict(image=img.cuda(), intrinsics=K.cuda()), compute_viewspace_depth=False)
    torch.cuda.synchronize()
    assert out["raw_gaussians"].shape == (16, 2, 256, 256, 86)
    errs = dict(pose=_rel(out["pred_extrins"][5:6].cpu(), z["f64_pred_extrins"]))
    raw = out["raw_gaussians"][5:6, :, LAT, LAT].cpu().numpy()
    for nm, sl in (("xyz", slice(0, 3)), ("opacity", slice(3, 4)), ("scale", slice(4, 7)), ("quat", slice(7, 11)), ("sh", slice(11, 86))):
        errs[nm] = _rel(raw[..., sl], z["f64_raw"][..., sl])
    g = out["gaussians"]
    for k in ("means", "covariances", "harmonics", "opacities"):
        errs["g_" + k] = _rel(getattr(g, k)[5:6, :, LAT, LAT].cpu().numpy(), z[f"f64_{k}"])
    print({k: f"{v:.2e}" for k, v in errs.items()})
    assert errs["pose"] <= F16_TOL[0], errs
    assert all(errs[k] <= F16_TOL[1] for k in errs if k not in ("pose", "g_covariances")) and errs["g_covariances"] <= F16_TOL[3], errs
    assert bool(torch.isfinite(out["raw_gaussians"]).all()) and bool(torch.isfinite(out["pred_extrins"]).all())


def test_encoder_bf16_path_runs():
    _check("tiny_v2", torch.bfloat16, 3e-2, 2e-1)


def test_scenes_are_independent_so_sharding_is_exact():
    """The multi-GPU path shards the scene batch with no collective: a scene must come out the same
    (to operand-rounding noise) whether it is encoded alone or inside a larger batch."""
    m = _model("full")
    img, K = er.synthetic_input(3, 2, 256, 11)
    both = m(dict(image=img.cuda(), intrinsics=K.cuda()), compute_viewspace_depth=False)
    for i in (0, 2):
        one = m(dict(image=img[i:i + 1].cuda(), intrinsics=K[i:i + 1].cuda()), compute_viewspace_depth=False)
        e = (_rel(one["pred_extrins"].cpu(), both["pred_extrins"][i:i + 1].cpu()),
             _rel(one["raw_gaussians"].cpu(), both["raw_gaussians"][i:i + 1].cpu()),
             _rel(one["gaussians"].means.cpu(), both["gaussians"].means[i:i + 1].cpu()))
        print("scene", i, "alone vs in batch:", ["%.2e" % v for v in e])
        # not bitwise: batch size picks different GEMM kernels (tile vs small-M / tail) whose f32 sums are ordered
        # differently, and a last-bit difference can flip a 16-bit operand rounding that 36 blocks then amplify; the result
        # stays inside the 16-bit-operand noise floor measured against the reference goldens (pose 1e-3, raw 5e-3)
        assert e[0] <= 1e-3 and e[1] <= 1e-2 and e[2] <= 1e-2, e


def test_deterministic_mode_is_bit_reproducible(monkeypatch):
    """N replicas of a scene shard must be bit-identical to one replica.  The default split-class forward is reproducible
    only to rounding: the f32 residual epilogue of GEMMs that underfill the chip (tails, skinny launches) splits K over workgroups whose
    partial sums meet through f32 atomics (csrc/gemm_common.h), so the last bit of a residual stream follows the arrival order and a few
    hundred of 2e8 tile instances flip downstream.  VS_DETERMINISTIC=1 runs those GEMMs unsplit (+1.3 % on the
    24-scene step, nothing at B = 1): (i) two runs of the full ViT-L split forward are torch.equal in every output; (ii) the rasterizer's
    instance count and image repeat exactly.  NOT guaranteed, and bounded here instead: scene 0 alone vs scene 0 inside a batch of 3 --
    the kernel a row is routed to (tile shape, LayerNorm / attention variant) depends on the batch's row count and the variants sum in
    different orders (forcing one GEMM tile shape alone, VS_GEMM_MI=4, does not remove it: tools/batch_invariance3.py); replicas of EQUAL
    shards, which is what scene sharding produces, take identical routes."""
    from vicasplat_amd.model.decoder.cuda_splatting import render_cuda
    from vicasplat_amd import raster
    monkeypatch.setenv("VS_DETERMINISTIC", "1")
    m = _model("full", "split")
    img, K = er.synthetic_input(3, 8, 256, 5)
    ctx = dict(image=img.cuda(), intrinsics=K.cuda())
    keys = ("pred_extrins", "raw_gaussians", "gaussian_camera_extrins")

    def run(c):
        o = m(c, compute_viewspace_depth=False)
        g = o["gaussians"]
        return {**{k: o[k].clone() for k in keys}, **{k: getattr(g, k).clone() for k in ("means", "covariances", "harmonics", "opacities")}}

    a, b = run(ctx), run(ctx)
    for k in a:
        assert torch.equal(a[k], b[k]), ("run to run", k, float((a[k] - b[k]).abs().max()))
    one = run(dict(image=img[:1].cuda(), intrinsics=K[:1].cuda()))
    for k in a:      # rounding-level only (measured: pose 3e-7 abs, raw 1e-4 abs at scale ~30, covariances 2e-5 of their largest element)
        d_ = float((one[k][0] - a[k][0]).abs().max()) / (float(a[k][0].abs().max()) + 1e-12)
        assert d_ <= 2e-4, ("scene 0 alone vs in a batch of 3", k, d_)
    # the rasterizer on top: identical instance counts and images, run to run
    d = torch.device("cuda:0")
    E = torch.eye(4, device=d).repeat(4, 1, 1); E[:, 0, 3] = torch.arange(4, device=d) * 0.05
    Kt = torch.tensor([[0.9, 0, 0.5], [0, 0.9, 0.5], [0, 0, 1.0]], device=d).repeat(4, 1, 1)
    near, far = torch.full((4,), 0.01, device=d), torch.full((4,), 100.0, device=d)
    outs = []
    for r in (a, b):
        col, _ = render_cuda(E, Kt, near, far, (256, 256), torch.zeros(4, 3, device=d), r["means"].flatten(1, 3)[0], r["covariances"].flatten(1, 3)[0],
                             r["harmonics"].flatten(1, 3)[0], r["opacities"].flatten(1)[0])
        outs.append((col.clone(), raster.last_call()["num_rendered"]))
    assert outs[0][1] == outs[1][1] and torch.equal(outs[0][0], outs[1][0])


def test_no_cpu_fallback():
    from vicasplat_amd.model.encoder import default_cfg, get_encoder
    m, _ = get_encoder(default_cfg(**TINY))
    img, K = er.synthetic_input(1, 2, 256, 0)
    with pytest.raises(RuntimeError):
        m(dict(image=img, intrinsics=K), compute_viewspace_depth=False)


# f16 on the reference's example frames: 1.5 x the largest value measured over the two scenes in round 4 -- (pose / c2w, raw and adapter
# quantities, normalised rotations, covariances); measured: golden pose 1.2e-3 / c2w 1.6e-3, xyz 8.5e-3 (synthetic input: 5.2e-3), rotations
# 9.6e-3, covariances 5.9e-3; conditioned checkpoint: c2w 1.2e-3, xyz 4.3e-3, rotations 6.1e-3, covariances 6.2e-4
F16_EXAMPLES_TOL = dict(golden=(2.5e-3, 1.3e-2, 1.5e-2, 9e-3), cond=(1.8e-3, 6.5e-3, 9.2e-3, 1e-3))


@pytest.mark.parametrize("wname", ["golden", "cond"])
def test_encoder_f16_on_the_reference_example_frames(wname):
    """The 16-bit class on REAL frames (fixture and protocol: tests/test_split_path_gpu.py::
    test_encoder_split_on_the_reference_example_frames): same bounds as on the synthetic inputs -- pose, raw / adapter quantities,
    covariances; real-image statistics (flat regions, 6.6 % saturated values) cost the f16 class up to 1.6x on the centres (8.5e-3 against
    5.2e-3 on the synthetic input), nothing on the poses."""
    from test_split_path_gpu import _example_model, example_frames_errors
    z = np.load(os.path.join(G, "encoder_full_v8_examples.npz"))
    m = _example_model(wname, torch.float16)
    tol_pose, tol_raw, tol_rot, tol_cov = F16_EXAMPLES_TOL[wname]
    for si in range(2):
        errs, _ = example_frames_errors(m, z, f"{wname}_s{si}", si)
        print(wname, "scene", si, "f16 vs reference f64:", {k: f"{v:.1e}" for k, v in errs.items()})
        assert errs["pose"] <= tol_pose and errs["c2w"] <= tol_pose, errs
        assert all(errs[k] <= tol_raw for k in ("xyz", "opacity", "scale", "quat", "sh", "g_opacities", "g_scales")), errs
        assert errs["g_rotations"] <= tol_rot and errs["g_covariances"] <= tol_cov, errs
