"""Training groundwork: one ViT-L frame-encoder block differentiated on the HIP backward operators vs torch autograd over
the oracle's restatement of the same block (oracle/encoder_ref.py: ln / lin / rope2d / sdpa / mlp)."""
import math

import pytest
import os
import torch

from oracle import encoder_ref as er

pytestmark = pytest.mark.gpu
F16_GRAD_TOL = (8e-2, 2e-2, 3e-2, 2e-2)      # (lattice, sum, |sum|, norm) of the f16-class gradient-golden test = 3x the measured 2.6e-2 / 5.8e-3 / 9.1e-3 / 6.6e-3


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_encoder_block_backward_matches_oracle_autograd(dt):
    from vicasplat_amd import train
    d = torch.device("cuda:0")
    frames, gh, C, H = 2, 16, 1024, 16
    tokens = gh * gh + 1
    M = frames * tokens
    g = torch.Generator().manual_seed(5)
    n = "backbone.enc_blocks.0"
    W = {}
    def mk(name, shape, std):
        W[name] = torch.randn(*shape, generator=g) * std
    mk(n + ".norm1.weight", (C,), 0.1); W[n + ".norm1.weight"] += 1
    mk(n + ".norm1.bias", (C,), 0.1)
    mk(n + ".norm2.weight", (C,), 0.1); W[n + ".norm2.weight"] += 1
    mk(n + ".norm2.bias", (C,), 0.1)
    for nm, shp in ((".attn.qkv", (3 * C, C)), (".attn.proj", (C, C)), (".mlp.fc1", (4 * C, C)), (".mlp.fc2", (C, 4 * C))):
        mk(n + nm + ".weight", shp, 1.0 / math.sqrt(shp[1])); mk(n + nm + ".bias", (shp[0],), 0.1)
    x0 = torch.randn(frames, tokens, C, generator=g)
    dy = torch.randn(frames, tokens, C, generator=g) * 0.1
    pos = er.patch_positions(frames, gh, gh)
    extra = pos[:, :1].clone(); extra[:, :, 0] += pos[:, -1:, 0] + 1
    pos = torch.cat([pos, extra], 1)                                                   # [frames, tokens, 2]

    # ---- reference: f32 autograd over the oracle's block, with the weights rounded to the operand dtype ----
    Wr = {k: (v.to(dt).float() if k.endswith("weight") and v.dim() == 2 else v.clone()).requires_grad_() for k, v in W.items()}
    xr = x0.clone().requires_grad_()
    h = er.ln(Wr, n + ".norm1", xr)
    qkv = er.lin(Wr, n + ".attn.qkv", h).reshape(frames, -1, 3, H, C // H).permute(2, 0, 3, 1, 4)
    q, k, v = er.rope2d(qkv[0], pos, 100.0), er.rope2d(qkv[1], pos, 100.0), qkv[2]
    xm = xr + er.lin(Wr, n + ".attn.proj", er.heads_merge(er.sdpa(q, k, v)))
    xo = xm + er.mlp(Wr, n + ".mlp", er.ln(Wr, n + ".norm2", xm))
    (xo * dy).sum().backward()

    # ---- ours ----
    f = lambda name: W[name].to(d)
    p = train.EncBlockParams(ln1_w=f(n + ".norm1.weight"), ln1_b=f(n + ".norm1.bias"), qkv_w=f(n + ".attn.qkv.weight").to(dt),
                             qkv_b=f(n + ".attn.qkv.bias"), proj_w=f(n + ".attn.proj.weight").to(dt), proj_b=f(n + ".attn.proj.bias"),
                             ln2_w=f(n + ".norm2.weight"), ln2_b=f(n + ".norm2.bias"), fc1_w=f(n + ".mlp.fc1.weight").to(dt),
                             fc1_b=f(n + ".mlp.fc1.bias"), fc2_w=f(n + ".mlp.fc2.weight").to(dt), fc2_b=f(n + ".mlp.fc2.bias"))
    pos_d = pos.reshape(M, 2).to(torch.int32).contiguous().to(d)
    x_out, tape = train.enc_block_forward_train(x0.reshape(M, C).to(d), p, pos_d, frames=frames, tokens=tokens, heads=H)
    rel = lambda a, b: float((a.detach().cpu().float() - b).abs().max() / (b.abs().max() + 1e-12))
    ft = 4e-3 if dt == torch.float16 else 3e-2
    assert rel(x_out, xo.detach().reshape(M, C)) <= ft
    dx_in, grads = train.enc_block_backward(dy.reshape(M, C).to(d), tape, p)
    gt = 1.5e-2 if dt == torch.float16 else 8e-2
    errs = {"dx": rel(dx_in, xr.grad.reshape(M, C))}
    for ours, ref in (("ln1_w", ".norm1.weight"), ("ln1_b", ".norm1.bias"), ("qkv_w", ".attn.qkv.weight"), ("qkv_b", ".attn.qkv.bias"),
                      ("proj_w", ".attn.proj.weight"), ("proj_b", ".attn.proj.bias"), ("ln2_w", ".norm2.weight"), ("ln2_b", ".norm2.bias"),
                      ("fc1_w", ".mlp.fc1.weight"), ("fc1_b", ".mlp.fc1.bias"), ("fc2_w", ".mlp.fc2.weight"), ("fc2_b", ".mlp.fc2.bias")):
        errs[ours] = rel(grads[ours], Wr[n + ref].grad)
    print(dt, {k: f"{v:.2e}" for k, v in errs.items()})
    assert max(errs.values()) <= gt, errs


TINY = dict(enc_depth=2, dec_embed_dim=192, dec_num_heads=3)


def _tiny_model(dt):
    import json, os
    from vicasplat_amd.model.encoder import default_cfg, get_encoder
    G = os.path.join(os.path.dirname(__file__), "golden")
    shapes = json.load(open(os.path.join(G, "shapes_tiny.json")))
    m, _ = get_encoder(default_cfg(**TINY))
    W = er.golden_weights(shapes, seed=0)
    m.load_state_dict(W, strict=True)
    m = m.cuda().train()
    m.set_compute_dtype(dt)
    return m, W


def test_training_forward_matches_reference_goldens():
    """The differentiable forward (autograd Functions on the HIP kernels) reproduces the REAL reference's f64 outputs like
    the inference forward does (same goldens, same tolerances as tests/test_encoder_gpu.py)."""
    import os
    import numpy as np
    from vicasplat_amd.model.encoder.train_forward import forward_train
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "encoder_tiny_v3.npz"))
    m, _ = _tiny_model(torch.float16)
    B, V = int(z["cfg_B"]), int(z["cfg_V"])
    img, K = er.synthetic_input(B, V, 256, int(z["cfg_seed"]))
    with torch.no_grad():
        out = forward_train(m, img.cuda(), K.cuda(), torch.float16)
    rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64).reshape(np.shape(a))).max() / (np.abs(b).max() + 1e-12))
    LAT = slice(8, 256, 16)
    e_pose = rel(out["pred_extrins"].cpu().numpy(), z["f64_pred_extrins"])
    e_raw = rel(out["raw_gaussians"][:, :, LAT, LAT].cpu().numpy(), z["f64_raw"])
    e_cov = rel(out["gaussians"]["covariances"][:, :, LAT, LAT].cpu().numpy(), z["f64_covariances"])
    print("train forward vs reference f64:", e_pose, e_raw, e_cov)
    assert e_pose <= 5e-3 and e_raw <= 3e-2 and e_cov <= 6e-2


@pytest.mark.parametrize("name", ["full_v2", "full_v8"])
def test_training_forward_full_vitl_matches_reference_goldens(name):
    """The differentiable forward at FULL size (ViT-L, 2 and 8 context views = BASELINE's training configuration) against the real
    reference's f64 goldens, same bounds as the inference forward (tests/test_encoder_gpu.py)."""
    import json, os
    import numpy as np
    from vicasplat_amd.model.encoder import default_cfg, get_encoder
    from vicasplat_amd.model.encoder.train_forward import forward_train
    G = os.path.join(os.path.dirname(__file__), "golden")
    z = np.load(os.path.join(G, f"encoder_{name}.npz"))
    m, _ = get_encoder(default_cfg())
    m.load_state_dict(er.golden_weights(json.load(open(os.path.join(G, "shapes_full.json"))), seed=0), strict=True)
    m = m.cuda().train()
    B, V = int(z["cfg_B"]), int(z["cfg_V"])
    img, K = er.synthetic_input(B, V, 256, int(z["cfg_seed"]))
    with torch.no_grad():
        out = forward_train(m, img.cuda(), K.cuda(), torch.float16)
    rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64).reshape(np.shape(a))).max() / (np.abs(b).max() + 1e-12))
    LAT = slice(8, 256, 16)
    errs = dict(pose=rel(out["pred_extrins"].cpu().numpy(), z["f64_pred_extrins"]))
    raw = out["raw_gaussians"][:, :, LAT, LAT].cpu().numpy()
    for nm, sl in (("xyz", slice(0, 3)), ("opacity", slice(3, 4)), ("scale", slice(4, 7)), ("quat", slice(7, 11)), ("sh", slice(11, 86))):
        errs[nm] = rel(raw[..., sl], z["f64_raw"][..., sl])
    g = out["gaussians"]
    for k in ("means", "harmonics", "opacities"):
        errs["g_" + k] = rel(g[k][:, :, LAT, LAT].cpu().numpy(), z[f"f64_{k}"])
    print("training forward", name, {k: f"{v:.2e}" for k, v in errs.items()})
    assert errs["pose"] <= 5e-3, errs
    assert max(v for k, v in errs.items() if k != "pose") <= 3e-2, errs
    assert out["raw_gaussians"].shape == (B, V, 256, 256, 86) and g["covariances"].shape == (B, V, 256, 256, 3, 3)


def test_training_backward_matches_oracle_autograd():
    """End-to-end gradients of the whole encoder (2 + 12 transformer blocks, both DPT heads, adapter, pose head) for a
    random linear functional of its outputs, HIP training path vs f32 torch autograd over the oracle (128 x 128 frames)."""
    from vicasplat_amd.model.encoder.train_forward import forward_train
    dt = torch.float16
    m, W = _tiny_model(dt)
    B, V, S = 1, 3, 128
    img, K = er.synthetic_input(B, V, S, 7)
    g = torch.Generator().manual_seed(1)
    r_raw = torch.randn(B, V, S, S, 86, generator=g) * 1e-3
    r_raw[..., :3] *= 0.1
    r_pose = torch.randn(B, V - 1, 8, generator=g)
    r_cov = torch.randn(B, V, S, S, 3, 3, generator=g) * 10.0
    # ---- oracle (f32 CPU autograd) ----
    Wr = {k: v.clone().float().requires_grad_() for k, v in W.items()}
    cfg = er.default_cfg(**TINY)
    o = er.forward.__wrapped__(Wr, cfg, img, K)
    loss_r = (o["raw_gaussians"] * r_raw).sum() + (o["pred_extrins"] * r_pose).sum() + (o["gaussians"]["covariances"] * r_cov).sum()
    loss_r.backward()
    # ---- ours ----
    out = forward_train(m, img.cuda(), K.cuda(), dt)
    loss = (out["raw_gaussians"] * r_raw.cuda()).sum() + (out["pred_extrins"] * r_pose.cuda()).sum() + \
        (out["gaussians"]["covariances"] * r_cov.cuda()).sum()
    S = 1024.0                       # static loss scale: 16-bit activation gradients would underflow otherwise (torch.amp practice)
    fe = lambda a, b: float((a.detach().cpu().float() - b.detach()).abs().max() / b.detach().abs().max())
    print("forward rel err: raw %.3e pose %.3e cov %.3e" % (fe(out["raw_gaussians"], o["raw_gaussians"]), fe(out["pred_extrins"], o["pred_extrins"]),
                                                          fe(out["gaussians"]["covariances"], o["gaussians"]["covariances"])))
    (loss * S).backward()
    assert abs(float(loss.detach()) - float(loss_r.detach())) <= 2e-2 * abs(float(loss_r.detach())) + 1e-2, (float(loss.detach()), float(loss_r.detach()))
    errs = {}
    for name, p in m.named_parameters():
        import re
        ref = Wr[re.sub(r"layer(\d)_rn", lambda mm: f"layer_rn.{int(mm.group(1)) - 1}", name)].grad   # aliased keys
        if ref is None:                      # unused by the model (e.g. refinenet4.resConfUnit1: no skip input)
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None, name
        errs[name] = float((p.grad.cpu().float() / S - ref).abs().max() / (ref.abs().max() + 1e-20))
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:8]
    vals = sorted(errs.values())
    print("params", len(errs), "median rel err %.2e" % vals[len(vals) // 2], "p90 %.2e" % vals[int(len(vals) * 0.9)], "worst", worst)
    # The residual error is the forward's 16-bit-operand noise (activations entering every weight gradient differ from the f32
    # oracle's by the same ~1-3 % that the forward parity tests allow on this tiny random-weight model), not a backward defect:
    # the backward operators alone are exact to 1e-6..7e-4 (tests/test_ops_gpu.py, the block test above).  What training
    # needs is the gradient DIRECTION: cosine similarity of the whole gradient and of every parameter's gradient.
    ours = torch.cat([p.grad.flatten().cpu().float() / S for n_, p in m.named_parameters() if n_ in errs])
    import re as _re
    refv = torch.cat([Wr[_re.sub(r"layer(\d)_rn", lambda mm: f"layer_rn.{int(mm.group(1)) - 1}", n_)].grad.flatten() for n_, p in m.named_parameters() if n_ in errs])
    cos_all = float((ours.double() @ refv.double()) / (ours.double().norm() * refv.double().norm()))
    cos_min, cos_arg = 1.0, None
    for n_, p in m.named_parameters():
        if n_ in errs:
            r_ = Wr[_re.sub(r"layer(\d)_rn", lambda mm: f"layer_rn.{int(mm.group(1)) - 1}", n_)].grad.flatten()
            o_ = p.grad.flatten().cpu().double()
            c_ = float((o_ @ r_.double()) / (o_.norm() * r_.double().norm() + 1e-300))
            if c_ < cos_min:
                cos_min, cos_arg = c_, n_
    print("cosine(all) %.6f  min per-parameter cosine %.5f (%s)" % (cos_all, cos_min, cos_arg))
    # (One more noise mechanism, seen when the RoPE moved into the GEMM epilogues: the pose head is ReLU -> Linear on 2 camera tokens
    # of this tiny model; a forward difference of 1e-3 flips the sign of a near-zero activation, a whole gradient element appears or
    # disappears, the f32 LayerNorm backward spreads it over the camera stream as a constant offset, and the max-norm error of the
    # small camera-path gradients jumps to 0.6 while their first elements agree to 4 digits.  Hence the slack below; a defect in a
    # backward operator shows up as a cosine far below these bounds for the parameters behind it.)
    assert cos_all >= 0.998 and cos_min >= 0.97, (cos_all, cos_min, cos_arg)
    assert vals[len(vals) // 2] <= 6e-2 and vals[int(len(vals) * 0.9)] <= 0.2 and vals[-1] <= 0.7, worst


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_training_step_reduces_the_photometric_loss(dt):
    """callers.training_step: encoder (HIP fwd+bwd) -> rasterizer (HIP fwd+bwd) -> MSE -> clip -> AdamW.  A few steps on one
    fixed batch must lower the loss; every parameter that the loss reaches gets a finite gradient."""
    from vicasplat_amd import callers
    from vicasplat_amd.model.decoder import DecoderSplattingCUDACfg, get_decoder
    import bench
    m, _ = _tiny_model(dt)
    d = torch.device("cuda:0")
    dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], False)).to(d)
    B, V, Vt, S = 1, 2, 2, 64
    img, K = er.synthetic_input(B, V, S, 3)
    tE, tK, tn, tf = bench.target_cameras(B, Vt, d)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, S), torch.linspace(0, 1, S), indexing="ij")
    target = torch.stack([0.5 + 0.4 * torch.sin(6 * xx), 0.5 + 0.4 * torch.cos(5 * yy), 0.5 * (xx + yy)], 0)[None, None].expand(B, Vt, 3, S, S).contiguous().to(d)
    batch = dict(context=dict(image=img.to(d), intrinsics=K.to(d)),
                 target=dict(image=target, extrinsics=tE, intrinsics=tK, near=tn, far=tf))
    opt, _ = callers.configure_optimizer(m, lr=4e-5, backbone_lr_multiplier=0.25)      # the reference's learning rates
    before = {n: p.detach().clone() for n, p in list(m.named_parameters())[:40]}
    hist = []
    # (Adam's first updates are sign steps of size lr on every weight: the loss jumps up by ~20 % after step 1 and then comes down along
    # a trajectory that differs run to run -- f32 atomics in the rasterizer backward -- so the check looks at the best of the last steps)
    for it in range(16):
        r = callers.training_step(m, dec, batch, opt, compute_dtype=dt)
        assert not r["skipped"] and torch.isfinite(r["loss"]) and torch.isfinite(r["grad_norm"]), r
        hist.append(float(r["loss"]))
    print("loss", ["%.5f" % v for v in hist], "grad_norm", float(r["grad_norm"]), "psnr", float(r["psnr"]))
    assert min(hist[-6:]) < hist[0] * 0.95 and float(r["grad_norm"]) > 0, hist
    changed = sum(int(not torch.equal(before[n], p.detach())) for n, p in list(m.named_parameters())[:40])
    assert changed >= 30


def test_training_step_with_gradient_allreduce_single_rank():
    """The DDP leg of the training step (bucketed all-reduce over RCCL, `allreduce=True`) on a world of one: the collective
    path executes on the GPU and leaves the step unchanged.  (World size 2 is covered on CPU/gloo in test_distributed_cpu.py.)"""
    import torch.distributed as dist
    from vicasplat_amd import callers
    from vicasplat_amd.model.decoder import DecoderSplattingCUDACfg, get_decoder
    import bench
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        dt = torch.float16
        d = torch.device("cuda:0")
        dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], False)).to(d)
        B, V, Vt, S = 1, 2, 2, 64
        img, K = er.synthetic_input(B, V, S, 3)
        tE, tK, tn, tf = bench.target_cameras(B, Vt, d)
        target = torch.rand(B, Vt, 3, S, S, device=d)
        batch = dict(context=dict(image=img.to(d), intrinsics=K.to(d)), target=dict(image=target, extrinsics=tE, intrinsics=tK, near=tn, far=tf))
        res = []
        for flag in (False, True):
            m, _ = _tiny_model(dt)
            opt, _ = callers.configure_optimizer(m, lr=4e-5)
            r = callers.training_step(m, dec, batch, opt, compute_dtype=dt, allreduce=flag)
            res.append((float(r["loss"]), float(r["grad_norm"])))
        assert abs(res[0][0] - res[1][0]) <= 1e-6 and abs(res[0][1] - res[1][1]) <= 1e-3 * res[0][1], res
    finally:
        dist.destroy_process_group()


def _full_model(dt=torch.float16):
    import json, os
    from vicasplat_amd.model.encoder import default_cfg, get_encoder
    G = os.path.join(os.path.dirname(__file__), "golden")
    shapes = json.load(open(os.path.join(G, "shapes_full.json")))
    m, _ = get_encoder(default_cfg())
    m.load_state_dict(er.golden_weights(shapes, seed=0), strict=True)
    m = m.cuda().train()
    m.set_compute_dtype(dt)
    return m


def _config4_batch(B, V, Vt, d, with_extrinsics=False):
    import bench
    img, K = er.synthetic_input(B, V, 256, 0)
    tE, tK, tn, tf = bench.target_cameras(B, Vt, d)
    g = torch.Generator().manual_seed(9)
    target = torch.rand(B, Vt, 3, 256, 256, generator=g).to(d)
    ctx = dict(image=img.to(d), intrinsics=K.to(d))
    if with_extrinsics:
        E = torch.eye(4).repeat(B, V, 1, 1)
        E[:, :, 0, 3] = 0.1 * torch.arange(V)[None]
        ctx["extrinsics"] = E.to(d)
    return dict(context=ctx, target=dict(image=target, extrinsics=tE, intrinsics=tK, near=tn, far=tf))


@pytest.mark.parametrize("cdt", ["split", "f16"])
def test_full_vit_l_backward_matches_the_real_references_gradient_goldens(cdt):
    """config 4's backward pinned to the REAL reference, not only to property checks.  tests/golden/
    encoder_full_v2_grads.npz holds float64 gradients of the imported reference encoder (full ViT-L: 24 + 12 blocks, both DPT heads, adapter,
    pose head; 1 scene x 2 views x 256 x 256) for a seeded linear functional of raw_gaussians, pred_extrins and the covariances
    (gen_encoder_grad_golden.py).  The split-class training path (f32 activations and gradients, three f16 MFMAs per product, forward and
    backward) runs the same functional; for 41 parameters spread over every part of the model the gradient's lattice elements, its sum
    and its sum of magnitudes must agree with the reference's, relative to the gradient's largest element."""
    import json
    import numpy as np
    from vicasplat_amd.model.encoder.train_forward import forward_train
    G = os.path.join(os.path.dirname(__file__), "golden")
    z = np.load(os.path.join(G, "encoder_full_v2_grads.npz"))
    B, V = int(z["cfg_B"]), int(z["cfg_V"])
    dt = "split" if cdt == "split" else torch.float16
    m = _full_model(dt).eval()               # eval: the heads' Dropout(0.1) is the identity, as in the generator
    m.requires_grad_(True)
    img, K = er.synthetic_input(B, V, 256, int(z["cfg_seed"]))
    g = torch.Generator().manual_seed(1)     # the functional of gen_encoder_grad_golden.functional()
    r_raw = torch.randn(B, V, 256, 256, 86, generator=g) * 1e-3
    r_raw[..., :3] *= 0.1
    r_pose = torch.randn(B, V - 1, 8, generator=g)
    r_cov = torch.randn(B, V, 256, 256, 3, 3, generator=g) * 10.0
    out = forward_train(m, img.cuda(), K.cuda(), dt)
    loss = (out["raw_gaussians"] * r_raw.cuda()).sum() + (out["pred_extrins"] * r_pose.cuda()).sum() + (out["gaussians"]["covariances"] * r_cov.cuda()).sum()
    S = 4096.0 if cdt == "split" else 1024.0  # power-of-two scale: keeps the (hi, lo) operand pairs / the 16-bit gradients of the backward in f16's normal range
    (loss * S).backward()
    torch.cuda.synchronize()
    ltol = 2e-4 if cdt == "split" else 5e-2
    assert abs(float(loss.detach()) - float(z["loss"])) <= ltol * abs(float(z["loss"])) + 1e-6, (float(loss.detach()), float(z["loss"]))
    named = dict(m.named_parameters())
    nlat = int(z["nlat"])
    worst = {}
    for i, n in enumerate(str(x) for x in z["names"]):
        gr = named[n].grad.detach().double().flatten().cpu() / S
        stats, lat = z[f"p{i}_stats"], z[f"p{i}_lat"]
        idx = torch.from_numpy((np.arange(nlat, dtype=np.int64) * 2654435761 % gr.numel()).astype(np.int64))
        scale = float(stats[2]) + 1e-300
        e_lat = float((gr[idx] - torch.from_numpy(lat)).abs().max()) / scale
        e_sum = abs(float(gr.sum()) - float(stats[0])) / (float(stats[1]) + 1e-300)
        e_abs = abs(float(gr.abs().sum()) - float(stats[1])) / (float(stats[1]) + 1e-300)
        worst[n] = (e_lat, e_sum, e_abs)
    top = sorted(worst.items(), key=lambda kv: -max(kv[1]))[:6]
    print("gradient goldens: worst parameters (lattice / sum / |sum|):", [(k, ["%.2e" % v for v in e]) for k, e in top])
    # split operands carry ~22 bits per product: the forward sits at 1e-5 of the reference's f64 outputs (tests/test_split_path_gpu.py); a
    # gradient element is a sum over up to 131 072 pixels / 514 tokens of such products.  Measured: lattice elements <= 1.6e-3 of the
    # gradient's largest element (the 7x7 stem's weight: a sum over every pixel of both frames; every transformer weight <= 6.1e-4), sums
    # and sums of magnitudes <= 1.9e-4.  Bounds = 3x that.
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters() if p.grad is not None)) / S
    print("whole-gradient norm: ours %.6e reference %.6e" % (float(gn), float(z["grad_norm_all"])), "max lattice %.2e max sum %.2e max |sum| %.2e" % tuple(
        max(v[k] for v in worst.values()) for k in range(3)))
    if cdt == "split":
        for n, (e_lat, e_sum, e_abs) in worst.items():
            assert e_lat <= 5e-3 and e_sum <= 6e-4 and e_abs <= 6e-4, (n, e_lat, e_sum, e_abs)
        assert abs(float(gn) - float(z["grad_norm_all"])) <= 1e-3 * float(z["grad_norm_all"]), (float(gn), float(z["grad_norm_all"]))
    else:
        # the 16-bit class (TF32-class operands, 16-bit activations and activation gradients): bounds = 3x the measured values
        for n, (e_lat, e_sum, e_abs) in worst.items():
            assert e_lat <= F16_GRAD_TOL[0] and e_sum <= F16_GRAD_TOL[1] and e_abs <= F16_GRAD_TOL[2], (n, e_lat, e_sum, e_abs)
        assert abs(float(gn) - float(z["grad_norm_all"])) <= F16_GRAD_TOL[3] * float(z["grad_norm_all"]), (float(gn), float(z["grad_norm_all"]))


@pytest.mark.parametrize("cdt", ["f16", "split"])
def test_config4_full_size_training_step(cdt):
    """(cdt = "split": the same step at the REFERENCE'S precision -- f32 activations and gradients, three f16 MFMAs per product in the
    forward AND the backward, VERDICT r2 item 4 "run config 4 once in f32".)
    BASELINE config 4 at FULL size: re10k_8view (config/experiment/re10k_8view.yaml:19-20,61): batch 2, 8 context views, 12 target
    views, ViT-L, 524 288 Gaussians per scene; encoder + decoder + rasterizer forward + backward + clip + AdamW, once.
    Checks: finite loss / gradient norm, every parameter the loss reaches is updated (only scratch.refinenet4.resConfUnit1 of the two
    DPT heads is unreachable), and the loss of the step is run-to-run reproducible (the TRAINING forward's kernels are
    deterministic -- the split-K residual epilogues that make the fused inference forward reproducible only to rounding, unless
    VS_DETERMINISTIC=1, are not on this path; the rasterizer backward's float atomics only touch the gradients)."""
    from vicasplat_amd import callers
    from vicasplat_amd.model.decoder import DecoderSplattingCUDACfg, get_decoder
    d = torch.device("cuda:0")
    dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], False)).to(d)
    batch = _config4_batch(2, 8, 12, d, with_extrinsics=True)
    res = []
    compute = torch.float16 if cdt == "f16" else "split"
    for rep in range(2):
        m = _full_model(compute)
        opt, sched = callers.configure_optimizer(m, lr=4e-5, backbone_lr_multiplier=0.25, warm_up_steps=100)
        before = {n: p.detach().clone() for n, p in m.named_parameters()}
        torch.cuda.reset_peak_memory_stats()
        r = callers.training_step(m, dec, batch, opt, scheduler=sched, camera_weight=1.0, compute_dtype=compute)
        torch.cuda.synchronize()
        assert not r["skipped"] and torch.isfinite(r["loss"]) and torch.isfinite(r["grad_norm"]) and float(r["grad_norm"]) > 0, r
        assert torch.isfinite(r["loss_camera"]) and float(r["loss_camera"]) > 0
        res.append((float(r["loss"]), float(r["loss_mse"]), float(r["grad_norm"])))
        if rep == 0:
            unreached = [n for n, p in m.named_parameters() if p.grad is None]
            assert unreached and all("refinenet4.resConfUnit1" in n for n in unreached), unreached
            same = [n for n, p in m.named_parameters() if p.grad is not None and torch.equal(before[n], p.detach())]
            assert not same, same[:10]
            print(f"config 4 step [{cdt}]: loss {res[0][0]:.6f} (mse {res[0][1]:.6f}) grad_norm {res[0][2]:.4f} peak mem "
                  f"{torch.cuda.max_memory_allocated() / 2**30:.1f} GB, {len(unreached)} unreachable parameters")
        del m, opt, sched, before
    assert res[0][0] == res[1][0] and res[0][1] == res[1][1], res            # forward: bit-reproducible
    assert abs(res[0][2] - res[1][2]) <= 1e-4 * res[0][2], res                 # gradient norm: f32 atomics re-associate


@pytest.mark.parametrize("blocks", [None, "auto"])
def test_config5_per_gpu_batch_split_class_checkpointed(blocks):
    """BASELINE config 5's per-GPU work item under the test record: 24 scenes per GPU (README.md:104, re10k_8view:
    batch 24 on 8 GPUs), 8 context + 12 target views, the reference's precision (split class: f32 activations and gradients) with
    enable_gradient_checkpointing() -- the reference trains with it on (re10k_8view.yaml:61), and 24 scenes only fit 288 GB that way.
    One full step (encoder + decoder + rasterizer forward and backward, clip, AdamW) through the GradReducer the N > 1 launch uses
    (world size 1 here: buckets and views, no collective -- the 8-GPU leg of the configuration is the driver's SCALE run):
    finite loss and gradient norm, no overflow skip, every reachable parameter updated, peak memory inside the device."""
    from vicasplat_amd import callers
    from vicasplat_amd import dist as vdist
    from vicasplat_amd.model.decoder import DecoderSplattingCUDACfg, get_decoder
    d = torch.device("cuda:0")
    dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], False)).to(d)
    batch = _config4_batch(24, 8, 12, d, with_extrinsics=True)
    m = _full_model("split")
    # None: the reference's policy (every block recomputed); "auto" (late round 5): only as many blocks as the 288 GB require
    m.enable_gradient_checkpointing(blocks)
    opt, sched = callers.configure_optimizer(m, lr=4e-5, backbone_lr_multiplier=0.25, warm_up_steps=100)
    reducer = vdist.GradReducer(m.parameters())
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    torch.cuda.reset_peak_memory_stats()
    r = callers.training_step(m, dec, batch, opt, scheduler=sched, camera_weight=1.0, compute_dtype="split", reducer=reducer)
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    print(f"config 5 per-GPU step [split, checkpointed ({blocks or 'all blocks'}), 24 scenes]: loss {float(r['loss']):.6f} grad_norm {float(r['grad_norm']):.4f} peak {peak:.1f} GB")
    assert not r["skipped"] and torch.isfinite(r["loss"]) and torch.isfinite(r["grad_norm"]) and float(r["grad_norm"]) > 0, r
    unreached = [n for n, p in m.named_parameters() if p.grad is None]
    assert unreached and all("refinenet4.resConfUnit1" in n for n in unreached), unreached[:5]
    same = [n for n, p in m.named_parameters() if p.grad is not None and torch.equal(before[n], p.detach())]
    assert not same, same[:10]
    assert peak < 262.0, peak
    reducer.remove()
    del m, opt, reducer, before, batch
    torch.cuda.empty_cache()


def test_gradient_checkpointing_recomputes_the_same_step():
    """enable_gradient_checkpointing() (vicasplat.py:140, backbone_vica.py:464-474,504-516): per-block recomputation gives the same
    outputs (bit-identical forward) and the same gradients, with a lower activation peak.  The backward kernels accumulate
    LayerNorm / bias column sums and dk / dv with f32 atomics, so two IDENTICAL runs already differ in the last bits: the
    checkpointed run must sit inside that run-to-run spread.  The loss is taken on the encoder outputs directly (the rasterizer
    backward's atomics would only widen the spread)."""
    from vicasplat_amd.model.encoder.train_forward import forward_train
    d = torch.device("cuda:0")
    img, K = er.synthetic_input(1, 3, 256, 0)
    img, K = img.to(d), K.to(d)
    g = torch.Generator().manual_seed(3)
    cot = torch.randn(1, 3, 256, 256, 86, generator=g).to(d) * 1e-3
    out = []
    for ck in (False, False, True, "partial"):
        m, _ = _tiny_model(torch.float16)
        if ck == "partial":      # late round 5: only the first encoder block and no decoder block recomputed (enable_gradient_checkpointing(blocks=...))
            m.enable_gradient_checkpointing(blocks=(1, 0))
        elif ck:
            m.enable_gradient_checkpointing()
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        o = forward_train(m, img, K, torch.float16)
        loss = (o["raw_gaussians"] * cot).sum() * 64.0 + o["pred_extrins"].square().sum()
        loss.backward()
        torch.cuda.synchronize()
        peak = torch.cuda.max_memory_allocated() - base
        grads = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
        out.append((float(loss), grads, peak))
        del m, o, loss
    assert out[0][0] == out[1][0] == out[2][0]
    assert set(out[0][1]) == set(out[2][1])
    spread = lambda a, b: max(float((a[n] - b[n]).abs().max() / (a[n].abs().max() + 1e-30)) for n in a)
    s_base, s_ck = spread(out[0][1], out[1][1]), spread(out[0][1], out[2][1])
    print(f"checkpointing: {len(out[0][1])} gradients; worst relative difference run-to-run {s_base:.2e}, checkpointed vs plain {s_ck:.2e}; "
          f"activation peak {out[0][2] / 2**20:.0f} MiB -> {out[2][2] / 2**20:.0f} MiB")
    # (the spread of ONE pair of identical runs is itself a random draw: 7e-4 .. 1e-3 here in f16, the checkpointed run 9e-4 .. 2.1e-3 over a dozen
    # repetitions on one box -- a floor at the documented f16 atomics spread keeps an unlucky baseline draw from failing the comparison)
    assert s_ck <= max(3 * s_base, 5e-3), (s_base, s_ck)
    assert out[2][2] < out[0][2]
    s_part = spread(out[0][1], out[3][1])
    assert out[3][0] == out[0][0] and s_part <= max(3 * s_base, 5e-3), (s_base, s_part)
    assert out[2][2] < out[3][2] < out[0][2], [o[2] for o in out]      # memory between the two policies


class _RoundGrad(torch.autograd.Function):
    """Identity whose incoming gradient is rounded to the 16-bit operand dtype: mirrors a place where the HIP backward materialises
    a gradient in 16 bit."""

    @staticmethod
    def forward(ctx, x, dt):
        ctx.dt = dt
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dt).to(g.dtype), None


def test_encoder_block_backward_tight_against_rounding_matched_autograd():
    """the loose bounds of the gradient tests above come from 16-bit rounding of the FORWARD activations, which
    hides a few-% defect of one operator.  Here the float64 autograd reference rounds exactly where the HIP block does -- forward: LN
    outputs, q|k|v after RoPE, softmax probabilities, attention output, fc1 pre-activation, GELU output; backward: every gradient the
    kernels materialise in 16 bit -- with straight-through rounding, so that only summation order is left: all 13 gradients of a
    ViT-L block must agree to 1.5e-3 of their scale (f16; measured <= 5.4e-4)."""
    from vicasplat_amd import train
    dt = torch.float16
    d = torch.device("cuda:0")
    frames, gh, C, H = 2, 16, 1024, 16
    tokens = gh * gh + 1
    M = frames * tokens
    g = torch.Generator().manual_seed(5)
    n = "backbone.enc_blocks.0"
    W = {}
    def mk(name, shape, std):
        W[name] = torch.randn(*shape, generator=g) * std
    mk(n + ".norm1.weight", (C,), 0.1); W[n + ".norm1.weight"] += 1
    mk(n + ".norm1.bias", (C,), 0.1)
    mk(n + ".norm2.weight", (C,), 0.1); W[n + ".norm2.weight"] += 1
    mk(n + ".norm2.bias", (C,), 0.1)
    for nm, shp in ((".attn.qkv", (3 * C, C)), (".attn.proj", (C, C)), (".mlp.fc1", (4 * C, C)), (".mlp.fc2", (C, 4 * C))):
        mk(n + nm + ".weight", shp, 1.0 / math.sqrt(shp[1])); mk(n + nm + ".bias", (shp[0],), 0.1)
    x0 = torch.randn(frames, tokens, C, generator=g)
    dy = torch.randn(frames, tokens, C, generator=g) * 0.1
    pos = er.patch_positions(frames, gh, gh)
    extra = pos[:, :1].clone(); extra[:, :, 0] += pos[:, -1:, 0] + 1
    pos = torch.cat([pos, extra], 1)

    rt = lambda t: t + (t.to(dt).to(t.dtype) - t).detach()            # forward rounding, straight-through gradient
    rg = lambda t: _RoundGrad.apply(t, dt)
    f64 = torch.float64
    Wr = {k: (v.to(dt) if k.endswith("weight") and v.dim() == 2 else v).to(f64).requires_grad_() for k, v in W.items()}
    xr = x0.to(f64).requires_grad_()
    h1 = rg(rt(er.ln(Wr, n + ".norm1", xr)))
    qkv = er.lin(Wr, n + ".attn.qkv", h1).reshape(frames, -1, 3, H, C // H).permute(2, 0, 3, 1, 4)
    q, k, v = rg(rt(er.rope2d(qkv[0], pos, 100.0))), rg(rt(er.rope2d(qkv[1], pos, 100.0))), rg(rt(qkv[2]))
    s = rg((q @ k.transpose(-2, -1)) * 0.125)
    att = rg(rt(er.heads_merge(rt(s.softmax(-1)) @ v)))
    xm = xr + rg(er.lin(Wr, n + ".attn.proj", att))
    h2 = rg(rt(er.ln(Wr, n + ".norm2", xm)))
    z = rg(rt(er.lin(Wr, n + ".mlp.fc1", h2)))
    a = rg(rt(torch.nn.functional.gelu(z)))
    xo = xm + rg(er.lin(Wr, n + ".mlp.fc2", a))
    (xo * dy.to(f64)).sum().backward()

    f = lambda name: W[name].to(d)
    p = train.EncBlockParams(ln1_w=f(n + ".norm1.weight"), ln1_b=f(n + ".norm1.bias"), qkv_w=f(n + ".attn.qkv.weight").to(dt),
                             qkv_b=f(n + ".attn.qkv.bias"), proj_w=f(n + ".attn.proj.weight").to(dt), proj_b=f(n + ".attn.proj.bias"),
                             ln2_w=f(n + ".norm2.weight"), ln2_b=f(n + ".norm2.bias"), fc1_w=f(n + ".mlp.fc1.weight").to(dt),
                             fc1_b=f(n + ".mlp.fc1.bias"), fc2_w=f(n + ".mlp.fc2.weight").to(dt), fc2_b=f(n + ".mlp.fc2.bias"))
    pos_d = pos.reshape(M, 2).to(torch.int32).contiguous().to(d)
    x_out, tape = train.enc_block_forward_train(x0.reshape(M, C).to(d), p, pos_d, frames=frames, tokens=tokens, heads=H)
    rel = lambda a_, b_: float((a_.detach().cpu().double() - b_.detach()).abs().max() / (b_.detach().abs().max() + 1e-30))
    e_fwd = rel(x_out, xo.reshape(M, C))
    dx_in, grads = train.enc_block_backward(dy.reshape(M, C).to(d), tape, p)
    errs = {"dx": rel(dx_in, xr.grad.reshape(M, C))}
    for ours, ref in (("ln1_w", ".norm1.weight"), ("ln1_b", ".norm1.bias"), ("qkv_w", ".attn.qkv.weight"), ("qkv_b", ".attn.qkv.bias"),
                      ("proj_w", ".attn.proj.weight"), ("proj_b", ".attn.proj.bias"), ("ln2_w", ".norm2.weight"), ("ln2_b", ".norm2.bias"),
                      ("fc1_w", ".mlp.fc1.weight"), ("fc1_b", ".mlp.fc1.bias"), ("fc2_w", ".mlp.fc2.weight"), ("fc2_b", ".mlp.fc2.bias")):
        errs[ours] = rel(grads[ours], Wr[n + ref].grad)
    print("rounding-matched block: forward", f"{e_fwd:.2e}", {k_: f"{v_:.2e}" for k_, v_ in errs.items()})
    assert e_fwd <= 5e-4
    assert max(errs.values()) <= 1.5e-3, errs          # measured 2e-4 .. 5.4e-4


@pytest.mark.parametrize("cdt", ["split", "f16"])
def test_module_api_trains_like_the_reference_wrapper(cdt):
    """SURVEY 8(b) / VERDICT r3 item 1: the reference's ModelWrapper.training_step calls the encoder MODULE under autograd
    (model_wrapper.py:207-230): `out = self.encoder(batch["context"], self.global_step)`, `self.decoder.forward(out["gaussians"], ...)`,
    the losses, `.backward()` -- no loss scale, plain `.grad`s.  The same sequence here must (1) return the dict of the fused inference
    path (same keys, same values to the class's rounding), (2) leave in `.grad` what the explicit training forward produces for fixed
    cotangents on the encoder's outputs, and (3) end to end (through the rasterizer backward) agree with callers.training_step's
    gradients -- both to the run-to-run spread measured in the same test (the LayerNorm-backward column sums, the cross-neighbour dK / dV
    and the rasterizer backward accumulate with f32 atomics, so two runs of the SAME code differ in the last bits; the internal gradient
    scale itself is exact, a power of two: tests/test_callers_cpu.py checks that part bit for bit)."""
    import bench
    from vicasplat_amd import callers
    from vicasplat_amd.model.decoder import DecoderSplattingCUDACfg, get_decoder
    from vicasplat_amd.model.encoder.train_forward import forward_train
    dt = torch.float16 if cdt == "f16" else "split"
    m, _ = _tiny_model(dt)
    d = torch.device("cuda:0")
    dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], False)).to(d)
    B, V, Vt, S = 1, 2, 2, 64
    img, K = er.synthetic_input(B, V, S, 3)
    tE, tK, tn, tf = bench.target_cameras(B, Vt, d)
    target = torch.rand(B, Vt, 3, S, S, generator=torch.Generator().manual_seed(5)).to(d)
    E = torch.eye(4).repeat(B, V, 1, 1)
    E[:, :, 0, 3] = 0.1 * torch.arange(V)[None]
    ctx = dict(image=img.to(d), intrinsics=K.to(d), extrinsics=E.to(d))
    batch = dict(context=ctx, target=dict(image=target, extrinsics=tE, intrinsics=tK, near=tn, far=tf))
    names = [n for n, _ in m.named_parameters()]
    grads = lambda: {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in m.named_parameters()}

    # ---- (1) same dict as the fused inference path ----
    with torch.no_grad():
        ref_out = m(ctx, 0)
    out = m(ctx, 0)
    assert set(out) == set(ref_out) and out["pred_extrins"].requires_grad and out["gaussians"].means.requires_grad
    tol = 2e-5 if cdt == "split" else 2e-2
    for k in ("pred_extrins", "raw_gaussians", "gaussian_camera_extrins", "gaussian_centers", "context_view_depths"):
        a, b = out[k].detach().float(), ref_out[k].float()
        assert a.shape == b.shape and float((a - b).abs().max() / (b.abs().max() + 1e-12)) <= tol, k
    for k in ("means", "covariances", "harmonics", "opacities", "scales", "rotations"):
        a, b = getattr(out["gaussians"], k).detach(), getattr(ref_out["gaussians"], k)
        assert a.shape == b.shape and float((a - b).abs().max() / (b.abs().max() + 1e-12)) <= tol, k

    # ---- (2) encoder half: fixed cotangents on the encoder outputs, Module API vs the explicit training forward ----
    rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-20))
    gen = torch.Generator().manual_seed(11)
    g = out["gaussians"]
    cots = [(torch.randn(t.shape, generator=gen) * 1e-3).to(d) for t in (g.means, g.covariances, g.harmonics, g.opacities, out["pred_extrins"])]

    def module_backward():
        m.zero_grad(set_to_none=True)
        mo = m(ctx, 0)
        gm = mo["gaussians"]
        torch.autograd.backward([gm.means, gm.covariances, gm.harmonics, gm.opacities, mo["pred_extrins"]], cots)
        return grads()
    ga, ga2 = module_backward(), module_backward()
    S_ = m._boundary_scaler.scale
    assert S_ == (8192.0 if cdt == "split" else 1024.0)
    m.zero_grad(set_to_none=True)
    o = forward_train(m, ctx["image"], ctx["intrinsics"], dt, global_step=0)
    gg = o["gaussians"]
    outs2 = [gg["means"], gg["covariances"], gg["harmonics"], gg["opacities"].unsqueeze(-1), o["pred_extrins"]]
    torch.autograd.backward(outs2, [c * S_ for c in cots])
    gb = grads()
    n_grad = sum(int(ga[n] is not None) for n in names)
    assert all((ga[n] is None) == (gb[n] is None) for n in names)
    assert n_grad >= len(names) - 16          # only scratch.refinenet4.resConfUnit1 of both heads is unreachable
    spread_e = max(rel(ga[n], ga2[n]) for n in names if ga[n] is not None)
    worst_e = max(rel(gb[n] / S_, ga[n]) for n in names if ga[n] is not None)
    print(cdt, "encoder half: run-to-run spread %.2e  module API vs forward_train %.2e" % (spread_e, worst_e))
    # f16 class: a last-bit difference of an f32 atomic sum occasionally flips the f16 rounding of one activation gradient (5e-4 of that
    # element), which the layers behind it spread -- measured with tools/dbg/module_api_diff.py: identical call sequences land 1.5e-3 ..
    # 2.5e-3 apart in the max norm, in a few discrete states; the split class keeps f32 gradients and stays at 1e-6
    floor = 1e-2 if cdt == "f16" else 1e-5
    assert worst_e <= max(4 * spread_e, floor), (worst_e, spread_e)

    # ---- (3) end to end, exactly the wrapper's sequence vs callers.training_step ----
    def wrapper_step():
        m.zero_grad(set_to_none=True)
        mo = m(batch["context"], 0)
        rp = dec.forward(mo["gaussians"], tE, tK, tn, tf, (S, S), depth_mode=None)
        loss = ((rp.color - target) ** 2).mean() + callers.camera_loss(mo["pred_extrins"], ctx["extrinsics"].float(), 1.0)
        loss.backward()
        return float(loss.detach()), grads()
    l1, g1 = wrapper_step()
    l2, g2 = wrapper_step()
    opt, _ = callers.configure_optimizer(m, lr=0.0)           # lr 0: the step leaves the weights alone; clip off: grads = unscaled gradients
    r = callers.training_step(m, dec, batch, opt, compute_dtype=dt, clip=1e30, camera_weight=1.0)
    g3 = grads()
    assert not r["skipped"] and abs(float(r["loss"]) - l1) <= 1e-6 * max(1.0, abs(l1)) and abs(l1 - l2) <= 1e-6 * max(1.0, abs(l1))
    spread = max(rel(g1[n], g2[n]) for n in names if g1[n] is not None)
    worst = max(rel(g3[n], g1[n]) for n in names if g1[n] is not None)
    print(cdt, "loss %.6f run-to-run spread %.2e  wrapper vs training_step %.2e" % (l1, spread, worst))
    assert all((g1[n] is None) == (g3[n] is None) for n in names)
    assert worst <= max(4 * spread, floor), (worst, spread)
    # an output no loss reads stays out of the backward pass: without a camera loss the pose head gets no gradient (None, as in the reference)
    m.zero_grad(set_to_none=True)
    mo = m(batch["context"], 0)
    ((dec.forward(mo["gaussians"], tE, tK, tn, tf, (S, S)).color - target) ** 2).mean().backward()
    assert m.camera_extrinsic_head[1].weight.grad is None and m.backbone.camera_extrinsic_token.grad is not None


def test_module_api_distillation_only_phase():
    """`self.encoder(batch["context"], self.global_step, distill=True)` (model_wrapper.py:207 during `distill_only_steps`): the reference returns
    the centres and poses only (vicasplat.py:234-243) and never runs the Gaussian-parameter head.  Same keys and values as the fused inference
    path; the distillation loss on `gaussian_centers` reaches the pts3d head and the backbone, not the Gaussian-parameter head."""
    m, _ = _tiny_model("split")
    d = torch.device("cuda:0")
    img, K = er.synthetic_input(1, 2, 64, 3)
    E = torch.eye(4).repeat(1, 2, 1, 1)
    ctx = dict(image=img.to(d), intrinsics=K.to(d), extrinsics=E.to(d))
    with torch.no_grad():
        ref = m(ctx, 0, distill=True)
    out = m(ctx, 0, distill=True)
    assert set(out) == set(ref) and "gaussians" not in out and out["gaussian_centers"].requires_grad
    for k in ("pred_extrins", "gaussian_camera_extrins", "gaussian_centers", "context_view_depths"):
        a, b = out[k].detach(), ref[k]
        assert a.shape == b.shape and float((a - b).abs().max() / (b.abs().max() + 1e-12)) <= 2e-5, k
    m.zero_grad(set_to_none=True)
    (out["gaussian_centers"].square().mean() + out["pred_extrins"].square().mean()).backward()
    g = {n: p.grad for n, p in m.named_parameters()}
    assert g["downstream_head1.dpt.head.4.weight"] is not None and torch.isfinite(g["downstream_head1.dpt.head.4.weight"]).all()
    assert g["backbone.enc_blocks.0.attn.qkv.weight"] is not None and g["camera_extrinsic_head.1.weight"] is not None
    assert all(v is None for n, v in g.items() if n.startswith("gaussian_param_head."))
