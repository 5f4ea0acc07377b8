"""The SPLIT-class BACKWARD (reference-precision training step): f32 activations and gradients, every matrix product
three f16 MFMAs on (hi, lo) pairs with f32 accumulation (C-ABI dtype 4; csrc/split_bwd.hip, gemm.hip vs_gemm_wgrad, attention_bwd.hip
vs_attention_backward_split).  Reference: the fp32 / TF32 training step of model_wrapper.py:184-321 (config/experiment/re10k_8view.yaml:75-80).

Operator level, against float64 torch autograd on the SAME f32 inputs: <= 1e-5 of the output scale (the 16-bit backward: ~2e-3) for
gradients of O(1) magnitude, and a stated graceful degradation for small ones (f16 subnormal lo halves: absolute floor 2^-25 per element).  -m gpu.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _dev():
    return torch.device("cuda:0")


def _rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / (b.double().abs().max().cpu() + 1e-300))


@pytest.mark.parametrize("R,C,relu", [(257, 64, False), (1000, 96, True), (64, 8, False), (4100, 256, False), (1000, 128, True), (300, 192, False)])
def test_transpose_f32_and_pack(R, C, relu):
    """vs_transpose_f32 is an exact transpose (+ ReLU, zero padding); vs_transpose_pack_split is vs_split_pack_weight of that transpose."""
    from vicasplat_amd import ops
    d = _dev()
    g = torch.Generator().manual_seed(R + C)
    x = torch.randn(R, C, generator=g).to(d)
    t = ops.transpose_f32(x, 128, relu=relu)
    Rp = (R + 127) // 128 * 128
    ref = torch.zeros(C, Rp, device=d)
    ref[:, :R] = (x.clamp_min(0) if relu else x).t()
    assert t.shape == (C, Rp) and torch.equal(t, ref)
    p = ops.transpose_pack_split(x, 128, relu=relu)
    q = ops.split_pack_weight(ref, 0)
    assert torch.equal(p.data, q.data) and p.acc_scale == 1.0
    if not relu:     # the bias gradient rides on the transpose of dY (f32 atomics over 64-row partial sums: order-dependent in the last bits)
        cs = torch.full((C,), 7.0, device=d)
        t2 = ops.transpose_f32(x, 128, colsum=cs)
        want = x.double().sum(0)
        assert torch.equal(t2, t) and float((cs.double() - want).abs().max()) <= 1e-5 * float(x.abs().sum(0).max())
        cs2 = torch.full((C,), -3.0, device=d)
        p2 = ops.transpose_pack_split(x, 128, scale_exp=3, colsum=cs2)
        assert torch.equal(p2.data, ops.split_pack_weight(ref, 3).data)
        assert float((cs2.double() - want).abs().max()) <= 1e-5 * float(x.abs().sum(0).max())


def test_transpose_conv_tap():
    from vicasplat_amd import ops
    d = _dev()
    N, H, W, C = 2, 7, 9, 32
    x = torch.randn(N, H, W, C, generator=torch.Generator().manual_seed(0)).to(d)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            t = ops.transpose_f32(x.view(-1, C), 64, conv_hw=(H, W), tap=(dy, dx))
            ref = torch.zeros(N, H, W, C, device=d)
            ys, ye = max(0, -dy), min(H, H - dy)
            xs, xe = max(0, -dx), min(W, W - dx)
            ref[:, ys:ye, xs:xe] = x[:, ys + dy:ye + dy, xs + dx:xe + dx]
            assert torch.equal(t[:, :N * H * W], ref.view(-1, C).t()), (dy, dx)
            assert float(t[:, N * H * W:].abs().max()) == 0.0


@pytest.mark.parametrize("T,M,N,ks,tr", [(128, 256, 256, 1, False), (1000, 512, 256, 2, True), (16448, 1024, 768, 4, True), (4100, 256, 1024, 8, False)])
def test_gemm_wgrad_split_atn_matches_float64(T, M, N, ks, tr):
    """vs_gemm_wgrad_split_atn: out = a^T b with a [T, M] f32 read reduction-major (LDS transpose reads of the in-place converted (hi, lo)
    halves) and b through the transposing pack; rows of `a` beyond T are never read (the buffer ends there)."""
    from vicasplat_amd import ops
    d = _dev()
    g = torch.Generator().manual_seed(T + M + N)
    a = torch.randn(T, M, generator=g).to(d)
    b = torch.randn(T, N, generator=g).to(d)
    Tp = (T + 64 * ks - 1) // (64 * ks) * (64 * ks)
    bp = ops.transpose_pack_split(b, 64 * ks)
    assert bp.data.shape == (N, Tp)
    out = torch.full((N, M) if tr else (M, N), float("nan"), device=d)
    ops.gemm_wgrad_split_atn(a, bp, out, ks, transpose_out=tr)
    want = a.double().t() @ b.double()
    assert _rel(out.t() if tr else out, want) <= TOL
    # the round-3 route (both operands transposed by a pass) computes the same products in the same K-tile order
    aT = ops.transpose_f32(a, 64 * ks)
    ref = torch.empty(M, N, device=d)
    ops.gemm_wgrad_split(aT, bp, ref, ks)
    assert _rel(out.t() if tr else out, ref) <= 2e-6


@pytest.mark.parametrize("M,N,K", [(514, 768, 768), (16448, 1024, 1024), (100, 192, 64), (257, 3072, 768), (300, 8, 128), (5000, 83, 256), (2056, 1024, 4096)])
def test_linear_backward_split_matches_float64(M, N, K):
    from vicasplat_amd import ops
    d = _dev()
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(d)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(d)
    dy = torch.randn(M, N, generator=g).to(d)
    dx, dw, db = ops.linear_backward_split(dy, x, w)
    assert _rel(dx, dy.double() @ w.double()) <= TOL
    assert _rel(dw, dy.double().t() @ x.double()) <= TOL
    assert _rel(db, dy.double().sum(0)) <= TOL


def test_linear_backward_split_small_gradients_degrade_gracefully():
    """Gradients far below the f16 normal range (no loss scale): lo underflows into subnormals, the error relative to the gradient's own
    scale grows like 2^-25 / |g| -- 3e-4 at |g| ~ 1e-4 -- and a power-of-two loss scale restores the full precision exactly."""
    from vicasplat_amd import ops
    d = _dev()
    g = torch.Generator().manual_seed(3)
    M, N, K = 2056, 768, 768
    x = torch.randn(M, K, generator=g).to(d)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(d)
    dy = (torch.randn(M, N, generator=g) * 1e-4).to(d)
    dx, dw, _ = ops.linear_backward_split(dy, x, w)
    e_small = max(_rel(dx, dy.double() @ w.double()), _rel(dw, dy.double().t() @ x.double()))
    S = 2.0 ** 14
    dx2, dw2, _ = ops.linear_backward_split(dy * S, x, w)
    e_scaled = max(_rel(dx2 / S, dy.double() @ w.double()), _rel(dw2 / S, dy.double().t() @ x.double()))
    print("small gradients: rel err %.2e unscaled, %.2e with loss scale 2^14" % (e_small, e_scaled))
    assert e_small <= 2e-3 and e_scaled <= TOL


def _attn_ref(q, k, v, keymask, scale):
    s = torch.einsum("qhd,khd->hqk", q, k) * scale
    s = s.masked_fill(~keymask[None], float("-inf"))
    return torch.einsum("hqk,khd->qhd", torch.softmax(s, -1), v)


@pytest.mark.parametrize("case", ["encoder", "video_prefix", "neighbour_segments"])
def test_attention_backward_split_matches_float64(case):
    from vicasplat_amd import ops
    d = _dev()
    g = torch.Generator(device="cpu").manual_seed(len(case))
    H, C = 3, 192
    if case == "encoder":
        nb, Lq = 3, 257
    elif case == "video_prefix":
        nb, Lq = 2, 3 * 86
    else:
        nb, Lq = 4, 100
    rows = nb * Lq
    qkv = (torch.randn(rows, 3 * C, generator=g) * 0.7).to(d)
    dout = torch.randn(rows, C, generator=g).to(d)
    q2, k2, v2 = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    kw, kvlen, seg = dict(nbatch=nb, H=H, Lq=Lq, q_batch_rows=Lq), None, None
    if case == "neighbour_segments":
        nbr = [[1, 1], [0, 2], [1, 3], [2, 2]]
        seg = torch.tensor([[a * Lq, Lq, b * Lq, Lq] for a, b in nbr], dtype=torch.int32, device=d)
        kw.update(kv_seg=seg)
    else:
        kw.update(Lk=Lq, k_batch_rows=Lq)
        if case == "video_prefix":
            kvlen = torch.full((nb, Lq), Lq, dtype=torch.int32)
            for t in range(3):
                kvlen[:, t * 86] = (t + 1) * 86
            kvlen = kvlen.reshape(-1).contiguous().to(d)
            kw.update(q_kvlen=kvlen)
    out = torch.empty(rows, C, dtype=torch.float32, device=d)
    lse = torch.empty(rows, H, dtype=torch.float32, device=d)
    ops.attention(q2, k2, v2, out, lse=lse, split=True, **kw)
    dqkv = torch.empty_like(qkv)
    if seg is None:
        ops.attention_backward_split(q2, k2, v2, out, dout, lse, dq_out=dqkv[:, :C], dk_out=dqkv[:, C:2 * C], dv_out=dqkv[:, 2 * C:], **kw)
        dq, dk, dv = dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:]
    else:
        dq, dk, dv = ops.attention_backward_split(q2, k2, v2, out, dout, lse, max_keys=2 * Lq, **kw)
    qf = qkv.double().clone().requires_grad_()
    outs = []
    for b in range(nb):
        qb = qf[b * Lq:(b + 1) * Lq, :C].reshape(Lq, H, 64)
        if seg is None:
            kb = qf[b * Lq:(b + 1) * Lq, C:2 * C].reshape(Lq, H, 64); vb = qf[b * Lq:(b + 1) * Lq, 2 * C:].reshape(Lq, H, 64)
            mask = torch.ones(Lq, Lq, dtype=torch.bool, device=d)
            if kvlen is not None:
                mask = torch.arange(Lq, device=d)[None, :] < kvlen[b * Lq:(b + 1) * Lq, None]
        else:
            a0, b0 = nbr[b]
            idx = torch.cat([torch.arange(a0 * Lq, (a0 + 1) * Lq), torch.arange(b0 * Lq, (b0 + 1) * Lq)]).to(d)
            kb = qf[idx, C:2 * C].reshape(2 * Lq, H, 64); vb = qf[idx, 2 * C:].reshape(2 * Lq, H, 64)
            mask = torch.ones(Lq, 2 * Lq, dtype=torch.bool, device=d)
        outs.append(_attn_ref(qb, kb, vb, mask, 0.125).reshape(Lq, C))
    ref_out = torch.cat(outs)
    (ref_out * dout.double()).sum().backward()
    gq, gk, gv = qf.grad[:, :C], qf.grad[:, C:2 * C], qf.grad[:, 2 * C:]
    errs = dict(out=_rel(out, ref_out), dq=_rel(dq, gq), dk=_rel(dk, gk), dv=_rel(dv, gv))
    print(case, {k: f"{v:.2e}" for k, v in errs.items()})
    assert max(errs.values()) <= TOL, errs


@pytest.mark.parametrize("N,H,W,Cin,Cout,relu_in", [(2, 16, 16, 64, 128, False), (1, 37, 21, 128, 256, True), (3, 64, 64, 256, 256, True),
                                                    (1, 8, 8, 768, 256, False), (2, 32, 32, 128, 128, True),
                                                    # Cin, Cout multiples of 256: X read reduction-major by vs_conv3x3_wgrad_split_atn
                                                    (1, 16, 16, 256, 512, False), (2, 37, 21, 256, 256, True), (5, 7, 50, 512, 256, False)])
def test_conv3x3_backward_split_matches_float64(N, H, W, Cin, Cout, relu_in):
    from vicasplat_amd import ops
    d = _dev()
    g = torch.Generator().manual_seed(N * H + Cin)
    x = torch.randn(N, H, W, Cin, generator=g).to(d)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(d)
    dy = torch.randn(N, H, W, Cout, generator=g).to(d)
    dx, dw, db = ops.conv3x3_backward_split(dy, x, w, relu_in=relu_in)
    xr = x.double().permute(0, 3, 1, 2).clone().requires_grad_()
    wr = w.double().clone().requires_grad_()
    br = torch.zeros(Cout, dtype=torch.float64, device=d, requires_grad=True)
    y = F.conv2d(F.relu(xr) if relu_in else xr, wr, br, padding=1)
    (y * dy.double().permute(0, 3, 1, 2)).sum().backward()
    errs = dict(dx=_rel(dx, xr.grad.permute(0, 2, 3, 1)), dw=_rel(dw, wr.grad), db=_rel(db, br.grad))
    print({k: f"{v:.2e}" for k, v in errs.items()})
    assert max(errs.values()) <= TOL, errs


def test_f32_elementwise_backward_operators():
    """GELU / its derivative, ReLU mask, gated residual (+ backward), bilinear x2 transpose and the inverse RoPE on f32 tensors vs torch."""
    from vicasplat_amd import ops
    d = _dev()
    g = torch.Generator().manual_seed(11)
    z = (torch.randn(1000, 256, generator=g) * 2).to(d)
    dy = torch.randn(1000, 256, generator=g).to(d)
    zr = z.double().clone().requires_grad_()
    a = F.gelu(zr)
    (a * dy.double()).sum().backward()
    assert _rel(ops.gelu16(z), a.detach()) <= 2e-6 and _rel(ops.gelu_backward(dy, z), zr.grad) <= 2e-6
    assert torch.equal(ops.relu_mask(dy, z), torch.where(z > 0, dy, torch.zeros_like(dy)))
    # gated residual with an interleaved f32 branch: rows 1..5 of every 6-row group of y
    M, C, G = 40, 192, 8
    x = torch.randn(M, C, generator=g).to(d)
    y = torch.randn(M // 5 * 6, C, generator=g).to(d)
    gate = torch.randn(G, C, generator=g).to(d)
    out = ops.gated_resid(x, y, gate, 5, grp_in=5, grp_out=6, grp_off=1)
    ysel = y.view(-1, 6, C)[:, 1:].reshape(M, C)
    ref = x + (1 + gate.repeat_interleave(5, 0)) * ysel
    assert _rel(out, ref) <= 1e-6
    dout = torch.randn(M, C, generator=g).to(d)
    dyb = torch.zeros_like(y)
    dgate = ops.gated_resid_backward(dout, y, gate, 5, dyb, grp_in=5, grp_out=6, grp_off=1)
    assert _rel(dyb.view(-1, 6, C)[:, 1:].reshape(M, C), dout * (1 + gate.repeat_interleave(5, 0))) <= 1e-6
    assert float(dyb.view(-1, 6, C)[:, 0].abs().max()) == 0.0
    assert _rel(dgate, (dout * ysel).view(G, 5, C).sum(1)) <= 1e-5
    # bilinear x2 transpose
    u = torch.randn(2, 16, 12, 64, generator=g).to(d)
    ur = torch.randn(2, 8, 6, 64, generator=g).to(d).double().requires_grad_()
    up = F.interpolate(ur.permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    (up * u.double()).sum().backward()
    assert _rel(ops.upsample2x_backward_nhwc(u), ur.grad) <= 2e-6
    # inverse RoPE undoes the forward rotation on an f32 packed q|k|v
    rows, Hh = 50, 3
    Cc = Hh * 64
    buf = torch.randn(rows, 3 * Cc, generator=g).to(d)
    pos = torch.randint(0, 16, (rows, 2), generator=g, dtype=torch.int32).to(d)
    b2 = buf.clone()
    ops.rope_qk(b2, Hh, Cc, pos, None, 100.0, 1.0)
    assert not torch.allclose(b2[:, :Cc], buf[:, :Cc]) and torch.equal(b2[:, 2 * Cc:], buf[:, 2 * Cc:])
    ops.rope_qk(b2, Hh, Cc, pos, None, 100.0, 1.0, inverse=True)
    assert _rel(b2, buf) <= 2e-6


# ---------------- composed: the whole tiny encoder differentiated in the split class vs float64 autograd over the oracle ----------------
TINY = dict(enc_depth=2, dec_embed_dim=192, dec_num_heads=3)


def _tiny_model():
    import json, os
    from oracle import encoder_ref as er
    from vicasplat_amd.model.encoder import default_cfg, get_encoder
    G = os.path.join(os.path.dirname(__file__), "golden")
    shapes = json.load(open(os.path.join(G, "shapes_tiny.json")))
    m, _ = get_encoder(default_cfg(**TINY))
    W = er.golden_weights(shapes, seed=0)
    m.load_state_dict(W, strict=True)
    m = m.cuda().train()
    m.set_compute_dtype("split")
    return m, W


def test_split_training_forward_matches_reference_goldens():
    """The differentiable forward in the split class reproduces the REAL reference's f64 outputs to the bar of the split inference path
    (tests/test_split_path_gpu.py: 2e-4; measured ~1e-5)."""
    import os
    import numpy as np
    from oracle import encoder_ref as er
    from vicasplat_amd.model.encoder.train_forward import forward_train
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "encoder_tiny_v3.npz"))
    m, _ = _tiny_model()
    B, V = int(z["cfg_B"]), int(z["cfg_V"])
    img, K = er.synthetic_input(B, V, 256, int(z["cfg_seed"]))
    with torch.no_grad():
        out = forward_train(m, img.cuda(), K.cuda(), "split")
    rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64).reshape(np.shape(a))).max() / (np.abs(b).max() + 1e-12))
    LAT = slice(8, 256, 16)
    e_pose = rel(out["pred_extrins"].cpu().numpy(), z["f64_pred_extrins"])
    e_raw = rel(out["raw_gaussians"][:, :, LAT, LAT].cpu().numpy(), z["f64_raw"])
    e_cov = rel(out["gaussians"]["covariances"][:, :, LAT, LAT].cpu().numpy(), z["f64_covariances"])
    print("split train forward vs reference f64:", e_pose, e_raw, e_cov)
    assert e_pose <= 2e-4 and e_raw <= 2e-4 and e_cov <= 2e-4


def test_split_training_backward_tight_against_float64_autograd():
    """end-to-end gradients of the WHOLE tiny encoder (2 + 12 transformer blocks, both DPT heads, adapter, pose head)
    for a random linear functional of its outputs (128 x 128 frames), HIP split-class training path against FLOAT64 torch autograd over
    the oracle -- with the yardstick measured beside it: plain f32 torch autograd over the same oracle (what the reference's fp32 training
    step computes) against the same float64 gradients.  On this random-weight network an f32 evaluation is itself 5e-4 (L2, every
    parameter alike) from the float64 gradient -- forward rounding amplified by the network, not a backward defect -- so 1e-3 in the
    max norm is not reachable by ANY f32 implementation (measured: f32 torch autograd median 4.8e-4, max 3.0e-3).  Asserted:
      * per-parameter L2-relative error <= max(2e-3, 1.25 x the f32 yardstick's worst) for EVERY parameter the loss reaches (the 16-bit backward: ~5e-2),
        median <= 1.5x and worst <= 2.5x the f32 yardstick's; whole-gradient cosine >= 0.9999990 (f32 torch: 0.99999985);
      * max-norm relative error: median <= 1.5x, 90th percentile <= 1.5x, worst <= 3x the f32 yardstick's;
      * the parameters right behind the loss (last 1x1 convolutions of both heads, pose head), where no forward noise has accumulated
        yet: <= 1e-5 in the max norm -- the backward operators' own precision."""
    import re
    from oracle import encoder_ref as er
    from vicasplat_amd.model.encoder.train_forward import forward_train
    m, W = _tiny_model()
    B, V, S = 1, 3, 128
    img, K = er.synthetic_input(B, V, S, 7)
    g = torch.Generator().manual_seed(1)
    r_raw = torch.randn(B, V, S, S, 86, generator=g) * 1e-3
    r_raw[..., :3] *= 0.1
    r_pose = torch.randn(B, V - 1, 8, generator=g)
    r_cov = torch.randn(B, V, S, S, 3, 3, generator=g) * 10.0
    cfg = er.default_cfg(**TINY)
    ref = {}
    for dt in (torch.float64, torch.float32):
        Wr = {k: v.clone().to(dt).requires_grad_() for k, v in W.items()}
        o = er.forward.__wrapped__(Wr, cfg, img.to(dt), K.to(dt))
        loss_r = (o["raw_gaussians"] * r_raw.to(dt)).sum() + (o["pred_extrins"] * r_pose.to(dt)).sum() + (o["gaussians"]["covariances"] * r_cov.to(dt)).sum()
        loss_r.backward()
        ref[dt] = ({k: v.grad for k, v in Wr.items()}, float(loss_r.detach()))
    g64, loss64 = ref[torch.float64]
    g32 = ref[torch.float32][0]
    out = forward_train(m, img.cuda(), K.cuda(), "split")
    loss = (out["raw_gaussians"] * r_raw.cuda()).sum() + (out["pred_extrins"] * r_pose.cuda()).sum() + \
        (out["gaussians"]["covariances"] * r_cov.cuda()).sum()
    S_ = 2.0 ** 12                  # power-of-two loss scale: exact in f32, keeps the (hi, lo) f16 operands of the gradients in range
    (loss * S_).backward()
    assert abs(float(loss.detach()) - loss64) <= 1e-4 * abs(loss64) + 1e-4
    mx, l2, mx32, l232 = {}, {}, {}, {}
    ours_all, ref_all, f32_all = [], [], []
    for name, p in m.named_parameters():
        key = re.sub(r"layer(\d)_rn", lambda mm: f"layer_rn.{int(mm.group(1)) - 1}", name)
        r = g64[key]
        if r is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
        o_ = p.grad.cpu().double() / S_
        f_ = g32[key].double()
        mx[name] = float((o_ - r).abs().max() / (r.abs().max() + 1e-300)); l2[name] = float((o_ - r).norm() / (r.norm() + 1e-300))
        mx32[name] = float((f_ - r).abs().max() / (r.abs().max() + 1e-300)); l232[name] = float((f_ - r).norm() / (r.norm() + 1e-300))
        ours_all.append(o_.flatten()); ref_all.append(r.flatten()); f32_all.append(f_.flatten())
    q = lambda d, f: sorted(d.values())[min(len(d) - 1, int(len(d) * f))]
    a, b, c = torch.cat(ours_all), torch.cat(ref_all), torch.cat(f32_all)
    cos = float(a @ b / (a.norm() * b.norm())); cos32 = float(c @ b / (c.norm() * b.norm()))
    print("params %d | max-norm rel err  ours: median %.2e p90 %.2e max %.2e | f32 torch: median %.2e p90 %.2e max %.2e" %
          (len(mx), q(mx, 0.5), q(mx, 0.9), q(mx, 1.0), q(mx32, 0.5), q(mx32, 0.9), q(mx32, 1.0)))
    print("           | L2 rel err        ours: median %.2e p90 %.2e max %.2e | f32 torch: median %.2e p90 %.2e max %.2e" %
          (q(l2, 0.5), q(l2, 0.9), q(l2, 1.0), q(l232, 0.5), q(l232, 0.9), q(l232, 1.0)))
    print("           | whole gradient cosine ours %.9f, f32 torch %.9f; worst ours:" % (cos, cos32), sorted(l2.items(), key=lambda kv: -kv[1])[:4])
    # (round 5: the worst parameter is held to the f32 yardstick's own worst, not to a fixed 2e-3 below it -- refinenet2.resConfUnit1.conv1 sits
    #  behind a ReLU whose flips decide it: plain f32 torch autograd is 3.1e-3 (L2) / 1.8e-2 (max norm) there, and which side of 2e-3 this path
    #  lands on changes with the summation order of the forward GEMMs: 1.4e-3 on the tile kernels, 2.9e-3 on the skinny kernel the <= 256-row
    #  GEMMs of this tiny model now take, whose median error is LOWER, 4.9e-4 against 6.9e-4)
    assert q(l2, 1.0) <= max(2e-3, 1.25 * q(l232, 1.0)) and q(l2, 0.5) <= 1.5 * q(l232, 0.5) and q(l2, 1.0) <= 2.5 * q(l232, 1.0), sorted(l2.items(), key=lambda kv: -kv[1])[:6]
    assert cos >= 0.999999
    assert q(mx, 0.5) <= 1.5 * q(mx32, 0.5) and q(mx, 0.9) <= 1.5 * q(mx32, 0.9) and q(mx, 1.0) <= 3.0 * q(mx32, 1.0)
    near = ["downstream_head1.dpt.head.4.weight", "downstream_head1.dpt.head.4.bias", "gaussian_param_head.dpt.head.4.weight",
            "gaussian_param_head.dpt.head.4.bias", "camera_extrinsic_head.1.weight", "camera_extrinsic_head.1.bias"]
    print("           | right behind the loss:", {k.split(".dpt.")[-1]: f"{mx[k]:.1e}" for k in near})
    assert max(mx[k] for k in near) <= 1e-5


@pytest.mark.parametrize("P,Cin,Cout,relu", [(64, 256, 83, True), (32 * 700, 256, 83, True), (4096, 128, 3, True), (4096, 128, 4, True),
                                             (2048, 128, 83, False), (2048, 256, 16, True), (32 * 999, 256, 96, True)])
def test_head1x1_backward_split_matches_float64(P, Cin, Cout, relu):
    """vs_head1x1_backward_split (csrc/head_bwd.hip): dt = (t > 0) * (dy W), dW = dy^T t, db = column sums of dy in ONE pass over dy and t --
    against float64 on the same f32 inputs; the mask is exact (taken from the f32 t, zeros and tiny positives included)."""
    from vicasplat_amd import ops
    d = _dev()
    g = torch.Generator().manual_seed(P + Cin + Cout)
    t = torch.randn(P, Cin, generator=g).clamp_min(0)
    t[0, :8] = torch.tensor([0.0, 1e-30, 1e-9, -0.0, 3e-8, 6e-8, 1.0, 0.0])        # mask edge cases: +0 / -0 off, any positive on
    dy = torch.randn(P, Cout, generator=g) * 0.25
    w = torch.randn(Cout, Cin, generator=g) * 0.05
    t, dy, w = t.to(d), dy.to(d), w.to(d)
    e = ops.split_scale_exp(w)
    dt, dw, db = ops.head1x1_backward_split(dy, t, w, relu=relu, scale_exp=e)
    want_dt = dy.double() @ w.double()
    if relu:
        want_dt = want_dt * (t > 0)
        assert torch.equal(dt == 0, ~(t > 0) | (want_dt == 0).to(dt.device))
    assert _rel(dt, want_dt) <= TOL
    assert _rel(dw, dy.double().t() @ t.double()) <= TOL
    assert _rel(db, dy.double().sum(0)) <= TOL
    # deterministic: per-workgroup partials summed in a fixed order
    dt2, dw2, db2 = ops.head1x1_backward_split(dy, t, w, relu=relu, scale_exp=e)
    assert torch.equal(dt, dt2) and torch.equal(dw, dw2) and torch.equal(db, db2)


@pytest.mark.parametrize("Cin,Cout", [(256, 83), (128, 3)])
def test_head_tail_fn_equals_the_operator_route(Cin, Cout):
    """HeadTailSplitFn behind Conv3x3Fn(relu_out=True) gives the gradients of the operator-by-operator route (linear_split + the
    convolution's own ReLU backward): same forward bits, backward to the split class's rounding, and the convolution skips its mask."""
    from vicasplat_amd import autograd as A, ops
    d = _dev()
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(2, 16, 32, Cin, generator=g).to(d).requires_grad_(True)
    w3 = (torch.randn(Cin, Cin, 3, 3, generator=g) * 0.02).to(d).requires_grad_(True)
    w1 = (torch.randn(Cout, Cin, generator=g) * 0.05).to(d).requires_grad_(True)
    b1 = torch.randn(Cout, generator=g).to(d).requires_grad_(True)
    gy = torch.randn(2, 16, 32, Cout, generator=g).to(d)
    outs = []
    calls = []
    orig = ops.relu_mask
    ops.relu_mask = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        for fused in (False, True):
            for p_ in (x, w3, w1, b1):
                p_.grad = None
            t = A.conv3x3(x, w3, None, relu_out=True)
            assert A.head_tail_ok(t, w1)
            y = A.HeadTailSplitFn.apply(t, w1, b1) if fused else A.linear_split(t, w1, b1)
            n0 = len(calls)
            (y * gy).sum().backward()
            assert len(calls) - n0 == (0 if fused else 1)
            outs.append((y.detach().clone(), x.grad.clone(), w3.grad.clone(), w1.grad.clone(), b1.grad.clone()))
    finally:
        ops.relu_mask = orig
    assert torch.equal(outs[0][0], outs[1][0])
    for a_, b_ in zip(outs[0][1:], outs[1][1:]):
        assert _rel(b_, a_) <= 2e-5


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("P,Cin,Cout,ldy", [(32 * 300, 256, 83, 88), (4096, 128, 3, 8), (2048, 128, 4, 4), (2048, 256, 83, 83)])
def test_head1x1_backward_16bit_classes(dtype, P, Cin, Cout, ldy):
    """The 16-bit instantiations of csrc/head_bwd.hip (one MFMA per product) on row-padded gradients (the adapter's backward writes 16-byte
    aligned rows; the padding holds garbage here and must not be read as data): against float64 on the same 16-bit inputs."""
    from vicasplat_amd import ops
    d = _dev()
    g = torch.Generator().manual_seed(P + Cin + Cout + ldy)
    t = torch.randn(P, Cin, generator=g).clamp_min(0).to(dtype).to(d)
    buf = torch.full((P, ldy), float("nan"), dtype=dtype)
    buf[:, :Cout] = (torch.randn(P, Cout, generator=g) * 0.25).to(dtype)
    buf = buf.to(d)
    dy = buf[:, :Cout]
    w = (torch.randn(Cout, Cin, generator=g) * 0.05).to(d)
    dt, dw, db = ops.head1x1_backward(dy, t, w, relu=True)
    w16 = w.to(dtype).double()
    want_dt = (dy.double() @ w16) * (t > 0)
    eps = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    assert dt.dtype == dtype and bool((dt[~(t > 0)] == 0).all())        # (f16 results below 2^-25 round to zero: no exact zero-pattern check)
    assert _rel(dt, want_dt) <= eps
    assert _rel(dw, dy.double().t() @ t.double()) <= 1e-5 and _rel(db, dy.double().sum(0)) <= 1e-5


@pytest.mark.parametrize("N,H,W,Cin,Cout,relu", [(1, 1, 32, 64, 64, False), (2, 5, 64, 128, 128, False), (3, 7, 32, 256, 128, True), (2, 3, 96, 128, 64, True),
                                                 (9, 4, 64, 64, 192, False)])
def test_conv3x3_wgrad_split_stream_matches_float64(N, H, W, Cin, Cout, relu):
    """vs_conv3x3_wgrad_split_stream (csrc/conv_wgrad_stream.hip): dW and db of a 3x3 convolution in one pass over X and dY as they are, against
    float64 autograd of F.conv2d on the same f32 inputs (image borders, strips of a row, several images per worker, relu_in)."""
    from vicasplat_amd import ops
    d = _dev()
    g = torch.Generator().manual_seed(N * 1000 + H * 100 + W + Cin + Cout)
    x = torch.randn(N, H, W, Cin, generator=g).to(d)
    dy = (torch.randn(N, H, W, Cout, generator=g) * 0.5).to(d)
    dw9, db = ops.conv3x3_wgrad_split_stream(dy, x, relu_in=relu)
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, device=d, requires_grad=True)
    xin = x.double().permute(0, 3, 1, 2)
    y = F.conv2d(xin.clamp_min(0) if relu else xin, w, None, padding=1)
    (y * dy.double().permute(0, 3, 1, 2)).sum().backward()
    want = w.grad.permute(2, 3, 1, 0).reshape(9, Cin, Cout)        # [ky, kx, ci, co]
    assert _rel(dw9, want) <= TOL
    assert _rel(db, dy.double().sum((0, 1, 2))) <= TOL
    dw9b, dbb = ops.conv3x3_wgrad_split_stream(dy, x, relu_in=relu)
    assert torch.equal(dw9, dw9b) and torch.equal(db, dbb)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_im2col7x7_rgb_equals_unfold(dtype):
    """vs_im2col7x7_rgb = F.unfold(frames.to(dtype), 7, padding=3).transpose(1, 2) padded with zero columns, bit for bit (a copy)."""
    from vicasplat_amd import ops
    d = _dev()
    fr = (torch.rand(3, 3, 20, 36, generator=torch.Generator().manual_seed(3)) * 2 - 1).to(d)
    got = ops.im2col7x7_rgb(fr, dtype, 256)
    want = F.pad(F.unfold(fr.to(dtype).float(), 7, padding=3).transpose(1, 2), (0, 256 - 147)).to(dtype)
    assert got.shape == (3, 20 * 36, 256) and torch.equal(got, want)


@pytest.mark.parametrize("M,N,K", [(1000, 256, 512), (200, 128, 256), (64, 64, 128), (4100, 192, 320)])
def test_linear_backward_split_with_gelu_derivative_in_the_epilogue(M, N, K):
    """vs_gemm_split epilogue 5: dz = (dy w) * GELU'(z) from the dX GEMM of fc2(gelu(z)) -- every tile route (256 x 256 tiles, 128 x 128 tiles, the
    skinny kernel's 64-row tail) against float64 and against the two-pass route (dX GEMM, then vs_gelu_backward_f32)."""
    from vicasplat_amd import ops
    d = _dev()
    g = torch.Generator().manual_seed(M + N + K)
    dy = (torch.randn(M, N, generator=g) * 0.1).to(d)
    z = (torch.randn(M, K, generator=g) * 1.5).to(d)
    w = (torch.randn(N, K, generator=g) * 0.03).to(d)
    a = ops.gelu16(z)
    e = ops.split_scale_exp(w)
    one, dw1, db1 = ops.linear_backward_split(dy, a, w, scale_exp=e, dgelu_z=z)
    two, dw2, db2 = ops.linear_backward_split(dy, a, w, scale_exp=e)
    two = ops.gelu_backward(two, z)
    zd = z.double()
    ref = (dy.double() @ w.double()) * (0.5 * (1 + torch.erf(zd / 2 ** 0.5)) + zd * torch.exp(-0.5 * zd * zd) / (2 * math.pi) ** 0.5)
    assert _rel(one, ref) <= 3e-6 and _rel(two, ref) <= 3e-6, (_rel(one, ref), _rel(two, ref))
    assert torch.equal(dw1, dw2) and _rel(db1, db2) <= 1e-6


def test_fc2_of_gelu_as_one_autograd_node_matches_the_two_node_chain():
    """autograd.linear(z, w, b, SPLIT, gelu_in=True) = linear(gelu(z)): same forward bits, gradients of z, w, b equal to float64 autograd to 3e-6."""
    from vicasplat_amd import autograd as A
    d = _dev()
    g = torch.Generator().manual_seed(5)
    z0 = (torch.randn(3, 130, 256, generator=g) * 1.2).to(d)
    w0, b0 = (torch.randn(128, 256, generator=g) * 0.05).to(d), (torch.randn(128, generator=g) * 0.1).to(d)
    gy = torch.randn(3, 130, 128, generator=g).to(d)
    outs = []
    for fused in (True, False):
        z, w, b = z0.clone().requires_grad_(True), w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        y = A.linear(z, w, b, A.SPLIT, gelu_in=True) if fused else A.linear(A.gelu(z), w, b, A.SPLIT)
        y.backward(gy)
        outs.append((y.detach(), z.grad, w.grad, b.grad))
    assert torch.equal(outs[0][0], outs[1][0])
    zd, wd, bd = z0.double().requires_grad_(True), w0.double().requires_grad_(True), b0.double().requires_grad_(True)
    torch.nn.functional.linear(torch.nn.functional.gelu(zd), wd, bd).backward(gy.double())
    for got, want in zip(outs[0][1:], (zd.grad, wd.grad, bd.grad)):
        assert _rel(got, want) <= 3e-6, _rel(got, want)


def test_ungated_f32_residual_backward_is_a_pass_through():
    """gated_resid(x, y) without a gate and with an f32 branch: out = x + y, and the backward hands the incoming gradient to both inputs as it is."""
    from vicasplat_amd import autograd as A
    d = _dev()
    g = torch.Generator().manual_seed(9)
    x, y = torch.randn(390, 128, generator=g).to(d).requires_grad_(True), torch.randn(390, 128, generator=g).to(d).requires_grad_(True)
    gy = torch.randn(390, 128, generator=g).to(d)
    out = A.gated_resid(x, y)
    assert torch.equal(out, x.detach() + y.detach())
    (out * 1.0).backward(gy)
    assert torch.equal(x.grad, gy) and torch.equal(y.grad, gy)


@pytest.mark.parametrize("mod", [False, True])
def test_layernorm_node_carries_the_residual_stream(mod):
    """layernorm_mod(x, ..., skip=True) -> (LN(x), x'): x' + branch(LN(x)) differentiates to d x' + LN'(x)^T d h in ONE kernel (dx_add of
    vs_layernorm_backward) -- equal to float64 autograd of the same expression, with and without the AdaLN modulation; an unused x' costs nothing."""
    from vicasplat_amd import autograd as A
    d = _dev()
    g = torch.Generator().manual_seed(3 + int(mod))
    M, C, rows = 6 * 50, 256, 50
    x0 = torch.randn(M, C, generator=g).to(d)
    w0, b0 = (1 + 0.1 * torch.randn(C, generator=g)).to(d), (0.1 * torch.randn(C, generator=g)).to(d)
    sc0, sh0 = (0.2 * torch.randn(M // rows, C, generator=g)).to(d), (0.2 * torch.randn(M // rows, C, generator=g)).to(d)
    gy = torch.randn(M, C, generator=g).to(d)
    x, w, b, sc, sh = (t.clone().requires_grad_(True) for t in (x0, w0, b0, sc0, sh0))
    kw = dict(scale=sc, shift=sh, mod_rows=rows) if mod else {}
    h, xs = A.layernorm_mod(x, w, b, out_dtype=torch.float32, skip=True, **kw)
    assert torch.equal(xs, x0)
    ((xs + 0.5 * h * h) * gy).sum().backward()
    xd, wd, bd, scd, shd = (t.double().requires_grad_(True) for t in (x0, w0, b0, sc0, sh0))
    hd = torch.nn.functional.layer_norm(xd, (C,), wd, bd, 1e-6)
    if mod:
        hd = hd * (1 + scd.repeat_interleave(rows, 0)) + shd.repeat_interleave(rows, 0)
    assert _rel(h, hd) <= 2e-6
    ((xd + 0.5 * hd * hd) * gy.double()).sum().backward()
    for got, want in ((x.grad, xd.grad), (w.grad, wd.grad), (b.grad, bd.grad)) + (((sc.grad, scd.grad), (sh.grad, shd.grad)) if mod else ()):
        assert _rel(got, want) <= 5e-6, _rel(got, want)
    # only LN(x) used: the skip gradient is None, the backward is the plain one
    x2 = x0.clone().requires_grad_(True)
    h2, _ = A.layernorm_mod(x2, w0, b0, out_dtype=torch.float32, skip=True)
    (h2 * gy).sum().backward()
    x3 = x0.clone().requires_grad_(True)
    (A.layernorm_mod(x3, w0, b0, out_dtype=torch.float32) * gy).sum().backward()
    assert torch.equal(x2.grad, x3.grad)
