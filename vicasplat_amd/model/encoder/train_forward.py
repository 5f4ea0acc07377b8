"""Differentiable (training) forward of the VicaSplat encoder, written over the module's parameters with the autograd
Functions of `vicasplat_amd.autograd`: every nn.Linear / 1x1 conv / ConvTranspose(k=s) (MFMA GEMM), LayerNorm + AdaLN
modulation, RoPE, attention and 3x3 convolution (stride 1 and the one stride-2 reassemble conv) runs on the HIP kernels
forward AND backward; PyTorch autograd differentiates the glue (residual adds, gates, token (de)interleaving, the tiny f32
camera-token MLPs / pose head and the per-pixel Gaussian adapter -- together < 1 % of the FLOPs; bilinear x2 and the 7x7
stem (as im2col rows) are on the HIP kernels too).  Every kernel on this path is run-to-run deterministic in the forward.

Structure and names follow vicasplat.py:158-278 / backbone_vica.py:526-582 (state_dict keys of SURVEY Appendix C).
First version of the training path: correct and kernel-backed, not yet tuned -- the inference path
(`VicaSplat.forward`) keeps its own fused, buffer-reusing implementation.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F
import torch.utils.checkpoint

from ... import autograd as A
from ... import ops

LN_EPS = 1e-6
_IM2COL = os.environ.get("VS_IM2COL", "1") != "0"            # 0: F.unfold + F.pad rows of the training stem (A/B runs)
_HEAD_TAIL = os.environ.get("VS_HEAD_TAIL", "1") != "0"      # 0: the operator-by-operator backward of the heads' last 1x1 convolution (A/B runs)


def _ln_f32(P, name, x):
    return F.layer_norm(x, (x.shape[-1],), P[name + ".weight"], P[name + ".bias"], LN_EPS)


def _lin_f32(P, name, x):
    # f32-class (split operands: three f16 MFMAs per product) on the hand-written GEMM, forward and backward -- no vendor-BLAS launch
    return A.linear_split(x, P[name + ".weight"], P.get(name + ".bias"))


def auto_checkpoint_blocks(scenes: int, scale: float, total_gb: float, n_enc: int, n_dec: int, half: bool = False, headroom: float = 0.82):
    """(n_enc, n_dec) blocks to checkpoint so that the step's peak memory stays below `headroom` x the device memory: recompute only what the
    memory requires.  Measured on MI355X (split class, 8 views of 256 x 256 = scale 1, bench_train.py --checkpoint): 6.2 GB per scene with every
    block checkpointed (+ ~3 GB of parameters / optimizer state), a decoder block's saved activations 0.142 GB per scene, an encoder block's
    0.134 GB; the 16-bit classes hold about half.  Decoder blocks are released first (12 x 3.4 GB at 24 scenes buy 59 ms), then encoder blocks
    from the last one down.  24 scenes on 288 GB -> (11, 0): ~240 GB peak in the bench process, ~970 ms instead of 1 070 - 1 100 ms per step."""
    k = (0.5 if half else 1.0) * scale * scenes
    mem, budget = 6.2 * k + 3.0, headroom * total_gb
    nd = n_dec
    while nd > 0 and mem + 0.142 * k <= budget:
        mem += 0.142 * k; nd -= 1
    ne = n_enc
    while nd == 0 and ne > 0 and mem + 0.134 * k <= budget:
        mem += 0.134 * k; ne -= 1
    return ne, nd


def forward_train(model, image: torch.Tensor, intrinsics: torch.Tensor, dt=torch.float16, global_step: int = 0,
                  distill: bool = False) -> dict:
    """image [B,V,3,H,W] normalised to [-1,1], intrinsics [B,V,3,3] -> dict(raw_gaussians [B,V,H,W,86] f32, pred_extrins
    [B,V-1,8], gaussians {means, covariances, harmonics, opacities, scales, rotations}, gaussian_centers, camera_tokens, pred_intrins).
    dt: torch.float16 / torch.bfloat16 (16-bit operands and activations) or "split" -- the reference-precision class: f32 activations
    and gradients, every product three f16 MFMAs on (hi, lo) pairs, forward AND backward (autograd.SPLIT).
    distill: the distillation-only phase of the reference (vicasplat.py:234-243) needs the centres and the poses only -- the
    Gaussian-parameter head and the adapter are skipped (`gaussian_centers` = post-processed pts3d, no `gaussians`).
    The exact-f32 MFMA class (torch.float32) has no backward kernels: it is rejected here instead of silently mixing operand classes
    (the convolution / attention Functions dispatch f32 tensors to the split kernels)."""
    if dt in (torch.float32, "f32", "f32x"):
        raise NotImplementedError('the exact-f32 operand class is inference-only; train in "split" (f32 activations and gradients, '
                                  'f32-class products), torch.float16 or torch.bfloat16')
    P = dict(model.named_parameters())
    adt = A.act_dtype(dt)                 # dtype of the activations between the operators
    cfg = model.backbone.config
    dev = image.device
    B, V, _, H, Wd = image.shape
    p = cfg.patch_size
    gh, gw = H // p, Wd // p
    n = gh * gw
    use_intr = cfg.use_intrinsic_embedding
    N1 = n + (1 if use_intr else 0)          # tokens per frame (the intrinsic token rides behind the patches, backbone_vica.py:455-459)
    BT = B * V
    tabs = model.backbone._pos_tables(B, V, gh, gw, dev)
    # per-block activation checkpointing (backbone_vica.py:464-474,504-516): enable_gradient_checkpointing() on the encoder
    ckpt = bool(getattr(model.backbone, "gradient_checkpointing", False)) and torch.is_grad_enabled()
    cb = getattr(model.backbone, "checkpoint_blocks", None)
    if cb == "auto":
        cb = auto_checkpoint_blocks(image.shape[0], image.shape[1] * image.shape[3] * image.shape[4] / (8.0 * 65536.0),
                                    torch.cuda.get_device_properties(image.device).total_memory / 2 ** 30, model.backbone.config.enc_depth,
                                    model.backbone.config.dec_depth, half=dt != A.SPLIT)
    if os.environ.get("VS_CKPT_BLOCKS"):                  # "n_enc,n_dec": experiment override of the policy
        cb = tuple(int(v) for v in os.environ["VS_CKPT_BLOCKS"].split(","))
    ck_enc = lambda i: ckpt and (cb is None or i < cb[0])
    ck_dec = lambda i: ckpt and (cb is None or i < cb[1])
    lin = lambda name, x: A.linear(x, P[name + ".weight"], P.get(name + ".bias"), dt)
    lin_gelu = lambda name, z: A.linear(z, P[name + ".weight"], P.get(name + ".bias"), dt, gelu_in=True)     # fc2(gelu(z))
    lnm = lambda name, x, **k: A.layernorm_mod(x, P[name + ".weight"], P[name + ".bias"], eps=LN_EPS, **k)

    # ---------------- frame encoder (backbone_vica.py:450-480) ----------------
    frames = image.reshape(BT, 3, H, Wd)
    cols = frames.reshape(BT, 3, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(BT, n, 3 * p * p)   # conv(k=s=16) as a GEMM
    Ce = cfg.enc_embed_dim
    x = A.linear(cols, P["backbone.patch_embed.proj.weight"].flatten(1), P["backbone.patch_embed.proj.bias"], dt).float()
    if use_intr:
        intr = _lin_f32(P, "backbone.intrinsic_encoder", intrinsics.reshape(BT, 1, 9).float())
        x = torch.cat([x, intr], 1)                                                                 # [BT, N1, Ce] f32 stream
    He = cfg.enc_num_heads
    x = x.reshape(BT * N1, Ce)
    def enc_block_split(i, x):              # the block composed of the split-class Functions (croco/blocks.py:114-130)
        nm = f"backbone.enc_blocks.{i}"
        h1, x = lnm(nm + ".norm1", x, out_dtype=adt, skip=True)        # (x leaves through the LayerNorm node: its backward adds the residual gradient)
        qkv = A.linear(h1, P[nm + ".attn.qkv.weight"], P[nm + ".attn.qkv.bias"], dt, rope=(tabs["pos_img"], None, He, Ce, 100.0, 1.0))
        att = A.AttentionFn.apply(qkv, BT, He, N1, N1, N1, N1, None, None, 0)
        x = A.gated_resid(x, lin(nm + ".attn.proj", att))
        h2, x = lnm(nm + ".norm2", x, out_dtype=adt, skip=True)
        return A.gated_resid(x, lin_gelu(nm + ".mlp.fc2", lin(nm + ".mlp.fc1", h2)))

    for i in range(cfg.enc_depth):          # one autograd node per block: LN / qkv+RoPE / attention / proj / LN / fc1 / GELU / fc2
        nm = f"backbone.enc_blocks.{i}"
        if dt == A.SPLIT:
            x = torch.utils.checkpoint.checkpoint(enc_block_split, i, x, use_reentrant=False) if ck_enc(i) else enc_block_split(i, x)
            continue
        names = ("norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias",
                 "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")
        x = A.EncBlockFn.apply(x, tabs["pos_img"], BT, N1, He, dt, ck_enc(i), *[P[f"{nm}.{k}"] for k in names])
    x = lnm("backbone.enc_norm", x.view(BT, N1, Ce), out_dtype=torch.float32)

    # ---------------- video / camera decoder (backbone_vica.py:482-524, block :280-335) ----------------
    T = V
    inter = [x]
    x = lin("backbone.decoder_embed", x).float().view(BT * N1, -1)                                    # image stream, f32 [BT*N1, C]
    C = x.shape[-1]
    Hd = cfg.dec_num_heads
    ti, te = P["backbone.camera_intrinsic_token"], P["backbone.camera_extrinsic_token"]
    cam = torch.cat([ti.expand(B, 1, C), (ti + te).expand(B, T - 1, C)], 1)                          # [B,T,C] f32
    theta = float(cfg.temporal_rope_theta)
    M2 = N1 + 1
    def dec_block(i, x, cam):
        # the elementwise work of the image stream lives in the HIP kernels: LayerNorm + AdaLN writes its rows straight behind
        # each frame's camera-token row, and every residual update x + (1 + gate) * branch is one gated_resid pass
        nm = f"backbone.dec_blocks.{i}"
        cn = _ln_f32(P, nm + ".cam_norm1", cam)
        s1, b1, g1 = _lin_f32(P, nm + ".modulation1.proj", F.silu(cn)).chunk(3, -1)                   # [B,T,C] each
        hmix, x = lnm(nm + ".norm1", x, skip=True, scale=s1.reshape(BT, C), shift=b1.reshape(BT, C), mod_rows=N1, out_dtype=adt,
                   lead=cn.to(adt).reshape(BT, C), lead_rows=N1)                                       # [BT*M2, C]: camera token first
        qkv = A.linear(hmix, P[nm + ".attn.qkv.weight"], P[nm + ".attn.qkv.bias"], dt,
                       rope=(tabs["pos_mix"], tabs["kind_mix"], Hd, C, 100.0, theta))                 # RoPE in the GEMM epilogue
        att = A.AttentionFn.apply(qkv, B, Hd, T * M2, T * M2, T * M2, T * M2, None, tabs["kvlen"], 0)
        x, o_cam = A.gated_resid(x, lin(nm + ".attn.proj", att), g1.reshape(BT, C), N1, N1, M2, 1)
        cam = cam + o_cam.view(B, T, C)
        cn = _ln_f32(P, nm + ".cam_norm2", cam)
        s2, b2, g2, s3, b3, g3 = _lin_f32(P, nm + ".modulation2.proj", F.silu(cn)).chunk(6, -1)
        # cross-neighbour attention (:152-191): q | k | v of frame t (one GEMM over the stacked projq / projk / projv weights),
        # keys gathered from frames t-1 / t+1 by row segments
        himg, x = lnm(nm + ".norm2", x, skip=True, scale=s2.reshape(BT, C), shift=b2.reshape(BT, C), mod_rows=N1, out_dtype=adt)
        ca = nm + ".cross_attn"
        wqkv = torch.cat([P[ca + ".projq.weight"], P[ca + ".projk.weight"], P[ca + ".projv.weight"]], 0)
        bqkv = torch.cat([P[ca + ".projq.bias"], P[ca + ".projk.bias"], P[ca + ".projv.bias"]], 0)
        # (split class: the exponent of the stacked temporary is the smallest of the three parameters' cached ones -- no host read)
        qkv = A.linear(himg, wqkv, bqkv, dt, rope=(tabs["pos_img"], None, Hd, C, 100.0, 1.0),
                       scale_sources=(P[ca + ".projq.weight"], P[ca + ".projk.weight"], P[ca + ".projv.weight"]))
        att = A.AttentionFn.apply(qkv, BT, Hd, N1, 0, N1, 0, tabs["seg"], None, 2 * N1)
        x = A.gated_resid(x, lin(ca + ".proj", att), g2.reshape(BT, C), N1)
        himg, x = lnm(nm + ".norm3", x, skip=True, scale=s3.reshape(BT, C), shift=b3.reshape(BT, C), mod_rows=N1, out_dtype=adt)
        x = A.gated_resid(x, lin_gelu(nm + ".mlp.fc2", lin(nm + ".mlp.fc1", himg)), g3.reshape(BT, C), N1)
        cam = cam + _lin_f32(P, nm + ".mlp_cam.fc2", F.gelu(_lin_f32(P, nm + ".mlp_cam.fc1", cn)))
        return x, cam

    for i in range(cfg.dec_depth):
        if ck_dec(i):      # keep only the block's inputs (x, cam); its Functions re-run in the backward (deterministic kernels)
            x, cam = torch.utils.checkpoint.checkpoint(dec_block, i, x, cam, use_reentrant=False)
        else:
            x, cam = dec_block(i, x, cam)
        inter.append(x.view(BT, N1, C))
    inter[-1] = lnm("backbone.dec_norm", inter[-1], out_dtype=torch.float32)
    cam = _ln_f32(P, "backbone.camera_dec_norm", cam)
    inter = [t[:, :n] for t in inter]                                                                 # drop the intrinsic token (:570-572)

    # ---------------- pose head (vicasplat.py:179-199; misc/dq.py:224-262), f32 ----------------
    d = _lin_f32(P, "camera_extrinsic_head.1", F.relu(cam[:, 1:]))
    d = torch.cat([d[..., :3], d[..., 3:4] + 1.0, d[..., 4:]], -1)
    d = d / d[..., :4].norm(dim=-1, keepdim=True)

    # ---------------- DPT heads (heads/dpt_block.py, dpt_head.py:35-70, dpt_gs_head.py:120-157), NHWC 16-bit ----------------
    def conv1x1(name, t):
        return A.linear(t, P[name + ".weight"].flatten(1), P.get(name + ".bias"), dt)

    def conv1x1_tail(name, t):              # the head's last 1x1 convolution behind a ReLU: one fused backward pass (csrc/head_bwd.hip)
        w = P[name + ".weight"].flatten(1)
        if _HEAD_TAIL and A.head_tail_ok(t, w) and t.dtype == adt:
            return A.head_tail(t, w, P.get(name + ".bias"), dt)
        return conv1x1(name, t)

    def convT(name, t, k):                                       # ConvTranspose2d(kernel = stride = k): a GEMM + depth-to-space
        w = P[name + ".weight"]                                  # [Cin, Cout, k, k]
        Cout = w.shape[1]
        y = A.linear(t, w.permute(2, 3, 1, 0).reshape(k * k * Cout, w.shape[0]), P[name + ".bias"].repeat(k * k), dt, scale_sources=(w,))
        n_, h_, w_ = t.shape[:3]
        return y.view(n_, h_, w_, k, k, Cout).permute(0, 1, 3, 2, 4, 5).reshape(n_, h_ * k, w_ * k, Cout)

    def chunked(fn, t, out_elems_per_item):                      # torch's kernels index with int32: keep every call < 2^31 elements
        step = max(1, (2 ** 31 - 1) // max(1, out_elems_per_item))
        if t.shape[0] <= step:
            return fn(t)
        return torch.cat([fn(t[i:i + step]) for i in range(0, t.shape[0], step)], 0)

    def up2(t):
        return A.upsample2x(t)

    def stem7x7(name, fr):                                       # 7x7, pad 3 conv on the RGB frames as im2col rows + the MFMA GEMM
        w = P[name + ".weight"]                                  # [Cout, 3, 7, 7]
        n_, _, h_, w_ = fr.shape
        if fr.requires_grad or fr.dtype != torch.float32 or not _IM2COL:      # a gradient for the image: torch differentiates unfold / pad
            f = lambda u: F.unfold(u.to(adt), 7, padding=3).transpose(1, 2)          # [n, h*w, 147] in (c, ky, kx) order = weight.flatten(1)
            cols = chunked(f, fr, h_ * w_ * 147)
            cols = F.pad(cols, (0, 256 - 147))     # 256 columns: whole tiles for the reduction-major weight-gradient kernel (no transposes)
        else:                                                  # the same rows written once, padded, by one HIP pass (vs_im2col7x7_rgb)
            cols = ops.im2col7x7_rgb(fr.contiguous(), adt, 256)
        wk = F.pad(w.flatten(1), (0, 256 - 147))
        return A.linear(cols, wk, P.get(name + ".bias"), dt, scale_sources=(w,)).view(n_, h_, w_, w.shape[0])

    def rcu(name, t):
        y = A.conv3x3(t, P[name + ".conv1.weight"], P[name + ".conv1.bias"], relu_in=True)
        return A.conv3x3(y, P[name + ".conv2.weight"], P[name + ".conv2.bias"], relu_in=True, residual=t)   # + t in the conv epilogue

    def fusion(name, t, skip=None):
        if skip is not None:
            t = t + rcu(name + ".resConfUnit1", skip)
        return conv1x1(name + ".out_conv", up2(rcu(name + ".resConfUnit2", t)))

    def trunk(pre):
        L = cfg.dec_depth
        hooks = [0, L * 2 // 4, L * 3 // 4, L]
        maps = [inter[h].to(adt).reshape(BT, gh, gw, -1) for h in hooks]
        a = pre + ".act_postprocess"
        l0 = convT(a + ".0.1", conv1x1(a + ".0.0", maps[0]), 4)
        l1 = convT(a + ".1.1", conv1x1(a + ".1.0", maps[1]), 2)
        l2 = conv1x1(a + ".2.0", maps[2])
        l3 = A.conv3x3(conv1x1(a + ".3.0", maps[3]).contiguous(), P[a + ".3.1.weight"], P.get(a + ".3.1.bias"), stride=2)
        s = pre + ".scratch"                      # (layer{j}_rn and layer_rn.{j-1} are one parameter under two state_dict keys)
        l0, l1, l2, l3 = [A.conv3x3(l.contiguous(), P[f"{s}.layer{j + 1}_rn.weight"], None) for j, l in enumerate((l0, l1, l2, l3))]
        p4 = fusion(s + ".refinenet4", l3)[:, :l2.shape[1], :l2.shape[2]]
        p3 = fusion(s + ".refinenet3", p4, l2)
        p2 = fusion(s + ".refinenet2", p3, l1)
        return fusion(s + ".refinenet1", p2, l0)

    pre = "downstream_head1.dpt"
    t = trunk(pre)
    t = A.conv3x3(t, P[pre + ".head.0.weight"], P[pre + ".head.0.bias"])
    t = A.conv3x3(up2(t), P[pre + ".head.2.weight"], P[pre + ".head.2.bias"], relu_out=True)
    pts16 = conv1x1_tail(pre + ".head.4", t)                                                              # [BT,H,W,3 | 4]: xyz (| confidence logit)
    # predict_conf (distill.yaml:24): confidence = 1 + exp(x) on the fourth channel (postprocess.py:17-18,66-75); element-wise glue in torch
    conf = (1.0 + torch.exp(pts16[..., 3].float())).unflatten(0, (B, V)) if pts16.shape[-1] == 4 else None

    pred_intrins = None
    if not use_intr:
        pred_intrins = _lin_f32(P, "camera_intrinsic_head.1", F.relu(cam[:, 0]))                      # fov head (vicasplat.py:201-205)
    if distill:      # 'exp' post-process of the centres in torch (postprocess.py:46-56): xyz * expm1(|xyz|) / |xyz|
        xyz = pts16[..., :3].float()
        nrm = xyz.norm(dim=-1, keepdim=True)
        centers = (xyz / nrm.clamp(min=1e-8) * torch.expm1(nrm)).unflatten(0, (B, V))
        return dict(pred_extrins=d, gaussian_centers=centers, camera_tokens=cam, pred_intrins=pred_intrins, confidence=conf)

    pre = "gaussian_param_head.dpt"
    t = A.upsample2x_add_relu(trunk(pre), stem7x7(pre + ".input_merger.0", frames))             # up2(trunk) + relu(stem), one launch
    t = A.conv3x3(t, P[pre + ".head.0.weight"], None, relu_out=True)
    gs16 = conv1x1_tail(pre + ".head.4", t)                                                               # [BT,H,W,8+3*d_sh] 16-bit

    # ---------------- 'exp' depth post-process (postprocess.py:46-56) + raw_gaussians concat (vicasplat.py:256) + Gaussian adapter
    # (common/gaussian_adapter.py:168-212): ONE fused HIP kernel per direction on the heads' 16-bit NHWC outputs ----------------
    ga = model.gaussian_adapter
    c = model.cfg.opacity_mapping
    exponent = -1.0 if model.cfg.predict_opacity else 2 ** (c.initial + min(global_step / c.warm_up, 1) * (c.final - c.initial))
    means, cov, sh, op, raw, scales, rot = A.gaussian_adapter(pts16, gs16, ga.sh_mask, scale_act=ga.cfg.scale_act, scale_min=ga.cfg.gaussian_scale_min,
                                                 scale_max=ga.cfg.gaussian_scale_max, opacity_exponent=float(exponent))
    un = lambda u: u.unflatten(0, (B, V))
    raw = un(raw)
    gaussians = dict(means=un(means), covariances=un(cov), harmonics=un(sh), opacities=un(op), scales=un(scales), rotations=un(rot))
    return dict(raw_gaussians=raw, pred_extrins=d, gaussians=gaussians, gaussian_centers=gaussians["means"], camera_tokens=cam,
                pred_intrins=pred_intrins, confidence=conf)
