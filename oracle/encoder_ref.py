"""CPU restatement of VicaSplat's encoder (ViT-L frame encoder + video/camera decoder + DPT heads + Gaussian
adapter + dual-quaternion pose head), written functionally over a plain `{name: tensor}` weight dict.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by
vicasplat_amd/.  Works in float32 or float64 (dtype of the weights decides).  Pinned against the REAL reference
imported on CPU (tests/golden/gen_encoder_golden.py -> tests/golden/encoder_*.npz; tests/test_encoder_oracle.py).

Every function cites the reference lines it restates (paths relative to /root/reference/src/model/encoder/).
Weight names are the reference's state_dict keys.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F

LN_EPS = 1e-6  # backbone/backbone_vica.py:370


def default_cfg(**over) -> dict:
    """config/model/encoder/backbone/vica.yaml + experiment overrides (re10k_8view.yaml:31-33)."""
    c = dict(enc_depth=24, enc_embed_dim=1024, enc_num_heads=16, dec_depth=12, dec_embed_dim=768, dec_num_heads=12,
             patch_size=16, temporal_rope_theta=30.0, rope_base=100.0, sh_degree=4)
    c.update(over)
    return c


# ---------------------------------------------------------------------------------------------------------
# primitives
# ---------------------------------------------------------------------------------------------------------
# ---------------------------------------------------------------------------------------------------------
# operand-rounding emulation of the reference's CUDA precision (test infrastructure)
# ---------------------------------------------------------------------------------------------------------
# The reference runs fp32 storage with TF32 matmuls on its GPU (backbone_vica.py:9: torch.backends.cuda.matmul.allow_tf32 = True;
# cudnn convolutions use TF32 by default): every matmul / convolution operand is rounded to a 10-bit mantissa, products accumulate
# in f32.  `operand_rounding(10)` reproduces that on the CPU oracle by rounding the operands of every linear, convolution and
# attention matmul (round-to-nearest-even on the f32 bit pattern), so the tests can state what the REFERENCE'S OWN CUDA PATH is
# worth against its f32/f64 evaluation on the same scene -- the yardstick for the 16-bit-operand HIP path.
_ROUND_BITS = None


class operand_rounding:
    def __init__(self, mantissa_bits):
        self.bits = mantissa_bits

    def __enter__(self):
        global _ROUND_BITS
        self.prev, _ROUND_BITS = _ROUND_BITS, self.bits

    def __exit__(self, *a):
        global _ROUND_BITS
        _ROUND_BITS = self.prev


def _r(x):
    if _ROUND_BITS is None or x is None or x.dtype != torch.float32:
        return x
    sh = 23 - _ROUND_BITS
    i = x.contiguous().view(torch.int32)
    return ((i + (((i >> sh) & 1) + ((1 << (sh - 1)) - 1))) >> sh << sh).view(torch.float32)


def lin(W, name, x):
    return F.linear(_r(x), _r(W[name + ".weight"]), W.get(name + ".bias"))


def ln(W, name, x):
    return F.layer_norm(x, (x.shape[-1],), W[name + ".weight"], W[name + ".bias"], LN_EPS)


def rope2d(x: torch.Tensor, pos: torch.Tensor, base: float) -> torch.Tensor:
    """backbone/croco/pos_embed.py:112-159 == curope/kernels.cu:39-80.  x [B,H,N,D], pos [B,N,2] (y,x) integer.
    Head dim = [Y half | X half]; inside a half, pairs (i, i+Q), Q = D/4, angle = p * base^(-i/Q)."""
    D = x.shape[-1]
    Q = D // 4
    inv = base ** (-torch.arange(Q, dtype=x.dtype) / Q)
    out = []
    for h in range(2):
        ang = pos[..., h].to(x.dtype)[:, None, :, None] * inv  # [B,1,N,Q]
        c, s = ang.cos(), ang.sin()
        u, v = x[..., h * 2 * Q:h * 2 * Q + Q], x[..., h * 2 * Q + Q:(h + 1) * 2 * Q]
        out += [u * c - v * s, v * c + u * s]
    return torch.cat(out, -1)


def rope1d_interleaved(x: torch.Tensor, theta: float) -> torch.Tensor:
    """src/misc/rope_utils.py:133-188,264-305: temporal RoPE on camera tokens.  x [B,H,T,D]; position t = 0..T-1;
    pairs (2j, 2j+1), angle = t * theta^(-2j/D)."""
    T, D = x.shape[-2], x.shape[-1]
    freqs = theta ** (-torch.arange(0, D, 2, dtype=torch.float32) / D)
    ang = torch.outer(torch.arange(T, dtype=torch.float32), freqs)
    cos = ang.cos().repeat_interleave(2, dim=1).to(x.dtype)
    sin = ang.sin().repeat_interleave(2, dim=1).to(x.dtype)
    xr = x.reshape(*x.shape[:-1], D // 2, 2)
    rot = torch.stack([-xr[..., 1], xr[..., 0]], -1).reshape(x.shape)
    return x * cos + rot * sin


def sdpa(q, k, v, mask: Optional[torch.Tensor] = None):
    """softmax(q k^T / sqrt(d)) v, explicit (croco/blocks.py:106-110; F.scaled_dot_product_attention default scale)."""
    att = (_r(q) @ _r(k).transpose(-2, -1)) * (q.shape[-1] ** -0.5)
    if mask is not None:
        att = att.masked_fill(~mask, float("-inf"))
    return _r(att.softmax(-1)) @ _r(v)


def mlp(W, name, x):
    """croco/blocks.py:73-79 (exact erf GELU)."""
    return lin(W, name + ".fc2", F.gelu(lin(W, name + ".fc1", x)))


def heads_split(x, H):  # [B,L,H*d] -> [B,H,L,d]
    B, L, C = x.shape
    return x.reshape(B, L, H, C // H).transpose(1, 2)


def heads_merge(x):  # [B,H,L,d] -> [B,L,H*d]
    B, H, L, d = x.shape
    return x.transpose(1, 2).reshape(B, L, H * d)


# ---------------------------------------------------------------------------------------------------------
# frame encoder  (backbone_vica.py:450-480; croco/blocks.py:94-130,195-236)
# ---------------------------------------------------------------------------------------------------------
def patch_positions(B, gh, gw):
    ys, xs = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing="ij")
    return torch.stack([ys, xs], -1).reshape(1, gh * gw, 2).expand(B, -1, -1).clone()


def encode_frames(W, cfg, frames, intr_tok):
    """frames [BT,3,H,W] (already normalised), intr_tok [BT,1,C] -> tokens [BT,N+1,C], pos [BT,N+1,2]; intr_tok None
    (use_intrinsic_embedding=false, the *_no_intrin checkpoints): N tokens, no extra position."""
    p = cfg["patch_size"]
    x = F.conv2d(_r(frames), _r(W["backbone.patch_embed.proj.weight"]), W["backbone.patch_embed.proj.bias"], stride=p)
    BT, C, gh, gw = x.shape
    x = x.flatten(2).transpose(1, 2)
    pos = patch_positions(BT, gh, gw)
    if intr_tok is not None:
        x = torch.cat([x, intr_tok], 1)
        extra = pos[:, :1].clone()
        extra[:, :, 0] += pos[:, -1:, 0] + 1  # (y,x) = (gh, 0): backbone_vica.py:455-459
        pos = torch.cat([pos, extra], 1)
    Hn = cfg["enc_num_heads"]
    for i in range(cfg["enc_depth"]):
        n = f"backbone.enc_blocks.{i}"
        h = ln(W, n + ".norm1", x)
        qkv = lin(W, n + ".attn.qkv", h).reshape(BT, -1, 3, Hn, C // Hn).permute(2, 0, 3, 1, 4)
        q, k, v = rope2d(qkv[0], pos, cfg["rope_base"]), rope2d(qkv[1], pos, cfg["rope_base"]), qkv[2]
        x = x + lin(W, n + ".attn.proj", heads_merge(sdpa(q, k, v)))
        x = x + mlp(W, n + ".mlp", ln(W, n + ".norm2", x))
    return ln(W, "backbone.enc_norm", x), pos


# ---------------------------------------------------------------------------------------------------------
# video / camera decoder  (backbone_vica.py:57-335,482-524,585-593)
# ---------------------------------------------------------------------------------------------------------
def camera_mask(T, n_per_frame, first_token_full_attn=False):
    """[T, T*(1+n)] bool: camera query t sees every key of frames <= t (backbone_vica.py:585-593); without the intrinsic
    embedding camera token 0 (the intrinsic token) sees every frame (:589-590)."""
    m = torch.ones(T, T, dtype=torch.bool).tril()
    if first_token_full_attn:
        m[:1] = True
    return m[:, :, None].expand(T, T, 1 + n_per_frame).reshape(T, T * (1 + n_per_frame))


def video_camera_attention(W, n, cfg, img, cam, pos, mask):
    """backbone_vica.py:76-126.  img [B,T,N,C] (modulated LN output), cam [B,T,C] (LN output)."""
    B, T, N, C = img.shape
    Hn = cfg["dec_num_heads"]
    d = C // Hn
    qkv_i = lin(W, n + ".qkv", img).reshape(B, T * N, 3, Hn, d).permute(2, 0, 3, 1, 4)
    posf = pos.reshape(B, T * N, 2)
    q_i, k_i, v_i = rope2d(qkv_i[0], posf, cfg["rope_base"]), rope2d(qkv_i[1], posf, cfg["rope_base"]), qkv_i[2]
    qkv_c = lin(W, n + ".qkv", cam).reshape(B, T, 3, Hn, d).permute(2, 0, 3, 1, 4)
    q_c, k_c, v_c = rope1d_interleaved(qkv_c[0], cfg["temporal_rope_theta"]), rope1d_interleaved(qkv_c[1], cfg["temporal_rope_theta"]), qkv_c[2]
    k = torch.cat([k_c[:, :, :, None], k_i.reshape(B, Hn, T, N, d)], 3).reshape(B, Hn, T * (N + 1), d)
    v = torch.cat([v_c[:, :, :, None], v_i.reshape(B, Hn, T, N, d)], 3).reshape(B, Hn, T * (N + 1), d)
    x_i = heads_merge(sdpa(q_i, k, v)).reshape(B, T, N, C)
    x_c = heads_merge(sdpa(q_c, k, v, mask))
    return lin(W, n + ".proj", x_i), lin(W, n + ".proj", x_c)


def cross_neighbor_attention(W, n, cfg, img, pos):
    """backbone_vica.py:152-191: frame t attends [k_{t-1}; k_{t+1}] (boundaries duplicated; T==2: the other frame)."""
    B, T, N, C = img.shape
    Hn = cfg["dec_num_heads"]
    d = C // Hn
    posf = pos.reshape(B * T, N, 2)
    flat = img.reshape(B * T, N, C)
    q = rope2d(heads_split(lin(W, n + ".projq", flat), Hn), posf, cfg["rope_base"]).reshape(B, T, Hn, N, d)
    k = rope2d(heads_split(lin(W, n + ".projk", flat), Hn), posf, cfg["rope_base"]).reshape(B, T, Hn, N, d)
    v = heads_split(lin(W, n + ".projv", flat), Hn).reshape(B, T, Hn, N, d)
    if T == 2:
        nb = [[1], [0]]
    else:
        nb = [[1, 1]] + [[t - 1, t + 1] for t in range(1, T - 1)] + [[T - 2, T - 2]]
    outs = []
    for t in range(T):
        kk = torch.cat([k[:, j] for j in nb[t]], 2)
        vv = torch.cat([v[:, j] for j in nb[t]], 2)
        outs.append(heads_merge(sdpa(q[:, t], kk, vv)))
    return lin(W, n + ".proj", torch.stack(outs, 1))


def decoder(W, cfg, x, pos):
    """x [B,T,N,Cenc] -> (13 intermediates, cam [B,T,C]).  backbone_vica.py:482-524, block :280-335."""
    B, T, N, _ = x.shape
    inter = [x]
    x = lin(W, "backbone.decoder_embed", x)
    C = x.shape[-1]
    ti, te = W["backbone.camera_intrinsic_token"], W["backbone.camera_extrinsic_token"]
    cam = torch.cat([ti.expand(B, 1, C), (ti + te).expand(B, T - 1, C)], 1)
    mask = camera_mask(T, N, first_token_full_attn=not cfg.get("use_intrinsic_embedding", True))
    for i in range(cfg["dec_depth"]):
        n = f"backbone.dec_blocks.{i}"
        cn = ln(W, n + ".cam_norm1", cam)
        s1, b1, g1 = lin(W, n + ".modulation1.proj", F.silu(cn))[:, :, None].chunk(3, -1)
        xa, ca = video_camera_attention(W, n + ".attn", cfg, ln(W, n + ".norm1", x) * (1 + s1) + b1, cn, pos, mask)
        x = x + (1 + g1) * xa
        cam = cam + ca
        cn = ln(W, n + ".cam_norm2", cam)
        s2, b2, g2, s3, b3, g3 = lin(W, n + ".modulation2.proj", F.silu(cn))[:, :, None].chunk(6, -1)
        x = x + (1 + g2) * cross_neighbor_attention(W, n + ".cross_attn", cfg, ln(W, n + ".norm2", x) * (1 + s2) + b2, pos)
        x = x + (1 + g3) * mlp(W, n + ".mlp", ln(W, n + ".norm3", x) * (1 + s3) + b3)
        cam = cam + mlp(W, n + ".mlp_cam", cn)
        inter.append(x)
    inter[-1] = ln(W, "backbone.dec_norm", inter[-1])
    return inter, ln(W, "backbone.camera_dec_norm", cam)


# ---------------------------------------------------------------------------------------------------------
# DPT heads  (heads/dpt_block.py:79-218,264-419; heads/dpt_head.py:35-70; heads/dpt_gs_head.py:120-157)
# ---------------------------------------------------------------------------------------------------------
def conv(W, name, x, stride=1, padding=0):
    return F.conv2d(_r(x), _r(W[name + ".weight"]), W.get(name + ".bias"), stride=stride, padding=padding)


def up2(x):
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)


def rcu(W, name, x):
    y = conv(W, name + ".conv1", F.relu(x), padding=1)
    y = conv(W, name + ".conv2", F.relu(y), padding=1)
    return y + x


def fusion(W, name, x, skip=None):
    if skip is not None:
        x = x + rcu(W, name + ".resConfUnit1", skip)
    x = rcu(W, name + ".resConfUnit2", x)
    return conv(W, name + ".out_conv", up2(x))


def dpt_trunk(W, pre, cfg, inter, gh, gw):
    """tokens of hooks [0, L/2, 3L/4, L] -> 256-ch feature map at 8x the patch grid (path_1)."""
    L = cfg["dec_depth"]
    hooks = [0, L * 2 // 4, L * 3 // 4, L]
    maps = [inter[h].transpose(1, 2).reshape(inter[h].shape[0], -1, gh, gw) for h in hooks]
    a = pre + ".act_postprocess"
    l0 = F.conv_transpose2d(_r(conv(W, a + ".0.0", maps[0])), _r(W[a + ".0.1.weight"]), W[a + ".0.1.bias"], stride=4)
    l1 = F.conv_transpose2d(_r(conv(W, a + ".1.0", maps[1])), _r(W[a + ".1.1.weight"]), W[a + ".1.1.bias"], stride=2)
    l2 = conv(W, a + ".2.0", maps[2])
    l3 = conv(W, a + ".3.1", conv(W, a + ".3.0", maps[3]), stride=2, padding=1)
    s = pre + ".scratch"
    l0, l1, l2, l3 = [F.conv2d(_r(l), _r(W[f"{s}.layer_rn.{i}.weight"]), None, padding=1) for i, l in enumerate((l0, l1, l2, l3))]
    p4 = fusion(W, s + ".refinenet4", l3)[:, :, :l2.shape[2], :l2.shape[3]]
    p3 = fusion(W, s + ".refinenet3", p4, l2)
    p2 = fusion(W, s + ".refinenet2", p3, l1)
    return fusion(W, s + ".refinenet1", p2, l0)


def pts3d_head(W, cfg, inter, gh, gw):
    """downstream_head1: regression head + 'exp' depth mode (heads/postprocess.py:46-56).  -> ([BT,H,W,3], confidence [BT,H,W] | None):
    a 4-row last convolution is the predict_conf=true layout (vicasplat.py:75), confidence = 1 + exp(x_3)
    (conf_mode ('exp', 1, inf), postprocess.py:17-18,66-75)."""
    pre = "downstream_head1.dpt"
    x = dpt_trunk(W, pre, cfg, inter, gh, gw)
    x = conv(W, pre + ".head.0", x, padding=1)
    x = conv(W, pre + ".head.2", up2(x), padding=1)
    f = conv(W, pre + ".head.4", F.relu(x)).permute(0, 2, 3, 1)
    conf = 1.0 + torch.exp(f[..., 3]) if f.shape[-1] == 4 else None
    x = f[..., :3]
    d = x.norm(dim=-1, keepdim=True)
    return x / d.clip(min=1e-8) * torch.expm1(d), conf


def gs_head(W, cfg, inter, frames, gh, gw):
    """gaussian_param_head: trunk -> x2 -> + ReLU(conv7(image)) -> conv3 -> ReLU -> conv1.  -> [BT,83,H,W]."""
    pre = "gaussian_param_head.dpt"
    x = up2(dpt_trunk(W, pre, cfg, inter, gh, gw)) + F.relu(conv(W, pre + ".input_merger.0", frames, padding=3))
    x = F.relu(F.conv2d(_r(x), _r(W[pre + ".head.0.weight"]), None, padding=1))
    return conv(W, pre + ".head.4", x)


# ---------------------------------------------------------------------------------------------------------
# pose head + Gaussian adapter  (vicasplat.py:179-199; misc/dq.py:224-262; common/gaussian_adapter.py:168-212)
# ---------------------------------------------------------------------------------------------------------
def quat_mul_xyzw(a, b):
    x1, y1, z1, w1 = a.unbind(-1)
    x2, y2, z2, w2 = b.unbind(-1)
    return torch.stack([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                        w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2], -1)


def quat_to_matrix_xyzw(q):
    x, y, z, w = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                        2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                        2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(*q.shape[:-1], 3, 3)


def pose_from_camera_tokens(W, cam_tokens):
    """cam_tokens [B,T-1,C] -> (dq [B,T-1,8], c2w [B,T,4,4]); frame 0 is the identity."""
    d = lin(W, "camera_extrinsic_head.1", F.relu(cam_tokens))
    d = torch.cat([d[..., :3], d[..., 3:4] + 1.0, d[..., 4:]], -1)
    d = d / d[..., :4].norm(dim=-1, keepdim=True)
    qr, qd = d[..., :4], d[..., 4:]
    conj = qr * torch.tensor([-1.0, -1.0, -1.0, 1.0], dtype=d.dtype)
    t = quat_mul_xyzw(2.0 * qd, conj)[..., :3]
    B, Tm1 = d.shape[:2]
    M = torch.zeros(B, Tm1, 4, 4, dtype=d.dtype)
    M[..., :3, :3] = quat_to_matrix_xyzw(qr)
    M[..., :3, 3] = t
    M[..., 3, 3] = 1
    eye = torch.eye(4, dtype=d.dtype).expand(B, 1, 4, 4)
    return d, torch.cat([eye, M], 1)


def sh_mask(degree: int, dtype):
    m = torch.ones((degree + 1) ** 2, dtype=dtype)
    for l in range(1, degree + 1):
        m[l * l:(l + 1) ** 2] = 0.1 * 0.25 ** l
    return m


def gaussian_adapter(raw, degree: int):
    """raw [...,86] = xyz | opacity | scale(3) | quat xyzw(4) | SH (3 x 25, rgb-major)."""
    xyz, op, sc, rot = raw[..., :3], raw[..., 3:4], raw[..., 4:7], raw[..., 7:11]
    nsh = (degree + 1) ** 2
    sh = raw[..., 11:].reshape(*raw.shape[:-1], 3, nsh) * sh_mask(degree, raw.dtype)
    op = torch.sigmoid(op)
    op = 0.5 * (1 - (1 - op) ** 1.0 + op ** 1.0)  # map_pdf_to_opacity with exponent 2^0 (vicasplat.py:143-156)
    sc = (0.001 * F.softplus(sc)).clamp_max(0.3)
    rot = F.normalize(rot, dim=-1)
    i, j, k, r = rot.unbind(-1)
    two_s = 2 / ((rot * rot).sum(-1) + 1e-8)  # common/gaussians.py:14
    R = torch.stack([1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)], -1).reshape(*rot.shape[:-1], 3, 3)
    RS = R * sc[..., None, :]
    cov = RS @ RS.transpose(-1, -2)
    return dict(means=xyz, covariances=cov, harmonics=sh, opacities=op, scales=sc, rotations=rot)


# ---------------------------------------------------------------------------------------------------------
# full forward  (vicasplat.py:158-278, backbone_vica.py:526-582)
# ---------------------------------------------------------------------------------------------------------
@torch.no_grad()
def forward(W: dict, cfg: dict, image: torch.Tensor, intrinsics: torch.Tensor, return_intermediates: bool = False) -> dict:
    """image [B,V,3,H,W] ALREADY normalised to [-1,1] (dataset/shims/normalize_shim.py:21-27), intrinsics [B,V,3,3]."""
    B, V, _, H, Wd = image.shape
    p = cfg["patch_size"]
    gh, gw = H // p, Wd // p
    frames = image.reshape(B * V, 3, H, Wd)
    use_intr = cfg.get("use_intrinsic_embedding", True)
    intr_tok = lin(W, "backbone.intrinsic_encoder", intrinsics.reshape(B * V, 1, 9)) if use_intr else None
    x, pos = encode_frames(W, cfg, frames, intr_tok)
    N1 = x.shape[1]
    inter, cam = decoder(W, cfg, x.reshape(B, V, N1, -1), pos.reshape(B, V, N1, 2))
    n_img = N1 - 1 if use_intr else N1
    inter = [t[:, :, :n_img].reshape(B * V, n_img, -1) for t in inter]  # drop the intrinsic token (:570-572)
    dq, c2w = pose_from_camera_tokens(W, cam[:, 1:])
    centers, conf = pts3d_head(W, cfg, inter, gh, gw)
    centers = centers.reshape(B, V, H, Wd, 3)
    params = gs_head(W, cfg, inter, frames, gh, gw).reshape(B, V, -1, H, Wd).permute(0, 1, 3, 4, 2)
    raw = torch.cat([centers, params], -1)
    out = dict(pred_extrins=dq, gaussian_camera_extrins=c2w, raw_gaussians=raw, gaussian_centers=centers,
               gaussians=gaussian_adapter(raw, cfg["sh_degree"]), pred_intrins=None, gaussian_camera_intrins=None,
               confidence=None if conf is None else conf.reshape(B, V, H, Wd))
    if not use_intr:   # fov head on camera token 0 (vicasplat.py:129-138,201-205) -> pinhole K (cam_utils.py:220-234)
        fov = lin(W, "camera_intrinsic_head.1", F.relu(cam[:, 0]))
        Kp = torch.eye(3, dtype=fov.dtype).repeat(B, 1, 1)
        Kp[:, 0, 0], Kp[:, 1, 1] = 0.5 / torch.tan(fov[:, 0] * 0.5), 0.5 / torch.tan(fov[:, 1] * 0.5)
        Kp[:, 0, 2] = Kp[:, 1, 2] = 0.5
        out["pred_intrins"], out["gaussian_camera_intrins"] = fov, Kp.float()[:, None].repeat(1, V, 1, 1)
    if return_intermediates:
        out["intermediates"] = inter
        out["camera_tokens"] = cam
    return out


# deterministic golden weights + synthetic inputs live in the (algorithm-free) vicasplat_amd.synthetic module so that
# bench.py / smoke() can build the SAME weights for the product path without importing the oracle
from vicasplat_amd.synthetic import golden_weights, synthetic_input  # noqa: E402,F401
