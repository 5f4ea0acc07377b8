"""Numerics of the ViT-block HIP kernels against a plain PyTorch fp32 reference of the same op.  -m gpu.

Tolerances are stated per test: 16-bit operands with f32 accumulation => relative error ~1e-3 (f16) / ~8e-3 (bf16)
against an fp32 reference evaluated on the SAME 16-bit-rounded inputs (so only accumulation order/rounding differ).
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("M,C", [(257, 1024), (2056, 768), (5, 256), (1, 1024)])
@pytest.mark.parametrize("odt", [torch.float32, torch.float16, torch.bfloat16])
def test_layernorm_mod(M, C, odt):
    from vicasplat_amd import ops
    d = _dev()
    g = torch.Generator(device="cpu").manual_seed(M * 7 + C)
    x = (torch.randn(M, C, generator=g) * 3 + 0.5).to(d)
    w, b = (1 + 0.1 * torch.randn(C, generator=g)).to(d), (0.1 * torch.randn(C, generator=g)).to(d)
    out = torch.empty(M, C, dtype=odt, device=d)
    ops.layernorm_mod(x, w, b, out)
    ref = F.layer_norm(x, (C,), w, b, 1e-6)
    tol = {torch.float32: 2e-5, torch.float16: 2e-3, torch.bfloat16: 2e-2}[odt]
    assert (out.float() - ref).abs().max() <= tol * ref.abs().max()
    # modulation + row remap: groups of `gi` rows, written behind one extra row per group
    gi = max(1, M // 4)
    G = (M + gi - 1) // gi
    mod = torch.randn(G, 3 * C, generator=g).to(d) * 0.3
    sc, sh = mod[:, :C], mod[:, C:2 * C]
    out2 = torch.zeros(G * (gi + 1), C, dtype=odt, device=d)
    ops.layernorm_mod(x, w, b, out2, scale=sc, shift=sh, mod_rows=gi, grp_in=gi, grp_out=gi + 1, grp_off=1)
    rows = torch.arange(M, device=d)
    refm = ref * (1 + sc[rows // gi]) + sh[rows // gi]
    got = out2[(rows // gi) * (gi + 1) + 1 + rows % gi].float()
    assert (got - refm).abs().max() <= tol * refm.abs().max()
    assert float(out2[0].abs().max()) == 0  # the skipped rows stay untouched


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(2056, 3072, 1024), (257, 1024, 4096), (100, 768, 768), (16, 2304, 768), (1, 128, 64), (300, 144, 192),
                                   # tail-split launches: 64 full 256-row tiles + 64 rows (small-M tail) / + 128 rows (128-row tail),
                                   # and the 128-row-tile variant (M=4112 -> 32 full tiles + 16 rows would not split: no new round)
                                   (16448, 1024, 64), (16512, 768, 128), (16448, 4096, 64), (98368, 128, 64),
                                   # 256x256 phase-interleaved kernel (K % 128 == 0, full tile rows fill 256 CUs) + its tails
                                   (16448, 1024, 256), (16400, 512, 1024), (65536 + 100, 256, 128)])
def test_gemm_epilogues(dt, M, N, K):
    from vicasplat_amd import ops
    d = _dev()
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    # asymmetric operands (a transposed C-write or swapped operand would not survive this)
    a = (torch.randn(M, K, generator=g) + 0.1 * torch.arange(K).float() / K).to(dt).to(d)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K) + 0.05 * torch.arange(N).float()[:, None] / N).to(dt).to(d)
    bias = torch.randn(N, generator=g).to(d)
    ref = a.float() @ w.float().t() + bias
    rtol = 2e-3 if dt == torch.float16 else 1.2e-2
    out = torch.empty(M, N, dtype=dt, device=d)
    ops.gemm(a, w, bias, out, ops.EPI_STORE16)
    assert (out.float() - ref).abs().max() <= rtol * ref.abs().max()
    ops.gemm(a, w, bias, out, ops.EPI_GELU16)
    assert (out.float() - F.gelu(ref)).abs().max() <= rtol * ref.abs().max()
    o32 = torch.empty(M, N, dtype=torch.float32, device=d)
    ops.gemm(a, w, None, o32, ops.EPI_STORE32)
    assert (o32 - (ref - bias)).abs().max() <= 2e-5 * ref.abs().max() * (1 if dt == torch.float16 else 1) + 1e-3 * 0 + \
        (1e-6 * K)  # f32 accumulation of exactly-representable products: only summation order differs
    # gated residual update into a remapped f32 buffer
    gi = max(1, M // 3)
    G = (M + gi - 1) // gi
    gate = (torch.randn(G, N, generator=g) * 0.5).to(d)
    x32 = torch.randn(G * (gi + 2), N, generator=g).to(d)
    x0 = x32.clone()
    ops.gemm(a, w, bias, x32, ops.EPI_RESID32, gate=gate, gate_rows=gi, grp_in=gi, grp_out=gi + 2, grp_off=2)
    rows = torch.arange(M, device=d)
    orow = (rows // gi) * (gi + 2) + 2 + rows % gi
    exp = x0.clone()
    exp[orow] += (1 + gate[rows // gi]) * ref
    assert (x32 - exp).abs().max() <= 1e-4 * exp.abs().max() + 2e-5 * K ** 0.5


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_gemm256_is_race_free_and_deterministic(dt):
    """The 256x256 kernel orders LDS-DMA against ds_reads with counted vmcnt + barriers only: a misplaced wait shows up
    as rare wrong tiles, so hammer one big shape and demand bit-identical outputs that also match the reference."""
    from vicasplat_amd import ops
    d = _dev()
    g = torch.Generator(device="cpu").manual_seed(7)
    M, N, K = 16384, 2048, 1024
    a = torch.randn(M, K, generator=g).to(dt).to(d)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dt).to(d)
    ref = a.float() @ w.float().t()
    first = None
    for it in range(25):
        out = torch.empty(M, N, dtype=dt, device=d)
        ops.gemm(a, w, None, out, ops.EPI_STORE16)
        if first is None:
            first = out
            assert (out.float() - ref).abs().max() <= (2e-3 if dt == torch.float16 else 1.2e-2) * ref.abs().max()
        else:
            assert torch.equal(out, first), f"run {it} differs in {(out != first).sum().item()} elements"


def _rope2d_ref(x, pos, base):  # x [rows, H, 64] fp32
    out = x.clone()
    inv = base ** (-torch.arange(16, device=x.device, dtype=torch.float32) / 16)
    for half in range(2):
        ang = pos[:, half].float()[:, None, None] * inv
        c, s = ang.cos(), ang.sin()
        u, v = x[..., half * 32:half * 32 + 16], x[..., half * 32 + 16:half * 32 + 32]
        out[..., half * 32:half * 32 + 16] = u * c - v * s
        out[..., half * 32 + 16:half * 32 + 32] = v * c + u * s
    return out


def _rope1d_ref(x, t, theta):  # interleaved pairs
    inv = theta ** (-torch.arange(0, 64, 2, device=x.device, dtype=torch.float32) / 64)
    ang = t.float()[:, None, None] * inv
    c, s = ang.cos(), ang.sin()
    out = x.clone()
    u, v = x[..., 0::2], x[..., 1::2]
    out[..., 0::2] = u * c - v * s
    out[..., 1::2] = v * c + u * s
    return out


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("frames,H,K", [(2, 12, 768), (64, 16, 1024), (5, 3, 192)])
def test_gemm_qkv_rope_fused(dt, frames, H, K):
    """qkv projection with RoPE in the epilogue == projection followed by the standalone rope (2-D image rows, 1-D
    temporal camera rows, untouched v), on every GEMM kernel the dispatcher can pick, tails included."""
    from vicasplat_amd import ops
    d = _dev()
    g = torch.Generator(device="cpu").manual_seed(frames * H)
    rows, C = frames * 258, H * 64
    a = torch.randn(rows, K, generator=g).to(dt).to(d)
    w = (torch.randn(3 * C, K, generator=g) / math.sqrt(K)).to(dt).to(d)
    bias = torch.randn(3 * C, generator=g).to(d)
    kind = torch.zeros(rows, dtype=torch.uint8, device=d)
    pos = torch.zeros(rows, 2, dtype=torch.int32, device=d)
    n = torch.arange(rows, device=d) % 258
    kind[n == 0] = 1
    pos[n == 0, 0] = (torch.arange(rows, device=d) // 258)[n == 0].int() % 8
    img = n > 0
    pos[img, 0] = ((n[img] - 1) // 16).int()
    pos[img, 1] = ((n[img] - 1) % 16).int()
    pos[n == 257] = torch.tensor([16, 0], dtype=torch.int32, device=d)
    kind[n == 5] = 2  # rows that skip the rope entirely
    ref = (a.float() @ w.float().t() + bias).reshape(rows, 3, H, 64)
    exp = ref.clone()
    for blk in (0, 1):
        x = ref[:, blk]
        r2, r1 = _rope2d_ref(x, pos, 100.0), _rope1d_ref(x, pos[:, 0], 30.0)
        exp[:, blk] = torch.where((kind == 1)[:, None, None], r1, torch.where((kind == 2)[:, None, None], x, r2))
    out = torch.empty(rows, 3 * C, dtype=dt, device=d)
    ops.gemm_qkv_rope(a, w, bias, out, C, pos, kind, 100.0, 30.0)
    tol = (3e-3 if dt == torch.float16 else 1.6e-2) * float(exp.abs().max())
    assert (out.float().reshape(rows, 3, H, 64) - exp).abs().max() <= tol
    out2 = torch.empty_like(out)  # kind=None: every row is a 2-D row
    ops.gemm_qkv_rope(a, w, bias, out2, C, pos, None, 100.0, 30.0)
    exp2 = ref.clone()
    for blk in (0, 1):
        exp2[:, blk] = _rope2d_ref(ref[:, blk], pos, 100.0)
    assert (out2.float().reshape(rows, 3, H, 64) - exp2).abs().max() <= tol


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_rope_qk_packed(dt):
    from vicasplat_amd import ops
    d = _dev()
    torch.manual_seed(0)
    rows, H = 2 * 258, 12
    buf = torch.randn(rows, 3 * H * 64, device=d).to(dt)
    ref_in = buf.float().clone()
    kind = torch.zeros(rows, dtype=torch.uint8, device=d)
    pos = torch.zeros(rows, 2, dtype=torch.int32, device=d)
    n = torch.arange(rows, device=d) % 258
    kind[n == 0] = 1                                       # camera tokens: temporal 1-D RoPE with t = frame index
    pos[n == 0, 0] = (torch.arange(rows, device=d) // 258)[n == 0].int() + 3
    img = n > 0
    pos[img, 0] = ((n[img] - 1) // 16).int()
    pos[img, 1] = ((n[img] - 1) % 16).int()
    pos[n == 257] = torch.tensor([16, 0], dtype=torch.int32, device=d)  # intrinsic token
    ops.rope_qk(buf, H, H * 64, pos, kind, 100.0, 30.0)
    got = buf.float()
    for col in (0, H * 64):
        x = ref_in[:, col:col + H * 64].reshape(rows, H, 64)
        exp = torch.where((kind == 1)[:, None, None], _rope1d_ref(x, pos[:, 0], 30.0), _rope2d_ref(x, pos, 100.0))
        tol = 4e-3 if dt == torch.float16 else 3e-2
        assert (got[:, col:col + H * 64].reshape(rows, H, 64) - exp).abs().max() <= tol
    assert torch.equal(got[:, 2 * H * 64:], ref_in[:, 2 * H * 64:])  # v untouched
    # the inverse rotation (backward of the embedding) undoes it
    ops.rope_qk(buf, H, H * 64, pos, kind, 100.0, 30.0, inverse=True)
    assert (buf.float() - ref_in).abs().max() <= (6e-3 if dt == torch.float16 else 5e-2)


def test_curope_drop_in_matches_reference_math():
    """curope.rope_2d / cuRoPE2D surface (curope2d.py:12-40): in-place on a [B,N,H,D] view; backward = inverse rotation."""
    from vicasplat_amd.curope import cuRoPE2D, rope_2d
    d = _dev()
    torch.manual_seed(1)
    B, Hh, N, D = 2, 4, 257, 64
    # the reference feeds q/k as [B,H,N,D] VIEWS of the token-major projection output (blocks.py:97-98), so that the
    # [B,N,H,D] view is contiguous in its last two dims; a contiguous [B,H,N,D] tensor is rejected (curope.cpp:54-59)
    tokens = torch.randn(B, N, Hh, D, device=d).transpose(1, 2)
    pos = torch.stack([torch.randint(0, 17, (B, N), device=d), torch.randint(0, 16, (B, N), device=d)], -1)
    exp = _rope2d_ref(tokens.transpose(1, 2).reshape(B * N, Hh, D), pos.reshape(B * N, 2), 100.0).reshape(B, N, Hh, D).transpose(1, 2)
    t2 = tokens.transpose(1, 2).contiguous().transpose(1, 2)
    out = cuRoPE2D(100.0)(t2, pos)
    assert out.data_ptr() == t2.data_ptr() and (out - exp).abs().max() < 2e-5
    rope_2d(t2.transpose(1, 2), pos, 100.0, -1.0)  # inverse
    assert (t2 - tokens).abs().max() < 2e-5
    with pytest.raises(RuntimeError):
        rope_2d(tokens, pos, 100.0, 1.0)  # [B,H,N,D] passed where [B,N,H,D] is expected -> shape check
    with pytest.raises(RuntimeError):
        rope_2d(tokens.contiguous().transpose(1, 2), pos, 100.0, 1.0)  # not contiguous along the last two dims
    x = tokens.transpose(1, 2).contiguous().requires_grad_(True)
    y = cuRoPE2D(100.0)((x * 1.0).transpose(1, 2), pos)
    y.square().sum().backward()
    assert (x.grad - 2 * x.detach()).abs().max() < 1e-4  # rotation is orthogonal: d/dx sum(rot(x)^2) = 2x


def _sdpa_ref(q, k, v, scale, kvlen=None):
    att = (q @ k.transpose(-1, -2)) * scale
    if kvlen is not None:
        j = torch.arange(k.shape[-2], device=q.device)
        att = att.masked_fill(j[None, :] >= kvlen[:, None], float("-inf"))
    return att.softmax(-1) @ v


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,H,L", [(3, 16, 257), (2, 12, 2 * 258), (1, 3, 64), (1, 2, 5)])
def test_attention_packed_self(dt, B, H, L):
    from vicasplat_amd import ops
    d = _dev()
    torch.manual_seed(B * 100 + L)
    C = H * 64
    qkv = (torch.randn(B * L, 3 * C, device=d) * 1.5).to(dt)
    out = torch.empty(B * L, C, dtype=dt, device=d)
    ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, nbatch=B, H=H, Lq=L, Lk=L, q_batch_rows=L, k_batch_rows=L)
    f = qkv.float().reshape(B, L, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = _sdpa_ref(f[0], f[1], f[2], 0.125).transpose(1, 2).reshape(B * L, C)
    tol = 3e-3 if dt == torch.float16 else 2e-2
    assert (out.float() - ref).abs().max() <= tol * max(1.0, float(ref.abs().max()))


def test_attention_prefix_mask_and_segments():
    from vicasplat_amd import ops
    d = _dev()
    torch.manual_seed(5)
    dt = torch.float16
    # (a) per-query key-prefix lengths == the camera tokens' blocked-causal mask
    B, H, T, n = 2, 3, 4, 10
    L = T * (n + 1)
    C = H * 64
    qkv = torch.randn(B * L, 3 * C, device=d).to(dt)
    kvlen = torch.full((B, L), L, dtype=torch.int32, device=d)
    for t in range(T):
        kvlen[:, t * (n + 1)] = (t + 1) * (n + 1)
    out = torch.empty(B * L, C, dtype=dt, device=d)
    ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, nbatch=B, H=H, Lq=L, Lk=L, q_batch_rows=L, k_batch_rows=L,
                  q_kvlen=kvlen.reshape(-1).contiguous())
    f = qkv.float().reshape(B, L, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = torch.stack([_sdpa_ref(f[0][b], f[1][b], f[2][b], 0.125, kvlen[b]) for b in range(B)]).transpose(1, 2).reshape(B * L, C)
    assert (out.float() - ref).abs().max() <= 3e-3 * float(ref.abs().max())
    # (b) two key segments per batch item == cross-neighbour attention without roll/cat copies
    Bf, N = 5, 37
    q = torch.randn(Bf * N, C, device=d).to(dt)
    kv = torch.randn(Bf * N, 2 * C, device=d).to(dt)
    nb = [[1, 1], [0, 2], [1, 3], [2, 4], [3, 3]]
    seg = torch.tensor([[a * N, N, b * N, N] for a, b in nb], dtype=torch.int32, device=d)
    out = torch.empty(Bf * N, C, dtype=dt, device=d)
    ops.attention(q, kv[:, :C], kv[:, C:], out, nbatch=Bf, H=H, Lq=N, q_batch_rows=N, kv_seg=seg)
    qf = q.float().reshape(Bf, N, H, 64).transpose(1, 2)
    kf = kv[:, :C].float().reshape(Bf, N, H, 64).transpose(1, 2)
    vf = kv[:, C:].float().reshape(Bf, N, H, 64).transpose(1, 2)
    ref = torch.stack([_sdpa_ref(qf[t], torch.cat([kf[a], kf[b]], 1), torch.cat([vf[a], vf[b]], 1), 0.125)
                       for t, (a, b) in enumerate(nb)]).transpose(1, 2).reshape(Bf * N, C)
    assert (out.float() - ref).abs().max() <= 3e-3 * float(ref.abs().max())


def test_attention_online_softmax_rescale_branch():
    """Force the running-max jump in a LATER key tile (guide rule 26): one key far more aligned with the query."""
    from vicasplat_amd import ops
    d = _dev()
    torch.manual_seed(9)
    H, L = 1, 200
    qkv = torch.randn(L, 3 * 64, device=d) * 0.3
    qkv[7, :64] = 4.0            # query 7
    qkv[150, 64:128] = 4.0       # key 150 (third 64-key tile) spikes against it
    qkv = qkv.half()
    out = torch.empty(L, 64, dtype=torch.float16, device=d)
    ops.attention(qkv[:, :64], qkv[:, 64:128], qkv[:, 128:], out, nbatch=1, H=H, Lq=L, Lk=L, q_batch_rows=L, k_batch_rows=L)
    f = qkv.float()
    ref = _sdpa_ref(f[:, :64], f[:, 64:128], f[:, 128:], 0.125)
    assert (out.float() - ref).abs().max() <= 3e-3


@pytest.mark.parametrize("dt,layout", [(torch.float16, "nhwc"), (torch.bfloat16, "nhwc"), (torch.float32, "nhwc"), (torch.float16, "nchw_slice")])
def test_gaussian_adapter_fused(dt, layout):
    """Fused post-process + adapter vs the plain PyTorch formulation (postprocess.py:46-56 + MyGaussianAdapter)."""
    from vicasplat_amd import ops
    from vicasplat_amd.model.encoder.common.gaussian_adapter import GaussianAdapterCfg, MyGaussianAdapter
    d = _dev()
    torch.manual_seed(3)
    N, H, W = 3, 20, 37  # 2220 pixels: not a multiple of 64 -> exercises the tail block
    pts = (torch.randn(N, 3, H, W, device=d) * 0.7).to(dt)
    gs = (torch.randn(N, 83, H, W, device=d) * 2.0).to(dt)
    if layout == "nhwc":
        pts_in, gs_in = pts.contiguous(memory_format=torch.channels_last), gs.contiguous(memory_format=torch.channels_last)
    else:  # a channel slice of a wider NHWC tensor: pixel stride != channel count -> generic kernel
        wide = torch.zeros(N, 90, H, W, device=d, dtype=dt).contiguous(memory_format=torch.channels_last)
        wide[:, :83] = gs
        gs_in = wide[:, :83]
        pts_in = pts.contiguous(memory_format=torch.channels_last)
    ad = MyGaussianAdapter(GaussianAdapterCfg(0.005, 0.04, 4, "softplus")).to(d)
    o = ops.gaussian_adapter(pts_in, gs_in, ad.sh_mask, scale_act="softplus", opacity_exponent=1.0)
    xyz = pts.float().permute(0, 2, 3, 1)
    dist = xyz.norm(dim=-1, keepdim=True)
    centers = xyz / dist.clip(min=1e-8) * torch.expm1(dist)
    raw = torch.cat([centers, gs.float().permute(0, 2, 3, 1)], -1)
    ref = ad(raw, lambda p: 0.5 * (1 - (1 - p) ** 1.0 + p ** 1.0))
    for k, r in (("means", ref.means), ("covariances", ref.covariances), ("harmonics", ref.harmonics), ("opacities", ref.opacities),
                 ("scales", ref.scales), ("rotations", ref.rotations), ("raw", raw)):
        assert o[k].shape == r.shape, k
        assert (o[k] - r).abs().max() <= 2e-5 * max(1.0, float(r.abs().max())), k


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 16, 16, 64, 128), (1, 37, 21, 128, 256), (3, 64, 64, 256, 256), (1, 8, 8, 768, 256), (1, 256, 256, 128, 128),
                                            # 256x256-tile kernel (Cout % 256 == 0, >= 224 pixel tiles), incl. a ragged last tile
                                            (16, 64, 64, 256, 256), (4, 128, 128, 128, 256), (1, 250, 250, 256, 256)])
def test_conv3x3_nhwc_implicit_gemm(dt, N, H, W, Cin, Cout):
    from vicasplat_amd import ops
    d = _dev()
    torch.manual_seed(N * H + Cin)
    x = torch.randn(N, Cin, H, W, device=d).to(dt)
    conv = torch.nn.Conv2d(Cin, Cout, 3, 1, 1).to(d)
    res = torch.randn(N, Cout, H, W, device=d).to(dt)
    xn = x.permute(0, 2, 3, 1).contiguous()
    wp = ops.pack_conv3x3_weight(conv.weight, dt)
    rtol = 3e-3 if dt == torch.float16 else 2e-2
    with torch.no_grad():
        ref = F.conv2d(x.float(), conv.weight.to(dt).float(), conv.bias, padding=1)
        out = ops.conv3x3_nhwc(xn, wp, conv.bias).permute(0, 3, 1, 2).float()
        assert (out - ref).abs().max() <= rtol * ref.abs().max()
        # fused pre-activation + residual + output ReLU (ResidualConvUnit pattern)
        ref2 = F.relu(F.conv2d(F.relu(x.float()), conv.weight.to(dt).float(), conv.bias, padding=1) + res.float())
        out2 = ops.conv3x3_nhwc(xn, wp, conv.bias, residual=res.permute(0, 2, 3, 1).contiguous(), relu_in=True, relu_out=True)
        assert (out2.permute(0, 3, 1, 2).float() - ref2).abs().max() <= rtol * ref2.abs().max()
        out3 = ops.conv3x3_nhwc(xn, wp, None)
        assert (out3.permute(0, 3, 1, 2).float() - (ref - conv.bias[None, :, None, None])).abs().max() <= rtol * ref.abs().max()
        if H <= 64:  # stride 2 (act_postprocess[3], dpt_block.py:404-409)
            ref4 = F.conv2d(x.float(), conv.weight.to(dt).float(), conv.bias, padding=1, stride=2)
            out4 = ops.conv3x3_nhwc(xn, wp, conv.bias, stride=2).permute(0, 3, 1, 2).float()
            assert out4.shape == ref4.shape and (out4 - ref4).abs().max() <= rtol * ref4.abs().max()


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("N,H,W,Cout", [(2, 32, 32, 128), (1, 37, 50, 64), (3, 256, 256, 128)])
def test_conv7x7_rgb_stem(dt, N, H, W, Cout):
    """gs-head input_merger conv (3 -> C, k=7, p=3) as a window GEMM over the zero-bordered NHWC image."""
    from vicasplat_amd import ops
    d = _dev()
    torch.manual_seed(H + Cout)
    x = torch.randn(N, 3, H, W, device=d)
    conv = torch.nn.Conv2d(3, Cout, 7, 1, 3).to(d)
    with torch.no_grad():
        ref = F.conv2d(x.to(dt).float(), conv.weight.to(dt).float(), conv.bias, padding=3)
        out = ops.conv7x7_rgb_nhwc(ops.pad_rgb_nhwc(x, dt), ops.pack_conv7x7_rgb_weight(conv.weight, dt), conv.bias.float(), H, W)
    assert out.shape == (N, H, W, Cout)
    rtol = 2e-3 if dt == torch.float16 else 1.2e-2
    assert (out.permute(0, 3, 1, 2).float() - ref).abs().max() <= rtol * ref.abs().max()


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_upsample2x_nhwc(dt):
    from vicasplat_amd import ops
    d = _dev()
    torch.manual_seed(2)
    for (N, H, W, C) in [(2, 8, 8, 256), (1, 5, 7, 16), (1, 128, 128, 256)]:
        x = torch.randn(N, C, H, W, device=d).to(dt)
        add = torch.randn(N, C, 2 * H, 2 * W, device=d).to(dt)
        ref = F.interpolate(x.float(), scale_factor=2, mode="bilinear", align_corners=True)
        out = ops.upsample2x_nhwc(x.permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2).float()
        tol = 2e-3 if dt == torch.float16 else 1.6e-2
        assert (out - ref).abs().max() <= tol * ref.abs().max()
        out = ops.upsample2x_nhwc(x.permute(0, 2, 3, 1).contiguous(), add.permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2).float()
        assert (out - (ref + add.float())).abs().max() <= tol * (ref + add.float()).abs().max()
        out = ops.upsample2x_nhwc(x.permute(0, 2, 3, 1).contiguous(), add.permute(0, 2, 3, 1).contiguous(), relu_add=True).permute(0, 3, 1, 2).float()
        assert (out - (ref + F.relu(add.float()))).abs().max() <= tol * (ref + add.float()).abs().max()


def test_abi_rejects_bad_arguments_loudly():
    """Error behaviour at the boundary: negative return + vs_last_error message -> RuntimeError (mirrors TORCH_CHECK in
    curope.cpp:54-59); nothing is silently rounded, padded or sent to a slow path."""
    from vicasplat_amd import ops
    d = _dev()
    a = torch.zeros(64, 96, dtype=torch.float16, device=d)
    w = torch.zeros(32, 96, dtype=torch.float16, device=d)
    with pytest.raises(RuntimeError, match="multiple of 64"):
        ops.gemm(a, w, None, torch.empty(64, 32, dtype=torch.float16, device=d), ops.EPI_STORE16)
    x = torch.zeros(1, 8, 8, 48, dtype=torch.float16, device=d)
    with pytest.raises(RuntimeError, match="multiple of 32"):
        ops.conv3x3_nhwc(x, torch.zeros(64, 3, 3, 48, dtype=torch.float16, device=d))
    with pytest.raises(RuntimeError, match="multiple of 8"):
        ops.upsample2x_nhwc(torch.zeros(1, 4, 4, 12, dtype=torch.float16, device=d))
    with pytest.raises((RuntimeError, AssertionError)):
        ops.gemm(a.cpu(), w, None, torch.empty(64, 32, dtype=torch.float16, device=d), ops.EPI_STORE16)  # host tensor
    q = torch.zeros(16, 3 * 64, dtype=torch.float16, device=d)
    with pytest.raises(RuntimeError, match="C % 64"):
        ops.gemm_qkv_rope(torch.zeros(16, 64, dtype=torch.float16, device=d), torch.zeros(192, 64, dtype=torch.float16, device=d), None, q,
                          48, torch.zeros(16, 2, dtype=torch.int32, device=d))


# ---- encoder backward building blocks (training groundwork) vs torch autograd ----
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(514, 768, 768), (16448, 1024, 1024), (100, 192, 64), (257, 3072, 768)])
def test_linear_backward_matches_autograd(dt, M, N, K):
    from vicasplat_amd import ops
    d = _dev()
    g = torch.Generator(device="cpu").manual_seed(M + N)
    x = torch.randn(M, K, generator=g).to(dt).to(d)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dt).to(d)
    dy = (torch.randn(M, N, generator=g) * 0.5).to(dt).to(d)
    xr, wr = x.float().requires_grad_(), w.float().requires_grad_()
    br = torch.zeros(N, device=d, requires_grad=True)
    (F.linear(xr, wr, br) * dy.float()).sum().backward()
    dx, dw, db = ops.linear_backward(dy, x, w)
    rt = 3e-3 if dt == torch.float16 else 1.6e-2
    assert (dx.float() - xr.grad).abs().max() <= rt * xr.grad.abs().max()
    assert (dw - wr.grad).abs().max() <= 2e-3 * wr.grad.abs().max() + 1e-4  # f32 accumulation of exact 16-bit products
    assert (db - br.grad).abs().max() <= 1e-4 * br.grad.abs().max() + 1e-3
    # transpose16 pads with zeros
    t = ops.transpose16(x, 64)
    assert t.shape == (K, (M + 63) // 64 * 64) and torch.equal(t[:, :M], x.t()) and float(t[:, M:].abs().sum()) == 0


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_transpose16_border_relu_colsum(dt):
    """vs_transpose16_ex: transposed zero-bordered pixel grid straight from NHWC, input ReLU and column sums in the same pass."""
    from vicasplat_amd import ops
    d = _dev()
    torch.manual_seed(11)
    for (N, H, W, C) in [(2, 5, 7, 96), (1, 16, 16, 64), (3, 1, 1, 8)]:
        x = torch.randn(N, H, W, C, device=d).to(dt)
        for relu in (False, True):
            cs = torch.full((C,), 7.0, device=d)
            P = N * (H + 2) * (W + 2)
            buf = torch.full((C, 8 + (P + 127) // 128 * 128 + 8), 3.0, dtype=dt, device=d)
            t = ops.transpose16(x.view(-1, C), 128, colsum_out=cs, out=buf[:, 8:-8], border_hw=(H, W), relu=relu)
            xr = torch.relu(x) if relu else x
            ref = F.pad(xr.permute(0, 3, 1, 2), (1, 1, 1, 1)).permute(1, 0, 2, 3).reshape(C, P)
            assert torch.equal(t[:, :P], ref) and float(t[:, P:].abs().sum()) == 0
            assert float((buf[:, :8] - 3).abs().sum()) == 0 and float((buf[:, -8:] - 3).abs().sum()) == 0   # outside the view: untouched
            want = xr.float().sum((0, 1, 2))
            assert (cs - want).abs().max() <= 1e-3 * want.abs().max() + 1e-3
    # slice-blocked output with halos: slice z = padded pixels [z*SL - halo, (z+1)*SL + halo), zeros outside [0, P)
    N, H, W, C = 2, 6, 5, 72
    x = torch.randn(N, H, W, C, device=d).to(dt)
    P = N * (H + 2) * (W + 2)
    for (ks, halo) in [(4, 8), (7, 0), (2, 16)]:
        cs = torch.zeros(C, device=d) if halo == 0 else None
        t = ops.transpose16(x.view(-1, C), 64 * ks, border_hw=(H, W), slices=ks, halo=halo, colsum_out=cs)
        Ppad = (P + 64 * ks - 1) // (64 * ks) * (64 * ks)
        SL = Ppad // ks
        assert t.shape == (ks, C, SL + 2 * halo)
        ref = F.pad(F.pad(x.permute(0, 3, 1, 2), (1, 1, 1, 1)).permute(1, 0, 2, 3).reshape(C, P), (halo, Ppad - P + halo))
        for z in range(ks):
            assert torch.equal(t[z], ref[:, z * SL:(z + 1) * SL + 2 * halo])
        if cs is not None:
            want = x.float().sum((0, 1, 2))
            assert (cs - want).abs().max() <= 1e-3 * want.abs().max() + 1e-3
    from vicasplat_amd import _lib as L
    x = torch.zeros(10, 8, dtype=dt, device=d); o = torch.zeros(8, 64, dtype=dt, device=d)
    with pytest.raises(RuntimeError, match="whole number"):
        L.check(L.lib().vs_transpose16_ex(L.ptr(x), 8, L.ptr(o), 64, 10, 8, 64, None, 1, 2, 2, 0, 1, 0, 0, L.stream_ptr(d)), "vs_transpose16_ex")
    with pytest.raises(RuntimeError, match="halo"):
        L.check(L.lib().vs_transpose16_ex(L.ptr(x), 8, L.ptr(o), 64, 10, 8, 64, None, 1, 0, 0, 0, 1, 8, 0, L.stream_ptr(d)), "vs_transpose16_ex")


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,ks", [(256, 256, 128 * 3, 3), (512, 256, 128 * 28 * 2, 28), (256, 768, 128, 1), (1024, 256, 1280, 5),
                                      (256, 256, 64 * 6, 6), (384, 256, 1024, 4)])
@pytest.mark.parametrize("mode", ["workspace", "atomics", "legacy"])
def test_gemm_splitk_accumulate_matches_matmul(dt, M, N, K, ks, mode):
    """Weight-gradient GEMM on both tilings (256x256 when the output is whole 256-tiles and K % (128 ks) == 0, else 128x128) and
    both ways of joining the K slices (partial tiles + reduce kernel / f32 atomics): out += A W^T onto a non-zero start."""
    from vicasplat_amd import ops
    d = _dev()
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) / math.sqrt(K)).to(dt).to(d)
    w = torch.randn(N, K, generator=g).to(dt).to(d)
    out = torch.randn(M, N, generator=g).to(d)
    want = out.double() + a.double() @ w.double().t()
    if mode == "legacy":
        ops.gemm_splitk_accumulate(a, w, out, ks)
    else:
        ops.gemm_wgrad(a, w, out, ks, workspace=mode == "workspace")
    assert (out.double() - want).abs().max() <= 2e-5 * want.abs().max() + 2e-5


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("Kred,M,N,ks", [(256, 256, 256, 1), (130, 256, 512, 1), (1000, 512, 256, 3), (16448, 768, 768, 28), (4099, 1024, 256, 5),
                                         (70000, 83, 256, 16), (70000, 4, 128, 8), (3000, 96, 200, 2)])
def test_gemm_wgrad_tn_matches_matmul(dt, Kred, M, N, ks):
    """Weight gradient from reduction-major operands (LDS transpose reads): out = a^T w, incl. reduction lengths that are not a
    multiple of the K tiling (zero-filled by the kernel), row strides larger than the row, accumulate and atomics modes."""
    from vicasplat_amd import ops
    d = _dev()
    g = torch.Generator(device="cpu").manual_seed(Kred + M + N)
    Mp, Np = (M + 7) // 8 * 8, (N + 7) // 8 * 8               # rows padded to 16-byte multiples (any M, N: skinny layers)
    abuf = (torch.randn(Kred, Mp + 24, generator=g) / math.sqrt(Kred)).to(dt).to(d)
    wbuf = torch.randn(Kred, Np + 8, generator=g).to(dt).to(d)
    a, w = abuf[:, 8:8 + M], wbuf[:, :N]                      # 16-byte aligned column-offset views (ld > row length)
    want = a.double().t() @ w.double()
    out = torch.full((M, N), 1e9, device=d)
    ops.gemm_wgrad_tn(a, w, out, ks, accumulate=False)
    tol = 2e-5 * float(want.abs().max()) + 2e-5
    assert float((out.double() - want).abs().max()) <= tol
    out2 = torch.randn(M, N, generator=g).to(d)
    want2 = out2.double() + want
    ops.gemm_wgrad_tn(a, w, out2, ks, accumulate=True)
    assert float((out2.double() - want2).abs().max()) <= tol
    out3 = torch.zeros(M, N, device=d)
    ops.gemm_wgrad_tn(a, w, out3, ks, workspace=False)
    assert float((out3.double() - want).abs().max()) <= tol


@pytest.mark.parametrize("M,N,K", [(49344 // 8, 1024, 1024), (6168 + 192, 1024, 4096), (200, 768, 768), (64, 1024, 1024), (1000, 768, 3072)])
def test_gemm_resid_matches_inplace_update(M, N, K):
    """vs_gemm_resid (residual read from a second buffer) == clone + vs_gemm_bias_act epilogue 2, bit for bit on the main tiles and to
    f32 rounding on split-K tails (the in-place form splits K there), incl. the small-M and tail launches, with and without a gate."""
    from vicasplat_amd import ops
    d = _dev()
    g = torch.Generator(device="cpu").manual_seed(M + K)
    a = (torch.randn(M, K, generator=g) / math.sqrt(K)).half().to(d)
    w = torch.randn(N, K, generator=g).half().to(d)
    b = torch.randn(N, generator=g).to(d)
    x = torch.randn(M, N, generator=g).to(d)
    for gate_rows in (0, 257):
        gate = None if gate_rows == 0 else (torch.randn((M + 256) // 257, N, generator=g) * 0.3).to(d)
        ref = x.clone()
        ops.gemm(a, w, b, ref, ops.EPI_RESID32, gate=gate, gate_rows=gate_rows)
        x0 = x.clone()
        out = ops.gemm_resid(a, w, b, x, gate=gate, gate_rows=gate_rows)
        assert torch.equal(x, x0)                                  # the source stream is untouched
        assert (out - ref).abs().max() <= 2e-6 * ref.abs().max() + 2e-6


def test_gemm_wgrad_taps_and_errors():
    """Tap-fused form: out[t] += A (W shifted by shifts[t])^T, incl. odd (2-byte aligned) shifts, with and without workspace."""
    from vicasplat_amd import ops, _lib as L
    d = _dev()
    g = torch.Generator(device="cpu").manual_seed(5)
    for (M, N, K, ks) in [(256, 256, 1024, 4), (128, 192, 512, 2), (256, 512, 256, 1)]:
        a = (torch.randn(M, K, generator=g) / 16).half().to(d)
        wfull = torch.randn(N, K + 64, generator=g).half().to(d)
        shifts = [-19, -1, 0, 1, 18]
        for wsflag in (True, False):
            out = torch.randn(len(shifts), M, N, generator=g).to(d)
            want = torch.stack([out[t].double() + a.double() @ wfull[:, 32 + sh:32 + sh + K].double().t() for t, sh in enumerate(shifts)])
            ops.gemm_wgrad(a, wfull[:, 32:32 + K], out, ks, shifts=shifts, workspace=wsflag)
            assert (out.double() - want).abs().max() <= 2e-5 * want.abs().max() + 2e-5
            if ks > 1:   # the same product from slice-blocked operands (halo 32 around every W slice)
                SL = K // ks
                a3 = a.view(M, ks, SL).permute(1, 0, 2).contiguous()
                w3 = torch.stack([wfull[:, z * SL:(z + 1) * SL + 64] for z in range(ks)]).contiguous()
                out2 = torch.zeros_like(out)
                ops.gemm_wgrad(a3, w3[:, :, 32:32 + SL], out2, ks, shifts=shifts, workspace=wsflag)
                want2 = torch.stack([a.double() @ wfull[:, 32 + sh:32 + sh + K].double().t() for sh in shifts])
                assert (out2.double() - want2).abs().max() <= 2e-5 * want2.abs().max() + 2e-5
    a = torch.zeros(256, 256, dtype=torch.float16, device=d); o = torch.zeros(256, 256, device=d); ws = torch.zeros(16, device=d)
    rc = L.lib().vs_gemm_wgrad(L.ptr(a), L.ptr(a), L.ptr(o), 256, 256, 256, 256, 256, 256, 0, 0, 0, None, 0, 2, 1, L.ptr(ws), 64, 1, L.stream_ptr(d))
    assert rc != 0 and b"workspace" in L.lib().vs_last_error()
    rc = L.lib().vs_gemm_wgrad(L.ptr(a), L.ptr(a), L.ptr(o), 256, 256, 256, 256, 256, 256, 0, 0, 0, None, 3, 2, 1, None, 0, 1, L.stream_ptr(d))
    assert rc != 0 and b"shifts" in L.lib().vs_last_error()
    # accumulate = 0 overwrites (workspace mode only)
    a = (torch.randn(256, 512, generator=g) / 16).half().to(d); w = torch.randn(256, 512, generator=g).half().to(d)
    o = torch.full((256, 256), 1e9, device=d)
    ops.gemm_wgrad(a, w, o, 2, accumulate=False)
    want = a.double() @ w.double().t()
    assert (o.double() - want).abs().max() <= 2e-5 * want.abs().max() + 2e-5
    rc = L.lib().vs_gemm_wgrad(L.ptr(a), L.ptr(w), L.ptr(o), 256, 256, 512, 512, 512, 256, 0, 0, 0, None, 0, 2, 1, None, 0, 0, L.stream_ptr(d))
    assert rc != 0 and b"accumulate" in L.lib().vs_last_error()


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_gated_resid_and_lead_layernorm_match_autograd(dt):
    """autograd.gated_resid (x + (1 + gate) * branch with the camera-token row split off), autograd.gelu and layernorm_mod(lead=)
    against the same expressions in plain torch, forward and backward."""
    from vicasplat_amd import autograd as A
    d = _dev()
    torch.manual_seed(21)
    BT, N1, C = 6, 37, 192
    M2 = N1 + 1
    x = torch.randn(BT * N1, C, device=d, requires_grad=True)
    y = torch.randn(BT * M2, C, device=d).to(dt).requires_grad_()
    gate = (torch.randn(BT, C, device=d) * 0.3).requires_grad_()
    out, extra = A.gated_resid(x, y, gate, N1, N1, M2, 1)
    yv = y.float().view(BT, M2, C)
    ref = x.view(BT, N1, C) + (1 + gate[:, None]) * yv[:, 1:]
    tol = 2e-3 if dt == torch.float16 else 1.6e-2
    assert (out.view(BT, N1, C) - ref).abs().max() <= 1e-5 * ref.abs().max() and torch.equal(extra.view(BT, C), yv[:, 0])
    go, ge = torch.randn_like(out), torch.randn_like(extra)
    gx, gy, gg = torch.autograd.grad((out * go).sum() + (extra * ge).sum(), (x, y, gate))
    rx, ry, rg = torch.autograd.grad((ref.reshape(-1, C) * go).sum() + (yv[:, :1] * ge).sum(), (x, y, gate))
    assert torch.equal(gx, rx) and (gy.float() - ry.float()).abs().max() <= tol * ry.float().abs().max()
    assert (gg - rg).abs().max() <= 1e-4 * rg.abs().max() + 1e-4
    # no gate, identity row map
    y2 = torch.randn(BT * N1, C, device=d).to(dt).requires_grad_()
    o2 = A.gated_resid(x, y2)
    assert (o2 - (x + y2.float())).abs().max() <= 1e-5 * o2.abs().max()
    g2 = torch.autograd.grad((o2 * go).sum(), y2)[0]
    assert (g2.float() - go).abs().max() <= tol * go.abs().max()
    # GELU
    z = (torch.randn(64, 256, device=d) * 2).to(dt).requires_grad_()
    a = A.gelu(z)
    zr = z.detach().float().requires_grad_()
    ar = F.gelu(zr)
    assert (a.float() - ar).abs().max() <= tol * ar.abs().max()
    da = torch.randn_like(ar)
    assert (torch.autograd.grad((a.float() * da).sum(), z)[0].float() - torch.autograd.grad((ar * da).sum(), zr)[0]).abs().max() <= 2 * tol * da.abs().max()
    # LayerNorm + modulation written behind a leading row per frame
    w = (torch.randn(C, device=d) * 0.2 + 1).requires_grad_(); b = (torch.randn(C, device=d) * 0.1).requires_grad_()
    sc = (torch.randn(BT, C, device=d) * 0.3).requires_grad_(); sh = (torch.randn(BT, C, device=d) * 0.3).requires_grad_()
    lead = torch.randn(BT, C, device=d).to(dt).requires_grad_()
    h = A.layernorm_mod(x, w, b, scale=sc, shift=sh, mod_rows=N1, out_dtype=dt, lead=lead, lead_rows=N1)
    ln = F.layer_norm(x, (C,), w, b, 1e-6).view(BT, N1, C) * (1 + sc[:, None]) + sh[:, None]
    href = torch.cat([lead.float()[:, None], ln], 1).reshape(BT * M2, C)
    assert h.shape == (BT * M2, C) and (h.float() - href).abs().max() <= tol * href.abs().max()
    gh = torch.randn_like(href)
    got = torch.autograd.grad((h.float() * gh).sum(), (x, w, b, sc, sh, lead))
    want = torch.autograd.grad((href * gh).sum(), (x, w, b, sc, sh, lead))
    for gname, u, v in zip(("x", "w", "b", "scale", "shift", "lead"), got, want):
        assert (u.float() - v.float()).abs().max() <= 2 * tol * v.float().abs().max() + 1e-4, gname


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("act,exponent", [("softplus", 1.0), ("bounded", 2.0), ("exp", -1.0)])
def test_gaussian_adapter_function_matches_autograd(dt, act, exponent):
    """autograd.gaussian_adapter (fused HIP kernel forward and backward) against the per-pixel PyTorch formulation of
    postprocess.py:46-56 + gaussian_adapter.py:168-212 differentiated by torch autograd, for all outputs incl. raw."""
    from vicasplat_amd import autograd as A
    d = _dev()
    torch.manual_seed(17)
    N, H, W, nsh = 2, 9, 13, 25
    pts = (torch.randn(N, H, W, 4, device=d) * 0.7).to(dt).requires_grad_()
    gs = torch.randn(N, H, W, 8 + 3 * nsh, device=d).to(dt).requires_grad_()
    with torch.no_grad():
        gs[..., 1:4] *= 3                       # some scales past the clamp / softplus knee
    mask = torch.linspace(1.0, 0.1, nsh, device=d)
    smin, smax = 0.5, 15.0
    means, cov, sh, op, raw, scales_o, rot_o = A.gaussian_adapter(pts, gs, mask, scale_act=act, scale_min=smin, scale_max=smax, opacity_exponent=exponent)

    pr, gr = pts.detach().float().requires_grad_(), gs.detach().float().requires_grad_()
    xyz = pr[..., :3]
    dist = xyz.norm(dim=-1, keepdim=True)
    centers = xyz / dist.clip(min=1e-8) * torch.expm1(dist)
    o = torch.sigmoid(gr[..., 0])
    if exponent > 0:
        o = 0.5 * (1 - (1 - o) ** exponent + o ** (1 / exponent))
    v = gr[..., 1:4]
    sc = {"softplus": lambda: (0.001 * F.softplus(v)).clamp_max(0.3), "exp": lambda: torch.exp(v).clamp_max(0.3),
          "bounded": lambda: smin + (smax - smin) * torch.sigmoid(v)}[act]()
    rot = F.normalize(gr[..., 4:8], dim=-1)
    qi, qj, qk, qr = rot.unbind(-1)
    two_s = 2 / ((rot * rot).sum(-1) + 1e-8)
    R = torch.stack([1 - two_s * (qj * qj + qk * qk), two_s * (qi * qj - qk * qr), two_s * (qi * qk + qj * qr),
                     two_s * (qi * qj + qk * qr), 1 - two_s * (qi * qi + qk * qk), two_s * (qj * qk - qi * qr),
                     two_s * (qi * qk - qj * qr), two_s * (qj * qk + qi * qr), 1 - two_s * (qi * qi + qj * qj)], -1).reshape(N, H, W, 3, 3)
    RS = R * sc[..., None, :]
    cov_r = RS @ RS.transpose(-1, -2)
    sh_r = gr[..., 8:].reshape(N, H, W, 3, nsh) * mask
    raw_r = torch.cat([centers, gr], -1)
    tol = 2e-3 if dt == torch.float16 else 1.6e-2
    for name, a_, b_ in (("means", means, centers), ("cov", cov, cov_r), ("sh", sh, sh_r), ("op", op, o), ("raw", raw, raw_r)):
        assert (a_ - b_.detach()).abs().max() <= 2e-5 * b_.detach().abs().max() + 1e-6, name
    ws = [torch.randn_like(t) for t in (centers, cov_r, sh_r, o, raw_r)]
    ws[1] = ws[1] * 30.0                         # covariances are ~1e-5..1e-1: weight them up so that every path matters
    loss = sum((a_ * w_).sum() for a_, w_ in zip((means, cov, sh, op, raw), ws))
    loss_r = sum((a_ * w_).sum() for a_, w_ in zip((centers, cov_r, sh_r, o, raw_r), ws))
    g_pts, g_gs = torch.autograd.grad(loss, (pts, gs), retain_graph=True)
    r_pts, r_gs = torch.autograd.grad(loss_r, (pr, gr), retain_graph=True)
    assert float(g_pts[..., 3].abs().max()) == 0
    assert (g_pts.float() - r_pts).abs().max() <= tol * r_pts.abs().max()
    for name, sl in (("opacity", slice(0, 1)), ("scales", slice(1, 4)), ("rotation", slice(4, 8)), ("harmonics", slice(8, None))):
        e = (g_gs.float()[..., sl] - r_gs[..., sl]).abs().max()
        assert e <= tol * r_gs[..., sl].abs().max() + 1e-6, (name, float(e), float(r_gs[..., sl].abs().max()))
    # without a gradient for raw (the training step's case) and for only one output
    g2 = torch.autograd.grad((cov * ws[1]).sum(), gs, retain_graph=True)[0]
    r2 = torch.autograd.grad((cov_r * ws[1]).sum(), gr, retain_graph=True)[0]
    assert (g2.float() - r2).abs().max() <= tol * r2.abs().max() + 1e-6
    # scales / rotations (the reference's Gaussians carry them, gaussian_adapter.py:150-155): values, and gradients when a loss reads them
    assert (scales_o - sc.detach()).abs().max() <= 2e-5 * sc.detach().abs().max() + 1e-7
    assert (rot_o - rot.detach()).abs().max() <= 2e-5
    w_s, w_r = torch.randn_like(sc), torch.randn_like(rot)
    g3 = torch.autograd.grad((scales_o * w_s).sum() + (rot_o * w_r).sum() + (op * ws[3]).sum(), gs, retain_graph=True)[0]
    r3 = torch.autograd.grad((sc * w_s).sum() + (rot * w_r).sum() + (o * ws[3]).sum(), gr, retain_graph=True)[0]
    for name, sl in (("opacity", slice(0, 1)), ("scales", slice(1, 4)), ("rotation", slice(4, 8))):
        assert (g3.float()[..., sl] - r3[..., sl]).abs().max() <= tol * r3[..., sl].abs().max() + 1e-6, name


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_gelu_backward_matches_autograd(dt):
    from vicasplat_amd import ops
    d = _dev()
    torch.manual_seed(3)
    z = (torch.randn(1000, 64, device=d) * 2).to(dt)
    dy = torch.randn(1000, 64, device=d).to(dt)
    zr = z.float().requires_grad_()
    (F.gelu(zr) * dy.float()).sum().backward()
    dz = ops.gelu_backward(dy, z)
    assert (dz.float() - zr.grad).abs().max() <= (2e-3 if dt == torch.float16 else 1.2e-2) * zr.grad.abs().max()


@pytest.mark.parametrize("do_dt", [torch.float32, torch.float16])
@pytest.mark.parametrize("M,C,gi", [(1028, 1024, 257), (600, 768, 200), (37, 192, 37)])
def test_layernorm_backward_matches_autograd(do_dt, M, C, gi):
    """LayerNorm + AdaLN modulation backward (dx, dw, db, dscale, dshift), incl. the remapped output rows."""
    from vicasplat_amd import ops
    d = _dev()
    g = torch.Generator(device="cpu").manual_seed(M + C)
    x = (torch.randn(M, C, generator=g) * 2 + 0.5).to(d)
    w = (torch.randn(C, generator=g) * 0.2 + 1).to(d)
    b = (torch.randn(C, generator=g) * 0.1).to(d)
    G = (M + gi - 1) // gi
    scale = (torch.randn(G, C, generator=g) * 0.3).to(d)
    shift = (torch.randn(G, C, generator=g) * 0.3).to(d)
    rows = torch.arange(M, device=d)
    dout_full = torch.randn(G * (gi + 1), C, generator=g).to(d).to(do_dt)           # forward wrote rows g*(gi+1) + 1 + m%gi
    orow = (rows // gi) * (gi + 1) + 1 + rows % gi
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    sr, hr = scale.clone().requires_grad_(), shift.clone().requires_grad_()
    y = F.layer_norm(xr, (C,), wr, br, 1e-6) * (1 + sr[rows // gi]) + hr[rows // gi]
    (y * dout_full[orow].float()).sum().backward()
    dx, dw, db, dsc, dsh = ops.layernorm_backward(dout_full, x, w, b, scale=scale, mod_rows=gi, grp_in=gi, grp_out=gi + 1, grp_off=1)
    tol = 2e-4 if do_dt == torch.float32 else 2e-3
    for got, ref, nm in ((dx, xr.grad, "dx"), (dw, wr.grad, "dw"), (db, br.grad, "db"), (dsc, sr.grad, "dscale"), (dsh, hr.grad, "dshift")):
        assert (got - ref).abs().max() <= tol * ref.abs().max() + 1e-5, nm
    # plain LayerNorm (no modulation), accumulate into an existing dx
    xr2 = x.clone().requires_grad_()
    dplain = dout_full[:M].float()
    (F.layer_norm(xr2, (C,), w, b, 1e-6) * dplain).sum().backward()
    base = torch.ones(M, C, device=d)
    dx2, _, _, n1, n2 = ops.layernorm_backward(dout_full[:M], x, w, b, dx=base.clone(), accumulate_dx=True)
    assert n1 is None and n2 is None
    assert (dx2 - base - xr2.grad).abs().max() <= tol * xr2.grad.abs().max() + 1e-5


def _attn_ref(q, k, v, keymask, scale):
    """q [Lq,H,64], k/v [Lk,H,64] f32, keymask [Lq,Lk] bool -> out [Lq,H,64] (plain softmax attention)."""
    s = torch.einsum("qhd,khd->hqk", q, k) * scale
    s = s.masked_fill(~keymask[None], float("-inf"))
    return torch.einsum("hqk,khd->qhd", torch.softmax(s, -1), v)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("case", ["encoder", "video_prefix", "neighbour_segments"])
def test_attention_backward_matches_autograd(dt, case):
    """dq / dk / dv of the three attention shapes (plain, key-prefix mask, two gathered key segments) vs torch autograd."""
    from vicasplat_amd import ops
    d = _dev()
    g = torch.Generator(device="cpu").manual_seed(len(case))
    H, C = 3, 192
    if case == "encoder":
        nb, Lq = 3, 257
        rows = nb * Lq
    elif case == "video_prefix":
        nb, Lq = 2, 3 * 86          # 3 frames of 86 tokens per scene; token 0 of a frame sees the frames up to its own
        rows = nb * Lq
    else:
        nb, Lq = 4, 100            # 4 frames; frame t attends to frames t-1 and t+1 (clamped like backbone_vica.py:172-186)
        rows = nb * Lq
    qkv = (torch.randn(rows, 3 * C, generator=g) * 0.7).to(dt).to(d)
    dout = torch.randn(rows, C, generator=g).to(dt).to(d)
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
    out = torch.empty(rows, C, dtype=dt, device=d)
    lse = torch.empty(rows, H, dtype=torch.float32, device=d)
    ops.attention(q2, k2, v2, out, lse=lse, **kw)
    dq, dk, dv = ops.attention_backward(q2, k2, v2, out, dout, lse, max_keys=2 * Lq if seg is not None else 0, **kw)
    # reference
    qf = qkv.float().clone().requires_grad_()
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
    (ref_out * dout.float()).sum().backward()
    gq, gk, gv = qf.grad[:, :C], qf.grad[:, C:2 * C], qf.grad[:, 2 * C:]
    rt = 6e-3 if dt == torch.float16 else 3e-2
    assert (out.float() - ref_out).abs().max() <= rt * ref_out.abs().max()
    # lse: log2-domain logsumexp of the scaled scores
    assert torch.isfinite(lse).all()
    for got, ref, nm in ((dq.float(), gq, "dq"), (dk, gk, "dk"), (dv, gv, "dv")):
        assert (got - ref).abs().max() <= rt * ref.abs().max(), (nm, float((got - ref).abs().max()), float(ref.abs().max()))
    if seg is None:   # direct mode (vs_attention_backward16): dk / dv stored as 16-bit values into the k | v blocks of a packed gradient buffer
        dqkv = torch.full((rows, 3 * C), float("nan"), dtype=dt, device=d)
        ops.attention_backward(q2, k2, v2, out, dout, lse, dq_out=dqkv[:, :C], dk_out=dqkv[:, C:2 * C], dv_out=dqkv[:, 2 * C:], **kw)
        assert torch.equal(dqkv[:, :C], dq)
        for got, f32, nm in ((dqkv[:, C:2 * C], dk, "dk16"), (dqkv[:, 2 * C:], dv, "dv16")):
            assert torch.equal(got, f32.to(dt)), nm        # one owner per row: the same sums in the same order, rounded once


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("N,H,W,Cin,Cout,relu_in", [(2, 16, 16, 64, 128, False), (1, 37, 21, 128, 256, True), (3, 64, 64, 256, 256, True),
                                                    (2, 13, 7, 256, 256, False), (1, 5, 100, 256, 512, True), (5, 8, 8, 512, 256, False), (2, 20, 12, 256, 128, True),
                                                    (1, 9, 33, 192, 256, False), (2, 11, 19, 128, 128, True), (1, 30, 8, 128, 256, False), (3, 6, 6, 64, 256, True)])
def test_conv3x3_backward_matches_autograd(dt, N, H, W, Cin, Cout, relu_in):
    from vicasplat_amd import ops
    d = _dev()
    torch.manual_seed(N * H + Cin)
    x = torch.randn(N, H, W, Cin, device=d).to(dt)
    conv = torch.nn.Conv2d(Cin, Cout, 3, 1, 1).to(d)
    wp = ops.pack_conv3x3_weight(conv.weight, dt)
    dy = (torch.randn(N, H, W, Cout, device=d) * 0.5).to(dt)
    xr = x.float().permute(0, 3, 1, 2).clone().requires_grad_()
    wr = wp.float().permute(0, 3, 1, 2).clone().requires_grad_()   # [Cout,Cin,3,3] from the 16-bit packed weights
    br = conv.bias.detach().clone().requires_grad_()
    y = F.conv2d(F.relu(xr) if relu_in else xr, wr, br, padding=1)
    (y * dy.float().permute(0, 3, 1, 2)).sum().backward()
    dx, dw, db = ops.conv3x3_backward(dy, x, wp, relu_in=relu_in)
    rt = 4e-3 if dt == torch.float16 else 2.5e-2
    gx = xr.grad.permute(0, 2, 3, 1)
    assert (dx.float() - gx).abs().max() <= rt * gx.abs().max()
    gw = wr.grad.permute(0, 2, 3, 1)
    assert (dw - gw).abs().max() <= 3e-3 * gw.abs().max() + 1e-3
    assert (db - br.grad).abs().max() <= 1e-3 * br.grad.abs().max() + 1e-2


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 16, 16, 128, 128), (1, 7, 5, 64, 192), (3, 4, 4, 192, 64)])
def test_conv3x3_stride2_function_matches_autograd(N, H, W, Cin, Cout):
    """autograd.Conv3x3Fn(stride=2) (the reassemble conv of dpt_block.py act_postprocess[3]): forward on the HIP conv, backward
    = the stride-1 backward of the zero-dilated gradient; both against torch's f32 conv on the 16-bit rounded operands."""
    from vicasplat_amd import autograd as A
    d = _dev()
    dt = torch.float16
    torch.manual_seed(H * W + Cin)
    conv = torch.nn.Conv2d(Cin, Cout, 3, 2, 1).to(d)
    x = torch.randn(N, H, W, Cin, device=d).to(dt).requires_grad_()
    w = conv.weight.detach().to(dt).float().requires_grad_()
    b = conv.bias.detach().clone().requires_grad_()
    y = A.conv3x3(x, w, b, stride=2)
    xr = x.detach().float().permute(0, 3, 1, 2).clone().requires_grad_()
    wr = w.detach().clone().requires_grad_()
    br = b.detach().clone().requires_grad_()
    yr = F.conv2d(xr, wr, br, stride=2, padding=1)
    assert y.shape == (N, (H + 1) // 2, (W + 1) // 2, Cout)
    assert (y.float() - yr.permute(0, 2, 3, 1)).abs().max() <= 4e-3 * yr.abs().max()
    dy = (torch.randn_like(yr) * 0.5).to(dt)
    (y.float() * dy.float().permute(0, 2, 3, 1)).sum().backward()
    (yr * dy.float()).sum().backward()
    gx = xr.grad.permute(0, 2, 3, 1)
    assert (x.grad.float() - gx).abs().max() <= 4e-3 * gx.abs().max()
    assert (w.grad - wr.grad).abs().max() <= 3e-3 * wr.grad.abs().max() + 1e-3
    assert (b.grad - br.grad).abs().max() <= 1e-3 * br.grad.abs().max() + 1e-2
    y2 = A.conv3x3(x.detach(), w.detach(), b.detach(), stride=2)
    assert torch.equal(y2, y.detach())


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_upsample2x_backward_matches_autograd(dt):
    from vicasplat_amd import ops
    d = _dev()
    torch.manual_seed(4)
    for (N, H, W, C) in [(2, 8, 8, 64), (1, 5, 7, 16), (1, 1, 3, 8), (2, 64, 64, 256)]:
        dy = torch.randn(N, 2 * H, 2 * W, C, device=d).to(dt)
        x = torch.zeros(N, C, H, W, device=d, requires_grad=True)
        (F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True) * dy.float().permute(0, 3, 1, 2)).sum().backward()
        got = ops.upsample2x_backward_nhwc(dy).permute(0, 3, 1, 2).float()
        tol = 2e-3 if dt == torch.float16 else 1.6e-2
        assert (got - x.grad).abs().max() <= tol * x.grad.abs().max()


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("Cin,Cout,C2", [(256, 256, 83), (128, 128, 3), (256, 256, 16), (64, 128, 4)])
def test_conv3x3_head1x1_fused_matches_the_two_layer_path(dt, Cin, Cout, C2):
    """conv3 -> ReLU -> conv1 of the DPT heads in one kernel (vs_conv3x3_head1x1_nhwc) against the unfused pair of HIP kernels
    (same 16-bit rounding of the intermediate activation) and against fp32 torch."""
    from vicasplat_amd import ops
    d = _dev()
    g = torch.Generator(device="cpu").manual_seed(Cin + Cout + C2)
    N, H, W = 2, 32, 48
    x = torch.randn(N, H, W, Cin, generator=g).to(dt).to(d)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(d)
    b = (torch.randn(Cout, generator=g) * 0.1).to(d) if Cout == 128 else None
    w2 = (torch.randn(C2, Cout, generator=g) / math.sqrt(Cout)).to(d)
    b2 = torch.randn(C2, generator=g).to(d)
    wp = ops.pack_conv3x3_weight(w, dt)
    pad = (C2 + 15) // 16 * 16 if Cout == 256 else 4
    w2p = torch.zeros(pad, Cout, dtype=dt, device=d); w2p[:C2] = w2.to(dt)
    b2p = torch.zeros(pad, device=d); b2p[:C2] = b2
    y = ops.conv3x3_head1x1_nhwc(x, wp, b, w2p, b2p, C2)
    assert y.shape == (N, H, W, pad)
    mid = ops.conv3x3_nhwc(x, wp, b, relu_out=True)
    two = torch.empty(N * H * W, C2, dtype=dt, device=d)
    if C2 % 8 == 0:
        ops.gemm(mid.view(-1, Cout), w2p[:C2].contiguous(), b2, two, ops.EPI_STORE16)
    else:
        two = (mid.view(-1, Cout).float() @ w2p[:C2].float().t() + b2).to(dt)
    tol = (2e-3 if dt == torch.float16 else 1.6e-2) * float(two.float().abs().max())
    assert (y[..., :C2].reshape(-1, C2).float() - two.float()).abs().max() <= tol
    if pad > C2 and Cout == 256:
        assert float(y[..., C2:].abs().max()) == 0.0
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.to(dt).float(), None if b is None else b, padding=1).relu().permute(0, 2, 3, 1)
    ref = ref.reshape(-1, Cout) @ w2.to(dt).float().t() + b2
    assert (y[..., :C2].reshape(-1, C2).float() - ref).abs().max() <= 4 * tol


def test_linear_f32_and_silu_cast():
    """The camera-token path's tiny f32 layers and the AdaLN SiLU on our own kernels (no vendor BLAS launch in the inference step)."""
    from vicasplat_amd import ops
    d = _dev()
    g = torch.Generator(device="cpu").manual_seed(1)
    for M, N, K, relu in ((192, 1024, 9, False), (168, 8, 768, True), (3, 2, 192, True), (1, 5, 70, False)):
        x = torch.randn(M, K, generator=g).to(d); w = torch.randn(N, K, generator=g).to(d); b = torch.randn(N, generator=g).to(d)
        ref = (x.relu() if relu else x).double() @ w.double().t() + b.double()
        got = ops.linear_f32(x, w, b, relu_in=relu)
        assert got.shape == (M, N) and (got.double() - ref).abs().max() <= 1e-5 * ref.abs().max()
    x = torch.randn(24, 768, generator=g).to(d) * 3
    for dt, tol in ((torch.float32, 1e-6), (torch.float16, 1e-3), (torch.bfloat16, 8e-3)):
        y = ops.silu_cast(x, dt)
        assert y.dtype == dt and (y.float() - F.silu(x)).abs().max() <= tol * F.silu(x).abs().max()


def test_upsample2x_add_relu_function_matches_autograd():
    from vicasplat_amd import autograd as A
    d = torch.device("cuda:0")
    torch.manual_seed(3)
    x = torch.randn(2, 12, 10, 16, device=d).half().requires_grad_(True)
    s = torch.randn(2, 24, 20, 16, device=d).half().requires_grad_(True)
    y = A.upsample2x_add_relu(x, s)
    g = torch.randn_like(y)
    y.backward(g)
    xr, sr = x.detach().float().requires_grad_(True), s.detach().float().requires_grad_(True)
    yr = torch.nn.functional.interpolate(xr.permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True).permute(0, 2, 3, 1) + torch.relu(sr)
    yr.backward(g.float())
    assert float((y.float() - yr).abs().max()) <= 4e-3
    assert float((x.grad.float() - xr.grad).abs().max()) <= 2e-2 * max(1.0, float(xr.grad.abs().max()))
    assert torch.equal(s.grad.float(), sr.grad.half().float())
