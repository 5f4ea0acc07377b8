"""Training groundwork: one ViT-L frame-encoder block differentiated on the HIP backward operators vs torch autograd over
the oracle's restatement of the same block (oracle/encoder_ref.py: ln / lin / rope2d / sdpa / mlp)."""
import math

import pytest
import torch

from oracle import encoder_ref as er

pytestmark = pytest.mark.gpu


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
