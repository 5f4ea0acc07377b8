"""One frame-encoder transformer block differentiated by hand on the HIP backward operators of `ops` -- LayerNorm -> packed
qkv projection + RoPE -> attention -> projection (+residual) -> LayerNorm -> fc1 -> GELU -> fc2 (+residual), i.e.
croco/blocks.py:114-130 as `VicaNet` runs it (backbone_vica.py:455-470) -- with an explicit tape instead of autograd.
13 gradients vs the oracle's autograd to 7e-4 in f16 (tests/test_train_gpu.py).  `autograd.EncBlockFn` wraps the pair as ONE
autograd node per encoder block for the training forward (`model/encoder/train_forward.py`): the residual adds live in the GEMM
epilogues and the LayerNorm backward's accumulate, so no elementwise PyTorch kernel runs inside a block in either direction.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn.functional as F

from . import ops


@dataclass
class EncBlockParams:
    """One encoder block: 16-bit GEMM operands, f32 everything else (the layout VicaNet caches for inference)."""
    ln1_w: torch.Tensor; ln1_b: torch.Tensor
    qkv_w: torch.Tensor; qkv_b: torch.Tensor      # [3C, C] 16-bit, [3C] f32
    proj_w: torch.Tensor; proj_b: torch.Tensor    # [C, C], [C]
    ln2_w: torch.Tensor; ln2_b: torch.Tensor
    fc1_w: torch.Tensor; fc1_b: torch.Tensor      # [4C, C], [4C]
    fc2_w: torch.Tensor; fc2_b: torch.Tensor      # [C, 4C], [C]


def enc_block_forward_train(x: torch.Tensor, p: EncBlockParams, pos: torch.Tensor, *, frames: int, tokens: int, heads: int,
                            rope_base: float = 100.0, eps: float = 1e-6):
    """x f32 [frames*tokens, C] (residual stream) -> (x_out f32, tape).  Same kernels as inference; the tape keeps what
    the backward needs: both residual inputs, both LayerNorm outputs, qkv (after RoPE), the attention output + its
    logsumexp, and the fc1 pre-activation."""
    M, C = x.shape
    dt = p.qkv_w.dtype
    dev = x.device
    h1 = torch.empty(M, C, dtype=dt, device=dev)
    ops.layernorm_mod(x, p.ln1_w, p.ln1_b, h1, eps=eps)
    qkv = torch.empty(M, 3 * C, dtype=dt, device=dev)
    ops.gemm_qkv_rope(h1, p.qkv_w, p.qkv_b, qkv, C, pos, None, rope_base, 1.0)
    att = torch.empty(M, C, dtype=dt, device=dev)
    lse = torch.empty(M, heads, dtype=torch.float32, device=dev)
    ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], att, nbatch=frames, H=heads, Lq=tokens, Lk=tokens,
                  q_batch_rows=tokens, k_batch_rows=tokens, lse=lse)
    x_mid = ops.gemm_resid(att, p.proj_w, p.proj_b, x)      # x + proj(att) into a new buffer: x stays for the LayerNorm backward
    h2 = torch.empty(M, C, dtype=dt, device=dev)
    ops.layernorm_mod(x_mid, p.ln2_w, p.ln2_b, h2, eps=eps)
    z = torch.empty(M, p.fc1_w.shape[0], dtype=dt, device=dev)
    ops.gemm(h2, p.fc1_w, p.fc1_b, z, ops.EPI_STORE16)       # pre-activation kept for the GELU backward
    a = ops.gelu16(z)
    x_out = ops.gemm_resid(a, p.fc2_w, p.fc2_b, x_mid)
    tape = dict(x=x, h1=h1, qkv=qkv, att=att, lse=lse, x_mid=x_mid, h2=h2, z=z, a=a, pos=pos, frames=frames, tokens=tokens,
                heads=heads, rope_base=rope_base, eps=eps)
    return x_out, tape


def enc_block_backward(dx_out: torch.Tensor, tape: dict, p: EncBlockParams):
    """dx_out f32 [M,C] = dL/d(block output) -> (dx_in f32 [M,C], dict of parameter gradients, all f32)."""
    t = tape
    C = dx_out.shape[1]
    dt = p.qkv_w.dtype
    g = {}
    # ---- MLP: x_out = x_mid + fc2(gelu(fc1(LN2(x_mid)))) ----
    da, g["fc2_w"], g["fc2_b"] = ops.linear_backward(dx_out.to(dt), t["a"], p.fc2_w)
    dz = ops.gelu_backward(da, t["z"])
    dh2, g["fc1_w"], g["fc1_b"] = ops.linear_backward(dz, t["h2"], p.fc1_w)
    # dx_mid = dx_out (residual path, read from its own buffer: no clone) + LayerNorm-2 gradient; that ONE f32 buffer then collects
    # the LayerNorm-1 gradient too, and the kernel also emits the 16-bit copy that the projection's backward GEMMs read (no cast pass)
    dx_mid = torch.empty_like(dx_out)
    dx_mid16 = torch.empty(dx_out.shape, dtype=dt, device=dx_out.device)
    _, g["ln2_w"], g["ln2_b"], _, _ = ops.layernorm_backward(dh2, t["x_mid"], p.ln2_w, p.ln2_b, eps=t["eps"], dx=dx_mid, dx_add=dx_out,
                                                             dx16=dx_mid16)
    # ---- attention: x_mid = x + proj(attn(rope(qkv(LN1(x))))) ----
    dx_in = dx_mid
    datt, g["proj_w"], g["proj_b"] = ops.linear_backward(dx_mid16, t["att"], p.proj_w)
    qkv = t["qkv"]
    dqkv = torch.empty_like(qkv)                              # dq lands in its block directly; dk / dv (f32) are cast into theirs
    _, dk, dv = ops.attention_backward(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], t["att"], datt, t["lse"], nbatch=t["frames"],
                                       H=t["heads"], Lq=t["tokens"], Lk=t["tokens"], q_batch_rows=t["tokens"],
                                       k_batch_rows=t["tokens"], dq_out=dqkv[:, :C])
    dqkv[:, C:2 * C] = dk
    dqkv[:, 2 * C:] = dv
    ops.rope_qk(dqkv, t["heads"], C, t["pos"], None, t["rope_base"], 1.0, inverse=True)   # backward of the rotation on dq | dk
    dh1, g["qkv_w"], g["qkv_b"] = ops.linear_backward(dqkv, t["h1"], p.qkv_w)
    _, g["ln1_w"], g["ln1_b"], _, _ = ops.layernorm_backward(dh1, t["x"], p.ln1_w, p.ln1_b, eps=t["eps"], dx=dx_in, accumulate_dx=True)
    return dx_in, g
