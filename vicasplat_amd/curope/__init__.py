"""Drop-in for the reference's `curope` extension + its autograd wrapper
(/root/reference/src/model/encoder/backbone/croco/curope/curope2d.py:12-40, curope.cpp:49-69).

`rope_2d(tokens, positions, base, fwd)` rotates IN PLACE a [B,N,H,D] view (stride(3)==1, stride(2)==D);
`cuRoPE2D(freq, F0)(tokens[B,H,N,D], positions[B,N,2])` is the nn.Module the backbone instantiates.
"""
from __future__ import annotations

import torch

from .. import _lib as L

_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def rope_2d(tokens: torch.Tensor, positions: torch.Tensor, base: float, fwd: float) -> None:
    if tokens.dim() != 4:
        raise RuntimeError("tokens must have 4 dimensions")
    if positions.dim() != 3:
        raise RuntimeError("positions must have 3 dimensions")
    B, N, H, D = tokens.shape
    if positions.shape[0] != B or positions.shape[1] != N:
        raise RuntimeError("batch size / number of tokens differ between tokens & positions")
    if positions.shape[2] != 2:
        raise RuntimeError("positions.shape[2] must be equal to 2")
    if tokens.stride(3) != 1 or tokens.stride(2) != D:
        raise RuntimeError("tokens are not contiguous along the last two dimensions")
    if D % 4 != 0:
        raise RuntimeError("token dim must be multiple of 4")
    if tokens.dtype not in _DT:
        raise RuntimeError(f"unsupported dtype {tokens.dtype}")
    dev = L.require_device(tokens, positions)
    pos = positions.to(torch.int64).contiguous()
    with torch.cuda.device(dev):
        rc = L.lib().vs_rope2d(L.ptr(tokens), L.ptr(pos), B, N, H, D, tokens.stride(0), tokens.stride(1), float(base),
                               float(fwd), _DT[tokens.dtype], L.stream_ptr(dev))
    L.check(rc, "vs_rope2d")


class _RoPE2DFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tokens, positions, base, F0=1.0):
        ctx.save_for_backward(positions)
        ctx.saved_base, ctx.saved_F0 = base, F0
        rope_2d(tokens, positions, base, F0)
        ctx.mark_dirty(tokens)
        return tokens

    @staticmethod
    def backward(ctx, grad):
        (positions,) = ctx.saved_tensors
        grad = grad.contiguous() if (grad.stride(3) != 1 or grad.stride(2) != grad.shape[3]) else grad
        rope_2d(grad, positions, ctx.saved_base, -ctx.saved_F0)
        ctx.mark_dirty(grad)
        return grad, None, None, None


class cuRoPE2D(torch.nn.Module):
    def __init__(self, freq: float = 100.0, F0: float = 1.0):
        super().__init__()
        self.base, self.F0 = freq, F0

    def forward(self, tokens: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        # tokens [B,H,N,D] -> operate on the [B,N,H,D] view, as the reference does (curope2d.py:39)
        _RoPE2DFn.apply(tokens.transpose(1, 2), positions, self.base, self.F0)
        return tokens
