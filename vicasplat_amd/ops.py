"""Python front-ends of the ViT-block HIP operators (thin ctypes wrappers over include/vicasplat_hip.h).

Every function requires HIP device tensors and raises otherwise -- there is no PyTorch/CPU fallback path.
16-bit operand dtype: torch.float16 (default; 10-bit mantissa like the TF32 the reference runs at) or torch.bfloat16.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import _lib as L

_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
# operand dtype codes of the MFMA kernels (GEMM / convolution / attention / upsample): 3 = f32, the reference-precision path --
# fp32 weights AND activations on the exact f32 MFMA (csrc/gemm_common.h, kDtF32); every "16-bit" output is then f32 too
_DTX = {torch.float32: 3, torch.float16: 1, torch.bfloat16: 2}
_OPERAND_DTYPES = (torch.float16, torch.bfloat16, torch.float32)
EPI_STORE16, EPI_GELU16, EPI_RESID32, EPI_STORE32 = 0, 1, 2, 3


class _RangeGuard:
    """Debug-mode range audit of the split operand class (csrc/range_guard.hip; VERDICT r3 "weak 3").  The split class keeps f32
    activations but multiplies their f16 (hi, lo) pairs: |x| >= 65520 becomes hi = +-inf, where the reference's fp32 / TF32 arithmetic still
    has range.  While enabled, every split-class GEMM / convolution / attention front-end below scans its activation operand
    (vs_range_check: one extra read pass, one atomic per wavefront that finds something) into one flag word per call; `report()` reads
    the table (one host synchronisation) and names the calls that saw out-of-range values.  Disabled (the default) it costs nothing."""
    LIMIT = 65520.0          # the smallest magnitude that rounds to +-inf in f16 (round to nearest even)
    SLOTS = 4096

    def __init__(self):
        self.enabled, self.flags, self.names = False, None, []

    def begin(self, device):
        self.enabled = True
        if self.flags is None or self.flags.device != torch.device(device):
            self.flags = torch.zeros(self.SLOTS, dtype=torch.int32, device=device)
        else:
            self.flags.zero_()
        self.names = []

    def end(self):
        self.enabled = False

    def check(self, what: str, t, rows=None, cols=None):
        """t: an f32 2-D view [rows, cols] (any row stride) / an N-D contiguous f32 tensor (flattened to rows of its last dimension) / a
        packed SplitWeight activation."""
        if not self.enabled or t is None:
            return
        packed = isinstance(t, SplitWeight)
        d = t.data if packed else t
        if d.dtype != (torch.int32 if packed else torch.float32) or d.numel() == 0:
            return
        if d.dim() != 2:
            d = d.reshape(-1, d.shape[-1])
        rows = d.shape[0] if rows is None else rows
        cols = d.shape[1] if cols is None else cols
        if d.stride(1) != 1 or cols % (32 if packed else 4) or d.stride(0) % 4 or d.data_ptr() % 16:
            d = d[:, :cols].contiguous() if not packed else d
            if packed:
                return
            pad = (-cols) % 4
            if pad:
                d = torch.nn.functional.pad(d, (0, pad))
                cols += pad
        slot = len(self.names)
        if slot >= self.SLOTS:
            return
        self.names.append(f"#{slot} {what} [{rows} x {cols}]{' packed' if packed else ''}")
        dev = d.device
        with torch.cuda.device(dev):
            rc = L.lib().vs_range_check(L.ptr(d), rows, cols, d.stride(0), 1 if packed else 0, self.LIMIT, L.ptr(self.flags), slot,
                                        L.stream_ptr(dev))
        L.check(rc, "vs_range_check")

    def report(self) -> list:
        """[(call, flags)] of the calls whose operand left the f16 range (1 = finite |x| >= 65520, 2 = non-finite f32 input, 4 = a packed hi
        half that is +-inf / NaN).  Synchronises."""
        if self.flags is None or not self.names:
            return []
        f = self.flags[:len(self.names)].cpu().tolist()
        return [(n, v) for n, v in zip(self.names, f) if v]


RANGE_GUARD = _RangeGuard()


class range_guard:
    """`with ops.range_guard(device) as g: ...; bad = g.report()` -- audit every split-class operand inside the block (see _RangeGuard)."""

    def __init__(self, device, raise_on_overflow: bool = False):
        self.device, self.raise_on_overflow = device, raise_on_overflow

    def __enter__(self):
        RANGE_GUARD.begin(self.device)
        return RANGE_GUARD

    def __exit__(self, et, ev, tb):
        RANGE_GUARD.end()
        if et is None and self.raise_on_overflow:
            bad = RANGE_GUARD.report()
            if bad:
                raise SplitRangeError(bad)
        return False


class SplitRangeError(RuntimeError):
    def __init__(self, bad):
        self.calls = bad
        head = "; ".join(f"{n} (flags {v})" for n, v in bad[:6])
        super().__init__(f"split operand class: {len(bad)} call(s) saw activations outside the f16 range of their hi halves (|x| >= 65520 -> "
                         f"+-inf; the reference's fp32 / TF32 arithmetic has fp32 range there): {head}{' ...' if len(bad) > 6 else ''}.  "
                         'Run this input in the exact-f32 class (set_compute_dtype("f32")) or rescale the offending layer.')


def split_act(rows: int, cols: int, device) -> "SplitWeight":
    """Uninitialised [rows, cols] activation buffer in the PACKED (hi, lo) form (scale 2^0) for producers that write it directly --
    layernorm_mod(out=...), gemm(..., out=...) with the store / GELU epilogues -- and GEMMs that read it as their A operand without a
    conversion (vs_gemm_split_packed).  cols % 32 == 0; 4 bytes per element like the f32 tensor it stands for."""
    assert cols % 32 == 0
    return SplitWeight(torch.empty((rows, cols), dtype=torch.int32, device=device), 1.0, (rows, cols))


class SplitWeight:
    """A weight packed for the SPLIT operand class (dtype code 4; csrc/gemm_common.h, kDtSplit): f16 (hi, lo) pairs of w * 2^scale_exp,
    blocked by 32 k (vs_split_pack_weight).  `data` is an int32 tensor of the logical shape (4 bytes per element); the product entry
    points multiply the accumulators by `acc_scale` = 2^-scale_exp.  Activations of this class are plain f32 tensors."""
    __slots__ = ("data", "acc_scale", "shape")

    def __init__(self, data: torch.Tensor, acc_scale: float, shape):
        self.data, self.acc_scale, self.shape = data, acc_scale, tuple(shape)

    @property
    def device(self):
        return self.data.device

    def is_contiguous(self):
        return self.data.is_contiguous()

    def numel(self):
        return self.data.numel()


def split_scale_exp(w: torch.Tensor, with_nonzero: bool = False):
    """Power-of-two exponent e with max|w| 2^e in [2^13, 2^14) (one host read of the maximum); 0 for an all-zero / non-finite weight.
    with_nonzero: also return whether the exponent came from an actual finite non-zero maximum (a cache must not keep the 0 of a
    zero-initialised layer: it would pack the layer unscaled long after its weights have moved)."""
    import math
    amax = float(w.detach().abs().max())
    ok = amax != 0.0 and math.isfinite(amax)
    e = max(-24, min(24, 13 - math.frexp(amax)[1] + 1)) if ok else 0
    return (e, ok) if with_nonzero else e


def split_pack_weight(w: torch.Tensor, scale_exp: Optional[int] = None) -> SplitWeight:
    """f32 weight [N, ...] (flattened to [N, K], K % 32 == 0) -> SplitWeight.  The power-of-two scale puts max|w| just below 2^14: lo stays a
    normal f16 for every element within 2^-16 of the largest, and the products of the f16 range cannot overflow the f32 accumulator."""
    dev = L.require_device(w)
    w2 = w.detach().float().reshape(w.shape[0], -1).contiguous()
    N, K = w2.shape
    assert K % 32 == 0, f"split operands need K % 32 == 0 (K={K}): pad the reduction dimension"
    # scale_exp given: no host synchronisation (callers that pack every step cache the exponent; 0 = unscaled, the activations' convention)
    e = split_scale_exp(w2) if scale_exp is None else int(scale_exp)
    out = torch.empty((N, K), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = L.lib().vs_split_pack_weight(L.ptr(w2), w2.stride(0), L.ptr(out), out.stride(0), N, K, e, L.stream_ptr(dev))
    L.check(rc, "vs_split_pack_weight")
    return SplitWeight(out.view(w.shape[0], *w.shape[1:]) if w.dim() > 2 else out, 2.0 ** (-e), w.shape)


def layernorm_mod(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, out: torch.Tensor, *, eps: float = 1e-6,
                  scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None, mod_rows: int = 0,
                  grp_in: int = 0, grp_out: int = 0, grp_off: int = 0) -> torch.Tensor:
    """x f32 [M,C] -> out [*,C] (f32/f16/bf16).  scale/shift: [G,C] f32 views (row stride taken from .stride(0))."""
    odt = None
    out_obj = out
    if isinstance(out, SplitWeight):      # packed (hi, lo) output of the split class: the consumer GEMM skips its conversion
        assert out.acc_scale == 1.0
        out, odt = out.data, 3
    dev = L.require_device(x, weight, bias, out, scale, shift)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    M, C = x.shape
    mod_ld = 0
    if scale is not None:
        assert scale.stride(-1) == 1 and (shift is None or shift.stride(0) == scale.stride(0))
        mod_ld = scale.stride(0)
    elif shift is not None:
        mod_ld = shift.stride(0)
    with torch.cuda.device(dev):
        rc = L.lib().vs_layernorm_mod(L.ptr(x), x.stride(0), L.ptr(weight), L.ptr(bias), L.ptr(scale), L.ptr(shift),
                                      mod_rows, mod_ld, L.ptr(out), out.stride(-2), _DT[out.dtype] if odt is None else odt, M, C, eps, grp_in,
                                      grp_out, grp_off, L.stream_ptr(dev))
    L.check(rc, "vs_layernorm_mod")
    return out_obj


def linear_f32(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], relu_in: bool = False) -> torch.Tensor:
    """[..., K] f32 @ w[N, K]^T + bias in f32 (vs_linear_f32): the tiny camera-token layers (intrinsic embedding, pose / fov heads)."""
    dev = L.require_device(x, w, bias)
    x2 = x.reshape(-1, x.shape[-1]).float()
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    wf = w.detach().float()
    if wf.stride(1) != 1:
        wf = wf.contiguous()
    bf = None if bias is None else bias.detach().float().contiguous()
    out = torch.empty((x2.shape[0], wf.shape[0]), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = L.lib().vs_linear_f32(L.ptr(x2), x2.stride(0), L.ptr(wf), wf.stride(0), L.ptr(bf), L.ptr(out), out.stride(0), x2.shape[0], wf.shape[0],
                                   x2.shape[1], int(relu_in), L.stream_ptr(dev))
    L.check(rc, "vs_linear_f32")
    return out.view(*x.shape[:-1], wf.shape[0])


def silu_cast(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """silu(x) of an f32 tensor in `dtype` (one pass; the GEMM operand of the AdaLN projections)."""
    dev = L.require_device(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.numel() % 4 == 0
    out = torch.empty(x.shape, dtype=dtype, device=dev)
    with torch.cuda.device(dev):
        rc = L.lib().vs_silu_cast(L.ptr(x), L.ptr(out), x.numel(), _DT[dtype], L.stream_ptr(dev))
    L.check(rc, "vs_silu_cast")
    return out


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, epilogue: int, *,
         gate: Optional[torch.Tensor] = None, gate_rows: int = 0, grp_in: int = 0, grp_out: int = 0, grp_off: int = 0,
         M: Optional[int] = None, a_grp_in: int = 0, a_grp_out: int = 0, a_grp_off: int = 0) -> torch.Tensor:
    """out = epilogue(a[M,K] @ w[N,K]^T + bias).  a, w 16-bit; out 16-bit (epilogue 0/1) or f32 (2: in-place residual
    update with optional per-group gate [G,N]; 3: store)."""
    if isinstance(w, SplitWeight):
        return _gemm_split(a, w, bias, out, epilogue, gate=gate, gate_rows=gate_rows, grp_in=grp_in, grp_out=grp_out, grp_off=grp_off, M=M,
                           a_grp_in=a_grp_in, a_grp_out=a_grp_out, a_grp_off=a_grp_off)
    dev = L.require_device(a, w, bias, out, gate)
    assert a.dtype == w.dtype and a.dtype in _OPERAND_DTYPES
    assert a.dim() == 2 and w.dim() == 2 and a.stride(1) == 1 and w.stride(1) == 1 and a.shape[1] == w.shape[1]
    K = a.shape[1]
    M = a.shape[0] if M is None else M  # with an input row map, M counts the rows actually consumed
    N = w.shape[0]
    if epilogue in (EPI_STORE16, EPI_GELU16):
        assert out.dtype == a.dtype
    else:
        assert out.dtype == torch.float32
    gate_ld = gate.stride(0) if gate is not None else 0
    with torch.cuda.device(dev):
        rc = L.lib().vs_gemm_bias_act(L.ptr(a), L.ptr(w), L.ptr(bias), L.ptr(out), L.ptr(gate), M, N, K, a.stride(0),
                                      w.stride(0), out.stride(-2), epilogue, _DTX[a.dtype], grp_in, grp_out, grp_off,
                                      gate_rows, gate_ld, a_grp_in, a_grp_out, a_grp_off, L.stream_ptr(dev))
    L.check(rc, "vs_gemm_bias_act")
    return out


def _gemm_split(a, w: SplitWeight, bias, out, epilogue, *, gate=None, gate_rows=0, grp_in=0, grp_out=0, grp_off=0, M=None, a_grp_in=0,
                a_grp_out=0, a_grp_off=0, resid=None, pos=None, kind=None, C=0, base2d=0.0, theta1d=0.0):
    """vs_gemm_split: a f32 [M,K], w SplitWeight [N,K], every output f32 (epilogue codes of `gemm`, 4 = q|k|v + RoPE)."""
    wd = w.data
    packed = isinstance(a, SplitWeight)       # A already in the packed (hi, lo) form (scale 2^0): the kernels skip their conversion
    if packed:
        assert a.acc_scale == 1.0
        a = a.data
    out_obj = out
    if isinstance(out, SplitWeight):          # packed output (epilogues 0 / 1 / 3): the A operand of the next GEMM, written by this one's epilogue
        assert out.acc_scale == 1.0 and epilogue in (EPI_STORE16, EPI_GELU16, EPI_STORE32, 4) and out.data.dtype == torch.int32
        out, epilogue = out.data.view(torch.float32), epilogue | 16
    dev = L.require_device(a, wd, bias, out, gate, resid, pos, kind)
    assert a.dtype == (torch.int32 if packed else torch.float32) and out.dtype == torch.float32 and a.dim() == 2 and wd.dim() == 2 and a.stride(1) == 1
    assert wd.stride(1) == 1 and a.shape[1] == wd.shape[1] and (resid is None or (resid.dtype == torch.float32 and resid.stride() == out.stride()))
    M = a.shape[0] if M is None else M
    if RANGE_GUARD.enabled:
        RANGE_GUARD.check(f"gemm A x W[{wd.shape[0]}] epi {epilogue}", SplitWeight(a, 1.0, a.shape) if packed else a)
    fn = L.lib().vs_gemm_split_packed if packed else L.lib().vs_gemm_split
    with torch.cuda.device(dev):
        rc = fn(L.ptr(a), L.ptr(wd), w.acc_scale, L.ptr(bias), L.ptr(out), L.ptr(gate), L.ptr(resid), M, wd.shape[0], a.shape[1],
                                   a.stride(0), wd.stride(0), out.stride(-2), epilogue, grp_in, grp_out, grp_off, gate_rows,
                                   gate.stride(0) if gate is not None else 0, a_grp_in, a_grp_out, a_grp_off, L.ptr(pos), L.ptr(kind), C,
                                   base2d, theta1d, L.stream_ptr(dev))
    L.check(rc, "vs_gemm_split")
    return out_obj


def gemm_resid(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], resid: torch.Tensor, *, gate: Optional[torch.Tensor] = None,
               gate_rows: int = 0) -> torch.Tensor:
    """resid [M,N] f32 + (1 + gate) * (a @ w^T + bias) into a NEW tensor (vs_gemm_resid): the residual update of gemm(..., EPI_RESID32)
    without touching (or cloning) the old stream."""
    if isinstance(w, SplitWeight):
        assert resid.is_contiguous() and resid.shape == (a.shape[0], w.shape[0])
        return _gemm_split(a, w, bias, torch.empty_like(resid), EPI_RESID32, gate=gate, gate_rows=gate_rows, resid=resid)
    dev = L.require_device(a, w, bias, resid, gate)
    assert a.dtype == w.dtype and a.dtype in _OPERAND_DTYPES and a.stride(1) == 1 and w.stride(1) == 1
    assert resid.dtype == torch.float32 and resid.is_contiguous() and resid.shape == (a.shape[0], w.shape[0])
    out = torch.empty_like(resid)
    with torch.cuda.device(dev):
        rc = L.lib().vs_gemm_resid(L.ptr(a), L.ptr(w), L.ptr(bias), L.ptr(resid), L.ptr(out), L.ptr(gate), a.shape[0], w.shape[0], a.shape[1],
                                   a.stride(0), w.stride(0), out.stride(0), _DTX[a.dtype], gate_rows, gate.stride(0) if gate is not None else 0,
                                   L.stream_ptr(dev))
    L.check(rc, "vs_gemm_resid")
    return out


def gemm_qkv_rope(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, C: int, pos: torch.Tensor,
                  kind: Optional[torch.Tensor] = None, base2d: float = 100.0, theta1d: float = 30.0, *, grp_in: int = 0,
                  grp_out: int = 0, grp_off: int = 0, M: Optional[int] = None, a_grp_in: int = 0, a_grp_out: int = 0,
                  a_grp_off: int = 0) -> torch.Tensor:
    """Packed q|k|v projection (w [N >= 2C, K]) with rope_qk(out, C // 64, C, pos, kind, ...) fused into the epilogue."""
    assert pos.dtype == torch.int32 and pos.is_contiguous() and (kind is None or (kind.dtype == torch.uint8 and kind.is_contiguous()))
    if isinstance(w, SplitWeight):
        return _gemm_split(a, w, bias, out, 4, grp_in=grp_in, grp_out=grp_out, grp_off=grp_off, M=M, a_grp_in=a_grp_in, a_grp_out=a_grp_out,
                           a_grp_off=a_grp_off, pos=pos, kind=kind, C=C, base2d=base2d, theta1d=theta1d)
    dev = L.require_device(a, w, bias, out, pos, kind)
    assert a.dtype == w.dtype == out.dtype and a.dtype in _OPERAND_DTYPES
    assert a.dim() == 2 and w.dim() == 2 and a.stride(1) == 1 and w.stride(1) == 1 and a.shape[1] == w.shape[1]
    assert pos.dtype == torch.int32 and pos.is_contiguous() and (kind is None or (kind.dtype == torch.uint8 and kind.is_contiguous()))
    M = a.shape[0] if M is None else M
    with torch.cuda.device(dev):
        rc = L.lib().vs_gemm_qkv_rope(L.ptr(a), L.ptr(w), L.ptr(bias), L.ptr(out), M, w.shape[0], a.shape[1], a.stride(0), w.stride(0),
                                      out.stride(-2), _DTX[a.dtype], grp_in, grp_out, grp_off, a_grp_in, a_grp_out, a_grp_off,
                                      L.ptr(pos), L.ptr(kind), C, base2d, theta1d, L.stream_ptr(dev))
    L.check(rc, "vs_gemm_qkv_rope")
    return out


def rope_qk(buf: torch.Tensor, H: int, k_col: int, pos: torch.Tensor, kind: Optional[torch.Tensor] = None,
            base2d: float = 100.0, theta1d: float = 30.0, inverse: bool = False) -> torch.Tensor:
    """In-place RoPE on q (col 0) and k (col k_col) of buf [rows, ld]; pos int32 [rows,2]; kind uint8 [rows] or None.
    inverse=True applies the inverse rotation (the backward pass of the embedding, on dq | dk)."""
    dev = L.require_device(buf, pos, kind)
    assert buf.dim() == 2 and buf.stride(1) == 1 and pos.dtype == torch.int32 and pos.is_contiguous()
    assert kind is None or (kind.dtype == torch.uint8 and kind.is_contiguous())
    with torch.cuda.device(dev):
        rc = L.lib().vs_rope_qk_dir(L.ptr(buf), buf.stride(0), buf.shape[0], H, k_col, L.ptr(pos), L.ptr(kind), base2d, theta1d,
                                    -1.0 if inverse else 1.0, _DT[buf.dtype], L.stream_ptr(dev))
    L.check(rc, "vs_rope_qk")
    return buf


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, *, nbatch: int, H: int, Lq: int,
              Lk: int = 0, q_batch_rows: int, k_batch_rows: int = 0, kv_seg: Optional[torch.Tensor] = None,
              q_kvlen: Optional[torch.Tensor] = None, scale: float = 0.125, lse: Optional[torch.Tensor] = None,
              split: bool = False) -> torch.Tensor:
    """q/k/v: 2-D views [rows, ld] whose column 0 is head 0 (e.g. slices of a packed q|k|v buffer); out [rows, H*64].
    lse (optional, f32 [rows, H] contiguous) receives the log2-domain logsumexp for attention_backward.
    split=True (f32 tensors only): the split operand class -- three f16 MFMAs per product instead of the exact-f32 MFMA (dtype code 4).
    q, k, v int32 (column slices of the `.data` of a packed (hi, lo) q|k|v buffer, written by gemm_qkv_rope(..., out=split_act(...))):
    the packed-input kernel (dtype 4 + 32: LDS-DMA staging without conversion, key-split waves, XCD-aware grid)."""
    out_obj, out_packed = out, isinstance(out, SplitWeight)
    in_packed = q.dtype == torch.int32
    if in_packed:
        assert split and k.dtype == torch.int32 and v.dtype == torch.int32
    if out_packed:          # split class: O written in the packed (hi, lo) form, the A operand of the projection GEMM as it is
        assert split and out.acc_scale == 1.0
        out = out.data.view(torch.float32)
    dev = L.require_device(q, k, v, out, kv_seg, q_kvlen, lse)
    for t in (q, k, v, out):
        assert t.dim() == 2 and t.stride(1) == 1 and (t.dtype == q.dtype or (in_packed and t is out))
    assert kv_seg is None or (kv_seg.dtype == torch.int32 and kv_seg.is_contiguous())
    assert q_kvlen is None or (q_kvlen.dtype == torch.int32 and q_kvlen.is_contiguous())
    assert lse is None or (lse.dtype == torch.float32 and lse.is_contiguous() and lse.shape == (q.shape[0], H))
    assert not split or q.dtype == torch.float32 or in_packed
    if split and RANGE_GUARD.enabled:
        for nm_, t_ in (("q", q), ("k", k), ("v", v)):
            RANGE_GUARD.check(f"attention {nm_} (H {H}, Lq {Lq})", SplitWeight(t_, 1.0, t_.shape) if in_packed else t_, cols=H * 64)
    with torch.cuda.device(dev):
        rc = L.lib().vs_attention_lse(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), nbatch, H, Lq, Lk, q_batch_rows, k_batch_rows,
                                      q.stride(0), k.stride(0), v.stride(0), out.stride(0), L.ptr(kv_seg), L.ptr(q_kvlen), scale,
                                      (4 + (16 if out_packed else 0) + (32 if in_packed else 0)) if split else _DTX[q.dtype], L.ptr(lse),
                                      L.stream_ptr(dev))
    L.check(rc, "vs_attention")
    return out_obj


def attention_backward(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, dout: torch.Tensor, lse: torch.Tensor, *,
                       nbatch: int, H: int, Lq: int, Lk: int = 0, q_batch_rows: int, k_batch_rows: int = 0,
                       kv_seg: Optional[torch.Tensor] = None, q_kvlen: Optional[torch.Tensor] = None, max_keys: int = 0,
                       scale: float = 0.125, dq_out: Optional[torch.Tensor] = None, dk_out: Optional[torch.Tensor] = None,
                       dv_out: Optional[torch.Tensor] = None):
    """Backward of `attention` (used by autograd.AttentionFn).  Returns dq (16-bit [rows, H*64]; written into dq_out, e.g. the q
    block of a packed [rows, 3*H*64] gradient buffer, when given) and dk, dv (f32 [key rows, H*64], accumulated from zero).
    Without key segments dk_out / dv_out (16-bit [key rows, H*64] views, e.g. the k | v blocks of the same packed buffer) may be
    given: they are written directly (vs_attention_backward16: no atomics, no zero fill, no cast pass) and returned."""
    dev = L.require_device(q, k, v, out, dout, lse, kv_seg, q_kvlen)
    for t in (q, k, v, out, dout):
        assert t.dim() == 2 and t.stride(1) == 1
    Cc = H * 64
    if dq_out is None:
        dq = torch.empty((q.shape[0], Cc), dtype=q.dtype, device=dev)
    else:
        dq = dq_out
        assert dq.shape == (q.shape[0], Cc) and dq.dtype == q.dtype and dq.stride(1) == 1
    delta = torch.empty((q.shape[0], H), dtype=torch.float32, device=dev)
    if dk_out is not None:
        assert kv_seg is None and dv_out is not None, "direct dk / dv need key rows with a single owner (no key segments)"
        for t, ref in ((dk_out, k), (dv_out, v)):
            assert t.shape == (ref.shape[0], Cc) and t.dtype == q.dtype and t.stride(1) == 1
        with torch.cuda.device(dev):
            rc = L.lib().vs_attention_backward16(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), L.ptr(dout), L.ptr(lse), L.ptr(delta), L.ptr(dq),
                                                 L.ptr(dk_out), L.ptr(dv_out), nbatch, H, Lq, Lk, q_batch_rows, k_batch_rows, q.stride(0),
                                                 k.stride(0), v.stride(0), out.stride(0), dout.stride(0), dq.stride(0), dk_out.stride(0),
                                                 dv_out.stride(0), L.ptr(q_kvlen), scale, _DT[q.dtype], L.stream_ptr(dev))
        L.check(rc, "vs_attention_backward16")
        return dq, dk_out, dv_out
    dk = torch.zeros((k.shape[0], Cc), dtype=torch.float32, device=dev)
    dv = torch.zeros((v.shape[0], Cc), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = L.lib().vs_attention_backward(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), L.ptr(dout), L.ptr(lse), L.ptr(delta), L.ptr(dq),
                                           L.ptr(dk), L.ptr(dv), nbatch, H, Lq, Lk, q_batch_rows, k_batch_rows, q.stride(0),
                                           k.stride(0), v.stride(0), out.stride(0), dout.stride(0), dq.stride(0), dk.stride(0),
                                           dv.stride(0), L.ptr(kv_seg), L.ptr(q_kvlen), max_keys, scale, _DT[q.dtype],
                                           L.stream_ptr(dev))
    L.check(rc, "vs_attention_backward")
    return dq, dk, dv


def gaussian_adapter(pts: torch.Tensor, gs: torch.Tensor, sh_mask: torch.Tensor, *, scale_act: str = "softplus",
                     scale_min: float = 0.0, scale_max: float = 0.0, opacity_exponent: float = 1.0, want_raw: bool = True):
    """pts [N,3,H,W], gs [N,8+3*d_sh,H,W] (any strides, f32/f16/bf16, same dtype) -> dict of f32 tensors shaped
    [N,H,W,...] (means, covariances, harmonics, opacities [...,1], scales, rotations, raw)."""
    dev = L.require_device(pts, gs, sh_mask)
    assert pts.dtype == gs.dtype and pts.dim() == 4 and gs.dim() == 4 and pts.shape[1] == 3
    N, Cg, H, W = gs.shape
    d_sh = (Cg - 8) // 3
    assert Cg == 8 + 3 * d_sh and sh_mask.numel() == d_sh and sh_mask.dtype == torch.float32
    # one flat pixel index must address both inputs: require (n,h,w) to be jointly contiguous in each (NCHW or NHWC)
    for t in (pts, gs):
        sn, sc, sh_, sw = t.stride()
        if not (sh_ == W * sw and sn == H * sh_):  # not pixel-linear (e.g. plain NCHW): make it channels-last
            raise RuntimeError("gaussian_adapter expects channels-last (NHWC) head outputs")
    f = dict(dtype=torch.float32, device=dev)
    npix = N * H * W
    out = dict(means=torch.empty(N, H, W, 3, **f), covariances=torch.empty(N, H, W, 3, 3, **f),
               harmonics=torch.empty(N, H, W, 3, d_sh, **f), opacities=torch.empty(N, H, W, 1, **f),
               scales=torch.empty(N, H, W, 3, **f), rotations=torch.empty(N, H, W, 4, **f),
               raw=torch.empty(N, H, W, 11 + 3 * d_sh, **f) if want_raw else None)
    act = {"bounded": 0, "exp": 1, "softplus": 2}[scale_act]
    with torch.cuda.device(dev):
        rc = L.lib().vs_gaussian_adapter(L.ptr(pts), pts.stride(3), pts.stride(1), L.ptr(gs), gs.stride(3), gs.stride(1),
                                         _DT[pts.dtype], npix, d_sh, L.ptr(sh_mask), act, scale_min, scale_max,
                                         opacity_exponent, L.ptr(out["means"]), L.ptr(out["covariances"]), L.ptr(out["harmonics"]),
                                         L.ptr(out["opacities"]), L.ptr(out["scales"]), L.ptr(out["rotations"]), L.ptr(out["raw"]),
                                         L.stream_ptr(dev))
    L.check(rc, "vs_gaussian_adapter")
    return out


def gaussian_adapter_backward(pts: torch.Tensor, gs: torch.Tensor, sh_mask: torch.Tensor, d_means: torch.Tensor, d_cov: torch.Tensor,
                              d_harm: torch.Tensor, d_op: torch.Tensor, d_raw: Optional[torch.Tensor] = None, *, scale_act: str = "softplus",
                              scale_min: float = 0.0, scale_max: float = 0.0, opacity_exponent: float = 1.0):
    """Backward of gaussian_adapter for dense NHWC 16-bit head outputs pts [N,H,W,pts_pix>=3], gs [N,H,W,8+3*d_sh]: f32 gradients of
    means [..,3], covariances [..,3,3], harmonics [..,3,d_sh], opacities [..] (and optionally raw [..,11+3*d_sh]) -> (d_pts, d_gs)
    in the inputs' shape and dtype, as views of buffers whose rows are padded to a multiple of 8 channels (16-byte aligned rows with
    zero padding: what the reduction-major weight-gradient kernel of the 1x1 convs in front of the adapter reads directly)."""
    dev = L.require_device(pts, gs, sh_mask, d_means, d_cov, d_harm, d_op, d_raw)
    assert pts.is_contiguous() and gs.is_contiguous() and pts.dtype == gs.dtype and pts.dtype in (torch.float16, torch.bfloat16, torch.float32)
    d_sh = (gs.shape[-1] - 8) // 3
    npix = gs.numel() // gs.shape[-1]
    f = lambda t: None if t is None else t.float().contiguous()
    d_means, d_cov, d_harm, d_op, d_raw = f(d_means), f(d_cov), f(d_harm), f(d_op), f(d_raw)
    assert d_means.numel() == npix * 3 and d_cov.numel() == npix * 9 and d_harm.numel() == npix * 3 * d_sh and d_op.numel() == npix
    assert d_raw is None or d_raw.numel() == npix * (11 + 3 * d_sh)
    pl, gl = (pts.shape[-1] + 7) // 8 * 8, (gs.shape[-1] + 7) // 8 * 8
    d_pts = torch.empty(pts.shape[:-1] + (pl,), dtype=pts.dtype, device=dev)
    d_gs = torch.empty(gs.shape[:-1] + (gl,), dtype=gs.dtype, device=dev)
    act = {"bounded": 0, "exp": 1, "softplus": 2}[scale_act]
    with torch.cuda.device(dev):
        rc = L.lib().vs_gaussian_adapter_backward(L.ptr(pts), pts.shape[-1], L.ptr(gs), _DT[pts.dtype], npix, d_sh, L.ptr(sh_mask), act,
                                                  scale_min, scale_max, opacity_exponent, L.ptr(d_means), L.ptr(d_cov), L.ptr(d_harm),
                                                  L.ptr(d_op), L.ptr(d_raw), L.ptr(d_pts), pl, L.ptr(d_gs), gl, L.stream_ptr(dev))
    L.check(rc, "vs_gaussian_adapter_backward")
    return d_pts[..., :pts.shape[-1]], d_gs[..., :gs.shape[-1]]


def conv3x3_nhwc(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
                 relu_in: bool = False, relu_out: bool = False, out: Optional[torch.Tensor] = None, stride: int = 1,
                 mask_by: Optional[torch.Tensor] = None, residual2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [N,H,W,Cin] contiguous 16-bit, w [Cout,3,3,Cin] (see pack_conv3x3_weight) -> [N,Ho,Wo,Cout] (k=3, pad=1).
    mask_by [N,Ho,Wo,Cout]: the result is zeroed where mask_by <= 0 (ReLU backward fused into a data-gradient conv)."""
    if mask_by is not None:
        assert residual is None and not relu_out
        residual, relu_out = mask_by, 2
    split = isinstance(w, SplitWeight)
    xin_packed = isinstance(x, SplitWeight)     # split class: the input already in the packed (hi, lo) form (its producer wrote it; ReLU applied there)
    if xin_packed:
        assert split and not relu_in and residual2 is None
        x = x.data
    dev = L.require_device(x, w.data if split else w, bias, residual, out)
    assert x.dim() == 4 and x.is_contiguous() and w.is_contiguous() and (x.dtype in _OPERAND_DTYPES or xin_packed)
    assert (split and x.dtype == (torch.int32 if xin_packed else torch.float32)) or (not split and x.dtype == w.dtype)
    N, H, W, Cin = x.shape
    Cout = w.shape[0]
    assert tuple(w.shape) == (Cout, 3, 3, Cin)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if out is None:
        out = torch.empty((N, Ho, Wo, Cout), dtype=torch.float32 if xin_packed else x.dtype, device=dev)
    assert residual is None or (residual.shape == out.shape and residual.is_contiguous() and residual.dtype == out.dtype)
    assert residual2 is None or (split and mask_by is None and residual2.shape == out.shape and residual2.is_contiguous() and residual2.dtype == torch.float32)
    if split and RANGE_GUARD.enabled:
        RANGE_GUARD.check(f"conv3x3 {Cin}->{Cout} @{H}x{W}", SplitWeight(x.view(-1, Cin), 1.0, (N * H * W, Cin)) if xin_packed else x)
    if split and residual2 is not None:      # split class: out = conv + bias + residual + residual2 in the epilogue (no add pass)
        L.require_device(residual2)
        with torch.cuda.device(dev):
            rc = L.lib().vs_conv3x3_split_res2_nhwc(L.ptr(x), L.ptr(w.data), w.acc_scale, L.ptr(bias), L.ptr(residual), L.ptr(residual2), L.ptr(out), N, H, W,
                                                    Cin, Cout, stride, int(relu_in), int(relu_out), L.stream_ptr(dev))
        L.check(rc, "vs_conv3x3_split_res2_nhwc")
        return out
    if split:
        with torch.cuda.device(dev):
            rc = L.lib().vs_conv3x3_split_nhwc(L.ptr(x), L.ptr(w.data), w.acc_scale, L.ptr(bias), L.ptr(residual), L.ptr(out), N, H, W, Cin, Cout,
                                               stride, int(relu_in) | (16 if xin_packed else 0), int(relu_out), L.stream_ptr(dev))
        L.check(rc, "vs_conv3x3_split_nhwc")
        return out
    with torch.cuda.device(dev):
        rc = L.lib().vs_conv3x3_nhwc(L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(residual), L.ptr(out), N, H, W, Cin, Cout, stride,
                                     int(relu_in), int(relu_out), _DTX[x.dtype], L.stream_ptr(dev))
    L.check(rc, "vs_conv3x3_nhwc")
    return out


def conv3x3_head1x1_nhwc(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], w2: torch.Tensor, bias2: torch.Tensor, n_out: int,
                         relu_out: bool = True) -> torch.Tensor:
    """[N,H,W,Cin] 16-bit -> [N,H,W,ld2] 16-bit = conv1x1(act(conv3x3(x) + bias)) + bias2 in one kernel (vs_conv3x3_head1x1_nhwc): the
    last two layers of a DPT head.  w [Cout,3,3,Cin]; w2 [C2pad, Cout] (rows >= n_out zero) with bias2 [C2pad] for Cout = 256, or
    w2 [<= 4, 128] with bias2 [4] for Cout = 128.  The result keeps its padded channel stride (ld2 = w2 rows, or 4): slice it."""
    if isinstance(w, SplitWeight) and isinstance(w2, SplitWeight):   # split operands, MFMA form: Cout = 256, w2 packed [C2pad <= 96, 256]
        xin_packed = isinstance(x, SplitWeight)      # the input already in the packed (hi, lo) form (upsample2x_nhwc(..., packed=True))
        if xin_packed:
            x = x.data
        dev = L.require_device(x, w.data, bias, w2.data, bias2)
        N, H, W, Cin = x.shape
        c2pad = w2.shape[0]
        assert x.is_contiguous() and x.dtype == (torch.int32 if xin_packed else torch.float32) and tuple(w.shape) == (256, 3, 3, Cin) and tuple(w2.shape) == (c2pad, 256)
        assert bias2.dtype == torch.float32 and bias2.numel() == c2pad and n_out <= c2pad
        out = torch.empty((N, H, W, c2pad), dtype=torch.float32, device=dev)
        if RANGE_GUARD.enabled:
            RANGE_GUARD.check(f"fused head conv3x3 {Cin}->256 -> 1x1 @{H}x{W}", SplitWeight(x.view(-1, Cin), 1.0, (N * H * W, Cin)) if xin_packed else x)
        with torch.cuda.device(dev):
            rc = L.lib().vs_conv3x3_head1x1_split_nhwc(L.ptr(x), L.ptr(w.data), w.acc_scale, L.ptr(bias), L.ptr(w2.data), w2.acc_scale, L.ptr(bias2),
                                                       L.ptr(out), N, H, W, Cin, n_out, c2pad, c2pad, int(relu_out) | (16 if xin_packed else 0), L.stream_ptr(dev))
        L.check(rc, "vs_conv3x3_head1x1_split_nhwc")
        return out
    if isinstance(w, SplitWeight):   # split operands: the dot-product form (Cout = 128, n_out <= 4), f32 in / out, w2 f32 [>= n_out, 128]
        xin_packed = isinstance(x, SplitWeight)      # the input already packed (upsample2x_nhwc(..., packed=True)): no conversion in the main loop
        if xin_packed:
            x = x.data
        dev = L.require_device(x, w.data, bias, w2, bias2)
        N, H, W, Cin = x.shape
        assert x.is_contiguous() and x.dtype == (torch.int32 if xin_packed else torch.float32) and tuple(w.shape) == (128, 3, 3, Cin) and n_out <= 4
        assert w2.dtype == torch.float32 and w2.is_contiguous() and w2.shape[1] == 128 and w2.shape[0] >= n_out and bias2.numel() >= 4
        out = torch.empty((N, H, W, 4), dtype=torch.float32, device=dev)
        if RANGE_GUARD.enabled:
            RANGE_GUARD.check(f"fused head conv3x3 {Cin}->128 -> dot @{H}x{W}", SplitWeight(x.view(-1, Cin), 1.0, (N * H * W, Cin)) if xin_packed else x)
        with torch.cuda.device(dev):
            rc = L.lib().vs_conv3x3_head_dot_split_nhwc(L.ptr(x), L.ptr(w.data), w.acc_scale, L.ptr(bias), L.ptr(w2), L.ptr(bias2), L.ptr(out), N, H, W,
                                                        Cin, n_out, 4, 0, int(relu_out) | (16 if xin_packed else 0), L.stream_ptr(dev))
        L.check(rc, "vs_conv3x3_head_dot_split_nhwc")
        return out
    dev = L.require_device(x, w, bias, w2, bias2)
    assert x.dim() == 4 and x.is_contiguous() and w.is_contiguous() and w2.is_contiguous() and x.dtype == w.dtype == w2.dtype
    assert x.dtype in (torch.float16, torch.bfloat16) and bias2.dtype == torch.float32 and bias2.is_contiguous()
    N, H, W, Cin = x.shape
    Cout = w.shape[0]
    assert w.shape == (Cout, 3, 3, Cin) and w2.shape[1] == Cout
    if Cout == 256:
        c2pad = ld2 = w2.shape[0]
        assert bias2.numel() == c2pad
    else:
        c2pad, ld2 = n_out, 4
        assert w2.shape[0] >= n_out and bias2.numel() >= 4
    out = torch.empty((N, H, W, ld2), dtype=x.dtype, device=dev)
    with torch.cuda.device(dev):
        rc = L.lib().vs_conv3x3_head1x1_nhwc(L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(w2), L.ptr(bias2), L.ptr(out), N, H, W, Cin, Cout, n_out,
                                             c2pad, ld2, 0, int(relu_out), _DT[x.dtype], L.stream_ptr(dev))
    L.check(rc, "vs_conv3x3_head1x1_nhwc")
    return out


def pack_conv3x3_weight(w: torch.Tensor, dtype: torch.dtype, cin_pad: int = 0) -> torch.Tensor:
    """nn.Conv2d weight [Cout,Cin,3,3] -> [Cout,3,3,Cin(+pad)] in the operand dtype (tap-major, channel-minor)."""
    wp = w.detach().permute(0, 2, 3, 1)
    if cin_pad > w.shape[1]:
        wp = torch.nn.functional.pad(wp, (0, cin_pad - w.shape[1]))
    if dtype == "split":
        return split_pack_weight(wp.float().contiguous())
    return wp.to(dtype).contiguous()


def pack_conv7x7_rgb_weight(w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """nn.Conv2d(3, Cout, 7) weight [Cout,3,7,7] -> [Cout, 8, 32] with w[co, dy, dx*3 + c], zeros in columns 21..31 and in the eighth row
    (K = 256 for the 256x256 tile kernel; the 7-row route reads the same buffer with a row stride of 256)."""
    Cout = w.shape[0]
    assert w.shape == (Cout, 3, 7, 7)
    split = dtype == "split"
    wp = torch.zeros(Cout, 8, 32, dtype=torch.float32 if split else dtype, device=w.device)
    wp[:, :7, :21] = w.detach().permute(0, 2, 3, 1).reshape(Cout, 7, 21).to(wp.dtype)
    return split_pack_weight(wp.reshape(Cout, 256)) if split else wp.contiguous()


def pad_rgb_nhwc(frames: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """frames [N,3,H,W] -> zero-bordered NHWC 16-bit image [N, H+6, W+8, 3] (3 before, 3/5 after) for conv7x7_rgb_nhwc;
    the storage carries one padded row + 64 spare halfs behind the last pixel (the last window slices -- and the zero-weight eighth
    kernel row of the 256-tile route -- read past it)."""
    N, C, H, W = frames.shape
    assert C == 3
    Hp, Wp = H + 6, W + 8
    flat = torch.zeros(N * Hp * Wp * 3 + Wp * 3 + 64, dtype=dtype, device=frames.device)
    img = flat[:N * Hp * Wp * 3].view(N, Hp, Wp, 3)
    img[:, 3:3 + H, 3:3 + W] = frames.permute(0, 2, 3, 1)
    return img


def conv7x7_rgb_nhwc(img_padded: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], H: int, W: int,
                     up_add: Optional[torch.Tensor] = None):
    """img_padded from pad_rgb_nhwc, w from pack_conv7x7_rgb_weight -> [N,H,W,Cout] = conv2d(k=7, s=1, p=3) + bias.
    up_add (split class only; [N, H/2, W/2, Cout] f32 contiguous): the Gaussian-parameter head's fusion (dpt_gs_head.py:142-150) -- returns
    the PACKED (hi, lo) activation bilinear_x2(up_add) + relu(conv + bias) as a SplitWeight [N,H,W,Cout] (what
    upsample2x_nhwc(up_add, add=conv, relu_add=True, packed=True) returns) without writing the f32 stem map."""
    if isinstance(w, SplitWeight):     # split operands: f32 image, packed [Cout, 256] weight (pack_conv7x7_rgb_weight(w, "split"))
        dev = L.require_device(img_padded, w.data, bias, up_add)
        N, Hp, Wp, C = img_padded.shape
        Cout = w.shape[0]
        assert C == 3 and img_padded.is_contiguous() and img_padded.dtype == torch.float32 and tuple(w.shape) == (Cout, 256)
        assert img_padded.untyped_storage().nbytes() >= (img_padded.storage_offset() + img_padded.numel() + Wp * 3 + 64) * 4
        if up_add is not None:
            assert up_add.dtype == torch.float32 and up_add.is_contiguous() and tuple(up_add.shape) == (N, H // 2, W // 2, Cout) and bias is not None
            if RANGE_GUARD.enabled:
                RANGE_GUARD.check(f"stem + upsample-add: trunk {Cout} @{H // 2}x{W // 2}", up_add)
            outp = torch.empty((N, H, W, Cout), dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                rc = L.lib().vs_conv7x7_rgb_split_up_nhwc(L.ptr(img_padded), L.ptr(w.data), w.acc_scale, L.ptr(bias), L.ptr(up_add), L.ptr(outp), N, H, W,
                                                          Hp, Wp, Cout, L.stream_ptr(dev))
            L.check(rc, "vs_conv7x7_rgb_split_up_nhwc")
            return SplitWeight(outp, 1.0, outp.shape)
        out = torch.empty((N, H, W, Cout), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = L.lib().vs_conv7x7_rgb_split_nhwc(L.ptr(img_padded), L.ptr(w.data), w.acc_scale, L.ptr(bias), L.ptr(out), N, H, W, Hp, Wp, Cout,
                                                   L.stream_ptr(dev))
        L.check(rc, "vs_conv7x7_rgb_split_nhwc")
        return out
    dev = L.require_device(img_padded, w, bias)
    N, Hp, Wp, C = img_padded.shape
    Cout = w.shape[0]
    assert C == 3 and img_padded.is_contiguous() and w.shape == (Cout, 8, 32) and w.is_contiguous() and img_padded.dtype == w.dtype
    assert img_padded.untyped_storage().nbytes() >= (img_padded.storage_offset() + img_padded.numel() + Wp * 3 + 64) * 2
    out = torch.empty((N, H, W, Cout), dtype=w.dtype, device=dev)
    with torch.cuda.device(dev):
        rc = L.lib().vs_conv7x7_rgb_nhwc(L.ptr(img_padded), L.ptr(w), L.ptr(bias), L.ptr(out), N, H, W, Hp, Wp, Cout,
                                         _DT[w.dtype], L.stream_ptr(dev))
    L.check(rc, "vs_conv7x7_rgb_nhwc")
    return out


def im2col7x7_rgb(frames: torch.Tensor, dtype: torch.dtype, ld: int = 256) -> torch.Tensor:
    """frames [N,3,H,W] f32 -> [N, H*W, ld] rows of the 7x7 / pad 3 windows in (c, ky, kx) order (columns 147.. zero), `dtype` f32 / f16 / bf16:
    F.unfold(frames.to(dtype), 7, padding=3).transpose(1, 2) padded to ld columns, in one pass (vs_im2col7x7_rgb)."""
    dev = L.require_device(frames)
    N, C, H, W = frames.shape
    assert C == 3 and frames.dtype == torch.float32 and frames.is_contiguous() and ld % 8 == 0 and ld >= 148
    out = torch.empty((N, H * W, ld), dtype=dtype, device=dev)
    with torch.cuda.device(dev):
        rc = L.lib().vs_im2col7x7_rgb(L.ptr(frames), L.ptr(out), N, H, W, ld, {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}[dtype], L.stream_ptr(dev))
    L.check(rc, "vs_im2col7x7_rgb")
    return out


def stem7x7_up_split_stream(img_padded: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, H: int, W: int, up_add: torch.Tensor, scale_exp: int):
    """conv7x7_rgb_nhwc(img_padded, pack(w), bias, H, W, up_add=up_add) of the split class as the streaming kernel (csrc/stem_stream.hip):
    w = the module's [256, 3, 7, 7] f32 parameter (packed in the kernel with 2^scale_exp), returns the packed SplitWeight [N,H,W,256]."""
    dev = L.require_device(img_padded, w, bias, up_add)
    N, Hp, Wp, C = img_padded.shape
    Cout = w.shape[0]
    assert C == 3 and img_padded.is_contiguous() and img_padded.dtype == torch.float32 and w.dtype == torch.float32 and w.is_contiguous()
    assert tuple(w.shape) == (256, 3, 7, 7) and W % 32 == 0 and H % 2 == 0 and bias.dtype == torch.float32 and bias.is_contiguous()
    assert up_add.dtype == torch.float32 and up_add.is_contiguous() and tuple(up_add.shape) == (N, H // 2, W // 2, Cout)
    if RANGE_GUARD.enabled:
        RANGE_GUARD.check(f"stem + upsample-add: trunk {Cout} @{H // 2}x{W // 2}", up_add)
    outp = torch.empty((N, H, W, Cout), dtype=torch.int32, device=dev)
    nwg = max(1, min(N * (W // 32), torch.cuda.get_device_properties(dev).multi_processor_count))
    with torch.cuda.device(dev):
        rc = L.lib().vs_stem7x7_up_split_stream(L.ptr(img_padded), L.ptr(w), int(scale_exp), L.ptr(bias), L.ptr(up_add), L.ptr(outp), N, H, W, Hp, Wp, Cout,
                                                nwg, L.stream_ptr(dev))
    L.check(rc, "vs_stem7x7_up_split_stream")
    return SplitWeight(outp, 1.0, outp.shape)


def upsample2x_nhwc(x: torch.Tensor, add: Optional[torch.Tensor] = None, relu_add: bool = False, packed: bool = False):
    """Bilinear x2, align_corners=True, on [N,H,W,C] contiguous 16-bit / f32; optional fused `+ add` / `+ relu(add)`.
    packed=True (f32 input, C % 32 == 0): the result is written in the packed (hi, lo) form of the split class and returned as a SplitWeight
    of shape [N,2H,2W,C] -- the operand of a split convolution that then skips its in-loop conversion."""
    dev = L.require_device(x, add)
    assert x.dim() == 4 and x.is_contiguous() and x.dtype in _OPERAND_DTYPES
    N, H, W, Cc = x.shape
    assert not packed or (x.dtype == torch.float32 and Cc % 32 == 0 and H >= 2 and W >= 2)
    out = torch.empty((N, 2 * H, 2 * W, Cc), dtype=torch.int32 if packed else x.dtype, device=dev)
    assert add is None or (add.shape == out.shape and add.is_contiguous() and add.dtype == x.dtype)
    with torch.cuda.device(dev):
        rc = L.lib().vs_upsample2x_nhwc(L.ptr(x), L.ptr(add), L.ptr(out), N, H, W, Cc, int(relu_add) | (16 if packed else 0), _DTX[x.dtype], L.stream_ptr(dev))
    L.check(rc, "vs_upsample2x_nhwc")
    return SplitWeight(out, 1.0, out.shape) if packed else out


# ---------------------------------------------------------------------------------------------------------------------
# Encoder backward building blocks: parity-tested against torch autograd (tests/test_ops_gpu.py), assembled into
# torch.autograd.Functions in vicasplat_amd/autograd.py.
# ---------------------------------------------------------------------------------------------------------------------
_DT3 = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def transpose16(x: torch.Tensor, pad_to: int = 1, *, colsum_out: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                border_hw: Optional[tuple] = None, relu: bool = False, slices: int = 1, halo: int = 0) -> torch.Tensor:
    """x [R,C] 16-bit (row stride any) -> [C, Rpad] with Rpad = R rounded up to `pad_to`, padding columns zero.
    colsum_out (f32 [C]) additionally receives the column sums of x (overwritten) from the same pass.
    border_hw = (H, W): x is [n*H*W, C] (NHWC pixels) and the transposed rows are the pixels of the zero-bordered
    [n, H+2, W+2] maps (R = n*(H+2)*(W+2)); relu zeroes negative inputs.  `out` [C, >= Rpad] may be a column-offset view.
    slices > 1: slice-blocked result [slices, C, Rpad/slices + 2*halo] (slice z = rows [z*SL - halo, (z+1)*SL + halo), zeros
    outside [0, R)): the operand layout of gemm_wgrad."""
    dev = L.require_device(x, colsum_out, out)
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype in (torch.float16, torch.bfloat16)
    R, Cc = x.shape
    bh = bw = 0
    if border_hw is not None:
        bh, bw = border_hw
        assert R % (bh * bw) == 0
        R = R // (bh * bw) * (bh + 2) * (bw + 2)
    Rpad = (R + pad_to - 1) // pad_to * pad_to
    if slices > 1:
        assert out is None and Rpad % slices == 0
        out = torch.empty((slices, Cc, Rpad // slices + 2 * halo), dtype=x.dtype, device=dev)
        ld_out, sstride = out.stride(1), out.stride(0)
    else:
        assert halo == 0
        if out is None:
            out = torch.empty((Cc, Rpad), dtype=x.dtype, device=dev)
        else:
            assert out.dtype == x.dtype and out.shape[0] == Cc and out.stride(1) == 1 and out.shape[1] >= Rpad
        ld_out, sstride = out.stride(0), 0
    if colsum_out is not None:
        assert colsum_out.dtype == torch.float32 and colsum_out.is_contiguous() and colsum_out.numel() == Cc
    with torch.cuda.device(dev):
        rc = L.lib().vs_transpose16_ex(L.ptr(x), x.stride(0), L.ptr(out), ld_out, R, Cc, Rpad, L.ptr(colsum_out), _DT[x.dtype],
                                       bh, bw, int(relu), slices, halo, sstride, L.stream_ptr(dev))
    L.check(rc, "vs_transpose16_ex")
    return out


def wgrad_ksplit(out_rows: int, out_cols: int, red: int, ntaps: int = 1) -> tuple[int, int]:
    """(ksplit, reduction padding unit) of a weight-gradient GEMM out[out_rows, out_cols] = sum over `red` rows.  Mirrors the
    dispatch of vs_gemm_splitk_accumulate: whole 256x256 output tiles run on the 8-wave kernel, one workgroup per CU (aim at
    ~256 workgroups, K slices of an even number of 64-wide tiles); anything else on 128x128 tiles, 3 per CU (~768)."""
    if out_rows % 256 == 0 and out_cols % 256 == 0:
        tiles = (out_rows // 256) * (out_cols // 256) * ntaps
        ks = max(1, min((256 + tiles // 2) // tiles, red // 512))
        return ks, 128 * ks
    tiles = ((out_rows + 127) // 128) * ((out_cols + 127) // 128) * ntaps
    ks = max(2, min(768 // tiles, red // 512, 1024))
    return ks, 64 * ks


def colsum(x: torch.Tensor) -> torch.Tensor:
    """f32 column sums of x [M,N] (f32 / f16 / bf16)."""
    dev = L.require_device(x)
    assert x.dim() == 2 and x.stride(1) == 1
    out = torch.empty(x.shape[1], dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = L.lib().vs_colsum(L.ptr(x), x.stride(0), L.ptr(out), x.shape[0], x.shape[1], _DT3[x.dtype], L.stream_ptr(dev))
    L.check(rc, "vs_colsum")
    return out


def gated_resid(x: torch.Tensor, y: torch.Tensor, gate: Optional[torch.Tensor] = None, gate_rows: int = 0, *, grp_in: int = 0,
                grp_out: int = 0, grp_off: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = x + (1 + gate[row // gate_rows]) * y[yrow(row)]: x f32 [M,C] contiguous, y 16-bit [rows,C] (row stride any), gate f32
    [G,C] or None; yrow(m) = (m // grp_in) * grp_out + grp_off + m % grp_in picks y's rows (identity by default)."""
    dev = L.require_device(x, y, gate, out)
    assert x.dim() == 2 and x.is_contiguous() and x.dtype == torch.float32 and y.stride(1) == 1 and y.shape[1] == x.shape[1]
    assert gate is None or (gate.is_contiguous() and gate.dtype == torch.float32 and gate.shape[1] == x.shape[1])
    if out is None:
        out = torch.empty_like(x)
    if y.dtype == torch.float32:      # f32 branch (split / f32 operand classes)
        with torch.cuda.device(dev):
            rc = L.lib().vs_gated_resid_f32(L.ptr(x), L.ptr(y), y.stride(0), L.ptr(gate), gate_rows, L.ptr(out), x.shape[0], x.shape[1], grp_in,
                                            grp_out, grp_off, L.stream_ptr(dev))
        L.check(rc, "vs_gated_resid_f32")
        return out
    with torch.cuda.device(dev):
        rc = L.lib().vs_gated_resid(L.ptr(x), L.ptr(y), y.stride(0), L.ptr(gate), gate_rows, L.ptr(out), x.shape[0], x.shape[1], grp_in,
                                    grp_out, grp_off, _DT[y.dtype], L.stream_ptr(dev))
    L.check(rc, "vs_gated_resid")
    return out


def gated_resid_backward(dout: torch.Tensor, y: torch.Tensor, gate: Optional[torch.Tensor], gate_rows: int, dy: torch.Tensor, *,
                         grp_in: int = 0, grp_out: int = 0, grp_off: int = 0) -> Optional[torch.Tensor]:
    """Branch-side backward of gated_resid: writes dy[yrow(m)] = dout[m] * (1 + gate) into the given 16-bit buffer (other rows are
    left alone) and returns dgate [G,C] f32 (None without a gate)."""
    dev = L.require_device(dout, y, gate, dy)
    assert dout.dim() == 2 and dout.is_contiguous() and dout.dtype == torch.float32 and dy.stride(1) == 1 and dy.dtype == y.dtype
    dgate = None if gate is None else torch.zeros_like(gate)
    if y.dtype == torch.float32:
        with torch.cuda.device(dev):
            rc = L.lib().vs_gated_resid_backward_f32(L.ptr(dout), L.ptr(y), y.stride(0), L.ptr(gate), gate_rows, L.ptr(dy), dy.stride(0),
                                                     L.ptr(dgate), dout.shape[0], dout.shape[1], grp_in, grp_out, grp_off, L.stream_ptr(dev))
        L.check(rc, "vs_gated_resid_backward_f32")
        return dgate
    with torch.cuda.device(dev):
        rc = L.lib().vs_gated_resid_backward(L.ptr(dout), L.ptr(y), y.stride(0), L.ptr(gate), gate_rows, L.ptr(dy), dy.stride(0),
                                             L.ptr(dgate), dout.shape[0], dout.shape[1], grp_in, grp_out, grp_off, _DT[y.dtype],
                                             L.stream_ptr(dev))
    L.check(rc, "vs_gated_resid_backward")
    return dgate


def gelu16(z: torch.Tensor) -> torch.Tensor:
    """gelu_erf(z) on a contiguous 16-bit tensor (numel % 8 == 0)."""
    dev = L.require_device(z)
    assert z.is_contiguous() and z.dtype in (torch.float16, torch.bfloat16, torch.float32)
    out = torch.empty_like(z)
    if z.dtype == torch.float32:
        with torch.cuda.device(dev):
            rc = L.lib().vs_gelu_f32(L.ptr(z), L.ptr(out), z.numel(), L.stream_ptr(dev))
        L.check(rc, "vs_gelu_f32")
        return out
    with torch.cuda.device(dev):
        rc = L.lib().vs_gelu16(L.ptr(z), L.ptr(out), z.numel(), _DT[z.dtype], L.stream_ptr(dev))
    L.check(rc, "vs_gelu16")
    return out


def gelu_backward(dy: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
    """dz = dy * gelu_erf'(z); dy, z contiguous 16-bit of the same shape (numel % 8 == 0)."""
    dev = L.require_device(dy, z)
    assert dy.shape == z.shape and dy.dtype == z.dtype and dy.is_contiguous() and z.is_contiguous()
    dz = torch.empty_like(dy)
    if z.dtype == torch.float32:
        with torch.cuda.device(dev):
            rc = L.lib().vs_gelu_backward_f32(L.ptr(dy), L.ptr(z), L.ptr(dz), dy.numel(), L.stream_ptr(dev))
        L.check(rc, "vs_gelu_backward_f32")
        return dz
    with torch.cuda.device(dev):
        rc = L.lib().vs_gelu_backward(L.ptr(dy), L.ptr(z), L.ptr(dz), dy.numel(), _DT[dy.dtype], L.stream_ptr(dev))
    L.check(rc, "vs_gelu_backward")
    return dz


def layernorm_backward(dout: torch.Tensor, x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, *, scale: Optional[torch.Tensor] = None,
                       mod_rows: int = 0, eps: float = 1e-6, dx: Optional[torch.Tensor] = None, accumulate_dx: bool = False,
                       grp_in: int = 0, grp_out: int = 0, grp_off: int = 0, dx_add: Optional[torch.Tensor] = None,
                       dx16: Optional[torch.Tensor] = None):
    """Backward of layernorm_mod.  x f32 [M,C]; dout [rows,C] in the forward's output layout (f32 or 16-bit).
    Returns (dx f32 [M,C], dw [C], db [C], dscale [G,C] | None, dshift [G,C] | None).  dx_add [M,C] f32: gradient arriving through the
    residual connection, added into dx (accumulate_dx=True is dx_add = dx); dx16 [M,C] 16-bit: receives a copy of dx for the next GEMMs."""
    dev = L.require_device(dout, x, w, b, scale, dx, dx_add, dx16)
    M, Cc = x.shape
    assert x.dtype == torch.float32 and x.stride(1) == 1 and dout.stride(-1) == 1
    if dx is None:
        assert not accumulate_dx
        dx = torch.empty((M, Cc), dtype=torch.float32, device=dev)
    if accumulate_dx:
        assert dx_add is None
        dx_add = dx
    assert dx_add is None or (dx_add.dtype == torch.float32 and dx_add.stride(1) == 1 and dx_add.shape == (M, Cc))
    assert dx16 is None or (dx16.dtype in (torch.float16, torch.bfloat16) and dx16.stride(1) == 1 and dx16.shape == (M, Cc))
    dw, db = torch.zeros((2, Cc), dtype=torch.float32, device=dev).unbind(0)      # (one fill for the pair: the accumulators of f32 atomics)
    dscale = dshift = None
    mod_ld = 0
    if scale is not None:
        assert scale.dtype == torch.float32 and scale.stride(-1) == 1
        mod_ld = scale.stride(0)
        # dscale / dshift are written with the row stride of `scale`: allocate [G, mod_ld] when the rows are wider than C
        dscale, dshift = torch.zeros((2, scale.shape[0], mod_ld if mod_ld != Cc else Cc), dtype=torch.float32, device=dev).unbind(0)
    with torch.cuda.device(dev):
        rc = L.lib().vs_layernorm_backward_ex(L.ptr(dout), dout.stride(-2), _DT3[dout.dtype], L.ptr(x), x.stride(0), L.ptr(w), L.ptr(b),
                                              L.ptr(scale), mod_rows, mod_ld, L.ptr(dx), dx.stride(0), L.ptr(dx_add),
                                              dx_add.stride(0) if dx_add is not None else 0, L.ptr(dx16),
                                              dx16.stride(0) if dx16 is not None else 0, _DT[dx16.dtype] if dx16 is not None else 0,
                                              L.ptr(dw), L.ptr(db), L.ptr(dscale), L.ptr(dshift), M, Cc, eps, grp_in, grp_out, grp_off,
                                              L.stream_ptr(dev))
    L.check(rc, "vs_layernorm_backward")
    if dscale is not None and dscale.shape[1] != Cc:
        dscale, dshift = dscale[:, :Cc], dshift[:, :Cc]
    return dx, dw, db, dscale, dshift


def linear_backward(dy: torch.Tensor, x: torch.Tensor, w: torch.Tensor, *, need_dx: bool = True, need_dw: bool = True,
                    need_db: bool = True, wT: Optional[torch.Tensor] = None):
    """Backward of y = x @ w^T + b on the NT GEMM kernels: dy [M,N], x [M,K], w [N,K] 16-bit.
    dx = dy @ w (16-bit), dw = dy^T @ x (f32, f32 accumulation over M), db = column sums of dy (f32).
    `wT` = transpose16(w, 64) can be cached by the caller (weights change once per optimiser step)."""
    M, N = dy.shape
    K = x.shape[1]
    dev = dy.device
    dx = dw = db = None
    if need_dx:
        if wT is None:
            wT = transpose16(w, 64)                       # [K, Npad]: reduction over N
        dyp = dy if N % 64 == 0 else torch.nn.functional.pad(dy, (0, wT.shape[1] - N))
        dx = torch.empty((M, K), dtype=dy.dtype, device=dev)
        gemm(dyp, wT, None, dx, EPI_STORE16)
    if need_dw:
        # reduction over the M rows: few output tiles, a very long K -> split it over workgroups (partial tiles + reduce kernel)
        ks, unit = wgrad_ksplit(N, K, M)
        dw = torch.empty((N, K), dtype=torch.float32, device=dev)
        skinny = M >= 1 << 20                    # millions of rows: HBM-bound whatever the tile fill, and the transposes cost 3 passes
        if (((N % 256 == 0 and K % 256 == 0) or skinny) and dy.stride(0) % 8 == 0 and x.stride(0) % 8 == 0
                and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 and dy.stride(0) >= (N + 7) // 8 * 8):
            # the reduction-major kernel reads dY and X as they are (LDS transpose reads), no transposed copies
            gemm_wgrad_tn(dy, x, dw, ks if (N % 256 == 0 and K % 256 == 0) else max(1, min(256, M // 512)), accumulate=False)
            if need_db:
                if N % 8 == 0:
                    db = colsum(dy)
                else:                             # rows padded with zeros up to a multiple of 8 (gaussian_adapter_backward): 16-byte loads
                    Np = (N + 7) // 8 * 8
                    db = colsum(torch.as_strided(dy, (M, Np), (dy.stride(0), 1)))[:N].contiguous()
        else:
            if need_db:
                db = torch.empty(N, dtype=torch.float32, device=dev)
            # slice-blocked [ks, N, Mpad/ks], [ks, K, Mpad/ks], zero padded
            dyT, xT = transpose16(dy, unit, colsum_out=db, slices=ks), transpose16(x, unit, slices=ks)
            gemm_wgrad(dyT, xT, dw, ks, accumulate=False)
    elif need_db:
        db = colsum(dy)
    return dx, dw, db


def relu_mask_(dx: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """In place dx = x > 0 ? dx : 0 (contiguous 16-bit, same shape): backward of a ReLU applied to x."""
    dev = L.require_device(dx, x)
    assert dx.shape == x.shape and dx.dtype == x.dtype and dx.is_contiguous() and x.is_contiguous()
    with torch.cuda.device(dev):
        rc = L.lib().vs_relu_mask16(L.ptr(dx), L.ptr(x), dx.numel(), L.stream_ptr(dev))
    L.check(rc, "vs_relu_mask16")
    return dx


def relu_mask(dy: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """x > 0 ? dy : 0 into a new tensor (contiguous 16-bit, same shape): ReLU backward that leaves the incoming gradient intact."""
    dev = L.require_device(dy, x)
    assert dy.shape == x.shape and dy.dtype == x.dtype and dy.is_contiguous() and x.is_contiguous()
    out = torch.empty_like(dy)
    if x.dtype == torch.float32:
        with torch.cuda.device(dev):
            rc = L.lib().vs_relu_mask_f32(L.ptr(dy), L.ptr(x), L.ptr(out), dy.numel(), L.stream_ptr(dev))
        L.check(rc, "vs_relu_mask_f32")
        return out
    with torch.cuda.device(dev):
        rc = L.lib().vs_relu_mask16_to(L.ptr(dy), L.ptr(x), L.ptr(out), dy.numel(), L.stream_ptr(dev))
    L.check(rc, "vs_relu_mask16_to")
    return out


def gemm_wgrad(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, ksplit: int, *, shifts=None, K: Optional[int] = None,
               workspace: bool = True, accumulate: bool = True) -> torch.Tensor:
    """out32[t][M,N] += a[M,K] @ (w shifted by shifts[t])[N,K]^T, K split over `ksplit` workgroups per tile (vs_gemm_wgrad).
    a, w: 2-D [rows, K] (slice s = columns [s K/ksplit, ...)) or slice-blocked 3-D [ksplit, rows, K/ksplit] as transpose16(slices=)
    writes them (w may be a column-offset view into a wider halo'd buffer; K = ksplit * slice length then).
    out is [M,N] (shifts None) or [len(shifts),M,N] contiguous.  workspace: the K slices store partial tiles that a second
    kernel sums (default); False: they meet through f32 atomics.  accumulate=False (workspace mode) overwrites out: no zero fill."""
    dev = L.require_device(a, w, out)
    assert a.dtype == w.dtype and a.stride(-1) == 1 and w.stride(-1) == 1 and out.dtype == torch.float32 and out.is_contiguous()
    assert a.dim() == w.dim()
    asl = wsl = 0
    if a.dim() == 3:
        assert a.shape[0] == ksplit and w.shape[0] == ksplit and a.shape[2] == w.shape[2]
        if ksplit == 1:
            a, w = a[0], w[0]
        else:
            asl, wsl = a.stride(0), w.stride(0)
            K = ksplit * a.shape[2] if K is None else K
    K = a.shape[-1] if K is None else K
    M, N = a.shape[-2], w.shape[-2]
    ntaps = 0 if shifts is None else len(shifts)
    assert out.shape == ((M, N) if shifts is None else (ntaps, M, N))
    import ctypes
    sh = None if shifts is None else (ctypes.c_int32 * ntaps)(*shifts)
    ws, ws_bytes = None, 0
    if workspace:
        ws = torch.empty(max(2, ksplit) * max(1, ntaps) * M * N, dtype=torch.float32, device=dev)
        ws_bytes = ws.numel() * 4
    with torch.cuda.device(dev):
        rc = L.lib().vs_gemm_wgrad(L.ptr(a), L.ptr(w), L.ptr(out), M, N, K, a.stride(-2), w.stride(-2), N, asl, wsl, M * N, sh, ntaps,
                                   ksplit, _DT[a.dtype], L.ptr(ws), ws_bytes, int(accumulate), L.stream_ptr(dev))
    L.check(rc, "vs_gemm_wgrad")
    return out


def gemm_wgrad_tn(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, ksplit: int, *, workspace: bool = True, accumulate: bool = True):
    """out32[M,N] (+)= a^T @ w for reduction-major operands a [Kred, M], w [Kred, N] (16-bit, row stride any multiple of 8): the weight
    gradient dW = dY^T X without transposed copies (vs_gemm_wgrad_tn; M, N multiples of 256)."""
    dev = L.require_device(a, w, out)
    assert a.dtype == w.dtype and a.stride(1) == 1 and w.stride(1) == 1 and a.shape[0] == w.shape[0]
    assert out.dtype == torch.float32 and out.is_contiguous() and out.shape == (a.shape[1], w.shape[1])
    M, N, Kred = a.shape[1], w.shape[1], a.shape[0]
    ws, ws_bytes = None, 0
    if workspace:
        ws = torch.empty(ksplit * M * N, dtype=torch.float32, device=dev)
        ws_bytes = ws.numel() * 4
    with torch.cuda.device(dev):
        rc = L.lib().vs_gemm_wgrad_tn(L.ptr(a), L.ptr(w), L.ptr(out), M, N, Kred, a.stride(0), w.stride(0), N, ksplit, _DT[a.dtype],
                                      L.ptr(ws), ws_bytes, int(accumulate), L.stream_ptr(dev))
    L.check(rc, "vs_gemm_wgrad_tn")
    return out


def gemm_splitk_accumulate(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, ksplit: int, *, K: Optional[int] = None) -> torch.Tensor:
    """out32 [M,N] += a[M,K] @ w[N,K]^T with K split over `ksplit` workgroups per tile, f32 atomics (vs_gemm_splitk_accumulate)."""
    dev = L.require_device(a, w, out)
    assert a.dtype == w.dtype and a.stride(1) == 1 and w.stride(1) == 1 and out.dtype == torch.float32
    K = a.shape[1] if K is None else K
    with torch.cuda.device(dev):
        rc = L.lib().vs_gemm_splitk_accumulate(L.ptr(a), L.ptr(w), L.ptr(out), a.shape[0], w.shape[0], K, a.stride(0), w.stride(0),
                                               out.stride(0), ksplit, _DT[a.dtype], L.stream_ptr(dev))
    L.check(rc, "vs_gemm_splitk_accumulate")
    return out


def conv3x3_backward(dy: torch.Tensor, x: torch.Tensor, w: torch.Tensor, *, relu_in: bool = False, need_dx: bool = True,
                     need_db: bool = True):
    """Backward of conv3x3_nhwc(x, w, bias, relu_in=relu_in) (stride 1, pad 1) for the gradient dy of its pre-activation output.
    dy [N,H,W,Cout], x [N,H,W,Cin] (the conv input, before the fused input ReLU), w [Cout,3,3,Cin], all 16-bit NHWC.
    Returns dx [N,H,W,Cin] 16-bit, dw [Cout,3,3,Cin] f32, db [Cout] f32.
      dx: the same implicit-GEMM kernel on dy with the spatially flipped, channel-transposed weights;
      dw: per tap one split-K GEMM over the zero-bordered, transposed activations -- the tap shift is a column shift of the
          [Cin, pixels] operand (LDS-DMA reads any 2-byte aligned address), so no im2col buffer exists here either."""
    N, H, W, Cin = x.shape
    Cout = w.shape[0]
    dev, dt = x.device, x.dtype
    dx = None
    if need_dx:
        wd = w.flip(1, 2).permute(3, 1, 2, 0).contiguous()          # [Cin, 3, 3, Cout]
        dx = conv3x3_nhwc(dy.contiguous(), wd, None, mask_by=x if relu_in else None)   # ReLU backward in the conv epilogue
    r256 = lambda c: (c + 255) // 256 * 256
    G = 256 // Cin if (Cin <= 128 and 256 % Cin == 0) else 1              # taps that share one 256-row tile (conv3x3_wgrad_tn_kernel)
    groups = (9 + G - 1) // G
    tile_rows = groups * 256 if G > 1 else 9 * r256(Cin)
    if (Cin % 8 == 0 and Cout % 8 == 0 and 9 * Cin * Cout >= 0.4 * tile_rows * r256(Cout) and x.is_contiguous() and dy.is_contiguous()):
        # 256 x 256 tiles at least 40 % full: the reduction-major kernel reads the NHWC tensors as they are (tap shift = pixel-row
        # shift, zero page outside the image, input ReLU on the fragments) -- no zero-bordered transposed copies
        P = N * H * W
        ks, _ = wgrad_ksplit(256 if G > 1 else r256(Cin), r256(Cout), P, groups if G > 1 else 9)
        dw9 = torch.empty((9, Cin, Cout), dtype=torch.float32, device=dev)
        ws = torch.empty(ks * 9 * Cin * Cout, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = L.lib().vs_conv3x3_wgrad_tn(L.ptr(x), L.ptr(dy), L.ptr(dw9), N, H, W, Cin, Cout, int(relu_in), ks, _DT[dt], L.ptr(ws),
                                             ws.numel() * 4, 0, L.stream_ptr(dev))
        L.check(rc, "vs_conv3x3_wgrad_tn")
        dw = dw9.view(3, 3, Cin, Cout).permute(3, 0, 1, 2).contiguous()            # [Cout, ky, kx, Cin]
        return dx, dw, (colsum(dy.view(P, Cout)) if need_db else None)
    # weight gradient: dY^T [Cout, pixels] and X^T [Cin, pixels] over the zero-bordered pixel grid, produced straight from the
    # NHWC tensors by the transposing kernel (border, input ReLU and the bias gradient folded into that one pass)
    Hp, Wp = H + 2, W + 2
    P = N * Hp * Wp
    ksplit, unit = wgrad_ksplit(Cout, Cin, P, 9)
    Ppad = (P + unit - 1) // unit * unit
    halo = (Wp + 2 + 7) // 8 * 8                                    # >= largest |tap shift| + 1; keeps the rows 16-byte aligned
    xT = transpose16(x.view(N * H * W, Cin), unit, border_hw=(H, W), relu=relu_in, slices=ksplit, halo=halo if ksplit > 1 else 0)
    if ksplit == 1:                                                 # one slice: the shifted views need zero columns on either side
        xT = torch.nn.functional.pad(xT, (halo, halo)).unsqueeze(0)
    db = torch.empty(Cout, dtype=torch.float32, device=dev)
    dyT = transpose16(dy.contiguous().view(N * H * W, Cout), unit, colsum_out=db, border_hw=(H, W), slices=ksplit)
    if ksplit == 1:
        dyT = dyT.unsqueeze(0)
    # all 9 taps in ONE launch (workgroup order: k-slice, tap, tile): the taps of a K slice run together, so A and the nine
    # overlapping shifted views of X^T are served from the caches instead of being streamed from HBM nine times
    dw9 = torch.empty((9, Cout, Cin), dtype=torch.float32, device=dev)
    shifts = [(ty - 1) * Wp + (tx - 1) for ty in range(3) for tx in range(3)]
    gemm_wgrad(dyT, xT[:, :, halo:halo + Ppad // ksplit], dw9, ksplit, shifts=shifts, accumulate=False)
    dw = dw9.view(3, 3, Cout, Cin).permute(2, 0, 1, 3).contiguous()
    return dx, dw, db


def upsample2x_backward_nhwc(dout: torch.Tensor) -> torch.Tensor:
    """Backward of upsample2x_nhwc (no add): dout [N,2H,2W,C] contiguous 16-bit -> din [N,H,W,C]."""
    dev = L.require_device(dout)
    assert dout.dim() == 4 and dout.is_contiguous() and dout.dtype in (torch.float16, torch.bfloat16, torch.float32)
    N, Ho, Wo, Cc = dout.shape
    din = torch.empty((N, Ho // 2, Wo // 2, Cc), dtype=dout.dtype, device=dev)
    if dout.dtype == torch.float32:
        with torch.cuda.device(dev):
            rc = L.lib().vs_upsample2x_backward_f32_nhwc(L.ptr(dout), L.ptr(din), N, Ho // 2, Wo // 2, Cc, L.stream_ptr(dev))
        L.check(rc, "vs_upsample2x_backward_f32_nhwc")
        return din
    with torch.cuda.device(dev):
        rc = L.lib().vs_upsample2x_backward_nhwc(L.ptr(dout), L.ptr(din), N, Ho // 2, Wo // 2, Cc, _DT[dout.dtype], L.stream_ptr(dev))
    L.check(rc, "vs_upsample2x_backward_nhwc")
    return din


# ------------------------------------------------------------------------------------------------------------------------------
# Split-class backward (reference precision: f32 tensors, three f16 MFMAs per product; csrc/split_bwd.hip, gemm.hip, attention_bwd.hip).
# Gradients must sit in the f16 range (|g| < 65504, typical magnitude >~ 1e-2 for full precision): callers.training_step keeps its
# power-of-two loss scale for this class too.
# ------------------------------------------------------------------------------------------------------------------------------
def _border_rows(R, conv_hw, border):
    if not border:
        return R
    h, w = conv_hw
    assert R % (h * w) == 0
    return R // (h * w) * (h + 2) * (w + 2)


def transpose_f32(x: torch.Tensor, pad_to: int = 64, *, relu: bool = False, conv_hw: Optional[tuple] = None, tap: tuple = (0, 0),
                  border: bool = False, halo: int = 0, colsum: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [R,C] f32 (row stride any multiple of 4 or contiguous) -> [C, Rpad] f32, Rpad = R rounded up to pad_to (a multiple of 64), zero padded.
    conv_hw = (H, W) with tap = (dy, dx): row r reads pixel r shifted by the tap, zero outside its image.  border=True: the transposed
    rows are the pixels of the zero-bordered (H+2) x (W+2) maps; halo > 0 (a multiple of 4): the result is the middle of a zero buffer with
    `halo` readable zero columns on both sides (the operand whose shifted views the tap-fused weight gradient reads).
    colsum (f32 [C], overwritten): the column sums of x -- the bias gradient, on the pass that reads dY anyway."""
    dev = L.require_device(x)
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32 and pad_to % 64 == 0 and halo % 4 == 0
    assert colsum is None or (colsum.dtype == torch.float32 and colsum.is_contiguous() and colsum.numel() == x.shape[1])
    R, Cc = x.shape
    Rb = _border_rows(R, conv_hw, border)
    Rpad = (Rb + pad_to - 1) // pad_to * pad_to
    if halo:
        buf = torch.empty((Cc, halo + Rpad + halo), dtype=torch.float32, device=dev)     # (the kernel writes all Rpad columns: only the halos need zeros)
        buf[:, :halo].zero_()
        buf[:, halo + Rpad:].zero_()
        out = buf[:, halo:halo + Rpad]
    else:
        out = torch.empty((Cc, Rpad), dtype=torch.float32, device=dev)
    h, w = conv_hw if conv_hw is not None else (0, 0)
    dy, dx = (2, 0) if border else tap
    with torch.cuda.device(dev):
        rc = L.lib().vs_transpose_f32(L.ptr(x), x.stride(0), L.ptr(out), out.stride(0), Rb, Cc, Rpad, int(relu), h, w, dy, dx, L.ptr(colsum), L.stream_ptr(dev))
    L.check(rc, "vs_transpose_f32")
    return out


def transpose_pack_split(x: torch.Tensor, pad_to: int = 64, *, relu: bool = False, conv_hw: Optional[tuple] = None, tap: tuple = (0, 0),
                         scale_exp: int = 0, border: bool = False, colsum: Optional[torch.Tensor] = None) -> SplitWeight:
    """transpose_f32 written as the packed split "weight" operand [C, Rpad] (SplitWeight, acc_scale = 2^-scale_exp)."""
    dev = L.require_device(x)
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32 and pad_to % 64 == 0
    R, Cc = x.shape
    R = _border_rows(R, conv_hw, border)
    Rpad = (R + pad_to - 1) // pad_to * pad_to
    out = torch.empty((Cc, Rpad), dtype=torch.int32, device=dev)
    h, w = conv_hw if conv_hw is not None else (0, 0)
    if border:
        tap = (2, 0)
    with torch.cuda.device(dev):
        rc = L.lib().vs_transpose_pack_split(L.ptr(x), x.stride(0), L.ptr(out), out.stride(0), R, Cc, Rpad, int(relu), h, w, tap[0], tap[1], scale_exp,
                                             L.ptr(colsum), L.stream_ptr(dev))
    L.check(rc, "vs_transpose_pack_split")
    return SplitWeight(out, 2.0 ** (-scale_exp), (Cc, Rpad))


def split16(x: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """f32 [rows, C] (row stride any multiple of 4) -> (hi, lo) f16 [rows, C] contiguous: hi = rne16(x), lo = rne16(x - hi)."""
    dev = L.require_device(x)
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32 and x.shape[1] % 8 == 0
    hi = torch.empty(x.shape, dtype=torch.float16, device=dev)
    lo = torch.empty_like(hi)
    with torch.cuda.device(dev):
        rc = L.lib().vs_split16(L.ptr(x), x.stride(0), L.ptr(hi), L.ptr(lo), hi.stride(0), x.shape[0], x.shape[1], L.stream_ptr(dev))
    L.check(rc, "vs_split16")
    return hi, lo


def gemm_wgrad_split(a: torch.Tensor, w: SplitWeight, out: torch.Tensor, ksplit: int, shifts=None) -> torch.Tensor:
    """out32 [M,N] = a [M,K] @ w [N,K]^T in the split class, the reduction cut into ksplit slices (vs_gemm_wgrad, dtype 4; partial
    tiles in a workspace + a reduce kernel).  K % (64 * ksplit) == 0.  shifts (<= 9 column offsets, in floats): out [len(shifts), M, N],
    tap t reads `a` shifted by shifts[t] columns -- `a` must be a view with that many readable (zero) columns on both sides."""
    wd = w.data
    dev = L.require_device(a, wd, out)
    assert a.dtype == torch.float32 and a.dim() == 2 and a.stride(1) == 1 and wd.dim() == 2 and wd.stride(1) == 1 and a.shape[1] == wd.shape[1]
    M, N, K = a.shape[0], wd.shape[0], a.shape[1]
    ntaps = 0 if shifts is None else len(shifts)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.shape == ((M, N) if shifts is None else (ntaps, M, N))
    assert K % (64 * ksplit) == 0, (K, ksplit)
    import ctypes
    sh = None if shifts is None else (ctypes.c_int32 * ntaps)(*shifts)
    ws = torch.empty(max(2, ksplit) * max(1, ntaps) * M * N, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = L.lib().vs_gemm_wgrad(L.ptr(a), L.ptr(wd), L.ptr(out), M, N, K, a.stride(0), wd.stride(0), N, 0, 0, M * N, sh, ntaps, ksplit, 4, L.ptr(ws),
                                   ws.numel() * 4, 0, L.stream_ptr(dev))
    L.check(rc, "vs_gemm_wgrad(split)")
    if w.acc_scale != 1.0:
        out.mul_(w.acc_scale)
    return out


def gemm_wgrad_split_atn(a: torch.Tensor, w: SplitWeight, out: torch.Tensor, ksplit: int, *, transpose_out: bool = False) -> torch.Tensor:
    """out32 [M,N] (or [N,M] with transpose_out) = a^T @ w^T in the split class: a [Kred, M] f32 AS IT IS in memory (reduction-major), w the packed
    transposed other operand [N, Kpad] (transpose_pack_split); M, N multiples of 256, Kpad % (64 * ksplit) == 0 (vs_gemm_wgrad_split_atn)."""
    wd = w.data
    dev = L.require_device(a, wd, out)
    assert a.dtype == torch.float32 and a.dim() == 2 and a.stride(1) == 1 and wd.dim() == 2 and wd.stride(1) == 1
    Kred, M = a.shape
    N, Kpad = wd.shape
    assert Kpad >= Kred and Kpad % (64 * ksplit) == 0 and M % 256 == 0 and N % 256 == 0, (a.shape, wd.shape, ksplit)
    assert out.dtype == torch.float32 and out.stride(1) == 1 and out.shape == ((N, M) if transpose_out else (M, N))
    ws = torch.empty(ksplit * M * N, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = L.lib().vs_gemm_wgrad_split_atn(L.ptr(a), L.ptr(wd), L.ptr(out), M, N, Kred, Kpad, a.stride(0), wd.stride(0), out.stride(0), ksplit,
                                             int(transpose_out), L.ptr(ws), ws.numel() * 4, 0, L.stream_ptr(dev))
    L.check(rc, "vs_gemm_wgrad_split_atn")
    if w.acc_scale != 1.0:
        out.mul_(w.acc_scale)
    return out


_WGRAD_ATN = os.environ.get("VS_WGRAD_ATN", "1") != "0"       # 0: the round-3 route (both operands through transposing passes): same-box A/B


def _wgrad_split(dyT: torch.Tensor, xT: SplitWeight, out: torch.Tensor) -> torch.Tensor:
    M, N, K = dyT.shape[0], xT.shape[0], dyT.shape[1]
    if M % 256 == 0 and N % 256 == 0:
        tiles = (M // 256) * (N // 256)
        ks = max(1, min((256 + tiles // 2) // tiles, K // 512))
        while ks > 1 and K % (64 * ks) != 0:
            ks -= 1
    else:
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        ks = max(2, min(768 // tiles, K // 512, 1024))
        while ks > 2 and K % (64 * ks) != 0:
            ks -= 1
    return gemm_wgrad_split(dyT, xT, out, ks)


def linear_backward_split(dy: torch.Tensor, x: torch.Tensor, w: torch.Tensor, *, need_dx: bool = True, need_dw: bool = True, need_db: bool = True,
                          scale_exp: Optional[int] = None, dgelu_z: Optional[torch.Tensor] = None):
    """Backward of y = x @ w^T + b in the split class: dy [M,N], x [M,K], w [N,K] all f32 -> dx [M,K], dw [N,K], db [N] f32.
    dx = vs_gemm_split(dy, pack(w^T)); dw = vs_gemm_wgrad(transpose(dy), transpose_pack(x)) over the M rows; db = column sums.
    dgelu_z [M,K] f32 (x = gelu(z), the MLP's second linear): the returned dx is the gradient of z, dy w^T * GELU'(z), multiplied in the dX
    GEMM's epilogue (vs_gemm_split epilogue 5) instead of a separate pass over dx and z."""
    M, N = dy.shape
    K = x.shape[1]
    dev = dy.device
    assert dy.dtype == torch.float32 and x.dtype == torch.float32 and dy.stride(1) == 1 and x.stride(1) == 1
    dx = dw = db = None
    if need_dx:
        wf = w.detach().float()
        if N % 64 == 0 and scale_exp is not None and wf.is_contiguous():
            dyp = dy if (dy.stride(0) % 4 == 0 and dy.data_ptr() % 16 == 0) else dy.contiguous()
            wtp = transpose_pack_split(wf, 64, scale_exp=int(scale_exp))          # pack(w^T) in one pass: no transposed f32 copy of the weight
        else:
            wt = wf.t()
            if N % 32 != 0:
                wt = torch.nn.functional.pad(wt, (0, (-N) % 32))
                dyp = torch.nn.functional.pad(dy, (0, (-N) % 32))
            else:
                dyp = dy if (dy.stride(0) % 4 == 0 and dy.data_ptr() % 16 == 0) else dy.contiguous()
            wtp = split_pack_weight(wt.contiguous(), scale_exp)
        dx = torch.empty((M, K), dtype=torch.float32, device=dev)
        if dgelu_z is not None and dyp.shape[1] % 64 == 0 and dgelu_z.shape == dx.shape and dgelu_z.is_contiguous() and dgelu_z.dtype == torch.float32:
            _gemm_split(dyp, wtp, None, dx, 5, resid=dgelu_z)
        else:
            _gemm_split(dyp, wtp, None, dx, EPI_STORE32)
            if dgelu_z is not None:
                dx = gelu_backward(dx, dgelu_z.contiguous())
    if need_dw:
        Mp = (M + 1023) // 1024 * 1024 if M >= 4096 else (M + 127) // 128 * 128      # (room for the K split; zero rows cost nothing exact)
        dys = dy if (dy.stride(0) % 4 == 0 and dy.data_ptr() % 16 == 0) else dy.contiguous()
        xs = x if (x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0) else x.contiguous()
        if need_db:
            db = torch.empty(N, dtype=torch.float32, device=dev)          # the bias gradient rides on the transpose of dY
        dw = torch.empty((N, K), dtype=torch.float32, device=dev)
        if _WGRAD_ATN and N % 256 == 0 and K % 256 == 0:
            # dW^T [K, N] = X^T dY with X read as it is (reduction-major A operand); only dY goes through a transposing (packing) pass
            dyTp = transpose_pack_split(dys, Mp, colsum=db)               # [N, Mp] packed
            tiles = (N // 256) * (K // 256)
            ks = max(1, min((256 + tiles // 2) // tiles, Mp // 512))
            while ks > 1 and Mp % (64 * ks) != 0:
                ks -= 1
            gemm_wgrad_split_atn(xs, dyTp, dw, ks, transpose_out=True)
        else:
            dyT = transpose_f32(dys, Mp, colsum=db)      # [N, Mp]
            xT = transpose_pack_split(xs, Mp)            # [K, Mp] packed
            _wgrad_split(dyT, xT, dw)
    if need_db and db is None:
        db = colsum(dy)
    return dx, dw, db


def head1x1_backward(dy: torch.Tensor, t: torch.Tensor, w: torch.Tensor, *, relu: bool = True, scale_exp: int = 0):
    """Backward of y = t @ w^T + b (the last 1x1 convolution of a DPT head) fused with the ReLU backward of t = relu(...), ONE pass over the
    full-resolution tensors (csrc/head_bwd.hip).  dy [P, Cout] (row stride >= Cout allowed: aligned rows of the adapter's backward), t [P, Cin]
    contiguous, both f32 (split class: three f16 MFMAs per product, w packed with 2^scale_exp) or both f16 / bf16; w [Cout, Cin] f32 ->
    dt [P, Cin] = (t > 0) * (dy @ w) in t's dtype, dw [Cout, Cin] = dy^T t and db [Cout] in f32 (per-workgroup partials summed here)."""
    dev = L.require_device(dy, t, w)
    P, Cout = dy.shape
    Cin = t.shape[1]
    assert dy.dtype == t.dtype and w.dtype == torch.float32 and dy.stride(1) == 1 and t.is_contiguous() and w.is_contiguous()
    assert t.shape[0] == P and w.shape == (Cout, Cin) and P % 32 == 0 and Cin in (128, 256) and 1 <= Cout <= 96
    split = dy.dtype == torch.float32
    ldy = dy.stride(0)
    assert Cout <= ldy <= 128 and dy.data_ptr() % 16 == 0
    rows = 96 if Cout > 16 else 16
    per_cu = 2 if (Cin == 128 and Cout <= 16) else 1        # resident workgroups per CU (101 / 206 VGPRs x 8 waves)
    nwg = max(1, min(P // 32, per_cu * torch.cuda.get_device_properties(dev).multi_processor_count))
    dt = torch.empty_like(t)
    dw_part = torch.empty((nwg, rows, Cin), dtype=torch.float32, device=dev)
    db_part = torch.empty((nwg, rows), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        if split:
            rc = L.lib().vs_head1x1_backward_split(L.ptr(dy), ldy, L.ptr(t), L.ptr(w), int(scale_exp), L.ptr(dt), L.ptr(dw_part), L.ptr(db_part), P, Cin,
                                                   Cout, int(relu), nwg, L.stream_ptr(dev))
        else:
            rc = L.lib().vs_head1x1_backward16(L.ptr(dy), ldy, L.ptr(t), L.ptr(w), L.ptr(dt), L.ptr(dw_part), L.ptr(db_part), P, Cin, Cout, int(relu),
                                               nwg, _DTX[dy.dtype], L.stream_ptr(dev))
    L.check(rc, "vs_head1x1_backward")
    return dt, dw_part.sum(0)[:Cout].contiguous(), db_part.sum(0)[:Cout].contiguous()


def head1x1_backward_split(dy, t, w, *, relu: bool = True, scale_exp: int = 0):
    assert dy.dtype == torch.float32
    return head1x1_backward(dy, t, w, relu=relu, scale_exp=scale_exp)


def attention_backward_split(qkv_q: torch.Tensor, qkv_k: torch.Tensor, qkv_v: torch.Tensor, out: torch.Tensor, dout: torch.Tensor, lse: torch.Tensor, *,
                             nbatch: int, H: int, Lq: int, Lk: int = 0, q_batch_rows: int, k_batch_rows: int = 0,
                             kv_seg: Optional[torch.Tensor] = None, q_kvlen: Optional[torch.Tensor] = None, max_keys: int = 0, scale: float = 0.125,
                             dq_out: Optional[torch.Tensor] = None, dk_out: Optional[torch.Tensor] = None, dv_out: Optional[torch.Tensor] = None):
    """Backward of attention(..., split=True): q / k / v f32 views [rows, H*64] (row stride any), out / dout f32.  Returns (dq, dk, dv) f32;
    dq_out / dk_out / dv_out: views to write into (e.g. the blocks of a packed [rows, 3*H*64] gradient).  With key segments dk / dv are
    accumulated with atomics (zeroed here)."""
    dev = L.require_device(qkv_q, qkv_k, qkv_v, out, dout, lse, kv_seg, q_kvlen)
    Cc = H * 64
    for t in (qkv_q, qkv_k, qkv_v, out, dout):
        assert t.dim() == 2 and t.stride(1) == 1 and t.dtype == torch.float32 and t.shape[1] == Cc
    dout = dout if (dout.stride(0) % 4 == 0 and dout.data_ptr() % 16 == 0) else dout.contiguous()
    qh, ql = split16(qkv_q)
    kh, kl = split16(qkv_k)
    vh, vl = split16(qkv_v)
    dh, dl = split16(dout)
    dq = dq_out if dq_out is not None else torch.empty((qkv_q.shape[0], Cc), dtype=torch.float32, device=dev)
    if kv_seg is None:
        dk = dk_out if dk_out is not None else torch.empty((qkv_k.shape[0], Cc), dtype=torch.float32, device=dev)
        dv = dv_out if dv_out is not None else torch.empty((qkv_v.shape[0], Cc), dtype=torch.float32, device=dev)
    else:       # (dk_out / dv_out given: accumulated where they are wanted -- zeroed here -- instead of in fresh buffers that are copied afterwards)
        dk = dk_out.zero_() if dk_out is not None else torch.zeros((qkv_k.shape[0], Cc), dtype=torch.float32, device=dev)
        dv = dv_out.zero_() if dv_out is not None else torch.zeros((qkv_v.shape[0], Cc), dtype=torch.float32, device=dev)
    for t in (dq, dk, dv):
        assert t.dtype == torch.float32 and t.stride(1) == 1
    delta = torch.empty((qkv_q.shape[0], H), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = L.lib().vs_attention_backward_split(L.ptr(qh), L.ptr(ql), L.ptr(kh), L.ptr(kl), L.ptr(vh), L.ptr(vl), L.ptr(dh), L.ptr(dl), L.ptr(out),
                                                 L.ptr(dout), L.ptr(lse), L.ptr(delta), L.ptr(dq), L.ptr(dk), L.ptr(dv), nbatch, H, Lq, Lk, q_batch_rows,
                                                 k_batch_rows, qh.stride(0), kh.stride(0), vh.stride(0), dh.stride(0), out.stride(0), dout.stride(0),
                                                 dq.stride(0), dk.stride(0), dv.stride(0), L.ptr(kv_seg), L.ptr(q_kvlen), max_keys, scale,
                                                 L.stream_ptr(dev))
    L.check(rc, "vs_attention_backward_split")
    return dq, dk, dv


_WGRAD_STREAM = os.environ.get("VS_WGRAD_STREAM", "1") != "0"


def conv3x3_wgrad_split_stream(dy: torch.Tensor, x: torch.Tensor, *, relu_in: bool = False):
    """dw9 [9, Cin, Cout] (tap = ky * 3 + kx) and db [Cout] of a 3x3 convolution (stride 1, pad 1) from x [N,H,W,Cin] and dy [N,H,W,Cout] f32 NHWC,
    split class, one streaming pass (vs_conv3x3_wgrad_split_stream); Cin, Cout multiples of 64, W a multiple of 32."""
    dev = L.require_device(dy, x)
    N, H, W, Cin = x.shape
    Cout = dy.shape[3]
    assert dy.dtype == x.dtype == torch.float32 and dy.is_contiguous() and x.is_contiguous() and dy.shape[:3] == x.shape[:3]
    assert Cin % 64 == 0 and Cout % 64 == 0 and W % 32 == 0
    nblk = (Cin // 64) * (Cout // 64)
    items = N * (W // 32)
    workers = max(8, min(torch.cuda.get_device_properties(dev).multi_processor_count // nblk, (items + 7) // 8 * 8) // 8 * 8)
    dw_part = torch.empty((workers, 9, Cin, Cout), dtype=torch.float32, device=dev)
    db_part = torch.empty((workers, Cout), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = L.lib().vs_conv3x3_wgrad_split_stream(L.ptr(x), L.ptr(dy), L.ptr(dw_part), L.ptr(db_part), N, H, W, Cin, Cout, int(relu_in), workers,
                                                   L.stream_ptr(dev))
    L.check(rc, "vs_conv3x3_wgrad_split_stream")
    return dw_part.sum(0), db_part.sum(0)


def conv3x3_backward_split(dy: torch.Tensor, x: torch.Tensor, w: torch.Tensor, *, relu_in: bool = False, need_dx: bool = True, need_db: bool = True,
                           scale_exp: Optional[int] = None):
    """Backward of conv3x3_nhwc(x, pack(w), relu_in=relu_in) (stride 1, pad 1) in the split class.  dy [N,H,W,Cout], x [N,H,W,Cin] f32 NHWC,
    w the module's [Cout,Cin,3,3] f32 parameter -> dx [N,H,W,Cin], dw [Cout,Cin,3,3], db [Cout], all f32.
      dx: the forward kernel on the spatially flipped, channel-transposed weights (ReLU backward in its epilogue);
      dw: per tap one weight-gradient GEMM dY^T [Cout, pixels] x act(X shifted by the tap)^T [Cin, pixels] -- the shifted operand is
          written by the transposing pack kernel (zero outside the image), so no im2col buffer and no zero-bordered copy exists."""
    N, H, W, Cin = x.shape
    Cout = w.shape[0]
    dev = x.device
    assert dy.dtype == torch.float32 and x.dtype == torch.float32 and dy.is_contiguous() and x.is_contiguous()
    dx = None
    if need_dx:
        wd = w.detach().float().flip(2, 3).permute(1, 2, 3, 0).contiguous()          # [Cin, 3, 3, Cout]: flipped taps, channels swapped
        pad_o = (-Cout) % 32
        dyp = dy
        if pad_o:
            wd = torch.nn.functional.pad(wd, (0, pad_o))
            dyp = torch.nn.functional.pad(dy, (0, pad_o))
        wdp = split_pack_weight(wd, scale_exp)
        dx = conv3x3_nhwc(dyp, wdp, None, mask_by=x if relu_in else None)
    # weight gradient: ONE tap-fused launch over the zero-bordered pixel grid.  A = act(X)^T [Cin, pixels] f32 with a zero halo (the nine tap
    # shifts are column offsets of this operand: LDS-DMA reads any 4-byte aligned f32 address), W = dY^T [Cout, pixels] packed (hi, lo);
    # out[tap] = [Cin, Cout].  Two transposing passes instead of nine packed shifted images.
    P = N * H * W
    if _WGRAD_ATN and Cin % 256 == 0 and Cout % 256 == 0:
        # X read as it is by the reduction-major A operand (tap shift = pixel-row shift, zero page outside the image): only dY goes through a
        # transposing (packing) pass, which also yields the bias gradient
        tiles = (Cin // 256) * (Cout // 256) * 9
        ks = max(1, min((256 + tiles // 2) // tiles, P // 512))
        db = torch.empty(Cout, dtype=torch.float32, device=dev) if need_db else None
        dyTp = transpose_pack_split(dy.view(P, Cout), 64 * ks, colsum=db)                                   # [Cout, Ppad] packed
        dw9 = torch.empty((9, Cin, Cout), dtype=torch.float32, device=dev)
        ws = torch.empty(ks * 9 * Cin * Cout, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = L.lib().vs_conv3x3_wgrad_split_atn(L.ptr(x), L.ptr(dyTp.data), L.ptr(dw9), N, H, W, Cin, Cout, dyTp.data.shape[1], dyTp.data.stride(0),
                                                    int(relu_in), ks, L.ptr(ws), ws.numel() * 4, 0, L.stream_ptr(dev))
        L.check(rc, "vs_conv3x3_wgrad_split_atn")
        return dx, dw9.view(3, 3, Cin, Cout).permute(3, 2, 0, 1).contiguous(), db
    if _WGRAD_STREAM and Cin % 64 == 0 and Cout % 64 == 0 and W % 32 == 0:
        # narrow layers (the pts3d head's 256 -> 128 and 128 -> 128 convolutions): one streaming pass over X and dY as they are
        # (csrc/conv_wgrad_stream.hip) instead of two transposing passes + nine 128 x 128 tiles per K slice on the 4-wave kernel
        dw9, db = conv3x3_wgrad_split_stream(dy, x, relu_in=relu_in)
        return dx, dw9.view(3, 3, Cin, Cout).permute(3, 2, 0, 1).contiguous(), (db if need_db else None)
    Wp = W + 2
    Pb = N * (H + 2) * Wp
    tiles = ((Cin + 255) // 256) * ((Cout + 255) // 256) * 9 if (Cin % 256 == 0 and Cout % 256 == 0) else ((Cin + 127) // 128) * ((Cout + 127) // 128) * 9
    if Cin % 256 == 0 and Cout % 256 == 0:
        ks = max(1, min((256 + tiles // 2) // tiles, Pb // 512))
    else:
        ks = max(2, min(768 // tiles, Pb // 512, 1024))
    unit = 64 * ks * (2 if (Cin % 256 == 0 and Cout % 256 == 0) else 1)
    halo = (Wp + 2 + 3) // 4 * 4
    xT = transpose_f32(x.view(P, Cin), unit, relu=relu_in, conv_hw=(H, W), border=True, halo=halo)          # [Cin, Pp] view of the haloed buffer
    db = torch.empty(Cout, dtype=torch.float32, device=dev) if need_db else None                            # (rides on the transpose of dY)
    dyT = transpose_pack_split(dy.view(P, Cout), unit, conv_hw=(H, W), border=True, colsum=db)              # [Cout, Pp] packed
    dw9 = torch.empty((9, Cin, Cout), dtype=torch.float32, device=dev)
    shifts = [(ty - 1) * Wp + (tx - 1) for ty in range(3) for tx in range(3)]
    gemm_wgrad_split(xT, dyT, dw9, ks, shifts=shifts)
    dw = dw9.view(3, 3, Cin, Cout).permute(3, 2, 0, 1).contiguous()                                          # [Cout, Cin, ky, kx]
    return dx, dw, db


def sustained_mfma_tflops(device, ms_target: float = 40.0) -> float:
    """TFLOP/s the chip sustains on back-to-back v_mfma_f32_16x16x32_f16 with register operands holding random data and no memory traffic
    (vs_probe_mfma_rate): the power-limited ceiling of every 16-bit MFMA kernel on this device, measured, beside the 2.5 PFLOP/s headline."""
    import ctypes
    dev = torch.device(device)
    src = (torch.rand(1 << 19, device=dev) * 4 - 2).half()
    scratch = torch.empty(512 * 256, dtype=torch.float32, device=dev)
    fl = ctypes.c_double(0.0)
    def run(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.device(dev):
            s.record()
            rc = L.lib().vs_probe_mfma_rate(L.ptr(src), L.ptr(scratch), iters, ctypes.byref(fl), L.stream_ptr(dev))
            e.record()
        L.check(rc, "vs_probe_mfma_rate")
        e.synchronize()
        return s.elapsed_time(e), fl.value
    ms, f = run(20000)                                   # warm-up and calibration (clocks settle within a few ms)
    iters = max(20000, int(20000 * ms_target / max(ms, 1e-3)))
    ms, f = run(iters)
    return f / (ms * 1e-3) / 1e12
