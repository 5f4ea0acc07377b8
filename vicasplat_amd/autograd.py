"""torch.autograd.Function wrappers of the HIP operators, forward AND backward on the hand-written kernels
(`vicasplat_amd.ops`).  They are what the training forward (`model/encoder/train_forward.py`) is written in, so that the
glue between the hot operators -- residual adds, reshapes, the tiny f32 camera-token math -- is differentiated by
PyTorch while every GEMM, LayerNorm, attention, RoPE and 3x3 convolution runs on gfx950 in both directions.

Reference: the training step differentiates the encoder with torch autograd (model_wrapper.py:184-321); these Functions
are the MI355X replacements of the autograd nodes of nn.Linear, nn.LayerNorm (+AdaLN modulation), RoPE2D / temporal RoPE,
scaled-dot-product attention and nn.Conv2d(k=3, s=1, p=1) on that path.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import ops

SPLIT = "split"      # compute-dtype tag of the split operand class: f32 tensors, three f16 MFMAs per product (csrc/gemm_common.h kDtSplit)


class BoundaryGradScale:
    """Internal power-of-two gradient scale of the Module API under autograd (VicaSplat.forward with grad enabled).

    The reference trains in fp32 (model_wrapper.py:184-321, trainer precision 32): `loss.backward()` with no loss scale.  Here the MFMA
    operands of the backward are 16-bit (f16 class) or f16 (hi, lo) pairs (split class), so gradients must sit in f16's range while they
    pass through the encoder.  `outputs(...)` multiplies the COTANGENTS of the encoder's outputs by S = 2^k on their way in (exact);
    parameter gradients therefore accumulate in units of S; when the backward pass ends (autograd engine callback) every parameter
    gradient is multiplied by 1/S (exact), so the caller observes plain unscaled `.grad`s as from the reference.  Gradients that were
    already in `.grad` before this backward (accumulation over micro-batches) are lifted by S first, so the sum stays consistent.  The
    callback re-queues itself once so that it runs after DDP's own end-of-backward callback (the bucket all-reduce runs on scaled values:
    exact for a power of two).  S = 1 disables all of it.

    * `inputs(...)`: non-parameter leaves (the image, the intrinsics) enter through the inverse node -- their cotangent leaves the encoder
      multiplied by 1/S, so d(loss)/d(image) is plain as well.
    * The set of parameters is re-read from `params` (the module itself, or a fixed list) at the START of every
      backward pass: parameters unfrozen after the scaler was built are unscaled like the rest.
    * Under this package's `dist.GradReducer` the `.grad`s are views of flat buckets whose all-reduces are still in flight when the
      callback runs: the unscale is then handed to the reducer (`defer_unscale`), which applies it in `finish()` after the wait, together
      with the averaging -- never on a buffer a collective is reading.
    * Overflow: the unscale pass also checks the gradients for inf / NaN on the device (one fused pass); the flag of the last backward is
      `last_overflow` (a 0-d device tensor, no host synchronisation here; under a GradReducer: `reducer.last_overflow` after `finish()`).  A cotangent with |dy| * S >= 65520 (e.g. a
      sum-reduced loss over many pixels) leaves f16's range where the reference's fp32 backward has it; lower `encoder.grad_scale` then.
    * `torch.autograd.grad(loss, params)` bypasses the end-of-backward callback: it returns gradients in units of S (divide by `scale`)."""

    def __init__(self, params, scale: float):
        self._module = params if isinstance(params, torch.nn.Module) else None      # re-read at every backward
        self._list = None if self._module is not None else list(params)
        self.scale = float(scale)
        m, e = __import__("math").frexp(self.scale)
        assert self.scale >= 1.0 and m == 0.5, "the boundary gradient scale must be a power of two"
        self._armed = False
        self._active = []
        self.last_overflow = None

    @property
    def params(self):
        src = self._module.parameters() if self._module is not None else self._list
        return [p for p in src if p.requires_grad]

    def _begin(self):
        if self._armed:
            return
        self._armed = True
        self._active = self.params
        # gradients already in .grad (accumulation over micro-batches) are lifted to units of S -- except those in a GradReducer's buckets
        # whose deferred unscale is still pending (a second backward before finish()): they ARE in units of S already
        from .dist import reducer_of
        old = []
        for p in self._active:
            if p.grad is None:
                continue
            red = reducer_of(p)
            if red is None or not red.pending_unscale(p):
                old.append(p.grad)
        if old:
            torch._foreach_mul_(old, self.scale)
        torch.autograd.Variable._execution_engine.queue_callback(self._requeue)

    def _requeue(self):
        torch.autograd.Variable._execution_engine.queue_callback(self._end)

    def _end(self):
        self._armed = False
        active, self._active = self._active, []
        inv = 1.0 / self.scale
        from .dist import reducer_of
        plain, reducers = [], {}
        for p in active:
            if p.grad is None:
                continue
            red = reducer_of(p)
            if red is not None:
                reducers.setdefault(id(red), (red, []))[1].append(p)
            else:
                plain.append(p.grad)
        for red, ps in reducers.values():      # gradients living in a GradReducer's buckets: unscaled by finish(), after the collectives --
            red.defer_unscale(inv, ps)         # exactly these parameters, not every bucket of the reducer
        if plain:
            with torch.no_grad():
                self.last_overflow = unscale_and_check_(plain, inv)

    def rearm(self):
        """Called at every forward: a backward pass that raised leaves `_armed` set (its callbacks never ran)."""
        self._armed = False
        self._active = []

    def outputs(self, *tensors):
        """Identity on the values; cotangents x S (None stays None).  One autograd node PER tensor: an output no loss reads (the poses
        without a camera loss, raw_gaussians, scales / rotations) then never enters the backward pass -- a shared node would hand its
        producers materialised zero cotangents, i.e. zero `.grad`s where the reference leaves None."""
        if self.scale == 1.0:
            return tensors
        return tuple(None if t is None else _BoundaryScaleFn.apply(self, t, self.scale) for t in tensors)

    def inputs(self, *tensors):
        """Identity on the values of non-parameter inputs that require grad; their cotangents x 1/S on the way OUT of the encoder."""
        if self.scale == 1.0:
            return tensors
        return tuple(t if (t is None or not t.requires_grad) else _BoundaryScaleFn.apply(None, t, 1.0 / self.scale) for t in tensors)


def unscale_and_check_(grads, inv_scale: float) -> torch.Tensor:
    """grads *= inv_scale in place (exact for a power of two) and a 0-d float flag (> 0: some gradient holds inf / NaN), one fused pass per
    device / dtype group on the GPU (the AMP foreach primitive), no host synchronisation."""
    dev = grads[0].device
    found = torch.zeros((), dtype=torch.float32, device=dev)
    if dev.type == "cuda" and all(g.device == dev for g in grads):
        by_dtype = {}
        for g in grads:
            by_dtype.setdefault(g.dtype, []).append(g)
        inv = torch.full((), inv_scale, dtype=torch.float32, device=dev)
        for gs in by_dtype.values():
            torch._amp_foreach_non_finite_check_and_unscale_(gs, found, inv)
        return found
    torch._foreach_mul_(grads, inv_scale)
    for g in grads:
        found = torch.maximum(found, (~torch.isfinite(g)).any().to(device=dev, dtype=torch.float32))
    return found


class _BoundaryScaleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scaler, t, factor):
        ctx.scaler, ctx.factor = scaler, factor
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        if ctx.scaler is not None:
            ctx.scaler._begin()
        return None, g * ctx.factor, None


def act_dtype(dt):
    """torch dtype of the activations of operand class `dt`."""
    return torch.float32 if dt == SPLIT else dt


class LinearFn(torch.autograd.Function):
    """y16 = x @ w^T + b on the MFMA GEMM; backward = dgrad / wgrad on the same kernels (ops.linear_backward).
    rope = (pos, kind, H, C, base2d, theta1d): w is a packed q | k | v projection and the rotary embedding of q and k runs in the
    GEMM epilogue (vs_gemm_qkv_rope); the backward applies the inverse rotation to dq | dk before the usual dgrad / wgrad -- in place
    only when the incoming gradient is the fresh buffer AttentionFn.backward produced for this node alone (it tags it), on a
    clone otherwise (retain_grad / tensor hooks / a second consumer must never observe a rotated gradient)."""

    @staticmethod
    def forward(ctx, x, w, b, dt, rope):
        K = x.shape[-1]
        x16 = x.reshape(-1, K).to(dt).contiguous()
        w16 = w.detach().to(dt).contiguous()
        bf = None if b is None else b.detach().float().contiguous()
        y = torch.empty((x16.shape[0], w16.shape[0]), dtype=dt, device=x.device)
        if rope is None:
            ops.gemm(x16, w16, bf, y, ops.EPI_STORE16)
        else:
            pos, kind, H, C, base2d, theta1d = rope
            ops.gemm_qkv_rope(x16, w16, bf, y, C, pos, kind, base2d, theta1d)
        ctx.save_for_backward(x16, w16)
        ctx.meta = (x.shape, x.dtype, b is not None, rope)
        return y if x.dim() == 2 else y.view(*x.shape[:-1], w16.shape[0])   # (no view node in front of a 2-D result: see backward)

    @staticmethod
    def backward(ctx, dy):
        x16, w16 = ctx.saved_tensors
        xshape, xdtype, has_b, rope = ctx.meta
        dy16 = dy.reshape(-1, dy.shape[-1]).to(w16.dtype)
        if dy16.stride(1) != 1 or dy16.stride(0) % 8 != 0 or dy16.data_ptr() % 16 != 0:
            dy16 = dy16.contiguous()          # (row-padded gradients -- 16-byte aligned rows -- are consumed as they are)
        if rope is not None:
            pos, kind, H, C, base2d, theta1d = rope
            if dy16.data_ptr() == dy.data_ptr() and not getattr(dy, "_vs_owned_grad", False):
                dy16 = dy16.clone()       # never mutate a gradient autograd handed in, unless its producer marked it as ours alone
            ops.rope_qk(dy16, H, C, pos, kind, base2d, theta1d, inverse=True)
        dx, dw, db = ops.linear_backward(dy16, x16, w16, need_dx=ctx.needs_input_grad[0], need_dw=ctx.needs_input_grad[1],
                                         need_db=has_b and ctx.needs_input_grad[2])
        return (None if dx is None else dx.view(xshape).to(xdtype)), dw, db, None, None


_MLP_DGELU_EPI = os.environ.get("VS_MLP_DGELU_EPI", "1") != "0"      # A/B switch: 0 = GELU and its backward as separate nodes in the split class too


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], dt: torch.dtype, rope=None, scale_sources=(), gelu_in: bool = False) -> torch.Tensor:
    """nn.Linear in the operand dtype `dt`.  K must be a multiple of 64 for the MFMA kernels; tiny odd shapes (the 9 -> C
    intrinsic embedding) stay on torch in f32.  gelu_in: y = gelu(x) @ w^T + b (croco/blocks.py:60-72: fc2(act(fc1(x)))) -- in the split class ONE
    node whose backward multiplies by GELU'(x) in the dX GEMM's epilogue."""
    if gelu_in and not (dt == SPLIT and _MLP_DGELU_EPI and rope is None):
        x, gelu_in = gelu(x), False
    if dt == SPLIT:
        return LinearSplitFn.apply(x, w, b, rope, tuple(scale_sources), gelu_in)
    K = x.shape[-1]
    if K % 64 != 0:
        assert rope is None
        if K < 32:
            return torch.nn.functional.linear(x.float(), w, b).to(dt)
        pad = (K + 63) // 64 * 64 - K        # e.g. the 96-channel reassemble stage: zero-pad the reduction dimension
        x, w = torch.nn.functional.pad(x, (0, pad)), torch.nn.functional.pad(w, (0, pad))
    return LinearFn.apply(x, w, b, dt, rope)


class LinearSplitFn(torch.autograd.Function):
    """y = x @ w^T + b in the SPLIT operand class (f32 in / out, every product three f16 MFMAs on hi+lo pairs: f32-class results; csrc/
    gemm_common.h kDtSplit), forward AND backward (ops.linear_backward_split: dx on the packed transposed weight, dw = dy^T x on the split
    weight-gradient GEMM over transposed / packed operands, db = column sums): the reference-precision nn.Linear of the training step, and
    the tiny f32 layers of the camera-token path without a vendor-BLAS launch in either direction.
    rope = (pos, kind, H, C, base2d, theta1d): packed q | k | v projection with the rotary embedding in the GEMM epilogue; the backward
    un-rotates dq | dk first (on a copy unless the producer marked the gradient buffer as ours alone)."""

    @staticmethod
    def _pad32(t, dim):
        n = t.shape[dim]
        pad = (-n) % 32
        if pad == 0:
            return t.contiguous()
        cfg = [0, 0] * (t.dim() - 1 - dim) + [0, pad]
        return torch.nn.functional.pad(t, cfg).contiguous()

    # power-of-two scale of each parameter's packed image, refreshed every 64 uses: reading max|w| is a host synchronisation, and these
    # layers are packed every step (the weights move).  2^e max|w| starts in [2^13, 2^14): a weight may grow 4x between refreshes (AdamW at
    # the reference's learning rates moves a weight by <= lr per step).  What the countdown cannot see is a REPLACEMENT of the values
    # under the same storage: load_state_dict / copy_ -- VicaSplat clears the cache from a load_state_dict post-hook, and
    # `clear_split_caches()` is there for callers that overwrite parameters by hand.  All-zero weights (the zero-initialised pose / fov
    # heads) are never cached: their exponent is re-read on every use until they have moved.
    _exp_cache: dict = {}

    @staticmethod
    def _scale_exp(w, *sources):
        """Exponent for packing `w`.  `sources`: the parameters a temporary `w` was assembled from (concatenation / re-layout): the
        exponent is then the smallest of theirs -- max|w| <= max over the sources -- and no host read happens for the temporary."""
        if sources:
            return min(LinearSplitFn._scale_exp(s_) for s_ in sources)
        base = w if w.is_leaf else w._base
        if base is None or not base.is_leaf:     # a temporary with unknown sources: nothing stable to key a cache on
            return ops.split_scale_exp(w)
        import weakref
        key = (id(base), w.data_ptr(), tuple(w.shape))
        ent = LinearSplitFn._exp_cache.get(key)
        if ent is None or ent[1] <= 0 or ent[2]() is not base:     # (the weak reference guards against a recycled id / address of a freed parameter)
            if len(LinearSplitFn._exp_cache) > 4096:
                LinearSplitFn._exp_cache.clear()
            e, nonzero = ops.split_scale_exp(w, with_nonzero=True)
            if not nonzero:
                LinearSplitFn._exp_cache.pop(key, None)
                return e
            ent = [e, 64, weakref.ref(base)]
            LinearSplitFn._exp_cache[key] = ent
        ent[1] -= 1
        return ent[0]

    @staticmethod
    def forward(ctx, x, w, b, rope=None, scale_sources=(), gelu_in=False):
        K, N = x.shape[-1], w.shape[0]
        x2 = x.reshape(-1, K).float()
        ctx.gelu_in = bool(gelu_in)
        if gelu_in:                       # x is the pre-activation z: a = gelu(z) is this linear's input; z is kept for GELU'(z) in the backward
            z = x2.contiguous()
            x2 = ops.gelu16(z)
        xp = LinearSplitFn._pad32(x2, 1)
        e = LinearSplitFn._scale_exp(w, *scale_sources)
        ctx.scale_exp = e
        wp = ops.split_pack_weight(LinearSplitFn._pad32(w.detach().float(), 1), e)
        y = torch.empty((x2.shape[0], N), dtype=torch.float32, device=x.device)
        bf = None if b is None else b.detach().float().contiguous()
        if rope is None:
            ops.gemm(xp, wp, bf, y, ops.EPI_STORE32)
        else:
            pos, kind, H, C, base2d, theta1d = rope
            ops.gemm_qkv_rope(xp, wp, bf, y, C, pos, kind, base2d, theta1d)
        if gelu_in:
            ctx.save_for_backward(xp, w, z)
        else:
            ctx.save_for_backward(xp, w)
        ctx.meta = (x.shape, b is not None, rope)
        return y if x.dim() == 2 else y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        z = None
        if ctx.gelu_in:
            xp, w, z = ctx.saved_tensors
        else:
            xp, w = ctx.saved_tensors
        xshape, has_b, rope = ctx.meta
        N, K = w.shape
        dy2 = dy.reshape(-1, N).float()
        if dy2.stride(1) != 1 or dy2.stride(0) % 4 != 0 or dy2.data_ptr() % 16 != 0:
            dy2 = dy2.contiguous()
        if rope is not None:
            pos, kind, H, C, base2d, theta1d = rope
            if dy2.data_ptr() == dy.data_ptr() and not getattr(dy, "_vs_owned_grad", False):
                dy2 = dy2.clone()
            ops.rope_qk(dy2, H, C, pos, kind, base2d, theta1d, inverse=True)
        Kp = xp.shape[1]
        wk = w.detach().float() if Kp == K else torch.nn.functional.pad(w.detach().float(), (0, Kp - K))
        if z is not None and Kp != K:      # (padded reduction dimension: GELU'(z) after the slice below)
            zz, z = z, None
        else:
            zz = None
        dx, dw, db = ops.linear_backward_split(dy2, xp, wk, need_dx=ctx.needs_input_grad[0], need_dw=ctx.needs_input_grad[1],
                                               need_db=has_b and ctx.needs_input_grad[2], scale_exp=ctx.scale_exp, dgelu_z=z)
        if dx is not None:
            dx = dx[:, :K].reshape(xshape) if Kp != K else dx.view(xshape)
            if zz is not None:
                dx = ops.gelu_backward(dx.reshape(-1, K).contiguous(), zz).view(xshape)
        if dw is not None and Kp != K:
            dw = dw[:, :K].contiguous()
        return dx, dw, db, None, None, None


def linear_split(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], rope=None, scale_sources=()) -> torch.Tensor:
    return LinearSplitFn.apply(x, w, b, rope, tuple(scale_sources), False)


def _tail_dy(dy, N, dtype):
    """The incoming gradient as [P, N] rows the fused kernel reads in place: unit column stride, a 16-byte aligned base, row stride <= 128
    elements (the adapter's backward pads its rows for alignment: the padding is skipped, not copied away)."""
    dy2 = dy.reshape(-1, N)
    if dy2.dtype != dtype:
        dy2 = dy2.to(dtype)
    if dy2.stride(1) != 1 or dy2.data_ptr() % 16 != 0 or not (N <= dy2.stride(0) <= 128):
        dy2 = dy2.contiguous()
    return dy2


class HeadTailSplitFn(torch.autograd.Function):
    """y = t @ w^T + b for the LAST 1x1 convolution of a DPT head whose input t is the output of a ReLU (Conv3x3Fn(..., relu_out=True)), split
    class.  Forward = LinearSplitFn's GEMM; backward = ONE fused pass (ops.head1x1_backward, csrc/head_bwd.hip): dt comes back ALREADY
    masked by t > 0 -- the producer's ReLU backward, which Conv3x3Fn.backward then skips (`_vs_relu_masked`) -- with dw and db from the same
    read of dy and t.  Only valid behind a ReLU (where t == 0 the producer's own backward would zero the gradient anyway)."""

    @staticmethod
    def forward(ctx, t, w, b):
        K, N = t.shape[-1], w.shape[0]
        t2 = t.reshape(-1, K)
        e = LinearSplitFn._scale_exp(w)
        ctx.scale_exp = e
        wp = ops.split_pack_weight(LinearSplitFn._pad32(w.detach().float(), 1), e)
        y = torch.empty((t2.shape[0], N), dtype=torch.float32, device=t.device)
        ops.gemm(t2, wp, None if b is None else b.detach().float().contiguous(), y, ops.EPI_STORE32)
        ctx.save_for_backward(t2, w)
        ctx.meta = (t.shape, b is not None)
        return y.view(*t.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        t2, w = ctx.saved_tensors
        tshape, has_b = ctx.meta
        dt, dw, db = ops.head1x1_backward(_tail_dy(dy, w.shape[0], torch.float32), t2, w.detach().float().contiguous(), relu=True,
                                          scale_exp=ctx.scale_exp)
        dt = dt.view(tshape)
        dt._vs_relu_masked = True      # Conv3x3Fn.backward: the trailing ReLU's mask is already applied
        return dt, dw, (db if has_b else None)


class HeadTail16Fn(torch.autograd.Function):
    """HeadTailSplitFn in the 16-bit operand classes (f16 / bf16 activations, one MFMA per product): LinearFn's forward GEMM, the fused backward."""

    @staticmethod
    def forward(ctx, t, w, b, dt):
        K = t.shape[-1]
        t2 = t.reshape(-1, K)
        w16 = w.detach().to(dt).contiguous()
        y = torch.empty((t2.shape[0], w16.shape[0]), dtype=dt, device=t.device)
        ops.gemm(t2, w16, None if b is None else b.detach().float().contiguous(), y, ops.EPI_STORE16)
        ctx.save_for_backward(t2, w)
        ctx.meta = (t.shape, b is not None)
        return y.view(*t.shape[:-1], w16.shape[0])

    @staticmethod
    def backward(ctx, dy):
        t2, w = ctx.saved_tensors
        tshape, has_b = ctx.meta
        dt, dw, db = ops.head1x1_backward(_tail_dy(dy, w.shape[0], t2.dtype), t2, w.detach().float().contiguous(), relu=True)
        dt = dt.view(tshape)
        dt._vs_relu_masked = True
        return dt, dw, (db if has_b else None), None


def head_tail_ok(t: torch.Tensor, w: torch.Tensor) -> bool:
    """Shapes csrc/head_bwd.hip serves: [.., Cin] contiguous rows (f32 = split class, f16, bf16), Cin 128 | 256, Cout <= 96, pixels a multiple of 32."""
    return (t.dtype in (torch.float32, torch.float16, torch.bfloat16) and t.is_contiguous() and t.shape[-1] in (128, 256) and w.shape[0] <= 96
            and w.shape[1] == t.shape[-1] and (t.numel() // t.shape[-1]) % 32 == 0 and _behind_relu(t))


def _behind_relu(t: torch.Tensor) -> bool:
    """The fused tail hands its input gradient over ALREADY masked by t > 0 and tells the producer to skip its own ReLU backward
    (`_vs_relu_masked`): only valid when t IS the output of Conv3x3Fn(..., relu_out=True).  The producer is identified by its autograd node
    (a custom Function's grad_fn is its ctx), not by convention at the call site; a t outside any graph has no consumer of dt."""
    fn = t.grad_fn
    return fn is None or (isinstance(fn, Conv3x3Fn._backward_cls) and bool(getattr(fn, "relu_out", False)))


def head_tail(t: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], dt) -> torch.Tensor:
    if not _behind_relu(t):
        raise ValueError("autograd.head_tail: t must be the output of conv3x3(..., relu_out=True) -- the fused backward masks dt by t > 0 and the "
                         "producer skips its ReLU backward; use autograd.linear for any other input")
    return HeadTailSplitFn.apply(t, w, b) if dt == SPLIT else HeadTail16Fn.apply(t, w, b, dt)


def clear_split_caches() -> None:
    """Forget the cached power-of-two weight exponents of the split class (call after overwriting parameter VALUES in place by hand;
    load_state_dict on a VicaSplat does it by itself)."""
    LinearSplitFn._exp_cache.clear()


class LayerNormModFn(torch.autograd.Function):
    """out = LN(x; w, b) [* (1 + scale[row // mod_rows]) + shift[...]], x f32 [M,C] -> out in `out_dtype`.
    lead (optional, [M // lead_rows, C]): the output gets one extra row in FRONT of every lead_rows rows holding lead (the
    decoder's camera token in front of each frame's image tokens, backbone_vica.py:95-118) -- the kernel writes the LayerNorm
    rows straight into that interleaved buffer, so no concatenation pass exists."""

    @staticmethod
    def forward(ctx, x, w, b, scale, shift, mod_rows, out_dtype, eps, lead, lead_rows, skip=False):
        M, C = x.shape
        ctx.skip = bool(skip)
        ctx.set_materialize_grads(False)          # (an unused output's gradient stays None instead of a zero tensor)
        xf = x.float().contiguous()
        sc = None if scale is None else scale.detach().float().contiguous()
        sh = None if shift is None else shift.detach().float().contiguous()
        wf, bf = w.detach().float().contiguous(), b.detach().float().contiguous()
        grp = (0, 0, 0)
        if lead is None:
            out = torch.empty((M, C), dtype=out_dtype, device=x.device)
        else:
            nf = M // lead_rows
            out = torch.empty((nf * (lead_rows + 1), C), dtype=out_dtype, device=x.device)
            out.view(nf, lead_rows + 1, C)[:, 0] = lead.detach().reshape(nf, C)
            grp = (lead_rows, lead_rows + 1, 1)
        ops.layernorm_mod(xf, wf, bf, out, eps=eps, scale=sc, shift=sh, mod_rows=mod_rows, grp_in=grp[0], grp_out=grp[1], grp_off=grp[2])
        ctx.save_for_backward(xf, wf, bf, sc)
        ctx.meta = (mod_rows, eps, x.dtype, scale is not None, grp, None if lead is None else (lead.shape, lead.dtype))
        if skip:      # (out, x): the residual stream leaves through THIS node too, so that its backward sees both gradients of x and adds the
            return out, x.view_as(x)      # residual one inside the LayerNorm backward kernel (dx_add) instead of in a separate autograd add pass
        return out

    @staticmethod
    def backward(ctx, dout, dskip=None):
        xf, w, b, sc = ctx.saved_tensors
        mod_rows, eps, xdtype, has_mod, grp, lead_meta = ctx.meta
        if dout is None:          # only the residual stream was used downstream
            return dskip, None, None, None, None, None, None, None, None, None, None
        dout = dout.contiguous()
        add = None
        if dskip is not None and dskip.dtype == torch.float32 and dskip.shape == xf.shape and dskip.stride(1) == 1:
            add, dskip = dskip, None
        dx, dw, db, dsc, dsh = ops.layernorm_backward(dout, xf, w, b, scale=sc, mod_rows=mod_rows, eps=eps, grp_in=grp[0], grp_out=grp[1],
                                                      grp_off=grp[2], dx_add=add)
        dlead = None
        if lead_meta is not None:
            dlead = dout.view(-1, grp[1], dout.shape[1])[:, 0].to(lead_meta[1]).reshape(lead_meta[0])
        dx = dx.to(xdtype)
        if dskip is not None:
            dx = dx + dskip.to(xdtype).reshape(dx.shape)
        return dx, dw, db, (dsc if has_mod else None), (dsh if has_mod else None), None, None, None, dlead, None, None


_LN_SKIP = os.environ.get("VS_LN_SKIP_FUSED", "1") != "0"       # A/B switch: 0 = the residual gradient meets the LayerNorm's in an autograd add pass


def layernorm_mod(x, w, b, *, scale=None, shift=None, mod_rows=0, out_dtype=torch.float16, eps=1e-6, lead=None, lead_rows=0, skip=False):
    """x [..., C] (any leading dims; scale/shift [G, C] apply to consecutive groups of mod_rows rows).  With `lead` the result is
    2-D [rows + rows // lead_rows, C] (see LayerNormModFn).  skip=True (x 2-D): returns (LN(x), x') with x' = x routed through the node -- use x' as
    the residual operand that follows (x' + branch(LN(x))): the backward then adds the residual gradient inside its kernel."""
    lead_shape = x.shape[:-1]
    if skip:
        assert x.dim() == 2
        if not _LN_SKIP:
            return layernorm_mod(x, w, b, scale=scale, shift=shift, mod_rows=mod_rows, out_dtype=out_dtype, eps=eps, lead=lead, lead_rows=lead_rows), x
        return LayerNormModFn.apply(x, w, b, scale, shift, mod_rows, out_dtype, eps, lead, lead_rows, True)
    y = LayerNormModFn.apply(x.reshape(-1, x.shape[-1]), w, b, scale, shift, mod_rows, out_dtype, eps, lead, lead_rows)
    return y if lead is not None else y.view(*lead_shape, x.shape[-1])


class RopeQKFn(torch.autograd.Function):
    """Rotary embedding of the q | k blocks of a packed [rows, 3C] projection; backward = the inverse rotation."""

    @staticmethod
    def forward(ctx, qkv, pos, kind, H, C, base2d, theta1d):
        out = qkv.contiguous().clone()
        ops.rope_qk(out, H, C, pos, kind, base2d, theta1d)
        ctx.meta = (pos, kind, H, C, base2d, theta1d)
        return out

    @staticmethod
    def backward(ctx, dy):
        pos, kind, H, C, base2d, theta1d = ctx.meta
        g = dy.contiguous().clone()
        ops.rope_qk(g, H, C, pos, kind, base2d, theta1d, inverse=True)
        return g, None, None, None, None, None, None


class AttentionFn(torch.autograd.Function):
    """softmax(q k^T / 8) v on packed q | k | v [rows, 3C] (head_dim 64) with the key-prefix mask / key segments of
    ops.attention; backward = ops.attention_backward (flash style, logsumexp saved)."""

    @staticmethod
    def forward(ctx, qkv, nbatch, H, Lq, Lk, q_batch_rows, k_batch_rows, kv_seg, q_kvlen, max_keys):
        C = H * 64
        qkv = qkv.contiguous()
        out = torch.empty((qkv.shape[0], C), dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty((qkv.shape[0], H), dtype=torch.float32, device=qkv.device)
        ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, nbatch=nbatch, H=H, Lq=Lq, Lk=Lk, q_batch_rows=q_batch_rows,
                      k_batch_rows=k_batch_rows, kv_seg=kv_seg, q_kvlen=q_kvlen, lse=lse, split=qkv.dtype == torch.float32)
        ctx.save_for_backward(qkv, out, lse)
        ctx.meta = (nbatch, H, Lq, Lk, q_batch_rows, k_batch_rows, kv_seg, q_kvlen, max_keys)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        nbatch, H, Lq, Lk, qbr, kbr, kv_seg, q_kvlen, max_keys = ctx.meta
        C = H * 64
        dqkv = torch.empty((qkv.shape[0], 3 * C), dtype=qkv.dtype, device=qkv.device)   # dq lands in its block directly; dk / dv are f32
        if qkv.dtype == torch.float32:     # split class (f32 tensors are never run on the exact-f32 MFMA in the training step)
            kw = dict(nbatch=nbatch, H=H, Lq=Lq, Lk=Lk, q_batch_rows=qbr, k_batch_rows=kbr, q_kvlen=q_kvlen, dq_out=dqkv[:, :C])
            if kv_seg is None:
                ops.attention_backward_split(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, dout.contiguous(), lse, dk_out=dqkv[:, C:2 * C],
                                             dv_out=dqkv[:, 2 * C:], **kw)
            else:       # (key segments: dk / dv meet through f32 atomics -- in their blocks of dqkv, zeroed by the call: no fresh buffers + copies)
                ops.attention_backward_split(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, dout.contiguous(), lse, kv_seg=kv_seg,
                                             max_keys=max_keys, dk_out=dqkv[:, C:2 * C], dv_out=dqkv[:, 2 * C:], **kw)
        elif kv_seg is None:   # every K/V row has one owner: dk / dv land in their blocks directly (no zero fill, atomics or cast pass)
            ops.attention_backward(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, dout.contiguous(), lse, nbatch=nbatch, H=H, Lq=Lq, Lk=Lk,
                                   q_batch_rows=qbr, k_batch_rows=kbr, q_kvlen=q_kvlen, dq_out=dqkv[:, :C], dk_out=dqkv[:, C:2 * C],
                                   dv_out=dqkv[:, 2 * C:])
        else:
            _, dk, dv = ops.attention_backward(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, dout.contiguous(), lse, nbatch=nbatch, H=H,
                                               Lq=Lq, Lk=Lk, q_batch_rows=qbr, k_batch_rows=kbr, kv_seg=kv_seg, q_kvlen=q_kvlen,
                                               max_keys=max_keys, dq_out=dqkv[:, :C])
            dqkv[:, C:2 * C] = dk                                                        # (one cast-copy each)
            dqkv[:, 2 * C:] = dv
        dqkv._vs_owned_grad = True     # fresh buffer with a single consumer: LinearFn.backward may un-rotate it in place
        return dqkv, None, None, None, None, None, None, None, None, None


class Conv3x3Fn(torch.autograd.Function):
    """nn.Conv2d(k=3, s=1|2, p=1) on NHWC 16-bit activations with the ResidualConvUnit's activation-before-conv (relu_in), the
    unit's skip connection (residual, added to the output) and a trailing ReLU (relu_out) fused into the kernel; w is the module's
    [Cout, Cin, 3, 3] f32 parameter.  Backward: ops.conv3x3_backward.  The stride-2 conv is the stride-1 conv sampled at even
    pixels, so its backward is the stride-1 backward of the zero-dilated gradient."""

    @staticmethod
    def forward(ctx, x, w, b, relu_in, stride, residual, relu_out):
        dt = x.dtype
        split = dt == torch.float32
        if split:     # split class: the module's f32 weight packed per call (scale exponent cached, see LinearSplitFn)
            e = LinearSplitFn._scale_exp(w)
            wp = ops.split_pack_weight(w.detach().float().permute(0, 2, 3, 1).contiguous(), e)
            ctx.scale_exp = e
        else:
            wp = ops.pack_conv3x3_weight(w, dt)
        x = x.contiguous()
        res = None if residual is None else residual.contiguous()
        y = ops.conv3x3_nhwc(x, wp, None if b is None else b.detach().float().contiguous(), residual=res, relu_in=relu_in,
                             relu_out=relu_out, stride=stride)
        if split:
            ctx.save_for_backward(x, w, y if relu_out else None)
        else:
            ctx.save_for_backward(x, wp, y if relu_out else None)
        ctx.meta = (relu_in, b is not None, stride, residual is not None, split)
        ctx.relu_out = bool(relu_out)      # read by head_tail_ok / head_tail through y.grad_fn
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wp, y = ctx.saved_tensors
        relu_in, has_b, stride, has_res, split = ctx.meta
        dy = dy.contiguous()
        if y is not None and not getattr(dy, "_vs_relu_masked", False):   # trailing ReLU: gradient only where the output is positive
            dy = ops.relu_mask(dy, y)                                     # (HeadTailSplitFn hands its input gradient over masked)
        dres = dy if has_res else None
        if stride != 1:
            full = torch.zeros(x.shape[:3] + (dy.shape[3],), dtype=dy.dtype, device=dy.device)
            full[:, ::stride, ::stride] = dy
            dy = full
        if split:
            dx, dw, db = ops.conv3x3_backward_split(dy, x, wp, relu_in=relu_in, need_dx=ctx.needs_input_grad[0], need_db=has_b,
                                                    scale_exp=ctx.scale_exp)
            return dx, dw, (db if has_b else None), None, None, dres, None
        dx, dw, db = ops.conv3x3_backward(dy, x, wp, relu_in=relu_in, need_dx=ctx.needs_input_grad[0], need_db=has_b)
        return dx, dw.permute(0, 3, 1, 2).contiguous(), (db if has_b else None), None, None, dres, None


def conv3x3(x_nhwc: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], relu_in: bool = False, stride: int = 1,
            residual: Optional[torch.Tensor] = None, relu_out: bool = False) -> torch.Tensor:
    return Conv3x3Fn.apply(x_nhwc, w, b, relu_in, stride, residual, relu_out)


class Upsample2xFn(torch.autograd.Function):
    """F.interpolate(scale_factor=2, bilinear, align_corners=True) on NHWC 16-bit; backward = its transpose (gather kernel)."""

    @staticmethod
    def forward(ctx, x):
        return ops.upsample2x_nhwc(x.contiguous())

    @staticmethod
    def backward(ctx, dy):
        return ops.upsample2x_backward_nhwc(dy.contiguous())


def upsample2x(x_nhwc: torch.Tensor) -> torch.Tensor:
    return Upsample2xFn.apply(x_nhwc)


class Upsample2xAddReluFn(torch.autograd.Function):
    """up2(x) + relu(s) in ONE kernel (the gs head's merge of the trunk with the 7x7 image features, dpt_gs_head.py:112-124; the inference
    path's fused launch): replaces a bilinear kernel, a ReLU and an add over the 256 x 256 x C tensor and their three backward passes.
    Backward: dx = up2^T(dy), ds = dy where s > 0."""

    @staticmethod
    def forward(ctx, x, s):
        s = s.contiguous()
        ctx.save_for_backward(s)
        return ops.upsample2x_nhwc(x.contiguous(), add=s, relu_add=True)

    @staticmethod
    def backward(ctx, dy):
        (s,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = ops.upsample2x_backward_nhwc(dy) if ctx.needs_input_grad[0] else None
        ds = ops.relu_mask(dy, s) if ctx.needs_input_grad[1] else None
        return dx, ds


def upsample2x_add_relu(x_nhwc: torch.Tensor, s_nhwc: torch.Tensor) -> torch.Tensor:
    return Upsample2xAddReluFn.apply(x_nhwc, s_nhwc)


class GeluFn(torch.autograd.Function):
    """Exact (erf) GELU on a 16-bit tensor; the pre-activation is kept for ops.gelu_backward."""

    @staticmethod
    def forward(ctx, z):
        z = z.contiguous()
        ctx.save_for_backward(z)
        return ops.gelu16(z)

    @staticmethod
    def backward(ctx, dy):
        (z,) = ctx.saved_tensors
        return ops.gelu_backward(dy.to(z.dtype).contiguous(), z)


def gelu(z: torch.Tensor) -> torch.Tensor:
    return GeluFn.apply(z)


class EncBlockFn(torch.autograd.Function):
    """One frame-encoder block (croco/blocks.py:114-130) as a single autograd node on the hand-differentiated pair of
    vicasplat_amd.train: x f32 [M,C] -> x_out f32 [M,C].  params = (norm1.w, norm1.b, qkv.w, qkv.b, proj.w, proj.b, norm2.w,
    norm2.b, fc1.w, fc1.b, fc2.w, fc2.b), the module's f32 parameters.
    ckpt = True is the reference's per-block activation checkpointing (backbone_vica.py:464-474): only the block INPUT is kept and
    the backward re-runs the block's forward kernels (run-to-run deterministic, so the recomputed tape is bit-identical) before
    differentiating it -- 9 saved tensors (~26 B per token-channel) shrink to one f32 copy of x."""

    @staticmethod
    def forward(ctx, x, pos, frames, tokens, heads, dt, ckpt, *params):
        from .train import EncBlockParams, enc_block_forward_train
        f = [t.detach().float().contiguous() for t in params]
        p = EncBlockParams(f[0], f[1], f[2].to(dt), f[3], f[4].to(dt), f[5], f[6], f[7], f[8].to(dt), f[9], f[10].to(dt), f[11])
        xin = x.detach().contiguous()
        x_out, tape = enc_block_forward_train(xin, p, pos, frames=frames, tokens=tokens, heads=heads)
        ctx.tape, ctx.p, ctx.live = (None if ckpt else tape), p, True
        ctx.recompute = (xin, pos, frames, tokens, heads) if ckpt else None
        return x_out

    @staticmethod
    def backward(ctx, dx_out):
        from .train import enc_block_backward, enc_block_forward_train
        if not ctx.live:
            raise RuntimeError("EncBlockFn.backward ran twice: the block's tape is released after the first backward "
                               "(backward(retain_graph=True) twice is not supported on the hand-differentiated encoder blocks)")
        tape = ctx.tape
        if tape is None:
            xin, pos, frames, tokens, heads = ctx.recompute
            _, tape = enc_block_forward_train(xin, ctx.p, pos, frames=frames, tokens=tokens, heads=heads)
        dx, g = enc_block_backward(dx_out.contiguous(), tape, ctx.p)
        ctx.tape = ctx.p = ctx.recompute = None
        ctx.live = False
        return (dx, None, None, None, None, None, None, g["ln1_w"], g["ln1_b"], g["qkv_w"], g["qkv_b"], g["proj_w"], g["proj_b"],
                g["ln2_w"], g["ln2_b"], g["fc1_w"], g["fc1_b"], g["fc2_w"], g["fc2_b"])


class GatedResidFn(torch.autograd.Function):
    """x32 [M,C] + (1 + gate[row // gate_rows]) * y16[yrow(row)] in one pass (ops.gated_resid); y's rows outside the map (the
    camera-token row in front of each frame's image tokens, grp_off) are returned as a second f32 output so that the camera
    stream can consume them.  Backward: dx = dout, dy rows / dgate from ops.gated_resid_backward."""

    @staticmethod
    def forward(ctx, x, y, gate, gate_rows, grp_in, grp_out, grp_off):
        y = y.contiguous()
        g = None if gate is None else gate.detach().float().contiguous()
        out = ops.gated_resid(x.detach().contiguous(), y, g, gate_rows, grp_in=grp_in, grp_out=grp_out, grp_off=grp_off)
        ctx.save_for_backward(y, g)
        ctx.meta = (gate_rows, grp_in, grp_out, grp_off, gate is not None and gate.dtype)
        extra = y.view(-1, grp_out, y.shape[1])[:, :grp_off].float() if grp_off > 0 else None
        return (out, extra) if grp_off > 0 else out

    @staticmethod
    def backward(ctx, dout, dextra=None):
        y, g = ctx.saved_tensors
        gate_rows, grp_in, grp_out, grp_off, gdt = ctx.meta
        if g is None and grp_off == 0 and grp_in == grp_out and y.dtype == torch.float32 and dout.dtype == torch.float32 and y.shape == dout.shape:
            # out = x + y with an f32 branch (the split class's frame-encoder blocks): both gradients ARE dout -- no copy pass (48 x 29 us per 8-scene
            # step).  (Measured and not kept beside it: the forward add in the GEMM's residual epilogue, autograd.linear(..., resid=x) -- identical
            # bits, 332.5 / 332.8 ms without against 331.5 / 333.4 with: the epilogue of a one-workgroup-per-CU kernel is exposed time, it costs what
            # the separate 30 us pass costs.)
            return dout, dout, None, None, None, None, None
        dy = torch.empty_like(y)
        if grp_off > 0:
            v = dy.view(-1, grp_out, y.shape[1])
            if dextra is None:
                v[:, :grp_off] = 0
            else:
                v[:, :grp_off] = dextra
        dgate = ops.gated_resid_backward(dout.contiguous().float(), y, g, gate_rows, dy, grp_in=grp_in, grp_out=grp_out, grp_off=grp_off)
        return dout, dy, (None if dgate is None else dgate.to(gdt)), None, None, None, None


def gated_resid(x, y, gate=None, gate_rows=0, grp_in=0, grp_out=0, grp_off=0):
    return GatedResidFn.apply(x, y, gate, gate_rows, grp_in, grp_out, grp_off)


class GaussianAdapterFn(torch.autograd.Function):
    """'exp' depth post-process + raw_gaussians concat + MyGaussianAdapter (heads/postprocess.py:46-56, vicasplat.py:256,
    common/gaussian_adapter.py:168-212) on the fused HIP kernel in both directions.  pts [N,H,W,>=3], gs [N,H,W,8+3*d_sh]: the
    heads' NHWC outputs -> (means [N,H,W,3], covariances [N,H,W,3,3], harmonics [N,H,W,3,d_sh], opacities [N,H,W], raw
    [N,H,W,11+3*d_sh], scales [N,H,W,3], rotations [N,H,W,4]), all f32.  `scales` / `rotations` (the reference's Gaussians carry them,
    gaussian_adapter.py:150-155; only exports and regularisers read them) are differentiable too: their cotangents, when some loss
    uses them, are folded into d_raw with the activation's derivative (a few element-wise torch ops on 7 channels; nothing runs when
    they are unused)."""

    @staticmethod
    def forward(ctx, pts, gs, sh_mask, scale_act, scale_min, scale_max, opacity_exponent):
        pts, gs = pts.contiguous(), gs.contiguous()
        mask = sh_mask.detach().float().contiguous()
        o = ops.gaussian_adapter(pts.permute(0, 3, 1, 2)[:, :3], gs.permute(0, 3, 1, 2), mask, scale_act=scale_act, scale_min=scale_min,
                                 scale_max=scale_max, opacity_exponent=opacity_exponent)
        ctx.save_for_backward(pts, gs, mask)
        ctx.meta = (scale_act, scale_min, scale_max, opacity_exponent)
        ctx.set_materialize_grads(False)      # unused outputs arrive as None (raw alone is 4.3 GB of zeros at 24 scenes otherwise)
        return o["means"], o["covariances"], o["harmonics"], o["opacities"][..., 0], o["raw"], o["scales"], o["rotations"]

    @staticmethod
    def backward(ctx, d_means, d_cov, d_harm, d_op, d_raw, d_scales=None, d_rot=None):
        pts, gs, mask = ctx.saved_tensors
        scale_act, scale_min, scale_max, opacity_exponent = ctx.meta
        z = lambda g, ref_shape: torch.zeros(ref_shape, dtype=torch.float32, device=pts.device) if g is None else g
        N, H, W = gs.shape[:3]
        d_sh = (gs.shape[-1] - 8) // 3
        if d_scales is not None or d_rot is not None:     # raw[..., 4:7] / raw[..., 7:11] are the pre-activation scales / quaternion
            import torch.nn.functional as F
            with torch.enable_grad():
                pre = gs[..., 1:8].detach().float().requires_grad_(True)
                s = pre[..., :3]
                if scale_act == "bounded":
                    s = scale_min + (scale_max - scale_min) * s.sigmoid()
                elif scale_act == "exp":
                    s = s.exp().clamp_max(0.3)
                else:
                    s = (0.001 * F.softplus(s)).clamp_max(0.3)
                r = F.normalize(pre[..., 3:], dim=-1)
                (g7,) = torch.autograd.grad([s, r], [pre], [z(d_scales, s.shape), z(d_rot, r.shape)])
            d_raw = z(d_raw, (N, H, W, 11 + 3 * d_sh)).clone()
            d_raw[..., 4:11] += g7
        d_pts, d_gs = ops.gaussian_adapter_backward(pts, gs, mask, z(d_means, (N, H, W, 3)), z(d_cov, (N, H, W, 3, 3)),
                                                    z(d_harm, (N, H, W, 3, d_sh)), z(d_op, (N, H, W)), d_raw, scale_act=scale_act,
                                                    scale_min=scale_min, scale_max=scale_max, opacity_exponent=opacity_exponent)
        return d_pts, d_gs, None, None, None, None, None


def gaussian_adapter(pts, gs, sh_mask, *, scale_act="softplus", scale_min=0.0, scale_max=0.0, opacity_exponent=1.0):
    return GaussianAdapterFn.apply(pts, gs, sh_mask, scale_act, scale_min, scale_max, opacity_exponent)
