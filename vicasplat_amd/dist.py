"""Multi-GPU host logic: one process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI on ROCm).

Inference / throughput: scenes are independent -> the scene batch is SHARDED over ranks and nothing
crosses devices on the data path; results are gathered on the host only if the caller asks.
Training: exactly one exchange per step -- a bucketed sum all-reduce of the gradients (the reference relies on
Lightning DDP, src/main.py:110-115).  xGMI is point-to-point (7 links/GPU), so buckets are large (default 64 MiB:
few, big ring collectives) and parameters that received no gradient (scratch.refinenet4.resConfUnit1 in both DPT
heads) are zero-filled so every rank reduces identical buckets without find_unused_parameters.
"""
from __future__ import annotations

import weakref
from typing import Iterable, List, Sequence

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced shard of `n_items` for `rank` (first n%world ranks get one more)."""
    q, r = divmod(n_items, world)
    start = rank * q + min(rank, r)
    return range(start, start + q + (1 if rank < r else 0))


def shard_batch(batch: dict, rank: int, world: int) -> dict:
    """Slice every tensor of a (nested) batch dict along dim 0 to this rank's scenes."""
    def cut(v):
        if isinstance(v, dict):
            return {k: cut(x) for k, x in v.items()}
        if torch.is_tensor(v):
            r = shard_range(v.shape[0], rank, world)
            return v[r.start:r.stop]
        return v
    return cut(batch)


def gather_scenes(t: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """All-gather per-rank scene results (possibly uneven) back to the full batch order; used off the hot path."""
    world = dist.get_world_size(group)
    sizes = [len(shard_range(n_total, r, world)) for r in range(world)]
    mx = max(sizes)
    pad = torch.zeros((mx, *t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad, group=group)
    return torch.cat([o[:s] for o, s in zip(outs, sizes)], 0)


def bucketed_allreduce_grads(params: Iterable[torch.nn.Parameter], bucket_bytes: int = 64 << 20, average: bool = True,
                             group=None) -> int:
    """Sum (or average) gradients across ranks in flat buckets.  Returns the number of collectives issued."""
    world = dist.get_world_size(group)
    ps: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
    for p in ps:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    n_coll = 0
    bucket: List[torch.nn.Parameter] = []
    size = 0

    def flush():
        nonlocal n_coll, bucket, size
        if not bucket:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat /= world
        off = 0
        for p in bucket:
            n = p.numel()
            p.grad.copy_(flat[off:off + n].view_as(p))
            off += n
        n_coll += 1
        bucket, size = [], 0

    for p in ps:
        b = p.numel() * p.element_size()
        if size and size + b > bucket_bytes:
            flush()
        bucket.append(p)
        size += b
    flush()
    return n_coll


# parameter -> the GradReducer whose buckets hold its .grad.  Kept OUTSIDE the Parameter: Parameter.__reduce_ex__ pickles __dict__, and a
# weakref there made torch.save(model) / pickling fail while a reducer was attached.  Weak values: a dropped reducer leaves no entry.
_REDUCER_OF: "weakref.WeakValueDictionary[int, GradReducer]" = weakref.WeakValueDictionary()


def reducer_of(p):
    """The live GradReducer that owns p.grad, or None."""
    r = _REDUCER_OF.get(id(p))
    return r if (r is not None and r.owns(p)) else None


class GradReducer:
    """Gradient exchange overlapped with the backward pass (what Lightning's DDP gives the reference, src/main.py:110-115).

    The trainable parameters are laid out, in REVERSE registration order (the order in which backward produces their gradients),
    in flat f32 buckets; every `p.grad` is a VIEW into its bucket, so autograd accumulates in place and nothing is concatenated
    or copied back.  A post-accumulate-grad hook per parameter counts arrivals; when a bucket is complete its all-reduce is
    issued asynchronously (RCCL over xGMI runs it on its own stream while the backward kernels continue).  `finish()` launches
    the buckets that never completed (parameters without a gradient this step -- scratch.refinenet4.resConfUnit1 of both DPT
    heads, SURVEY 2.2 -- contribute the zeros `zero_grad()` left there, so every rank reduces identical buckets), waits, and leaves
    `.grad = None` on those parameters (DDP's behaviour: AdamW then neither updates nor weight-decays them).  Buckets are launched
    strictly in index order, whatever order the hooks fire in.  Because of that a parameter that NEVER gets a gradient would hold back
    its bucket -- and every later one -- until finish(): the model's graph is static (DDP's `static_graph`), so the set of such
    parameters is learned from the first step (or passed as `unused_params=`) and counted as arrived from zero_grad() on; the overlap
    is then lost on the first step only.  If a parameter of that set does receive a gradient later, it is accepted as long as its
    bucket has not been launched, and is an error (a gradient that would miss the exchange) otherwise.

    xGMI is point-to-point (7 links x ~153 GB/s per GPU): ring collectives are per-link bound, so buckets are large (64 MiB
    default -> ~36 collectives for the 2.31 GB gradient) rather than DDP's 25 MB.  `comm_dtype=torch.bfloat16` halves the bytes
    on the links (1.16 GB): the bucket is cast into a staging buffer, reduced, and cast back.

    The Module API's internal gradient scale (autograd.BoundaryGradScale: cotangents x S through the encoder, gradients / S when backward
    ends) meets this class through `defer_unscale`: the end-of-backward callback runs while the bucket all-reduces are in flight, so it
    does not touch the buckets; it hands 1/S over and `finish()` applies it after the wait, fused with the averaging.  The sum of scaled
    gradients times 1/S is exact for a power of two (barring overflow: `last_overflow`, a 0-d device flag set by `finish()`).
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 64 << 20, average: bool = True, group=None,
                 comm_dtype: torch.dtype | None = None, unused_params: Iterable[torch.nn.Parameter] | None = None):
        self.group, self.average, self.comm_dtype = group, average, comm_dtype
        self._unused_ids = {id(p) for p in (unused_params or ())}     # parameters known to receive no gradient (static graph)
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        ps = [p for p in params if p.requires_grad]
        for p in ps:
            if p.dtype != torch.float32:
                raise TypeError(f"GradReducer keeps f32 master gradients: parameter of dtype {p.dtype} (cast the module with .float())")
        self.params = ps[::-1]
        self.buckets: List[dict] = []
        cur, size = [], 0
        for p in self.params:
            b = p.numel() * 4
            if cur and size + b > bucket_bytes:
                self._close(cur)
                cur, size = [], 0
            cur.append(p)
            size += b
        if cur:
            self._close(cur)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._pending: List[tuple] = []
        self._unused_learned = unused_params is not None     # the static set is learned from the first COMPLETE step only
        self._unscale, self._unscale_ids, self._unscale_any = 1.0, None, False
        self.last_overflow = None
        self.zero_grad()

    def owns(self, p) -> bool:
        """True while `p.grad` is (supposed to be) a view of one of this reducer's buckets, i.e. until remove()."""
        return bool(self._hooks) and getattr(p, "_vs_bucket", None) is not None

    def defer_unscale(self, inv_scale: float, params=None):
        """Called by autograd.BoundaryGradScale when backward ends: the gradients of `params` (None: of every parameter of this reducer)
        sit in the buckets in units of 1 / inv_scale; finish() multiplies exactly those by it after their collective has completed.
        Parameters of the reducer that the scaler does not cover (outside the encoder module, or whose gradient did not flow through the
        boundary node) keep plain gradients."""
        inv_scale = float(inv_scale)
        if self._unscale != 1.0 and self._unscale != inv_scale:
            raise RuntimeError("GradReducer: two backward passes with different boundary gradient scales before one finish()")
        self._unscale = inv_scale
        if params is None:
            self._unscale_ids = None
        elif self._unscale_ids is not None or not self._unscale_any:
            self._unscale_ids = (self._unscale_ids or set()) | {id(p) for p in params}
        self._unscale_any = True

    def pending_unscale(self, p) -> bool:
        """True while p's gradient in the bucket is still in units of the boundary scale (between the end of a backward and finish()): a
        second backward before finish() accumulates onto it as it is -- the scaler must not lift it by S again."""
        return self._unscale_any and (self._unscale_ids is None or id(p) in self._unscale_ids)

    def reset_unused(self):
        """Forget the learned set of gradient-less parameters (call when the graph changes on purpose: e.g. the distillation phase, which
        skips the Gaussian-parameter head, followed by the full objective).  The next complete step learns it again."""
        self._unused_ids = set()
        self._unused_learned = False

    def _close(self, plist):
        dev = plist[0].device
        n = sum(p.numel() for p in plist)
        flat = torch.zeros(n, dtype=torch.float32, device=dev)
        views, off = [], 0
        for p in plist:
            views.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.buckets.append(dict(params=plist, flat=flat, views=views, ready=0, launched=False, got=[False] * len(plist),
                                 stage=torch.empty(n, dtype=self.comm_dtype, device=dev) if self.comm_dtype else None))
        for i, p in enumerate(plist):
            p._vs_bucket, p._vs_slot = len(self.buckets) - 1, i      # plain ints: a Parameter's __dict__ is pickled with it
            _REDUCER_OF[id(p)] = self                               # (a weakref stored ON the parameter broke torch.save(model))

    def zero_grad(self):
        """Zero the flat buckets and (re)attach every p.grad as a view of its bucket."""
        for b in self.buckets:
            b["flat"].zero_()
            b["launched"], b["got"] = False, [False] * len(b["params"])
            b["ready"] = sum(1 for p in b["params"] if id(p) in self._unused_ids)     # known-unused parameters count as arrived
            for p, v in zip(b["params"], b["views"]):
                p.grad = v
        self._pending = []
        self._unscale, self._unscale_ids, self._unscale_any = 1.0, None, False
        self._next = 0            # buckets are launched strictly in index order: every rank issues the same collective sequence

    def _launch(self, b):
        if b["launched"]:
            return
        b["launched"] = True
        if self.world == 1:
            return
        buf = b["flat"]
        if b["stage"] is not None:
            b["stage"].copy_(buf)
            buf = b["stage"]
        self._pending.append((b, dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)))

    def _on_grad(self, p):
        b = self.buckets[p._vs_bucket]
        v = b["views"][p._vs_slot]
        if p.grad is not v:      # someone replaced .grad (zero_grad(set_to_none=True)): fold it back
            v.copy_(p.grad)
            p.grad = v
        if not b["got"][p._vs_slot]:
            b["got"][p._vs_slot] = True
            if id(p) not in self._unused_ids:
                b["ready"] += 1
            elif b["launched"]:
                raise RuntimeError("GradReducer: a parameter recorded as unused (no gradient in the previous step) received a gradient after "
                                   "its bucket was exchanged -- the graph is not static; call reset_unused() at a phase change (or rebuild the reducer)")
        # launch in INDEX order only (a bucket that completes before an earlier one waits for it): the order in which hooks fire may
        # differ between ranks or steps, the order of the collectives must not
        while self._next < len(self.buckets) and self.buckets[self._next]["ready"] == len(self.buckets[self._next]["params"]):
            self._launch(self.buckets[self._next])
            self._next += 1

    def finish(self) -> int:
        """Issue what is left, wait for every collective, finish the averaging.  Returns the number of collectives of the step."""
        for b in self.buckets[self._next:]:
            self._launch(b)
        self._next = len(self.buckets)
        n = len(self._pending)
        waited = set()
        for b, work in self._pending:
            work.wait()
            if b["stage"] is not None:
                b["flat"].copy_(b["stage"])
            waited.add(id(b))
        # averaging and the deferred 1/S of the Module API's internal gradient scale, one pass per bucket, after its collective
        found = None
        for b in self.buckets:
            avg = 1.0 / (self.world if (self.average and id(b) in waited) else 1)
            if self._unscale_any and self._unscale != 1.0:
                from .autograd import unscale_and_check_
                ids = self._unscale_ids
                cov = [ids is None or id(p) in ids or not got for p, got in zip(b["params"], b["got"])]
                if all(cov):      # the usual case: every gradient of the bucket came through the boundary node (or is absent: zeros)
                    fl = unscale_and_check_([b["flat"]], self._unscale * avg)
                else:             # a reducer over encoder + other parameters: only the scaler's own are in units of S
                    fl = unscale_and_check_([v for v, c in zip(b["views"], cov) if c], self._unscale * avg) if any(cov) else None
                    rest = [v for v, c in zip(b["views"], cov) if not c]
                    if avg != 1.0 and rest:
                        torch._foreach_mul_(rest, avg)
                if fl is not None:
                    found = fl if found is None else torch.maximum(found, fl)
            elif avg != 1.0:
                b["flat"] *= avg
        if found is not None:
            self.last_overflow = found
        self._unscale, self._unscale_ids, self._unscale_any = 1.0, None, False
        self._pending = []
        # parameters that received no gradient this step keep .grad = None, as under DDP: the optimizer skips them (a zero gradient
        # would still be weight-decayed by AdamW every step).  The model's graph does not depend on the rank or the data, so the set is
        # the same on every rank; the zeros they contributed to the buckets kept the collectives identical.  zero_grad() re-attaches.
        unused, arrivals = set(), 0
        for b in self.buckets:
            for p, got in zip(b["params"], b["got"]):
                if not got:
                    p.grad = None
                    unused.add(id(p))
                else:
                    arrivals += 1
        # The static set is LEARNED from the first step in which a backward pass actually ran, and only shrinks afterwards (a parameter
        # that does get a gradient leaves it).  A finish() without arrivals (skipped / failed step) teaches nothing, and a step with
        # FEWER gradients than usual (a partial loss, the distillation phase) does not grow the set: growing it would let the next full
        # step launch a bucket before one of its gradients has arrived.  `reset_unused()` relearns on purpose.
        if arrivals:
            if not self._unused_learned:
                self._unused_ids, self._unused_learned = unused, True
            else:
                self._unused_ids &= unused
        return n

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self.params:
            for a in ("_vs_bucket", "_vs_slot"):
                if hasattr(p, a):
                    delattr(p, a)
            if _REDUCER_OF.get(id(p)) is self:
                del _REDUCER_OF[id(p)]
