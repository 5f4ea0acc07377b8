"""N>1 host logic on CPU: world_size 2, gloo backend (rendezvous on 127.0.0.1)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vicasplat_amd import dist as vd


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # 1. scene sharding: every scene exactly once, order preserved after the gather
        n = 5
        batch = dict(context=dict(image=torch.arange(n, dtype=torch.float32)[:, None].repeat(1, 3), idx=torch.arange(n)), tag="x")
        mine = vd.shard_batch(batch, rank, world)
        assert mine["tag"] == "x" and mine["context"]["image"].shape[0] == len(vd.shard_range(n, rank, world))
        res = mine["context"]["image"] * 2  # "process" the scenes
        full = vd.gather_scenes(res, n)
        assert torch.equal(full, batch["context"]["image"] * 2)
        # 2. replicas are bit-identical: same shard processed on two ranks gives the same bytes
        torch.manual_seed(0)
        w = torch.randn(16, 16)
        y = (torch.arange(16.0) @ w).sum()
        ys = [torch.zeros(()) for _ in range(world)]
        dist.all_gather(ys, y)
        assert all(torch.equal(ys[0], t) for t in ys)
        # 3. bucketed gradient all-reduce == sum over ranks / world, unused parameters zero-filled
        torch.manual_seed(1)
        ps = [torch.nn.Parameter(torch.randn(300)), torch.nn.Parameter(torch.randn(7, 5)), torch.nn.Parameter(torch.randn(1000)),
              torch.nn.Parameter(torch.randn(3))]
        for i, p in enumerate(ps[:3]):
            p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
        ncoll = vd.bucketed_allreduce_grads(ps, bucket_bytes=2048)
        for i, p in enumerate(ps[:3]):
            assert torch.allclose(p.grad, torch.full_like(p, (1 + 2) / 2 * (i + 1)))
        assert ps[3].grad is not None and float(ps[3].grad.abs().sum()) == 0.0
        assert ncoll >= 2
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(out) == [(0, "ok"), (1, "ok")], out


def test_shard_range_properties():
    for n in range(0, 20):
        for w in (1, 2, 3, 8):
            rs = [vd.shard_range(n, r, w) for r in range(w)]
            assert sum(len(r) for r in rs) == n
            assert [i for r in rs for i in r] == list(range(n))
            assert max(len(r) for r in rs) - min(len(r) for r in rs) <= 1


# ---- overlapped gradient exchange (GradReducer): 2 ranks x B/2 scenes == 1 rank x B scenes ----
class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.net = torch.nn.Sequential(torch.nn.Linear(12, 64), torch.nn.GELU(), torch.nn.Linear(64, 64), torch.nn.GELU(), torch.nn.Linear(64, 5))
        self.unused = torch.nn.Linear(4, 4)   # never reached by the loss: its bucket is zero-filled on every rank

    def forward(self, x):
        return self.net(x)


def _toy_model():
    torch.manual_seed(3)
    return _Toy()


def _toy_batch(n=8):
    g = torch.Generator().manual_seed(4)
    return torch.randn(n, 12, generator=g), torch.randn(n, 5, generator=g)


def _reducer_worker(rank, world, port, q, comm_bf16):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = _toy_model()
        red = vd.GradReducer(m.parameters(), bucket_bytes=4096, comm_dtype=torch.bfloat16 if comm_bf16 else None)
        assert len(red.buckets) >= 3
        x, y = _toy_batch()
        r = vd.shard_range(x.shape[0], rank, world)
        for step in range(2):                   # two steps: zero_grad() must re-arm the buckets
            red.zero_grad()
            loss = ((m(x[r.start:r.stop]) - y[r.start:r.stop]) ** 2).mean()
            loss.backward()
            # `unused` is registered last = laid out FIRST (bucket 0): on the first step it holds every bucket back until finish(); from
            # the second step on it is known (static graph) and every collective is already in flight when backward returns
            assert red._next == (0 if step == 0 else len(red.buckets)), (step, red._next)
            assert len(red._pending) == (0 if step == 0 else len(red.buckets))
            ncoll = red.finish()
            assert ncoll == len(red.buckets)
            assert m.unused.weight.grad is None and m.net[0].weight.grad is not None
        # numpy arrays travel by value; tensors would travel as shared-memory handles that die with this process (the parent may read late)
        q.put((rank, {n: (None if p.grad is None else p.grad.detach().numpy().copy()) for n, p in m.named_parameters()}))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def _run_reducer(comm_bf16):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reducer_worker, args=(r, 2, port, q, comm_bf16)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert all(isinstance(v, dict) for v in out.values()), out
    out = {r: {n: (None if a is None else torch.from_numpy(a)) for n, a in v.items()} for r, v in out.items()}
    m = _toy_model()
    x, y = _toy_batch()
    ((m(x) - y) ** 2).mean().backward()
    return out, {n: p.grad for n, p in m.named_parameters()}


def test_gradient_equivalence_two_ranks_half_batch_each():
    out, ref = _run_reducer(False)
    for rank in (0, 1):
        for n, g in out[rank].items():
            if ref[n] is None:
                assert g is None, n       # no gradient this step: .grad stays None, as under DDP (AdamW then skips the parameter)
            else:
                assert torch.allclose(g, ref[n], rtol=1e-5, atol=1e-7), (rank, n)
    for n in out[0]:
        assert (out[0][n] is None and out[1][n] is None) or torch.equal(out[0][n], out[1][n])          # replicas stay bit-identical


def test_gradient_exchange_in_bf16_is_close():
    out, ref = _run_reducer(True)
    for n, g in out[0].items():
        if ref[n] is not None:
            assert torch.allclose(g, ref[n], rtol=2e-2, atol=1e-4), n
        else:
            assert g is None


# ---- callers.training_step under the GradReducer at world size 2: the step's control flow -- loss scaling, the
# gradient exchange launched from hooks, unscale + clip, the skipped-step decision from the REDUCED norm, AdamW -- with a toy encoder /
# decoder on CPU (forward_fn injection; the HIP forward needs a GPU).  The toy registers its LAST-used layer FIRST, so gradients arrive
# in an order that is not the reverse registration order the buckets are laid out in. ----
class _ToyEncoder(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.head = torch.nn.Linear(16, 4 * 14)                   # registered first, used last: its gradient arrives first
        self.body = torch.nn.Sequential(torch.nn.Linear(12, 32), torch.nn.GELU(), torch.nn.Linear(32, 16))
        self.pose = torch.nn.Linear(16, 8)
        self.unused = torch.nn.Linear(3, 3)


def _toy_forward(enc, image, intrinsics, compute_dtype, global_step=0):
    B, V = image.shape[:2]
    f = enc.body(image.flatten(2).mean(1)[:, :12])                # [B, 16]
    o = enc.head(f).view(B, 1, 1, 4, 14)
    cov = torch.eye(3).expand(B, 1, 1, 4, 3, 3) * (0.1 + o[..., 3:4, None].abs())
    return dict(gaussians=dict(means=o[..., :3], covariances=cov, harmonics=o[..., 4:13].unflatten(-1, (3, 3)), opacities=torch.sigmoid(o[..., 13])),
                pred_extrins=enc.pose(f)[:, None].expand(B, V - 1, 8))


class _ToyDecoder:
    def forward(self, gs, extr, intr, near, far, shape, **kw):
        B, Vt = extr.shape[:2]
        col = torch.sigmoid(gs.means.mean(1) + gs.harmonics.mean((1, 3)) * gs.opacities.mean(1, keepdim=True) + gs.covariances.mean((1, 2)))
        from types import SimpleNamespace
        return SimpleNamespace(color=col[:, None, :, None, None].expand(B, Vt, 3, *shape))


def _toy_train_batch(n=4, V=3, Vt=2):
    g = torch.Generator().manual_seed(6)
    E = torch.eye(4).repeat(n, V, 1, 1)
    return dict(context=dict(image=torch.randn(n, V, 3, 4, 4, generator=g), intrinsics=torch.eye(3).repeat(n, V, 1, 1), extrinsics=E),
                target=dict(image=torch.rand(n, Vt, 3, 4, 4, generator=g), extrinsics=torch.eye(4).repeat(n, Vt, 1, 1),
                            intrinsics=torch.eye(3).repeat(n, Vt, 1, 1), near=torch.ones(n, Vt), far=torch.ones(n, Vt)))


def _train_worker(rank, world, port, q, comm_bf16, poison_step):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vicasplat_amd import callers
        torch.manual_seed(5)
        enc = _ToyEncoder()
        opt = torch.optim.AdamW(enc.parameters(), lr=1e-2, weight_decay=0.05)
        red = vd.GradReducer(enc.parameters(), bucket_bytes=1024, comm_dtype=torch.bfloat16 if comm_bf16 else None) if world > 1 else None
        assert red is None or len(red.buckets) >= 3
        scaler = callers.LossScaler(1024.0)
        full = _toy_train_batch()
        sh = vd.shard_range(4, rank, world)
        batch = {k: {kk: vv[sh.start:sh.stop] for kk, vv in v.items()} for k, v in full.items()}
        skipped, scales = [], []
        for step in range(5):
            poison = (lambda render, b, out: out["pred_extrins"].sum() * float("inf")) if (step == poison_step and rank == world - 1) else None
            r = callers.training_step(enc, _ToyDecoder(), batch, opt, compute_dtype=torch.float16, loss_scale=scaler, camera_weight=0.1,
                                      extra_losses=(poison,) if poison else (), reducer=red, forward_fn=_toy_forward)
            skipped.append(bool(r["skipped"])); scales.append(scaler.scale)
        assert enc.unused.weight.grad is None                      # never touched: no zero gradient for AdamW's weight decay to act on
        q.put((rank, dict(params={n: p.detach().numpy().copy() for n, p in enc.named_parameters()}, skipped=skipped, scales=scales)))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, repr(e) + traceback.format_exc()))
    finally:
        if world > 1:
            dist.destroy_process_group()


def _run_train(world, comm_bf16=False, poison_step=-1):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q, comm_bf16, poison_step)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert all(isinstance(v, dict) for v in out.values()), out
    return out


def test_training_step_with_reducer_two_ranks_matches_one_rank():
    """2 ranks x 2 scenes with the overlapped exchange == 1 rank x 4 scenes, parameter by parameter after 5 AdamW steps; the replicas
    stay bit-identical although the gradients arrive in a different order than the buckets are indexed."""
    two, one = _run_train(2), _run_train(1)
    for n, p in one[0]["params"].items():
        assert np.allclose(two[0]["params"][n], p, rtol=2e-4, atol=2e-6), n
        assert np.array_equal(two[0]["params"][n], two[1]["params"][n]), n
    assert two[0]["skipped"] == [False] * 5 and two[0]["scales"] == two[1]["scales"]


def test_overflow_on_one_rank_skips_the_step_on_every_rank():
    """Rank 1 produces an infinite loss at step 2: the reduced gradient norm is non-finite on BOTH ranks, both skip the step, both halve
    the loss scale, and the replicas stay bit-identical (the decision comes from the reduced norm, never from a local one)."""
    out = _run_train(2, poison_step=2)
    assert out[0]["skipped"] == out[1]["skipped"] == [False, False, True, False, False]
    assert out[0]["scales"] == out[1]["scales"] and out[0]["scales"][2] < out[0]["scales"][1]
    for n in out[0]["params"]:
        assert np.array_equal(out[0]["params"][n], out[1]["params"][n]), n
        assert np.isfinite(out[0]["params"][n]).all(), n


def test_bf16_wire_with_loss_scale_is_close_to_the_f32_wire():
    """comm_dtype=bfloat16 with a loss scale of 1024: the scaled gradients travel as bf16 (8 exponent bits: no range issue), the update
    stays within bf16 rounding of the f32-wire run."""
    a, b = _run_train(2), _run_train(2, comm_bf16=True)
    for n, p in a[0]["params"].items():
        # (AdamW normalises the gradient: an element whose tiny gradient changes sign under bf16 rounding moves by lr per step -- the
        #  worst element is bounded by steps x lr, the bulk by bf16's 2^-8)
        d = np.abs(b[0]["params"][n] - p)
        assert float(d.max()) <= 5 * 1e-2 * 0.5 and float(d.mean()) <= 1e-3, (n, float(d.max()), float(d.mean()))
        assert np.array_equal(b[0]["params"][n], b[1]["params"][n]), n


def test_bench_dry_run_dist_two_ranks():
    """bench.py's N > 1 control flow -- process-group init from the torchrun environment, barriers, the MAX-reduce of
    the elapsed time, the training leg's watchdog, one GradReducer exchange, rank 0 printing ONE JSON line -- runs on CPU with gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run-dist"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["steps"] == 3 and d["scaling"] == "weak" and d["value"] > 0
    assert d["train"]["gradient_exchange"].startswith("GradReducer") and d["train"]["replicas_identical"] is True
    assert d["distributed"]["ranks_seen"] == 2


def _bench_env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    return env


def test_bench_gpus_2_launches_two_ranks_by_itself():
    """`python bench.py --gpus 2` with NO launcher around it starts the two ranks itself (a plain `python -m src.main`
    does under Lightning, src/main.py:104-116) and the line says n_gpus 2 with the collective library's own rank count beside it."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run-dist"],
                       capture_output=True, text=True, timeout=300, cwd=root, env=_bench_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["distributed"]["ranks_seen"] == 2 and d["train"]["replicas_identical"] is True
    # --gpus 1: unchanged, no launcher, one rank
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "0", "--dry-run-dist"],
                       capture_output=True, text=True, timeout=300, cwd=root, env=_bench_env())
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["distributed"]["ranks_seen"] == 1


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """--gpus is asserted against the launcher's WORLD_SIZE: an 8-rank line can never come out of a 1-rank run (or the reverse)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = _bench_env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run-dist"], capture_output=True, text=True,
                       timeout=300, cwd=root, env=env)
    assert r.returncode != 0 and "--gpus 2" in (r.stderr + r.stdout) and not [l for l in r.stdout.splitlines() if l.startswith("{")]


# ---- the Module API's internal gradient scale under torch's DistributedDataParallel (what Lightning wraps the reference's ModelWrapper in,
# src/main.py:110-115): cotangents x S on the way in, parameter gradients / S from an autograd-engine callback that re-queues itself once so
# that it runs AFTER DDP's own end-of-backward callback (the bucket all-reduce and the copy back into .grad work on scaled values) ----
class _ScaledToy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.net = torch.nn.Sequential(torch.nn.Linear(12, 32), torch.nn.GELU(), torch.nn.Linear(32, 5))
        self.scaler = None

    def forward(self, x):
        from vicasplat_amd import autograd as A
        y = self.net(x)
        if self.scaler is None:
            self.scaler = A.BoundaryGradScale(list(self.parameters()), 8192.0)
        return self.scaler.outputs(y)[0]


def _ddp_scale_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(3)
        m = _ScaledToy()
        ddp = torch.nn.parallel.DistributedDataParallel(m)          # gradient_as_bucket_view=False: DDP copies the reduced buckets back at the end
        x, y = _toy_batch()
        r = vd.shard_range(x.shape[0], rank, world)
        for step in range(2):
            ddp.zero_grad(set_to_none=(step == 0))
            loss = ((ddp(x[r.start:r.stop]) - y[r.start:r.stop]) ** 2).mean()
            loss.backward()
        q.put((rank, {n: p.grad.detach().numpy().copy() for n, p in m.named_parameters()}))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_boundary_grad_scale_under_ddp_matches_the_unscaled_full_batch_gradient():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_scale_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert all(isinstance(v, dict) for v in out.values()), out
    torch.manual_seed(3)
    ref = _ScaledToy()
    x, y = _toy_batch()
    ((ref.net(x) - y) ** 2).mean().backward()                        # 1 rank x full batch, no scale, no DDP
    for n, p in ref.named_parameters():
        for rank in (0, 1):
            g = torch.from_numpy(out[rank][n])
            assert torch.allclose(g, p.grad, rtol=1e-5, atol=1e-8), (rank, n, float((g - p.grad).abs().max()))
        assert np.array_equal(out[0][n], out[1][n]), n               # replicas bit-identical


# ---- the same internal scale under THIS package's GradReducer: p.grad are views of flat buckets whose all-reduces are in
# flight when backward ends, so the end-of-backward callback must not touch them; the 1/S is applied by finish() after the wait ----
def _reducer_scale_worker(rank, world, port, q, comm_bf16):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(3)
        m = _ScaledToy()
        # small buckets: the first ones are launched (and, staged, copied out) long before backward ends -- the case the advisor described
        red = vd.GradReducer(m.parameters(), bucket_bytes=256, comm_dtype=torch.bfloat16 if comm_bf16 else None)
        x, y = _toy_batch()
        r = vd.shard_range(x.shape[0], rank, world)
        launched_early = 0
        for step in range(2):
            red.zero_grad()
            loss = ((m(x[r.start:r.stop]) - y[r.start:r.stop]) ** 2).mean()
            loss.backward()
            launched_early = max(launched_early, len(red._pending))
            red.finish()
        ovf = float(red.last_overflow)
        q.put((rank, dict(grads={n: p.grad.detach().numpy().copy() for n, p in m.named_parameters()}, early=launched_early, ovf=ovf)))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _run_reducer_scale(comm_bf16):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reducer_scale_worker, args=(r, 2, port, q, comm_bf16)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert all(isinstance(v, dict) for v in out.values()), out
    return out


def test_boundary_grad_scale_under_the_grad_reducer_f32_wire():
    out = _run_reducer_scale(False)
    torch.manual_seed(3)
    ref = _ScaledToy()
    x, y = _toy_batch()
    ((ref.net(x) - y) ** 2).mean().backward()
    assert out[0]["early"] >= 2 and out[0]["ovf"] == 0.0            # buckets really were in flight when backward ended
    for n, p in ref.named_parameters():
        for rank in (0, 1):
            g = torch.from_numpy(out[rank]["grads"][n])
            assert torch.allclose(g, p.grad, rtol=1e-5, atol=1e-8), (rank, n, float((g - p.grad).abs().max()))
        assert np.array_equal(out[0]["grads"][n], out[1]["grads"][n]), n


def test_boundary_grad_scale_under_the_grad_reducer_bf16_wire():
    """comm_dtype staging: finish() copies the reduced stage back over the bucket AFTER backward ended -- a bucket unscaled by the callback
    would come back S times too large (early buckets) next to correct late ones.  Every gradient must be the plain one to bf16 rounding."""
    out = _run_reducer_scale(True)
    torch.manual_seed(3)
    ref = _ScaledToy()
    x, y = _toy_batch()
    ((ref.net(x) - y) ** 2).mean().backward()
    for n, p in ref.named_parameters():
        g = torch.from_numpy(out[0]["grads"][n])
        assert float((g - p.grad).abs().max()) <= 2.0 ** -7 * float(p.grad.abs().max()) + 1e-8, (n, float(g.abs().max()), float(p.grad.abs().max()))
        assert np.array_equal(out[0]["grads"][n], out[1]["grads"][n]), n


def test_grad_reducer_unused_set_is_learned_once_and_only_shrinks():
    """a finish() without a backward (skipped step) or a step with fewer gradients than usual must not grow the learned set
    of gradient-less parameters -- the next full step would launch buckets before their gradients arrive and raise."""
    torch.manual_seed(0)
    a, b, c = torch.nn.Linear(4, 4), torch.nn.Linear(4, 4), torch.nn.Linear(4, 4)     # c never used
    red = vd.GradReducer(list(a.parameters()) + list(b.parameters()) + list(c.parameters()), bucket_bytes=64)
    x = torch.randn(2, 4)
    red.zero_grad(); red.finish()                                    # a skipped step: nothing learned
    assert not red._unused_learned and not red._unused_ids
    red.zero_grad(); b(a(x)).sum().backward(); red.finish()          # first complete step: c's parameters are the static unused set
    assert red._unused_learned and red._unused_ids == {id(p) for p in c.parameters()}
    red.zero_grad(); a(x).sum().backward(); red.finish()             # partial step (b unused this time): the set does not grow
    assert red._unused_ids == {id(p) for p in c.parameters()}
    red.zero_grad(); red.finish()                                    # another skipped step
    assert red._unused_ids == {id(p) for p in c.parameters()}
    red.zero_grad(); b(a(x)).sum().backward(); red.finish()          # the full step again: no "graph is not static" error
    assert all(p.grad is not None for p in list(a.parameters()) + list(b.parameters())) and all(p.grad is None for p in c.parameters())
    red.reset_unused()
    assert not red._unused_ids and not red._unused_learned
    red.remove()
    assert not hasattr(next(a.parameters()), "_vs_bucket")


# ---- (i) a reducer attached to a module must not break pickling; (ii) the deferred 1/S of the boundary scale applies to the
# scaler's parameters only; (iii) two backward passes before one finish() accumulate consistently ----
def test_module_with_a_reducer_attached_pickles():
    import io
    import pickle
    m = torch.nn.Linear(4, 3)
    red = vd.GradReducer(m.parameters(), bucket_bytes=64)
    m(torch.randn(2, 4)).sum().backward()
    red.finish()
    assert vd.reducer_of(m.weight) is red
    pickle.dumps(m.weight)                          # was: TypeError: cannot pickle 'weakref.ReferenceType'
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m2 = torch.load(buf, weights_only=False)
    assert torch.equal(m2.weight, m.weight) and vd.reducer_of(m2.weight) is None
    red.remove()
    assert vd.reducer_of(m.weight) is None


def test_deferred_unscale_touches_only_the_scalers_parameters():
    """A reducer built over the encoder's parameters AND others (a decoder-side trainable, say): the others' gradients are plain when
    backward ends and must not be divided by S in finish()."""
    from vicasplat_amd import autograd as A
    torch.manual_seed(0)
    enc, other = torch.nn.Linear(6, 6), torch.nn.Linear(6, 2)
    x = torch.randn(5, 6)
    other(enc(x)).square().mean().backward()
    ref = {id(p): p.grad.clone() for p in list(enc.parameters()) + list(other.parameters())}
    for p in list(enc.parameters()) + list(other.parameters()):
        p.grad = None
    S = 4096.0
    scaler = A.BoundaryGradScale(enc, S)
    for bucket_bytes in (1 << 20, 32):               # one mixed bucket / one bucket per tensor
        red = vd.GradReducer(list(enc.parameters()) + list(other.parameters()), bucket_bytes=bucket_bytes)
        red.zero_grad()
        scaler.rearm()
        h = scaler.outputs(enc(x))[0]
        other(h).square().mean().backward()
        red.finish()
        for p in list(enc.parameters()) + list(other.parameters()):
            assert torch.allclose(p.grad, ref[id(p)], rtol=1e-5, atol=1e-9), float((p.grad - ref[id(p)]).abs().max())
        assert float(red.last_overflow) == 0.0
        red.remove()


def test_two_backwards_before_one_finish_accumulate_under_the_boundary_scale():
    from vicasplat_amd import autograd as A
    torch.manual_seed(1)
    enc = torch.nn.Linear(6, 3)
    x1, x2 = torch.randn(4, 6), torch.randn(4, 6)
    (enc(x1).square().mean() + enc(x2).square().mean()).backward()
    ref = [p.grad.clone() for p in enc.parameters()]
    for p in enc.parameters():
        p.grad = None
    scaler = A.BoundaryGradScale(enc, 1024.0)
    red = vd.GradReducer(enc.parameters())
    red.zero_grad()
    for x in (x1, x2):                               # micro-batches: the second backward finds gradients still in units of S
        scaler.rearm()
        scaler.outputs(enc(x))[0].square().mean().backward()
    red.finish()
    for p, r in zip(enc.parameters(), ref):
        assert torch.allclose(p.grad, r, rtol=1e-5, atol=1e-9)
    red.remove()
