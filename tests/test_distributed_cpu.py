"""N>1 host logic on CPU: world_size 2, gloo backend (rendezvous on 127.0.0.1)."""
import os
import socket

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
            ncoll = red.finish()
            assert ncoll == len(red.buckets)
        # numpy arrays travel by value; tensors would travel as shared-memory handles that die with this process (the parent may read late)
        q.put((rank, {n: p.grad.detach().numpy().copy() for n, p in m.named_parameters()}))
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
    out = {r: {n: torch.from_numpy(a) for n, a in v.items()} for r, v in out.items()}
    m = _toy_model()
    x, y = _toy_batch()
    ((m(x) - y) ** 2).mean().backward()
    return out, {n: p.grad for n, p in m.named_parameters()}


def test_gradient_equivalence_two_ranks_half_batch_each():
    out, ref = _run_reducer(False)
    for rank in (0, 1):
        for n, g in out[rank].items():
            if ref[n] is None:
                assert float(g.abs().sum()) == 0.0, n
            else:
                assert torch.allclose(g, ref[n], rtol=1e-5, atol=1e-7), (rank, n)
    for n in out[0]:
        assert torch.equal(out[0][n], out[1][n])          # replicas stay bit-identical


def test_gradient_exchange_in_bf16_is_close():
    out, ref = _run_reducer(True)
    for n, g in out[0].items():
        if ref[n] is not None:
            assert torch.allclose(g, ref[n], rtol=2e-2, atol=1e-4), n
