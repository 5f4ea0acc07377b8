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
