"""Same-process A/B of the packed split attention with the first tile's DMA issued before / after the Q loads (VS_ATTN_EARLY_DMA), the three
shapes of the default bench step; outputs compared bit for bit.  python tools/ab_attn_early.py [scenes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import ops
d = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 24
T = 8
def bench(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
tot = {0: 0.0, 1: 0.0}
for name, nb, H, Lq, Lk, seg, mask, reps in [("encoder", S * T, 16, 257, 257, False, False, 24), ("video", S, 12, T * 258, T * 258, False, True, 12),
                                             ("neighbor", S * T, 12, 257, 514, True, False, 12)]:
    C = H * 64
    rows = nb * Lq
    qkv = torch.randn(rows, 3 * C, device=d)
    qp = ops.split_pack_weight(qkv, 0).data
    kw = dict(nbatch=nb, H=H, Lq=Lq, q_batch_rows=Lq, split=True)
    if seg:
        segs = []
        for b in range(nb // T):
            for t in range(T):
                a_, b_ = (1, 1) if t == 0 else ((T - 2, T - 2) if t == T - 1 else (t - 1, t + 1))
                segs.append([(b * T + a_) * Lq, Lq, (b * T + b_) * Lq, Lq])
        kw["kv_seg"] = torch.tensor(segs, dtype=torch.int32, device=d)
    else:
        kw.update(Lk=Lk, k_batch_rows=Lk)
    if mask:
        kv = torch.full((nb, T, 258), Lk, dtype=torch.int32)
        kv[:, :, 0] = (torch.arange(T, dtype=torch.int32) + 1)[None] * 258
        kw["q_kvlen"] = kv.reshape(-1).contiguous().to(d)
    q, k, v = qp[:, :C], qp[:, C:2 * C], qp[:, 2 * C:]
    outs = {}
    for rnd in range(2):
        for early in (0, 1):
            os.environ["VS_ATTN_EARLY_DMA"] = str(early)
            out = ops.split_act(rows, C, d)
            t = bench(lambda: ops.attention(q, k, v, out, **kw))
            outs[early] = out.clone() if not isinstance(out, ops.SplitWeight) else out.data.clone()
            if rnd == 1:
                tot[early] += t * reps
            print(f"{name:9s} early={early} {t*1e6:8.1f} us x {reps} = {t*reps*1e3:.2f} ms", flush=True)
    print(f"{name:9s} bit-identical: {torch.equal(outs[0], outs[1])}")
print("per step: late %.2f ms, early %.2f ms" % (tot[0] * 1e3, tot[1] * 1e3))
