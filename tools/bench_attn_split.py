"""Split-class attention, the three shapes of the default bench step (24 scenes x 8 views): f32-input kernel (round 3) vs the packed-input
kernel (round 4, attention_sp_kernel).  python tools/bench_attn_split.py [scenes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import ops
d = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 24
T = 8
def bench(fn, n=10):
    for _ in range(2): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
tot = {"f32": 0.0, "packed": 0.0}
for name, nb, H, Lq, Lk, seg, mask, reps in [("encoder", S * T, 16, 257, 257, False, False, 24), ("video", S, 12, T * 258, T * 258, False, True, 12),
                                             ("neighbor", S * T, 12, 257, 514, True, False, 12)]:
    C = H * 64
    rows = nb * Lq
    qkv = torch.randn(rows, 3 * C, device=d)
    qp = ops.split_pack_weight(qkv, 0).data
    out = ops.split_act(rows, C, d)
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
    sl = lambda t: (t[:, :C], t[:, C:2 * C], t[:, 2 * C:])
    fl = 4.0 * nb * H * Lq * Lk * 64
    for tag, src in (("f32", qkv), ("packed", qp)):
        t = bench(lambda: ops.attention(*sl(src), out, **kw))
        tot[tag] += t * reps
        print(f"{name:9s} {tag:7s} {t*1e6:8.1f} us  {fl/t/1e12:6.1f} TF/s algorithmic ({3*fl/t/1e12:6.1f} executed) x {reps} launches/step = {t*reps*1e3:.2f} ms")
print("per step: f32-input %.2f ms, packed %.2f ms" % (tot["f32"] * 1e3, tot["packed"] * 1e3))
