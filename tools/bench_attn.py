import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import ops
d = torch.device("cuda:0")
def bench(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
# encoder: 64 frames x 16 heads x 257
for name, nb, H, Lq, Lk, seg in [("encoder", 64, 16, 257, 257, False), ("encoder192", 192, 16, 257, 257, False), ("video", 8, 12, 2064, 2064, False), ("neighbor", 64, 12, 257, 514, True)]:
    C = H * 64
    rows = nb * Lq
    qkv = torch.randn(rows, 3 * C, device=d).half()
    out = torch.empty(rows, C, device=d, dtype=torch.float16)
    if seg:
        T = 8
        segs = []
        for b in range(nb // T):
            for t in range(T):
                a_, b_ = (1, 1) if t == 0 else ((T - 2, T - 2) if t == T - 1 else (t - 1, t + 1))
                segs.append([(b * T + a_) * Lq, Lq, (b * T + b_) * Lq, Lq])
        kv = torch.tensor(segs, dtype=torch.int32, device=d)
        fn = lambda: ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, nbatch=nb, H=H, Lq=Lq, q_batch_rows=Lq, kv_seg=kv)
    else:
        fn = lambda: ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, nbatch=nb, H=H, Lq=Lq, Lk=Lk, q_batch_rows=Lq, k_batch_rows=Lk)
    t = bench(fn)
    fl = 4.0 * nb * H * Lq * Lk * 64
    print(f"{name:9s} {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s")
    # torch SDPA reference speed
    q = qkv[:, :C].reshape(nb, Lq, H, 64).transpose(1, 2).contiguous(); k = torch.randn(nb, H, Lk, 64, device=d).half(); v = torch.randn_like(k)
    t2 = bench(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v))
    print(f"   torch SDPA {t2*1e6:8.1f} us  {fl/t2/1e12:7.1f} TF/s")
