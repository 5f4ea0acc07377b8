"""Attention backward micro-benchmark at the training step's three attention shapes (scenes x 8 frames)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import ops
ap = argparse.ArgumentParser(); ap.add_argument("--scenes", type=int, default=8); a = ap.parse_args()
d = torch.device("cuda:0")
B, T, N1 = a.scenes, 8, 257
def run(name, rows, H, nbatch, Lq, Lk, qbr, kbr, kvlen=None, seg=None, max_keys=0):
    C = H * 64
    qkv = torch.randn(rows, 3 * C, device=d).half() * 0.5
    out = torch.empty(rows, C, device=d).half(); lse = torch.empty(rows, H, device=d)
    ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, nbatch=nbatch, H=H, Lq=Lq, Lk=Lk, q_batch_rows=qbr, k_batch_rows=kbr, kv_seg=seg, q_kvlen=kvlen, lse=lse)
    dout = torch.randn_like(out)
    f = lambda: ops.attention_backward(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, dout, lse, nbatch=nbatch, H=H, Lq=Lq, Lk=Lk, q_batch_rows=qbr, k_batch_rows=kbr, kv_seg=seg, q_kvlen=kvlen, max_keys=max_keys)
    ff = lambda: ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, nbatch=nbatch, H=H, Lq=Lq, Lk=Lk, q_batch_rows=qbr, k_batch_rows=kbr, kv_seg=seg, q_kvlen=kvlen, lse=lse)
    res = []
    for fn in (ff, f):
        for _ in range(2): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize(); res.append((time.perf_counter() - t) / 10)
    print(f"{name:10s} fwd {res[0] * 1e6:8.1f} us   bwd {res[1] * 1e6:8.1f} us   ratio {res[1] / res[0]:.2f}")
run("encoder", B * T * N1, 16, B * T, N1, N1, N1, N1)
M2 = N1 + 1
# the real table (backbone_vica._pos_tables): image queries see all T*M2 keys, only the camera token of frame t is limited to frames <= t
kv = torch.full((B, T, M2), T * M2, dtype=torch.int32, device=d)
kv[:, :, 0] = ((torch.arange(T, device=d) + 1) * M2).int()[None]
kv = kv.reshape(-1).contiguous()
run("video", B * T * M2, 12, B, T * M2, T * M2, T * M2, T * M2, kvlen=kv)
seg = []
for b in range(B):
    for t in range(T):
        p, n = (t - 1 if t > 0 else 1), (t + 1 if t < T - 1 else T - 2)
        seg += [(b * T + p) * N1, N1, (b * T + n) * N1, N1]
run("cross", B * T * N1, 12, B * T, N1, 0, N1, 0, seg=torch.tensor(seg, dtype=torch.int32, device=d), max_keys=2 * N1)
