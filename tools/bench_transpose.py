"""Bandwidth of the split-class transposes (vs_transpose_f32 / vs_transpose_pack_split) and of colsum on the shapes of the training step."""
import sys, torch
sys.path.insert(0, ".")
from vicasplat_amd import ops

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

dev = torch.device("cuda:0")
for (R, C) in [(16384, 768), (16384, 1024), (16384, 3072), (16384, 4096), (49344, 1024), (64 * 128 * 128, 256), (64 * 256 * 256, 128)]:
    x = torch.randn(R, C, device=dev)
    Rp = (R + 1023) // 1024 * 1024
    gb = 2 * R * C * 4 / 1e9
    a = t(lambda: ops.transpose_f32(x, Rp))
    b = t(lambda: ops.transpose_pack_split(x, Rp))
    c = t(lambda: ops.colsum(x))
    d = t(lambda: ops.split16(x))
    print(f"[{R} x {C}] transpose {a*1e3:.0f} us {gb/a:.2f} TB/s | transpose_pack {b*1e3:.0f} us {gb/b:.2f} TB/s | colsum {c*1e3:.0f} us {gb/2/c:.2f} TB/s | split16 {d*1e3:.0f} us {gb/d:.2f} TB/s", flush=True)
    del x
