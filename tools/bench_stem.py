"""The fused stem (7x7 RGB conv + upsample-add, packed output): streaming kernel against the tile route.  python tools/bench_stem.py [frames=192]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 192
d = torch.device("cuda:0")
frames = torch.rand(N, 3, 256, 256, device=d) * 2 - 1
w = torch.randn(256, 3, 7, 7, device=d) * 0.1
b = torch.randn(256, device=d) * 0.1
trunk = torch.randn(N, 128, 128, 256, device=d)
img = ops.pad_rgb_nhwc(frames, torch.float32)
wp = ops.pack_conv7x7_rgb_weight(w, "split"); e = ops.split_scale_exp(w)
def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) / n
a = timeit(lambda: ops.conv7x7_rgb_nhwc(img, wp, b, 256, 256, up_add=trunk))
c = timeit(lambda: ops.stem7x7_up_split_stream(img, w, b, 256, 256, trunk, e))
gb = N * 65536 * 1024 / 1e9 + trunk.numel() * 4 / 1e9
print(f"{N} frames: tile route {a:.3f} ms, streaming {c:.3f} ms = {gb / c:.2f} TB/s of {gb:.1f} GB")
