"""vs_im2col7x7_rgb on the training shapes.  python tools/bench_im2col.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import ops
d = torch.device("cuda:0")
for N, dt in ((64, torch.float32), (192, torch.float16)):
    fr = torch.rand(N, 3, 256, 256, device=d)
    ops.im2col7x7_rgb(fr, dt); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): o = ops.im2col7x7_rgb(fr, dt)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    print(f"{N} frames {dt}: {ms:.3f} ms = {o.numel() * o.element_size() / ms / 1e9:.2f} TB/s written")
