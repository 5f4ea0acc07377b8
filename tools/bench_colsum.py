"""vs_colsum on the bias-gradient shapes of the training step.  python tools/bench_colsum.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import ops
d = torch.device("cuda:0")
for dt in (torch.float16, torch.float32):
    for M, N in ((49344, 1024), (49344, 4096), (49344, 3072), (49536, 2304), (16448, 1024)):
        x = torch.randn(M, N, device=d).to(dt)
        want = x.double().sum(0)
        got = ops.colsum(x)
        err = float((got.double() - want).abs().max() / want.abs().max())
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): ops.colsum(x)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 20 * 1e3
        print(f"{str(dt):14s} [{M}, {N}] {us:7.1f} us {x.numel() * x.element_size() / us / 1e6:5.2f} TB/s rel err {err:.1e}")
