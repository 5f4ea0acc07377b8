"""LayerNorm (+AdaLN) micro-benchmark at the bench's row count: python tools/bench_ln.py  (VICASPLAT_HIP_LIB selects an alternative build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import ops
d = torch.device("cuda:0")
def once(fn, n=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for M, C in ((49344, 1024), (49536, 768)):
    x = torch.randn(M, C, device=d); w = torch.randn(C, device=d); b = torch.randn(C, device=d)
    o = torch.empty(M, C, device=d, dtype=torch.float16)
    sc = torch.randn(M // 257 + 1, C, device=d) * 0.1; sh = torch.randn(M // 257 + 1, C, device=d) * 0.1
    fns = {"plain": lambda: ops.layernorm_mod(x, w, b, o), "adaln": lambda: ops.layernorm_mod(x, w, b, o, scale=sc, shift=sh, mod_rows=257)}
    best = {k: 1e9 for k in fns}
    for f in fns.values(): once(f, 3)
    for _ in range(5):
        for k, f in fns.items(): best[k] = min(best[k], once(f))
    print(f"M={M} C={C}: " + "  ".join(f"{k} {v:6.1f} us = {M * C * 6 / v / 1e6:5.2f} TB/s" for k, v in best.items()))
