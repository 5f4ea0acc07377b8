"""Time of the 256x256 GEMM kernel against K at whole rounds of tiles (M = 65536, N = 1024: 1024 tiles = 4 rounds of 256): the
intercept of the line is the per-tile cost that is not K loop (dispatch, prologue latency, epilogue).  python tools/gemm_fixed_cost.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import ops
d = torch.device("cuda:0")
M, N = 65536, 1024
def bench(fn, n=30):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
rows = []
SPLIT = os.environ.get("VS_DTYPE", "split") == "split"     # VS_DTYPE=f16: the 16-bit kernel
for K in (128, 256, 512, 1024, 2048, 4096):
    a = torch.randn(M, K, device=d); w = (torch.randn(N, K, device=d) / K ** 0.5); b = torch.randn(N, device=d)
    if SPLIT: w = ops.split_pack_weight(w)
    else: a, w = a.half(), w.half()
    o = torch.empty(M, N, device=d, dtype=a.dtype); x = torch.zeros(M, N, device=d)
    t0 = bench(lambda: ops.gemm(a, w, b, o, ops.EPI_STORE16)); t2 = bench(lambda: ops.gemm(a, w, b, x, ops.EPI_RESID32))
    rows.append((K, t0, t2))
    print(f"K={K:5d}  store16 {t0:7.1f} us = {t0 / 4:6.1f} us per round ({2.0 * M * N * K / t0 / 1e6:6.0f} TF/s)   resid32 {t2:7.1f} us = {t2 / 4:6.1f} per round")
(k1, a1, _), (k2, a2, _) = rows[3], rows[5]
slope = (a2 - a1) / (k2 - k1)
print(f"store16: slope {slope * (32 if SPLIT else 64) / 4:.3f} us per K-tile ({32 if SPLIT else 64} k) per round, intercept {(a1 - slope * k1) / 4:.1f} us per round (tile)")
