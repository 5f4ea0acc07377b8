"""Interleaved, min-of-rounds timing of the GEMM epilogues on the ViT-L shapes (robust against clock drift between variants).
python tools/bench_epi2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import ops
d = torch.device("cuda:0")
def once(fn, n=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
M = 49152
for N, K in ((1024, 1024), (4096, 1024), (1024, 4096), (3072, 1024)):
    a = torch.randn(M, K, device=d).half(); w = (torch.randn(N, K, device=d) / K ** 0.5).half(); b = torch.randn(N, device=d)
    o = torch.empty(M, N, device=d, dtype=torch.float16); x = torch.zeros(M, N, device=d)
    gate = torch.randn(M // 257 + 1, N, device=d) * 0.1
    fns = {"store16": lambda: ops.gemm(a, w, b, o, ops.EPI_STORE16), "gelu16": lambda: ops.gemm(a, w, b, o, ops.EPI_GELU16),
           "resid32": lambda: ops.gemm(a, w, b, x, ops.EPI_RESID32), "resid32+gate": lambda: ops.gemm(a, w, b, x, ops.EPI_RESID32, gate=gate, gate_rows=257)}
    if N % 192 == 0:
        C = N // 3
        pos = torch.zeros(M, 2, dtype=torch.int32, device=d); pos[:, 0] = torch.arange(M, device=d) % 16; pos[:, 1] = (torch.arange(M, device=d) // 16) % 16
        fns["rope16"] = lambda: ops.gemm_qkv_rope(a, w, b, o, C, pos, None, 100.0, 30.0)
    best = {k: 1e9 for k in fns}
    for f in fns.values(): once(f, 3)
    for _ in range(6):
        for k, f in fns.items(): best[k] = min(best[k], once(f))
    fl = 2.0 * M * N * K
    print(f"N={N:5d} K={K:5d} " + "  ".join(f"{k} {v:7.1f}us {fl / v / 1e6:5.0f}TF" for k, v in best.items()))
