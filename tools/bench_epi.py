"""Epilogue micro-benchmark of the 256x256 GEMM kernel: the ViT-L shapes at an exact tile multiple (M = 49152) and with the
192-row tail of the 24-scene batch (M = 49344), for every fused epilogue.  python tools/bench_epi.py"""
import os, sys, math
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
for M in (49152, 49344):
    for N, K in ((1024, 1024), (4096, 1024), (1024, 4096), (3072, 1024), (768, 768), (2304, 768)):
        a = torch.randn(M, K, device=d).half(); w = (torch.randn(N, K, device=d) / K ** 0.5).half(); b = torch.randn(N, device=d)
        o = torch.empty(M, N, device=d, dtype=torch.float16); x = torch.zeros(M, N, device=d)
        gate = torch.randn(M // 257 + 1, N, device=d) * 0.1
        fl = 2.0 * M * N * K
        r = {}
        r["store16"] = bench(lambda: ops.gemm(a, w, b, o, ops.EPI_STORE16))
        r["gelu16"] = bench(lambda: ops.gemm(a, w, b, o, ops.EPI_GELU16))
        r["store32"] = bench(lambda: ops.gemm(a, w, b, x, ops.EPI_STORE32))
        r["resid32"] = bench(lambda: ops.gemm(a, w, b, x, ops.EPI_RESID32))
        r["resid32+gate"] = bench(lambda: ops.gemm(a, w, b, x, ops.EPI_RESID32, gate=gate, gate_rows=257))
        if N % 192 == 0:
            C = N // 3
            pos = torch.zeros(M, 2, dtype=torch.int32, device=d); pos[:, 0] = torch.arange(M, device=d) % 16; pos[:, 1] = (torch.arange(M, device=d) // 16) % 16
            r["rope16"] = bench(lambda: ops.gemm_qkv_rope(a, w, b, o, C, pos, None, 100.0, 30.0))
        print(f"M={M} N={N:5d} K={K:5d} " + "  ".join(f"{k} {v*1e6:7.1f}us {fl/v/1e12:5.0f}TF" for k, v in r.items()))
