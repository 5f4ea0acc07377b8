"""Does padding the row stride of A / W away from a power of two help the big GEMMs (L2-channel aliasing)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import ops
d = torch.device("cuda:0")
def bench(fn, n=30):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
for M, N, K in [(16384, 3072, 1024), (16384, 1024, 1024), (16384, 4096, 1024), (16384, 1024, 4096)]:
    for pad_a, pad_w, pad_o in [(0, 0, 0), (64, 0, 0), (0, 64, 0), (64, 64, 0), (64, 64, 64), (128, 128, 128)]:
        a = torch.randn(M, K + pad_a, device=d).half()[:, :K]
        w = (torch.randn(N, K + pad_w, device=d) / K ** 0.5).half()[:, :K]
        b = torch.randn(N, device=d)
        o = torch.empty(M, N + pad_o, device=d, dtype=torch.float16)[:, :N]
        t = bench(lambda: ops.gemm(a, w, b, o, ops.EPI_STORE16))
        print(f"M={M} N={N} K={K} pad a/w/o {pad_a}/{pad_w}/{pad_o}: {t*1e6:7.1f} us {2.0*M*N*K/t/1e12:7.1f} TF/s")
