"""Split GEMM with the A operand converted in the kernel (f32 A) vs already packed (hi, lo): what a producer that writes the packed form buys.
python tools/bench_split_packed.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from vicasplat_amd import ops

d = torch.device("cuda:0")
M = 49344
for N, K, epi in ((3072, 1024, ops.EPI_STORE32), (1024, 1024, ops.EPI_STORE32), (4096, 1024, ops.EPI_GELU16), (1024, 4096, ops.EPI_STORE32), (2304, 768, ops.EPI_STORE32), (768, 768, ops.EPI_STORE32)):
    g = torch.Generator().manual_seed(N + K)
    a = torch.randn(M, K, generator=g).to(d)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(d)
    b = torch.randn(N, generator=g).to(d)
    wp = ops.split_pack_weight(w)
    ap = ops.split_pack_weight(a, 0)
    o1 = torch.empty(M, N, device=d); o2 = torch.empty(M, N, device=d)
    ops.gemm(a, wp, b, o1, epi); ops.gemm(ap, wp, b, o2, epi)
    same = torch.equal(o1, o2)
    def t(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    t1 = t(lambda: ops.gemm(a, wp, b, o1, epi)); t2 = t(lambda: ops.gemm(ap, wp, b, o2, epi)); t3 = t(lambda: ops.split_pack_weight(a, 0))
    fl = 2.0 * M * N * K / 1e9
    print(f"N={N} K={K} epi={epi}: f32-A {t1:.3f} ms ({fl / t1:.0f} TF/s)  packed-A {t2:.3f} ms ({fl / t2:.0f} TF/s)  pack pass {t3:.3f} ms  identical={same}", flush=True)
