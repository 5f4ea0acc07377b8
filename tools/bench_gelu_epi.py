"""Cost of the GELU epilogue: the fc1 GEMMs of the 24-scene step with epilogue 0 (store) and 1 (exact-erf GELU)."""
import os, sys
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
for M, N, K in [(49344, 4096, 1024), (49344, 3072, 768)]:
    a = torch.randn(M, K, device=d).half(); w = (torch.randn(N, K, device=d) / K ** 0.5).half(); b = torch.randn(N, device=d)
    o = torch.empty(M, N, device=d, dtype=torch.float16)
    t0 = bench(lambda: ops.gemm(a, w, b, o, ops.EPI_STORE16)); t1 = bench(lambda: ops.gemm(a, w, b, o, ops.EPI_GELU16))
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: store {t0*1e6:7.1f} us {fl/t0/1e12:6.1f} TF/s | gelu {t1*1e6:7.1f} us {fl/t1/1e12:6.1f} TF/s | epilogue cost {100*(t1-t0)/t0:.1f} %")
