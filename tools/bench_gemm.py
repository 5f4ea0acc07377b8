"""GEMM micro-benchmark: vs_gemm_bias_act against torch (hipBLASLt) on the encoder/decoder shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from vicasplat_amd import ops
d = torch.device("cuda:0")
def bench(fn, n=30):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
shapes = [(16448, 3072, 1024), (16448, 1024, 1024), (16448, 4096, 1024), (16448, 1024, 4096), (16512, 2304, 768), (16448, 768, 768), (16448, 3072, 768), (16448, 768, 3072), (64, 2304, 768), (64, 768, 3072),
          (4112, 3072, 1024), (4112, 1024, 1024), (4112, 4096, 1024), (4112, 1024, 4096), (4112, 2304, 768), (4112, 768, 768),
          (4112, 3072, 768), (4112, 768, 3072), (8224, 4096, 1024), (8224, 1024, 4096), (16, 2304, 768), (16, 768, 3072)]
for M, N, K in shapes:
    a = torch.randn(M, K, device=d).half(); w = (torch.randn(N, K, device=d) / K ** 0.5).half(); b = torch.randn(N, device=d)
    o = torch.empty(M, N, device=d, dtype=torch.float16)
    t1 = bench(lambda: ops.gemm(a, w, b, o, ops.EPI_STORE16))
    bh = b.half()
    t2 = bench(lambda: F.linear(a, w, bh)) if os.environ.get("NO_BLAS") is None else 1.0
    fl = 2.0 * M * N * K
    print(f"M={M:5d} N={N:5d} K={K:5d}  ours {t1*1e6:7.1f} us {fl/t1/1e12:7.1f} TF/s | hipBLASLt {t2*1e6:7.1f} us {fl/t2/1e12:7.1f} TF/s")
