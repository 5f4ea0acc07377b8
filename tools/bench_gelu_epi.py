import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from vicasplat_amd import ops
d = torch.device("cuda:0")
def bench(fn, n=30):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for M, N, K in ((49152, 4096, 1024), (49344, 4096, 1024), (49152, 3072, 768)):
    a = torch.randn(M, K, device=d).half(); w = (torch.randn(N, K, device=d) / K ** 0.5).half(); b = torch.randn(N, device=d)
    o = torch.empty(M, N, device=d, dtype=torch.float16)
    t0 = bench(lambda: ops.gemm(a, w, b, o, ops.EPI_STORE16)); t1 = bench(lambda: ops.gemm(a, w, b, o, ops.EPI_GELU16))
    ref = torch.nn.functional.gelu(a[:4096].float() @ w.float().t() + b)
    err = (o[:4096].float() - ref).abs().max().item()
    print(f"M={M} N={N} K={K}: store16 {t0:7.1f} us  gelu16 {t1:7.1f} us   max err vs f32 gelu {err:.2e}")
