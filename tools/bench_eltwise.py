"""Streaming element-wise kernels at the training step's sizes (python tools/bench_eltwise.py): TB/s of relu_mask, gelu, gelu_backward."""
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
n = 48 * 256 * 256 * 128
dy = torch.randn(n, device=d).half(); x = torch.randn(n, device=d).half()
fns = {"relu_mask (3 x 2 B)": (lambda: ops.relu_mask(dy, x), 6 * n)}
z = torch.randn(49344, 4096, device=d).half()
if hasattr(ops, "gelu"): fns["gelu (2 x 2 B)"] = (lambda: ops.gelu(z), 4 * z.numel())
if hasattr(ops, "gelu_backward"): fns["gelu_backward (3 x 2 B)"] = (lambda: ops.gelu_backward(z, z), 6 * z.numel())
for N_ in (1024, 3072, 4096):
    zz = torch.randn(49344, N_, device=d).half()
    fns[f"colsum N={N_}"] = ((lambda t: (lambda: ops.colsum(t)))(zz), 2 * zz.numel())
for k, (f, byt) in fns.items():
    once(f, 2); t = min(once(f) for _ in range(4))
    print(f"{k:28s} {t:8.1f} us  {byt / t / 1e6:5.2f} TB/s")
