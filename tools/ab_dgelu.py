"""Epilogue 5 of the split GEMM (dX of the MLP's second linear x GELU'(z) in the epilogue) against the two-pass route: values against float64
and the time of the pair, on the encoder / decoder MLP shapes of the 8-scene training step.  python tools/ab_dgelu.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import ops
d = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in [(16448, 1024, 4096), (16512, 768, 3072), (200, 1024, 4096), (64, 768, 3072), (16448 * 3, 1024, 4096)]:
    g = torch.Generator().manual_seed(M)
    dy = torch.randn(M, N, generator=g).to(d) * 0.1
    z = torch.randn(M, K, generator=g).to(d) * 1.5
    a = ops.gelu16(z)
    w = torch.randn(N, K, generator=g).to(d) * 0.03
    e = ops.split_scale_exp(w)
    f2 = lambda: ops.linear_backward_split(dy, a, w, need_dw=False, need_db=False, scale_exp=e)[0]
    two = lambda: ops.gelu_backward(f2(), z)
    one = lambda: ops.linear_backward_split(dy, a, w, need_dw=False, need_db=False, scale_exp=e, dgelu_z=z)[0]
    r2, r1 = two(), one()
    zd = z.double()
    ref = (dy.double() @ w.double()) * (0.5 * (1 + torch.erf(zd / 2 ** 0.5)) + zd * torch.exp(-0.5 * zd * zd) / (2 * torch.pi) ** 0.5)
    sc = float(ref.abs().max())
    print(f"[{M} x {N} -> {K}] two-pass {t(two):7.1f} us (err {float((r2.double() - ref).abs().max()) / sc:.1e}) | epilogue {t(one):7.1f} us "
          f"(err {float((r1.double() - ref).abs().max()) / sc:.1e}) | plain dX GEMM {t(f2):7.1f} us", flush=True)
