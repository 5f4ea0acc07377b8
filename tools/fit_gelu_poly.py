"""Coefficients of gelu_poly2 (vicasplat_amd/csrc/gemm_common.h): erf(x / sqrt 2) ~ t * P(t^2) on t = clamp(x, -X, X) / X, P of degree
deg - 1, weighted (Lawson) minimax fit of the GELU error 0.5 * x * (approx - erf) under the constraint P(1) = 1 (seamless clamp);
prints the f32-evaluated maximum error over [-30, 30].  python tools/fit_gelu_poly.py [X=4.2] [deg=8]"""
import sys
import numpy as np
from scipy.special import erf

X = float(sys.argv[1]) if len(sys.argv) > 1 else 4.2
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 8
xs = np.cos(np.pi * (np.arange(6000) + 0.5) / 6000) * X / 2 + X / 2
t = xs / X
A = np.stack([t ** (2 * k + 1) for k in range(deg)], 1)
y = erf(xs / np.sqrt(2))
B = A[:, 1:] - A[:, :1]          # c0 = 1 - sum_{k >= 1} c_k
w = np.ones_like(xs)
best = None
for _ in range(300):
    d = np.linalg.lstsq(B * (w * xs)[:, None], (y - t) * w * xs, rcond=None)[0]
    c = np.concatenate([[1 - d.sum()], d])
    err = np.abs((A @ c - y) * xs * 0.5)
    if best is None or err.max() < best[0]:
        best = (err.max(), c.copy())
    w = w * (1 + 2 * err / err.max()); w /= w.mean()
c32 = best[1].astype(np.float32)
xt = np.linspace(-30, 30, 600001).astype(np.float32)
tt = (np.clip(xt, -X, X).astype(np.float32) * np.float32(1.0 / X)).astype(np.float32)
u = (tt * tt).astype(np.float32)
p = c32[-1] * np.ones_like(u)
for k in range(deg - 2, -1, -1):
    p = (p * u + c32[k]).astype(np.float32)
h = (np.float32(0.5) * xt).astype(np.float32)
g = (h * (p * tt).astype(np.float32) + h).astype(np.float32)
ref = 0.5 * xt.astype(np.float64) * (1 + erf(xt.astype(np.float64) / np.sqrt(2)))
print(f"X = {X}, {deg} coefficients (ascending powers of t^2):", ", ".join(f"{v:.9g}f" for v in c32))
print(f"fit error {best[0]:.2e}, f32-evaluated max |gelu_poly - gelu| = {np.abs(g - ref).max():.2e}")
