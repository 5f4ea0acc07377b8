"""Weight-gradient GEMM micro-benchmark: dW[N,K] = dY^T X over M rows (ViT-L / decoder / DPT-head shapes of the training step),
including the transposes + bias gradient that feed it.  python tools/bench_wgrad.py [--scenes 8]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import ops

ap = argparse.ArgumentParser(); ap.add_argument("--scenes", type=int, default=8); ap.add_argument("--atomics", action="store_true")
a = ap.parse_args()
WS = not a.atomics
d = torch.device("cuda:0")
M = a.scenes * 8 * 257
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
print(f"{'shape':28s} {'wgrad us':>9s} {'TF/s':>7s} {'ks':>4s} | {'transposes us':>13s} | {'dgrad us':>9s}")
for (m, N, K) in [(M, 3072, 1024), (M, 1024, 1024), (M, 4096, 1024), (M, 1024, 4096), (M, 2304, 768), (M, 768, 768), (M, 3072, 768), (M, 768, 3072),
                  (a.scenes * 8 * 128 * 128, 256, 256), (a.scenes * 8 * 256 * 256, 128, 128), (a.scenes * 8 * 256 * 256, 83, 128)]:
    dy = torch.randn(m, N, device=d).half(); x = torch.randn(m, K, device=d).half(); w = torch.randn(N, K, device=d).half()
    ks, unit = ops.wgrad_ksplit(N, K, m)
    db = torch.empty(N, device=d)
    dyT, xT = ops.transpose16(dy, unit, colsum_out=db, slices=ks), ops.transpose16(x, unit, slices=ks)
    dw = torch.zeros(N, K, device=d)
    tw = timeit(lambda: ops.gemm_wgrad(dyT, xT, dw, ks, workspace=WS))
    tt = timeit(lambda: (ops.transpose16(dy, unit, colsum_out=db, slices=ks), ops.transpose16(x, unit, slices=ks)))
    td = timeit(lambda: ops.linear_backward(dy, x, w, need_dw=False, need_db=False))
    extra = ""
    if N % 256 == 0 and K % 256 == 0:
        ttn = timeit(lambda: ops.gemm_wgrad_tn(dy, x, dw, ks, accumulate=False))
        tcs = timeit(lambda: ops.colsum(dy))
        extra = f" | TN (no transposes) {ttn * 1e6:8.1f} us {2 * m * N * K / ttn / 1e12:7.1f} TF/s + colsum {tcs * 1e6:6.1f} us"
    print(f"{str((m, N, K)):28s} {tw * 1e6:9.1f} {2 * m * N * K / tw / 1e12:7.1f} {ks:4d} | {tt * 1e6:13.1f} | {td * 1e6:9.1f}{extra}")
for (n, H, C1, C2) in [(a.scenes * 8, 256, 128, 128), (a.scenes * 8, 128, 256, 256), (a.scenes * 8, 64, 256, 256), (a.scenes * 8, 256, 256, 128)]:
    x = torch.randn(n, H, H, C1, device=d).half(); dy = torch.randn(n, H, H, C2, device=d).half()
    wp = torch.randn(C2, 3, 3, C1, device=d).half()
    t = timeit(lambda: ops.conv3x3_backward(dy, x, wp, relu_in=True, need_dx=False), 5)
    t2 = timeit(lambda: ops.conv3x3_backward(dy, x, wp, relu_in=True, need_dx=True), 5)
    fl = 2 * 9 * n * H * H * C1 * C2
    print(f"conv wgrad {n}x{H}x{H} {C1}->{C2}: {t * 1e3:7.2f} ms  {fl / t / 1e12:6.1f} TF/s (incl. transposes) | with dgrad {t2 * 1e3:7.2f} ms")
