"""vs_conv3x3_wgrad_split_stream against the tile route on the two narrow head layers of an S-scene training step.
python tools/bench_conv_wgrad_stream.py [scenes=8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import ops
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
d = torch.device("cuda:0")
def timeit(f, n=3):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for (H, Cin, Cout) in ((256, 128, 128), (128, 256, 128)):
    N = S * 8
    x = torch.randn(N, H, H, Cin, device=d)
    dy = torch.randn(N, H, H, Cout, device=d)
    w = torch.randn(Cout, Cin, 3, 3, device=d) * 0.02
    e = ops.split_scale_exp(w)
    ms = timeit(lambda: ops.conv3x3_wgrad_split_stream(dy, x))
    fl = 2.0 * N * H * H * Cin * Cout * 9
    ops._WGRAD_STREAM = False
    ms_old = timeit(lambda: ops.conv3x3_backward_split(dy, x, w, need_dx=False, scale_exp=e))
    ops._WGRAD_STREAM = True
    a, b = ops.conv3x3_wgrad_split_stream(dy, x)[0], None
    print(f"[{N},{H},{H}] {Cin}->{Cout}: stream {ms:.3f} ms = {fl / ms / 1e9:.0f} TF/s algorithmic x3 = {3 * fl / ms / 1e9 / 2500:.2f} of peak; tile route (wgrad only) {ms_old:.3f} ms")
    del x, dy
