"""vs_head1x1_backward_split against the operator-by-operator route on the two head shapes of an S-scene training step.
python tools/bench_head_bwd.py [scenes=8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import ops
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
d = torch.device("cuda:0")
P = S * 8 * 65536
def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for Cin, Cout in ((256, 83), (128, 3)):
    t = torch.randn(P, Cin, device=d).clamp_min_(0)
    dy = torch.randn(P, Cout, device=d)
    w = torch.randn(Cout, Cin, device=d) * 0.05
    e = ops.split_scale_exp(w)
    ms = timeit(lambda: ops.head1x1_backward_split(dy, t, w, relu=True, scale_exp=e))
    gb = P * (Cout + 2 * Cin) * 4 / 1e9
    def old():
        dx, dw, db = ops.linear_backward_split(dy, t, w, scale_exp=e)
        return ops.relu_mask(dx, t)
    ms_old = timeit(old, 2)
    print(f"Cin {Cin} Cout {Cout} P {P}: fused {ms:.3f} ms = {gb / ms:.2f} TB/s algorithmic ({gb:.1f} GB); operator route {ms_old:.3f} ms")
    del t, dy
