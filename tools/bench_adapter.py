"""vs_gaussian_adapter on the bench shape (24 scenes x 8 frames x 65536 pixels, f32 heads of the split class).  python tools/bench_adapter.py [frames=192]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 192
d = torch.device("cuda:0"); torch.manual_seed(0)
P = N * 65536
gs = torch.randn(N, 256, 256, 96, device=d)[..., :83].permute(0, 3, 1, 2)      # channels-last views, rows padded as the fused head kernels write them
pts = torch.randn(N, 256, 256, 4, device=d)[..., :3].permute(0, 3, 1, 2)
mask = torch.ones(25, device=d)
f = lambda: ops.gaussian_adapter(pts, gs, mask, scale_act="softplus", scale_min=0.0, scale_max=0.0, opacity_exponent=1.0)
o = f(); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5): f()
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 5
gb = P * (400 + 724) / 1e9
print(f"{N} frames: {ms:.3f} ms = {gb / ms:.2f} TB/s of {gb:.1f} GB; checksum {float(o['covariances'].double().sum()):.6e} {float(o['harmonics'].double().sum()):.6e} {float(o['raw'].double().sum()):.6e}")
