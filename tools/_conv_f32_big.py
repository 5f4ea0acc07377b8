import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from vicasplat_amd import ops
d = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
N, H, W, Cin, Cout = 64, 32, 32, 256, 256
x = torch.randn(N, H, W, Cin, generator=g).to(d)
w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(d)
b = torch.randn(Cout, generator=g).to(d)
xn = x.permute(0, 3, 1, 2).double()
for relu_in in (False, True):
  for relu_out in (False, True):
    ref = F.conv2d(F.relu(xn) if relu_in else xn, w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
    if relu_out: ref = F.relu(ref)
    for name, wp in (("f32", ops.pack_conv3x3_weight(w, torch.float32)), ("split", ops.pack_conv3x3_weight(w, "split")), ("f16", ops.pack_conv3x3_weight(w, torch.float16))):
        xx = x.half() if name == "f16" else x
        y = ops.conv3x3_nhwc(xx, wp, b, relu_in=relu_in, relu_out=relu_out)
        e = (y.double() - ref).abs()
        bad = e > 1e-2 * ref.abs().max()
        idx = bad.nonzero()
        print(name, "relu_in", relu_in, "relu_out", relu_out, "rel err", float(e.max() / ref.abs().max()), "bad frac", float(bad.double().mean()),
              "bad channels%64", sorted(set((idx[:, 3] % 64).tolist()))[:20] if len(idx) else [], "bad px%256", sorted(set((((idx[:,0]*H+idx[:,1])*W+idx[:,2]) % 256).tolist()))[:40] if len(idx) else [])
