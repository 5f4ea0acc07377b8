"""The fused Gaussian-parameter head convolution (conv3 256->256 -> ReLU -> conv1 256->83, split class, packed input) at the bench's size, a
few launches: the target of rocprofv3 --pmc passes.  python tools/one_conv.py [frames=192]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import ops
d = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 192
x = torch.randn(N, 128, 128, 256, device=d)
add = torch.randn(N, 256, 256, 256, device=d)
w = ops.split_pack_weight((torch.randn(256, 3, 3, 256, device=d) / 48.0))
w2 = ops.split_pack_weight(torch.nn.functional.pad(torch.randn(83, 256, device=d) / 16.0, (0, 0, 0, 13)))
b2 = torch.zeros(96, device=d)
xp = ops.upsample2x_nhwc(x, add=add, relu_add=True, packed=os.environ.get("VS_A_PACKED", "1") != "0")
for _ in range(3):
    y = ops.conv3x3_head1x1_nhwc(xp, w, None, w2, b2, 83)
torch.cuda.synchronize()
