"""One GEMM shape, a few launches: the target of rocprofv3 --pmc passes (M N K from argv, default ViT-L fc1 at 64 frames; VS_DTYPE=split (default) | f16; VS_A_PACKED=0: f32 A converted in the kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import ops
d = torch.device("cuda:0")
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (16384, 4096, 1024)
split = os.environ.get("VS_DTYPE", "split") == "split"
a = torch.randn(M, K, device=d); w = torch.randn(N, K, device=d) / K ** 0.5; b = torch.randn(N, device=d)
o = torch.empty(M, N, device=d, dtype=torch.float32 if split else torch.float16)
if split:
    w = ops.split_pack_weight(w)
    if os.environ.get("VS_A_PACKED", "1") != "0":      # the A operand as the LayerNorm / GELU producers of the model write it (round 3): packed (hi, lo)
        a = ops.split_pack_weight(a, 0)
else: a, w = a.half(), w.half()
for _ in range(5): ops.gemm(a, w, b, o, ops.EPI_STORE16)
torch.cuda.synchronize()
