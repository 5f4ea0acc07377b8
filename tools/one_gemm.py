"""One GEMM shape, a few launches: the target of rocprofv3 --pmc passes (M N K from argv, default ViT-L fc1 at 64 frames; VS_DTYPE=split (default) | f16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import ops
d = torch.device("cuda:0")
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (16384, 4096, 1024)
split = os.environ.get("VS_DTYPE", "split") == "split"
a = torch.randn(M, K, device=d); w = torch.randn(N, K, device=d) / K ** 0.5; b = torch.randn(N, device=d)
if split: w = ops.split_pack_weight(w)
else: a, w = a.half(), w.half()
o = torch.empty(M, N, device=d, dtype=a.dtype)
for _ in range(5): ops.gemm(a, w, b, o, ops.EPI_STORE16)
torch.cuda.synchronize()
