"""One split-class attention shape, a few launches: the target of rocprofv3 --pmc passes.  python tools/one_attn.py [encoder|video]   (VS_ATTN_PACKED=0: the f32-input kernel)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import ops
d = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "encoder"
nb, H, L = (192, 16, 257) if which == "encoder" else (24, 12, 2064)
C = H * 64
qkv = torch.randn(nb * L, 3 * C, device=d)
if os.environ.get("VS_ATTN_PACKED", "1") != "0":       # round 4: packed q | k | v -> attention_sp_kernel
    qkv = ops.split_pack_weight(qkv, 0).data
out = torch.empty(nb * L, C, device=d)
for _ in range(5):
    ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, nbatch=nb, H=H, Lq=L, Lk=L, q_batch_rows=L, k_batch_rows=L, split=True)
torch.cuda.synchronize()
