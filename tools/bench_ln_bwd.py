"""LayerNorm backward micro-benchmark at the training step's shapes (24 scenes x 8 frames x 257 tokens)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import ops
d = torch.device("cuda:0")
for (M, C, mod) in [(49344, 1024, 0), (49344, 768, 257), (16448, 1024, 0), (16448, 768, 257)]:
    x = torch.randn(M, C, device=d); w = torch.randn(C, device=d); b = torch.randn(C, device=d)
    dout = torch.randn(M, C, device=d).half(); dx = torch.zeros(M, C, device=d)
    G = M // mod if mod else 1
    sc = torch.randn(G, C, device=d) if mod else None
    f = lambda: ops.layernorm_backward(dout, x, w, b, scale=sc, mod_rows=mod, dx=dx, accumulate_dx=True)
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); t = (time.perf_counter() - t) / 20
    by = M * C * (2 + 4 + 8)
    print(f"M={M} C={C} mod={mod}: {t * 1e6:7.1f} us  {by / t / 1e12:5.2f} TB/s algorithmic (dout 2 B + x 4 B + dx read-modify-write 8 B per element)")
