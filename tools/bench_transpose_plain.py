"""A/B of the plain packing transpose (vs_transpose_pack_split without a tap): tile kernel (VS_TP_PLAIN=0) against the chunk-store kernel at
64 / 128 / 256 rows per tile, both block orders; outputs and the fused column sums are compared bit for bit / to 1e-6."""
import os, sys, torch
sys.path.insert(0, ".")
from vicasplat_amd import ops

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

dev = torch.device("cuda:0")
for (R, C) in [(16448, 768), (16448, 1024), (16448, 2304), (16448, 3072), (16448, 4096), (16512, 768), (16512, 3072), (49344, 1024), (49344, 4096),
               (64 * 128 * 128, 256), (64 * 256 * 256, 128)]:
    x = torch.randn(R, C, device=dev)
    Rp = (R + 1023) // 1024 * 1024
    gb = 2 * R * C * 4 / 1e9
    os.environ["VS_TP_PLAIN"] = "0"
    db0 = torch.empty(C, device=dev)
    ref = ops.transpose_pack_split(x, Rp, colsum=db0).data.clone()
    line = f"[{R} x {C}]"
    a = t(lambda: ops.transpose_pack_split(x, Rp, colsum=db0))
    line += f" tile {a*1e3:.0f} us {gb/a:.2f} TB/s |"
    for tr in (64, 128, 256):
        for rf in (0, 1):
            os.environ["VS_TP_PLAIN"], os.environ["VS_TP_RFAST"] = str(tr), str(rf)
            db = torch.empty(C, device=dev)
            o = ops.transpose_pack_split(x, Rp, colsum=db).data
            same = torch.equal(o[:, : R // 32 * 32], ref[:, : R // 32 * 32]) and torch.equal(o[:, :(R + 31) // 32 * 32], ref[:, :(R + 31) // 32 * 32])
            dberr = float((db - db0).abs().max() / db0.abs().max())
            b = t(lambda: ops.transpose_pack_split(x, Rp, colsum=db))
            line += f" {tr}/{rf}: {b*1e3:.0f} us {gb/b:.2f}{'' if same and dberr < 1e-5 else ' MISMATCH %s %.1e' % (same, dberr)} |"
    print(line, flush=True)
    del x
