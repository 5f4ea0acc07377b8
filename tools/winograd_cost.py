"""is Winograd F(2x2, 3x3) worth building for the Gaussian-parameter head's conv 256 -> 256 @ 256^2 (34 ms per 24-scene step,
the largest kernel of the step)?  Its matrix work is 16 GEMMs [tiles, Cin] x [Cin, Cout] with K = Cin = 256 -- 2.25 x fewer MFMAs than the
implicit GEMM (K = 9 * 256), but SHORT K: the 256 x 256 tile's fixed cost (tools/gemm_fixed_cost.py: ~11 us per tile + 2.35 us per K-tile
of 32) is paid for 8 K-tiles instead of 72.  This script times exactly that matrix part on the product's own split GEMM (packed A, i.e. a
free input transform; plain f32 store, i.e. a free output transform): a LOWER bound of any Winograd kernel built on this main loop.
python tools/winograd_cost.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import ops
d = torch.device("cuda:0")
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 192
tiles = frames * (256 // 2) * (256 // 2)
C = 256
a = ops.split_pack_weight(torch.randn(tiles, C, device=d), 0)          # one position's transformed input, packed (hi, lo)
a = ops.SplitWeight(a.data, 1.0, a.shape)
w = ops.split_pack_weight(torch.randn(C, C, device=d) / 16)
out = torch.empty(tiles, C, device=d)
def run():
    for p in range(16):
        ops.gemm(a, w, None, out, ops.EPI_STORE32)
run(); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(3): run()
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 3
fl = 16 * 2.0 * tiles * C * C
print(f"{frames} frames: 16 position GEMMs [{tiles} x {C}] x [{C} x {C}]: {ms:.2f} ms = {fl / ms / 1e9:.0f} TFLOP/s algorithmic ({3 * fl / ms / 1e9:.0f} executed); "
      f"the implicit-GEMM convolution does the same layer in 34.1 ms per 192 frames -> Winograd's matrix part alone: {34.1 * frames / 192 / ms:.2f} x")
