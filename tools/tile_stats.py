"""Distribution of the rasterizer's per-(camera, tile) list lengths and per-Gaussian tile counts on the bench scene
(python tools/tile_stats.py [scenes=2]): what the binning / sort / render kernels see."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from vicasplat_amd import raster, synthetic, _lib as L
from vicasplat_amd.model.decoder import DecoderSplattingCUDACfg, get_decoder
from vicasplat_amd.model.encoder import default_cfg, get_encoder
from vicasplat_amd.model.types import Gaussians
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
d = torch.device("cuda:0")
shapes = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "shapes_full.json")))
enc, _ = get_encoder(default_cfg()); enc.load_state_dict(synthetic.golden_weights(shapes, 0), strict=True); enc = enc.to(d).eval().requires_grad_(False); enc.set_compute_dtype(torch.float16)
dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], False)).to(d)
img, K = synthetic.synthetic_input(B, 8, 256, 0)
tE, tK, tn, tf = bench.target_cameras(B, 12, d)
states = []
orig = raster._forward_impl
def spy(*a, **k):
    r = orig(*a, **k); states.append(r[1]); return r
raster._forward_impl = spy
out = enc(dict(image=img.to(d), intrinsics=K.to(d)), compute_viewspace_depth=False)
g = out["gaussians"]
dec(Gaussians(g.means, g.covariances, g.harmonics, g.opacities), tE, tK, tn, tf, (256, 256))
torch.cuda.synchronize()
st = states[0]; S, P, Cn, M, H, W, _ = st["dims"]; t = st["alloc"].tensors
rg = t[L.VS_BUF_TILE_RANGES].view(torch.int32)[:Cn * 256 * 2].view(Cn * 256, 2).cpu().numpy()
n = (rg[:, 1] - rg[:, 0]).astype(np.int64)
print(f"cameras {Cn}, P {P}, R {st['num_rendered']}, tiles {n.size}")
print("tile list length: mean %.0f  p50 %d  p90 %d  p99 %d  max %d" % (n.mean(), *np.percentile(n, [50, 90, 99]).astype(int), n.max()))
for lo, hi in ((0, 1), (1, 1025), (1025, 4097), (4097, 16385), (16385, 1 << 30)):
    m = (n >= lo) & (n < hi); print(f"  tiles with {lo} <= n < {hi}: {m.sum()} ({100.0 * m.mean():.1f} %), keys {n[m].sum()} ({100.0 * n[m].sum() / max(n.sum(), 1):.1f} %)")
# ---- what dispatch order costs: greedy list scheduling of the (camera, tile) workgroups over S resident slots, cost = a + n (entries),
# in the launch order (camera-major, tile-minor) vs longest-first
import heapq
def makespan(costs, S):
    h = [0.0] * S
    for c in costs:
        t = heapq.heappop(h); heapq.heappush(h, t + c)
    return max(h)
for a in (64.0, 256.0):
    c = n.astype(np.float64) + a
    for S in (512, 1024, 2048):
        ideal = c.sum() / S
        print(f"  overhead {a:.0f} entries/tile, {S} slots: launch order {makespan(c, S) / ideal:.3f} x ideal, longest-first {makespan(np.sort(c)[::-1], S) / ideal:.3f} x, max tile / ideal {c.max() / ideal:.3f}")

if L.VS_BUF_RECT not in t: sys.exit(0)   # (the autograd path frees the transient buffers)
rect = t[L.VS_BUF_RECT].view(torch.int16)[:Cn * P * 4].view(Cn * P, 4).cpu().numpy().astype(np.int64)
cells = (rect[:, 2] - rect[:, 0]) * (rect[:, 3] - rect[:, 1])
vis = cells > 0
print("visible pairs %.1f %%; tiles per visible pair: mean %.2f p50 %d p90 %d p99 %d max %d" % (100.0 * vis.mean(), cells[vis].mean(), *np.percentile(cells[vis], [50, 90, 99]).astype(int), cells.max()))
geom = t[L.VS_BUF_GEOM].view(torch.float32)[:Cn * P * 12].view(Cn * P, 12)[:, 2:4].cpu().numpy()
ext = geom[vis]
print("footprint half extents (px) of visible pairs: x p50 %.2f p90 %.2f p99 %.2f; y p50 %.2f p90 %.2f p99 %.2f; culled by opacity (ext<0): %.1f %%" % (
    *np.percentile(ext[:, 0], [50, 90, 99]), *np.percentile(ext[:, 1], [50, 90, 99]), 100.0 * (ext[:, 0] < 0).mean()))
