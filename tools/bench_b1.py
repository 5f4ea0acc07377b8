"""Single-scene latency (B = 1, 8 context views): encoder only and encoder + 12 rendered views, synchronised per iteration (median of --iters).
python tools/bench_b1.py [--iters 30] [--dtype split]"""
import argparse, json, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import raster, synthetic
from vicasplat_amd.model.decoder import DecoderSplattingCUDACfg, get_decoder
from vicasplat_amd.model.encoder import default_cfg, get_encoder
from vicasplat_amd.model.types import Gaussians
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=30); ap.add_argument("--dtype", default="split"); ap.add_argument("--scenes", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda:0")
shapes = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "shapes_full.json")))
enc, _ = get_encoder(default_cfg()); enc.load_state_dict(synthetic.golden_weights(shapes, 0), strict=True); enc = enc.to(dev).eval().requires_grad_(False)
enc.set_compute_dtype({"split": "split", "f16": torch.float16, "f32": torch.float32}[a.dtype])
dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], False)).to(dev)
B = a.scenes
img, K = synthetic.synthetic_input(B, 8, 256, 0)
ctx = dict(image=img.to(dev), intrinsics=K.to(dev))
tE, tK, tn, tf = bench.target_cameras(B, 12, dev)
cap = {"n": None}
def enc_only(): return enc(ctx, compute_viewspace_depth=False)
def full():
    o = enc(ctx, compute_viewspace_depth=False); g = o["gaussians"]
    with raster.instance_capacity(cap["n"]) as sc:
        r = dec(Gaussians(g.means, g.covariances, g.harmonics, g.opacities), tE, tK, tn, tf, (256, 256))
    if cap["n"] is None: cap["n"] = int(max(n for n, _ in sc.calls) * 1.25) + 65536
    return r
res = {}
for name, fn in (("encoder_only", enc_only), ("encoder_plus_12_views", full)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(a.iters):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    res[name] = round(statistics.median(ts), 3)
print(json.dumps(dict(scenes=B, dtype=a.dtype, **res)))
