"""Rasterizer-only timing on the BASELINE config-4/5 sized scene (8 views -> 524k Gaussians, 12 target views)."""
import argparse, json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import raster_ref as rr
from vicasplat_amd.model.decoder.cuda_splatting import render_batched

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=8); ap.add_argument("--targets", type=int, default=12)
ap.add_argument("--iters", type=int, default=20); ap.add_argument("--scenes", type=int, default=1)
a = ap.parse_args()
d = torch.device("cuda:0")
sc = rr.synthetic_scene(V=a.views, res=256, Vt=a.targets, seed=0)
T = lambda x: torch.tensor(x, dtype=torch.float32, device=d)
S = a.scenes
rep = lambda t: t[None].expand(S, *t.shape).contiguous()
m, cv, sh, op = rep(T(sc["means"])), rep(T(sc["covariances"])), rep(T(sc["harmonics"])), rep(T(sc["opacities"]))
E = T(sc["extrinsics"]).repeat(S, 1, 1); K = T(sc["intrinsics"]).repeat(S, 1, 1)
near = T(sc["near"]).repeat(S); far = T(sc["far"]).repeat(S)
cam_scene = torch.arange(S, dtype=torch.int32, device=d).repeat_interleave(a.targets)
bg = torch.zeros(S * a.targets, 3, device=d)
def step():
    return render_batched(E, K, near, far, (256, 256), bg, m, cv, sh, op, cam_scene)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(a.iters): step()
torch.cuda.synchronize(); dt = (time.time() - t0) / a.iters
print(json.dumps({"P": int(m.shape[1]), "scenes": S, "targets": a.targets, "ms_per_call": dt * 1e3, "views_per_s": S * a.targets / dt}))
