"""Tile-population statistics + per-kernel timing of the rasterizer on the bench scene (encoder-predicted Gaussians)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import synthetic, raster
from vicasplat_amd.model.encoder import default_cfg, get_encoder
from vicasplat_amd.model.decoder.cuda_splatting import camera_matrices
import bench

dev = torch.device("cuda:0")
shapes = json.load(open("tests/golden/shapes_full.json"))
enc, _ = get_encoder(default_cfg()); enc.load_state_dict(synthetic.golden_weights(shapes, 0), strict=True); enc = enc.to(dev).eval().requires_grad_(False)
B, V, Vt = 2, 8, 12
img, K = synthetic.synthetic_input(B, V, 256, 0)
out = enc(dict(image=img.to(dev), intrinsics=K.to(dev)), compute_viewspace_depth=False)
g = out["gaussians"]
tE, tK, tn, tf = bench.target_cameras(B, Vt, dev)
view_t, full_t, proj_t, campos, tanfov = camera_matrices(tE.flatten(0, 1), tK.flatten(0, 1), tn.flatten(), tf.flatten())
cam_scene = torch.arange(B, dtype=torch.int32, device=dev).repeat_interleave(Vt)
m, cv, sh, op = g.means.flatten(1, 3), g.covariances.flatten(1, 3), g.harmonics.flatten(1, 3), g.opacities.flatten(1)
torch.save(dict(means=m.cpu(), cov=cv.cpu(), sh=sh.cpu().half(), op=op.cpu()), "gpurun_out/bench_scene.pt") if len(sys.argv) > 1 else None
d = raster.forward_debug(m, cv, op, view_t, full_t, campos, tanfov, torch.zeros(B * Vt, 3, device=dev), 256, 256, shs=sh, sh_degree=4,
                         sh_rgb_major=True, cam_scene=cam_scene, count_touched=False)
n = (d["ranges"][..., 1] - d["ranges"][..., 0]).flatten().float()
print("R", d["R"], "tiles", n.numel(), "mean", n.mean().item(), "max", n.max().item(), "p50", n.median().item(),
      "p90", n.quantile(0.9).item(), "p99", n.quantile(0.99).item(), ">8192:", int((n > 8192).sum()), ">4096:", int((n > 4096).sum()),
      ">2048:", int((n > 2048).sum()))
rad = d["radii"].float(); vis = rad > 0
print("visible frac", vis.float().mean().item(), "radius mean", rad[vis].mean().item(), "max", rad.max().item())
nc = d["n_contrib"].float()
print("n_contrib mean", nc.mean().item(), "max", nc.max().item(), "final_T mean", d["final_T"].mean().item())
# depth-tie statistics of the sorted lists: runs of identical depth bits are ordered by an insertion sort in tile_sort_kernel
pl = d["point_list"].long()
Pn = m.shape[1]
cam_of = torch.repeat_interleave(torch.arange(B * Vt, device=dev), (d["ranges"][..., 1].max(1).values - d["ranges"][..., 0].min(1).values))
dep = d["geom"][cam_of, pl, 11]
print("adjacent equal-depth fraction in the sorted lists", (dep[1:] == dep[:-1]).float().mean().item())
hist = torch.histc(n, bins=16, min=0, max=32768)
print("tile population histogram (2048-wide bins):", hist.int().tolist())
