"""Rasterizer-only forward + backward on the bench scene (encoder-predicted Gaussians of `--scenes` 8-view scenes, 12 target views each, MSE
against a random target): event-timed forward and backward, scaled to the bench step's 288 views.  Runs under rocprofv3 for the per-kernel
picture (tools/raster_fb_prof.sh).  python tools/bench_raster_fb.py [--scenes 8] [--iters 5] [--no-bwd] [--check]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import synthetic, raster
from vicasplat_amd.model.encoder import default_cfg, get_encoder
from vicasplat_amd.model.decoder.cuda_splatting import camera_matrices
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--scenes", type=int, default=8); ap.add_argument("--targets", type=int, default=12); ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--no-bwd", action="store_true"); ap.add_argument("--check", action="store_true", help="print gradient checksums (A/B of two builds)")
ap.add_argument("--tau", action="store_true", help="also differentiate w.r.t. the camera twist")
a = ap.parse_args()
dev = torch.device("cuda:0")
shapes = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "shapes_full.json")))
enc, _ = get_encoder(default_cfg()); enc.load_state_dict(synthetic.golden_weights(shapes, 0), strict=True); enc = enc.to(dev).eval().requires_grad_(False)
B, V, Vt = a.scenes, 8, a.targets
m, cv, sh, op = [], [], [], []
for s0 in range(0, B, 4):      # four scenes per encoder pass: the tool's footprint stays small
    img, K = synthetic.synthetic_input(B, V, 256, 0)
    out = enc(dict(image=img[s0:s0 + 4].to(dev), intrinsics=K[s0:s0 + 4].to(dev)), compute_viewspace_depth=False)
    g = out["gaussians"]
    m.append(g.means.flatten(1, 3).clone()); cv.append(g.covariances.flatten(1, 3).clone()); sh.append(g.harmonics.flatten(1, 3).clone()); op.append(g.opacities.flatten(1).clone())
    del out, g
m, cv, sh, op = (torch.cat(x) for x in (m, cv, sh, op))
del enc; torch.cuda.empty_cache()
tE, tK, tn, tf = bench.target_cameras(B, Vt, dev)
view_t, full_t, proj_t, campos, tanfov = camera_matrices(tE.flatten(0, 1), tK.flatten(0, 1), tn.flatten(), tf.flatten())
cam_scene = torch.arange(B, dtype=torch.int32, device=dev).repeat_interleave(Vt)
bg = torch.zeros(B * Vt, 3, device=dev)
target = torch.rand(B * Vt, 3, 256, 256, generator=torch.Generator().manual_seed(7)).to(dev)
leaves = [t.requires_grad_(not a.no_bwd) for t in (m, cv, sh, op)]
theta = torch.zeros(B * Vt, 3, device=dev, requires_grad=True) if a.tau else None
rho = torch.zeros(B * Vt, 3, device=dev, requires_grad=True) if a.tau else None


def fwd():
    return raster.rasterize(m, cv, op, view_t, full_t, campos, tanfov, bg, 256, 256, shs=sh, sh_degree=4, sh_rgb_major=True, cam_scene=cam_scene,
                            theta=theta, rho=rho, projmatrix_raw=proj_t)


ev = lambda: torch.cuda.Event(enable_timing=True)
tf_, tb_ = [], []
for it in range(a.iters + 1):
    for t in leaves + ([theta, rho] if a.tau else []):
        t.grad = None
    e0, e1, e2 = ev(), ev(), ev()
    e0.record()
    color = fwd()[0]
    e1.record()
    if not a.no_bwd:
        gcol = (color - target) * (2.0 / color.numel())
        e1.record()
        color.backward(gcol)
    e2.record()
    torch.cuda.synchronize()
    if it:
        tf_.append(e0.elapsed_time(e1)); tb_.append(e1.elapsed_time(e2))
R = raster.last_call()["num_rendered"]
sc = 288.0 / (B * Vt)
res = dict(scenes=B, views=B * Vt, P=int(m.shape[1]), num_rendered=int(R), fwd_ms=round(sum(tf_) / len(tf_), 3), bwd_ms=round(sum(tb_) / len(tb_), 3),
           fwd_ms_per_288_views=round(sum(tf_) / len(tf_) * sc, 2), bwd_ms_per_288_views=round(sum(tb_) / len(tb_) * sc, 2))
if a.check and not a.no_bwd:
    res["checks"] = {n: [float(t.grad.double().sum()), float(t.grad.double().abs().sum())] for n, t in zip(("means", "cov", "sh", "op"), leaves)}
    if a.tau:
        res["checks"]["theta"] = [float(theta.grad.double().sum()), float(theta.grad.double().abs().sum())]
        res["checks"]["rho"] = [float(rho.grad.double().sum()), float(rho.grad.double().abs().sum())]
    res["color_sum"] = float(color.double().sum())
print(json.dumps(res))
