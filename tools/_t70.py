import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import callers, raster, synthetic
from vicasplat_amd.model.decoder import DecoderSplattingCUDACfg, get_decoder
from vicasplat_amd.model.encoder import default_cfg, get_encoder
from vicasplat_amd.model.types import Gaussians
B, V = 8, 8
d = torch.device("cuda:0")
shapes = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "shapes_full.json")))
enc, _ = get_encoder(default_cfg()); enc.load_state_dict(synthetic.golden_weights(shapes, 0), strict=True); enc = enc.to(d).eval(); enc.set_compute_dtype("f16")
dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], False)).to(d)
img, K = synthetic.synthetic_input(B, V, 256, 0)
ctx = dict(image=img.to(d), intrinsics=K.to(d))
out = enc(ctx, compute_viewspace_depth=False)
t_ = torch.linspace(0, 1, 10, dtype=torch.float32, device=d)
P_ = out["gaussian_camera_extrins"]
print("poses", P_[0, :3])
def T(f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, (time.perf_counter() - t) * 1e3
for _ in range(2):
    E70, ms = T(lambda: callers.interpolate_extrinsics(P_[:, :-1].reshape(-1, 4, 4), P_[:, 1:].reshape(-1, 4, 4), t_).reshape(B, 70, 4, 4))
    print("interp ms", ms, E70.dtype)
Kc = ctx["intrinsics"].float()
K70 = callers.interpolate_intrinsics(Kc[:, :-1].reshape(-1, 3, 3), Kc[:, 1:].reshape(-1, 3, 3), t_).reshape(B, 70, 3, 3)
g = out["gaussians"]
gs = Gaussians(g.means, g.covariances, g.harmonics, g.opacities)
near, far = torch.full((B, 70), 0.01, device=d), torch.full((B, 70), 100.0, device=d)
for _ in range(2):
    r, ms = T(lambda: dec(gs, E70, K70, near, far, (256, 256)))
    print("render 70 ms", ms, raster.last_call()["num_rendered"])
tE = torch.eye(4, device=d).repeat(B, 70, 1, 1); tE[:, :, 0, 3] = (torch.arange(70, device=d) * 0.01)[None]
tK = torch.tensor([[0.9, 0, 0.5], [0, 0.9, 0.5], [0, 0, 1.0]], device=d).repeat(B, 70, 1, 1)
for _ in range(2):
    r, ms = T(lambda: dec(gs, tE, tK, near, far, (256, 256)))
    print("render 70 (translated cams) ms", ms, raster.last_call()["num_rendered"])
