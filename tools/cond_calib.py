import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import synthetic
from vicasplat_amd.model.encoder import default_cfg, get_encoder
d = torch.device("cuda:0")
shapes = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "shapes_full.json")))
synthetic._COND_CALIB = None      # measure WITHOUT the conditioned calibration
W = synthetic.conditioned_weights(shapes, 0)
enc, _ = get_encoder(default_cfg()); enc.load_state_dict(W, strict=True); enc = enc.to(d).eval().requires_grad_(False); enc.set_compute_dtype("f32")
for V in (8, 2):
    img, K = synthetic.smooth_input(1, V, 256, 0)
    o = enc(dict(image=img.to(d), intrinsics=K.to(d)), compute_viewspace_depth=False)
    m = o["gaussians"].means.double().reshape(-1, 3)
    dn = m.norm(dim=-1, keepdim=True)
    xyz = m / dn.clamp_min(1e-12) * torch.log1p(dn)
    print("V", V, "pre-exp xyz mean", [float(v) for v in xyz.mean(0)], "std", [float(v) for v in xyz.std(0)])
    wb = W["downstream_head1.dpt.head.4.bias"]; print("bias", wb[:3].tolist())
