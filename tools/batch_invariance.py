"""Scene 0 of a B-scene batch must equal the same scene run alone (scenes are independent: SURVEY 8e).  python tools/batch_invariance.py [B] [dtype ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import synthetic
from vicasplat_amd.model.encoder import default_cfg, get_encoder
B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dts = sys.argv[2:] or ["split", "f32", "f16"]
d = torch.device("cuda:0")
shapes = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "shapes_full.json")))
enc, _ = get_encoder(default_cfg()); enc.load_state_dict(synthetic.golden_weights(shapes, 0), strict=True); enc = enc.to(d).eval().requires_grad_(False)
img, K = synthetic.synthetic_input(B, 8, 256, 0)
img, K = img.to(d), K.to(d)
for dt in dts:
    enc.set_compute_dtype(dt)
    a = enc(dict(image=img[:1], intrinsics=K[:1]), compute_viewspace_depth=False)
    b = enc(dict(image=img, intrinsics=K), compute_viewspace_depth=False)
    torch.cuda.synchronize()
    for k in ("means", "covariances", "harmonics", "opacities"):
        x, y = getattr(a["gaussians"], k)[0], getattr(b["gaussians"], k)[0]
        print(dt, k, "max|d| scene0 alone vs in batch:", float((x - y).abs().max()), "scale", float(x.abs().max()))
    print(dt, "pose", float((a["pred_extrins"][0] - b["pred_extrins"][0]).abs().max()))
