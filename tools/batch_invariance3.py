"""Scene 0 alone vs scene 0 in a batch of 3 (8 views), split class: max |difference| per output (0.0 = bit-identical).  Env switches are read by the library."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import synthetic
from vicasplat_amd.model.encoder import default_cfg, get_encoder
d = torch.device("cuda:0")
shapes = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "shapes_full.json")))
enc, _ = get_encoder(default_cfg()); enc.load_state_dict(synthetic.golden_weights(shapes, 0), strict=True); enc = enc.to(d).eval().requires_grad_(False)
enc.set_compute_dtype("split")
img, K = synthetic.synthetic_input(3, 8, 256, 5)
img, K = img.to(d), K.to(d)
try:
    a = enc(dict(image=img[:1], intrinsics=K[:1]), compute_viewspace_depth=False)
    b = enc(dict(image=img, intrinsics=K), compute_viewspace_depth=False)
    b2 = enc(dict(image=img, intrinsics=K), compute_viewspace_depth=False)
    torch.cuda.synchronize()
    res = {k: float((a[k][0] - b[k][0]).abs().max()) for k in ("pred_extrins", "raw_gaussians")}
    res.update({k: float((getattr(a["gaussians"], k)[0] - getattr(b["gaussians"], k)[0]).abs().max()) for k in ("means", "covariances", "harmonics", "opacities")})
    res["run_to_run_raw"] = float((b["raw_gaussians"] - b2["raw_gaussians"]).abs().max())
    print({k: os.environ[k] for k in os.environ if k.startswith("VS_")}, res)
except Exception as e:
    print({k: os.environ[k] for k in os.environ if k.startswith("VS_")}, "ERROR", repr(e)[:300])
