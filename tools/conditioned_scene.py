"""Tunes / reports the conditioned synthetic checkpoint (synthetic.conditioned_weights): PSNR of the f16 / bf16 / split renders against the
exact-f32 HIP path on the same scene (GPU only: the f32 path is within 1e-5 dB of the CPU oracle chain, tests/test_f32_path_gpu.py).
python tools/conditioned_scene.py [resid_gain scale_bias opacity_bias dc_gain]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vicasplat_amd import synthetic
from vicasplat_amd.model.decoder.cuda_splatting import render_cuda
from vicasplat_amd.model.encoder import default_cfg, get_encoder
a = [float(x) for x in sys.argv[1:5]] + [0.1, 30.0, -1.5, 1.2][len(sys.argv) - 1:]
d = torch.device("cuda:0")
shapes = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "shapes_full.json")))
V, Vt = 2, 3
for name, W, (img, K) in (("plain", synthetic.golden_weights(shapes, 0), synthetic.synthetic_input(1, V, 256, 0)),
                          ("conditioned", synthetic.conditioned_weights(shapes, 0, *a), synthetic.smooth_input(1, V, 256, 0))):
    enc, _ = get_encoder(default_cfg()); enc.load_state_dict(W, strict=True); enc = enc.to(d).eval().requires_grad_(False)
    E = torch.eye(4, device=d).repeat(Vt, 1, 1); E[:, 0, 3] = torch.arange(Vt, device=d) * 0.05
    Kt = torch.tensor([[0.9, 0, 0.5], [0, 0.9, 0.5], [0, 0, 1.0]], device=d).repeat(Vt, 1, 1)
    near, far = torch.full((Vt,), 0.01, device=d), torch.full((Vt,), 100.0, device=d)
    res = {}
    for dt in ("f32", "split", "f16", "bf16"):
        enc.set_compute_dtype(dt)
        o = enc(dict(image=img.to(d), intrinsics=K.to(d)), compute_viewspace_depth=False)
        g = o["gaussians"]
        col, _ = render_cuda(E, Kt, near, far, (256, 256), torch.zeros(Vt, 3, device=d), g.means.flatten(1, 3)[0], g.covariances.flatten(1, 3)[0],
                             g.harmonics.flatten(1, 3)[0], g.opacities.flatten(1)[0])
        res[dt] = (col.double().cpu(), g)
    ref, gr = res["f32"]
    print(name, "render mean %.3f std %.3f; opacity mean %.3f; sigma mean %.4f; means z mean %.2f" % (
        float(ref.mean()), float(ref.std()), float(gr.opacities.mean()), float(gr.scales.mean()), float(gr.means[..., 2].mean())))
    for dt in ("split", "f16", "bf16"):
        mse = ((res[dt][0] - ref) ** 2).flatten(1).mean(1)
        print("   ", dt, "PSNR vs f32 per view:", [round(float(-10 * torch.log10(m)), 1) for m in mse],
              "max |d means| %.2e" % float((res[dt][1].means - gr.means).abs().max()))
