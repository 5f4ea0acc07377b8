"""Probe of the split-class backward on the tiny encoder: per-parameter gradient error vs float64 autograd over the oracle as a function of the
loss scale (range of the f16 (hi, lo) operand pairs), and the magnitude spread of the gradients.  python tools/split_bwd_probe.py"""
import json, os, re, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import encoder_ref as er
from vicasplat_amd.model.encoder import default_cfg, get_encoder
from vicasplat_amd.model.encoder.train_forward import forward_train

TINY = dict(enc_depth=2, dec_embed_dim=192, dec_num_heads=3)
G = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
shapes = json.load(open(os.path.join(G, "shapes_tiny.json")))
W = er.golden_weights(shapes, seed=0)
B, V, S = 1, 3, 128
img, K = er.synthetic_input(B, V, S, 7)
g = torch.Generator().manual_seed(1)
r_raw = torch.randn(B, V, S, S, 86, generator=g) * 1e-3
r_raw[..., :3] *= 0.1
r_pose = torch.randn(B, V - 1, 8, generator=g)
r_cov = torch.randn(B, V, S, S, 3, 3, generator=g) * 10.0
Wr = {k: v.clone().double().requires_grad_() for k, v in W.items()}
o = er.forward.__wrapped__(Wr, er.default_cfg(**TINY), img.double(), K.double())
((o["raw_gaussians"] * r_raw.double()).sum() + (o["pred_extrins"] * r_pose.double()).sum() + (o["gaussians"]["covariances"] * r_cov.double()).sum()).backward()
for lg in [int(a) for a in sys.argv[1:]] or [4, 8, 12, 16, 20]:
    m, _ = get_encoder(default_cfg(**TINY))
    m.load_state_dict(W, strict=True)
    m = m.cuda().train()
    m.set_compute_dtype("split")
    out = forward_train(m, img.cuda(), K.cuda(), "split")
    loss = (out["raw_gaussians"] * r_raw.cuda()).sum() + (out["pred_extrins"] * r_pose.cuda()).sum() + (out["gaussians"]["covariances"] * r_cov.cuda()).sum()
    Sc = 2.0 ** lg
    (loss * Sc).backward()
    errs = {}
    for name, p in m.named_parameters():
        ref = Wr[re.sub(r"layer(\d)_rn", lambda mm: f"layer_rn.{int(mm.group(1)) - 1}", name)].grad
        if ref is None or p.grad is None:
            continue
        errs[name] = float((p.grad.cpu().double() / Sc - ref).abs().max() / (ref.abs().max() + 1e-300)) if torch.isfinite(p.grad).all() else float("inf")
    vals = sorted(errs.values())
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    if os.environ.get("VS_PROBE_ALL"):
        for k, v in errs.items():
            print(f"  {v:.2e}  {k}")
    print(f"scale 2^{lg}: median {vals[len(vals)//2]:.2e} p90 {vals[int(len(vals)*0.9)]:.2e} max {vals[-1]:.2e}  worst:", [(k, f"{v:.1e}") for k, v in worst], flush=True)
