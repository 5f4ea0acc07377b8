"""Measures the statistics of the pts3d head's pre-expm1 output under the UNCALIBRATED golden weights on the
8-view synthetic input (CPU oracle, ViT-L) -> constants _PTS3D_FULL_MEAN/_STD in vicasplat_amd/synthetic.py.
Run once in the build container:  python tools/calibrate_scene.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vicasplat_amd.synthetic as syn
from oracle import encoder_ref as er

shapes = json.load(open("tests/golden/shapes_full.json"))
syn._calibrate_scene = lambda W, s: None  # measure the raw layer
W = syn.golden_weights(shapes, seed=0)
cfg = er.default_cfg()
img, K = syn.synthetic_input(1, 8, 256, 0)
B, V = 1, 8
frames = img.reshape(V, 3, 256, 256)
tok = er.lin(W, "backbone.intrinsic_encoder", K.reshape(V, 1, 9))
x, pos = er.encode_frames(W, cfg, frames, tok)
inter, cam = er.decoder(W, cfg, x.reshape(1, V, 257, -1), pos.reshape(1, V, 257, 2))
inter = [t[:, :, :-1].reshape(V, 256, -1) for t in inter]
pre = "downstream_head1.dpt"
h = er.dpt_trunk(W, pre, cfg, inter, 16, 16)
h = er.conv(W, pre + ".head.0", h, padding=1)
h = er.conv(W, pre + ".head.2", er.up2(h), padding=1)
o = er.conv(W, pre + ".head.4", torch.relu(h))
print("MEAN", tuple(o.mean(dim=(0, 2, 3)).tolist()))
print("STD", tuple(o.std(dim=(0, 2, 3)).tolist()))
print("bias", W[pre + ".head.4.bias"].tolist())
