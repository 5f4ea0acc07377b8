"""Shapes and rocprof-independent timings of every conv / upsample call of one inference step (python tools/conv_shapes.py [scenes=24])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import ops

calls = []
def wrap(name):
    f = getattr(ops, name)
    def g(x, w, *a, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); r = f(x, w, *a, **k); e.record()
        calls.append((name, tuple(x.shape), tuple(w.shape), k.get("stride", 1), s, e))
        return r
    setattr(ops, name, g)
for n in ("conv3x3_nhwc", "conv3x3_head1x1_nhwc"):
    wrap(n)
scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 24
import json
from vicasplat_amd import synthetic
from vicasplat_amd.model.encoder import default_cfg, get_encoder
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "shapes_full.json")))
enc, _ = get_encoder(default_cfg())
enc.load_state_dict(synthetic.golden_weights(shapes, seed=0), strict=True)
enc = enc.to(dev).eval().requires_grad_(False); enc.set_compute_dtype(os.environ.get('VS_DTYPE', 'split'))
img, K = synthetic.synthetic_input(scenes, 8, 256, seed=0)
ctx = dict(image=img.to(dev), intrinsics=K.to(dev))
def step():
    with torch.no_grad():
        return enc(ctx, compute_viewspace_depth=False)
step(); calls.clear(); step(); torch.cuda.synchronize()
tot = 0.0
for name, xs, ws, st, s, e in calls:
    ms = s.elapsed_time(e)
    N, H, W, Cin = xs; Cout = ws[0]
    fl = 2.0 * N * (H // st) * (W // st) * Cin * Cout * 9
    tot += ms
    print(f"{name:22s} x {xs} w {ws} stride {st}: {ms:7.3f} ms  {fl / ms / 1e9:7.0f} TF/s")
print(f"total {tot:.2f} ms")
