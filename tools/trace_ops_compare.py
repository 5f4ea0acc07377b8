"""Per-op output checksums of one encoder forward in two compute dtypes: the first op whose outputs part ways.  python tools/trace_ops_compare.py [B] [dtA] [dtB]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import ops, synthetic
from vicasplat_amd.model.encoder import default_cfg, get_encoder
B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dA, dB = (sys.argv[2], sys.argv[3]) if len(sys.argv) > 3 else ("split", "f32")
d = torch.device("cuda:0")
shapes = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "shapes_full.json")))
enc, _ = get_encoder(default_cfg()); enc.load_state_dict(synthetic.golden_weights(shapes, 0), strict=True); enc = enc.to(d).eval().requires_grad_(False)
img, K = synthetic.synthetic_input(B, 8, 256, 0)
ctx = dict(image=img.to(d), intrinsics=K.to(d))
NAMES = ["gemm", "gemm_qkv_rope", "attention", "layernorm_mod", "conv3x3_nhwc", "upsample2x_nhwc", "silu_cast", "linear_f32", "gaussian_adapter"]
log = []
def wrap(name):
    orig = getattr(ops, name)
    def f(*a, **k):
        r = orig(*a, **k)
        t = r if torch.is_tensor(r) else (r["means"] if isinstance(r, dict) else None)
        if t is not None:
            tf = t.float()
            log.append((name, tuple(t.shape), float(tf.abs().double().sum()), float(tf.double().sum())))
        return r
    setattr(ops, name, f)
for n in NAMES: wrap(n)
res = {}
for dt in (dA, dB):
    enc.set_compute_dtype(dt); log.clear()
    enc(ctx, compute_viewspace_depth=False); torch.cuda.synchronize()
    res[dt] = list(log)
bad = 0
for i, (x, y) in enumerate(zip(res[dA], res[dB])):
    rel = abs(x[2] - y[2]) / (abs(x[2]) + 1e-30)
    flag = rel > 1e-4
    if flag or i % 50 == 0:
        print(i, x[0], x[1], f"{x[2]:.6e} {y[2]:.6e} rel {rel:.2e}", "<<<<" if flag else "")
    bad += flag
    if bad > 12: break
