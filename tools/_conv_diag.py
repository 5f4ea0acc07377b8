import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from vicasplat_amd import ops, synthetic
from vicasplat_amd.model.encoder import default_cfg, get_encoder
B = 24
d = torch.device("cuda:0")
shapes = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "shapes_full.json")))
enc, _ = get_encoder(default_cfg()); enc.load_state_dict(synthetic.golden_weights(shapes, 0), strict=True); enc = enc.to(d).eval()
img, K = synthetic.synthetic_input(B, 8, 256, 0)
ctx = dict(image=img.to(d), intrinsics=K.to(d))
orig = ops.conv3x3_nhwc
n = [0]
def f(x, w, bias=None, residual=None, relu_in=False, relu_out=False, out=None, stride=1, mask_by=None):
    r = orig(x, w, bias, residual, relu_in, relu_out, out, stride, mask_by)
    if x.shape[1] in (32, 64) and x.shape[0] == 192 and n[0] < 10 and not isinstance(w, ops.SplitWeight):
        n[0] += 1
        xs = x[:4].permute(0, 3, 1, 2).double()
        ref = F.conv2d(F.relu(xs) if relu_in else xs, w.permute(0, 3, 1, 2).double(), None if bias is None else bias.double(), stride=stride, padding=1).permute(0, 2, 3, 1)
        if residual is not None: ref = ref + residual[:4].double()
        if relu_out: ref = F.relu(ref)
        e = (r[:4].double() - ref).abs()
        print(tuple(x.shape), tuple(w.shape), "relu_in", relu_in, "relu_out", relu_out, "res", residual is not None, "bias", bias is not None,
              "max err", float(e.max()), "scale", float(ref.abs().max()), "x absmax", float(x.abs().max()), "nan", bool(torch.isnan(x).any()),
              "bad frac", float((e > 1e-4 * ref.abs().max()).double().mean()))
    return r
ops.conv3x3_nhwc = f
enc.set_compute_dtype("f32")
enc(ctx, compute_viewspace_depth=False); torch.cuda.synchronize()
