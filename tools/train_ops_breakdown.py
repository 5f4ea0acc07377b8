"""Per-(operator, shape) breakdown of ONE training step (event-timed on the launch stream): every function of vicasplat_amd.ops is
wrapped, so nested operators appear under both names (linear_backward_split AND the GEMMs / transposes inside it).
python tools/train_ops_breakdown.py [--scenes 8] [--dtype split] [--top 60]"""
import argparse, json, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import callers, ops, synthetic
from vicasplat_amd.model.decoder import DecoderSplattingCUDACfg, get_decoder
from vicasplat_amd.model.encoder import default_cfg, get_encoder
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--scenes", type=int, default=8); ap.add_argument("--dtype", default="split"); ap.add_argument("--top", type=int, default=60)
a = ap.parse_args()
CDT = {"f16": torch.float16, "bf16": torch.bfloat16, "split": "split"}[a.dtype]
d = torch.device("cuda:0")
shapes = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "shapes_full.json")))
enc, _ = get_encoder(default_cfg()); enc.load_state_dict(synthetic.golden_weights(shapes, 0), strict=True); enc = enc.to(d).train()
enc.set_compute_dtype(CDT)
dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], False)).to(d)
B, V, Vt = a.scenes, 8, 12
img, K = synthetic.synthetic_input(B, V, 256, 0)
tE, tK, tn, tf = bench.target_cameras(B, Vt, d)
target = torch.rand(B, Vt, 3, 256, 256, device=d)
batch = dict(context=dict(image=img.to(d), intrinsics=K.to(d)), target=dict(image=target, extrinsics=tE, intrinsics=tK, near=tn, far=tf))
opt, _ = callers.configure_optimizer(enc, lr=1e-12)
for _ in range(2):
    callers.training_step(enc, dec, batch, opt, compute_dtype=CDT)
rec = []
depth = [0]
def wrap(name, orig):
    def f(*args, **kw):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        shp = tuple(tuple(t.shape) for t in args if isinstance(t, torch.Tensor))[:4]
        extra = tuple(x for x in args if isinstance(x, (int, bool)) and not isinstance(x, torch.Tensor))[:3]
        depth[0] += 1; s.record()
        try:
            return orig(*args, **kw)
        finally:
            e.record(); depth[0] -= 1
            rec.append(((depth[0], name, shp, extra), s, e))
    return f
for name, obj in list(vars(ops).items()):
    if isinstance(obj, types.FunctionType) and obj.__module__ == ops.__name__ and name not in ("_dtype_code",):
        setattr(ops, name, wrap(name, obj))
torch.cuda.synchronize()
s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s0.record(); callers.training_step(enc, dec, batch, opt, compute_dtype=CDT); e0.record()
torch.cuda.synchronize()
agg = {}
for key, s, e in rec:
    v = agg.setdefault(key, [0, 0.0]); v[0] += 1; v[1] += s.elapsed_time(e)
print(f"step {s0.elapsed_time(e0):.1f} ms (with the wrappers' events)")
byname = {}
for (dep, name, shp, extra), (n, ms) in agg.items():
    v = byname.setdefault((dep, name), [0, 0.0]); v[0] += n; v[1] += ms
print("--- by operator (depth = nesting inside other ops functions)")
for (dep, name), (n, ms) in sorted(byname.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{dep} {name:34s} {n:5d} {ms:9.2f} ms")
print("--- by operator and shape")
for (dep, name, shp, extra), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
    print(f"{dep} {name:30s} {n:4d} {ms:9.3f} ms {ms / n * 1e3:9.1f} us  {shp} {extra}")
