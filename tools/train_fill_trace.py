"""Which autograd nodes launch the small PyTorch kernels (fills, copies, adds) of the training step: one profiled step, device-kernel-launching
aten ops grouped by (op, enclosing autograd Function / top-level op).  python tools/train_fill_trace.py [scenes=8] [f16|split]"""
import json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from vicasplat_amd import callers, synthetic
from vicasplat_amd.model.decoder import DecoderSplattingCUDACfg, get_decoder
from vicasplat_amd.model.encoder import default_cfg, get_encoder
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
d = torch.device("cuda:0")
shapes = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "shapes_full.json")))
enc, _ = get_encoder(default_cfg()); enc.load_state_dict(synthetic.golden_weights(shapes, 0), strict=True); enc = enc.to(d).train()
CDT = "split" if (len(sys.argv) <= 2 or sys.argv[2] == "split") else torch.float16; enc.set_compute_dtype(CDT)
dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], False)).to(d)
img, K = synthetic.synthetic_input(B, 8, 256, 0)
tE, tK, tn, tf = bench.target_cameras(B, 12, d)
batch = dict(context=dict(image=img.to(d), intrinsics=K.to(d)), target=dict(image=torch.rand(B, 12, 3, 256, 256, device=d), extrinsics=tE, intrinsics=tK, near=tn, far=tf))
opt, _ = callers.configure_optimizer(enc, lr=1e-12)
for _ in range(2): callers.training_step(enc, dec, batch, opt, compute_dtype=CDT)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    callers.training_step(enc, dec, batch, opt, compute_dtype=CDT); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0, collections.Counter()])
for e in prof.events():
    if not e.name.startswith("aten::") or not e.kernels: continue
    t = sum(k.duration for k in e.kernels)
    p, chain = e.cpu_parent, []
    while p is not None:
        chain.append(p.name); p = p.cpu_parent
    owner = next((c for c in chain if "Backward" in c or "Fn" in c or "autograd" in c.lower()), chain[-1] if chain else "(top)")
    top = chain[0] if chain else "(top)"
    a = agg[(e.name, top[:40], owner[:60])]
    a[0] += t; a[1] += 1; a[2][str(e.input_shapes)[:50]] += 1
tot = 0.0
for (k, top, owner), (t, n, shp) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    tot += t; print(f"{t / 1e3:7.2f} ms x{n:5d}  {k:22s} <- {top:40s} <- {owner:60s} {shp.most_common(1)[0][0]}")
print(f"listed: {tot / 1e3:.1f} ms")
