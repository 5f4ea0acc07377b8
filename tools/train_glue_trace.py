"""Which Python lines launch the PyTorch (non-HIP-library) kernels of the training step: one profiled step, aten ops grouped by the
innermost vicasplat_amd stack frames.  python tools/train_glue_trace.py [scenes=24] [f16|split]"""
import json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from vicasplat_amd import callers, synthetic
from vicasplat_amd.model.decoder import DecoderSplattingCUDACfg, get_decoder
from vicasplat_amd.model.encoder import default_cfg, get_encoder
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
d = torch.device("cuda:0")
shapes = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "shapes_full.json")))
enc, _ = get_encoder(default_cfg()); enc.load_state_dict(synthetic.golden_weights(shapes, 0), strict=True); enc = enc.to(d).train(); CDT = "split" if (len(sys.argv) > 2 and sys.argv[2] == "split") else torch.float16; enc.set_compute_dtype(CDT)
dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], False)).to(d)
img, K = synthetic.synthetic_input(B, 8, 256, 0)
tE, tK, tn, tf = bench.target_cameras(B, 12, d)
batch = dict(context=dict(image=img.to(d), intrinsics=K.to(d)), target=dict(image=torch.rand(B, 12, 3, 256, 256, device=d), extrinsics=tE, intrinsics=tK, near=tn, far=tf))
opt, _ = callers.configure_optimizer(enc, lr=1e-12)
for _ in range(2): callers.training_step(enc, dec, batch, opt, compute_dtype=CDT)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    callers.training_step(enc, dec, batch, opt, compute_dtype=CDT); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
for e in prof.key_averages(group_by_stack_n=12):
    t = getattr(e, "self_device_time_total", None) or getattr(e, "self_cuda_time_total", 0)
    if not e.key.startswith("aten::") or (t < 200 and e.count < 20): continue
    frames = [f for f in e.stack if "vicasplat_amd" in f or "bench" in f]
    where = " <- ".join(f.split("/")[-1] for f in frames[:3]) or (e.stack[0] if e.stack else "?")
    shp = str(getattr(e, "input_shapes", ""))[:60]
    a = agg[(e.key, where)]; a[0] += t; a[1] += e.count
tot = 0.0
for (k, w), (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    tot += t; print(f"{t / 1e3:8.2f} ms x{n:5d}  {k:28s} {w[:150]}")
print(f"listed: {tot / 1e3:.1f} ms")
