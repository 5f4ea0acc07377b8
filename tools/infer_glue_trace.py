"""Which Python lines launch the PyTorch (non-HIP-library) kernels of the inference step: one profiled step, aten ops grouped by the
innermost vicasplat_amd stack frames.  python tools/infer_glue_trace.py [scenes=24]"""
import json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from vicasplat_amd import synthetic
from vicasplat_amd.model.decoder import DecoderSplattingCUDACfg, get_decoder
from vicasplat_amd.model.encoder import default_cfg, get_encoder
from vicasplat_amd.model.types import Gaussians
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
d = torch.device("cuda:0")
shapes = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "shapes_full.json")))
enc, _ = get_encoder(default_cfg()); enc.load_state_dict(synthetic.golden_weights(shapes, 0), strict=True); enc = enc.to(d).eval().requires_grad_(False); enc.set_compute_dtype(os.environ.get("VS_DTYPE", "split"))
dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], False)).to(d)
img, K = synthetic.synthetic_input(B, 8, 256, 0)
ctx = dict(image=img.to(d), intrinsics=K.to(d))
tE, tK, tn, tf = bench.target_cameras(B, 12, d)
def step():
    out = enc(ctx, compute_viewspace_depth=False); g = out["gaussians"]
    return dec(Gaussians(g.means, g.covariances, g.harmonics, g.opacities), tE, tK, tn, tf, (256, 256))
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
for e in prof.key_averages(group_by_stack_n=12):
    t = getattr(e, "self_device_time_total", None) or getattr(e, "self_cuda_time_total", 0)
    if t < 20: continue
    frames = [f for f in e.stack if "vicasplat_amd" in f or "bench" in f]
    where = " <- ".join(f.split("/")[-1] for f in frames[:3]) or (e.stack[0] if e.stack else "?")
    a = agg[(e.key, where)]; a[0] += t; a[1] += e.count
tot = 0.0
ATEN = len(sys.argv) > 2 and sys.argv[2] == 'aten'      # only the PyTorch ops (the glue between the HIP kernels)
for (k, w), (t, n) in [kv for kv in sorted(agg.items(), key=lambda kv: -kv[1][0]) if (not ATEN or kv[0][0].startswith('aten::'))][:60 if ATEN else 40]:
    tot += t; print(f"{t / 1e3:8.2f} ms x{n:5d}  {k:28s} {w[:170]}")
print(f"listed: {tot / 1e3:.1f} ms")
