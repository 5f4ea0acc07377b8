"""BASELINE config 3: re10k_8view full pipeline, ONE training step (encoder + decoder + rasterizer, forward + backward + AdamW)
on 1 x MI355X, synthetic data, random-init ViT-L weights.  First-version training path (vicasplat_amd.autograd)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import callers, synthetic
from vicasplat_amd.model.decoder import DecoderSplattingCUDACfg, get_decoder
from vicasplat_amd.model.encoder import default_cfg, get_encoder
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--scenes", type=int, default=1); ap.add_argument("--views", type=int, default=8); ap.add_argument("--targets", type=int, default=12)   # re10k_8view.yaml:20
ap.add_argument("--steps", type=int, default=3); ap.add_argument("--warmup", type=int, default=1); ap.add_argument("--checkpoint", action="store_true")
ap.add_argument("--dtype", default="f16", choices=["f16", "bf16", "split"])
a = ap.parse_args()
CDT = {"f16": torch.float16, "bf16": torch.bfloat16, "split": "split"}[a.dtype]
d = torch.device("cuda:0")
shapes = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "shapes_full.json")))
enc, _ = get_encoder(default_cfg()); enc.load_state_dict(synthetic.golden_weights(shapes, 0), strict=True); enc = enc.to(d).train()
enc.set_compute_dtype(CDT)
if a.checkpoint: enc.enable_gradient_checkpointing()
dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], False)).to(d)
B, V, Vt = a.scenes, a.views, a.targets
img, K = synthetic.synthetic_input(B, V, 256, 0)
tE, tK, tn, tf = bench.target_cameras(B, Vt, d)
target = torch.rand(B, Vt, 3, 256, 256, device=d)
batch = dict(context=dict(image=img.to(d), intrinsics=K.to(d)), target=dict(image=target, extrinsics=tE, intrinsics=tK, near=tn, far=tf))
opt, _ = callers.configure_optimizer(enc, lr=1e-12)   # timing only: random-init weights + real learning rates throw the scene off screen
for _ in range(a.warmup):
    r = callers.training_step(enc, dec, batch, opt, compute_dtype=CDT)
st0 = dict(torch.cuda.memory_stats())
torch.cuda.synchronize(); t0 = time.perf_counter()
per_step = []
for _ in range(a.steps):
    ts = time.perf_counter(); m0 = torch.cuda.memory_stats()["num_device_alloc"]
    r = callers.training_step(enc, dec, batch, opt, compute_dtype=CDT)
    if os.environ.get("VS_TRAIN_STEP_TIMES"):      # per-step wall time, device mallocs, live GB after the step (adds one sync per step)
        torch.cuda.synchronize()
        per_step.append((round((time.perf_counter() - ts) * 1e3, 1), torch.cuda.memory_stats()["num_device_alloc"] - m0,
                         round(torch.cuda.memory_allocated() / 2**30, 2)))
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
if per_step: print("per-step (ms, device mallocs, live GB):", per_step)
print(json.dumps(dict(config="re10k_8view training step fwd+bwd+AdamW", dtype=a.dtype, checkpointing=bool(a.checkpoint), scenes=B, views=V, targets=Vt, ms_per_step=round(dt * 1e3, 1),
                      scenes_per_s=round(B / dt, 3), loss=float(r["loss"]), grad_norm=float(r["grad_norm"]), skipped=bool(r["skipped"]),
                      peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2**30, 1),
                      reserved_gb=round(torch.cuda.memory_stats()["reserved_bytes.all.peak"] / 2**30, 1),
                      device_mallocs_in_timed_steps=torch.cuda.memory_stats()["num_device_alloc"] - st0["num_device_alloc"],
                      device_frees_in_timed_steps=torch.cuda.memory_stats()["num_device_free"] - st0["num_device_free"],
                      alloc_retries=torch.cuda.memory_stats()["num_alloc_retries"])))
