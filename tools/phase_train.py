"""Wall-clock phases of one training step (GPU-synchronised) + caching-allocator statistics: where the time outside kernels goes."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import callers, synthetic
from vicasplat_amd.model.decoder import DecoderSplattingCUDACfg, get_decoder
from vicasplat_amd.model.encoder import default_cfg, get_encoder
from vicasplat_amd.model.encoder.train_forward import forward_train
from vicasplat_amd.model.types import Gaussians
import bench

ap = argparse.ArgumentParser(); ap.add_argument("--scenes", type=int, default=24); a = ap.parse_args()
d = torch.device("cuda:0")
shapes = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "shapes_full.json")))
enc, _ = get_encoder(default_cfg()); enc.load_state_dict(synthetic.golden_weights(shapes, 0), strict=True); enc = enc.to(d).train()
dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], False)).to(d)
B, V, Vt = a.scenes, 8, 4
img, K = synthetic.synthetic_input(B, V, 256, 0); img, K = img.to(d), K.to(d)
tE, tK, tn, tf = bench.target_cameras(B, Vt, d)
target = torch.rand(B, Vt, 3, 256, 256, device=d)
opt, _ = callers.configure_optimizer(enc, lr=1e-12)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(3):
    st0 = torch.cuda.memory_stats()
    t0 = sync(); opt.zero_grad(set_to_none=True)
    out = forward_train(enc, img, K, torch.float16); t1 = sync()
    g = out["gaussians"]
    gs = Gaussians(g["means"].flatten(1, 3), g["covariances"].flatten(1, 3), g["harmonics"].flatten(1, 3), g["opacities"].flatten(1))
    render = dec.forward(gs, tE, tK, tn, tf, (256, 256)); loss = callers.mse_loss(render.color, target, 1.0); t2 = sync()
    (loss * 1024.0).backward(); t3 = sync()
    params = [p for p in enc.parameters() if p.grad is not None]
    torch._foreach_mul_([p.grad for p in params], 1 / 1024.0); gn = torch.nn.utils.clip_grad_norm_(params, 0.5); ok = bool(torch.isfinite(gn)); opt.step(); t4 = sync()
    st1 = torch.cuda.memory_stats()
    dd = lambda k: st1[k] - st0[k]
    print(json.dumps(dict(it=it, enc_fwd_ms=round((t1 - t0) * 1e3, 1), raster_fwd_loss_ms=round((t2 - t1) * 1e3, 1), backward_ms=round((t3 - t2) * 1e3, 1),
                          clip_adamw_ms=round((t4 - t3) * 1e3, 1), total_ms=round((t4 - t0) * 1e3, 1), device_mallocs=dd("num_device_alloc"), device_frees=dd("num_device_free"),
                          alloc_retries=dd("num_alloc_retries"), reserved_gb=round(st1["reserved_bytes.all.peak"] / 2**30, 1), allocated_peak_gb=round(st1["allocated_bytes.all.peak"] / 2**30, 1))))
    del out, g, gs, render, loss
