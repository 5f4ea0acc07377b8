import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bench
from oracle import encoder_ref as er
from vicasplat_amd import callers
from vicasplat_amd.model.decoder import DecoderSplattingCUDACfg, get_decoder
from test_train_gpu import _tiny_model
dt = torch.float16
m, _ = _tiny_model(dt)
d = torch.device("cuda:0")
dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], False)).to(d)
B, V, Vt, S = 1, 2, 2, 64
img, K = er.synthetic_input(B, V, S, 3)
tE, tK, tn, tf = bench.target_cameras(B, Vt, d)
target = torch.rand(B, Vt, 3, S, S, generator=torch.Generator().manual_seed(5)).to(d)
E = torch.eye(4).repeat(B, V, 1, 1); E[:, :, 0, 3] = 0.1 * torch.arange(V)[None]
ctx = dict(image=img.to(d), intrinsics=K.to(d), extrinsics=E.to(d))
batch = dict(context=ctx, target=dict(image=target, extrinsics=tE, intrinsics=tK, near=tn, far=tf))
grads = lambda: {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in m.named_parameters()}
def wrapper_step(cam=1.0):
    m.zero_grad(set_to_none=True)
    mo = m(batch["context"], 0)
    rp = dec.forward(mo["gaussians"], tE, tK, tn, tf, (S, S), depth_mode=None)
    loss = ((rp.color - target) ** 2).mean()
    if cam: loss = loss + callers.camera_loss(mo["pred_extrins"], ctx["extrinsics"].float(), cam)
    loss.backward()
    return float(loss), grads()
rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-20))
def poison(val):
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    xs = [torch.full((64 << 20,), val, device=d) for _ in range(8)]      # 2 GB of `val` handed back to the caching allocator
    del xs
    torch.cuda.synchronize()
def ts():
    opt, _ = callers.configure_optimizer(m, lr=0.0)
    r = callers.training_step(m, dec, batch, opt, compute_dtype=dt, clip=1e30, camera_weight=1.0)
    return grads()
def worst(a, b):
    return sorted(((rel(a[n], b[n]), n) for n in a if a[n] is not None), reverse=True)[:3]
wrapper_step(); wrapper_step()
poison(0.0); W0 = wrapper_step()[1]
for val in (0.0, float("nan"), 1.0, 0.0, 1e-4, float("nan")):
    poison(val); W1 = wrapper_step()[1]
    errs = [(rel(W1[n], W0[n]), n) for n in W0 if W0[n] is not None]
    bad = [n for e, n in errs if e > 2e-6]
    print("poison", val, "max err %.2e" % max(errs)[0], "n_bad", len(bad), "last bad in registration order:", bad[-1] if bad else None)
