"""Per-shape breakdown of every GEMM launch group of ONE bench step (event-timed on the launch stream): where the GEMM family's
time goes -- main 256x256 launches, tails, skinny head GEMMs.  python tools/gemm_breakdown.py [scenes_per_gpu]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import ops, synthetic
from vicasplat_amd.model.encoder import default_cfg, get_encoder

B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
d = torch.device("cuda:0")
shapes = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "shapes_full.json")))
enc, _ = get_encoder(default_cfg()); enc.load_state_dict(synthetic.golden_weights(shapes, 0), strict=True); enc = enc.to(d).eval().requires_grad_(False); enc.set_compute_dtype(os.environ.get('VS_DTYPE', 'split'))
img, K = synthetic.synthetic_input(B, 8, 256, 0)
ctx = dict(image=img.to(d), intrinsics=K.to(d))
for _ in range(2):
    enc(ctx, compute_viewspace_depth=False)
rec = []
def wrap(name):
    orig = getattr(ops, name)
    def f(a, w, bias, out, *rest, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); r = orig(a, w, bias, out, *rest, **k); e.record()
        M = k.get("M") or a.shape[0]
        epi = rest[0] if (name == "gemm" and rest) else ("rope" if name == "gemm_qkv_rope" else k.get("epilogue"))
        rec.append(((name, M, w.shape[0], a.shape[1], epi, bool(k.get("gate") is not None)), s, e))
        return r
    setattr(ops, name, f)
wrap("gemm"); wrap("gemm_qkv_rope")
enc(ctx, compute_viewspace_depth=False)
torch.cuda.synchronize()
agg = {}
for key, s, e in rec:
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += s.elapsed_time(e)
tot = sum(v[1] for v in agg.values()); totf = 0.0
print(f"{'op':14s} {'M':>7s} {'N':>5s} {'K':>5s} {'epi':>5s} gate calls   ms_total  us/call   TF/s  tiles256 rounds")
for key, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    name, M, N, Kk, epi, gate = key
    fl = 2.0 * M * N * Kk
    totf += fl * n
    tiles = -(-M // 256) * -(-N // 256)
    print(f"{name:14s} {M:7d} {N:5d} {Kk:5d} {str(epi):>5s} {int(gate):4d} {n:5d} {ms:10.3f} {ms / n * 1e3:8.1f} {fl * n / (ms * 1e-3) / 1e12:6.0f} {tiles:9d} {tiles / 256:6.2f}")
print(f"total {tot:.2f} ms, {totf / 1e12:.2f} TFLOP, {totf / (tot * 1e-3) / 1e12:.0f} TF/s")
