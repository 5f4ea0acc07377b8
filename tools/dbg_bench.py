import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = ["bench.py"]
import torch, bench
from vicasplat_amd import synthetic
from vicasplat_amd.model.decoder import DecoderSplattingCUDACfg, get_decoder
from vicasplat_amd.model.encoder import default_cfg, get_encoder
from vicasplat_amd.model.types import Gaussians
dev = torch.device("cuda:0")
shapes = json.load(open(os.path.join(os.path.dirname(bench.__file__), "tests", "golden", "shapes_full.json")))
enc, _ = get_encoder(default_cfg()); enc.load_state_dict(synthetic.golden_weights(shapes, 0), strict=True); enc = enc.to(dev).eval()
dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], False)).to(dev)
for B in (8, 4):
    img, K = synthetic.synthetic_input(B, 8, 256, 0)
    ctx = dict(image=img.to(dev), intrinsics=K.to(dev))
    tE, tK, tn, tf = bench.target_cameras(B, 12, dev)
    for it in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = enc(ctx, compute_viewspace_depth=False)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        g = out["gaussians"]
        r = dec(Gaussians(g.means, g.covariances, g.harmonics, g.opacities), tE, tK, tn, tf, (256, 256))
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(B, it, "enc %.1f ms  dec %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), "mem GB %.1f reserved %.1f" % (torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9), flush=True)
