"""Phase timing of tile_sort_kernel on the bench scene via the VS_SORT_DEBUG early-return bits (run under rocprofv3)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vicasplat_amd import synthetic, raster
from vicasplat_amd.model.encoder import default_cfg, get_encoder
from vicasplat_amd.model.decoder.cuda_splatting import camera_matrices
import bench
dev = torch.device("cuda:0")
shapes = json.load(open("tests/golden/shapes_full.json"))
enc, _ = get_encoder(default_cfg()); enc.load_state_dict(synthetic.golden_weights(shapes, 0), strict=True); enc = enc.to(dev).eval()
B, V, Vt = 2, 8, 12
img, K = synthetic.synthetic_input(B, V, 256, 0)
out = enc(dict(image=img.to(dev), intrinsics=K.to(dev)), compute_viewspace_depth=False)
g = out["gaussians"]
tE, tK, tn, tf = bench.target_cameras(B, Vt, dev)
view_t, full_t, proj_t, campos, tanfov = camera_matrices(tE.flatten(0, 1), tK.flatten(0, 1), tn.flatten(), tf.flatten())
cam_scene = torch.arange(B, dtype=torch.int32, device=dev).repeat_interleave(Vt)
m, cv, sh, op = g.means.flatten(1, 3), g.covariances.flatten(1, 3), g.harmonics.flatten(1, 3), g.opacities.flatten(1)
for mode in (0,):
    os.environ["VS_SORT_DEBUG"] = str(mode)
    for _ in range(3):
        raster.forward_debug(m, cv, op, view_t, full_t, campos, tanfov, torch.zeros(B * Vt, 3, device=dev), 256, 256, shs=sh, sh_degree=4,
                             sh_rgb_major=True, cam_scene=cam_scene, count_touched=False)
    torch.cuda.synchronize()
