"""Generates tests/golden/encoder_*.npz by running the REAL reference encoder (imported from /root/reference on
CPU, see ref_import.py) on seeded synthetic inputs with key-seeded golden weights, in float32 AND float64.

Run in the build container only:   PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_encoder_golden.py [tiny_v3 ... | examples]
The fixtures are data (inputs are re-derivable from seeds; outputs are sampled), never reference source.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_import  # noqa: E402
from oracle import encoder_ref as er  # noqa: E402  (only golden_weights / synthetic_input: shared seeded generators)

LATTICE = slice(8, 256, 16)  # 16x16 pixel lattice
PROBE = [0, 1, 7, 100, 1000, 5000, 20000, 50000]  # fixed flat indices (mod numel)


def checksum(t: torch.Tensor) -> np.ndarray:
    f = t.detach().double().flatten()
    idx = torch.tensor([p % f.numel() for p in PROBE])
    return torch.cat([f.mean()[None], f.abs().mean()[None], f[idx]]).numpy()


def run(model, image, K, dtype):
    model = model.to(dtype)
    sums = {}
    hooks = []
    def enc_hook(i):
        def h(m, a, o):
            sums[f"enc{i:02d}"] = checksum(o)  # returns None: a forward hook's return value would replace the output
        return h

    def dec_hook(i):
        def h(m, a, o):
            sums[f"dec{i:02d}_img"] = checksum(o[0])
            sums[f"dec{i:02d}_cam"] = checksum(o[1])
        return h

    for i, blk in enumerate(model.backbone.enc_blocks):
        hooks.append(blk.register_forward_hook(enc_hook(i)))
    for i, blk in enumerate(model.backbone.dec_blocks):
        hooks.append(blk.register_forward_hook(dec_hook(i)))
    with torch.no_grad():
        out = model(dict(image=image.to(dtype), intrinsics=K.to(dtype)), compute_viewspace_depth=False)
    for h in hooks:
        h.remove()
    g = out["gaussians"]
    lat = lambda t: t[:, :, LATTICE, LATTICE].double().numpy()
    res = dict(pred_extrins=out["pred_extrins"].double().numpy(), c2w=out["gaussian_camera_extrins"].double().numpy(),
               raw=lat(out["raw_gaussians"]), means=lat(g.means), covariances=lat(g.covariances), harmonics=lat(g.harmonics),
               opacities=lat(g.opacities), scales=lat(g.scales), rotations=lat(g.rotations),
               blocks=np.stack([sums[k] for k in sorted(sums)]), block_names=np.array(sorted(sums)))
    if out.get("confidence") is not None:        # predict_conf=true (config/experiment/distill.yaml:24): 1 + exp(4th channel of the pts3d head)
        res["confidence"] = out["confidence"][:, :, LATTICE, LATTICE].double().numpy()
        with torch.no_grad():                    # the stage that configuration runs: distill=True returns the centres + confidence only
            od = model(dict(image=image.to(dtype), intrinsics=K.to(dtype)), compute_viewspace_depth=False, distill=True)
        res["distill_centers"] = lat(od["gaussian_centers"])
        res["distill_confidence"] = od["confidence"][:, :, LATTICE, LATTICE].double().numpy()
    if out.get("pred_intrins") is not None:      # use_intrinsic_embedding=false: fov head + pinhole intrinsics
        res["pred_intrins"] = out["pred_intrins"].double().numpy()
        res["intrins_3x3"] = out["gaussian_camera_intrins"].double().numpy()
    return res


def make(name, overrides, B, V, seed=0, do_f64=True, predict_conf=False):
    t0 = time.time()
    model = ref_import.build_reference_encoder(overrides, predict_conf=predict_conf)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    if name.startswith("tiny_noint"):
        import json
        json.dump({k: list(v) for k, v in shapes.items()}, open(os.path.join(HERE, "shapes_tiny_noint.json"), "w"))
    W = er.golden_weights(shapes, seed=seed)
    missing, unexpected = model.load_state_dict(W, strict=True), None
    image, K = er.synthetic_input(B, V, 256, seed=seed)
    out = {}
    r32 = run(model, image, K, torch.float32)
    out.update({f"f32_{k}": v for k, v in r32.items()})
    if do_f64:
        r64 = run(model, image, K, torch.float64)
        out.update({f"f64_{k}": v for k, v in r64.items()})
    out["cfg_B"], out["cfg_V"], out["cfg_seed"] = B, V, seed
    out["cfg_overrides"] = np.array(repr(sorted((overrides or {}).items())))
    out["n_params"] = sum(int(np.prod(s)) for s in shapes.values())
    path = os.path.join(HERE, f"encoder_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"[golden] {path}: {os.path.getsize(path)/1e6:.2f} MB in {time.time()-t0:.1f}s; "
          f"f32 vs f64 max|d raw| = {np.abs(r32['raw'] - r64['raw']).max() if do_f64 else float('nan'):.3e}")


EXAMPLES = ("05b1462991e38e4d", "6c99592614256138")     # /root/reference/examples/<scene>/*.png: 8 real 256x256 frames each


def make_examples(fov_deg=50.0):
    """Goldens on the reference's OWN example frames: both scenes, 8 views, through the reference's
    demo pre-processing (demo.py:75-132 `load_images`, run from its source -- ref_import.reference_load_images) and its `inference`
    conventions (demo.py:180-202: pinhole K of the given field of view from cam_utils.simple_intrin_matrix_from_fov), with the key-seeded
    synthetic checkpoint AND the conditioned one.  Real images have what the synthetic sin + noise input lacks: flat regions, saturated
    pixels (6.6 % of the first scene's values are exactly +1), strong edges.  Stored: the pre-processed frames as uint8 (the tensors are
    exactly (u8 / 255 - 0.5) / 0.5), K, and the reference's f64 outputs on the 16 x 16 pixel lattice (poses in full) + f32 pose / raw."""
    from vicasplat_amd import synthetic
    t0 = time.time()
    ref_import.install()
    from src.misc.cam_utils import simple_intrin_matrix_from_fov
    load_images = ref_import.reference_load_images()
    frames = torch.stack([load_images(os.path.join(ref_import.REF, "examples", sc), size=256, verbose=False) for sc in EXAMPLES])   # [2,8,3,256,256]
    u8 = torch.round((frames * 0.5 + 0.5) * 255.0).to(torch.uint8)
    assert torch.equal((u8.float() / 255.0 - 0.5) / 0.5, frames)
    fov = torch.as_tensor([np.deg2rad(fov_deg), np.deg2rad(fov_deg)], dtype=torch.float)
    K1 = simple_intrin_matrix_from_fov(fov[None])                                    # demo.py:188-190
    K = K1.reshape(1, 1, 3, 3).expand(1, frames.shape[1], 3, 3).contiguous()
    out = dict(frames_u8=u8.permute(0, 1, 3, 4, 2).contiguous().numpy(), K=K.double().numpy(), fov_deg=fov_deg, scenes=np.array(EXAMPLES))
    model = ref_import.build_reference_encoder(None)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    for wname, W in (("golden", er.golden_weights(shapes, seed=0)), ("cond", synthetic.conditioned_weights(shapes, seed=0))):
        for si in range(len(EXAMPLES)):
            model.load_state_dict(W, strict=True)
            img = frames[si:si + 1]
            r64 = run(model, img, K, torch.float64)
            r32 = run(model, img, K, torch.float32)
            tag = f"{wname}_s{si}"
            for k in ("pred_extrins", "c2w", "raw", "covariances", "opacities", "scales", "rotations", "blocks"):
                out[f"{tag}_f64_{k}"] = r64[k]
            out[f"{tag}_f32_pred_extrins"] = r32["pred_extrins"].astype(np.float32)
            out[f"{tag}_f32_raw"] = r32["raw"].astype(np.float32)
            out[f"{tag}_f32_covariances"] = r32["covariances"].astype(np.float32)
            print(f"[golden] {tag}: f32 vs f64 max|d raw| {np.abs(r32['raw'] - r64['raw']).max():.3e}  pose {np.abs(r32['pred_extrins'] - r64['pred_extrins']).max():.3e}"
                  f"  max|enc activation probe| {np.abs(r64['blocks']).max():.3e}  ({time.time() - t0:.0f}s)")
    out["block_names"] = r64["block_names"]
    path = os.path.join(HERE, "encoder_full_v8_examples.npz")
    np.savez_compressed(path, **out)
    print(f"[golden] {path}: {os.path.getsize(path) / 1e6:.2f} MB in {time.time() - t0:.1f}s")


TINY = dict(enc_depth=2, dec_embed_dim=192, dec_num_heads=3)

if __name__ == "__main__":
    which = sys.argv[1:] or ["tiny_v3", "tiny_v2", "full_v2", "full_v8", "tiny_noint_v3"]
    if "examples" in which:
        make_examples()
    if "tiny_noint_v3" in which:    # the *_no_intrin checkpoints' architecture (README.md:51-53): no intrinsic token, fov head
        make("tiny_noint_v3", dict(TINY, use_intrinsic_embedding=False), B=2, V=3)
    if "tiny_v3" in which:
        make("tiny_v3", TINY, B=2, V=3)
    if "tiny_v2" in which:
        make("tiny_v2", TINY, B=1, V=2)
    if "tiny_conf_v3" in which:     # predict_conf=true (distill.yaml:24): 4-channel pts3d head, confidence = 1 + exp(x)
        make("tiny_conf_v3", TINY, B=2, V=3, predict_conf=True)
    if "full_conf_v2" in which:
        make("full_conf_v2", None, B=1, V=2, predict_conf=True)
    if "full_v2" in which:
        make("full_v2", None, B=1, V=2)
    if "full_v8" in which:
        make("full_v8", None, B=1, V=8)
