"""Generates tests/golden/encoder_full_v2_grads.npz: GRADIENTS of the REAL reference encoder (imported from /root/reference on CPU, see
ref_import.py; full ViT-L, 1 scene x 2 views x 256 x 256, float64) for a seeded linear functional of its outputs -- the same functional
tests/test_train_gpu.py already uses against the oracle at reduced depth:

    loss = <raw_gaussians, r_raw> + <pred_extrins, r_pose> + <gaussians.covariances, r_cov>

For ~40 parameters spread over the frame encoder, the decoder, both DPT heads and the pose head the file holds the gradient's sum, sum of
absolute values, largest magnitude and a fixed lattice of elements.  The fixture is data (inputs, weights and the functional are re-derivable
from seeds), never reference source.  Run in the build container only (about ten minutes):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_encoder_grad_golden.py
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

PARAMS = [
    "backbone.patch_embed.proj.weight", "backbone.intrinsic_encoder.weight", "backbone.camera_extrinsic_token",
    "backbone.enc_blocks.0.attn.qkv.weight", "backbone.enc_blocks.0.mlp.fc1.weight", "backbone.enc_blocks.0.norm1.weight",
    "backbone.enc_blocks.5.attn.proj.weight", "backbone.enc_blocks.11.mlp.fc2.weight", "backbone.enc_blocks.11.attn.qkv.bias",
    "backbone.enc_blocks.17.mlp.fc1.bias", "backbone.enc_blocks.23.attn.qkv.weight", "backbone.enc_blocks.23.mlp.fc2.weight",
    "backbone.enc_norm.weight", "backbone.decoder_embed.weight",
    "backbone.dec_blocks.0.attn.qkv.weight", "backbone.dec_blocks.0.modulation1.proj.weight", "backbone.dec_blocks.0.cross_attn.projk.weight",
    "backbone.dec_blocks.5.cross_attn.projq.weight", "backbone.dec_blocks.5.mlp.fc1.weight", "backbone.dec_blocks.5.mlp_cam.fc2.weight",
    "backbone.dec_blocks.5.cam_norm1.weight", "backbone.dec_blocks.11.modulation2.proj.weight", "backbone.dec_blocks.11.cross_attn.proj.weight",
    "backbone.dec_blocks.11.mlp.fc2.weight", "backbone.dec_blocks.11.norm3.weight", "backbone.dec_norm.weight", "backbone.camera_dec_norm.weight",
    "camera_extrinsic_head.1.weight", "camera_extrinsic_head.1.bias",
    "downstream_head1.dpt.act_postprocess.0.1.weight", "downstream_head1.dpt.scratch.layer3_rn.weight",
    "downstream_head1.dpt.scratch.refinenet2.resConfUnit1.conv1.weight", "downstream_head1.dpt.scratch.refinenet1.out_conv.weight",
    "downstream_head1.dpt.head.0.weight",
    "gaussian_param_head.dpt.input_merger.0.weight", "gaussian_param_head.dpt.act_postprocess.0.0.weight",
    "gaussian_param_head.dpt.scratch.layer1_rn.weight", "gaussian_param_head.dpt.scratch.refinenet4.resConfUnit2.conv2.weight",
    "gaussian_param_head.dpt.scratch.refinenet1.resConfUnit2.conv1.bias", "gaussian_param_head.dpt.head.0.weight", "gaussian_param_head.dpt.head.4.bias",
]
NLAT = 64


def functional(B, V, S=256):
    """The seeded cotangents (float32 values; the test regenerates them with the same CPU generator)."""
    g = torch.Generator().manual_seed(1)
    r_raw = torch.randn(B, V, S, S, 86, generator=g) * 1e-3
    r_raw[..., :3] *= 0.1
    r_pose = torch.randn(B, V - 1, 8, generator=g)
    r_cov = torch.randn(B, V, S, S, 3, 3, generator=g) * 10.0
    return r_raw, r_pose, r_cov


def lattice(n):
    return (np.arange(NLAT, dtype=np.int64) * 2654435761 % max(n, 1)).astype(np.int64)


def main():
    t0 = time.time()
    B, V, seed = 1, 2, 0
    model = ref_import.build_reference_encoder(None)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(er.golden_weights(shapes, seed=seed), strict=True)
    model = model.double().eval()          # eval: the heads' Dropout(0.1) is the identity on both sides of the comparison
    for p in model.parameters():
        p.requires_grad_(True)
    image, K = er.synthetic_input(B, V, 256, seed=seed)
    r_raw, r_pose, r_cov = functional(B, V)
    out = model(dict(image=image.double(), intrinsics=K.double()), compute_viewspace_depth=False)
    print(f"[grad golden] forward {time.time() - t0:.0f}s", flush=True)
    loss = (out["raw_gaussians"] * r_raw.double()).sum() + (out["pred_extrins"] * r_pose.double()).sum() + (out["gaussians"].covariances * r_cov.double()).sum()
    loss.backward()
    print(f"[grad golden] backward {time.time() - t0:.0f}s  loss {float(loss):.9e}", flush=True)
    named = dict(model.named_parameters())
    res = dict(cfg_B=B, cfg_V=V, cfg_seed=seed, loss=float(loss), names=np.array(PARAMS), nlat=NLAT)
    allsq = 0.0
    for n, p in named.items():
        if p.grad is not None:
            allsq += float((p.grad.double() ** 2).sum())
    res["grad_norm_all"] = allsq ** 0.5
    for i, n in enumerate(PARAMS):
        g = named[n].grad.detach().double().flatten()
        idx = lattice(g.numel())
        res[f"p{i}_stats"] = np.array([float(g.sum()), float(g.abs().sum()), float(g.abs().max()), float(g.numel())])
        res[f"p{i}_lat"] = g[torch.from_numpy(idx)].numpy()
        print(f"  {n:80s} |g|max {float(g.abs().max()):.3e}  sum {float(g.sum()):+.3e}")
    path = os.path.join(HERE, "encoder_full_v2_grads.npz")
    np.savez_compressed(path, **res)
    print(f"[grad golden] {path}: {os.path.getsize(path) / 1e3:.1f} kB in {time.time() - t0:.0f}s")


if __name__ == "__main__":
    main()
