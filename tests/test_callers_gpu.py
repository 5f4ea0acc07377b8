"""Test-time pose alignment: Adam on camera twists through the HIP rasterizer forward+backward."""
import numpy as np
import pytest
import torch

from oracle import raster_ref as rr
from vicasplat_amd import callers

pytestmark = pytest.mark.gpu


def _smooth_scene(d, res=64, Vt=3):
    sc = rr.synthetic_scene(V=2, res=res, Vt=Vt, seed=5)
    m = sc["means"]
    sh = np.zeros_like(sc["harmonics"])
    sh[:, 0, 0] = np.sin(3 * m[:, 0]) * 1.2      # smooth albedo so the photometric loss has a basin
    sh[:, 1, 0] = np.cos(4 * m[:, 1]) * 1.2
    sh[:, 2, 0] = np.sin(2 * m[:, 0] + 3 * m[:, 1])
    sc["harmonics"] = sh
    sc["covariances"] = sc["covariances"] * 4.0
    sc["opacities"] = np.full_like(sc["opacities"], 0.8)
    return {k: torch.tensor(v, dtype=torch.float32, device=d) for k, v in sc.items()}


def test_align_poses_recovers_perturbed_cameras():
    from vicasplat_amd.model.decoder import DecoderSplattingCUDACfg, get_decoder
    from vicasplat_amd.model.types import Gaussians
    d = torch.device("cuda:0")
    sc = _smooth_scene(d)
    dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], False)).to(d)
    g = Gaussians(sc["means"][None], sc["covariances"][None], sc["harmonics"][None], sc["opacities"][None])
    E, K, near, far = sc["extrinsics"][None], sc["intrinsics"][None], sc["near"][None], sc["far"][None]
    with torch.no_grad():
        target = dec(g, E, K, near, far, (64, 64)).color
    gen = torch.Generator().manual_seed(0)
    tau = torch.cat([torch.randn(3, 3, generator=gen) * 0.03, torch.randn(3, 3, generator=gen) * 0.01], -1).to(d)
    E0 = callers.update_pose(tau[:, :3], tau[:, 3:], E[0])[None]
    E1, hist = callers.align_poses(dec, g, target, E0, K, near, far, steps=60, rot_lr=0.003, trans_lr=0.003,
                                   return_history=True)
    assert hist[-1] < 0.15 * hist[0], (hist[0].item(), hist[-1].item())
    err0 = (E0[0, :, :3, 3] - E[0, :, :3, 3]).norm(dim=-1).mean()
    err1 = (E1[0, :, :3, 3] - E[0, :, :3, 3]).norm(dim=-1).mean()
    assert err1 < 0.5 * err0, (err0.item(), err1.item())
    psnr0 = callers.compute_psnr(target[0], dec(g, E0, K, near, far, (64, 64)).color[0].detach())
    psnr1 = callers.compute_psnr(target[0], dec(g, E1, K, near, far, (64, 64)).color[0].detach())
    assert (psnr1 > psnr0 + 3).all(), (psnr0, psnr1)


def test_render_video_interpolation_shares_one_gaussian_set():
    """demo.py:204-243: 10 cameras per interval between the predicted key poses, rendered from ONE Gaussian set in a single batched
    rasterizer call, forward path then its reverse; every frame equals the per-camera render_cuda of the same interpolated camera."""
    from vicasplat_amd.model.decoder.cuda_splatting import render_cuda
    from vicasplat_amd.model.types import Gaussians
    d = torch.device("cuda:0")
    sc = _smooth_scene(d, res=64, Vt=4)
    g = Gaussians(sc["means"].view(2, 64, 64, 3), sc["covariances"].view(2, 64, 64, 3, 3), sc["harmonics"].view(2, 64, 64, 3, 25),
                  sc["opacities"].view(2, 64, 64))
    poses, K = sc["extrinsics"], sc["intrinsics"][:1]
    frames = callers.render_video_interpolation(g, poses, K, n_interp_per_interv=10, image_shape=(64, 64))
    assert frames.shape == (2 * 3 * 10, 3, 64, 64) and torch.equal(frames[:30].flip(0), frames[30:])
    t = torch.linspace(0, 1, 10, device=d)
    E = callers.interpolate_extrinsics(poses[:-1], poses[1:], t).reshape(-1, 4, 4)
    for j in (0, 9, 17, 29):
        one, _ = render_cuda(E[j:j + 1], K, sc["near"][:1], sc["far"][:1], (64, 64), torch.zeros(1, 3, device=d), sc["means"],
                             sc["covariances"], sc["harmonics"], sc["opacities"])
        assert torch.equal(one[0], frames[j])
    assert float(frames.mean()) > 0.01
