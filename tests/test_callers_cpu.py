"""Host-side callers: pose update, PSNR/MSE, PLY and transforms.json writers. CPU only."""
import json

import numpy as np
import torch

from vicasplat_amd import callers


def _twist_matrix(tau):
    rho, th = tau[:3], tau[3:]
    X = torch.zeros(4, 4, dtype=tau.dtype)
    X[0, 1], X[0, 2], X[1, 0], X[1, 2], X[2, 0], X[2, 1] = -th[2], th[1], th[2], -th[0], -th[1], th[0]
    X[:3, 3] = rho
    return X


def test_se3_exp_matches_matrix_exponential():
    g = torch.Generator().manual_seed(0)
    tau = torch.randn(16, 6, generator=g, dtype=torch.float64) * 0.3
    tau[0] = 0
    tau[1, 3:] = 1e-7  # small-angle branch (cam_utils.py:76-77)
    T = callers.se3_exp(tau)
    ref = torch.stack([torch.linalg.matrix_exp(_twist_matrix(t)) for t in tau])
    assert torch.allclose(T, ref, atol=1e-12)


def test_update_pose_composes_on_the_world_to_camera_side():
    g = torch.Generator().manual_seed(1)
    tau = torch.randn(5, 6, generator=g, dtype=torch.float64) * 0.1
    ext = callers.se3_exp(torch.randn(5, 6, generator=g, dtype=torch.float64))
    new = callers.update_pose(tau[:, :3], tau[:, 3:], ext)
    assert torch.allclose(new.inverse(), callers.se3_exp(tau) @ ext.inverse(), atol=1e-12)
    zero = torch.zeros(5, 3, dtype=torch.float64)
    assert torch.allclose(callers.update_pose(zero, zero, ext), ext, atol=1e-12)


def test_psnr_and_mse():
    gt = torch.rand(3, 3, 8, 8)
    pred = gt + 0.1
    psnr = callers.compute_psnr(gt, pred)
    want = -10 * torch.log10(((gt.clip(0, 1) - pred.clip(0, 1)) ** 2).reshape(3, -1).mean(1))
    assert torch.allclose(psnr, want)
    assert torch.isclose(callers.mse_loss(pred, gt, 2.0), torch.tensor(2.0 * 0.01), rtol=1e-4)


def test_ply_round_trip(tmp_path):
    g = torch.Generator().manual_seed(2)
    n = 200
    means = torch.randn(n, 3, generator=g)
    scales = torch.rand(n, 3, generator=g) * 0.1 + 1e-3
    rot = torch.randn(n, 4, generator=g)
    sh = torch.randn(n, 3, 25, generator=g)
    op = torch.rand(n, generator=g) * 0.9 + 0.05
    op[:10] = 0.001  # pruned (< 0.005)
    for dc_only in (True, False):
        path = tmp_path / f"g{int(dc_only)}.ply"
        kept = callers.export_ply(means, scales, rot, sh, op, path, save_sh_dc_only=dc_only)
        assert kept == n - 10
        ply = callers.read_ply(path)
        assert list(ply)[:9] == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
        assert len(ply) == (17 if dc_only else 17 + 72)
        order = torch.sort(op[10:], descending=True).indices + 10
        assert np.allclose(ply["x"], means[order, 0].numpy())
        assert np.allclose(1 / (1 + np.exp(-ply["opacity"])), op[order].numpy(), atol=1e-6)
        assert np.all(np.diff(ply["opacity"]) <= 0)
        assert np.allclose(np.exp(ply["scale_1"]), scales[order, 1].numpy(), rtol=1e-5)
        q = rot[order] / rot[order].norm(dim=-1, keepdim=True)
        assert np.allclose(ply["rot_0"], q[:, 3].numpy(), atol=1e-6)  # w first
        assert np.allclose(ply["rot_1"], q[:, 0].numpy(), atol=1e-6)
        assert np.allclose(ply["f_dc_2"], sh[order, 2, 0].numpy())
        if not dc_only:
            assert np.allclose(ply["f_rest_24"], sh[order, 1, 1].numpy())  # channel-major flatten of bands 1..


def test_transforms_json(tmp_path):
    ext = torch.eye(4).repeat(3, 1, 1)
    ext[:, 0, 3] = torch.arange(3.0)
    callers.export_transforms(ext, tmp_path / "s" / "transforms.json")
    frames = json.load(open(tmp_path / "s" / "transforms.json"))
    assert [f["file_path"] for f in frames] == ["context/000000.png", "context/000001.png", "context/000002.png"]
    assert frames[2]["transform_matrix"][0][3] == 2.0


def test_load_images_resize_crop_normalise(tmp_path):
    from PIL import Image
    # 400 x 300 landscape: left half red, right half blue, a green band over the top 30 rows
    a = np.zeros((300, 400, 3), np.uint8)
    a[:, :200, 0] = 255; a[:, 200:, 2] = 255; a[:30, :, 1] = 255
    Image.fromarray(a).save(tmp_path / "b.png")
    Image.fromarray(np.full((100, 60, 3), 128, np.uint8)).save(tmp_path / "a.png")   # small portrait: enlarged (bicubic)
    (tmp_path / "notes.txt").write_text("ignored")
    x = callers.load_images(str(tmp_path), size=256)
    assert x.shape == (2, 3, 256, 256) and x.dtype == torch.float32
    assert float(x.min()) >= -1.0 and float(x.max()) <= 1.0
    assert torch.allclose(x[0], torch.full_like(x[0], 128 / 255 * 2 - 1), atol=1e-6)    # name order: a.png first
    img = x[1]                                                                           # 400x300 -> 341x256 -> centre 256x256
    assert float(img[0, 128, 10]) > 0.9 and float(img[2, 128, 10]) < -0.9                # left: red
    assert float(img[2, 128, 245]) > 0.9 and float(img[0, 128, 245]) < -0.9              # right: blue
    assert float(img[1, 5, 128]) > 0.9 and float(img[1, 200, 128]) < -0.9                # green band survives at the top
    # the crop is centred: the red/blue boundary sits in the middle column
    assert abs(int((img[0, 128] > 0).sum()) - 128) <= 2
    import pytest
    with pytest.raises(FileNotFoundError):
        callers.load_images([str(tmp_path / "notes.txt")])


def test_wgrad_split_heuristic_matches_the_kernel_contract():
    """ops.wgrad_ksplit (host logic): the (ksplit, padding unit) it picks must satisfy what vs_gemm_wgrad checks -- whole 256-tiles
    run on the 8-wave kernel with an even number (>= 2) of 64-wide K tiles per slice, everything else on 128x128 tiles with
    32-wide K steps and at least two slices -- and keep the launch near one wave of workgroups."""
    from vicasplat_amd.ops import wgrad_ksplit
    for rows, cols, red, taps in [(4096, 1024, 49344, 1), (768, 768, 16448, 1), (256, 256, 4_260_000, 9), (128, 128, 4_194_304, 9),
                                  (83, 128, 12_582_912, 1), (96, 1024, 514, 1), (1024, 1024, 100, 1), (256, 256, 130, 9), (3072, 1024, 514, 1)]:
        ks, unit = wgrad_ksplit(rows, cols, red, taps)
        assert ks >= 1 and unit >= 64
        padded = (red + unit - 1) // unit * unit
        slice_len = padded // ks
        if rows % 256 == 0 and cols % 256 == 0:
            assert unit == 128 * ks and slice_len % 128 == 0                       # even number of 64-wide K tiles per slice
            assert (rows // 256) * (cols // 256) * taps * ks <= 384 or ks == 1     # about one workgroup per CU
        else:
            assert ks >= 2 and unit == 64 * ks and slice_len % 64 == 0
            tiles = -(-rows // 128) * -(-cols // 128) * taps
            assert tiles * ks <= 768 or ks == 2
        assert padded - red < unit                                                 # never more than one unit of zero padding
