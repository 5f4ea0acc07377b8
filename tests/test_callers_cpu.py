"""Host-side callers: pose update, PSNR/MSE, PLY and transforms.json writers. CPU only."""
import json

import numpy as np
import pytest
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
        q = q * torch.sign(torch.gather(q, 1, q.abs().argmax(1, keepdim=True)))   # largest component positive (reference: scipy round trip)
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


# ---- fixtures produced by the REAL reference (tests/golden/gen_callers_golden.py imports it on CPU in the build container) ----
import os

_G = os.path.join(os.path.dirname(__file__), "golden")


def _ply_inputs(seed=0, n=64):
    """Same seeded inputs as gen_callers_golden.ply_inputs."""
    g = torch.Generator().manual_seed(seed)
    means = torch.randn(n, 3, generator=g)
    scales = torch.rand(n, 3, generator=g) * 0.05 + 1e-3
    rot = torch.randn(n, 4, generator=g)
    rot = rot / rot.norm(dim=-1, keepdim=True)
    harm = torch.randn(n, 3, 25, generator=g) * 0.3
    op = torch.rand(n, generator=g)
    op[::7] = 0.001
    return means, scales, rot, harm, op


def test_export_ply_matches_the_reference_vertex_table(tmp_path):
    """src/model/ply_export.py:31-90 run on the same seeded Gaussians: attribute names/order and every value of the vertex table
    (pruning, opacity sort, logit / log transforms, quaternion sign + wxyz order), for the full and the DC-only variant."""
    z = np.load(os.path.join(_G, "callers_ply.npz"))
    for tag, dc in (("full", False), ("dc", True)):
        n = callers.export_ply(*_ply_inputs(), tmp_path / f"{tag}.ply", save_sh_dc_only=dc)
        got = callers.read_ply(tmp_path / f"{tag}.ply")
        names = [str(s) for s in z[f"{tag}_names"]]
        assert list(got.keys()) == names and n == z[f"{tag}_table"].shape[0]
        table = np.stack([got[k] for k in names], 1)
        assert np.abs(table - z[f"{tag}_table"]).max() <= 2e-6
        head = open(tmp_path / f"{tag}.ply", "rb").read(96).decode("ascii", "replace")
        assert head.startswith("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\n" % n)


def test_camera_path_interpolation_matches_the_reference():
    """interpolate_extrinsics / interpolate_intrinsics as demo.py:207-221 calls them (incl. one interval with parallel look vectors)."""
    z = np.load(os.path.join(_G, "callers_interp.npz"))
    P, K, t = torch.tensor(z["poses"]), torch.tensor(z["K"]), torch.tensor(z["t"])
    E = callers.interpolate_extrinsics(P[:-1], P[1:], t).reshape(-1, 4, 4)
    assert E.shape == z["extrinsics"].shape and np.abs(E.numpy() - z["extrinsics"]).max() <= 2e-6
    Ki = callers.interpolate_intrinsics(K[:-1], K[1:], t).reshape(-1, 3, 3)
    assert np.abs(Ki.numpy() - z["intrinsics"]).max() <= 1e-7
    assert np.abs(E[0].numpy() - z["poses"][0]).max() <= 2e-6          # every interval starts / ends on its key cameras
    assert np.abs(E[9].numpy() - z["poses"][1]).max() <= 2e-6


def test_dual_quaternion_camera_loss_matches_the_reference():
    """src/misc/dq.py algebra (product, conjugate, translation / homogeneous matrix) and loss_camera.py:30-45 on seeded inputs."""
    from vicasplat_amd.model.encoder.vicasplat import camera_matrix_from_dq_array
    d = np.load(os.path.join(_G, "callers_dq.npz"))
    a, b = torch.tensor(d["dq_a"]), torch.tensor(d["dq_b"])
    assert (callers.dq_mul(a, b) - torch.tensor(d["prod_ab"])).abs().max() <= 1e-6
    assert (callers.dq_conj(a) - torch.tensor(d["conj_a"])).abs().max() == 0
    assert (camera_matrix_from_dq_array(a) - torch.tensor(d["mat_a"])).abs().max() <= 1e-6
    assert abs(float(callers.camera_dq_loss(a, b)) - float(d["dq_loss"])) <= 1e-6
    # (R, t) -> dual quaternion (cam_utils.py:213-218): the inverse of homogeneous_matrix up to the quaternion's sign
    M = torch.tensor(d["mat_a"])
    back = callers.camera_dq_array_from_Rt(M[:, :3, :3], M[:, :3, 3])
    sgn = torch.sign((back[:, :4] * a[:, :4]).sum(-1, keepdim=True))
    assert (back * sgn - a).abs().max() <= 1e-6
    # the loss is zero for a perfect prediction and grows with the pose error
    E = torch.eye(4).repeat(1, 3, 1, 1)
    E[0, 1:, :3, :3] = M[:2, :3, :3]; E[0, 1:, :3, 3] = M[:2, :3, 3]
    perfect = callers.camera_dq_array_from_Rt(E[:, 1:, :3, :3], E[:, 1:, :3, 3])
    assert float(callers.camera_loss(perfect, E)) <= 1e-6
    assert float(callers.camera_loss(perfect + 0.05, E)) > 0.05


def test_camera_loss_fov_term_matches_the_reference():
    """with a fov head (use_intrinsic_embedding=False) LossCamera adds l2(pred_intrins, get_fov(mean_v K)) (loss_camera.py:76-79).
    Fixture from the reference's get_fov / l2_loss (tests/golden/gen_callers_golden.py)."""
    from vicasplat_amd.geometry.projection import get_fov
    z = np.load(os.path.join(_G, "callers_fov.npz"))
    K, pred = torch.tensor(z["K"]), torch.tensor(z["pred_intrins"])
    assert np.abs(get_fov(K.mean(1)).numpy() - z["fov"]).max() <= 2e-6
    B, V = K.shape[:2]
    E = torch.eye(4).repeat(B, V, 1, 1)
    perfect = callers.camera_dq_array_from_Rt(E[:, 1:, :3, :3], E[:, 1:, :3, 3])
    base = float(callers.camera_loss(perfect, E))
    with_fov = float(callers.camera_loss(perfect, E, pred_intrins=pred, context_intrinsics=K))
    assert abs(with_fov - base - float(z["l2"])) <= 1e-6
    assert abs(float(callers.camera_loss(perfect, E, weight=0.5, pred_intrins=pred, context_intrinsics=K)) - 0.5 * (base + float(z["l2"]))) <= 1e-6
    with pytest.raises(ValueError):
        callers.camera_loss(perfect, E, pred_intrins=pred)
    # the fov head receives a gradient through it
    p = pred.clone().requires_grad_(True)
    callers.camera_loss(perfect, E, pred_intrins=p, context_intrinsics=K).backward()
    assert float(p.grad.abs().min()) > 0


def test_matrix_to_quaternion_is_standardised_for_large_angles():
    """rotations beyond ~120 degrees pick a candidate whose real part can come out negative; pytorch3d's
    matrix_to_quaternion ends with standardize_quaternion (real part >= 0), which the sign-sensitive camera losses rely on."""
    g = torch.Generator().manual_seed(4)
    axis = torch.nn.functional.normalize(torch.randn(64, 3, generator=g), dim=-1)
    ang = torch.linspace(2.2, 3.1, 64) * torch.where(torch.arange(64) % 2 == 0, 1.0, -1.0)
    q = torch.cat([torch.cos(ang / 2)[:, None], axis * torch.sin(ang / 2)[:, None]], -1)          # wxyz, w > 0 for |angle| < pi
    w, x, y, z = q.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                     2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    back = callers.matrix_to_quaternion_wxyz(R)
    assert float(back[:, 0].min()) >= 0
    assert float((back - q).abs().max()) <= 1e-5
    dq = callers.camera_dq_array_from_Rt(R, torch.zeros(64, 3))
    assert float(dq[:, 3].min()) >= 0                                                           # xyzw layout: real part last


def test_optimizer_groups_and_schedule_follow_the_reference():
    """model_wrapper.py:884-951: new_param_keywords [gaussian_param_head, intrinsic_encoder] train at lr, the rest at lr * 0.25; warm-up
    then cosine annealing to 0.1 * lr at max_steps (SequentialLR)."""
    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = torch.nn.ModuleDict(dict(intrinsic_encoder=torch.nn.Linear(9, 4), dec=torch.nn.Linear(4, 4)))
            self.gaussian_param_head = torch.nn.Linear(4, 4)
            self.downstream_head1 = torch.nn.Linear(4, 4)
    m = M()
    opt, sched = callers.configure_optimizer(m, lr=4e-5, backbone_lr_multiplier=0.25, warm_up_steps=10, lr_cosine_annealing=True, max_steps=100)
    new = {id(p) for n, p in m.named_parameters() if "gaussian_param_head" in n or "intrinsic_encoder" in n}
    assert {id(p) for p in opt.param_groups[0]["params"]} == new and len(opt.param_groups[1]["params"]) == 4
    ref_opt = torch.optim.AdamW([dict(params=opt.param_groups[0]["params"], lr=4e-5), dict(params=opt.param_groups[1]["params"], lr=1e-5)],
                                lr=4e-5, weight_decay=0.05, betas=(0.9, 0.95))
    warm = torch.optim.lr_scheduler.LinearLR(ref_opt, 1 / 10, 1, total_iters=10)
    cos = torch.optim.lr_scheduler.CosineAnnealingLR(ref_opt, T_max=100, eta_min=4e-5 * 0.1)
    ref = torch.optim.lr_scheduler.SequentialLR(ref_opt, schedulers=[warm, cos], milestones=[10])
    for _ in range(60):
        assert [g["lr"] for g in opt.param_groups] == [g["lr"] for g in ref_opt.param_groups]
        opt.step(); sched.step(); ref_opt.step(); ref.step()
    assert opt.param_groups[0]["lr"] < 4e-5 and opt.defaults["betas"] == (0.9, 0.95) and opt.defaults["weight_decay"] == 0.05
    opt1, _ = callers.configure_optimizer(m, new_param_keywords=None)          # distillation: one group at lr
    assert len(opt1.param_groups) == 1 and opt1.param_groups[0]["lr"] == 4e-5


def test_loss_scaler_backs_off_and_recovers():
    s = callers.LossScaler(1024.0, growth_interval=3)
    s.update(False); assert s.scale == 512.0
    for _ in range(3):
        s.update(True)
    assert s.scale == 1024.0


def _fake_lpips_state_dict(seed=0):
    """Key names and shapes of lpips.LPIPS(net='vgg').state_dict() with seeded random values (structure test only: the real weights
    are not available offline and LossLpips never substitutes them)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for s_, (idxs, ch) in enumerate(zip(callers.LossLpips._SLICES, callers.LossLpips._CHANNELS)):
        for j, li in enumerate(idxs):
            sd[f"net.slice{s_ + 1}.{li}.weight"] = torch.randn(ch[j + 1], ch[j], 3, 3, generator=g) * (2.0 / (9 * ch[j])) ** 0.5
            sd[f"net.slice{s_ + 1}.{li}.bias"] = torch.randn(ch[j + 1], generator=g) * 0.01
        sd[f"lin{s_}.model.1.weight"] = torch.rand(1, ch[-1], 1, 1, generator=g)
    return sd


def test_lpips_structure_and_properties():
    """src/loss/loss_lpips.py:27-54 with the LPIPS-VGG algorithm restated: taps / channels of VGG-16, identity gives 0, symmetric,
    non-negative (non-negative heads), differentiable, gated by apply_after_step, and refuses to run without weights."""
    import pytest
    with pytest.raises(RuntimeError):
        callers.LossLpips(None)
    loss = callers.LossLpips(_fake_lpips_state_dict(), weight=0.05, apply_after_step=3)
    g = torch.Generator().manual_seed(1)
    a = torch.rand(1, 2, 3, 64, 64, generator=g)
    b = torch.rand(1, 2, 3, 64, 64, generator=g)
    taps = loss.features(a.flatten(0, 1))
    assert [t.shape[1] for t in taps] == [64, 128, 256, 512, 512] and [t.shape[-1] for t in taps] == [64, 32, 16, 8, 4]
    assert float(loss(a, b, global_step=0)) == 0.0
    d_ab, d_ba, d_aa = float(loss(a, b, 3)), float(loss(b, a, 3)), float(loss(a, a, 3))
    assert d_aa == 0.0 and d_ab > 0 and abs(d_ab - d_ba) <= 1e-6 * d_ab
    near = float(loss(a, (a + 0.01 * (b - a)).clamp(0, 1), 3))
    assert near < 0.2 * d_ab
    x = a.clone().requires_grad_()
    loss(x, b, 3).backward()
    assert x.grad is not None and torch.isfinite(x.grad).all() and float(x.grad.abs().sum()) > 0


def test_split_class_training_step_defaults_and_operand_geometry():
    """Host logic of the split-class (reference-precision) training step, no GPU: callers.training_step(compute_dtype="split") creates the
    scaler the f16 (hi, lo) gradient operands need (8192, growing to 2^24, halved on overflow -- exact powers of two), keeps the 16-bit
    defaults unchanged, and the border geometry of the tap-fused convolution weight gradient (ops._border_rows, the halo of
    conv3x3_backward_split) matches the kernel contract of vs_transpose_f32 / vs_gemm_wgrad(dtype 4)."""
    from tests.test_distributed_cpu import _ToyDecoder, _ToyEncoder, _toy_forward, _toy_train_batch
    from vicasplat_amd import autograd as A
    from vicasplat_amd import ops
    assert A.act_dtype("split") == torch.float32 and A.act_dtype(torch.float16) == torch.float16 and A.SPLIT == "split"
    torch.manual_seed(0)
    for cdt, init in (("split", 8192.0), (torch.float16, 1024.0), (torch.bfloat16, 1.0)):
        enc = _ToyEncoder()
        opt = torch.optim.AdamW(enc.parameters(), lr=1e-3)
        r = callers.training_step(enc, _ToyDecoder(), _toy_train_batch(), opt, compute_dtype=cdt, forward_fn=_toy_forward)
        sc = opt._vs_loss_scaler
        assert r["loss_scale"] == init and sc.scale == init and not r["skipped"]
        assert sc.max_scale == (2.0 ** 24 if cdt == "split" else 65536.0)
    sc = callers.LossScaler(8192.0, max_scale=2.0 ** 24, growth_interval=1)
    for _ in range(20):
        sc.update(True)
    assert sc.scale == 2.0 ** 24
    sc.update(False)
    assert sc.scale == 2.0 ** 23
    # zero-bordered pixel grid of the tap-fused weight gradient: N images of (H + 2) x (W + 2) pixels; halo >= largest |tap shift| + 1, 16-byte rows
    assert ops._border_rows(2 * 5 * 7, (5, 7), True) == 2 * 7 * 9 and ops._border_rows(70, (5, 7), False) == 70
    for W in (4, 16, 37, 256):
        Wp = W + 2
        halo = (Wp + 2 + 3) // 4 * 4
        shifts = [(ty - 1) * Wp + (tx - 1) for ty in range(3) for tx in range(3)]
        assert halo % 4 == 0 and halo > max(abs(s) for s in shifts)


def test_boundary_grad_scale_leaves_plain_gradients_and_handles_accumulation():
    """autograd.BoundaryGradScale (the Module API's internal power-of-two gradient scale, VicaSplat.forward under autograd): cotangents
    enter the network multiplied by S, parameter gradients leave divided by S when the backward pass ends -- bit-identical to an unscaled
    backward (power of two), also when `.grad` already holds a gradient (micro-batch accumulation) and with unused outputs."""
    from vicasplat_amd import autograd as A
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 4))
    ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 4))
    ref.load_state_dict(net.state_dict())
    seen = []

    class Spy(torch.autograd.Function):      # sits where the encoder's operators sit: records the size of the cotangent it receives
        @staticmethod
        def forward(ctx, t):
            return t.view_as(t)

        @staticmethod
        def backward(ctx, g):
            seen.append(float(g.abs().max()))
            return g

    sc = A.BoundaryGradScale(list(net.parameters()), 4096.0)
    for it in range(2):                      # second iteration accumulates onto the first one's gradients
        x = torch.randn(3, 6)
        y = Spy.apply(net(x))
        a, b, unused = sc.outputs(y[:, :2], y[:, 2:], None)
        assert unused is None and torch.equal(a, y[:, :2])
        (a.square().sum() + b.sum()).backward()
        yr = ref(x)
        (yr[:, :2].square().sum() + yr[:, 2:].sum()).backward()
        for p, q in zip(net.parameters(), ref.parameters()):
            assert torch.equal(p.grad, q.grad), it
    assert seen[0] >= 4096.0 * 0.99         # the network saw scaled cotangents
    with pytest.raises(AssertionError):
        A.BoundaryGradScale(list(net.parameters()), 1000.0)
    # S = 1: pass-through, no node
    one = A.BoundaryGradScale(list(net.parameters()), 1.0)
    t = torch.ones(2, requires_grad=True)
    assert one.outputs(t)[0] is t


def test_boundary_grad_scale_late_unfreeze_input_grads_failed_backward_and_overflow():
    """ADVICE r4 on autograd.BoundaryGradScale: (1) parameters unfrozen after the scaler was built are unscaled too (the set is re-read at
    every backward); (2) non-parameter leaves entering through `inputs()` get PLAIN gradients; (3) a backward that raises does not leave
    the scaler armed; (4) an inf / NaN gradient raises the device-side overflow flag."""
    from vicasplat_amd import autograd as A
    torch.manual_seed(1)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 4))
    ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 4))
    ref.load_state_dict(net.state_dict())
    net[0].requires_grad_(False)
    sc = A.BoundaryGradScale(net, 8192.0)
    x = torch.randn(3, 6)
    sc.outputs(net(x))[0].square().sum().backward()
    assert net[0].weight.grad is None and float(sc.last_overflow) == 0.0
    net[0].requires_grad_(True)                                       # unfrozen AFTER the scaler exists
    net.zero_grad()
    xi = x.clone().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    sc.rearm()
    (xin,) = sc.inputs(xi)
    sc.outputs(net(xin))[0].square().sum().backward()
    ref(xr).square().sum().backward()
    for p, q in zip(net.parameters(), ref.parameters()):
        assert torch.equal(p.grad, q.grad)
    assert torch.equal(xi.grad, xr.grad)                              # (2)

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.view_as(t)

        @staticmethod
        def backward(ctx, g):
            raise ValueError("boom")

    net.zero_grad()
    with pytest.raises(ValueError):
        sc.outputs(Boom.apply(net(x)))[0].sum().backward()
    assert sc._armed                                                   # the failed pass left it armed ...
    sc.rearm()                                                         # ... the next forward clears that (VicaSplat._forward_autograd)
    net.zero_grad(); ref.zero_grad()
    sc.outputs(net(x))[0].square().sum().backward()
    ref(x).square().sum().backward()
    for p, q in zip(net.parameters(), ref.parameters()):
        assert torch.equal(p.grad, q.grad)
    net.zero_grad()
    (sc.outputs(net(x))[0].sum() * float("inf")).backward()            # (4)
    assert float(sc.last_overflow) > 0


def test_split_weight_exponent_cache_policy():
    """autograd.LinearSplitFn._scale_exp (the cached power-of-two scale of a split-class weight pack): an all-zero weight is
    never cached (the zero-initialised pose head would otherwise be packed unscaled for 64 steps), a stacked temporary takes the smallest
    exponent of its source parameters without a host read of its own, and clear_split_caches() / VicaSplat's load_state_dict hook drop
    exponents that new values under the same storage have made stale."""
    from vicasplat_amd import autograd as A
    from vicasplat_amd.model.encoder import default_cfg, get_encoder
    A.clear_split_caches()
    z = torch.nn.Parameter(torch.zeros(8, 32))
    assert A.LinearSplitFn._scale_exp(z) == 0 and not A.LinearSplitFn._exp_cache
    with torch.no_grad():
        z.normal_(0, 0.02)
    e = A.LinearSplitFn._scale_exp(z)
    amax = float(z.detach().abs().max())
    assert 2 ** 13 <= amax * 2.0 ** e < 2 ** 14 and len(A.LinearSplitFn._exp_cache) == 1
    big = torch.nn.Parameter(z.detach() * 16)
    assert A.LinearSplitFn._scale_exp(torch.cat([z, big], 0), z, big) == e - 4          # smallest exponent of the sources
    with torch.no_grad():
        z.mul_(1000.0)                                                                     # new values under the same storage ...
    assert A.LinearSplitFn._scale_exp(z) == e                                              # ... the cache cannot see them
    A.clear_split_caches()
    assert A.LinearSplitFn._scale_exp(z) < e - 8
    # load_state_dict on the encoder clears the cache by itself
    m, _ = get_encoder(default_cfg(enc_depth=1, dec_depth=10, dec_embed_dim=64, dec_num_heads=1, enc_embed_dim=64, enc_num_heads=1))
    w = m.backbone.decoder_embed.weight
    e0 = A.LinearSplitFn._scale_exp(w)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd["backbone.decoder_embed.weight"] *= 256.0
    m.load_state_dict(sd)
    assert A.LinearSplitFn._scale_exp(w) == e0 - 8


def test_load_images_reproduces_the_reference_preprocessing_of_its_example_frames(tmp_path):
    """demo.py:75-132 on the reference's own examples: the fixture holds what the REAL `load_images` returned for
    examples/<scene>/*.png (uint8-exact: the tensors are (u8 / 255 - 0.5) / 0.5); written back as PNGs, callers.load_images must return the
    same tensor, files in name order.  callers_load_images.npz (the reference's `load_images` run on three non-square pictures) pins the
    resize rule of demo.py:62-69 -- Lanczos when shrinking, bicubic when enlarging -- and the centre crop."""
    import os
    from PIL import Image
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "encoder_full_v8_examples.npz"))
    u8 = z["frames_u8"][0]                                   # [8,256,256,3]
    names = [f"{5 + 39 * i:06d}.png" for i in range(8)]     # the reference's file names: 000005.png ... 000278.png
    for nm, fr in zip(reversed(names), reversed(list(u8))):  # written in reverse: the loader must sort by name
        Image.fromarray(fr).save(tmp_path / nm)
    (tmp_path / "notes.txt").write_text("ignored")
    x = callers.load_images(str(tmp_path))
    ref = (torch.from_numpy(u8).permute(0, 3, 1, 2).float() / 255.0 - 0.5) / 0.5
    assert x.shape == (8, 3, 256, 256) and torch.equal(x, ref)
    assert torch.equal(callers.load_images([str(tmp_path / n) for n in names[::-1]]), ref)
    # non-square sources: the reference's own output for a shrunk (Lanczos), an enlarged (bicubic) and a portrait picture
    zl = np.load(os.path.join(os.path.dirname(__file__), "golden", "callers_load_images.npz"))
    for name in ("shrink", "enlarge", "tall"):
        Image.fromarray(zl["src_" + name]).save(tmp_path / (name + ".png"))
        y = callers.load_images([str(tmp_path / (name + ".png"))])
        want = (torch.from_numpy(zl["out_" + name]).permute(2, 0, 1).float() / 255.0 - 0.5) / 0.5
        assert y.shape == (1, 3, 256, 256) and torch.equal(y[0], want), name


def test_state_dict_with_the_confidence_channel_loads_strict_in_both_layouts():
    """a stage-1 checkpoint (predict_conf=true, distill.yaml:24: `downstream_head1.dpt.head.4` has 4 rows) loads
    `strict=True` into a predict_conf model as it is, and into a model WITHOUT the confidence channel through the slicing rule the
    reference applies when it loads such a checkpoint (src/main.py:146-151: rows 0..2 kept)."""
    import dataclasses
    from vicasplat_amd.model.encoder import default_cfg, get_encoder
    tiny = dict(enc_depth=1, dec_depth=10, dec_embed_dim=64, dec_num_heads=1, enc_embed_dim=64, enc_num_heads=1)
    mc, _ = get_encoder(dataclasses.replace(default_cfg(**tiny), predict_conf=True))
    assert mc.predict_confidence and mc.downstream_head1.dpt.head[4].out_channels == 4
    sd4 = {k: v.clone() for k, v in mc.state_dict().items()}
    torch.manual_seed(0)
    sd4["downstream_head1.dpt.head.4.weight"] = torch.randn_like(sd4["downstream_head1.dpt.head.4.weight"])
    sd4["downstream_head1.dpt.head.4.bias"] = torch.randn_like(sd4["downstream_head1.dpt.head.4.bias"])
    mc.load_state_dict(sd4, strict=True)
    m3, _ = get_encoder(default_cfg(**tiny))
    assert not m3.predict_confidence and m3.downstream_head1.dpt.head[4].out_channels == 3
    m3.load_state_dict(sd4, strict=True)                                         # 4 -> 3 inside load_state_dict
    assert torch.equal(m3.downstream_head1.dpt.head[4].weight, sd4["downstream_head1.dpt.head.4.weight"][:3])
    assert torch.equal(m3.downstream_head1.dpt.head[4].bias, sd4["downstream_head1.dpt.head.4.bias"][:3])
    assert sd4["downstream_head1.dpt.head.4.bias"].shape[0] == 4                 # the caller's dict is not modified
    with pytest.raises(RuntimeError):                                            # the other direction is a real mismatch
        mc.load_state_dict(m3.state_dict(), strict=True)


def test_auto_checkpoint_policy_fills_the_device_memory():
    """train_forward.auto_checkpoint_blocks: recompute only as many blocks as the device memory requires -- decoder blocks are released
    first, then encoder blocks; measured anchors on a 288 GB MI355X (split class, 8 views of 256 x 256)."""
    from vicasplat_amd.model.encoder.train_forward import auto_checkpoint_blocks as f
    assert f(8, 1.0, 288.0, 24, 12) == (0, 0) and f(16, 1.0, 288.0, 24, 12) == (0, 0)      # fit without recomputation
    ne, nd = f(24, 1.0, 288.0, 24, 12)
    assert nd == 0 and 6 <= ne <= 12                                                          # config 5's batch: 241 GB at (9, 0)
    assert f(48, 1.0, 288.0, 24, 12) == (24, 12)                                              # does not fit either way: everything checkpointed
    assert f(24, 1.0, 288.0, 24, 12, half=True) == (0, 0)                                     # 16-bit classes hold half
    assert f(24, 1.0, 80.0, 24, 12) == (24, 12)                                               # the reference's 80 GB devices: its all-blocks policy
