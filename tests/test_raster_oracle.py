"""Pins the CPU rasterizer oracle (oracle/raster_ref.c) -- CPU only, runs in the `-m "not gpu"` suite.

The reference rasterizer is not in /root/reference (requirements.txt:17), so the oracle is pinned by
(i) analytic known-answer tests, (ii) an independent float64 PyTorch restatement (oracle/raster_torch.py),
(iii) autograd + finite differences for the backward incl. the camera twist.
"""
import math

import numpy as np
import pytest
import torch

from oracle import raster_ref as rr
from oracle import raster_torch as rt

K09 = np.array([[0.9, 0, 0.5], [0, 0.9, 0.5], [0, 0, 1]], np.float32)


def _cam(E=None, K=K09, near=0.01, far=100.0):
    E = np.eye(4, dtype=np.float32) if E is None else np.asarray(E, np.float32)
    return rr.make_cameras(E[None], K[None], np.array([near], np.float32), np.array([far], np.float32))[0]


def _iso(P, s2):
    c = np.zeros((P, 6), np.float32)
    c[:, 0] = c[:, 3] = c[:, 5] = s2
    return c


def test_single_isotropic_gaussian_known_answer():
    # One Gaussian on the optical axis at z=2, sigma3d = 0.02, flat colour (band 0 only), opacity 0.8.
    W = H = 64
    cam = _cam()
    sig = 0.02
    sh = np.zeros((1, 25, 3), np.float32)
    sh[0, 0] = (np.array([0.9, 0.4, 0.1], np.float32) - 0.5) / 0.28209479177387814
    o = rr.rasterize_forward(cam, W, H, np.zeros(3, np.float32), np.array([[0, 0, 2.0]], np.float32), _iso(1, sig * sig), sh,
                             np.array([0.8], np.float32))
    # projection: centre of the image is pixel coordinate (W-1)/2
    assert np.allclose(o["xy"][0], [(W - 1) / 2, (H - 1) / 2], atol=1e-4)
    f = W / (2 * cam.tanfovx)
    assert abs(cam.tanfovx - 0.5 / 0.9) < 1e-6
    var2d = (f * sig / 2.0) ** 2 + 0.3
    assert np.allclose(o["conic_opacity"][0], [1 / var2d, 0, 1 / var2d, 0.8], rtol=1e-5, atol=1e-6)
    assert o["radii"][0] == math.ceil(3 * math.sqrt(var2d))
    assert np.allclose(o["rgb"][0], [0.9, 0.4, 0.1], atol=1e-6)
    # pixel (32,32) is at offset (0.5,0.5) from the centre
    d2 = 0.5
    alpha = 0.8 * math.exp(-0.5 * d2 / var2d)
    assert np.allclose(o["color"][:, 32, 32], np.array([0.9, 0.4, 0.1]) * alpha, rtol=1e-5)
    assert np.isclose(o["depth"][32, 32], 2.0 * alpha, rtol=1e-5)
    assert np.isclose(o["opacity"][32, 32], alpha, rtol=1e-5)
    assert o["n_contrib"][32, 32] == 1
    # a far-away pixel in an untouched tile sees nothing
    assert o["color"][:, 0, 0].sum() == 0 and o["n_contrib"][0, 0] == 0
    # tiles touched: centre +- radius straddles the 4 centre tiles
    assert o["tiles_touched"][0] == 4 and o["R"] == 4


def test_alpha_clamp_and_background():
    W = H = 32
    cam = _cam()
    sh = np.zeros((1, 25, 3), np.float32)  # rgb = 0.5
    o = rr.rasterize_forward(cam, W, H, np.array([1.0, 0.0, 0.5], np.float32), np.array([[0, 0, 1.0]], np.float32),
                             _iso(1, 0.5 ** 2), sh, np.array([1.0], np.float32))
    # huge Gaussian, opacity 1: alpha clamps to 0.99 everywhere near the centre
    a = 0.99
    assert np.allclose(o["color"][:, 16, 16], 0.5 * a + (1 - a) * np.array([1.0, 0.0, 0.5]), atol=2e-3)
    assert np.isclose(o["final_T"][16, 16], 0.01, atol=2e-3)


def test_two_gaussians_depth_order_and_ties():
    W = H = 32
    cam = _cam()
    sh = np.zeros((3, 25, 3), np.float32)
    for i, c in enumerate([(1, 0, 0), (0, 1, 0), (0, 0, 1)]):
        sh[i, 0] = (np.array(c, np.float32) - 0.5) / 0.28209479177387814
    means = np.array([[0, 0, 3.0], [0, 0, 2.0], [0, 0, 2.0]], np.float32)  # idx1 and idx2 tie in depth
    op = np.array([0.5, 0.5, 0.5], np.float32)
    o = rr.rasterize_forward(cam, W, H, np.zeros(3, np.float32), means, _iso(3, 0.2 ** 2), sh, op)
    # sorted order in the centre tile: 1, 2 (tie -> index order), then 0
    t = 1 * 2 + 1
    r0, r1 = o["ranges"][t]
    assert list(o["point_list"][r0:r1]) == [1, 2, 0]
    a = [min(0.99, 0.5 * math.exp(-0.5 * 0.5 * o["conic_opacity"][i, 0])) for i in (1, 2, 0)]
    exp_col = np.array([0.0, 1.0, 0.0]) * a[0] + np.array([0, 0, 1.0]) * a[1] * (1 - a[0]) + np.array([1.0, 0, 0]) * a[2] * (1 - a[0]) * (1 - a[1])
    assert np.allclose(o["color"][:, 16, 16], exp_col, rtol=1e-5)


def test_near_cull_and_empty():
    W = H = 32
    cam = _cam()
    sh = np.zeros((2, 25, 3), np.float32)
    means = np.array([[0, 0, 0.2], [0, 0, -1.0]], np.float32)  # z == 0.2 is culled (<=), behind camera culled
    o = rr.rasterize_forward(cam, W, H, np.array([0.1, 0.2, 0.3], np.float32), means, _iso(2, 0.01), sh, np.ones(2, np.float32))
    assert o["R"] == 0 and (o["radii"] == 0).all()
    assert np.allclose(o["color"], np.array([0.1, 0.2, 0.3], np.float32)[:, None, None])
    assert (o["opacity"] == 0).all() and (o["depth"] == 0).all()
    # P = 0
    o = rr.rasterize_forward(cam, W, H, np.zeros(3, np.float32), np.zeros((0, 3), np.float32), np.zeros((0, 6), np.float32),
                             np.zeros((0, 25, 3), np.float32), np.zeros(0, np.float32))
    assert o["R"] == 0 and o["color"].sum() == 0


def _random_small(P=48, seed=0, W=48, H=32):
    rng = np.random.default_rng(seed)
    means = np.stack([rng.uniform(-0.9, 0.9, P), rng.uniform(-0.6, 0.6, P), rng.uniform(1.5, 4.0, P)], -1)
    A = rng.standard_normal((P, 3, 3)) * 0.08
    cov = A @ A.transpose(0, 2, 1) + 1e-4 * np.eye(3)
    sh = rng.standard_normal((P, 25, 3)) * rr.SH_MASK[None, :, None]
    sh[:, 0] = rng.standard_normal((P, 3)) * 0.7
    op = rng.uniform(0.2, 0.95, P)
    yaw = 0.1
    E = np.eye(4)
    E[:3, :3] = [[math.cos(yaw), 0, math.sin(yaw)], [0, 1, 0], [-math.sin(yaw), 0, math.cos(yaw)]]
    E[:3, 3] = [0.1, -0.05, 0.02]
    return means, cov, sh, op, E


def _torch_args(means, cov, sh, op, E, dt=torch.float64, grad=False):
    t = lambda a: torch.tensor(a, dtype=dt, requires_grad=grad)
    return t(means), t(cov), t(sh), t(op), torch.tensor(E, dtype=dt), torch.tensor(K09, dtype=dt)


@pytest.mark.parametrize("seed", [0, 1])
def test_c_oracle_matches_torch_forward(seed):
    W, H = 48, 32
    means, cov, sh, op, E = _random_small(seed=seed, W=W, H=H)
    bg = np.array([0.2, 0.1, 0.3], np.float32)
    cam = _cam(E)
    o = rr.rasterize_forward(cam, W, H, bg, means, rr.cov6(cov.astype(np.float32)), sh, op)
    tm, tc, ts, to, tE, tK = _torch_args(means, cov, sh, op, E)
    color, depth, opac, aux = rt.rasterize(tm, tc, ts, to, tE, tK, 0.01, 100.0, H, W, bg, return_aux=True)
    vis = aux["visible"].numpy()
    assert (o["radii"] > 0).tolist() == vis.tolist()
    assert np.array_equal(o["radii"][vis], aux["radius"].numpy()[vis].astype(np.int32))
    assert np.array_equal(o["rect"][vis], aux["rect"].numpy()[vis].astype(np.int32))
    assert np.allclose(o["xy"][vis], aux["pix"].numpy()[vis], atol=2e-4)
    assert np.allclose(o["conic_opacity"][vis, :3], aux["conic"].numpy()[vis], rtol=2e-4, atol=1e-6)
    assert np.allclose(o["rgb"][vis], aux["rgb"].numpy()[vis], atol=2e-5)
    assert np.allclose(o["color"], color.numpy(), atol=2e-5)
    assert np.allclose(o["depth"], depth.numpy(), atol=1e-4)
    assert np.allclose(o["opacity"], opac.numpy(), atol=2e-5)
    assert np.array_equal(o["n_touched"], aux["n_touched"].numpy())
    assert o["R"] == o["tiles_touched"].sum()


def test_c_oracle_backward_matches_autograd():
    W, H = 48, 32
    means, cov, sh, op, E = _random_small(seed=3, W=W, H=H)
    bg = np.array([0.2, 0.1, 0.3], np.float32)
    rng = np.random.default_rng(5)
    gC = rng.standard_normal((3, H, W)).astype(np.float32)
    gD = rng.standard_normal((H, W)).astype(np.float32) * 0.3
    cam = _cam(E)
    c6 = rr.cov6(cov.astype(np.float32))
    fwd = rr.rasterize_forward(cam, W, H, bg, means, c6, sh, op)
    bwd = rr.rasterize_backward(cam, W, H, bg, means, c6, sh, op, fwd, gC, gD)

    tm, tc, ts, to, tE, tK = _torch_args(means, cov, sh, op, E, grad=True)
    tau = torch.zeros(6, dtype=torch.float64, requires_grad=True)
    color, depth, _ = rt.rasterize(tm, tc, ts, to, tE, tK, 0.01, 100.0, H, W, bg, tau=tau)
    loss = (color * torch.tensor(gC, dtype=torch.float64)).sum() + (depth * torch.tensor(gD, dtype=torch.float64)).sum()
    loss.backward()

    def close(a, b, name, rtol=2e-3):
        a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
        scale = np.abs(b).max() + 1e-12
        assert np.abs(a - b).max() <= rtol * scale, (name, np.abs(a - b).max(), scale)

    close(bwd["means3D"], tm.grad.numpy(), "means3D")
    close(bwd["opacities"], to.grad.numpy(), "opacity")
    g33 = tc.grad.numpy()  # autograd treats the 9 entries independently: fold to the 6 unique parameters
    g6 = np.stack([g33[:, 0, 0], g33[:, 0, 1] + g33[:, 1, 0], g33[:, 0, 2] + g33[:, 2, 0], g33[:, 1, 1],
                   g33[:, 1, 2] + g33[:, 2, 1], g33[:, 2, 2]], -1)
    close(bwd["cov3D"], g6, "cov3D")
    close(bwd["shs"], ts.grad.numpy(), "shs")
    assert np.abs(bwd["shs"][:, 16:]).max() == 0  # band 4 never read -> zero gradient
    close(bwd["tau"], tau.grad.numpy(), "tau", rtol=5e-3)


def test_single_isotropic_gaussian_backward_known_answer():
    """Closed-form gradients of ONE isotropic Gaussian on the optical axis -- the backward counterpart of
    test_single_isotropic_gaussian_known_answer; the rasterizer parity stays UNPINNED against upstream (no source in the reference tree),
    these vectors pin the oracle's backward against SURVEY Appendix B.5 written out by hand:
      C_r(p) = c_r alpha,  alpha = o G,  G = exp(-(dx^2 + dy^2) / (2 v)),  v = (f sigma / Z)^2 + 0.3   [B.1 dilation, B.3 alpha]
      dL/do = c_r G                       [B.5: dL/dopacity += G dL/dalpha, dL/dalpha = c T dL/dC]
      dL/dSH[0, r] = C0 alpha             [B.5: dL/drgb += alpha T dL/dC; band 0 = C0 * coefficient + 0.5]
      dL/dx_c = c_r o G (-dx / v), dx = x_c - x_p   -> means2D = dL/dx_c * W/2   [B.5: dG/dDelta (W/2, H/2)]
      dL/dX = dL/dx_c f / Z  (x_c = f X / Z + (W-1)/2);  dL/dZ = dalpha/dv dv/dZ,  dv/dZ = -2 (f sigma)^2 / Z^3
      dL/dSigma_xx = c_r o G dx^2 / (2 v^2) (f/Z)^2,  dL/dSigma_xy = c_r o G dx dy / v^2 (f/Z)^2   (J's third column vanishes on the axis)
      dL/dtau = (g, p x ... ): left perturbation T' = Exp(tau) T: dp = rho + theta x p, p = (0, 0, Z) -> (gX, gY, gZ, -Z gY, Z gX, 0)."""
    W = H = 64
    cam = _cam()
    sig, Z, o_, c = 0.05, 2.0, 0.8, np.array([0.9, 0.4, 0.1])
    C0 = 0.28209479177387814
    sh = np.zeros((1, 25, 3), np.float32)
    sh[0, 0] = (c.astype(np.float32) - 0.5) / C0
    means = np.array([[0, 0, Z]], np.float32)
    cov = _iso(1, sig * sig)
    op = np.array([o_], np.float32)
    bg = np.zeros(3, np.float32)
    fwd = rr.rasterize_forward(cam, W, H, bg, means, cov, sh, op)
    px, py = 33, 31                                  # pixel offsets from the projected centre (31.5, 31.5): dx = x_c - x_p = -1.5, dy = +0.5
    gC = np.zeros((3, H, W), np.float32)
    gC[0, py, px] = 1.0                              # L = C_r(px, py)
    bwd = rr.rasterize_backward(cam, W, H, bg, means, cov, sh, op, fwd, gC, np.zeros((H, W), np.float32))
    f = W / (2 * cam.tanfovx)
    v = (f * sig / Z) ** 2 + 0.3
    dx, dy = 31.5 - px, 31.5 - py
    G = math.exp(-0.5 * (dx * dx + dy * dy) / v)
    cr = float(c[0])
    assert np.isclose(fwd["color"][0, py, px], cr * o_ * G, rtol=1e-5)
    rel = lambda a, b: abs(a - b) <= 2e-5 * max(abs(b), 1e-12) + 1e-9
    assert rel(bwd["opacities"][0], cr * G)
    # colour = 0.5 + sum_k SH_k Y_k(dir), dir = (0, 0, 1): dL/dSH_k = Y_k(dir) alpha for the red channel, nothing for green / blue;
    # Y_0 = C0, band 1 = (-C1 y, C1 z, -C1 x) -> only its z term, band 4 is never read (B.1)
    assert rel(bwd["shs"][0, 0, 0], C0 * o_ * G) and np.abs(bwd["shs"][0, :, 1:]).max() == 0
    assert rel(bwd["shs"][0, 2, 0], 0.4886025119029199 * o_ * G) and bwd["shs"][0, 1, 0] == 0 and bwd["shs"][0, 3, 0] == 0
    assert np.abs(bwd["shs"][0, 16:]).max() == 0
    gxc, gyc = cr * o_ * G * (-dx / v), cr * o_ * G * (-dy / v)
    assert rel(bwd["means2D"][0, 0], gxc * W / 2) and rel(bwd["means2D"][0, 1], gyc * H / 2)
    gX, gY = gxc * f / Z, gyc * f / Z
    gZ = cr * o_ * G * (0.5 * (dx * dx + dy * dy) / v ** 2) * (-2 * (f * sig) ** 2 / Z ** 3)
    assert rel(bwd["means3D"][0, 0], gX) and rel(bwd["means3D"][0, 1], gY) and rel(bwd["means3D"][0, 2], gZ)
    a = (f / Z) ** 2
    want6 = [cr * o_ * G * dx * dx / (2 * v * v) * a, cr * o_ * G * dx * dy / (v * v) * a, 0.0, cr * o_ * G * dy * dy / (2 * v * v) * a, 0.0, 0.0]
    for k in range(6):
        assert rel(bwd["cov3D"][0, k], want6[k]) or abs(bwd["cov3D"][0, k] - want6[k]) <= 1e-7 * abs(want6[0]), (k, bwd["cov3D"][0], want6)
    want_tau = [gX, gY, gZ, -Z * gY, Z * gX, 0.0]
    for k in range(6):
        assert abs(bwd["tau"][k] - want_tau[k]) <= 3e-5 * max(abs(t) for t in want_tau), (k, bwd["tau"], want_tau)


def test_autograd_tau_matches_finite_differences():
    # pins the twist convention itself: T_cw' = Exp(tau) T_cw, all of view/proj/campos derived from it
    W, H = 32, 32
    means, cov, sh, op, E = _random_small(P=12, seed=7, W=W, H=H)
    tm, tc, ts, to, tE, tK = _torch_args(means, cov, sh, op, E)
    wgt = torch.tensor(np.random.default_rng(1).standard_normal((3, H, W)))

    def f(tau):
        c, d, _ = rt.rasterize(tm, tc, ts, to, tE, tK, 0.01, 100.0, H, W, (0.1, 0.2, 0.3), tau=tau)
        return (c * wgt).sum() + 0.1 * d.sum()

    tau = torch.zeros(6, dtype=torch.float64, requires_grad=True)
    f(tau).backward()
    g = tau.grad.numpy()
    eps = 1e-6
    for k in range(6):
        e = torch.zeros(6, dtype=torch.float64); e[k] = eps
        fd = (f(e) - f(-e)).item() / (2 * eps)
        assert abs(fd - g[k]) <= 2e-4 * (abs(g).max() + 1e-9), (k, fd, g[k])


def test_synthetic_scene_statistics():
    # guards the config-3 scene generator: R ~ 1.5-2.5 P, most Gaussians visible
    sc = rr.synthetic_scene(V=2, res=64, Vt=2, seed=0)
    outs = rr.render_views(sc, res=64)
    P = sc["means"].shape[0]
    for o in outs:
        assert 1.0 * P < o["R"] < 8.0 * P
        assert (o["radii"] > 0).mean() > 0.5
        assert 0.3 < o["color"].mean() < 0.7
