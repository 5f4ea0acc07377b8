"""Differentiable dense PyTorch restatement of the rasterizer (small cases only; O(P*H*W)).

TEST INFRASTRUCTURE ONLY.  Second, independent statement of SURVEY.md Appendix B used to pin the C oracle
(oracle/raster_ref.c): autograd on this forward gives reference gradients for B.5 (means3D, cov3D, SH, opacity,
and the camera twist tau for T_cw' = Exp(tau) T_cw), and float64 makes finite differences meaningful.

Conventions follow /root/reference/src/model/decoder/cuda_splatting.py:187-235 (c2w OpenCV extrinsics,
normalised intrinsics, z in [0,1] projection).  Upstream quirks reproduced on purpose:
  * tile rectangle from radius = ceil(3*sqrt(lambda_max)), all 16x16 tiles in the rectangle evaluate the Gaussian;
  * the min(0.99, .) alpha clamp is ignored by the backward (straight-through), and t.x/t.y are treated as
    constants when the 1.3*tanfov clamp is active.
"""
from __future__ import annotations

import math

import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435)


def skew(v):
    z = torch.zeros((), dtype=v.dtype)
    return torch.stack([torch.stack([z, -v[2], v[1]]), torch.stack([v[2], z, -v[0]]), torch.stack([-v[1], v[0], z])])


def se3_exp(tau):
    """tau = (rho, theta) -> 4x4, standard SE(3) exponential (MonoGS / cam_utils.py:59-142 convention)."""
    rho, theta = tau[:3], tau[3:]
    W = skew(theta)
    ang = theta.norm()
    I = torch.eye(3, dtype=tau.dtype)
    if float(ang) < 1e-8:
        R = I + W + 0.5 * W @ W
        V = I + 0.5 * W + W @ W / 6.0
    else:
        R = I + torch.sin(ang) / ang * W + (1 - torch.cos(ang)) / ang ** 2 * W @ W
        V = I + (1 - torch.cos(ang)) / ang ** 2 * W + (ang - torch.sin(ang)) / ang ** 3 * W @ W
    T = torch.eye(4, dtype=tau.dtype)
    T = T.clone()
    T[:3, :3] = R
    T[:3, 3] = V @ rho
    return T


def eval_sh(deg, sh, dirs):
    """sh [P,M,3], dirs [P,3] normalised -> rgb+0.5 (unclamped) [P,3]; bands 0..min(deg,3)."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    r = C0 * sh[:, 0]
    if deg > 0:
        r = r - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = (r + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6]
             + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        r = (r + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10] + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11]
             + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12] + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13]
             + C3[5] * z * (xx - yy) * sh[:, 14] + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return r + 0.5


def rasterize(means, cov33, shs, opac, c2w, K, near, far, H, W, bg, tau=None, sh_degree=4, return_aux=False):
    """means [P,3], cov33 [P,3,3], shs [P,M,3], opac [P]; c2w [4,4]; K normalised [3,3]. dtype of `means`."""
    dt = means.dtype
    P = means.shape[0]
    Kinv = torch.linalg.inv(K)

    def unit(v):
        r = Kinv @ torch.tensor(v, dtype=dt)
        return r / r.norm()

    fov_x = torch.acos((unit([0, .5, 1]) * unit([1, .5, 1])).sum())
    fov_y = torch.acos((unit([.5, 0, 1]) * unit([.5, 1, 1])).sum())
    tfx, tfy = torch.tan(0.5 * fov_x), torch.tan(0.5 * fov_y)
    Pm = torch.zeros(4, 4, dtype=dt)
    Pm[0, 0] = 1 / tfx; Pm[1, 1] = 1 / tfy; Pm[3, 2] = 1
    Pm[2, 2] = far / (far - near); Pm[2, 3] = -(far * near) / (far - near)
    Tcw = torch.linalg.inv(c2w)
    if tau is not None:
        Tcw = se3_exp(tau) @ Tcw
    Rv, tv = Tcw[:3, :3], Tcw[:3, 3]
    campos = -Rv.T @ tv
    fx, fy = W / (2 * tfx), H / (2 * tfy)

    pc = means @ Rv.T + tv
    vz = pc[:, 2]
    hom = torch.cat([pc, torch.ones(P, 1, dtype=dt)], 1) @ Pm.T
    pw = 1.0 / (hom[:, 3] + 1e-7)
    proj = hom[:, :2] * pw[:, None]
    limx, limy = 1.3 * tfx, 1.3 * tfy
    txtz, tytz = pc[:, 0] / vz, pc[:, 1] / vz
    cx = (txtz < -limx) | (txtz > limx)
    cy = (tytz < -limy) | (tytz > limy)
    tx = torch.where(cx, (txtz.clamp(-limx, limx) * vz).detach(), pc[:, 0])
    ty = torch.where(cy, (tytz.clamp(-limy, limy) * vz).detach(), pc[:, 1])
    zero = torch.zeros_like(vz)
    Jac = torch.stack([torch.stack([fx / vz, zero, -fx * tx / vz ** 2], -1),
                       torch.stack([zero, fy / vz, -fy * ty / vz ** 2], -1)], 1)  # [P,2,3]
    M = Jac @ Rv
    cov2 = M @ cov33 @ M.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3; b = cov2[:, 0, 1]; c = cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    A, B, Cc = c / det, -b / det, a / det
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3 * torch.sqrt(lam)).detach()
    pix = ((proj + 1) * torch.tensor([W, H], dtype=dt) - 1) * 0.5
    gx, gy = (W + 15) // 16, (H + 15) // 16

    def ti(v, g):
        return torch.clamp(torch.trunc(v.detach()), 0, g)

    rminx = ti((pix[:, 0] - radius) / 16, gx); rmaxx = ti((pix[:, 0] + radius + 15) / 16, gx)
    rminy = ti((pix[:, 1] - radius) / 16, gy); rmaxy = ti((pix[:, 1] + radius + 15) / 16, gy)
    visible = (vz > 0.2) & (det != 0) & ((rmaxx - rminx) * (rmaxy - rminy) > 0)

    dirs = means - campos
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    rgb = torch.clamp(eval_sh(sh_degree, shs, dirs), min=0.0)

    order = sorted(range(P), key=lambda i: (float(vz[i]), i))
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    tyi, txi = torch.div(ys, 16, rounding_mode="floor"), torch.div(xs, 16, rounding_mode="floor")
    T = torch.ones(H, W, dtype=dt)
    Cimg = torch.zeros(3, H, W, dtype=dt)
    Dimg = torch.zeros(H, W, dtype=dt)
    done = torch.zeros(H, W, dtype=torch.bool)
    n_touched = torch.zeros(P, dtype=torch.int64)
    for i in order:
        if not bool(visible[i]):
            continue
        intile = (txi >= rminx[i]) & (txi < rmaxx[i]) & (tyi >= rminy[i]) & (tyi < rmaxy[i])
        dx, dy = pix[i, 0] - xs, pix[i, 1] - ys
        power = -0.5 * (A[i] * dx * dx + Cc[i] * dy * dy) - B[i] * dx * dy
        G = torch.exp(power)
        raw = opac[i] * G
        alpha = raw + (torch.clamp(raw, max=0.99) - raw).detach()  # straight-through, as upstream's backward
        ok = intile & (power <= 0) & (alpha >= 1.0 / 255.0) & (~done)
        test_T = T * (1 - alpha)
        stop = ok & (test_T < 1e-4)
        done = done | stop
        ok = ok & (~stop)
        w = torch.where(ok, alpha * T, torch.zeros_like(T))
        Cimg = Cimg + rgb[i][:, None, None] * w
        Dimg = Dimg + vz[i] * w
        n_touched[i] = int((ok & (test_T > 0.5)).sum())
        T = torch.where(ok, test_T, T)
    color = Cimg + T * torch.as_tensor(bg, dtype=dt)[:, None, None]
    if return_aux:
        return color, Dimg, 1 - T, dict(radius=radius, visible=visible, n_touched=n_touched, pix=pix, rgb=rgb,
                                        conic=torch.stack([A, B, Cc], -1), depth=vz,
                                        rect=torch.stack([rminx, rminy, rmaxx, rmaxy], -1))
    return color, Dimg, 1 - T
