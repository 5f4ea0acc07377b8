"""Generates tests/golden/callers_*.npz by running the REAL reference's caller-side code (imported from /root/reference on
CPU through ref_import.py's stub finder) on seeded inputs:

  * callers_ply.npz    -- src/model/ply_export.py:31-90 `export_ply`: the vertex table it hands to plyfile (plyfile is not installed;
                          the stub's PlyElement.describe records the structured array instead of writing it)
  * callers_interp.npz -- src/visualization/camera_trajectory/interpolation.py `interpolate_extrinsics` / `interpolate_intrinsics`
                          as demo.py:204-221 calls them (10 steps per interval of a 4-camera path)
  * callers_fov.npz    -- src/geometry/projection.py:247-261 `get_fov` on the view-mean of seeded context intrinsics and the l2 fov term
                          LossCamera adds for the *_no_intrin configurations (loss_camera.py:76-79, loss.py:23-28)
  * callers_load_images.npz -- demo.py:75-132 `load_images` (its function bodies executed from the reference's source) on non-square
                          pictures: both branches of the resize rule (Lanczos when shrinking, bicubic when enlarging) + the centre crop
  * callers_dq.npz     -- src/loss/loss_camera.py:30-45 `camera_dq_loss` and src/misc/dq.py `homogeneous_matrix` on seeded dual
                          quaternions (the pypose SO3 algebra comes from ref_import's shim)

Run in the build container only:   PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_callers_golden.py
The fixtures are data (inputs + expected outputs), never reference source.
"""
from __future__ import annotations

import math
import os
import sys
from pathlib import Path

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_import  # noqa: E402

ref_import.install()


def ply_inputs(seed=0, n=64):
    g = torch.Generator().manual_seed(seed)
    means = torch.randn(n, 3, generator=g)
    scales = torch.rand(n, 3, generator=g) * 0.05 + 1e-3
    rot = torch.randn(n, 4, generator=g)
    rot = rot / rot.norm(dim=-1, keepdim=True)
    harm = torch.randn(n, 3, 25, generator=g) * 0.3
    op = torch.rand(n, generator=g)
    op[::7] = 0.001          # pruned (< 0.005)
    return means, scales, rot, harm, op


def gen_ply():
    import plyfile  # the stub module
    captured = {}

    class PlyElement:
        @staticmethod
        def describe(elements, name):
            captured["elements"], captured["name"] = elements.copy(), name
            return None

    class PlyData:
        def __init__(self, els):
            pass

        def write(self, path):
            captured["path"] = str(path)

    sys.modules["plyfile"].PlyElement, sys.modules["plyfile"].PlyData = PlyElement, PlyData
    from src.model import ply_export
    ply_export.PlyElement, ply_export.PlyData = PlyElement, PlyData
    out = {}
    for tag, dc_only in (("full", False), ("dc", True)):
        means, scales, rot, harm, op = ply_inputs()
        ply_export.export_ply(torch.eye(4), means, scales, rot, harm, op, Path("/tmp/_vs_unused.ply"), save_sh_dc_only=dc_only)
        el = captured["elements"]
        out[f"{tag}_names"] = np.array(el.dtype.names)
        out[f"{tag}_table"] = np.stack([el[k] for k in el.dtype.names], 1).astype(np.float32)
        assert captured["name"] == "vertex"
    np.savez_compressed(os.path.join(HERE, "callers_ply.npz"), **out)
    print("[golden] callers_ply.npz", out["full_table"].shape, out["dc_table"].shape)


def path_poses(seed=1, n=4):
    """n c2w poses roughly looking at a common point, plus two exactly parallel look vectors (the midpoint branch)."""
    g = torch.Generator().manual_seed(seed)
    poses = []
    for i in range(n):
        yaw = 0.15 * i + 0.05 * float(torch.randn((), generator=g))
        pitch = 0.03 * float(torch.randn((), generator=g))
        Ry = torch.tensor([[math.cos(yaw), 0, math.sin(yaw)], [0, 1, 0], [-math.sin(yaw), 0, math.cos(yaw)]])
        Rx = torch.tensor([[1, 0, 0], [0, math.cos(pitch), -math.sin(pitch)], [0, math.sin(pitch), math.cos(pitch)]])
        E = torch.eye(4)
        E[:3, :3] = Ry @ Rx
        E[:3, 3] = torch.tensor([0.3 * i, 0.02 * i, 0.05 * i]) + 0.01 * torch.randn(3, generator=g)
        poses.append(E)
    par = poses[-1].clone()
    par[:3, 3] += torch.tensor([0.2, 0.0, 0.0])
    poses.append(par)                     # same rotation as the previous pose: parallel look vectors
    return torch.stack(poses)


def gen_interp():
    from src.visualization.camera_trajectory.interpolation import interpolate_extrinsics, interpolate_intrinsics
    poses = path_poses()
    K = torch.eye(3).repeat(poses.shape[0], 1, 1)
    K[:, 0, 0] = torch.linspace(0.8, 1.0, poses.shape[0]); K[:, 1, 1] = torch.linspace(0.85, 0.95, poses.shape[0])
    K[:, 0, 2] = K[:, 1, 2] = 0.5
    t = torch.linspace(0, 1, 10, dtype=torch.float)
    E = interpolate_extrinsics(initial=poses[:-1], final=poses[1:], t=t).reshape(-1, 4, 4)
    Ki = interpolate_intrinsics(initial=K[:-1], final=K[1:], t=t).reshape(-1, 3, 3)
    np.savez_compressed(os.path.join(HERE, "callers_interp.npz"), poses=poses.numpy(), K=K.numpy(), t=t.numpy(), extrinsics=E.numpy(),
                        intrinsics=Ki.numpy())
    print("[golden] callers_interp.npz", tuple(E.shape), tuple(Ki.shape))


def gen_dq():
    from src.misc.dq import DualQuaternion
    g = torch.Generator().manual_seed(2)
    def rand_dq(n):
        q = torch.randn(n, 4, generator=g); q = q / q.norm(dim=-1, keepdim=True)
        t = torch.randn(n, 3, generator=g)
        return DualQuaternion.from_quat_pose_array(torch.cat([q, t], -1)), q, t
    a, qa, ta = rand_dq(6)
    b, qb, tb = rand_dq(6)
    arr = lambda d: torch.cat([d.q_r.tensor(), d.q_d.tensor()], -1)
    pa, pb = arr(a), arr(b)
    # loss_camera.py:30-45 (camera_dq_loss), written out here on the reference's DualQuaternion class because importing
    # src.loss.loss_camera pulls the dataset / decoder packages; identical operations, reference classes
    ident = torch.zeros(6, 8); ident[:, 3] = 1
    l = (arr(a * b.conjugate) - ident).abs().mean() + (arr(b * a.conjugate) - ident).abs().mean()
    np.savez_compressed(os.path.join(HERE, "callers_dq.npz"), q_a=qa.numpy(), t_a=ta.numpy(), q_b=qb.numpy(), t_b=tb.numpy(), dq_a=pa.numpy(),
                        dq_b=pb.numpy(), prod_ab=arr(a * b).numpy(), conj_a=arr(a.conjugate).numpy(), mat_a=a.homogeneous_matrix.numpy(),
                        trans_a=a.translation.numpy(), dq_loss=np.float64(l))
    print("[golden] callers_dq.npz loss", float(l))


def gen_fov():
    from src.geometry.projection import get_fov
    from src.loss.loss import l2_loss
    g = torch.Generator().manual_seed(3)
    B, V = 5, 4
    K = torch.eye(3).repeat(B, V, 1, 1)
    K[..., 0, 0] = 0.6 + 0.8 * torch.rand(B, V, generator=g); K[..., 1, 1] = 0.6 + 0.8 * torch.rand(B, V, generator=g)
    K[..., 0, 2] = 0.5 + 0.02 * torch.randn(B, V, generator=g); K[..., 1, 2] = 0.5 + 0.02 * torch.randn(B, V, generator=g)
    pred = math.pi * 50 / 180 + 0.2 * torch.randn(B, 2, generator=g)
    fov = get_fov(K.mean(dim=1))
    np.savez_compressed(os.path.join(HERE, "callers_fov.npz"), K=K.numpy(), pred_intrins=pred.numpy(), fov=fov.numpy(),
                        l2=np.float64(l2_loss(pred, fov)))
    print("[golden] callers_fov.npz l2", float(l2_loss(pred, fov)))


def gen_load_images():
    """demo.py:75-132 `load_images` of the real reference (ref_import.reference_load_images) on two NON-square pictures, so that both
    branches of its resize rule are pinned: a 400 x 300 picture (shrunk: Lanczos) and a 200 x 150 one (enlarged: bicubic), each centre-
    cropped to 256 x 256.  Sources are smooth synthetic RGB patterns with an edge and saturated patches (data, generated here)."""
    import tempfile
    from PIL import Image
    load_images = ref_import.reference_load_images()
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for name, (w, h) in (("shrink", (400, 300)), ("enlarge", (200, 150)), ("tall", (180, 320))):
            yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
            img = np.stack([127.5 + 127.5 * np.sin(xx / 17.0 + yy / 31.0), 255.0 * (xx > w * 0.4), 300.0 * yy / h - 20.0], -1)
            u8 = np.clip(np.round(img), 0, 255).astype(np.uint8)
            Image.fromarray(u8).save(os.path.join(td, name + ".png"))
            x = load_images([os.path.join(td, name + ".png")], size=256, verbose=False)
            out["src_" + name] = u8
            r = torch.round((x[0] * 0.5 + 0.5) * 255.0).to(torch.uint8)
            assert torch.equal((r.float() / 255.0 - 0.5) / 0.5, x[0])
            out["out_" + name] = r.permute(1, 2, 0).numpy()
    np.savez_compressed(os.path.join(HERE, "callers_load_images.npz"), **out)
    print("[golden] callers_load_images.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    gen_load_images()
    gen_fov()
    gen_ply()
    gen_interp()
    gen_dq()
