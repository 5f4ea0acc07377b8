"""Harness for the callers either side of the hot path.

What is here runs on the HIP rasterizer through `DecoderSplattingCUDA` and is what the reference's evaluation and demo
scripts do around the encoder/decoder:
  * `mse_loss`                — LossMse.forward, src/loss/loss_mse.py:23-31
  * `compute_psnr`            — src/evaluation/metrics.py:21-29
  * `se3_exp`, `update_pose`  — src/misc/cam_utils.py:59-142 (batched, no per-camera Python loop)
  * `align_poses`             — ModelWrapper.test_step_align, src/model/model_wrapper.py:442-513
  * `export_ply`              — src/model/ply_export.py:31-90 (own binary writer, no plyfile dependency)
  * `export_transforms`       — src/model/model_wrapper.py:390-400 (transforms.json)
  * `load_images`             — demo.py:75-132 (resize short side to 256, centre crop, normalise to [-1, 1])
  * `configure_optimizer`, `training_step` — ModelWrapper.configure_optimizers / training_step (model_wrapper.py:884-951,
    184-321): AdamW(lr, wd 0.05, betas 0.9/0.95) with the backbone-lr multiplier, encoder -> rasterizer -> MSE ->
    backward on the HIP kernels (vicasplat_amd.autograd) -> optional gradient all-reduce -> clip 0.5 -> step.
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import torch
from torch import Tensor

from .model.types import Gaussians


def mse_loss(color: Tensor, target: Tensor, weight: float = 1.0) -> Tensor:
    return weight * ((color - target) ** 2).mean()


@torch.no_grad()
def compute_psnr(ground_truth: Tensor, predicted: Tensor) -> Tensor:
    gt = ground_truth.clip(0, 1)
    pr = predicted.clip(0, 1)
    return -10 * ((gt - pr) ** 2).flatten(1).mean(1).log10()


def _skew(w: Tensor) -> Tensor:
    z = torch.zeros_like(w[..., 0])
    return torch.stack([z, -w[..., 2], w[..., 1], w[..., 2], z, -w[..., 0], -w[..., 1], w[..., 0], z], -1).unflatten(-1, (3, 3))


def se3_exp(tau: Tensor) -> Tensor:
    """tau [..., 6] = (rho, theta) -> 4x4; small-angle series below 1e-5 rad exactly as cam_utils.py:70-115."""
    rho, theta = tau[..., :3], tau[..., 3:]
    W = _skew(theta)
    W2 = W @ W
    angle = theta.norm(dim=-1)[..., None, None]
    small = angle < 1e-5
    a = torch.where(small, torch.ones_like(angle), angle)
    eye = torch.eye(3, dtype=tau.dtype, device=tau.device).expand_as(W)
    Rm = torch.where(small, eye + W + 0.5 * W2, eye + (torch.sin(a) / a) * W + ((1 - torch.cos(a)) / a**2) * W2)
    Vm = torch.where(small, eye + 0.5 * W + W2 / 6.0, eye + W * ((1 - torch.cos(a)) / a**2) + W2 * ((a - torch.sin(a)) / a**3))
    T = torch.zeros(*tau.shape[:-1], 4, 4, dtype=tau.dtype, device=tau.device)
    T[..., :3, :3] = Rm
    T[..., :3, 3] = (Vm @ rho[..., None])[..., 0]
    T[..., 3, 3] = 1
    return T


def update_pose(cam_trans_delta: Tensor, cam_rot_delta: Tensor, extrinsics: Tensor) -> Tensor:
    """c2w' = (Exp([trans, rot]) · w2c)^-1   (cam_utils.py:118-137)."""
    tau = torch.cat([cam_trans_delta, cam_rot_delta], dim=-1)
    return (se3_exp(tau) @ extrinsics.inverse()).inverse()


def align_poses(decoder, gaussians, target_image: Tensor, extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor,
                steps: int = 100, rot_lr: float = 0.005, trans_lr: float = 0.005, mse_weight: float = 1.0,
                return_history: bool = False):
    """Test-time pose alignment: Adam on per-camera twist deltas through the rasterizer's camera-Jacobian backward,
    folding the delta into the extrinsics after every step (model_wrapper.py:442-513).
    target_image [b,v,3,h,w]; extrinsics [b,v,4,4] c2w. Returns the refined extrinsics (and the loss history)."""
    b, v, _, h, w = target_image.shape
    dev = target_image.device
    gaussians = Gaussians(gaussians.means.detach(), gaussians.covariances.detach(), gaussians.harmonics.detach(),
                          gaussians.opacities.detach())
    rot = torch.nn.Parameter(torch.zeros(b, v, 3, device=dev))
    trans = torch.nn.Parameter(torch.zeros(b, v, 3, device=dev))
    opt = torch.optim.Adam([{"params": [rot], "lr": rot_lr}, {"params": [trans], "lr": trans_lr}])
    extrinsics = extrinsics.clone()
    history = []
    with torch.enable_grad():
        for _ in range(steps):
            opt.zero_grad()
            out = decoder.forward(gaussians, extrinsics, intrinsics, near, far, (h, w), cam_rot_delta=rot, cam_trans_delta=trans)
            loss = mse_loss(out.color, target_image, mse_weight)
            loss.backward()
            history.append(loss.detach())
            with torch.no_grad():
                opt.step()
                extrinsics = update_pose(trans.flatten(0, 1), rot.flatten(0, 1), extrinsics.flatten(0, 1)).unflatten(0, (b, v))
                rot.zero_()
                trans.zero_()
    return (extrinsics, torch.stack(history)) if return_history else extrinsics


def _ply_attributes(num_rest: int) -> list[str]:
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
    names += [f"f_rest_{i}" for i in range(num_rest)]
    names += ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    return names


def export_ply(means: Tensor, scales: Tensor, rotations: Tensor, harmonics: Tensor, opacities: Tensor, path,
               save_sh_dc_only: bool = False) -> int:
    """3DGS-compatible binary PLY: pruned at opacity < 0.005, sorted by opacity (descending), opacity as a logit, scales
    as logs, rotation xyzw -> unit wxyz (ply_export.py:31-90). Returns the number of vertices written."""
    mask = opacities >= 0.005
    op, idx = torch.sort(opacities[mask], descending=True)
    means, scales, rotations, harmonics = (t[mask][idx] for t in (means, scales, rotations, harmonics))
    q = rotations / rotations.norm(dim=-1, keepdim=True)
    q = torch.cat([q[:, 3:], q[:, :3]], dim=-1)
    f_dc = harmonics[..., 0]
    f_rest = harmonics[..., 1:].flatten(1)
    cols = [means, torch.zeros_like(means), f_dc]
    if not save_sh_dc_only:
        cols.append(f_rest)
    cols += [torch.log(op / (1 - op))[:, None], scales.log(), q]
    table = torch.cat([c.detach().float().cpu() for c in cols], dim=1).numpy().astype("<f4")
    names = _ply_attributes(0 if save_sh_dc_only else f_rest.shape[1])
    assert table.shape[1] == len(names)
    path = Path(path)
    path.parent.mkdir(exist_ok=True, parents=True)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % table.shape[0]
    header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(table).tobytes())
    return table.shape[0]


def read_ply(path) -> dict[str, np.ndarray]:
    """Reader for the files `export_ply` writes (round-trip tests)."""
    with open(path, "rb") as f:
        names, n = [], 0
        while True:
            line = f.readline().decode("ascii").strip()
            if line.startswith("element vertex"):
                n = int(line.split()[-1])
            elif line.startswith("property float"):
                names.append(line.split()[-1])
            elif line == "end_header":
                break
        table = np.frombuffer(f.read(), dtype="<f4").reshape(n, len(names))
    return {k: table[:, i] for i, k in enumerate(names)}


def export_transforms(extrinsics: Tensor, path, file_names: list[str] | None = None) -> None:
    """transforms.json: a list of `{file_path, transform_matrix}` (4x4 c2w) per context frame (model_wrapper.py:390-400)."""
    ext = extrinsics.detach().float().cpu().reshape(-1, 4, 4)
    frames = [{"file_path": file_names[i] if file_names else f"context/{i:0>6}.png", "transform_matrix": ext[i].tolist()}
              for i in range(ext.shape[0])]
    path = Path(path)
    path.parent.mkdir(exist_ok=True, parents=True)
    with open(path, "w") as f:
        json.dump(frames, f, indent=4)


_IMAGE_EXT = (".jpg", ".jpeg", ".png")


def load_images(folder_or_list, size: int = 256) -> Tensor:
    """Demo pre-processing (demo.py:75-132): EXIF-upright RGB, short side resized to `size` (Lanczos when shrinking,
    bicubic when enlarging), centre square crop, then (x/255 - 0.5) / 0.5.  Returns [V, 3, size, size] f32, files in
    name order.  Host-side (PIL); the frames go to the GPU with the batch."""
    import os
    from PIL import Image, ImageOps
    if isinstance(folder_or_list, (str, os.PathLike)):
        root = os.fspath(folder_or_list)
        paths = [os.path.join(root, f) for f in sorted(os.listdir(root))]
    else:
        paths = sorted((os.fspath(f) for f in folder_or_list), key=lambda f: f.split("/")[-1])
    frames = []
    for path in paths:
        if not path.lower().endswith(_IMAGE_EXT):
            continue
        img = ImageOps.exif_transpose(Image.open(path)).convert("RGB")
        w, h = img.size
        long_edge = round(size * max(w / h, h / w))   # short side -> size
        big = max(w, h)
        interp = Image.LANCZOS if big > long_edge else Image.BICUBIC
        img = img.resize((int(round(w * long_edge / big)), int(round(h * long_edge / big))), interp)
        w, h = img.size
        cx, cy = w // 2, h // 2
        half = min(cx, cy)
        img = img.crop((cx - half, cy - half, cx + half, cy + half))
        x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0
        frames.append((x - 0.5) / 0.5)
    if not frames:
        raise FileNotFoundError(f"no .jpg/.jpeg/.png images in {folder_or_list!r}")
    return torch.stack(frames, 0)


def configure_optimizer(encoder, lr: float = 4e-5, backbone_lr_multiplier: float = 0.25, new_param_keywords=("camera", "mlp_cam", "cam_norm", "modulation"),
                        weight_decay: float = 0.05, warm_up_steps: int = 0):
    """AdamW with the reference's two learning-rate groups (model_wrapper.py:884-951): parameters whose name contains one of
    `new_param_keywords` train at `lr`, the pretrained rest at `lr * backbone_lr_multiplier`; linear warm-up as the reference."""
    new, old = [], []
    for name, p in encoder.named_parameters():
        if p.requires_grad:
            (new if any(k in name for k in new_param_keywords) else old).append(p)
    groups = [dict(params=new, lr=lr), dict(params=old, lr=lr * backbone_lr_multiplier)] if new else [dict(params=old, lr=lr)]
    kw = dict(lr=lr, weight_decay=weight_decay, betas=(0.9, 0.95))
    opt = None
    if old and old[0].is_cuda:
        try:   # one multi-tensor kernel per step instead of ~10 foreach passes over the 578 M parameters
            opt = torch.optim.AdamW(groups, fused=True, **kw)
        except (RuntimeError, TypeError, ValueError):
            opt = None
    if opt is None:
        opt = torch.optim.AdamW(groups, **kw)
    sched = torch.optim.lr_scheduler.LinearLR(opt, 1 / warm_up_steps, 1, total_iters=warm_up_steps) if warm_up_steps > 0 else None
    return opt, sched


def training_step(encoder, decoder, batch: dict, optimizer, *, scheduler=None, compute_dtype: torch.dtype = torch.float16,
                  loss_scale: float = 1024.0, clip: float = 0.5, mse_weight: float = 1.0, allreduce: bool = False) -> dict:
    """One optimisation step of the reference's photometric objective (training_step, model_wrapper.py:184-321, with the MSE
    loss of loss_mse.py; LPIPS / camera losses need external weights / ground-truth poses and are not part of this harness).

    batch: {"context": {"image" [B,V,3,H,W] in [-1,1], "intrinsics" [B,V,3,3]},
            "target":  {"image" [B,Vt,3,H,W] in [0,1], "extrinsics", "intrinsics", "near", "far"}}
    The encoder runs its differentiable HIP forward (train_forward.forward_train), the Gaussians are rendered by the HIP
    rasterizer, and the backward pass runs on the HIP backward kernels.  `loss_scale` keeps 16-bit activation gradients out
    of the f16 underflow range (gradients are unscaled before clipping).  allreduce=True averages gradients over
    torch.distributed ranks with vicasplat_amd.dist.bucketed_allreduce_grads (RCCL over xGMI on a node)."""
    from .model.encoder.train_forward import forward_train
    ctx, tgt = batch["context"], batch["target"]
    optimizer.zero_grad(set_to_none=True)
    out = forward_train(encoder, ctx["image"], ctx["intrinsics"], compute_dtype)
    g = out["gaussians"]
    gs = Gaussians(g["means"].flatten(1, 3), g["covariances"].flatten(1, 3), g["harmonics"].flatten(1, 3), g["opacities"].flatten(1))
    h, w = tgt["image"].shape[-2:]
    render = decoder.forward(gs, tgt["extrinsics"], tgt["intrinsics"], tgt["near"], tgt["far"], (h, w))
    loss = mse_loss(render.color, tgt["image"], mse_weight)
    (loss * loss_scale).backward()
    params = [p for p in encoder.parameters() if p.grad is not None]
    if allreduce:
        from . import dist as vdist
        vdist.bucketed_allreduce_grads(params)
    grads = [p.grad for p in params]
    # unscale and clip in ONE pass over the gradients: norm of the scaled gradients, then a single multiply by
    # min(1, clip / (norm + 1e-6)) / loss_scale (= torch's clip_grad_norm_ on the unscaled gradients)
    gnorm = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(grads))) / loss_scale
    torch._foreach_mul_(grads, torch.clamp(clip / (gnorm + 1e-6), max=1.0) / loss_scale)   # a non-finite norm marks an overflowed step ...
    finite = bool(torch.isfinite(gnorm))
    if finite:                                                # ... which is skipped, as torch.amp's GradScaler would
        optimizer.step()
        if scheduler is not None:
            scheduler.step()
    with torch.no_grad():
        psnr = compute_psnr(tgt["image"].flatten(0, 1), render.color.detach().flatten(0, 1)).mean()
    return dict(loss=loss.detach(), psnr=psnr, grad_norm=gnorm, skipped=not finite, pred_extrins=out["pred_extrins"].detach())
