"""VicaSplat encoder, MI355X-native, with the reference's module API (vicasplat.py:38-290):
`VicaSplat(cfg).forward(context, global_step=0, visualization_dump=None, distill=False,
compute_viewspace_depth=True) -> dict`, attributes `.cfg`, `.backbone.config`, `enable_gradient_checkpointing()`,
`get_data_shim()`, and the state_dict keys of SURVEY.md Appendix C."""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Literal, Optional

import torch
import torch.nn.functional as F
from torch import nn

from ... import ops
from .backbone.backbone_vica import VicaNet
from .common.gaussian_adapter import GaussianAdapterCfg, Gaussians, MyGaussianAdapter
from .encoder import Encoder
from .heads.dpt import PixelwiseTaskWithDPT


@dataclass
class OpacityMappingCfg:
    initial: float
    final: float
    warm_up: int


@dataclass
class VicaSplatCfg:
    name: str
    backbone: dict
    visualizer: object
    gaussian_adapter: GaussianAdapterCfg
    apply_bounds_shim: bool
    opacity_mapping: OpacityMappingCfg
    predict_opacity: bool
    input_mean: tuple = (0.5, 0.5, 0.5)
    input_std: tuple = (0.5, 0.5, 0.5)
    pretrained_weights: str = ""
    gs_center_head_type: str = "dpt"
    gs_param_head_type: str = "dpt_gs"
    predict_conf: bool = False
    camera_type: Literal["dq", "qt"] = "dq"
    # MI355X-side field (not in the reference's schema; a config that omits it gets the default): operand class of the MFMA kernels the
    # module is built with -- "split" (default: f32 activations, products as three f16 MFMAs on (hi, lo) pairs; meets the 1e-4 dB render
    # tolerance against an fp32 evaluation of the reference), "f16" / "bf16" (opt-in fast path, TF32-class products), "f32" (exact-f32 MFMA)
    compute_class: str = "split"


_WARNED = set()


def _warn_once(key, msg):
    if key not in _WARNED:
        _WARNED.add(key)
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


def _clear_split_caches():
    from ... import autograd as A
    A.clear_split_caches()


def quat_mul_xyzw(a, b):
    x1, y1, z1, w1 = a.unbind(-1)
    x2, y2, z2, w2 = b.unbind(-1)
    return torch.stack([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                        w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2], -1)


def camera_matrix_from_dq_array(dq: torch.Tensor) -> torch.Tensor:
    """[...,8] (q_r xyzw | q_d xyzw, q_r unit) -> [...,4,4]: R = rotmat(q_r), t = vec(2 q_d (x) conj(q_r))
    (misc/cam_utils.py:203-207, misc/dq.py:224-262; the quaternion algebra pypose provides there is restated)."""
    qr, qd = dq[..., :4], dq[..., 4:]
    x, y, z, w = qr.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                     2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(*qr.shape[:-1], 3, 3)
    t = quat_mul_xyzw(2.0 * qd, qr * qr.new_tensor([-1.0, -1.0, -1.0, 1.0]))[..., :3]
    M = torch.zeros(*qr.shape[:-1], 4, 4, dtype=dq.dtype, device=dq.device)
    M[..., :3, :3] = R
    M[..., :3, 3] = t
    M[..., 3, 3] = 1
    return M


class VicaSplat(Encoder[VicaSplatCfg]):
    patch_size: int = 16

    def __init__(self, cfg: VicaSplatCfg, weight_dtype: Optional[torch.dtype] = None, device=None,
                 compute_dtype=None) -> None:
        super().__init__(cfg)
        if compute_dtype is None:       # get_encoder(cfg) / `VicaSplat(cfg)` as src/main.py:128 and demo.py:367 build it: the config decides
            compute_dtype = getattr(cfg, "compute_class", "split")
        if cfg.camera_type != "dq" or cfg.gs_center_head_type != "dpt" or cfg.gs_param_head_type != "dpt_gs":
            raise NotImplementedError("only the released configurations (dq camera, dpt + dpt_gs heads) are implemented")
        self.camera_extrinsic_channels = 8
        self.backbone = VicaNet(**dict(cfg.backbone))
        self.gaussian_adapter = MyGaussianAdapter(cfg.gaussian_adapter)
        self.raw_gs_dim = 1 + self.gaussian_adapter.d_in
        # predict_conf (config/experiment/distill.yaml:24, vicasplat.py:75,87-91): a fourth channel on the pts3d head's last 1x1
        # convolution, confidence = 1 + exp(x) (conf_mode ('exp', 1, inf), heads/postprocess.py:17-18,66-75)
        self.predict_confidence = bool(cfg.predict_conf)
        self.downstream_head1 = PixelwiseTaskWithDPT(self.backbone, 3 + int(self.predict_confidence), "regression")
        self.gaussian_param_head = PixelwiseTaskWithDPT(self.backbone, self.raw_gs_dim, "gs_params")
        self.camera_extrinsic_head = nn.Sequential(nn.ReLU(), nn.Linear(self.backbone.config.dec_embed_dim, 8))
        nn.init.zeros_(self.camera_extrinsic_head[1].weight)  # predicts the identity pose at init (vicasplat.py:126-127)
        nn.init.zeros_(self.camera_extrinsic_head[1].bias)
        if self.backbone.config.use_intrinsic_embedding:
            self.camera_intrinsic_head = None
        else:   # the *_no_intrin checkpoints: fov head on camera token 0, initialised to 50 degrees (vicasplat.py:79-82,129-138)
            self.camera_intrinsic_head = nn.Sequential(nn.ReLU(), nn.Linear(self.backbone.config.dec_embed_dim, 2))
            nn.init.zeros_(self.camera_intrinsic_head[1].weight)
            nn.init.constant_(self.camera_intrinsic_head[1].bias, math.pi * 50 / 180)
        self.set_compute_dtype(compute_dtype)
        # new VALUES under the same storage: the cached split-class weight exponents of the training Functions are stale
        self.register_load_state_dict_post_hook(lambda module, incompatible_keys: _clear_split_caches())
        self._register_load_state_dict_pre_hook(self._slice_confidence_channel, with_module=False)
        if device is not None or weight_dtype is not None:
            self.to(device=device, dtype=weight_dtype)

    def _slice_confidence_channel(self, state_dict, prefix, *unused):
        """A stage-1 (predict_conf=true) checkpoint loaded into a model without the confidence channel: the 4-row last convolution of
        the pts3d head is cut to its xyz rows -- the rule the reference applies when it loads such a checkpoint (src/main.py:146-151),
        here inside load_state_dict so that `strict=True` works for both layouts."""
        kb, kw = prefix + "downstream_head1.dpt.head.4.bias", prefix + "downstream_head1.dpt.head.4.weight"
        if not self.predict_confidence and kb in state_dict and kw in state_dict and state_dict[kb].shape[0] == 4:
            state_dict[kw], state_dict[kb] = state_dict[kw][0:3], state_dict[kb][0:3]

    def set_compute_dtype(self, dt):
        """Operand dtype of the MFMA kernels: torch.float16 (default; 10-bit mantissa = the TF32 products the reference runs at,
        backbone_vica.py:9), torch.bfloat16, or torch.float32 / "f32" -- the reference-precision path: fp32 weights AND activations
        through exact-f32 MFMA (1/16 of the 16-bit matrix rate), matching an fp32 evaluation of the reference to f32 rounding; or "split" --
        f32 activations with every weight / activation operand split into f16 (hi, lo) pairs and three f16 MFMAs per product
        (ops.SplitWeight; csrc/gemm_common.h kDtSplit): f32-class results at a third of the 16-bit matrix rate."""
        split = dt == "split"
        dt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32, "f32x": torch.float32, "split": torch.float32}.get(dt, dt)
        assert dt in (torch.float16, torch.bfloat16, torch.float32), dt
        for m in (self.backbone, self.downstream_head1, self.gaussian_param_head):
            m.compute_dtype = dt
            m.split = split

    def enable_gradient_checkpointing(self, blocks=None):
        self.backbone.enable_gradient_checkpointing(blocks)

    def map_pdf_to_opacity(self, pdf: torch.Tensor, global_step: int) -> torch.Tensor:
        c = self.cfg.opacity_mapping
        exponent = 2 ** (c.initial + min(global_step / c.warm_up, 1) * (c.final - c.initial))
        return 0.5 * (1 - (1 - pdf) ** exponent + pdf ** (1 / exponent))

    # Internal power-of-two gradient scale of the differentiable forward per operand class (autograd.BoundaryGradScale): the caller sees
    # unscaled gradients, as from the reference's fp32 training.  `encoder.grad_scale = 1.0` turns it off (e.g. when the caller runs its
    # own loss scaling, as callers.training_step does through forward_train).
    grad_scale: Optional[float] = None

    @property
    def last_backward_overflow(self):
        """0-d device flag (> 0: the last Module-API backward left an inf / NaN in some parameter gradient -- a cotangent times the
        internal scale left f16's range) or None before the first backward.  Reading it with bool() / .item() synchronises.  Under a
        dist.GradReducer the check happens in its finish(): see `reducer.last_overflow`."""
        sc = getattr(self, "_boundary_scaler", None)
        return None if sc is None else sc.last_overflow
    # Debug mode of the split operand class (`encoder.range_guard = True`): the class multiplies UNSCALED f16 (hi, lo) pairs of the f32
    # activations, so an activation with |x| >= 65520 turns into +-inf where the reference's fp32 / TF32 arithmetic has range.  With the
    # guard on, every forward audits its operands on the device (ops.range_guard) and raises ops.SplitRangeError instead of returning
    # silently wrong Gaussians.
    range_guard: bool = False
    _DEFAULT_GRAD_SCALE = {"split": 8192.0, torch.float16: 1024.0, torch.bfloat16: 1.0}

    def train_compute_class(self):
        """Operand class of the differentiable forward: "split" | torch.float16 | torch.bfloat16 (from set_compute_dtype)."""
        if self.backbone.split:
            return "split"
        return self.backbone.compute_dtype

    def forward(self, context: dict, global_step: int = 0, visualization_dump: Optional[dict] = None, distill: bool = False,
                compute_viewspace_depth: bool = True, **kwargs) -> dict:
        """The reference's module call (vicasplat.py:158-278).  Under autograd -- grad mode on and a parameter (or the input image) requires
        grad, i.e. exactly how ModelWrapper.training_step calls it (model_wrapper.py:207) -- the differentiable HIP forward runs
        (train_forward.forward_train: every operator on the hand-written kernels in both directions) and `loss.backward()` leaves plain
        gradients in `.grad`; otherwise (torch.no_grad() / eval with frozen weights) the fused, buffer-reusing inference path runs.  Both
        return the same dict."""
        image = context["image"]
        if not image.is_cuda:
            raise RuntimeError("VicaSplat.forward needs HIP device tensors: vicasplat_amd has no CPU fallback path")
        if self.range_guard and self.backbone.split:
            # debug mode of the split class: audit every MFMA operand of this forward against the f16 range of its hi halves and raise
            # ops.SplitRangeError naming the calls that left it (one extra read pass per operand + one host synchronisation)
            with ops.range_guard(image.device, raise_on_overflow=True):
                return self._dispatch(context, global_step, visualization_dump, distill, compute_viewspace_depth)
        return self._dispatch(context, global_step, visualization_dump, distill, compute_viewspace_depth)

    def _dispatch(self, context, global_step, visualization_dump, distill, compute_viewspace_depth) -> dict:
        image = context["image"]
        if torch.is_grad_enabled() and (image.requires_grad or any(p.requires_grad for p in self.parameters())):
            if self.train_compute_class() == torch.float32:
                # the exact-f32 MFMA class has no backward kernels.  In train() mode this is a training loop: fail HERE with the reason,
                # not later in loss.backward() with "does not require grad" (ADVICE r5).  In eval() mode run inference (as under no_grad)
                # and say so.
                if self.training:
                    raise RuntimeError("VicaSplat.forward: the exact-f32 operand class is inference-only (no backward kernels), but the module is in "
                                       "train() mode with grad enabled and trainable parameters.  Train in the \"split\" class "
                                       "(set_compute_dtype(\"split\"): f32 activations and gradients), or call under torch.no_grad() / eval().")
                _warn_once("f32-no-grad", "VicaSplat.forward: the exact-f32 operand class is inference-only; grad mode is on and a parameter "
                           "requires grad, but this call runs the fused no-grad path (outputs carry no graph).  Train in the \"split\" class.")
            else:
                if not self.training:
                    _warn_once("eval-autograd", "VicaSplat.forward: eval() mode with grad enabled and trainable parameters -> the differentiable "
                               "(non-fused, slower, memory-hungry) forward runs, as it would for the reference module.  For inference wrap the "
                               "call in torch.no_grad() or freeze the weights (requires_grad_(False)) to get the fused path.")
                return self._forward_autograd(context, global_step, visualization_dump, distill, compute_viewspace_depth)
        with torch.no_grad():
            return self._forward_fused(context, global_step, visualization_dump, distill, compute_viewspace_depth)

    def _forward_autograd(self, context, global_step, visualization_dump, distill, compute_viewspace_depth) -> dict:
        from ... import autograd as A
        from .train_forward import forward_train
        image = context["image"]
        B, T = image.shape[:2]
        cls = self.train_compute_class()
        S = self.grad_scale if self.grad_scale is not None else self._DEFAULT_GRAD_SCALE[cls]
        sc = getattr(self, "_boundary_scaler", None)
        if sc is None or sc.scale != S:
            # (the scaler reads self.parameters() anew at every backward: parameters unfrozen later are unscaled like the rest)
            sc = self._boundary_scaler = A.BoundaryGradScale(self, S)
        sc.rearm()
        # non-parameter leaves: their cotangents leave the encoder through the inverse node, so d(loss)/d(image) is plain too
        image, intr = sc.inputs(image, context.get("intrinsics", None))
        o = forward_train(self, image, intr, cls, global_step=global_step, distill=distill)
        g = o.get("gaussians")
        names = ("means", "covariances", "harmonics", "opacities", "scales", "rotations")
        outs = sc.outputs(o["pred_extrins"], o["pred_intrins"], o.get("raw_gaussians"), o["gaussian_centers"] if distill else None,
                          o.get("confidence"), *([g[k] for k in names] if g is not None else []))
        pred_extrins, pred_intrins, raw_gaussians, centers, conf = outs[:5]
        dev = image.device
        eye = torch.eye(4, device=dev, dtype=pred_extrins.dtype).expand(B, 1, 4, 4)
        pred_extrinsics_4x4 = torch.cat([eye, camera_matrix_from_dq_array(pred_extrins)], dim=1)
        pred_K = None
        if pred_intrins is not None:     # fov head -> pinhole K (vicasplat.py:201-205, cam_utils.py:220-234), differentiable
            fx, fy = 0.5 / torch.tan(pred_intrins[:, 0] * 0.5), 0.5 / torch.tan(pred_intrins[:, 1] * 0.5)
            zero, one, half = torch.zeros_like(fx), torch.ones_like(fx), torch.full_like(fx, 0.5)
            pred_K = torch.stack([fx, zero, half, zero, fy, half, zero, zero, one], -1).view(B, 1, 3, 3).repeat(1, T, 1, 1)
        if distill:
            return dict(pred_extrins=pred_extrins, pred_intrins=pred_intrins, gaussian_camera_extrins=pred_extrinsics_4x4,
                        gaussian_camera_intrins=pred_K, gaussian_centers=centers, confidence=conf,
                        context_view_depths=self._viewspace_depth(context, centers) if compute_viewspace_depth else None)
        gv = dict(zip(names, outs[5:]))
        gaussians = Gaussians(means=gv["means"], covariances=gv["covariances"], harmonics=gv["harmonics"],
                              opacities=gv["opacities"].unsqueeze(-1), scales=gv["scales"], rotations=gv["rotations"])
        if visualization_dump is not None:
            visualization_dump["depth"] = gaussians.means[..., -1:]
        return dict(gaussians=gaussians, pred_extrins=pred_extrins, pred_intrins=pred_intrins, raw_gaussians=raw_gaussians,
                    gaussian_camera_extrins=pred_extrinsics_4x4, gaussian_camera_intrins=pred_K, gaussian_centers=gaussians.means,
                    confidence=conf,
                    context_view_depths=self._viewspace_depth(context, gaussians.means) if compute_viewspace_depth else None)

    def _forward_fused(self, context: dict, global_step: int = 0, visualization_dump: Optional[dict] = None, distill: bool = False,
                       compute_viewspace_depth: bool = True) -> dict:
        image = context["image"]
        B, T, _, H, Wd = image.shape
        dev = image.device
        gh, gw = H // self.patch_size, Wd // self.patch_size
        video = image.permute(0, 2, 1, 3, 4)
        _, camera_embeds, _global, interms = self.backbone(video, context.get("intrinsics", None))

        head = self.camera_extrinsic_head[1]                      # nn.Sequential(ReLU, Linear): ReLU fused into the f32 linear kernel
        pred = ops.linear_f32(camera_embeds, head.weight, head.bias, relu_in=True)
        pred = torch.cat([pred[..., :3], pred[..., 3:4] + 1.0, pred[..., 4:]], -1)
        pred_extrins = pred / pred[..., :4].norm(dim=-1, keepdim=True)
        eye = torch.eye(4, device=dev, dtype=pred_extrins.dtype).expand(B, 1, 4, 4)
        pred_extrinsics_4x4 = torch.cat([eye, camera_matrix_from_dq_array(pred_extrins)], dim=1)
        pred_intrins = pred_K = None
        if _global is not None:      # no intrinsic embedding: predict the field of view -> pinhole K (vicasplat.py:201-205, cam_utils.py:220-234)
            pred_intrins = ops.linear_f32(_global, self.camera_intrinsic_head[1].weight, self.camera_intrinsic_head[1].bias, relu_in=True)
            pred_K = torch.eye(3, device=dev, dtype=torch.float32).repeat(B, 1, 1)
            pred_K[:, 0, 0], pred_K[:, 1, 1] = 0.5 / torch.tan(pred_intrins[:, 0] * 0.5), 0.5 / torch.tan(pred_intrins[:, 1] * 0.5)
            pred_K[:, 0, 2] = pred_K[:, 1, 2] = 0.5
            pred_K = pred_K[:, None].repeat(1, T, 1, 1)

        tokens = [None if t is None else t.flatten(0, 1) for t in interms]
        conf_of = lambda raw: (1.0 + torch.exp(raw[:, 3].float())).unflatten(0, (B, T)) if self.predict_confidence else None
        if distill:
            pts_raw = self.downstream_head1.forward_pts3d_raw(tokens, gh, gw)
            gs_centers = self.downstream_head1.postprocess_pts3d(pts_raw).unflatten(0, (B, T))
            return dict(pred_extrins=pred_extrins, pred_intrins=pred_intrins, gaussian_camera_extrins=pred_extrinsics_4x4,
                        gaussian_camera_intrins=pred_K, gaussian_centers=gs_centers, confidence=conf_of(pts_raw),
                        context_view_depths=self._viewspace_depth(context, gs_centers) if compute_viewspace_depth else None)
        # heads -> ONE fused kernel: 'exp' depth post-process + raw_gaussians concat + Gaussian adapter
        pts_raw = self.downstream_head1.forward_pts3d_raw(tokens, gh, gw)
        conf = conf_of(pts_raw)
        pts_raw = pts_raw[:, :3]
        gs_raw = self.gaussian_param_head.forward_gs(tokens, image.flatten(0, 1), gh, gw)
        if pts_raw.dtype != gs_raw.dtype:
            pts_raw = pts_raw.to(gs_raw.dtype)
        ga = self.gaussian_adapter
        c = self.cfg.opacity_mapping
        exponent = -1.0 if self.cfg.predict_opacity else 2 ** (c.initial + min(global_step / c.warm_up, 1) * (c.final - c.initial))
        o = ops.gaussian_adapter(pts_raw, gs_raw, ga.sh_mask, scale_act=ga.cfg.scale_act, scale_min=ga.cfg.gaussian_scale_min,
                                 scale_max=ga.cfg.gaussian_scale_max, opacity_exponent=float(exponent))
        un = lambda t: t.unflatten(0, (B, T))
        gaussians = Gaussians(means=un(o["means"]), covariances=un(o["covariances"]), harmonics=un(o["harmonics"]),
                              opacities=un(o["opacities"]), scales=un(o["scales"]), rotations=un(o["rotations"]))
        raw_gaussians = un(o["raw"])
        gs_centers = gaussians.means
        viewspace_depth = self._viewspace_depth(context, gs_centers) if compute_viewspace_depth else None
        if visualization_dump is not None:
            visualization_dump["depth"] = gaussians.means[..., -1:]
        return dict(gaussians=gaussians, pred_extrins=pred_extrins, pred_intrins=pred_intrins, raw_gaussians=raw_gaussians,
                    gaussian_camera_extrins=pred_extrinsics_4x4, gaussian_camera_intrins=pred_K, gaussian_centers=gs_centers,
                    confidence=conf, context_view_depths=viewspace_depth)

    @staticmethod
    def _viewspace_depth(context: dict, gs_centers: torch.Tensor) -> torch.Tensor:
        """z of the predicted centres in each context camera (vicasplat.py:224-232)."""
        E = context["extrinsics"]
        vp = torch.einsum("bvij,bvhwj->bvhwi", torch.linalg.inv(E[:, :, :3, :3]), gs_centers - E[:, :, None, None, :3, 3])
        return vp[..., -1]

    def get_data_shim(self):
        mean, std = self.cfg.input_mean, self.cfg.input_std

        def data_shim(batch):
            """(image - mean) / std on the context images (dataset/shims/normalize_shim.py:21-27)."""
            ctx = batch["context"]
            m = ctx["image"].new_tensor(mean)[:, None, None]
            s = ctx["image"].new_tensor(std)[:, None, None]
            return {**batch, "context": {**ctx, "image": (ctx["image"] - m) / s}}

        return data_shim
