"""Import the REAL reference encoder (/root/reference) on CPU in THIS container to generate golden vectors.

Never used at test/bench time on the GPU box (the reference does not travel); only tests/golden/gen_*.py call it.
The reference needs packages that are not installed (jaxtyping, diffusers, omegaconf, pypose, ...): a
sys.meta_path finder hands out permissive stub modules, and three tiny behavioural shims supply what the hot path
really uses:
  * diffusers ModelMixin / ConfigMixin / register_to_config  (config capture + __getattr__ fallback to config)
  * pypose.SO3 (Hamilton product on xyzw quaternions, Inv, matrix, tensor, scalar mul / div)
  * jaxtyping.Float[...] subscriptable
"""
from __future__ import annotations

import functools
import importlib.abc
import importlib.machinery
import inspect
import os
import sys
import types

import torch
from torch import nn

REF = "/root/reference"
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

_STUB_ROOTS = {"jaxtyping", "diffusers", "omegaconf", "torchvision", "cv2", "pytorch3d", "pypose", "pyquaternion",
               "lightning", "pytorch_lightning", "dacite", "hydra", "wandb", "colorama", "lpips", "skimage", "evo",
               "moviepy", "imageio", "plotly", "tabulate", "timm", "e3nn", "plyfile", "beartype", "gsplat",
               "diff_gaussian_rasterization", "matplotlib", "roma", "trimesh", "viser", "nerfview", "kornia", "svg",
               "sklearn_extra", "open3d", "huggingface_hub_stub"}


# `curope` is the reference's compiled CUDA extension: it cannot exist here, and stubbing it would silently turn
# RoPE into a no-op.  Leaving it unresolvable makes pos_embed.py:106-110 fall back to its Python RoPE2D.
_NEVER_STUB = {"curope"}


class _Any:
    """Permissive placeholder: subscriptable, callable, attribute-able."""

    def __init__(self, *a, **k):
        pass

    def __class_getitem__(cls, item):
        return cls

    def __call__(self, *a, **k):
        return _Any()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Any()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return type(name, (_Any,), {})


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        root = fullname.split(".")[0]
        if root in _NEVER_STUB:
            return None  # must raise ImportError so the reference takes its own pure-PyTorch fallback
        if root in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        if root != "src" and "." not in fullname and importlib.machinery.PathFinder.find_spec(root) is None \
                and root not in sys.builtin_module_names:
            _STUB_ROOTS.add(root)  # any other package the reference imports but the image lacks
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


# ---- behavioural shims ------------------------------------------------------------------------------
class _Config(dict):
    __getattr__ = dict.__getitem__


def register_to_config(init):
    @functools.wraps(init)
    def wrapper(self, *args, **kwargs):
        sig = inspect.signature(init)
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
        object.__setattr__(self, "_vs_config", _Config(cfg))
        init(self, *args, **kwargs)

    return wrapper


class ConfigMixin:
    @property
    def config(self):
        return self._vs_config


class ModelMixin(nn.Module):
    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            cfg = self.__dict__.get("_vs_config")
            if cfg is not None and name in cfg:
                return cfg[name]
            raise


class SO3:
    """xyzw quaternion with the handful of pypose operations src/misc/dq.py uses."""

    def __init__(self, t):
        self.t = t.t if isinstance(t, SO3) else t

    def tensor(self):
        return self.t

    def Inv(self):
        return SO3(self.t * torch.tensor([-1.0, -1.0, -1.0, 1.0], dtype=self.t.dtype, device=self.t.device))

    def __mul__(self, o):
        if isinstance(o, SO3):
            x1, y1, z1, w1 = self.t.unbind(-1)
            x2, y2, z2, w2 = o.t.unbind(-1)
            return SO3(torch.stack([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                                    w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2], -1))
        return SO3(self.t * o)

    def __rmul__(self, o):
        return SO3(self.t * o)

    def __truediv__(self, o):
        return SO3(self.t / (o.t if isinstance(o, SO3) else o))

    def norm(self, *a, **k):
        return self.t.norm(*a, **k)

    def matrix(self):
        x, y, z, w = self.t.unbind(-1)
        return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                            2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                            2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(*self.t.shape[:-1], 3, 3)

    def __getitem__(self, i):
        return self.t[i]


def install():
    if any(isinstance(f, _Finder) for f in sys.meta_path):
        return
    sys.meta_path.insert(0, _Finder())
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import diffusers.models, diffusers.configuration_utils, pypose, pypose.lietensor.lietensor  # noqa
    sys.modules["diffusers.models"].ModelMixin = ModelMixin
    sys.modules["diffusers.configuration_utils"].ConfigMixin = ConfigMixin
    sys.modules["diffusers.configuration_utils"].register_to_config = register_to_config
    sys.modules["pypose"].SO3 = SO3
    sys.modules["pypose"].LieTensor = SO3
    sys.modules["pypose.lietensor.lietensor"].SO3Type = SO3
    sys.modules["pypose.lietensor.lietensor"].LieType = SO3


BACKBONE_CFG = dict(img_size=256, patch_size=16, enc_embed_dim=1024, enc_depth=24, enc_num_heads=16, dec_embed_dim=768,
                    dec_depth=12, dec_num_heads=12, mlp_ratio=4.0, temporal_rope_theta=30, rope_dim_list=[32, 32],
                    use_blocked_causal_attention=True, use_framewise_modulation=True, use_cross_neighbor_attention=True,
                    use_intrinsic_embedding=True)


def build_reference_encoder(backbone_overrides: dict | None = None, predict_conf: bool = False):
    """Returns the reference's VicaSplat(nn.Module) built from config/model/encoder/vicasplat.yaml + vica.yaml +
    the experiment overrides (use_intrinsic_embedding=true, temporal_rope_theta)."""
    install()
    from src.model.encoder.vicasplat import OpacityMappingCfg, VicaSplat, VicaSplatCfg
    from src.model.encoder.common.gaussian_adapter import GaussianAdapterCfg

    bb = dict(BACKBONE_CFG)
    bb.update(backbone_overrides or {})
    cfg = VicaSplatCfg(name="vicasplat", backbone=bb, visualizer=None,
                       gaussian_adapter=GaussianAdapterCfg(0.005, 0.04, 4, "softplus"), apply_bounds_shim=True,
                       opacity_mapping=OpacityMappingCfg(0.0, 0.0, 1), predict_opacity=False, predict_conf=predict_conf)
    return VicaSplat(cfg).eval()


def reference_load_images():
    """demo.py:75-132 `load_images` of the REAL reference, for fixture generation.  demo.py as a module cannot be imported here (gradio,
    tensorboard, the Lightning wrapper), so the two functions are taken out of its source with `ast` AT GENERATION TIME and executed
    with their own globals: PIL (installed), `exif_transpose`, and one behavioural shim -- `ImgNorm`, torchvision's
    Compose([ToTensor(), Normalize(0.5, 0.5)]) (uint8 HWC -> float CHW / 255, then (x - 0.5) / 0.5; torchvision is not installed)."""
    import ast
    import numpy as np
    from PIL import Image
    from PIL.ImageOps import exif_transpose
    src = open(os.path.join(REF, "demo.py")).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("_resize_pil_image", "load_images")]
    assert len(keep) == 2

    def ImgNorm(img):
        a = np.asarray(img, dtype=np.uint8)
        return (torch.from_numpy(a.copy()).permute(2, 0, 1).float() / 255.0 - 0.5) / 0.5

    ns = dict(os=os, Image=Image, exif_transpose=exif_transpose, torch=torch, ImgNorm=ImgNorm, heif_support_enabled=False)
    exec(compile(ast.Module(body=keep, type_ignores=[]), os.path.join(REF, "demo.py"), "exec"), ns)
    return ns["load_images"]
