"""ctypes binding of libvicasplat_hip.so (the C ABI declared in include/vicasplat_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a tensor is not on a HIP device the
call raises.  PyTorch is used only for device memory (caching allocator) and streams.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("VICASPLAT_HIP_LIB") or os.path.join(_HERE, "libvicasplat_hip.so")   # (override: A/B runs of two builds)
_lock = threading.Lock()
ABI_VERSION = 8     # == vs_abi_version() of csrc/api.hip; INTEGRATION.md lists the entries of every version
_lib = None

VS_BUF_GEOM, VS_BUF_RECT, VS_BUF_CLAMPED, VS_BUF_TILE_RANGES, VS_BUF_TILE_CURSOR, VS_BUF_KEYS, VS_BUF_POINT_LIST, \
    VS_BUF_SORT_SCRATCH, VS_BUF_FINAL_T, VS_BUF_N_CONTRIB, VS_BUF_MISC, VS_BUF_DEPTH, VS_BUF_CHECKPOINT, VS_BUF_COUNT = range(14)
VS_RASTER_COUNT_TOUCHED = 1
VS_RASTER_SAVE_FOR_BACKWARD = 2
VS_RASTER_SH_RGB_MAJOR = 4
VS_RASTER_COV_3X3 = 8

AllocFn = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int32, C.c_size_t)


class VsRasterIn(C.Structure):
    _fields_ = [
        ("num_cameras", C.c_int32), ("num_scenes", C.c_int32), ("P", C.c_int32), ("sh_degree", C.c_int32),
        ("sh_coeffs", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("flags", C.c_int32),
        ("means3D", C.c_void_p), ("cov3D", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p),
        ("opacities", C.c_void_p), ("cam_scene", C.c_void_p), ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p),
        ("campos", C.c_void_p), ("tanfov", C.c_void_p), ("background", C.c_void_p), ("capacity", C.c_int64),
    ]


class VsRasterOut(C.Structure):
    _fields_ = [
        ("color", C.c_void_p), ("depth", C.c_void_p), ("opacity", C.c_void_p), ("radii", C.c_void_p),
        ("n_touched", C.c_void_p), ("num_rendered", C.c_int64), ("buffers", C.c_void_p * VS_BUF_COUNT),
    ]


class VsRasterGrads(C.Structure):
    _fields_ = [
        ("dL_dcolor", C.c_void_p), ("dL_ddepth", C.c_void_p), ("dL_dmeans3D", C.c_void_p), ("dL_dcov3D", C.c_void_p),
        ("dL_dshs", C.c_void_p), ("dL_dcolors_precomp", C.c_void_p), ("dL_dopacities", C.c_void_p),
        ("dL_dmeans2D", C.c_void_p), ("dL_dtau", C.c_void_p),
    ]


def build(force: bool = False) -> str:
    """Compile every HIP source for gfx950 (cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    newest = max(os.path.getmtime(os.path.join(src_dir, f)) for f in os.listdir(src_dir)
                 if f.endswith((".hip", ".h", "Makefile")))
    newest = max(newest, os.path.getmtime(os.path.join(_HERE, "..", "include", "vicasplat_hip.h")))
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < newest:
        subprocess.check_call(["make", "-C", src_dir, "-j8"], stdout=subprocess.DEVNULL)
    return _SO


def lib() -> C.CDLL:
    """Load the C-ABI library; raise loudly when it is absent (no silent fallback)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(_SO):
                raise RuntimeError(
                    f"{_SO} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(vicasplat_amd has no CPU / PyTorch fallback path)")
            L = C.CDLL(_SO)
            L.vs_last_error.restype = C.c_char_p
            L.vs_abi_version.restype = C.c_int
            if L.vs_abi_version() != ABI_VERSION:     # the ctypes mirrors of the structs below are for exactly this layout
                raise RuntimeError(f"{_SO} has ABI version {L.vs_abi_version()}, this package needs {ABI_VERSION}: rebuild it (python -c 'import __graft_entry__ as g; g.build()')")
            L.vs_raster_forward.restype = C.c_int64
            L.vs_raster_forward.argtypes = [C.POINTER(VsRasterIn), C.POINTER(VsRasterOut), AllocFn, C.c_void_p, C.c_void_p]
            L.vs_rope2d.restype = C.c_int
            L.vs_rope2d.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64,
                                    C.c_int64, C.c_float, C.c_float, C.c_int32, C.c_void_p]
            i32, i64, vp, f32 = C.c_int32, C.c_int64, C.c_void_p, C.c_float
            L.vs_range_check.restype = C.c_int
            L.vs_range_check.argtypes = [vp, i64, i32, i64, i32, f32, vp, i32, vp]
            L.vs_layernorm_mod.restype = C.c_int
            L.vs_layernorm_mod.argtypes = [vp, i64, vp, vp, vp, vp, i32, i32, vp, i64, i32, i32, i32, f32, i32, i32, i32, vp]
            L.vs_gemm_bias_act.restype = C.c_int
            L.vs_gemm_bias_act.argtypes = [vp, vp, vp, vp, vp] + [i32] * 16 + [vp]
            L.vs_rope_qk.restype = C.c_int
            L.vs_rope_qk.argtypes = [vp, i64, i32, i32, i32, vp, vp, f32, f32, i32, vp]
            L.vs_attention.restype = C.c_int
            L.vs_attention.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i64, i64, i32, i32, i32, i32, vp, vp, f32, i32, vp]
            L.vs_gaussian_adapter_backward.restype = C.c_int
            L.vs_gaussian_adapter_backward.argtypes = [vp, i32, vp, i32, i64, i32, vp, i32, f32, f32, f32, vp, vp, vp, vp, vp, vp, i32, vp, i32, vp]
            L.vs_gaussian_adapter.restype = C.c_int
            L.vs_gaussian_adapter.argtypes = [vp, i64, i64, vp, i64, i64, i32, i64, i32, vp, i32, f32, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp]
            L.vs_rope_qk_dir.restype = C.c_int
            L.vs_rope_qk_dir.argtypes = [vp, i64, i32, i32, i32, vp, vp, f32, f32, f32, i32, vp]
            L.vs_attention_lse.restype = C.c_int
            L.vs_attention_lse.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i64, i64, i32, i32, i32, i32, vp, vp, f32, i32, vp, vp]
            L.vs_attention_backward.restype = C.c_int
            L.vs_attention_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i64, i64, i32, i32, i32, i32, i32,
                                                i32, i32, i32, vp, vp, i32, f32, i32, vp]
            L.vs_attention_backward16.restype = C.c_int
            L.vs_attention_backward16.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i64, i64, i32, i32, i32, i32, i32,
                                                  i32, i32, i32, vp, f32, i32, vp]
            L.vs_upsample2x_backward_nhwc.restype = C.c_int
            L.vs_upsample2x_backward_nhwc.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
            L.vs_relu_mask16.restype = C.c_int
            L.vs_relu_mask16.argtypes = [vp, vp, i64, vp]
            L.vs_relu_mask16_to.restype = C.c_int
            L.vs_relu_mask16_to.argtypes = [vp, vp, vp, i64, vp]
            L.vs_gemm_splitk_accumulate.restype = C.c_int
            L.vs_gemm_splitk_accumulate.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]
            L.vs_gemm_taps_accumulate.restype = C.c_int
            L.vs_gemm_taps_accumulate.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i64, vp, i32, i32, i32, vp]
            L.vs_gemm_wgrad.restype = C.c_int
            L.vs_gemm_wgrad.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i64, i64, i64, vp, i32, i32, i32, vp, i64, i32, vp]
            L.vs_conv3x3_wgrad_tn.restype = C.c_int
            L.vs_conv3x3_wgrad_tn.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, i64, i32, vp]
            L.vs_gemm_wgrad_tn.restype = C.c_int
            L.vs_gemm_wgrad_tn.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, i64, i32, vp]
            L.vs_gemm_resid.restype = C.c_int
            L.vs_gemm_resid.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]
            L.vs_transpose16.restype = C.c_int
            L.vs_transpose16.argtypes = [vp, i64, vp, i64, i32, i32, i32, vp]
            L.vs_transpose16_ex.restype = C.c_int
            L.vs_transpose16_ex.argtypes = [vp, i64, vp, i64, i32, i32, i32, vp, i32, i32, i32, i32, i32, i32, i64, vp]
            L.vs_colsum.restype = C.c_int
            L.vs_colsum.argtypes = [vp, i64, vp, i32, i32, i32, vp]
            L.vs_gated_resid.restype = C.c_int
            L.vs_gated_resid.argtypes = [vp, vp, i64, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp]
            L.vs_gated_resid_backward.restype = C.c_int
            L.vs_gated_resid_backward.argtypes = [vp, vp, i64, vp, i32, vp, i64, vp, i32, i32, i32, i32, i32, i32, vp]
            L.vs_gelu16.restype = C.c_int
            L.vs_gelu16.argtypes = [vp, vp, i64, i32, vp]
            L.vs_gelu_backward.restype = C.c_int
            L.vs_gelu_backward.argtypes = [vp, vp, vp, i64, i32, vp]
            L.vs_layernorm_backward.restype = C.c_int
            L.vs_layernorm_backward.argtypes = [vp, i64, i32, vp, i64, vp, vp, vp, i32, i32, vp, i64, i32, vp, vp, vp, vp, i32, i32, f32,
                                                i32, i32, i32, vp]
            L.vs_layernorm_backward_ex.restype = C.c_int
            L.vs_layernorm_backward_ex.argtypes = [vp, i64, i32, vp, i64, vp, vp, vp, i32, i32, vp, i64, vp, i64, vp, i64, i32, vp, vp, vp, vp,
                                                   i32, i32, f32, i32, i32, i32, vp]
            L.vs_gemm_qkv_rope.restype = C.c_int
            L.vs_gemm_qkv_rope.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, i32,
                                           C.c_float, C.c_float, vp]
            L.vs_conv7x7_rgb_nhwc.restype = C.c_int
            L.vs_conv7x7_rgb_nhwc.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
            L.vs_conv3x3_nhwc.restype = C.c_int
            L.vs_conv3x3_nhwc.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]
            L.vs_linear_f32.restype = C.c_int
            L.vs_linear_f32.argtypes = [vp, i64, vp, i64, vp, vp, i64, i32, i32, i32, i32, vp]
            L.vs_silu_cast.restype = C.c_int
            L.vs_silu_cast.argtypes = [vp, vp, i64, i32, vp]
            L.vs_conv3x3_head1x1_nhwc.restype = C.c_int
            L.vs_conv3x3_head1x1_nhwc.argtypes = [vp, vp, vp, vp, vp, vp] + [i32] * 11 + [vp]
            L.vs_split_pack_weight.restype = C.c_int
            L.vs_split_pack_weight.argtypes = [vp, i64, vp, i64, i32, i32, i32, vp]
            L.vs_gemm_split.restype = C.c_int
            L.vs_gemm_split.argtypes = [vp, vp, f32, vp, vp, vp, vp] + [i32] * 15 + [vp, vp, i32, f32, f32, vp]
            L.vs_gemm_split_packed.restype = C.c_int
            L.vs_gemm_split_packed.argtypes = [vp, vp, f32, vp, vp, vp, vp] + [i32] * 15 + [vp, vp, i32, f32, f32, vp]
            L.vs_conv3x3_split_nhwc.restype = C.c_int
            L.vs_conv3x3_split_nhwc.argtypes = [vp, vp, f32, vp, vp, vp] + [i32] * 8 + [vp]
            L.vs_conv3x3_split_res2_nhwc.restype = C.c_int
            L.vs_conv3x3_split_res2_nhwc.argtypes = [vp, vp, f32, vp, vp, vp, vp] + [i32] * 8 + [vp]
            L.vs_conv3x3_head1x1_split_nhwc.restype = C.c_int
            L.vs_conv3x3_head1x1_split_nhwc.argtypes = [vp, vp, f32, vp, vp, f32, vp, vp] + [i32] * 8 + [vp]
            L.vs_conv3x3_head_dot_split_nhwc.restype = C.c_int
            L.vs_conv3x3_head_dot_split_nhwc.argtypes = [vp, vp, f32, vp, vp, vp, vp] + [i32] * 8 + [vp]
            L.vs_conv7x7_rgb_split_nhwc.restype = C.c_int
            L.vs_conv7x7_rgb_split_nhwc.argtypes = [vp, vp, f32, vp, vp, i32, i32, i32, i32, i32, i32, vp]
            L.vs_conv7x7_rgb_split_up_nhwc.restype = C.c_int
            L.vs_conv7x7_rgb_split_up_nhwc.argtypes = [vp, vp, f32, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
            L.vs_upsample2x_nhwc.restype = C.c_int
            L.vs_upsample2x_nhwc.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
            L.vs_probe_mfma_rate.restype = C.c_int
            L.vs_probe_mfma_rate.argtypes = [vp, vp, i32, C.POINTER(C.c_double), vp]
            L.vs_transpose_f32.restype = C.c_int
            L.vs_transpose_f32.argtypes = [vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]
            L.vs_gemm_wgrad_split_atn.restype = C.c_int
            L.vs_gemm_wgrad_split_atn.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i64, i32, vp]
            L.vs_conv3x3_wgrad_split_atn.restype = C.c_int
            L.vs_conv3x3_wgrad_split_atn.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i64, i32, vp]
            L.vs_transpose_pack_split.restype = C.c_int
            L.vs_transpose_pack_split.argtypes = [vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]
            L.vs_split16.restype = C.c_int
            L.vs_split16.argtypes = [vp, i64, vp, vp, i64, i64, i32, vp]
            L.vs_attention_backward_split.restype = C.c_int
            L.vs_attention_backward_split.argtypes = [vp] * 15 + [i32, i32, i32, i32, i64, i64] + [i32] * 9 + [vp, vp, i32, f32, vp]
            for nm, at in (("vs_gelu_f32", [vp, vp, i64, vp]), ("vs_gelu_backward_f32", [vp, vp, vp, i64, vp]), ("vs_relu_mask_f32", [vp, vp, vp, i64, vp]),
                           ("vs_gated_resid_f32", [vp, vp, i64, vp, i32, vp, i32, i32, i32, i32, i32, vp]),
                           ("vs_gated_resid_backward_f32", [vp, vp, i64, vp, i32, vp, i64, vp, i32, i32, i32, i32, i32, vp]),
                           ("vs_upsample2x_backward_f32_nhwc", [vp, vp, i32, i32, i32, i32, vp]),
                           ("vs_head1x1_backward_split", [vp, i64, vp, vp, i32, vp, vp, vp, i64, i32, i32, i32, i32, vp]),
                           ("vs_head1x1_backward16", [vp, i64, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, vp]),
                           ("vs_conv3x3_wgrad_split_stream", [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
                           ("vs_stem7x7_up_split_stream", [vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
                           ("vs_im2col7x7_rgb", [vp, vp, i32, i32, i32, i32, i32, vp])):
                getattr(L, nm).restype = C.c_int
                getattr(L, nm).argtypes = at
            if hasattr(L, "vs_raster_backward"):
                L.vs_raster_backward.restype = C.c_int
                L.vs_raster_backward.argtypes = [C.POINTER(VsRasterIn), C.POINTER(VsRasterOut), C.POINTER(VsRasterGrads),
                                                 AllocFn, C.c_void_p, C.c_void_p]
            _lib = L
    return _lib


def check(rc: int, what: str) -> int:
    if rc < 0:
        raise RuntimeError(f"{what}: {lib().vs_last_error().decode()}")
    return rc


def require_device(*tensors: torch.Tensor) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("vicasplat_amd kernels need HIP device tensors (got a CPU tensor); there is no CPU fallback")
        dev = dev or t.device
        if t.device != dev:
            raise RuntimeError("all tensors must live on the same device")
    return dev


def stream_ptr(device: torch.device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t) -> C.c_void_p:
    return C.c_void_p(None) if t is None else C.c_void_p(t.data_ptr())


class TorchAllocator:
    """VsAllocFn backed by the PyTorch caching allocator; keeps the tensors alive, indexed by tag."""

    def __init__(self, device: torch.device):
        self.device = device
        self.tensors: dict[int, torch.Tensor] = {}
        self.fn = AllocFn(self._alloc)

    def _alloc(self, _ctx, tag, nbytes):
        t = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        self.tensors[int(tag)] = t
        return t.data_ptr()
