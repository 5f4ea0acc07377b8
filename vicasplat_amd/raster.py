"""Batched Gaussian rasterizer op (autograd-aware) on top of the C ABI (vs_raster_forward / vs_raster_backward).

One call renders every camera of every scene in the batch: Gaussians are stored once per scene and shared by the
scene's cameras (the reference replicates them per view, decoder_splatting_cuda.py:86-89, and loops over views in
Python, cuda_splatting.py:199-238).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import threading
from typing import Optional

import torch

from . import _lib as L


def _f32c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


_hint = threading.local()


class CapacityScope:
    """What `instance_capacity` yields: the record of EVERY rasterizer call made inside the context (a decoder may issue several per
    forward, or none), not only the most recent one."""

    def __init__(self, n):
        self.n = n
        self.calls: list = []      # per call: (host num_rendered, device int64[4] misc = [R, largest tile, overflow flag, 0])

    def overflow_flag(self):
        """Device int64 scalar: non-zero when some call of the scope needed more than its capacity (it rendered background only).  No
        host synchronisation: read it whenever the caller next synchronises.  None if no call was made."""
        flags = [m[2] for _, m in self.calls]
        return None if not flags else torch.stack(flags).max()

    def instances(self):
        """Device int64 scalar: the largest instance count any call of the scope needed (exact mode: produced)."""
        cnt = [m[0] for _, m in self.calls]
        return None if not cnt else torch.stack(cnt).max()


@contextlib.contextmanager
def instance_capacity(n: Optional[int]):
    """Inside this context every rasterizer call runs in the CAPACITY mode of vs_raster_forward (include/vicasplat_hip.h): buffers sized
    for `n` (Gaussian, tile) instances, NO host synchronisation per call (the exact mode copies the instance count back, as upstream's
    extension does per view).  A call that needs more than `n` renders nothing and raises a device-side flag; the yielded CapacityScope
    accumulates the flags and counts of ALL calls of the context (`overflow_flag()`, `instances()`), which the caller reads whenever
    it next synchronises, repeating with a larger capacity on overflow (callers.align_poses, bench.py).  n = None: exact mode (the scope
    still records the calls)."""
    prev, prev_scope = getattr(_hint, "n", None), getattr(_hint, "scope", None)
    scope = CapacityScope(None if n is None else int(n))
    _hint.n, _hint.scope = scope.n, scope
    try:
        yield scope
    finally:
        _hint.n, _hint.scope = prev, prev_scope


def last_call() -> Optional[dict]:
    """{'num_rendered': host int (== capacity in the capacity mode), 'misc': device int64[4] = [R, largest tile, overflow flag, 0]} of
    this thread's most recent rasterizer call."""
    return getattr(_hint, "last", None)


def _forward_impl(means3D, cov3D, shs, colors_precomp, opacities, viewmatrix, projmatrix, campos, tanfov, background,
                  cam_scene, H, W, sh_degree, flags):
    """Calls vs_raster_forward; returns (outputs, state) where state keeps every device buffer alive."""
    dev = L.require_device(means3D, cov3D, shs, colors_precomp, opacities, viewmatrix, projmatrix, campos, tanfov, background)
    lib = L.lib()
    S, P = means3D.shape[0], means3D.shape[1]
    Cn = viewmatrix.shape[0]
    cov33 = cov3D.dim() == 4
    M = 0
    if shs is not None:
        M = shs.shape[3] if (flags & L.VS_RASTER_SH_RGB_MAJOR) else shs.shape[2]
    if cov33:
        flags |= L.VS_RASTER_COV_3X3
    inp = L.VsRasterIn()
    inp.num_cameras, inp.num_scenes, inp.P = Cn, S, P
    inp.sh_degree, inp.sh_coeffs, inp.width, inp.height, inp.flags = int(sh_degree), int(M), int(W), int(H), int(flags)
    inp.means3D, inp.cov3D, inp.shs, inp.colors_precomp = L.ptr(means3D), L.ptr(cov3D), L.ptr(shs), L.ptr(colors_precomp)
    inp.opacities, inp.cam_scene = L.ptr(opacities), L.ptr(cam_scene)
    inp.viewmatrix, inp.projmatrix, inp.campos = L.ptr(viewmatrix), L.ptr(projmatrix), L.ptr(campos)
    inp.tanfov, inp.background = L.ptr(tanfov), L.ptr(background)
    inp.capacity = getattr(_hint, "n", None) or 0

    color = torch.empty((Cn, 3, H, W), dtype=torch.float32, device=dev)
    depth = torch.empty((Cn, H, W), dtype=torch.float32, device=dev)
    opacity = torch.empty((Cn, H, W), dtype=torch.float32, device=dev)
    radii = torch.empty((Cn, P), dtype=torch.int32, device=dev)
    n_touched = torch.empty((Cn, P), dtype=torch.int32, device=dev) if (flags & L.VS_RASTER_COUNT_TOUCHED) else None
    out = L.VsRasterOut()
    out.color, out.depth, out.opacity, out.radii, out.n_touched = L.ptr(color), L.ptr(depth), L.ptr(opacity), L.ptr(radii), L.ptr(n_touched)
    alloc = L.TorchAllocator(dev)
    with torch.cuda.device(dev):
        R = lib.vs_raster_forward(C.byref(inp), C.byref(out), alloc.fn, None, L.stream_ptr(dev))
    alloc.fn = None  # break the allocator <-> ctypes-callback reference cycle so the buffers die with their last user
    L.check(R, "vs_raster_forward")
    if n_touched is None:
        n_touched = torch.zeros((Cn, P), dtype=torch.int32, device=dev)
    _hint.last = dict(num_rendered=int(R), misc=alloc.tensors[L.VS_BUF_MISC].view(torch.int64)[:4])
    if getattr(_hint, "scope", None) is not None:
        _hint.scope.calls.append((int(R), _hint.last["misc"]))
    state = dict(inp=inp, out=out, alloc=alloc, dims=(S, P, Cn, M, H, W, cov33), num_rendered=int(R),
                 keep=(means3D, cov3D, shs, colors_precomp, opacities, viewmatrix, projmatrix, campos, tanfov, background,
                       cam_scene, color, depth, opacity, radii))
    return (color, radii, depth, opacity, n_touched), state


def forward_debug(means3D, cov3D, opacities, viewmatrix, projmatrix, campos, tanfov, background, H, W, *, shs=None,
                  colors_precomp=None, sh_degree=0, sh_rgb_major=False, cam_scene=None, count_touched=True) -> dict:
    """Forward + typed views of the internal buffers (for the parity tests: integer data must be bit-exact)."""
    flags = (L.VS_RASTER_SH_RGB_MAJOR if sh_rgb_major else 0) | (L.VS_RASTER_COUNT_TOUCHED if count_touched else 0)
    Cn = viewmatrix.shape[0]
    if cam_scene is not None:
        cam_scene = cam_scene.to(torch.int32).contiguous()
    outs, st = _forward_impl(_f32c(means3D), _f32c(cov3D), _f32c(shs), _f32c(colors_precomp), _f32c(opacities),
                             _f32c(viewmatrix).reshape(Cn, 16), _f32c(projmatrix).reshape(Cn, 16), _f32c(campos),
                             _f32c(tanfov), _f32c(background), cam_scene, int(H), int(W), int(sh_degree), flags)
    S, P, Cn, M, H, W, _ = st["dims"]
    t = st["alloc"].tensors
    R = st["num_rendered"]
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    return dict(color=outs[0], radii=outs[1], depth=outs[2], opacity=outs[3], n_touched=outs[4], R=R,
                geom=t[L.VS_BUF_GEOM].view(torch.float32)[:Cn * P * 12].view(Cn, P, 12),
                rect=t[L.VS_BUF_RECT].view(torch.int16)[:Cn * P * 4].view(Cn, P, 4),
                clamped=t[L.VS_BUF_CLAMPED][:Cn * P].view(Cn, P),
                ranges=t[L.VS_BUF_TILE_RANGES].view(torch.int32)[:Cn * tiles * 2].view(Cn, tiles, 2),
                point_list=t[L.VS_BUF_POINT_LIST].view(torch.int32)[:R],
                final_T=t[L.VS_BUF_FINAL_T].view(torch.float32)[:Cn * H * W].view(Cn, H, W),
                n_contrib=t[L.VS_BUF_N_CONTRIB].view(torch.int32)[:Cn * H * W].view(Cn, H, W), _state=st)


def _backward_impl(inp, out, grads, dev) -> None:
    """Calls vs_raster_backward (its scratch -- the per-(camera, Gaussian) gradient records -- comes from torch's caching allocator and goes
    back to it on return).  A module-level function so that bench.py can bracket it with events."""
    lib = L.lib()
    alloc = L.TorchAllocator(dev)
    with torch.cuda.device(dev):
        rc = lib.vs_raster_backward(C.byref(inp), C.byref(out), C.byref(grads), alloc.fn, None, L.stream_ptr(dev))
    alloc.fn = None
    L.check(rc, "vs_raster_backward")


class _Rasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, cov3D, shs, colors_precomp, opacities, viewmatrix, projmatrix, campos, tanfov, background,
                cam_scene, theta, rho, projmatrix_raw, H, W, sh_degree, flags):
        # a differentiated call also stores the blending checkpoints (VS_BUF_CHECKPOINT): the backward then replays the 512-entry segments
        # of every tile list independently instead of walking each list in one wave (round 6)
        if any(ctx.needs_input_grad):
            flags |= L.VS_RASTER_SAVE_FOR_BACKWARD
        outs, st = _forward_impl(means3D, cov3D, shs, colors_precomp, opacities, viewmatrix, projmatrix, campos, tanfov,
                                 background, cam_scene, H, W, sh_degree, flags)
        color, radii, depth, opacity, n_touched = outs
        # binning scratch the backward never reads (keys, sort ping-pong, depth keys, rectangles, cursors: ~25 bytes per instance)
        # goes back to the caching allocator now; later work on this stream may reuse it
        for tag in (L.VS_BUF_KEYS, L.VS_BUF_SORT_SCRATCH, L.VS_BUF_DEPTH, L.VS_BUF_RECT, L.VS_BUF_TILE_CURSOR):
            st["alloc"].tensors.pop(tag, None)
        ctx.inp, ctx.out, ctx.alloc = st["inp"], st["out"], st["alloc"]
        # inputs whose device pointers sit in ctx.inp.  NO output tensor may be stored on ctx directly: an output holds its
        # grad_fn (this node) and the node would hold the output -- a cycle through C++ references that Python's collector never
        # sees, i.e. one leaked copy of the scene's Gaussians (and of everything their graph saved) per differentiated render.
        # The backward needs exactly one output, radii (ctx.out.radii): it goes through save_for_backward.
        ctx.keep = (means3D, cov3D, shs, colors_precomp, opacities, viewmatrix, projmatrix, campos, tanfov, background,
                    cam_scene, projmatrix_raw)
        # outputs the backward reads (radii always; the rendered colour / depth on the checkpoint route): through save_for_backward only
        ctx.save_for_backward(radii, color, depth)
        ctx.set_materialize_grads(False)     # an output no loss reads (the depth image, usually) arrives as None, not as a zero image
        ctx.want_tau = theta is not None or rho is not None
        ctx.dims = st["dims"]
        ctx.num_rendered = st["num_rendered"]
        ctx.mark_non_differentiable(radii, n_touched, opacity)
        return color, radii, depth, opacity, n_touched

    @staticmethod
    def backward(ctx, g_color, g_radii, g_depth, g_opacity, g_touched):
        lib = L.lib()
        if not hasattr(lib, "vs_raster_backward"):
            raise RuntimeError("libvicasplat_hip.so was built without vs_raster_backward")
        (means3D, cov3D, shs, colors_precomp, opacities, viewmatrix, projmatrix, campos, tanfov, background, cam_scene,
         projmatrix_raw) = ctx.keep
        radii, color, depth = ctx.saved_tensors    # keeps the buffers behind ctx.out.radii / color / depth alive
        ctx.out.radii, ctx.out.color, ctx.out.depth = L.ptr(radii), L.ptr(color), L.ptr(depth)
        S, P, Cn, M, H, W, cov33 = ctx.dims
        dev = means3D.device
        g_color = _f32c(g_color) if g_color is not None else torch.zeros((Cn, 3, H, W), dtype=torch.float32, device=dev)
        g_depth = _f32c(g_depth)
        g = L.VsRasterGrads()
        # every element of these is stored by preprocess_backward_kernel (one thread per Gaussian)
        d_means = torch.empty_like(means3D)
        d_cov = torch.empty((S, P, 3, 3) if cov33 else (S, P, 6), dtype=torch.float32, device=dev)   # layout of the input covariances
        d_shs = torch.empty_like(shs) if shs is not None else None
        d_cp = torch.empty_like(colors_precomp) if colors_precomp is not None else None
        d_op = torch.empty_like(opacities)
        # dL_dmeans2D (upstream's screen-space gradient holder, a densification statistic) has no consumer on this path: not requested --
        # [C,P,2] f32 that the C side would clear and preprocess_backward_kernel would fill (2.4 GB of traffic per 288 views)
        d_tau = torch.zeros((Cn, 6), dtype=torch.float32, device=dev) if ctx.want_tau else None
        g.dL_dcolor, g.dL_ddepth = L.ptr(g_color), L.ptr(g_depth)
        g.dL_dmeans3D, g.dL_dcov3D, g.dL_dshs, g.dL_dcolors_precomp = L.ptr(d_means), L.ptr(d_cov), L.ptr(d_shs), L.ptr(d_cp)
        g.dL_dopacities, g.dL_dmeans2D, g.dL_dtau = L.ptr(d_op), None, L.ptr(d_tau)
        _backward_impl(ctx.inp, ctx.out, g, dev)
        d_theta = d_tau[:, 3:] if d_tau is not None else None
        d_rho = d_tau[:, :3] if d_tau is not None else None
        return (d_means, d_cov, d_shs, d_cp, d_op, None, None, None, None, None, None, d_theta, d_rho, None, None, None,
                None, None)


def rasterize(means3D, cov3D, opacities, viewmatrix, projmatrix, campos, tanfov, background, image_height, image_width, *,
              shs=None, colors_precomp=None, sh_degree=0, sh_rgb_major=False, cam_scene=None, theta=None, rho=None,
              projmatrix_raw=None, count_touched=False):
    """means3D [S,P,3]; cov3D [S,P,6] | [S,P,3,3]; shs [S,P,M,3] | [S,P,3,M] (sh_rgb_major); opacities [S,P];
    viewmatrix / projmatrix [C,16] (or [C,4,4], row-major flatten == the reference's transposed storage);
    campos [C,3]; tanfov [C,2]; background [C,3]; cam_scene int32 [C] or None.
    Returns (color [C,3,H,W], radii [C,P], depth [C,H,W], opacity [C,H,W], n_touched [C,P])."""
    flags = 0
    if sh_rgb_major:
        flags |= L.VS_RASTER_SH_RGB_MAJOR
    if count_touched:
        flags |= L.VS_RASTER_COUNT_TOUCHED
    Cn = viewmatrix.shape[0]
    if cam_scene is not None:
        cam_scene = cam_scene.to(torch.int32).contiguous()
    return _Rasterize.apply(
        _f32c(means3D), _f32c(cov3D), _f32c(shs), _f32c(colors_precomp), _f32c(opacities), _f32c(viewmatrix).reshape(Cn, 16),
        _f32c(projmatrix).reshape(Cn, 16), _f32c(campos), _f32c(tanfov), _f32c(background), cam_scene, theta, rho,
        _f32c(projmatrix_raw), int(image_height), int(image_width), int(sh_degree), flags)
