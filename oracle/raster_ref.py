"""ctypes front-end of the CPU rasterizer oracle (oracle/raster_ref.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never by the product package.  PARITY UNPINNED (see header of raster_ref.c): the reference rasterizer is an
un-vendored pip dependency (/root/reference/requirements.txt:17); this restates its published algorithm.

Also restates, in numpy float32, the host-side camera set-up of render_cuda
(/root/reference/src/model/decoder/cuda_splatting.py:18-45,187-194 and
/root/reference/src/geometry/projection.py:247-261) so oracle and product get identical matrices only
if both implement the same formulas independently.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libraster_ref.so")
    src = os.path.join(_HERE, "raster_ref.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libraster_ref.so"], stdout=subprocess.DEVNULL)
    return so


class _In(C.Structure):
    _fields_ = [
        ("P", C.c_int), ("D", C.c_int), ("M", C.c_int), ("W", C.c_int), ("H", C.c_int),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("bg", C.c_void_p), ("means3D", C.c_void_p), ("cov3D", C.c_void_p), ("shs", C.c_void_p),
        ("colors_precomp", C.c_void_p), ("opacities", C.c_void_p), ("viewmatrix", C.c_void_p),
        ("projmatrix", C.c_void_p), ("campos", C.c_void_p),
    ]


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.ref_preprocess.restype = C.c_long
        _LIB.ref_bin.restype = C.c_int
    return _LIB


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))


# ------------------------------------------------------------------------------------------------
# host-side camera math (float32, numpy)
# ------------------------------------------------------------------------------------------------
def get_fov(intrinsics: np.ndarray) -> np.ndarray:
    """projection.py:247-261 -- fov from normalised intrinsics, [b,3,3] -> [b,2] (x,y)."""
    K = np.asarray(intrinsics, dtype=np.float32)
    Kinv = np.linalg.inv(K.astype(np.float64)).astype(np.float32)

    def proc(v):
        v = np.asarray(v, dtype=np.float32)
        r = np.einsum("bij,j->bi", Kinv, v).astype(np.float32)
        return r / np.linalg.norm(r, axis=-1, keepdims=True).astype(np.float32)

    left, right = proc([0, 0.5, 1]), proc([1, 0.5, 1])
    top, bottom = proc([0.5, 0, 1]), proc([0.5, 1, 1])
    fov_x = np.arccos(np.clip((left * right).sum(-1), -1, 1)).astype(np.float32)
    fov_y = np.arccos(np.clip((top * bottom).sum(-1), -1, 1)).astype(np.float32)
    return np.stack([fov_x, fov_y], -1)


def get_projection_matrix(near, far, fov_x, fov_y) -> np.ndarray:
    """cuda_splatting.py:18-45 (z in [0,1], no principal-point term)."""
    near = np.asarray(near, np.float32); far = np.asarray(far, np.float32)
    tx = np.tan(0.5 * np.asarray(fov_x, np.float32)).astype(np.float32)
    ty = np.tan(0.5 * np.asarray(fov_y, np.float32)).astype(np.float32)
    top = ty * near; bottom = -top; right = tx * near; left = -right
    b = near.shape[0]
    r = np.zeros((b, 4, 4), np.float32)
    r[:, 0, 0] = 2 * near / (right - left)
    r[:, 1, 1] = 2 * near / (top - bottom)
    r[:, 0, 2] = (right + left) / (right - left)
    r[:, 1, 2] = (top + bottom) / (top - bottom)
    r[:, 3, 2] = 1
    r[:, 2, 2] = far / (far - near)
    r[:, 2, 3] = -(far * near) / (far - near)
    return r


@dataclass
class Camera:
    viewmatrix: np.ndarray   # [16] column-major  (== (c2w^-1)^T flattened row-major)
    projmatrix: np.ndarray   # [16]
    projmatrix_raw: np.ndarray  # [16]
    campos: np.ndarray       # [3]
    tanfovx: float
    tanfovy: float


def make_cameras(extrinsics, intrinsics, near, far) -> list[Camera]:
    """cuda_splatting.py:187-194: view = (c2w^-1)^T, full = view @ P^T (row-vector convention)."""
    E = np.asarray(extrinsics, np.float32)
    fov = get_fov(intrinsics)
    P = get_projection_matrix(near, far, fov[:, 0], fov[:, 1])
    PT = np.transpose(P, (0, 2, 1))
    view = np.transpose(np.linalg.inv(E.astype(np.float64)).astype(np.float32), (0, 2, 1))
    full = np.matmul(view, PT).astype(np.float32)
    cams = []
    for i in range(E.shape[0]):
        cams.append(Camera(np.ascontiguousarray(view[i].reshape(16)), np.ascontiguousarray(full[i].reshape(16)),
                           np.ascontiguousarray(PT[i].reshape(16)), np.ascontiguousarray(E[i, :3, 3]),
                           float(np.tan(0.5 * fov[i, 0])), float(np.tan(0.5 * fov[i, 1]))))
    return cams


# ------------------------------------------------------------------------------------------------
# rasterizer
# ------------------------------------------------------------------------------------------------
def _mk_in(cam: Camera, W, H, bg, means3D, cov6, shs, colors_precomp, opac, sh_degree):
    s = _In()
    keep = dict(bg=_f32(bg), means=_f32(means3D), cov=_f32(cov6), shs=_f32(shs), cp=_f32(colors_precomp),
                op=_f32(opac).reshape(-1), vm=_f32(cam.viewmatrix), pm=_f32(cam.projmatrix), cpos=_f32(cam.campos))
    P = keep["means"].shape[0]
    s.P = P; s.D = int(sh_degree); s.M = 0 if shs is None else keep["shs"].shape[1]; s.W = W; s.H = H
    s.tanfovx = cam.tanfovx; s.tanfovy = cam.tanfovy; s.scale_modifier = 1.0
    s.bg = _p(keep["bg"]); s.means3D = _p(keep["means"]); s.cov3D = _p(keep["cov"]); s.shs = _p(keep["shs"])
    s.colors_precomp = _p(keep["cp"]); s.opacities = _p(keep["op"]); s.viewmatrix = _p(keep["vm"])
    s.projmatrix = _p(keep["pm"]); s.campos = _p(keep["cpos"])
    return s, keep


def rasterize_forward(cam: Camera, W: int, H: int, bg, means3D, cov6, shs, opacities, sh_degree=4,
                      colors_precomp=None) -> dict:
    """One view.  shs: [P,M,3] (coefficient-major, rgb minor) or None with colors_precomp [P,3]."""
    L = lib()
    s, keep = _mk_in(cam, W, H, bg, means3D, cov6, shs, colors_precomp, opacities, sh_degree)
    P = s.P
    gx, gy = (W + 15) // 16, (H + 15) // 16
    o = dict(
        depths=np.zeros(P, np.float32), radii=np.zeros(P, np.int32), xy=np.zeros((P, 2), np.float32),
        conic_opacity=np.zeros((P, 4), np.float32), rgb=np.zeros((P, 3), np.float32),
        clamped=np.zeros((P, 3), np.uint8), rect=np.zeros((P, 4), np.int32), tiles_touched=np.zeros(P, np.int32))
    R = L.ref_preprocess(C.byref(s), _p(o["depths"]), _p(o["radii"]), _p(o["xy"]), _p(o["conic_opacity"]),
                         _p(o["rgb"]), _p(o["clamped"]), _p(o["rect"]), _p(o["tiles_touched"]))
    o["R"] = int(R)
    o["point_list"] = np.zeros(max(R, 1), np.uint32)[:R]
    o["point_keys"] = np.zeros(max(R, 1), np.uint64)[:R]
    o["ranges"] = np.zeros((gx * gy, 2), np.int32)
    pl = np.zeros(max(R, 1), np.uint32); pk = np.zeros(max(R, 1), np.uint64)
    rc = L.ref_bin(C.c_int(P), C.c_int(W), C.c_int(H), _p(o["depths"]), _p(o["rect"]), _p(o["tiles_touched"]),
                   C.c_long(R), _p(pl), _p(pk), _p(o["ranges"]))
    assert rc == 0, rc
    o["point_list"], o["point_keys"] = pl[:R], pk[:R]
    o["color"] = np.zeros((3, H, W), np.float32); o["depth"] = np.zeros((H, W), np.float32)
    o["opacity"] = np.zeros((H, W), np.float32); o["final_T"] = np.zeros((H, W), np.float32)
    o["n_contrib"] = np.zeros((H, W), np.int32); o["n_touched"] = np.zeros(P, np.int32)
    L.ref_render(C.c_int(W), C.c_int(H), _p(keep["bg"]), _p(o["ranges"]), _p(pl), _p(o["xy"]), _p(o["conic_opacity"]),
                 _p(o["rgb"]), _p(o["depths"]), _p(o["color"]), _p(o["depth"]), _p(o["opacity"]), _p(o["final_T"]),
                 _p(o["n_contrib"]), _p(o["n_touched"]))
    o["_in"] = (s, keep); o["_pl_full"] = pl
    return o


def rasterize_backward(cam: Camera, W, H, bg, means3D, cov6, shs, opacities, fwd: dict, dL_dcolor, dL_ddepth=None,
                       sh_degree=4, colors_precomp=None) -> dict:
    L = lib()
    s, keep = _mk_in(cam, W, H, bg, means3D, cov6, shs, colors_precomp, opacities, sh_degree)
    P = s.P
    g = dict(mean2D=np.zeros((P, 2), np.float32), conic=np.zeros((P, 3), np.float32), opacity=np.zeros(P, np.float32),
             colors=np.zeros((P, 3), np.float32), depths=np.zeros(P, np.float32))
    dLc = _f32(dL_dcolor); dLd = _f32(dL_ddepth)
    L.ref_render_backward(C.c_int(W), C.c_int(H), _p(keep["bg"]), _p(fwd["ranges"]), _p(fwd["_pl_full"]), _p(fwd["xy"]),
                          _p(fwd["conic_opacity"]), _p(fwd["rgb"]), _p(fwd["depths"]), _p(fwd["final_T"]),
                          _p(fwd["n_contrib"]), _p(dLc), _p(dLd), _p(g["mean2D"]), _p(g["conic"]), _p(g["opacity"]),
                          _p(g["colors"]), _p(g["depths"]))
    out = dict(means3D=np.zeros((P, 3), np.float32), cov3D=np.zeros((P, 6), np.float32),
               shs=None if shs is None else np.zeros_like(keep["shs"]), tau=np.zeros(6, np.float32),
               opacities=g["opacity"], means2D=g["mean2D"], colors=g["colors"])
    praw = _f32(cam.projmatrix_raw)
    L.ref_preprocess_backward(C.byref(s), _p(praw), _p(fwd["radii"]), _p(fwd["clamped"]), _p(g["mean2D"]), _p(g["conic"]),
                              _p(g["colors"]), _p(g["depths"]), _p(out["means3D"]), _p(out["cov3D"]), _p(out["shs"]),
                              _p(out["tau"]))
    out["_render"] = g
    return out


# ------------------------------------------------------------------------------------------------
# synthetic pixel-aligned scene of SURVEY.md 8(d) config 3 (numpy, seeded)
# ------------------------------------------------------------------------------------------------
SH_MASK = np.ones(25, np.float32)
for _deg in range(1, 5):
    SH_MASK[_deg ** 2:(_deg + 1) ** 2] = 0.1 * 0.25 ** _deg


def synthetic_scene(V: int = 2, res: int = 256, Vt: int = 4, seed: int = 0, f: float = 0.9):
    """V context cameras on the x axis (baseline 1, yaw 2deg*i); one Gaussian per context pixel."""
    rng = np.random.default_rng(seed)
    K = np.array([[f, 0, 0.5], [0, f, 0.5], [0, 0, 1]], np.float32)

    def c2w(i, n):
        yaw = np.deg2rad(2.0 * i)
        Rm = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]], np.float64)
        E = np.eye(4); E[:3, :3] = Rm; E[0, 3] = i / max(n - 1, 1)
        return E

    means, covs, ops, shs = [], [], [], []
    u = (np.arange(res) + 0.5) / res
    uu, vv = np.meshgrid(u, u, indexing="xy")
    for i in range(V):
        E = c2w(i, V)
        z = 2 + 0.5 * np.sin(2 * np.pi * uu) * np.cos(2 * np.pi * vv) + 0.05 * rng.standard_normal((res, res))
        ray = np.stack([(uu - 0.5) / f, (vv - 0.5) / f, np.ones_like(uu)], -1)
        pc = ray * z[..., None]
        pw = pc @ E[:3, :3].T + E[:3, 3]
        scale = (z / (res * f))[..., None] * np.exp(0.3 * rng.standard_normal((res, res, 3)))
        q = rng.standard_normal((res, res, 4)); q /= np.linalg.norm(q, axis=-1, keepdims=True)
        x, y, zq, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
        Rq = np.stack([1 - 2 * (y * y + zq * zq), 2 * (x * y - zq * w), 2 * (x * zq + y * w),
                       2 * (x * y + zq * w), 1 - 2 * (x * x + zq * zq), 2 * (y * zq - x * w),
                       2 * (x * zq - y * w), 2 * (y * zq + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(res, res, 3, 3)
        RS = Rq * scale[..., None, :]
        cov = RS @ np.swapaxes(RS, -1, -2)
        means.append(pw.reshape(-1, 3)); covs.append(cov.reshape(-1, 3, 3))
        ops.append(1 / (1 + np.exp(-1.5 * rng.standard_normal(res * res))))
        sh = rng.standard_normal((res * res, 3, 25)) * SH_MASK
        sh[..., 0] = 0.5 * rng.standard_normal((res * res, 3))
        shs.append(sh)
    means = np.concatenate(means).astype(np.float32); covs = np.concatenate(covs).astype(np.float32)
    ops = np.concatenate(ops).astype(np.float32); shs = np.concatenate(shs).astype(np.float32)
    tgt = []
    for j in range(Vt):
        a = (j + 0.5) / Vt
        tgt.append(c2w(a * (V - 1), V))
    tgt = np.stack(tgt).astype(np.float32)
    Kt = np.broadcast_to(K, (Vt, 3, 3)).copy()
    return dict(means=means, covariances=covs, harmonics=shs, opacities=ops, extrinsics=tgt, intrinsics=Kt,
                near=np.full(Vt, 0.01, np.float32), far=np.full(Vt, 100.0, np.float32))


def cov6(cov33: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(np.stack([cov33[:, 0, 0], cov33[:, 0, 1], cov33[:, 0, 2], cov33[:, 1, 1],
                                          cov33[:, 1, 2], cov33[:, 2, 2]], -1).astype(np.float32))


def render_views(scene: dict, res: int = 256, bg=(0, 0, 0), threads: int = 1) -> list[dict]:
    """Oracle equivalent of render_cuda(...) on a scene dict (shared Gaussians).  threads > 1: the views on that many host threads."""
    cams = make_cameras(scene["extrinsics"], scene["intrinsics"], scene["near"], scene["far"])
    shs = np.ascontiguousarray(np.transpose(scene["harmonics"], (0, 2, 1)))  # [P,25,3]
    c6 = cov6(scene["covariances"])
    one = lambda cam: rasterize_forward(cam, res, res, np.asarray(bg, np.float32), scene["means"], c6, shs, scene["opacities"])
    if threads > 1 and len(cams) > 1:      # views are independent: one host thread each (the C call releases the GIL)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(threads, len(cams))) as ex:
            return list(ex.map(one, cams))
    return [one(cam) for cam in cams]
