"""The C-ABI library builds for gfx950 without a GPU, loads, and exports every symbol include/*.h declares.
No compute calls here (CPU suite).  Also: argument validation returns an error code + message, never crashes."""
import ctypes as C
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    syms = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        txt = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        syms |= set(re.findall(r"\b(vs_[a-z0-9_]+)\s*\(", txt))
    return syms


def test_library_exports_every_declared_symbol():
    from vicasplat_amd import _lib
    so = _lib.build()
    assert os.path.exists(so)
    L = C.CDLL(so)
    syms = _declared_symbols()
    assert {"vs_raster_forward", "vs_raster_backward", "vs_rope2d", "vs_gemm_bias_act", "vs_attention", "vs_layernorm_mod",
            "vs_rope_qk", "vs_gaussian_adapter", "vs_last_error", "vs_abi_version"} <= syms
    missing = [s for s in sorted(syms) if not hasattr(L, s)]
    assert not missing, missing
    assert _lib.lib().vs_abi_version() == _lib.ABI_VERSION
    # the documented version is the library's (INTEGRATION.md said 7 while the library said 6)
    doc = open(os.path.join(os.path.dirname(__file__), '..', 'INTEGRATION.md')).read()
    assert f'(ABI {_lib.ABI_VERSION})' in doc


def test_argument_validation_reports_errors():
    from vicasplat_amd import _lib
    L = _lib.lib()
    rc = L.vs_rope2d(None, None, 1, 1, 1, 64, 64, 64, 100.0, 1.0, 0, None)
    assert rc < 0 and b"null" in L.vs_last_error()
    rc = L.vs_gemm_bias_act(C.c_void_p(16), C.c_void_p(16), None, C.c_void_p(16), None, 4, 4, 30, 32, 32, 4, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, None)
    assert rc < 0 and b"multiple of 64" in L.vs_last_error()
    rc = L.vs_raster_forward(None, None, _lib.AllocFn(lambda a, b, c: 0), None, None)
    assert rc < 0


def test_product_package_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under vicasplat_amd/ may import it."""
    bad = []
    for f in glob.glob(os.path.join(ROOT, "vicasplat_amd", "**", "*.py"), recursive=True):
        src = open(f).read()
        if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
            bad.append(f)
    assert not bad, bad


def test_python_constants_mirror_the_header_enums():
    """The ctypes side indexes VsRasterOut.buffers and sets VsRasterIn.flags by number: the numbers are the header's."""
    from vicasplat_amd import _lib
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "vicasplat_hip.h")).read()
    enums = {m.group(1): int(m.group(2)) for m in re.finditer(r"\b(VS_(?:BUF|RASTER)_[A-Z_0-9]+)\s*=\s*(\d+)", hdr)}
    assert enums["VS_BUF_COUNT"] == 13 and enums["VS_BUF_CHECKPOINT"] == 12
    for name, val in enums.items():
        if hasattr(_lib, name):
            assert getattr(_lib, name) == val, (name, getattr(_lib, name), val)
    for name in ("VS_BUF_GEOM", "VS_BUF_CHECKPOINT", "VS_BUF_COUNT", "VS_RASTER_SAVE_FOR_BACKWARD", "VS_RASTER_SH_RGB_MAJOR", "VS_RASTER_COV_3X3"):
        assert hasattr(_lib, name) and name in enums, name
    assert C.sizeof(_lib.VsRasterOut) == 6 * 8 + enums["VS_BUF_COUNT"] * 8


def test_head_tail_refuses_an_input_that_is_not_a_relu_output():
    """the fused tail of a DPT head masks its input gradient by t > 0 and tells the producer to skip its ReLU backward -- only
    valid behind Conv3x3Fn(..., relu_out=True).  The producer is identified by its autograd node; anything else must not take the route."""
    import pytest
    import torch
    from vicasplat_amd import autograd as A
    t = torch.randn(32, 256, requires_grad=True) * 2.0          # produced by a multiplication, not by the ReLU convolution
    w = torch.randn(83, 256)
    assert not A.head_tail_ok(t, w)
    with pytest.raises(ValueError, match="relu_out=True"):
        A.head_tail(t, w, None, A.SPLIT)
    leaf = torch.randn(32, 256)                                   # outside any graph: no consumer of dt, shapes decide
    assert A.head_tail_ok(leaf, w)
