"""vicasplat_amd -- MI355X-native (gfx950) implementation of VicaSplat's feed-forward hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all compute on the path goes
through hand-written HIP kernels behind the C ABI in include/vicasplat_hip.h (libvicasplat_hip.so).
The sub-packages mirror the reference's import surface:

    vicasplat_amd.diff_gaussian_rasterization   <- `diff_gaussian_rasterization` (cuda_splatting.py:5-8)
    vicasplat_amd.curope                        <- croco/curope (curope2d.py, curope.cpp)
    vicasplat_amd.model.encoder / .decoder      <- src/model/encoder, src/model/decoder
"""
__version__ = "0.1.0"
