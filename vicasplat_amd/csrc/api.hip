// Error channel + ABI version of libvicasplat_hip.so.
#include "common.h"

namespace vs {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace vs

extern "C" const char *vs_last_error(void) { return vs::g_err; }
extern "C" int vs_abi_version(void) { return 8; }   // 8 (round 6): VS_BUF_CHECKPOINT / VS_RASTER_SAVE_FOR_BACKWARD (VsRasterOut.buffers[13]), vs_raster_backward reads saved->color / depth on that route; 7: the round-5 entries (vs_head1x1_backward_split/16, vs_conv3x3_wgrad_split_stream, vs_stem7x7_up_split_stream, vs_im2col7x7_rgb); 6: colsum argument of vs_transpose_f32 / vs_transpose_pack_split; 5: round 4 -- vs_range_check (split-class range guard), ...; 2: VsRasterIn.capacity, VS_BUF_DEPTH (round 2); 3: split operands (dtype 4, vs_gemm_split, ...);
                                                      // 4: split-class backward entries, packed activations (+16 flags, vs_gemm_split_packed), vs_probe_mfma_rate

// ---- measurement aid (bench.py `roofline.sustained_mfma_tflops`): the rate the chip SUSTAINS on the matrix pipe alone.  The MFMA kernels
// of this library run power-limited (DVFS: DESIGN 5); the 2.5 PFLOP/s dense-f16 figure assumes 2.4 GHz on every CU, which gfx950 does
// not hold under matrix load.  This kernel issues v_mfma_f32_16x16x32_f16 back to back on register operands holding the caller's data
// (random values: zeros clock ~20 % higher) -- 16 independent accumulators per wave, 2 waves per SIMD, no memory traffic in the loop --
// so its rate is the ceiling any kernel built on that instruction can approach on this hardware (measured: ~1.67 PFLOP/s = 0.67 of the
// headline; v_mfma_f32_32x32x16_f16: the same, tools/probe/mfma_power.hip). ----
namespace {
typedef _Float16 half8p __attribute__((ext_vector_type(8)));
typedef float f4p __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) mfma_rate_kernel(const half8p *__restrict__ in, float *__restrict__ out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    half8p a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = in[(t * 8 + i) & 65535]; b[i] = in[(t * 8 + 4 + i) & 65535]; }
    f4p c[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = f4p{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], c[i * 4 + j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[t] = s;
}
}  // namespace

// operands: >= 1 MiB of f16 data (65536 x 16 bytes); scratch: >= 512 x 256 floats; `iters` loop trips of 16 MFMAs per wave on 512 workgroups.
// Asynchronous on `stream`; *flop_out (host) receives the FLOP count of the launch so that the caller divides by its own event timing.
extern "C" int vs_probe_mfma_rate(const void *operands, float *scratch, int32_t iters, double *flop_out_host, vs_stream_t stream_) {
    VS_CHECK(operands && scratch && iters > 0 && flop_out_host, "vs_probe_mfma_rate: bad argument");
    const int blocks = 512;
    hipLaunchKernelGGL(mfma_rate_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, (const half8p *)operands, scratch, iters);
    VS_HIP(hipGetLastError());
    *flop_out_host = (double)blocks * 4 * iters * 16 * (2.0 * 16 * 16 * 32);
    return 0;
}
