// Fused "DPT post-process + Gaussian adapter" for gfx950: one pass from the two heads' raw outputs to
// rasterizer-ready Gaussian attributes.  Memory-bound.
//
// Replaces, on the hot path:  heads/postprocess.py:46-56 (xyz * expm1(|xyz|) / |xyz|), the torch.cat that builds
// raw_gaussians (vicasplat.py:256) and MyGaussianAdapter.forward (common/gaussian_adapter.py:168-212,
// common/gaussians.py:8-44, vicasplat.py:143-156): split 86 channels, sigmoid + pdf->opacity map, softplus scale,
// quaternion normalise, Sigma = R diag(s)^2 R^T, SH * sh_mask.  The PyTorch formulation launches ~40 elementwise
// kernels plus a batched 3x3 GEMM (8.9 ms per 2 scenes on MI355X); this is one launch.
//
// Layout: one thread per pixel; inputs are read with arbitrary (pixel, channel) strides so the NHWC (channels-last)
// conv outputs are consumed in place; outputs are written per Gaussian in the layouts the decoder / rasterizer take
// (means [P,3], covariances [P,3,3], harmonics [P,3,d_sh], opacities [P], scales [P,3], rotations [P,4], raw [P,11+3*d_sh]).
#include "common.h"

namespace {

struct AdapterArgs {
    const void *pts;  // 3 channels
    const void *gs;   // 8 + 3*d_sh channels: opacity | scale3 | quat4 (xyzw) | sh (rgb-major)
    long long pts_pix, pts_ch, gs_pix, gs_ch;  // element strides
    int in_dtype;     // 0 f32, 1 f16, 2 bf16
    long long npix;
    int d_sh;
    const float *sh_mask;
    int scale_act;    // 0 bounded, 1 exp, 2 softplus
    float scale_min, scale_max, opacity_exponent;
    float *means, *cov, *harmonics, *opacities, *scales, *rotations, *raw;
};

template <int DT>
__device__ __forceinline__ float ld(const void *p, long long i) {
    if constexpr (DT == 0) return reinterpret_cast<const float *>(p)[i];
    else if constexpr (DT == 1) return (float)reinterpret_cast<const _Float16 *>(p)[i];
    else return __uint_as_float(((unsigned)reinterpret_cast<const unsigned short *>(p)[i]) << 16);
}

template <int DT>
__global__ void __launch_bounds__(256) adapter_kernel(const AdapterArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.npix) return;
    const int nsh = a.d_sh, craw = 11 + 3 * nsh;
    // ---- centre: 'exp' depth mode ----
    const float x = ld<DT>(a.pts, i * a.pts_pix), y = ld<DT>(a.pts, i * a.pts_pix + a.pts_ch), z = ld<DT>(a.pts, i * a.pts_pix + 2 * a.pts_ch);
    const float d = sqrtf(x * x + y * y + z * z);
    const float k = expm1f(d) / fmaxf(d, 1e-8f);
    const float mx = x * k, my = y * k, mz = z * k;
    a.means[3 * i] = mx; a.means[3 * i + 1] = my; a.means[3 * i + 2] = mz;
    const long long g0 = i * a.gs_pix;
    const float o_raw = ld<DT>(a.gs, g0);
    float sr[3], qr[4];
#pragma unroll
    for (int c = 0; c < 3; ++c) sr[c] = ld<DT>(a.gs, g0 + (1 + c) * a.gs_ch);
#pragma unroll
    for (int c = 0; c < 4; ++c) qr[c] = ld<DT>(a.gs, g0 + (4 + c) * a.gs_ch);
    // ---- opacity ----
    float p = 1.0f / (1.0f + expf(-o_raw));
    if (a.opacity_exponent > 0.0f) {
        const float e = a.opacity_exponent;
        p = e == 1.0f ? 0.5f * (1.0f - (1.0f - p) + p) : 0.5f * (1.0f - powf(1.0f - p, e) + powf(p, 1.0f / e));
    }
    a.opacities[i] = p;
    // ---- scales ----
    float s[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = sr[c];
        if (a.scale_act == 0) v = a.scale_min + (a.scale_max - a.scale_min) / (1.0f + expf(-v));
        else if (a.scale_act == 1) v = fminf(expf(v), 0.3f);
        else v = fminf(0.001f * (v > 20.0f ? v : log1pf(expf(v))), 0.3f);  // F.softplus threshold 20
        s[c] = v;
        a.scales[3 * i + c] = v;
    }
    // ---- rotation (xyzw), covariance ----
    const float qn = fmaxf(sqrtf(qr[0] * qr[0] + qr[1] * qr[1] + qr[2] * qr[2] + qr[3] * qr[3]), 1e-12f);
    const float qi = qr[0] / qn, qj = qr[1] / qn, qk = qr[2] / qn, qw = qr[3] / qn;
    a.rotations[4 * i] = qi; a.rotations[4 * i + 1] = qj; a.rotations[4 * i + 2] = qk; a.rotations[4 * i + 3] = qw;
    const float two_s = 2.0f / (qi * qi + qj * qj + qk * qk + qw * qw + 1e-8f);
    const float R[3][3] = {{1 - two_s * (qj * qj + qk * qk), two_s * (qi * qj - qk * qw), two_s * (qi * qk + qj * qw)},
                           {two_s * (qi * qj + qk * qw), 1 - two_s * (qi * qi + qk * qk), two_s * (qj * qk - qi * qw)},
                           {two_s * (qi * qk - qj * qw), two_s * (qj * qk + qi * qw), 1 - two_s * (qi * qi + qj * qj)}};
    float RS[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) RS[r][c] = R[r][c] * s[c];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) a.cov[9 * i + 3 * r + c] = RS[r][0] * RS[c][0] + RS[r][1] * RS[c][1] + RS[r][2] * RS[c][2];
    // ---- spherical harmonics + raw copy ----
    float *raw = a.raw ? a.raw + i * craw : nullptr;
    if (raw) {
        raw[0] = mx; raw[1] = my; raw[2] = mz; raw[3] = o_raw;
#pragma unroll
        for (int c = 0; c < 3; ++c) raw[4 + c] = sr[c];
#pragma unroll
        for (int c = 0; c < 4; ++c) raw[7 + c] = qr[c];
    }
    float *h = a.harmonics + i * 3 * nsh;
    for (int c = 0; c < 3 * nsh; ++c) {
        const float v = ld<DT>(a.gs, g0 + (8 + c) * a.gs_ch);
        h[c] = v * a.sh_mask[c % nsh];
        if (raw) raw[11 + c] = v;
    }
}

}  // namespace

extern "C" int vs_gaussian_adapter(const void *pts, int64_t pts_pix, int64_t pts_ch, const void *gs, int64_t gs_pix,
                                   int64_t gs_ch, int32_t in_dtype, int64_t npix, int32_t d_sh, const float *sh_mask,
                                   int32_t scale_act, float scale_min, float scale_max, float opacity_exponent, float *means,
                                   float *cov, float *harmonics, float *opacities, float *scales, float *rotations, float *raw,
                                   vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(pts && gs && sh_mask && means && cov && harmonics && opacities && scales && rotations, "vs_gaussian_adapter: null pointer");
    VS_CHECK(in_dtype >= 0 && in_dtype <= 2 && d_sh > 0 && scale_act >= 0 && scale_act <= 2, "vs_gaussian_adapter: bad argument");
    if (npix <= 0) return 0;
    AdapterArgs a{pts, gs, pts_pix, pts_ch, gs_pix, gs_ch, in_dtype, npix, d_sh, sh_mask, scale_act, scale_min, scale_max,
                  opacity_exponent, means, cov, harmonics, opacities, scales, rotations, raw};
    dim3 grid((unsigned)vs::cdiv64(npix, 256)), block(256);
    if (in_dtype == 0) hipLaunchKernelGGL(adapter_kernel<0>, grid, block, 0, stream, a);
    else if (in_dtype == 1) hipLaunchKernelGGL(adapter_kernel<1>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(adapter_kernel<2>, grid, block, 0, stream, a);
    VS_HIP(hipGetLastError());
    return 0;
}
