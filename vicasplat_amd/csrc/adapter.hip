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

#include <cstdlib>

#include <type_traits>

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


// ---- fast path: dense NHWC 16-bit inputs (what the heads produce).  One wave handles 64 consecutive pixels: the
// 64 x (8+3*d_sh) input block and every output block are CONTIGUOUS in HBM, so they move as 16-byte coalesced
// vectors through an LDS staging tile; the per-pixel math runs on LDS-resident values. ----
constexpr int kMaxCh = 96;  // 11 + 3*d_sh <= 96  (d_sh <= 28)
constexpr int kMaxPixStride = 96;  // largest pixel stride (channels incl. padding) of the gs input of the dense kernel

template <bool BF16>
__device__ __forceinline__ float cvt16(unsigned short h) {
    if constexpr (BF16) return __uint_as_float(((unsigned)h) << 16);
    else { _Float16 f = *reinterpret_cast<_Float16 *>(&h); return (float)f; }
}

template <int DT>
__device__ __forceinline__ float cvtin(float v) { return v; }
template <int DT>
__device__ __forceinline__ float cvtin(unsigned short h) { return cvt16<DT == 2>(h); }

// x / d == umulhi(x, div_magic(d)) for every index used here (x < 64 * 96, 2 <= d <= 96): the error x * (m d - 2^32) stays below 2^32
__device__ __forceinline__ unsigned div_magic(unsigned d) { return (unsigned)((0x100000000ull + d - 1) / d); }   // d >= 2

// One block of `per` floats per pixel for np pixels, written as 16-byte vectors straight to HBM (dst is 16-byte aligned: block start
// * per * 64 pixels); element (px, c) comes from f(px, c).  No LDS image of the output: the wide rows (raw 86 floats, harmonics 75)
// are conversions of the staged 16-bit input row, and an output tile in LDS (24.6 KiB) held this kernel to 4 waves per CU.
template <class F>
__device__ __forceinline__ void write_rows(float *__restrict__ dst, int np, int per, int lane, F f) {
    const unsigned m = div_magic((unsigned)per);
    const int n = np * per, nv = n >> 2;
    for (int k4 = lane; k4 < nv; k4 += 64) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned k = (unsigned)(4 * k4 + j), px = __umulhi(k, m);
            v[j] = f((int)px, (int)(k - px * (unsigned)per));
        }
        // streaming (nontemporal) stores: 4.8 GB of attributes written once, read much later by the rasterizer; cycle stamps showed the
        // workgroups waiting 56 % of their time for their 12 KB of INPUT behind this write traffic (3.08 -> 2.73 ms)
        typedef float f4nt __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(f4nt{v[0], v[1], v[2], v[3]}, reinterpret_cast<f4nt *>(dst) + k4);
    }
    for (int k = (nv << 2) + lane; k < n; k += 64) {
        const unsigned px = __umulhi((unsigned)k, m);
        dst[k] = f((int)px, (int)((unsigned)k - px * (unsigned)per));
    }
}

// DT: 0 = f32 inputs (the f32 / split operand classes), 1 = f16, 2 = bf16
// NPX pixels per workgroup (64 or 32).  Round 5: the f32 inputs of the split class need 34.5 KB of LDS per 64-pixel block -- four one-wave
// workgroups per CU, i.e. ONE wave per SIMD for a kernel that only moves bytes (4.46 ms per 24-scene step = 3.1 TB/s of its 14 GB).  With 32
// pixels per block (the upper half of the lanes idles in the per-pixel arithmetic, which is nothing beside the traffic) nine workgroups fit.
template <int DT, int NPX = 64>
__global__ void __launch_bounds__(64) adapter_nhwc16_kernel(const AdapterArgs a) {
    typedef typename std::conditional<DT == 0, float, unsigned short>::type elem_t;
    constexpr int EV = 16 / (int)sizeof(elem_t);   // elements per 16-byte vector
    // pixel strides may exceed the channel counts (gs rows padded to a multiple of 16 channels, pts rows to 4: what the fused
    // conv3 -> conv1 head kernel writes); the staged blocks keep the stride
    constexpr int kPx = 28;   // per-pixel results: means 3 | raw opacity 1 | raw scale 3 | raw quat 4 | cov 9 | scale 3 | quat 4 | opacity 1
    __shared__ __attribute__((aligned(16))) elem_t sin[NPX * kMaxPixStride + NPX * 8 + 16];
    __shared__ float spx[NPX][kPx + 1];
    __shared__ float smask[kMaxCh];
    const int lane = threadIdx.x;
    const long long p0 = (long long)blockIdx.x * NPX;
    const int np = (int)min((long long)NPX, a.npix - p0);
    const int nsh = a.d_sh, cg = (int)a.gs_pix, cp = (int)a.pts_pix, craw = 11 + 3 * nsh;
    elem_t *sgs = sin, *spt = sin + ((NPX * cg + 7) & ~7);
    {   // coalesced 16-byte loads of the two input blocks
        const elem_t *ggs = reinterpret_cast<const elem_t *>(a.gs) + p0 * cg;
        const elem_t *gpt = reinterpret_cast<const elem_t *>(a.pts) + p0 * cp;
        const int n1 = np * cg, n2 = np * cp;
        for (int k = lane; k < n1 / EV; k += 64) reinterpret_cast<uint4 *>(sgs)[k] = reinterpret_cast<const uint4 *>(ggs)[k];
        for (int k = (n1 / EV) * EV + lane; k < n1; k += 64) sgs[k] = ggs[k];
        for (int k = lane; k < n2 / EV; k += 64) reinterpret_cast<uint4 *>(spt)[k] = reinterpret_cast<const uint4 *>(gpt)[k];
        for (int k = (n2 / EV) * EV + lane; k < n2; k += 64) spt[k] = gpt[k];
        for (int c = lane; c < 3 * nsh; c += 64) smask[c] = a.sh_mask[c % nsh];   // mask per (channel, coefficient) column
    }
    __syncthreads();
    const bool live = lane < np;
    const elem_t *mg = sgs + lane * cg;
    float mx = 0.f, my = 0.f, mz = 0.f, p = 0.f, s[3] = {0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 1.f}, cov[9];
    float o_raw = 0.f, sr[3] = {0.f, 0.f, 0.f}, qr[4] = {0.f, 0.f, 0.f, 1.f};
#pragma unroll
    for (int c = 0; c < 9; ++c) cov[c] = 0.f;
    if (live) {
        const float x = cvtin<DT>(spt[lane * cp]), y = cvtin<DT>(spt[lane * cp + 1]), z = cvtin<DT>(spt[lane * cp + 2]);
        const float d = sqrtf(x * x + y * y + z * z);
        const float k = expm1f(d) / fmaxf(d, 1e-8f);
        mx = x * k; my = y * k; mz = z * k;
        o_raw = cvtin<DT>(mg[0]);
#pragma unroll
        for (int c = 0; c < 3; ++c) sr[c] = cvtin<DT>(mg[1 + c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) qr[c] = cvtin<DT>(mg[4 + c]);
        p = 1.0f / (1.0f + expf(-o_raw));
        if (a.opacity_exponent > 0.0f && a.opacity_exponent != 1.0f) {
            const float e = a.opacity_exponent;
            p = 0.5f * (1.0f - powf(1.0f - p, e) + powf(p, 1.0f / e));
        } else if (a.opacity_exponent == 1.0f) {
            p = 0.5f * (1.0f - (1.0f - p) + p);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = sr[c];
            if (a.scale_act == 0) v = a.scale_min + (a.scale_max - a.scale_min) / (1.0f + expf(-v));
            else if (a.scale_act == 1) v = fminf(expf(v), 0.3f);
            else v = fminf(0.001f * (v > 20.0f ? v : log1pf(expf(v))), 0.3f);
            s[c] = v;
        }
        const float qn = fmaxf(sqrtf(qr[0] * qr[0] + qr[1] * qr[1] + qr[2] * qr[2] + qr[3] * qr[3]), 1e-12f);
#pragma unroll
        for (int c = 0; c < 4; ++c) q[c] = qr[c] / qn;
        const float qi = q[0], qj = q[1], qk = q[2], qw = q[3];
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
            for (int c = 0; c < 3; ++c) cov[3 * r + c] = RS[r][0] * RS[c][0] + RS[r][1] * RS[c][1] + RS[r][2] * RS[c][2];
    }
    if (lane < NPX) {   // per-pixel results -> LDS rows (29-float stride: conflict-free for the row-per-lane stores), read back by the block writers
        float *r = spx[lane];
        r[0] = mx; r[1] = my; r[2] = mz; r[3] = o_raw;
#pragma unroll
        for (int c = 0; c < 3; ++c) r[4 + c] = sr[c];
#pragma unroll
        for (int c = 0; c < 4; ++c) r[7 + c] = qr[c];
#pragma unroll
        for (int c = 0; c < 9; ++c) r[11 + c] = cov[c];
#pragma unroll
        for (int c = 0; c < 3; ++c) r[20 + c] = s[c];
#pragma unroll
        for (int c = 0; c < 4; ++c) r[23 + c] = q[c];
        r[27] = p;
    }
    __syncthreads();
    // raw [px][11 + 3 d_sh] = means | pre-activation opacity, scale, quaternion | SH as the head wrote them
    if (a.raw)
        write_rows(a.raw + p0 * craw, np, craw, lane, [&](int px, int c) { return c < 11 ? spx[px][c] : cvtin<DT>(sgs[px * cg + c - 3]); });
    // harmonics [px][3][d_sh] = SH * sh_mask
    write_rows(a.harmonics + p0 * 3 * nsh, np, 3 * nsh, lane, [&](int px, int c) { return cvtin<DT>(sgs[px * cg + 8 + c]) * smask[c]; });
    write_rows(a.cov + p0 * 9, np, 9, lane, [&](int px, int c) { return spx[px][11 + c]; });
    write_rows(a.means + p0 * 3, np, 3, lane, [&](int px, int c) { return spx[px][c]; });
    write_rows(a.scales + p0 * 3, np, 3, lane, [&](int px, int c) { return spx[px][20 + c]; });
    write_rows(a.rotations + p0 * 4, np, 4, lane, [&](int px, int c) { return spx[px][23 + c]; });
    if (live) a.opacities[p0 + lane] = p;
}

// ---- backward of the adapter (training): gradients of the Gaussian attributes -> gradients of the two heads' 16-bit NHWC
// outputs.  One wave per 64 consecutive pixels; the wide per-pixel gradient rows (harmonics 3*d_sh floats, raw 11+3*d_sh
// floats) and the 16-bit outputs move through LDS as contiguous blocks, the per-pixel chain rule runs in registers:
//   means  m = x k(d), k = expm1(d)/d            dx = g k + (g.x) k'(d) x / d
//   opacity p = sigmoid(o) [pdf->opacity map]     do = gp p (1-p) [* map']
//   scale   s = min(0.001 softplus(v), 0.3) ...   dv = gs ds/dv
//   cov     = (R S)(R S)^T                         dRS = (G + G^T) RS,  ds_c = sum_r dRS_rc R_rc,  dR_rc = dRS_rc s_c
//   R(q), q = qr/|qr|, two_s = 2/(q.q + 1e-8)      dq from the 9 entries + the two_s term, then through the normalisation
//   harmonics = sh * mask                           dsh = g * mask
// plus d_raw (gradient of the raw_gaussians output: means | pre-activation channels), added where given. ----
struct AdapterBwdArgs {
    const void *pts, *gs;      // forward inputs, dense NHWC 16-bit: pts [npix, pts_pix] (3 used), gs [npix, 8 + 3 d_sh]
    int pts_pix;
    long long npix;
    int d_sh;
    const float *sh_mask;
    int scale_act;
    float scale_min, scale_max, opacity_exponent;
    const float *d_means, *d_cov, *d_harm, *d_op, *d_raw;   // d_raw may be null
    void *d_pts, *d_gs;        // outputs in the inputs' dtype: d_pts [npix, d_pts_ld], d_gs [npix, d_gs_ld] (row strides >= channels;
    int d_pts_ld, d_gs_ld;     // padding columns are written as zeros, so that the rows can be 16-byte aligned for the consumers)
};

template <bool BF16>
__device__ __forceinline__ unsigned short to16a(float v) {
    if constexpr (BF16) {
        unsigned u = __float_as_uint(v);
        u += 0x7FFFu + ((u >> 16) & 1u);
        return (unsigned short)(u >> 16);
    } else {
        _Float16 h = (_Float16)v;
        return *reinterpret_cast<unsigned short *>(&h);
    }
}

// DT 0: f32 inputs and gradients (split / f32 operand classes), 1 f16, 2 bf16
template <int DT>
__global__ void __launch_bounds__(64) adapter_backward_kernel(const AdapterBwdArgs a) {
    // per-pixel results (8 leading d_gs channels) and the mask per SH column are all that goes through LDS: the wide rows (d_harm, d_raw
    // -> d_gs) are an element-wise map written straight from global to global as 16-byte vectors (the [64][75] + [64][86] float tiles
    // plus the 16-bit output tile of the first version were 58 KiB: two waves per CU)
    __shared__ float s_dg8[64][9];
    __shared__ float smask[kMaxCh];
    const int lane = threadIdx.x;
    const long long p0 = (long long)blockIdx.x * 64;
    const int np = (int)min((long long)64, a.npix - p0);
    const int nsh = a.d_sh, cg = 8 + 3 * nsh, craw = 11 + 3 * nsh, nh = 3 * nsh;
    for (int c = lane; c < nh; c += 64) smask[c] = a.sh_mask[c % nsh];
    const bool live = lane < np;
    const long long i = p0 + lane;
    float dpt[3] = {0.f, 0.f, 0.f}, dg8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (live) {
        typedef typename std::conditional<DT == 0, float, unsigned short>::type in_t;
        const in_t *pp = reinterpret_cast<const in_t *>(a.pts) + i * a.pts_pix;
        const in_t *gg = reinterpret_cast<const in_t *>(a.gs) + i * cg;
        const float *rr = a.d_raw ? a.d_raw + i * craw : nullptr;   // the 11 leading columns of this pixel's raw-gradient row
        // ---- means ----
        const float x = cvtin<DT>(pp[0]), y = cvtin<DT>(pp[1]), z = cvtin<DT>(pp[2]);
        float gm[3] = {a.d_means[3 * i], a.d_means[3 * i + 1], a.d_means[3 * i + 2]};
        if (rr) { gm[0] += rr[0]; gm[1] += rr[1]; gm[2] += rr[2]; }
        const float d = sqrtf(x * x + y * y + z * z);
        if (d > 1e-8f) {
            const float em = expm1f(d), k = em / d, kp = ((em + 1.0f) * d - em) / (d * d);
            const float gx = (gm[0] * x + gm[1] * y + gm[2] * z) * kp / d;
            dpt[0] = gm[0] * k + gx * x; dpt[1] = gm[1] * k + gx * y; dpt[2] = gm[2] * k + gx * z;
        } else {   // m = x * expm1(d) / 1e-8 in this branch of the forward's clamp
            const float k = expm1f(d) / 1e-8f;
            dpt[0] = gm[0] * k; dpt[1] = gm[1] * k; dpt[2] = gm[2] * k;
        }
        // ---- opacity ----
        const float o_raw = cvtin<DT>(gg[0]);
        const float p = 1.0f / (1.0f + expf(-o_raw));
        float dmap = 1.0f;
        if (a.opacity_exponent > 0.0f && a.opacity_exponent != 1.0f) {
            const float e = a.opacity_exponent;
            dmap = 0.5f * (e * powf(1.0f - p, e - 1.0f) + (1.0f / e) * powf(p, 1.0f / e - 1.0f));
        }
        dg8[0] = a.d_op[i] * dmap * p * (1.0f - p) + (rr ? rr[3] : 0.f);
        // ---- scales, rotation (forward recomputed) ----
        float sr[3], qr[4], s[3], dsdv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) sr[c] = cvtin<DT>(gg[1 + c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) qr[c] = cvtin<DT>(gg[4 + c]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = sr[c];
            if (a.scale_act == 0) {
                const float sg = 1.0f / (1.0f + expf(-v));
                s[c] = a.scale_min + (a.scale_max - a.scale_min) * sg;
                dsdv[c] = (a.scale_max - a.scale_min) * sg * (1.0f - sg);
            } else if (a.scale_act == 1) {
                const float e = expf(v);
                s[c] = fminf(e, 0.3f);
                dsdv[c] = e < 0.3f ? e : 0.f;
            } else {
                const float sp = 0.001f * (v > 20.0f ? v : log1pf(expf(v)));
                s[c] = fminf(sp, 0.3f);
                dsdv[c] = sp < 0.3f ? 0.001f * (v > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-v))) : 0.f;
            }
        }
        const float qn = fmaxf(sqrtf(qr[0] * qr[0] + qr[1] * qr[1] + qr[2] * qr[2] + qr[3] * qr[3]), 1e-12f);
        const float qi = qr[0] / qn, qj = qr[1] / qn, qk = qr[2] / qn, qw = qr[3] / qn;
        const float t = 2.0f / (qi * qi + qj * qj + qk * qk + qw * qw + 1e-8f);
        const float R[3][3] = {{1 - t * (qj * qj + qk * qk), t * (qi * qj - qk * qw), t * (qi * qk + qj * qw)},
                               {t * (qi * qj + qk * qw), 1 - t * (qi * qi + qk * qk), t * (qj * qk - qi * qw)},
                               {t * (qi * qk - qj * qw), t * (qj * qk + qi * qw), 1 - t * (qi * qi + qj * qj)}};
        float G[3][3], RS[3][3], dRS[3][3], dR[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) { G[r][c] = a.d_cov[9 * i + 3 * r + c]; RS[r][c] = R[r][c] * s[c]; }
        float ds[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float v = 0.f;
#pragma unroll
                for (int m = 0; m < 3; ++m) v += (G[r][m] + G[m][r]) * RS[m][c];
                dRS[r][c] = v;
                ds[c] += v * R[r][c];
                dR[r][c] = v * s[c];
            }
#pragma unroll
        for (int c = 0; c < 3; ++c) dg8[1 + c] = ds[c] * dsdv[c] + (rr ? rr[4 + c] : 0.f);
        // dL/dt (entries are 1 - t*u or t*u) and the explicit q dependence at fixed t
        const float dt_ = -dR[0][0] * (qj * qj + qk * qk) + dR[0][1] * (qi * qj - qk * qw) + dR[0][2] * (qi * qk + qj * qw)
                          + dR[1][0] * (qi * qj + qk * qw) - dR[1][1] * (qi * qi + qk * qk) + dR[1][2] * (qj * qk - qi * qw)
                          + dR[2][0] * (qi * qk - qj * qw) + dR[2][1] * (qj * qk + qi * qw) - dR[2][2] * (qi * qi + qj * qj);
        float dq[4];
        dq[0] = t * (dR[0][1] * qj + dR[0][2] * qk + dR[1][0] * qj - 2.0f * dR[1][1] * qi - dR[1][2] * qw + dR[2][0] * qk + dR[2][1] * qw - 2.0f * dR[2][2] * qi);
        dq[1] = t * (-2.0f * dR[0][0] * qj + dR[0][1] * qi + dR[0][2] * qw + dR[1][0] * qi + dR[1][2] * qk - dR[2][0] * qw + dR[2][1] * qk - 2.0f * dR[2][2] * qj);
        dq[2] = t * (-2.0f * dR[0][0] * qk - dR[0][1] * qw + dR[0][2] * qi + dR[1][0] * qw - 2.0f * dR[1][1] * qk + dR[1][2] * qj + dR[2][0] * qi + dR[2][1] * qj);
        dq[3] = t * (-dR[0][1] * qk + dR[0][2] * qj + dR[1][0] * qk - dR[1][2] * qi - dR[2][0] * qj + dR[2][1] * qi);
        const float q[4] = {qi, qj, qk, qw};
        const float dtq = -t * t * dt_;            // dt/dq_c = -t^2 q_c
#pragma unroll
        for (int c = 0; c < 4; ++c) dq[c] += dtq * q[c];
        // through q = qr / max(|qr|, 1e-12)
        const float qdq = q[0] * dq[0] + q[1] * dq[1] + q[2] * dq[2] + q[3] * dq[3];
#pragma unroll
        for (int c = 0; c < 4; ++c) dg8[4 + c] = (dq[c] - q[c] * qdq) / qn + (rr ? rr[7 + c] : 0.f);
    }
    // ---- d_gs block [np][gld] 16-bit: columns 0..7 from the per-pixel chain rule, 8.. = d_harm * mask (+ d_raw), padding zero ----
    const int gld = a.d_gs_ld;
#pragma unroll
    for (int c = 0; c < 8; ++c) s_dg8[lane][c] = dg8[c];
    __syncthreads();
    {
        const float *gh = a.d_harm + p0 * nh;
        const float *gr = a.d_raw ? a.d_raw + p0 * craw : nullptr;
        const unsigned m = div_magic((unsigned)gld);
        auto elemf = [&](unsigned k) -> float {
            const unsigned px = __umulhi(k, m), c = k - px * (unsigned)gld;
            float v = 0.f;
            if (c < 8u) v = s_dg8[px][c];
            else if (c < (unsigned)cg) v = gh[px * nh + (c - 8)] * smask[c - 8] + (gr ? gr[px * craw + 3 + c] : 0.f);
            return v;
        };
        const int n = np * gld;
        if constexpr (DT == 0) {
            float *dst = reinterpret_cast<float *>(a.d_gs) + p0 * gld;
            const int nv = n >> 2;
            for (int k4 = lane; k4 < nv; k4 += 64)
                reinterpret_cast<float4 *>(dst)[k4] = make_float4(elemf((unsigned)(4 * k4)), elemf((unsigned)(4 * k4 + 1)), elemf((unsigned)(4 * k4 + 2)),
                                                                   elemf((unsigned)(4 * k4 + 3)));
            for (int k = (nv << 2) + lane; k < n; k += 64) dst[k] = elemf((unsigned)k);
        } else {
            unsigned short *dst = reinterpret_cast<unsigned short *>(a.d_gs) + p0 * gld;
            auto elem = [&](unsigned k) -> unsigned short { return to16a<DT == 2>(elemf(k)); };
            const int nv = n >> 3;
            for (int k8 = lane; k8 < nv; k8 += 64) {
                unsigned w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = (unsigned)elem((unsigned)(8 * k8 + 2 * j)) | ((unsigned)elem((unsigned)(8 * k8 + 2 * j + 1)) << 16);
                reinterpret_cast<uint4 *>(dst)[k8] = make_uint4(w[0], w[1], w[2], w[3]);
            }
            for (int k = (nv << 3) + lane; k < n; k += 64) dst[k] = elem((unsigned)k);
        }
    }
    if (live) {
        if constexpr (DT == 0) {
            float *dp = reinterpret_cast<float *>(a.d_pts) + i * a.d_pts_ld;
#pragma unroll
            for (int c = 0; c < 3; ++c) dp[c] = dpt[c];
            for (int c = 3; c < a.d_pts_ld; ++c) dp[c] = 0.f;
        } else {
            unsigned short *dp = reinterpret_cast<unsigned short *>(a.d_pts) + i * a.d_pts_ld;
#pragma unroll
            for (int c = 0; c < 3; ++c) dp[c] = to16a<DT == 2>(dpt[c]);
            for (int c = 3; c < a.d_pts_ld; ++c) dp[c] = 0;
        }
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
    const int ev = in_dtype == 0 ? 4 : 8;   // elements per 16-byte vector: every 64-pixel block must start on one
    const bool dense16 = pts_ch == 1 && pts_pix >= 3 && pts_pix <= 8 && gs_ch == 1 && gs_pix >= 8 + 3 * d_sh &&
                         gs_pix <= kMaxPixStride && 11 + 3 * d_sh <= kMaxCh && ((uintptr_t)pts & 15) == 0 && ((uintptr_t)gs & 15) == 0 &&
                         ((64 * pts_pix) % ev == 0) && ((64 * gs_pix) % ev == 0);
    if (dense16) {
        dim3 g64((unsigned)vs::cdiv64(npix, 64));
        static const int npx = [] { const char *e = getenv("VS_ADAPTER_NPX"); return e ? atoi(e) : 32; }();
        if (in_dtype == 0 && npx == 32 && ((32 * pts_pix) % ev == 0) && ((32 * gs_pix) % ev == 0))
            hipLaunchKernelGGL((adapter_nhwc16_kernel<0, 32>), dim3((unsigned)vs::cdiv64(npix, 32)), dim3(64), 0, stream, a);
        else if (in_dtype == 0) hipLaunchKernelGGL(adapter_nhwc16_kernel<0>, g64, dim3(64), 0, stream, a);
        else if (in_dtype == 1) hipLaunchKernelGGL(adapter_nhwc16_kernel<1>, g64, dim3(64), 0, stream, a);
        else hipLaunchKernelGGL(adapter_nhwc16_kernel<2>, g64, dim3(64), 0, stream, a);
        VS_HIP(hipGetLastError());
        return 0;
    }
    dim3 grid((unsigned)vs::cdiv64(npix, 256)), block(256);
    if (in_dtype == 0) hipLaunchKernelGGL(adapter_kernel<0>, grid, block, 0, stream, a);
    else if (in_dtype == 1) hipLaunchKernelGGL(adapter_kernel<1>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(adapter_kernel<2>, grid, block, 0, stream, a);
    VS_HIP(hipGetLastError());
    return 0;
}

// Backward of vs_gaussian_adapter for dense NHWC 16-bit head outputs (the training path): gradients of means [npix,3],
// covariances [npix,3,3], harmonics [npix,3,d_sh], opacities [npix] and, optionally, of the raw output [npix, 11+3 d_sh]
// -> d_pts [npix, d_pts_ld] and d_gs [npix, d_gs_ld] in the inputs' 16-bit dtype (row strides >= the channel counts; channels of pts
// beyond 3 and all padding columns get zero, so a consumer can read 16-byte aligned rows).
extern "C" int vs_gaussian_adapter_backward(const void *pts, int32_t pts_pix, const void *gs, int32_t in_dtype, int64_t npix, int32_t d_sh,
                                            const float *sh_mask, int32_t scale_act, float scale_min, float scale_max,
                                            float opacity_exponent, const float *d_means, const float *d_cov, const float *d_harmonics,
                                            const float *d_opacities, const float *d_raw, void *d_pts, int32_t d_pts_ld, void *d_gs,
                                            int32_t d_gs_ld, vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(pts && gs && sh_mask && d_means && d_cov && d_harmonics && d_opacities && d_pts && d_gs, "vs_gaussian_adapter_backward: null pointer");
    VS_CHECK(in_dtype >= 0 && in_dtype <= 2 && d_sh > 0 && 11 + 3 * d_sh <= kMaxCh && scale_act >= 0 && scale_act <= 2 && pts_pix >= 3,
             "vs_gaussian_adapter_backward: bad argument (in_dtype 0 f32 / 1 f16 / 2 bf16, 11 + 3 d_sh <= %d)", kMaxCh);
    VS_CHECK(((uintptr_t)d_gs & 15) == 0, "vs_gaussian_adapter_backward: d_gs must be 16-byte aligned");
    VS_CHECK(d_pts_ld >= pts_pix && d_gs_ld >= 8 + 3 * d_sh && d_gs_ld <= kMaxCh, "vs_gaussian_adapter_backward: output row strides must cover the channels (d_gs_ld <= %d)", kMaxCh);
    if (npix <= 0) return 0;
    AdapterBwdArgs a{pts, gs, pts_pix, npix, d_sh, sh_mask, scale_act, scale_min, scale_max, opacity_exponent, d_means, d_cov, d_harmonics,
                     d_opacities, d_raw, d_pts, d_gs, d_pts_ld, d_gs_ld};
    dim3 grid((unsigned)vs::cdiv64(npix, 64));
    if (in_dtype == 2) hipLaunchKernelGGL(adapter_backward_kernel<2>, grid, dim3(64), 0, stream, a);
    else if (in_dtype == 1) hipLaunchKernelGGL(adapter_backward_kernel<1>, grid, dim3(64), 0, stream, a);
    else hipLaunchKernelGGL(adapter_backward_kernel<0>, grid, dim3(64), 0, stream, a);
    VS_HIP(hipGetLastError());
    return 0;
}
