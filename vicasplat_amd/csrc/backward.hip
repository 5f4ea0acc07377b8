// Building blocks of the ENCODER backward pass (training_step, model_wrapper.py:184-321, where the
// reference relies on torch autograd).  Wired into torch.autograd.Functions by vicasplat_amd/autograd.py; each one is a complete,
// parity-tested operator:
//
//   vs_transpose16        [R,C] -> [C,Rpad] 16-bit, zero padded: feeds the existing NT GEMM kernels for
//                         dX = dY W   (A = dY [M,N], "W" = W^T [K,N])      and
//                         dW = dY^T X (A = dY^T [N,M], "W" = X^T [K,M], f32 store) of nn.Linear (croco/blocks.py:60-112)
//   vs_colsum             db[n] = sum_m dY[m,n]                                           (16-bit or f32 input, f32 output)
//   vs_gelu_backward      dz = dy * gelu'(z), exact-erf GELU (nn.GELU(), blocks.py:60)
//   vs_layernorm_backward nn.LayerNorm(eps) followed by the AdaLN modulation of backbone_vica.py:268-273:
//                         out = (xhat * w + b) * (1 + scale[g]) + shift[g];  dx, dw, db, dscale, dshift
#include "common.h"

#include <algorithm>

namespace {

template <bool BF16>
__device__ __forceinline__ float ld16(unsigned short h) {
    if constexpr (BF16) return __uint_as_float(((unsigned)h) << 16);
    else return (float)*reinterpret_cast<_Float16 *>(&h);
}
template <bool BF16>
__device__ __forceinline__ unsigned short st16(float v) {
    if constexpr (BF16) {
        unsigned u = __float_as_uint(v);
        u += 0x7FFFu + ((u >> 16) & 1u);
        return (unsigned short)(u >> 16);
    } else {
        _Float16 h = (_Float16)v;
        return *reinterpret_cast<unsigned short *>(&h);
    }
}

// 64 x 64 tile through LDS (65-half rows: conflict-free both ways); rows >= R of the output padding are written as zeros.
// CS = 1 / 2 (f16 / bf16): the column sums of the input (the bias gradient of the layer whose dY is being transposed for its
// weight-gradient GEMM) are accumulated on the way through -- one f32 atomic per column and block, no second pass over dY.
// bH > 0: the R input rows are the pixels of a ZERO-BORDERED NHWC map [n, bH+2, bW+2] whose interior is read from the
// unpadded [n, bH, bW, C] tensor `in` (border rows read as zero): the conv weight gradient's operands are produced
// straight from the activations, without a padded copy.  relu: negative inputs are written as zero (activation-before-conv).
// Slice-blocked output (gridDim.z slices of SL rows + `halo` rows of the neighbouring slices on either side): slice z holds
// the transposed rows [z SL - halo, (z+1) SL + halo) at out + z * slice_stride, i.e. out is [slice][C][SL + 2 halo] -- the
// operand layout of vs_gemm_wgrad (one K slice per block; the halo is what the shifted tap views of a 3x3 weight gradient read
// beyond their slice).
template <int CS>
__global__ void __launch_bounds__(256)
transpose16_kernel(const unsigned short *__restrict__ in, long long ld_in, unsigned short *__restrict__ out, long long ld_out,
                   int R, int C, int SL, int halo, long long slice_stride, float *__restrict__ colsum, int bH, int bW, int relu) {
    // 64 x 64 tile; every thread moves 4 consecutive elements (8 bytes) per access in both directions: rows of 68 halfs keep the
    // 8-byte LDS accesses aligned.  Pass p: thread (q = t & 15, rr = t >> 4) reads input row p*16 + rr, columns 4q..4q+3, and
    // later writes output row (= input column) p*16 + rr, slice-local rows 4q..4q+3.
    __shared__ __attribute__((aligned(16))) unsigned short tile[64][68];
    [[maybe_unused]] __shared__ float part[16][64];
    const int l0 = blockIdx.y * 64, c0 = blockIdx.x * 64;       // l0: first slice-local row of this block
    const int r0 = (int)blockIdx.z * SL - halo + l0;            // ... and the input row it corresponds to (may be < 0)
    const int Rpad = SL + 2 * halo;                              // slice-local rows to write
    out += (long long)blockIdx.z * slice_stride;
    const int q = threadIdx.x & 15, rr = threadIdx.x >> 4;
    const bool vec_in = (ld_in & 3) == 0 && (reinterpret_cast<uintptr_t>(in) & 7) == 0 && c0 + 64 <= C;
    [[maybe_unused]] float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int i = p * 16 + rr, r = r0 + i, c = c0 + 4 * q;
        long long src = r;
        bool ok = r >= 0 && r < R;
        if (bH > 0 && ok) {
            const int Wp = bW + 2, Hp = bH + 2;
            const int n = r / (Hp * Wp), rem = r - n * (Hp * Wp);
            const int y = rem / Wp - 1, x = rem - (rem / Wp) * Wp - 1;
            ok = y >= 0 && y < bH && x >= 0 && x < bW;
            src = ((long long)n * bH + y) * bW + x;
        }
        unsigned short v[4] = {0, 0, 0, 0};
        if (ok) {
            const unsigned short *ip = in + src * ld_in + c;
            if (vec_in) {
                const uint2 u = *reinterpret_cast<const uint2 *>(ip);
                v[0] = (unsigned short)(u.x & 0xffffu); v[1] = (unsigned short)(u.x >> 16);
                v[2] = (unsigned short)(u.y & 0xffffu); v[3] = (unsigned short)(u.y >> 16);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c + e < C) v[e] = ip[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (relu && (v[e] & 0x8000u)) v[e] = 0;
            if constexpr (CS != 0) s[e] += ld16<CS == 2>(v[e]);
        }
        uint2 w;
        w.x = (unsigned)v[0] | ((unsigned)v[1] << 16);
        w.y = (unsigned)v[2] | ((unsigned)v[3] << 16);
        *reinterpret_cast<uint2 *>(&tile[i][4 * q]) = w;
    }
    if constexpr (CS != 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) part[rr][4 * q + e] = s[e];
    }
    __syncthreads();
    if constexpr (CS != 0) {
        if (threadIdx.x < 64 && c0 + (int)threadIdx.x < C && r0 < R && r0 + 64 > 0) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) a += part[k][threadIdx.x];
            unsafeAtomicAdd(colsum + c0 + threadIdx.x, a);
        }
    }
    const bool vec_out = (ld_out & 3) == 0 && (slice_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int cl = p * 16 + rr, c = c0 + cl, l = l0 + 4 * q;
        if (c >= C || l >= Rpad) continue;
        unsigned short v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = tile[4 * q + e][cl];
        unsigned short *op = out + (long long)c * ld_out + l;
        if (vec_out && l + 3 < Rpad) {
            uint2 w;
            w.x = (unsigned)v[0] | ((unsigned)v[1] << 16);
            w.y = (unsigned)v[2] | ((unsigned)v[3] << 16);
            *reinterpret_cast<uint2 *>(op) = w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (l + e < Rpad) op[e] = v[e];
        }
    }
}

// column sums: block = 64 columns x 4 row lanes; rows strided over gridDim.y blocks; one f32 atomic per (block, column)
template <int DT>  // 0 f32, 1 f16, 2 bf16
__global__ void __launch_bounds__(256)
colsum_kernel(const void *__restrict__ x, long long ld, float *__restrict__ out, int M, int N) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), ty = threadIdx.x >> 6;
    float s = 0.f;
    if (c < N) {
        for (int m = blockIdx.y * 4 + ty; m < M; m += gridDim.y * 4) {
            if constexpr (DT == 0) s += reinterpret_cast<const float *>(x)[(long long)m * ld + c];
            else s += ld16<DT == 2>(reinterpret_cast<const unsigned short *>(x)[(long long)m * ld + c]);
        }
    }
    part[ty][threadIdx.x & 63] = s;
    __syncthreads();
    if (ty == 0 && c < N) unsafeAtomicAdd(out + c, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// 16-bit fast path (N % 8 == 0, 16-byte aligned rows): thread = 8 consecutive columns (one 16-byte load per row), block = 16 column
// groups (128 columns, 256 contiguous bytes per row) x 16 row lanes; rows strided over gridDim.y blocks; LDS reduction over the row
// lanes, one f32 atomic per (block, column).
template <bool BF16>
__global__ void __launch_bounds__(256)
colsum16_kernel(const unsigned short *__restrict__ x, long long ld, float *__restrict__ out, int M, int N) {
    __shared__ float part[16][129];
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 128 + cg * 8;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < N) {
        // four rows per trip: four independent 16-byte loads in flight per thread (one per trip ran at 2.75 TB/s: 12 ms of column sums per
        // 24-scene f16 training step)
        const long long step = (long long)gridDim.y * 16;
        long long m = blockIdx.y * 16 + rl;
        auto add = [&](const uint4 v) {
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s[2 * k] += ld16<BF16>((unsigned short)(w[k] & 0xffffu));
                s[2 * k + 1] += ld16<BF16>((unsigned short)(w[k] >> 16));
            }
        };
        for (; m + 3 * step < M; m += 4 * step) {
            const uint4 v0 = *reinterpret_cast<const uint4 *>(x + m * ld + c), v1 = *reinterpret_cast<const uint4 *>(x + (m + step) * ld + c);
            const uint4 v2 = *reinterpret_cast<const uint4 *>(x + (m + 2 * step) * ld + c), v3 = *reinterpret_cast<const uint4 *>(x + (m + 3 * step) * ld + c);
            add(v0); add(v1); add(v2); add(v3);
        }
        for (; m < M; m += step) add(*reinterpret_cast<const uint4 *>(x + m * ld + c));
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) part[rl][cg * 8 + k] = s[k];
    __syncthreads();
    if (threadIdx.x < 128 && blockIdx.x * 128 + threadIdx.x < N) {
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) a += part[r][threadIdx.x];
        unsafeAtomicAdd(out + blockIdx.x * 128 + threadIdx.x, a);
    }
}

// f32 fast path (N % 4 == 0, 16-byte aligned rows; the bias gradients of the split-class backward): thread = 4 consecutive columns (one
// 16-byte load per row), block = 32 column groups (128 columns, 512 contiguous bytes per row) x 8 row lanes, as colsum16_kernel.
__global__ void __launch_bounds__(256)
colsum32_kernel(const float *__restrict__ x, long long ld, float *__restrict__ out, int M, int N) {
    __shared__ float part[8][129];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 128 + cg * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < N) {
        const long long step = (long long)gridDim.y * 8;
        long long m = blockIdx.y * 8 + rl;
        for (; m + 3 * step < M; m += 4 * step) {      // four independent loads in flight (see colsum16_kernel)
            const float4 v0 = *reinterpret_cast<const float4 *>(x + m * ld + c), v1 = *reinterpret_cast<const float4 *>(x + (m + step) * ld + c);
            const float4 v2 = *reinterpret_cast<const float4 *>(x + (m + 2 * step) * ld + c), v3 = *reinterpret_cast<const float4 *>(x + (m + 3 * step) * ld + c);
            s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y);
            s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
        }
        for (; m < M; m += step) {
            const float4 v = *reinterpret_cast<const float4 *>(x + m * ld + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    part[rl][cg * 4] = s.x; part[rl][cg * 4 + 1] = s.y; part[rl][cg * 4 + 2] = s.z; part[rl][cg * 4 + 3] = s.w;
    __syncthreads();
    if (threadIdx.x < 128 && blockIdx.x * 128 + threadIdx.x < N) {
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) a += part[r][threadIdx.x];
        unsafeAtomicAdd(out + blockIdx.x * 128 + threadIdx.x, a);
    }
}

template <bool BF16>
__global__ void __launch_bounds__(256)
gelu_backward_kernel(const unsigned short *__restrict__ dy, const unsigned short *__restrict__ z, unsigned short *__restrict__ dz,
                     long long n) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= n) return;
    const uint4 a = *reinterpret_cast<const uint4 *>(dy + i), b = *reinterpret_cast<const uint4 *>(z + i);
    const unsigned aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
    unsigned r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float o[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float g = ld16<BF16>((unsigned short)(h ? aw[k] >> 16 : aw[k] & 0xffffu));
            const float x = ld16<BF16>((unsigned short)(h ? bw[k] >> 16 : bw[k] & 0xffffu));
            // d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
            const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
            const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
            o[h] = g * (cdf + x * pdf);
        }
        r[k] = (unsigned)st16<BF16>(o[0]) | ((unsigned)st16<BF16>(o[1]) << 16);
    }
    *reinterpret_cast<uint4 *>(dz + i) = make_uint4(r[0], r[1], r[2], r[3]);
}

// a = gelu_erf(z) (exact GELU, croco/blocks.py:60) on 16-bit elements: the training forward keeps the pre-activation z for the
// backward, so the activation is its own pass here instead of the GEMM epilogue
template <bool BF16>
__global__ void __launch_bounds__(256)
gelu16_kernel(const unsigned short *__restrict__ z, unsigned short *__restrict__ out, long long n) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= n) return;
    const uint4 b = *reinterpret_cast<const uint4 *>(z + i);
    const unsigned bw[4] = {b.x, b.y, b.z, b.w};
    unsigned r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float x0 = ld16<BF16>((unsigned short)(bw[k] & 0xffffu)), x1 = ld16<BF16>((unsigned short)(bw[k] >> 16));
        const float o0 = 0.5f * x0 * (1.0f + erff(x0 * 0.70710678118654752440f));
        const float o1 = 0.5f * x1 * (1.0f + erff(x1 * 0.70710678118654752440f));
        r[k] = (unsigned)st16<BF16>(o0) | ((unsigned)st16<BF16>(o1) << 16);
    }
    *reinterpret_cast<uint4 *>(out + i) = make_uint4(r[0], r[1], r[2], r[3]);
}

// out = x > 0 ? dy : 0 on packed 16-bit pairs (backward of a ReLU whose INPUT x was saved); out may be dy (in place)
__global__ void __launch_bounds__(256)
relu_mask16_kernel(const unsigned short *dy, const unsigned short *__restrict__ x, unsigned short *dx, long long n) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= n) return;
    uint4 d = *reinterpret_cast<const uint4 *>(dy + i);
    const uint4 v = *reinterpret_cast<const uint4 *>(x + i);
    auto m = [](unsigned dd, unsigned xx) {  // keep a half iff x is positive: sign bit clear and not (+/-)zero
        const unsigned lo = ((xx & 0x8000u) == 0 && (xx & 0x7fffu) != 0) ? 0xffffu : 0u;
        const unsigned hi = ((xx & 0x80000000u) == 0 && (xx & 0x7fff0000u) != 0) ? 0xffff0000u : 0u;
        return dd & (lo | hi);
    };
    d.x = m(d.x, v.x); d.y = m(d.y, v.y); d.z = m(d.z, v.z); d.w = m(d.w, v.w);
    *reinterpret_cast<uint4 *>(dx + i) = d;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One wave per row, lane owns columns lane*4 + 256*k (k < NV).  Per row: recompute mean / rstd, then
//   g  = dout * (1 + scale) * w,   dx = rstd * (g - mean(g) - xhat * mean(g * xhat))
// Work split: blockIdx.y = modulation group (rows [g*mod_rows, (g+1)*mod_rows); one group = all rows without modulation),
// blockIdx.x = chunk of that group; the block's 4 waves interleave the chunk's rows.  Column sums (dw, db) and the group's
// (dscale, dshift) stay in registers across the wave's rows, are summed over the block's waves through LDS and leave the block
// as ONE f32 atomic per column -- the host picks ~512 blocks, so the atomic traffic stays at a few MB per call.
constexpr int kLnVec = 8;  // C <= 64 * 4 * 8 = 2048
template <int DT, int NV, bool MOD>  // dout dtype: 0 f32, 1 f16, 2 bf16; NV * 256 >= C; MOD: scale / shift modulation present
__global__ void __launch_bounds__(256)
layernorm_backward_kernel(const void *__restrict__ dout, long long ld_do, const float *__restrict__ x, long long ldx,
                          const float *__restrict__ w, const float *__restrict__ b, const float *__restrict__ scale, int mod_rows,
                          int mod_ld, float *dx, long long ld_dx, const float *dx_add, long long ld_add,   // (dx_add may alias dx)
                          unsigned short *__restrict__ dx16, long long ld_dx16, int dx16_bf16, float *__restrict__ dw,
                          float *__restrict__ db, float *__restrict__ dscale, float *__restrict__ dshift, int M, int C, float eps,
                          int grp_in, int grp_out, int grp_off, int rows_per_chunk) {
    __shared__ float red[4][NV * 256];
    const int lane = threadIdx.x & 63, wv_id = threadIdx.x >> 6;
    const int gidx = blockIdx.y;
    const int g_lo = gidx * mod_rows, g_hi = min(M, g_lo + mod_rows);
    const int m_lo = g_lo + blockIdx.x * rows_per_chunk, m_hi = min(g_hi, m_lo + rows_per_chunk);
    if (m_lo >= m_hi) return;
    float4 wv[NV], bvv[NV], sc[NV], aw[NV], ab[NV];
    [[maybe_unused]] float4 as[NV], ah[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = (lane + 64 * k) * 4;
        wv[k] = c < C ? *reinterpret_cast<const float4 *>(w + c) : make_float4(0, 0, 0, 0);
        bvv[k] = c < C ? *reinterpret_cast<const float4 *>(b + c) : make_float4(0, 0, 0, 0);
        sc[k] = make_float4(0, 0, 0, 0);
        if constexpr (MOD) {
            if (c < C) sc[k] = *reinterpret_cast<const float4 *>(scale + (long long)gidx * mod_ld + c);
            as[k] = ah[k] = make_float4(0, 0, 0, 0);
        }
        aw[k] = ab[k] = make_float4(0, 0, 0, 0);
    }
    for (int m = m_lo + wv_id; m < m_hi; m += 4) {
        const float *xr = x + (long long)m * ldx;
        const long long orow = (long long)(m / grp_in) * grp_out + grp_off + (m % grp_in);  // row of dout (the forward's output row)
        float4 xv[NV], gv[NV];
        float d[NV][4];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = (lane + 64 * k) * 4;
            xv[k] = c < C ? *reinterpret_cast<const float4 *>(xr + c) : make_float4(0, 0, 0, 0);
            s += xv[k].x + xv[k].y + xv[k].z + xv[k].w;
            d[k][0] = d[k][1] = d[k][2] = d[k][3] = 0.f;
            if (c < C) {  // issue the dout loads before the first reduction: they do not depend on it
                if constexpr (DT == 0) {
                    const float4 t = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(dout) + orow * ld_do + c);
                    d[k][0] = t.x; d[k][1] = t.y; d[k][2] = t.z; d[k][3] = t.w;
                } else {
                    const uint2 t = *reinterpret_cast<const uint2 *>(reinterpret_cast<const unsigned short *>(dout) + orow * ld_do + c);
                    d[k][0] = ld16<DT == 2>((unsigned short)(t.x & 0xffffu)); d[k][1] = ld16<DT == 2>((unsigned short)(t.x >> 16));
                    d[k][2] = ld16<DT == 2>((unsigned short)(t.y & 0xffffu)); d[k][3] = ld16<DT == 2>((unsigned short)(t.y >> 16));
                }
            }
        }
        const float mean = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = (lane + 64 * k) * 4;
            if (c < C) {
                xv[k].x -= mean; xv[k].y -= mean; xv[k].z -= mean; xv[k].w -= mean;
                q += xv[k].x * xv[k].x + xv[k].y * xv[k].y + xv[k].z * xv[k].z + xv[k].w * xv[k].w;
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = (lane + 64 * k) * 4;
            gv[k] = make_float4(0, 0, 0, 0);
            if (c < C) {
                const float xh[4] = {xv[k].x * rstd, xv[k].y * rstd, xv[k].z * rstd, xv[k].w * rstd};
                const float ww[4] = {wv[k].x, wv[k].y, wv[k].z, wv[k].w}, bb[4] = {bvv[k].x, bvv[k].y, bvv[k].z, bvv[k].w};
                const float ss[4] = {sc[k].x, sc[k].y, sc[k].z, sc[k].w};
                float gg[4];
                float *paw = &aw[k].x, *pab = &ab[k].x;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dy = d[k][e] * (1.0f + ss[e]);     // gradient w.r.t. the LN output before modulation
                    if constexpr (MOD) {
                        (&as[k].x)[e] += d[k][e] * (xh[e] * ww[e] + bb[e]);
                        (&ah[k].x)[e] += d[k][e];
                    }
                    paw[e] += dy * xh[e];
                    pab[e] += dy;
                    gg[e] = dy * ww[e];
                    sg += gg[e];
                    sgx += gg[e] * xh[e];
                }
                gv[k] = make_float4(gg[0], gg[1], gg[2], gg[3]);
            }
        }
        const float mg = wave_sum(sg) / (float)C, mgx = wave_sum(sgx) / (float)C;
        float *dxr = dx + (long long)m * ld_dx;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = (lane + 64 * k) * 4;
            if (c < C) {
                float4 r;
                r.x = rstd * (gv[k].x - mg - xv[k].x * rstd * mgx);
                r.y = rstd * (gv[k].y - mg - xv[k].y * rstd * mgx);
                r.z = rstd * (gv[k].z - mg - xv[k].z * rstd * mgx);
                r.w = rstd * (gv[k].w - mg - xv[k].w * rstd * mgx);
                if (dx_add) {   // gradient arriving through the residual connection (dx_add may be dx itself: in-place accumulate)
                    const float4 o = *reinterpret_cast<const float4 *>(dx_add + (long long)m * ld_add + c);
                    r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
                }
                *reinterpret_cast<float4 *>(dxr + c) = r;
                if (dx16) {     // 16-bit copy for the GEMMs that consume this gradient next (no separate cast pass)
                    uint2 pk;
                    if (dx16_bf16) {
                        pk.x = (unsigned)st16<true>(r.x) | ((unsigned)st16<true>(r.y) << 16);
                        pk.y = (unsigned)st16<true>(r.z) | ((unsigned)st16<true>(r.w) << 16);
                    } else {
                        pk.x = (unsigned)st16<false>(r.x) | ((unsigned)st16<false>(r.y) << 16);
                        pk.y = (unsigned)st16<false>(r.z) | ((unsigned)st16<false>(r.w) << 16);
                    }
                    *reinterpret_cast<uint2 *>(dx16 + (long long)m * ld_dx16 + c) = pk;
                }
            }
        }
    }
    // block reduction of the column sums, then one atomic per column
    auto flush = [&](float4 (&acc)[NV], float *dst) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NV; ++k) *reinterpret_cast<float4 *>(&red[wv_id][(lane + 64 * k) * 4]) = acc[k];
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) unsafeAtomicAdd(dst + c, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
    };
    flush(aw, dw);
    flush(ab, db);
    if constexpr (MOD) {
        flush(as, dscale + (long long)gidx * mod_ld);
        flush(ah, dshift + (long long)gidx * mod_ld);
    }
}

// Gated residual update of the decoder's image stream (backbone_vica.py:274-278,302,327,331):
//   out[m, c] = x[m, c] + (1 + gate[m / gate_rows, c]) * y[yrow(m), c],   yrow(m) = (m / grp_in) * grp_out + grp_off + m % grp_in
// x, out f32 (may alias), y 16-bit (the branch output, possibly interleaved with other rows: the camera token in front of each
// frame's image tokens), gate f32 [G, C] or null.  One pass instead of cast + multiply + add.
template <bool BF16>
__global__ void __launch_bounds__(256)
gated_resid_kernel(const float *__restrict__ x, const unsigned short *__restrict__ y, long long ldy, const float *__restrict__ gate,
                   int gate_rows, float *__restrict__ out, int M, int C, int grp_in, int grp_out, int grp_off) {
    const int C4 = C >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)M * C4) return;
    const int m = (int)(idx / C4), c = (int)(idx - (long long)m * C4) * 4;
    const long long yr = (long long)(m / grp_in) * grp_out + grp_off + (m % grp_in);
    const float4 xv = *reinterpret_cast<const float4 *>(x + (long long)m * C + c);
    const uint2 t = *reinterpret_cast<const uint2 *>(y + yr * ldy + c);
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gate) g = *reinterpret_cast<const float4 *>(gate + (long long)(m / gate_rows) * C + c);
    float4 o;
    o.x = xv.x + (1.0f + g.x) * ld16<BF16>((unsigned short)(t.x & 0xffffu));
    o.y = xv.y + (1.0f + g.y) * ld16<BF16>((unsigned short)(t.x >> 16));
    o.z = xv.z + (1.0f + g.z) * ld16<BF16>((unsigned short)(t.y & 0xffffu));
    o.w = xv.w + (1.0f + g.w) * ld16<BF16>((unsigned short)(t.y >> 16));
    *reinterpret_cast<float4 *>(out + (long long)m * C + c) = o;
}

// Backward of the branch side: dy[yrow(m), c] = dout[m, c] * (1 + gate) in 16 bits and dgate[g, c] += sum_m dout[m, c] * y[yrow(m), c]
// (dx = dout needs no kernel).  blockIdx.y = gate group, blockIdx.x = chunk of its rows; wave per row, lane owns columns
// lane*4 + 256 k; the group sums leave the block as one f32 atomic per column (as in layernorm_backward_kernel).
template <bool BF16, int NV>
__global__ void __launch_bounds__(256)
gated_resid_backward_kernel(const float *__restrict__ dout, const unsigned short *__restrict__ y, long long ldy,
                            const float *__restrict__ gate, int gate_rows, unsigned short *__restrict__ dy, long long lddy,
                            float *__restrict__ dgate, int M, int C, int grp_in, int grp_out, int grp_off, int rows_per_chunk) {
    __shared__ float red[4][NV * 256];
    const int lane = threadIdx.x & 63, wv_id = threadIdx.x >> 6;
    const int gidx = blockIdx.y;
    const int g_lo = gidx * gate_rows, g_hi = min(M, g_lo + gate_rows);
    const int m_lo = g_lo + blockIdx.x * rows_per_chunk, m_hi = min(g_hi, m_lo + rows_per_chunk);
    if (m_lo >= m_hi) return;
    float4 gt[NV], acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = (lane + 64 * k) * 4;
        gt[k] = (gate && c < C) ? *reinterpret_cast<const float4 *>(gate + (long long)gidx * C + c) : make_float4(0, 0, 0, 0);
        acc[k] = make_float4(0, 0, 0, 0);
    }
    for (int m = m_lo + wv_id; m < m_hi; m += 4) {
        const long long yr = (long long)(m / grp_in) * grp_out + grp_off + (m % grp_in);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = (lane + 64 * k) * 4;
            if (c >= C) continue;
            const float4 d = *reinterpret_cast<const float4 *>(dout + (long long)m * C + c);
            if (gate) {
                const uint2 t = *reinterpret_cast<const uint2 *>(y + yr * ldy + c);
                acc[k].x += d.x * ld16<BF16>((unsigned short)(t.x & 0xffffu)); acc[k].y += d.y * ld16<BF16>((unsigned short)(t.x >> 16));
                acc[k].z += d.z * ld16<BF16>((unsigned short)(t.y & 0xffffu)); acc[k].w += d.w * ld16<BF16>((unsigned short)(t.y >> 16));
            }
            uint2 o;
            o.x = (unsigned)st16<BF16>(d.x * (1.0f + gt[k].x)) | ((unsigned)st16<BF16>(d.y * (1.0f + gt[k].y)) << 16);
            o.y = (unsigned)st16<BF16>(d.z * (1.0f + gt[k].z)) | ((unsigned)st16<BF16>(d.w * (1.0f + gt[k].w)) << 16);
            *reinterpret_cast<uint2 *>(dy + yr * lddy + c) = o;
        }
    }
    if (!gate) return;
#pragma unroll
    for (int k = 0; k < NV; ++k) *reinterpret_cast<float4 *>(&red[wv_id][(lane + 64 * k) * 4]) = acc[k];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) unsafeAtomicAdd(dgate + (long long)gidx * C + c, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
}

}  // namespace

namespace {
int transpose_entry(const char *fn, const void *in, int64_t ld_in, void *out, int64_t ld_out, int32_t R, int32_t C, int32_t Rpad,
                    float *colsum, int32_t dtype, int32_t bH, int32_t bW, int32_t relu, int32_t nslices, int32_t halo,
                    int64_t slice_stride, hipStream_t stream) {
    VS_CHECK(in && out, "%s: null pointer", fn);
    VS_CHECK(R >= 0 && C >= 0 && Rpad >= R && ld_in >= C, "%s: bad sizes R=%d C=%d Rpad=%d", fn, R, C, Rpad);
    VS_CHECK((bH == 0 && bW == 0) || (bH > 0 && bW > 0 && R % ((bH + 2) * (bW + 2)) == 0),
             "%s: with a border, R=%d must be a whole number of (H+2) x (W+2) maps (H=%d W=%d)", fn, R, bH, bW);
    if (nslices < 1) nslices = 1;
    VS_CHECK(nslices <= 65535 && Rpad % nslices == 0 && halo >= 0, "%s: Rpad=%d must be a multiple of nslices=%d (<= 65535), halo >= 0", fn, Rpad, nslices);
    VS_CHECK(nslices > 1 || halo == 0, "%s: a halo needs nslices > 1", fn);
    const int SL = Rpad / nslices;
    VS_CHECK(ld_out >= SL + 2 * halo, "%s: ld_out=%lld < slice length + 2 halo = %d", fn, (long long)ld_out, SL + 2 * halo);
    VS_CHECK(!(colsum && halo), "%s: column sums need halo == 0 (rows would be counted twice)", fn);
    if (C == 0) return 0;
    if (colsum) {
        VS_CHECK(dtype == 1 || dtype == 2, "%s: dtype must be 1 (f16) or 2 (bf16)", fn);
        VS_HIP(hipMemsetAsync(colsum, 0, (size_t)C * sizeof(float), stream));
    }
    if (Rpad == 0) return 0;
    dim3 grid(vs::cdiv(C, 64), vs::cdiv(SL + 2 * halo, 64), nslices), block(256);
    VS_CHECK(grid.y <= 65535u, "%s: too many rows per slice", fn);
    const unsigned short *ip = (const unsigned short *)in;
    unsigned short *op = (unsigned short *)out;
    if (!colsum) hipLaunchKernelGGL(transpose16_kernel<0>, grid, block, 0, stream, ip, (long long)ld_in, op, (long long)ld_out, R, C, SL, halo, (long long)slice_stride, colsum, bH, bW, relu);
    else if (dtype == 2) hipLaunchKernelGGL(transpose16_kernel<2>, grid, block, 0, stream, ip, (long long)ld_in, op, (long long)ld_out, R, C, SL, halo, (long long)slice_stride, colsum, bH, bW, relu);
    else hipLaunchKernelGGL(transpose16_kernel<1>, grid, block, 0, stream, ip, (long long)ld_in, op, (long long)ld_out, R, C, SL, halo, (long long)slice_stride, colsum, bH, bW, relu);
    VS_HIP(hipGetLastError());
    return 0;
}
}  // namespace

extern "C" int vs_transpose16(const void *in, int64_t ld_in, void *out, int64_t ld_out, int32_t R, int32_t C, int32_t Rpad,
                              vs_stream_t stream_) {
    return transpose_entry("vs_transpose16", in, ld_in, out, ld_out, R, C, Rpad, nullptr, 0, 0, 0, 0, 1, 0, 0, (hipStream_t)stream_);
}

extern "C" int vs_transpose16_ex(const void *in, int64_t ld_in, void *out, int64_t ld_out, int32_t R, int32_t C, int32_t Rpad,
                                 float *colsum, int32_t dtype, int32_t border_h, int32_t border_w, int32_t relu, int32_t nslices,
                                 int32_t halo, int64_t slice_stride, vs_stream_t stream_) {
    return transpose_entry("vs_transpose16_ex", in, ld_in, out, ld_out, R, C, Rpad, colsum, dtype, border_h, border_w, relu, nslices, halo,
                           slice_stride, (hipStream_t)stream_);
}

extern "C" int vs_colsum(const void *x, int64_t ld, float *out, int32_t M, int32_t N, int32_t dtype, vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(x && out, "vs_colsum: null pointer");
    VS_CHECK(dtype >= 0 && dtype <= 2, "vs_colsum: dtype must be 0 (f32), 1 (f16) or 2 (bf16)");
    VS_CHECK(M >= 0 && N > 0 && ld >= N, "vs_colsum: bad sizes");
    VS_HIP(hipMemsetAsync(out, 0, (size_t)N * sizeof(float), stream));
    if (M == 0) return 0;
    if (dtype != 0 && N % 8 == 0 && ld % 8 == 0 && ((uintptr_t)x & 15) == 0) {   // 16-byte loads
        const int gx = vs::cdiv(N, 128);
        dim3 g16(gx, std::max(1, std::min(std::max(1, 2048 / gx), vs::cdiv(M, 64)))), block(256);
        if (dtype == 2) hipLaunchKernelGGL(colsum16_kernel<true>, g16, block, 0, stream, (const unsigned short *)x, (long long)ld, out, M, N);
        else hipLaunchKernelGGL(colsum16_kernel<false>, g16, block, 0, stream, (const unsigned short *)x, (long long)ld, out, M, N);
        VS_HIP(hipGetLastError());
        return 0;
    }
    if (dtype == 0 && N % 4 == 0 && ld % 4 == 0 && ((uintptr_t)x & 15) == 0) {   // 16-byte loads
        const int gx = vs::cdiv(N, 128);
        dim3 g32(gx, std::max(1, std::min(std::max(1, 2048 / gx), vs::cdiv(M, 32)))), block(256);
        hipLaunchKernelGGL(colsum32_kernel, g32, block, 0, stream, (const float *)x, (long long)ld, out, M, N);
        VS_HIP(hipGetLastError());
        return 0;
    }
    dim3 grid(vs::cdiv(N, 64), std::min(std::max(256, 2048 / vs::cdiv(N, 64)), vs::cdiv(M, 64))), block(256);
    if (dtype == 0) hipLaunchKernelGGL(colsum_kernel<0>, grid, block, 0, stream, x, (long long)ld, out, M, N);
    else if (dtype == 1) hipLaunchKernelGGL(colsum_kernel<1>, grid, block, 0, stream, x, (long long)ld, out, M, N);
    else hipLaunchKernelGGL(colsum_kernel<2>, grid, block, 0, stream, x, (long long)ld, out, M, N);
    VS_HIP(hipGetLastError());
    return 0;
}

extern "C" int vs_gelu_backward(const void *dy, const void *z, void *dz, int64_t n, int32_t dtype, vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(dy && z && dz, "vs_gelu_backward: null pointer");
    VS_CHECK(dtype == 1 || dtype == 2, "vs_gelu_backward: dtype must be 1 (f16) or 2 (bf16)");
    VS_CHECK(n >= 0 && n % 8 == 0, "vs_gelu_backward: n=%lld must be a multiple of 8", (long long)n);
    VS_CHECK((((uintptr_t)dy | (uintptr_t)z | (uintptr_t)dz) & 15) == 0, "vs_gelu_backward: 16-byte alignment required");
    if (n == 0) return 0;
    dim3 grid((unsigned)vs::cdiv64(n / 8, 256)), block(256);
    if (dtype == 2) hipLaunchKernelGGL(gelu_backward_kernel<true>, grid, block, 0, stream, (const unsigned short *)dy, (const unsigned short *)z, (unsigned short *)dz, (long long)n);
    else hipLaunchKernelGGL(gelu_backward_kernel<false>, grid, block, 0, stream, (const unsigned short *)dy, (const unsigned short *)z, (unsigned short *)dz, (long long)n);
    VS_HIP(hipGetLastError());
    return 0;
}

extern "C" int vs_gelu16(const void *z, void *out, int64_t n, int32_t dtype, vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(z && out, "vs_gelu16: null pointer");
    VS_CHECK(dtype == 1 || dtype == 2, "vs_gelu16: dtype must be 1 (f16) or 2 (bf16)");
    VS_CHECK(n >= 0 && n % 8 == 0, "vs_gelu16: n=%lld must be a multiple of 8", (long long)n);
    VS_CHECK((((uintptr_t)z | (uintptr_t)out) & 15) == 0, "vs_gelu16: 16-byte alignment required");
    if (n == 0) return 0;
    dim3 grid((unsigned)vs::cdiv64(n / 8, 256)), block(256);
    if (dtype == 2) hipLaunchKernelGGL(gelu16_kernel<true>, grid, block, 0, stream, (const unsigned short *)z, (unsigned short *)out, (long long)n);
    else hipLaunchKernelGGL(gelu16_kernel<false>, grid, block, 0, stream, (const unsigned short *)z, (unsigned short *)out, (long long)n);
    VS_HIP(hipGetLastError());
    return 0;
}

namespace {
int layernorm_backward_entry(const void *dout, int64_t ld_do, int32_t do_dtype, const float *x, int64_t ldx, const float *w, const float *b,
                             const float *scale, int32_t mod_rows, int32_t mod_ld, float *dx, int64_t ld_dx, const float *dx_add,
                             int64_t ld_add, void *dx16v, int64_t ld16, int32_t dx16_dtype, float *dw, float *db, float *dscale,
                             float *dshift, int32_t M, int32_t C, float eps, int32_t grp_in, int32_t grp_out, int32_t grp_off,
                             vs_stream_t stream_) {
    unsigned short *dx16 = (unsigned short *)dx16v;
    const int dx16_bf16 = dx16_dtype == 2;
    VS_CHECK(!dx16 || ((dx16_dtype == 1 || dx16_dtype == 2) && ld16 % 4 == 0 && ((uintptr_t)dx16 & 7) == 0), "vs_layernorm_backward: bad 16-bit gradient output");
    VS_CHECK(!dx_add || ld_add % 4 == 0, "vs_layernorm_backward: ld_add must be a multiple of 4");
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(dout && x && w && b && dx && dw && db, "vs_layernorm_backward: null pointer");
    VS_CHECK(C > 0 && C % 4 == 0 && C <= 64 * 4 * kLnVec, "vs_layernorm_backward: C=%d must be a multiple of 4 and <= %d", C, 64 * 4 * kLnVec);
    VS_CHECK(do_dtype >= 0 && do_dtype <= 2, "vs_layernorm_backward: bad dout dtype %d", do_dtype);
    VS_CHECK(ldx % 4 == 0 && ld_dx % 4 == 0 && ld_do % 4 == 0, "vs_layernorm_backward: row strides must be multiples of 4 elements");
    VS_CHECK(!scale || (dscale && dshift), "vs_layernorm_backward: scale given without dscale/dshift");
    if (M <= 0) return 0;
    if (grp_in <= 0) { grp_in = M; grp_out = M; grp_off = 0; }
    if (mod_rows <= 0) mod_rows = M;
    if (mod_ld <= 0) mod_ld = C;
    // ~512 blocks of 4 waves (2 resident blocks per CU): enough waves to hide the row reductions' latency, few enough that the
    // per-block column-sum atomics stay a few MB.  One grid row per modulation group.
    const int G = scale ? vs::cdiv(M, mod_rows) : 1;
    const int group_rows = scale ? mod_rows : M;
    if (!scale) mod_rows = M;
    const int chunks = std::max(1, std::min(512 / G, vs::cdiv(group_rows, 4)));
    const int rows_per_chunk = vs::cdiv(group_rows, chunks);
    VS_CHECK(G <= 65535, "vs_layernorm_backward: too many modulation groups (%d)", G);
    dim3 grid(vs::cdiv(group_rows, rows_per_chunk), G), block(256);
#define VS_LNB3(DT_, NV_, MOD_) hipLaunchKernelGGL((layernorm_backward_kernel<DT_, NV_, MOD_>), grid, block, 0, stream, dout, (long long)ld_do, x, \
                                      (long long)ldx, w, b, scale, mod_rows, mod_ld, dx, (long long)ld_dx, dx_add, (long long)ld_add, dx16, (long long)ld16, dx16_bf16, dw, db, dscale, dshift, M, C, eps, \
                                      grp_in, grp_out, grp_off, rows_per_chunk)
#define VS_LNB2(DT_, NV_) { if (scale) VS_LNB3(DT_, NV_, true); else VS_LNB3(DT_, NV_, false); }
#define VS_LNB(DT_) { if (C <= 256) VS_LNB2(DT_, 1) else if (C <= 512) VS_LNB2(DT_, 2) else if (C <= 768) VS_LNB2(DT_, 3) \
                      else if (C <= 1024) VS_LNB2(DT_, 4) else VS_LNB2(DT_, 8) }
    if (do_dtype == 0) VS_LNB(0) else if (do_dtype == 1) VS_LNB(1) else VS_LNB(2)
#undef VS_LNB3
#undef VS_LNB2
#undef VS_LNB
    VS_HIP(hipGetLastError());
    return 0;
}
}  // namespace

extern "C" int vs_layernorm_backward(const void *dout, int64_t ld_do, int32_t do_dtype, const float *x, int64_t ldx, const float *w,
                                     const float *b, const float *scale, int32_t mod_rows, int32_t mod_ld, float *dx, int64_t ld_dx,
                                     int32_t accumulate_dx, float *dw, float *db, float *dscale, float *dshift, int32_t M, int32_t C,
                                     float eps, int32_t grp_in, int32_t grp_out, int32_t grp_off, vs_stream_t stream_) {
    return layernorm_backward_entry(dout, ld_do, do_dtype, x, ldx, w, b, scale, mod_rows, mod_ld, dx, ld_dx, accumulate_dx ? dx : nullptr, ld_dx,
                                    nullptr, 0, 0, dw, db, dscale, dshift, M, C, eps, grp_in, grp_out, grp_off, stream_);
}

// vs_layernorm_backward with the residual-path gradient read from its own buffer (dx = dx_add + LayerNorm gradient; dx_add may be
// null or dx itself) and an optional 16-bit copy of dx (dx16, dtype 1 f16 / 2 bf16) for the GEMMs that consume it next.
extern "C" int vs_layernorm_backward_ex(const void *dout, int64_t ld_do, int32_t do_dtype, const float *x, int64_t ldx, const float *w,
                                        const float *b, const float *scale, int32_t mod_rows, int32_t mod_ld, float *dx, int64_t ld_dx,
                                        const float *dx_add, int64_t ld_add, void *dx16, int64_t ld16, int32_t dx16_dtype, float *dw,
                                        float *db, float *dscale, float *dshift, int32_t M, int32_t C, float eps, int32_t grp_in,
                                        int32_t grp_out, int32_t grp_off, vs_stream_t stream_) {
    return layernorm_backward_entry(dout, ld_do, do_dtype, x, ldx, w, b, scale, mod_rows, mod_ld, dx, ld_dx, dx_add, ld_add, dx16, ld16,
                                    dx16_dtype, dw, db, dscale, dshift, M, C, eps, grp_in, grp_out, grp_off, stream_);
}

extern "C" int vs_gated_resid(const float *x, const void *y, int64_t ldy, const float *gate, int32_t gate_rows, float *out, int32_t M,
                              int32_t C, int32_t grp_in, int32_t grp_out, int32_t grp_off, int32_t dtype, vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(x && y && out, "vs_gated_resid: null pointer");
    VS_CHECK(dtype == 1 || dtype == 2, "vs_gated_resid: dtype must be 1 (f16) or 2 (bf16)");
    VS_CHECK(M >= 0 && C > 0 && C % 4 == 0 && ldy % 4 == 0, "vs_gated_resid: C and ldy must be multiples of 4");
    VS_CHECK((((uintptr_t)x | (uintptr_t)out | (uintptr_t)gate) & 15) == 0 && ((uintptr_t)y & 7) == 0, "vs_gated_resid: misaligned pointer");
    if (M == 0) return 0;
    if (grp_in <= 0) { grp_in = M; grp_out = M; grp_off = 0; }
    if (gate_rows <= 0) gate_rows = M;
    dim3 grid((unsigned)vs::cdiv64((long long)M * (C / 4), 256)), block(256);
    if (dtype == 2) hipLaunchKernelGGL(gated_resid_kernel<true>, grid, block, 0, stream, x, (const unsigned short *)y, (long long)ldy, gate, gate_rows, out, M, C, grp_in, grp_out, grp_off);
    else hipLaunchKernelGGL(gated_resid_kernel<false>, grid, block, 0, stream, x, (const unsigned short *)y, (long long)ldy, gate, gate_rows, out, M, C, grp_in, grp_out, grp_off);
    VS_HIP(hipGetLastError());
    return 0;
}

extern "C" int vs_gated_resid_backward(const float *dout, const void *y, int64_t ldy, const float *gate, int32_t gate_rows, void *dy,
                                       int64_t lddy, float *dgate, int32_t M, int32_t C, int32_t grp_in, int32_t grp_out,
                                       int32_t grp_off, int32_t dtype, vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(dout && dy && (!gate || (y && dgate)), "vs_gated_resid_backward: null pointer");
    VS_CHECK(dtype == 1 || dtype == 2, "vs_gated_resid_backward: dtype must be 1 (f16) or 2 (bf16)");
    VS_CHECK(M >= 0 && C > 0 && C % 4 == 0 && C <= 2048 && ldy % 4 == 0 && lddy % 4 == 0, "vs_gated_resid_backward: C (<= 2048), ldy, lddy must be multiples of 4");
    VS_CHECK((((uintptr_t)dout | (uintptr_t)gate | (uintptr_t)dgate) & 15) == 0 && (((uintptr_t)y | (uintptr_t)dy) & 7) == 0, "vs_gated_resid_backward: misaligned pointer");
    if (M == 0) return 0;
    if (grp_in <= 0) { grp_in = M; grp_out = M; grp_off = 0; }
    if (!gate || gate_rows <= 0) gate_rows = M;
    const int G = vs::cdiv(M, gate_rows);
    VS_CHECK(G <= 65535, "vs_gated_resid_backward: too many gate groups (%d)", G);
    const int chunks = std::max(1, std::min(1024 / G, vs::cdiv(gate_rows, 4)));
    const int rows_per_chunk = vs::cdiv(gate_rows, chunks);
    dim3 grid(vs::cdiv(gate_rows, rows_per_chunk), G), block(256);
#define VS_GRB2(BF_, NV_) hipLaunchKernelGGL((gated_resid_backward_kernel<BF_, NV_>), grid, block, 0, stream, dout, (const unsigned short *)y, (long long)ldy, \
                                          gate, gate_rows, (unsigned short *)dy, (long long)lddy, dgate, M, C, grp_in, grp_out, grp_off, rows_per_chunk)
#define VS_GRB(BF_) { if (C <= 256) VS_GRB2(BF_, 1); else if (C <= 512) VS_GRB2(BF_, 2); else if (C <= 768) VS_GRB2(BF_, 3); \
                      else if (C <= 1024) VS_GRB2(BF_, 4); else VS_GRB2(BF_, 8); }
    if (dtype == 2) VS_GRB(true) else VS_GRB(false)
#undef VS_GRB
#undef VS_GRB2
    VS_HIP(hipGetLastError());
    return 0;
}

extern "C" int vs_relu_mask16_to(const void *dy, const void *x, void *out, int64_t n, vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(dy && x && out, "vs_relu_mask16: null pointer");
    VS_CHECK(n >= 0 && n % 8 == 0, "vs_relu_mask16: n must be a multiple of 8");
    VS_CHECK((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)out) & 15) == 0, "vs_relu_mask16: 16-byte alignment required");
    if (n == 0) return 0;
    hipLaunchKernelGGL(relu_mask16_kernel, dim3((unsigned)vs::cdiv64(n / 8, 256)), dim3(256), 0, stream, (const unsigned short *)dy,
                       (const unsigned short *)x, (unsigned short *)out, (long long)n);
    VS_HIP(hipGetLastError());
    return 0;
}

extern "C" int vs_relu_mask16(void *dx, const void *x, int64_t n, vs_stream_t stream_) { return vs_relu_mask16_to(dx, x, dx, n, stream_); }
