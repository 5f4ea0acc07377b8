// Operand preparation and f32 element-wise kernels of the SPLIT-class backward (the reference-precision training step; gfx950 only).
//
// In the split operand class (gemm_common.h, kDtSplit) every activation and every gradient is an f32 tensor in HBM and an MFMA product is
// three f16 MFMAs on (hi, lo) pairs.  The forward needs no preparation pass (activations are converted inside the kernels, weights are
// packed once); the backward has products whose BOTH operands are activations reduced over their ROW index (dW = dY^T X) and the
// attention backward's transposed operands.  This file holds the passes that put them into the forms the MFMA kernels take:
//   vs_transpose_f32          out[c][r] = in[r][c]           (the f32 "A" operand of the weight-gradient GEMM: dY^T)
//   vs_transpose_pack_split   the same, written as the packed (hi, lo) "W" operand of vs_gemm_split / vs_gemm_wgrad (X^T), optionally
//                             through a 3x3-convolution tap (row r = output pixel, source = pixel + (dy, dx), zero outside the image)
//                             and the ResidualConvUnit's ReLU -- the convolution weight gradient as nine GEMMs without an im2col buffer
//   vs_split16                f32 -> separate hi / lo 16-bit images (Q, K, V, dO of vs_attention_backward_split)
// and the f32 forms of the element-wise backward operators (GELU, ReLU mask, gated residual, bilinear x2 transpose).
// Reference: torch autograd of nn.Linear / nn.Conv2d / F.gelu / F.interpolate in the reference's fp32 training step
// (model_wrapper.py:184-321, config/experiment/re10k_8view.yaml:75-80).
#include "common.h"
#include "gemm_common.h"

namespace {

typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split_pair(float a, float b, unsigned &h, unsigned &l) {
    const h2v hh = __builtin_convertvector(f2v{a, b}, h2v);
    const h2v ll = __builtin_convertvector(f2v{a - (float)hh.x, b - (float)hh.y}, h2v);
    h = __builtin_bit_cast(unsigned, hh);
    l = __builtin_bit_cast(unsigned, ll);
}

// 64 (r) x 64 (c) tile through LDS.  PACK = false: out f32 [C, ld_out], out[c][r].  PACK = true: out = packed split rows (4 bytes per
// element, ld_out in elements): per block of 32 r, 32 hi halves then 32 lo halves; inside a block the 8-half chunk g holds
// r = {4g..4g+3, 16+4g..16+4g+3} (split_pack_kernel's order, gemm.hip).  Rows r >= R (and taps outside the image) read as zero.
template <bool PACK>
__global__ void __launch_bounds__(256)
transpose_f32_kernel(const float *__restrict__ in, long long ld_in, void *__restrict__ out_, long long ld_out, int R, int C, int relu,
                     int cH, int cW, int dy, int dx, float scale, float *__restrict__ colsum) {
    __shared__ float t[64][65];
    const int tiles_c = (C + 63) >> 6;
    const int r0 = (int)(blockIdx.x / tiles_c) * 64, c0 = (int)(blockIdx.x % tiles_c) * 64;
    const bool vec_in = (ld_in & 3) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0 && (C & 3) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int rr = idx >> 4, c4 = (idx & 15) * 4;
        const int r = r0 + rr, c = c0 + c4;
        bool ok = r < R;
        long long src = r;
        if (cH > 0 && ok) {
            if (dy == 2) {   // border mode: r indexes the pixels of zero-bordered (cH + 2) x (cW + 2) maps
                const int Wp = cW + 2, HWp = (cH + 2) * Wp;
                const int n = r / HWp, rem = r - n * HWp;
                const int y = rem / Wp - 1, x = rem - (rem / Wp) * Wp - 1;
                ok = y >= 0 && y < cH && x >= 0 && x < cW;
                src = ((long long)n * cH + y) * cW + x;
            } else {
                const int x = r % cW, y = (r / cW) % cH;
                const int sy = y + dy, sx = x + dx;
                ok = sy >= 0 && sy < cH && sx >= 0 && sx < cW;
                src = (long long)r + dy * cW + dx;
            }
        }
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
            const float *ip = in + src * ld_in + c;
            if (vec_in && c + 3 < C) {
                const float4 u = *reinterpret_cast<const float4 *>(ip);
                v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c + e < C) v[e] = ip[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float w = v[e];
            if (relu) w = fmaxf(w, 0.f);
            t[rr][c4 + e] = w * scale;
        }
    }
    __syncthreads();
    if (colsum) {   // bias gradient on the way: thread (column tid >> 2, quarter tid & 3) sums 16 rows of the tile, the quarters meet by DPP
        const int cc = threadIdx.x >> 2, part = threadIdx.x & 3;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += t[part * 16 + i][cc];
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (part == 0 && c0 + cc < C) unsafeAtomicAdd(colsum + c0 + cc, s / scale);   // (scale is a power of two: exact)
    }
    if constexpr (!PACK) {
        float *out = reinterpret_cast<float *>(out_);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = threadIdx.x + 256 * i;
            const int r4 = (idx & 15) * 4, cc = idx >> 4;
            if (c0 + cc >= C) continue;
            const float4 o = make_float4(t[r4][cc], t[r4 + 1][cc], t[r4 + 2][cc], t[r4 + 3][cc]);
            *reinterpret_cast<float4 *>(out + (long long)(c0 + cc) * ld_out + r0 + r4) = o;
        }
    } else {
        unsigned short *out = reinterpret_cast<unsigned short *>(out_);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = threadIdx.x + 256 * i;
            const int q = idx & 7, cc = (idx >> 3) & 63, blk = idx >> 9;
            if (c0 + cc >= C) continue;
            const int rb = blk * 32 + q * 4;
            unsigned h0, l0, h1, l1;
            split_pair(t[rb][cc], t[rb + 1][cc], h0, l0);
            split_pair(t[rb + 2][cc], t[rb + 3][cc], h1, l1);
            const int pos = (q & 3) * 8 + (q >> 2) * 4;
            unsigned short *o = out + (long long)(c0 + cc) * (2 * ld_out) + ((long long)(r0 >> 5) + blk) * 64 + pos;
            *reinterpret_cast<uint2 *>(o) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(o + 32) = make_uint2(l0, l1);
        }
    }
}

// The plain form of the packing transpose (no convolution tap; C a multiple of 64, 16-byte aligned rows) -- the dY^T operand of every linear /
// 1x1 weight gradient of the split-class training step.  TR x 64 tile; the (hi, lo) conversion and the store take one 8-half CHUNK per
// thread (rows {4g..4g+3, 16+4g..16+4g+3} of a 32-row block: 8 LDS reads -> one 16-byte hi store + one 16-byte lo store; four lanes fill
// the 64 + 64 bytes of a block), LDS rows of 66 words (the chunk reads of 32 lanes fall into 32 different banks), the bias gradient summed
// from the registers of the load phase (two DPP steps + one LDS row per wave) instead of a second walk over the tile.
// RFAST: consecutive workgroups walk DOWN the rows of one 64-column band (their stores extend the same output rows).
template <int TR, bool RFAST>
__global__ void __launch_bounds__(256)
transpose_pack_plain_kernel(const float *__restrict__ in, long long ld_in, unsigned short *__restrict__ out, long long ld_out, int R, int C, int Rpad,
                            int relu, float scale, float *__restrict__ colsum) {
    constexpr int S = 66;
    __shared__ float t[TR][S];
    __shared__ float cs[4][64];
    const int tiles_c = C >> 6, tiles_r = Rpad / TR;
    const int r0 = (RFAST ? (int)(blockIdx.x % tiles_r) : (int)(blockIdx.x / tiles_c)) * TR;
    const int c0 = (RFAST ? (int)(blockIdx.x / tiles_r) : (int)(blockIdx.x % tiles_c)) * 64;
    const int c4 = (threadIdx.x & 15) * 4;
    float4 v[TR / 16];
#pragma unroll
    for (int i = 0; i < TR / 16; ++i) {
        const int r = r0 + (int)(threadIdx.x >> 4) + 16 * i;
        v[i] = r < R ? *reinterpret_cast<const float4 *>(in + (long long)r * ld_in + c0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < TR / 16; ++i) {
        float4 w = v[i];
        if (relu) { w.x = fmaxf(w.x, 0.f); w.y = fmaxf(w.y, 0.f); w.z = fmaxf(w.z, 0.f); w.w = fmaxf(w.w, 0.f); }
        sum.x += w.x; sum.y += w.y; sum.z += w.z; sum.w += w.w;
        float *p = &t[(threadIdx.x >> 4) + 16 * i][c4];
        *reinterpret_cast<float2 *>(p) = make_float2(w.x * scale, w.y * scale);
        *reinterpret_cast<float2 *>(p + 2) = make_float2(w.z * scale, w.w * scale);
    }
    if (colsum) {
        sum.x += __shfl_xor(sum.x, 16); sum.y += __shfl_xor(sum.y, 16); sum.z += __shfl_xor(sum.z, 16); sum.w += __shfl_xor(sum.w, 16);
        sum.x += __shfl_xor(sum.x, 32); sum.y += __shfl_xor(sum.y, 32); sum.z += __shfl_xor(sum.z, 32); sum.w += __shfl_xor(sum.w, 32);
        if ((threadIdx.x & 63) < 16) *reinterpret_cast<float4 *>(&cs[threadIdx.x >> 6][c4]) = sum;
    }
    __syncthreads();
    if (colsum && threadIdx.x < 64) unsafeAtomicAdd(colsum + c0 + threadIdx.x, (cs[0][threadIdx.x] + cs[1][threadIdx.x]) + (cs[2][threadIdx.x] + cs[3][threadIdx.x]));
#pragma unroll
    for (int i = 0; i < TR / 32; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int g = idx & 3, cc = (idx >> 2) & 63, blk = idx >> 8;
        const int rb = blk * 32 + 4 * g;
        unsigned h0, h1, h2, h3, l0, l1, l2, l3;
        split_pair(t[rb][cc], t[rb + 1][cc], h0, l0);
        split_pair(t[rb + 2][cc], t[rb + 3][cc], h1, l1);
        split_pair(t[rb + 16][cc], t[rb + 17][cc], h2, l2);
        split_pair(t[rb + 18][cc], t[rb + 19][cc], h3, l3);
        unsigned short *o = out + (long long)(c0 + cc) * (2 * ld_out) + ((long long)(r0 >> 5) + blk) * 64 + g * 8;
        *reinterpret_cast<uint4 *>(o) = make_uint4(h0, h1, h2, h3);
        *reinterpret_cast<uint4 *>(o + 32) = make_uint4(l0, l1, l2, l3);
    }
}

__global__ void __launch_bounds__(256)
split16_kernel(const float *__restrict__ in, long long ld_in, unsigned short *__restrict__ hi, unsigned short *__restrict__ lo, long long ld_out,
               long long rows, int C4) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * C4) return;
    const long long r = idx / C4;
    const int c = (int)(idx - r * C4) * 4;
    const float4 v = *reinterpret_cast<const float4 *>(in + r * ld_in + c);
    unsigned h0, l0, h1, l1;
    split_pair(v.x, v.y, h0, l0);
    split_pair(v.z, v.w, h1, l1);
    *reinterpret_cast<uint2 *>(hi + r * ld_out + c) = make_uint2(h0, h1);
    *reinterpret_cast<uint2 *>(lo + r * ld_out + c) = make_uint2(l0, l1);
}

// exact (erf) GELU and its derivative in f32 (croco/blocks.py:60; the 16-bit forms are in backward.hip)
__global__ void __launch_bounds__(256) gelu_f32_kernel(const float *__restrict__ z, float *__restrict__ out, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 v = reinterpret_cast<const float4 *>(z)[i];
    auto f = [](float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); };
    reinterpret_cast<float4 *>(out)[i] = make_float4(f(v.x), f(v.y), f(v.z), f(v.w));
}
__global__ void __launch_bounds__(256)
gelu_backward_f32_kernel(const float *__restrict__ dy, const float *__restrict__ z, float *__restrict__ dz, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 g = reinterpret_cast<const float4 *>(dy)[i], v = reinterpret_cast<const float4 *>(z)[i];
    auto f = [](float gg, float x) {   // d/dx [x Phi(x)] = Phi(x) + x phi(x)
        const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
        const float pdf = 0.3989422804014327f * expf(-0.5f * x * x);
        return gg * (cdf + x * pdf);
    };
    reinterpret_cast<float4 *>(dz)[i] = make_float4(f(g.x, v.x), f(g.y, v.y), f(g.z, v.z), f(g.w, v.w));
}
__global__ void __launch_bounds__(256)
relu_mask_f32_kernel(const float *dy, const float *__restrict__ x, float *out, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 g = reinterpret_cast<const float4 *>(dy)[i], v = reinterpret_cast<const float4 *>(x)[i];
    reinterpret_cast<float4 *>(out)[i] = make_float4(v.x > 0.f ? g.x : 0.f, v.y > 0.f ? g.y : 0.f, v.z > 0.f ? g.z : 0.f, v.w > 0.f ? g.w : 0.f);
}

// gated residual update with an f32 branch (backward.hip: gated_resid_kernel for the 16-bit classes)
__global__ void __launch_bounds__(256)
gated_resid_f32_kernel(const float *__restrict__ x, const float *__restrict__ y, long long ldy, const float *__restrict__ gate, int gate_rows,
                       float *__restrict__ out, int M, int C, int grp_in, int grp_out, int grp_off) {
    const int C4 = C >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)M * C4) return;
    const int m = (int)(idx / C4), c = (int)(idx - (long long)m * C4) * 4;
    const long long yr = (long long)(m / grp_in) * grp_out + grp_off + (m % grp_in);
    const float4 xv = *reinterpret_cast<const float4 *>(x + (long long)m * C + c);
    const float4 t = *reinterpret_cast<const float4 *>(y + yr * ldy + c);
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gate) g = *reinterpret_cast<const float4 *>(gate + (long long)(m / gate_rows) * C + c);
    *reinterpret_cast<float4 *>(out + (long long)m * C + c) =
        make_float4(xv.x + (1.0f + g.x) * t.x, xv.y + (1.0f + g.y) * t.y, xv.z + (1.0f + g.z) * t.z, xv.w + (1.0f + g.w) * t.w);
}

template <int NV>
__global__ void __launch_bounds__(256)
gated_resid_backward_f32_kernel(const float *__restrict__ dout, const float *__restrict__ y, long long ldy, const float *__restrict__ gate,
                                int gate_rows, float *__restrict__ dy, long long lddy, float *__restrict__ dgate, int M, int C, int grp_in,
                                int grp_out, int grp_off, int rows_per_chunk) {
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
                const float4 t = *reinterpret_cast<const float4 *>(y + yr * ldy + c);
                acc[k].x += d.x * t.x; acc[k].y += d.y * t.y; acc[k].z += d.z * t.z; acc[k].w += d.w * t.w;
            }
            *reinterpret_cast<float4 *>(dy + yr * lddy + c) =
                make_float4(d.x * (1.0f + gt[k].x), d.y * (1.0f + gt[k].y), d.z * (1.0f + gt[k].z), d.w * (1.0f + gt[k].w));
        }
    }
    if (!gate) return;
#pragma unroll
    for (int k = 0; k < NV; ++k) *reinterpret_cast<float4 *>(&red[wv_id][(lane + 64 * k) * 4]) = acc[k];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) unsafeAtomicAdd(dgate + (long long)gidx * C + c, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
}

// transpose of the bilinear x2 (align_corners = True) interpolation, f32 NHWC (conv.hip: upsample2x_backward_kernel for 16-bit)
__global__ void __launch_bounds__(256)
upsample2x_backward_f32_kernel(const float *__restrict__ dout, float *__restrict__ din, int Nimg, int H, int W, int C) {
    const int Ho = 2 * H, Wo = 2 * W, c4 = C >> 2;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= W * c4) return;
    const int x = e / c4, cc = e - x * c4;
    const int row = blockIdx.z * gridDim.y + blockIdx.y;
    if (row >= Nimg * H) return;
    const int n = row / H, y = row - n * H;
    const float ry = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, rx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int yo = max(0, 2 * y - 3); yo <= min(Ho - 1, 2 * y + 3); ++yo) {
        const float sy = (float)yo * ry;
        const int y0 = min((int)sy, H - 1), y1 = min(y0 + 1, H - 1);
        const float ly = sy - (float)y0;
        const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
        if (wy == 0.f) continue;
        for (int xo = max(0, 2 * x - 3); xo <= min(Wo - 1, 2 * x + 3); ++xo) {
            const float sx = (float)xo * rx;
            const int x0 = min((int)sx, W - 1), x1 = min(x0 + 1, W - 1);
            const float lx = sx - (float)x0;
            const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
            if (wx == 0.f) continue;
            const float wgt = wy * wx;
            const float4 v = *reinterpret_cast<const float4 *>(dout + ((((size_t)n * Ho + yo) * Wo + xo) * C + cc * 4));
            acc.x += wgt * v.x; acc.y += wgt * v.y; acc.z += wgt * v.z; acc.w += wgt * v.w;
        }
    }
    *reinterpret_cast<float4 *>(din + ((((size_t)n * H + y) * W + x) * C + cc * 4)) = acc;
}

int transpose_f32_entry(const char *fn, bool pack, const float *in, int64_t ld_in, void *out, int64_t ld_out, int32_t R, int32_t C, int32_t Rpad,
                        int32_t relu, int32_t cH, int32_t cW, int32_t dy, int32_t dx, int32_t scale_exp, float *colsum, hipStream_t stream) {
    VS_CHECK(in && out, "%s: null pointer", fn);
    VS_CHECK(!colsum || (!relu && (cH == 0 || dy == 2)), "%s: colsum goes with the plain and the bordered forms, without ReLU (every input row counted once)", fn);
    VS_CHECK(R >= 0 && C > 0 && Rpad >= R && Rpad % 64 == 0 && ld_in >= C && ld_out >= Rpad, "%s: bad sizes R=%d C=%d Rpad=%d (Rpad %% 64 == 0, ld_out >= Rpad)", fn, R, C, Rpad);
    VS_CHECK((cH == 0 && cW == 0) || (cH > 0 && cW > 0 && dy == 2 && R % ((cH + 2) * (cW + 2)) == 0) ||
                 (cH > 0 && cW > 0 && R % (cH * cW) == 0 && dy >= -1 && dy <= 1 && dx >= -1 && dx <= 1),
             "%s: a convolution tap needs R = whole H x W images and |dy|, |dx| <= 1 (tap_dy = 2: R = whole zero-bordered (H+2) x (W+2) images)", fn);
    VS_CHECK(ld_out % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0, "%s: out must be 16-byte aligned with ld_out %% 4 == 0", fn);
    VS_CHECK(scale_exp >= -30 && scale_exp <= 30, "%s: scale_exp out of range", fn);
    if (colsum) VS_HIP(hipMemsetAsync(colsum, 0, sizeof(float) * (size_t)C, stream));
    if (Rpad == 0) return 0;
    const long long nblk = (long long)vs::cdiv(C, 64) * (Rpad / 64);
    VS_CHECK(nblk <= 0x7fffffffLL, "%s: too many tiles", fn);
    dim3 grid((unsigned)nblk), block(256);
    const float scale = ldexpf(1.0f, scale_exp);
    // plain packing transposes (every linear / 1x1 weight gradient's dY^T) take the chunk-store kernel; VS_TP_PLAIN = 0 / 64 / 128 / 256 (rows per
    // tile) and VS_TP_RFAST for the A/B
    if (pack && cH == 0 && C % 64 == 0 && ld_in % 4 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0) {
        const char *e = getenv("VS_TP_PLAIN"), *e2 = getenv("VS_TP_RFAST");
        int tr = e ? atoi(e) : 128;
        const bool rfast = e2 ? atoi(e2) != 0 : false;
        while (tr > 64 && Rpad % tr != 0) tr >>= 1;
        if (tr == 64 || tr == 128 || tr == 256) {
            dim3 g2((unsigned)((long long)(C / 64) * (Rpad / tr)));
            unsigned short *o16 = reinterpret_cast<unsigned short *>(out);
#define VS_TP(TR_, RF_) hipLaunchKernelGGL((transpose_pack_plain_kernel<TR_, RF_>), g2, block, 0, stream, in, (long long)ld_in, o16, (long long)ld_out, R, C, Rpad, relu, scale, colsum)
            if (tr == 64) { if (rfast) VS_TP(64, true); else VS_TP(64, false); }
            else if (tr == 128) { if (rfast) VS_TP(128, true); else VS_TP(128, false); }
            else { if (rfast) VS_TP(256, true); else VS_TP(256, false); }
#undef VS_TP
            VS_HIP(hipGetLastError());
            return 0;
        }
    }
    if (pack) hipLaunchKernelGGL(transpose_f32_kernel<true>, grid, block, 0, stream, in, (long long)ld_in, out, (long long)ld_out, R, C, relu, cH, cW, dy, dx, scale, colsum);
    else hipLaunchKernelGGL(transpose_f32_kernel<false>, grid, block, 0, stream, in, (long long)ld_in, out, (long long)ld_out, R, C, relu, cH, cW, dy, dx, scale, colsum);
    VS_HIP(hipGetLastError());
    return 0;
}

// im2col rows of the 7x7 / pad 3 RGB stem for the TRAINING path (the stem runs as a GEMM over these rows: train_forward.stem7x7): out[p, k] =
// frames[n, c, y + ky - 3, x + kx - 3] (0 outside) for k = c * 49 + ky * 7 + kx < 147, zeros for 147 <= k < ld -- F.unfold + transpose + F.pad in
// ONE pass (the three PyTorch passes moved 13.6 GB per 8-scene step for the 4.3 GB this writes).  One wave writes one pixel's ld = 256 columns.
template <int OUT>   // 0 f32, 1 f16, 2 bf16
__global__ void __launch_bounds__(256) im2col7x7_rgb_kernel(const float *__restrict__ fr, void *__restrict__ out, long long npix, int H, int W, int ld) {
    // column -> (plane offset c * H * W, ky - 3, kx - 3), decoded once per block (three integer divisions per ELEMENT made this pass 1.5 TB/s)
    __shared__ int s_off[256];
    __shared__ signed char s_dy[256], s_dx[256];
    for (int k = threadIdx.x; k < 256; k += 256) {
        const int c = k / 49, r = k - c * 49, ky = r / 7, kx = r - ky * 7;
        s_off[k] = k < 147 ? c * H * W : -1;
        s_dy[k] = (signed char)(ky - 3); s_dx[k] = (signed char)(kx - 3);
    }
    __syncthreads();
    constexpr int EPT = OUT == 0 ? 4 : 8;       // columns per thread: one 16-byte store
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int per = ld / EPT;
    const long long p = t / per;
    if (p >= npix) return;
    const int k0 = (int)(t - p * per) * EPT;
    const long long hw = (long long)H * W;
    const long long n = p / hw, rem = p - n * hw;
    const int y = (int)(rem / W), x = (int)(rem - (long long)y * W);
    const float *img = fr + n * 3 * hw;
    float v[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int k = k0 + i;
        float a = 0.f;
        if (k < 256 && s_off[k] >= 0) {
            const int yy = y + s_dy[k], xx = x + s_dx[k];
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) a = img[s_off[k] + yy * W + xx];
        }
        v[i] = a;
    }
    if constexpr (OUT == 0) {
        *reinterpret_cast<float4 *>(reinterpret_cast<float *>(out) + p * ld + k0) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        constexpr int B = OUT == 2 ? 1 : 0;
        *reinterpret_cast<uint4 *>(reinterpret_cast<unsigned short *>(out) + p * ld + k0) =
            make_uint4(pack16x2<B>(v[0], v[1]), pack16x2<B>(v[2], v[3]), pack16x2<B>(v[4], v[5]), pack16x2<B>(v[6], v[7]));
    }
}

}  // namespace

extern "C" int vs_transpose_f32(const float *in, int64_t ld_in, float *out, int64_t ld_out, int32_t R, int32_t C, int32_t Rpad, int32_t relu,
                                int32_t conv_H, int32_t conv_W, int32_t tap_dy, int32_t tap_dx, float *colsum, vs_stream_t stream_) {
    return transpose_f32_entry("vs_transpose_f32", false, in, ld_in, out, ld_out, R, C, Rpad, relu, conv_H, conv_W, tap_dy, tap_dx, 0, colsum, (hipStream_t)stream_);
}

extern "C" int vs_transpose_pack_split(const float *in, int64_t ld_in, void *out, int64_t ld_out, int32_t R, int32_t C, int32_t Rpad, int32_t relu,
                                       int32_t conv_H, int32_t conv_W, int32_t tap_dy, int32_t tap_dx, int32_t scale_exp, float *colsum,
                                       vs_stream_t stream_) {
    return transpose_f32_entry("vs_transpose_pack_split", true, in, ld_in, out, ld_out, R, C, Rpad, relu, conv_H, conv_W, tap_dy, tap_dx, scale_exp,
                               colsum, (hipStream_t)stream_);
}

extern "C" int vs_split16(const float *in, int64_t ld_in, void *hi, void *lo, int64_t ld_out, int64_t rows, int32_t C, vs_stream_t stream_) {
    VS_CHECK(in && hi && lo, "vs_split16: null pointer");
    VS_CHECK(rows >= 0 && C > 0 && C % 4 == 0 && ld_in % 4 == 0 && ld_out % 4 == 0 && ld_in >= C && ld_out >= C, "vs_split16: C and the row strides must be multiples of 4");
    VS_CHECK((((uintptr_t)in) & 15) == 0 && (((uintptr_t)hi | (uintptr_t)lo) & 7) == 0, "vs_split16: misaligned pointer");
    if (rows == 0) return 0;
    const long long items = rows * (C / 4);
    hipLaunchKernelGGL(split16_kernel, dim3((unsigned)vs::cdiv64(items, 256)), dim3(256), 0, (hipStream_t)stream_, in, (long long)ld_in, (unsigned short *)hi,
                       (unsigned short *)lo, (long long)ld_out, (long long)rows, C / 4);
    VS_HIP(hipGetLastError());
    return 0;
}

extern "C" int vs_im2col7x7_rgb(const float *frames, void *out, int32_t N, int32_t H, int32_t W, int32_t ld, int32_t out_dtype, vs_stream_t stream_) {
    VS_CHECK(frames && out && N >= 0 && H > 0 && W > 0, "vs_im2col7x7_rgb: null pointer / bad sizes");
    VS_CHECK(ld >= 148 && ld % 8 == 0 && out_dtype >= 0 && out_dtype <= 2, "vs_im2col7x7_rgb: ld=%d must be a multiple of 8 covering 147 columns; out_dtype 0 f32 / 1 f16 / 2 bf16", ld);
    VS_CHECK(((uintptr_t)out & 15) == 0, "vs_im2col7x7_rgb: out must be 16-byte aligned");
    const long long npix = (long long)N * H * W, items = npix * (ld / (out_dtype == 0 ? 4 : 8));
    if (items == 0) return 0;
    VS_CHECK(vs::cdiv64(items, 256) < (1LL << 31), "vs_im2col7x7_rgb: too large");
    dim3 grid((unsigned)vs::cdiv64(items, 256)), block(256);
    hipStream_t s = (hipStream_t)stream_;
    if (out_dtype == 0) hipLaunchKernelGGL(im2col7x7_rgb_kernel<0>, grid, block, 0, s, frames, out, npix, H, W, ld);
    else if (out_dtype == 1) hipLaunchKernelGGL(im2col7x7_rgb_kernel<1>, grid, block, 0, s, frames, out, npix, H, W, ld);
    else hipLaunchKernelGGL(im2col7x7_rgb_kernel<2>, grid, block, 0, s, frames, out, npix, H, W, ld);
    VS_HIP(hipGetLastError());
    return 0;
}

extern "C" int vs_gelu_f32(const float *z, float *out, int64_t n, vs_stream_t stream_) {
    VS_CHECK(z && out && n >= 0 && n % 4 == 0 && (((uintptr_t)z | (uintptr_t)out) & 15) == 0, "vs_gelu_f32: n %% 4 == 0 and 16-byte alignment required");
    if (n == 0) return 0;
    hipLaunchKernelGGL(gelu_f32_kernel, dim3((unsigned)vs::cdiv64(n / 4, 256)), dim3(256), 0, (hipStream_t)stream_, z, out, (long long)(n / 4));
    VS_HIP(hipGetLastError());
    return 0;
}

extern "C" int vs_gelu_backward_f32(const float *dy, const float *z, float *dz, int64_t n, vs_stream_t stream_) {
    VS_CHECK(dy && z && dz && n >= 0 && n % 4 == 0 && (((uintptr_t)dy | (uintptr_t)z | (uintptr_t)dz) & 15) == 0,
             "vs_gelu_backward_f32: n %% 4 == 0 and 16-byte alignment required");
    if (n == 0) return 0;
    hipLaunchKernelGGL(gelu_backward_f32_kernel, dim3((unsigned)vs::cdiv64(n / 4, 256)), dim3(256), 0, (hipStream_t)stream_, dy, z, dz, (long long)(n / 4));
    VS_HIP(hipGetLastError());
    return 0;
}

extern "C" int vs_relu_mask_f32(const float *dy, const float *x, float *out, int64_t n, vs_stream_t stream_) {
    VS_CHECK(dy && x && out && n >= 0 && n % 4 == 0 && (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)out) & 15) == 0,
             "vs_relu_mask_f32: n %% 4 == 0 and 16-byte alignment required");
    if (n == 0) return 0;
    hipLaunchKernelGGL(relu_mask_f32_kernel, dim3((unsigned)vs::cdiv64(n / 4, 256)), dim3(256), 0, (hipStream_t)stream_, dy, x, out, (long long)(n / 4));
    VS_HIP(hipGetLastError());
    return 0;
}

extern "C" int vs_gated_resid_f32(const float *x, const float *y, int64_t ldy, const float *gate, int32_t gate_rows, float *out, int32_t M, int32_t C,
                                  int32_t grp_in, int32_t grp_out, int32_t grp_off, vs_stream_t stream_) {
    VS_CHECK(x && y && out, "vs_gated_resid_f32: null pointer");
    VS_CHECK(M >= 0 && C > 0 && C % 4 == 0 && ldy % 4 == 0, "vs_gated_resid_f32: C and ldy must be multiples of 4");
    VS_CHECK((((uintptr_t)x | (uintptr_t)out | (uintptr_t)gate | (uintptr_t)y) & 15) == 0, "vs_gated_resid_f32: misaligned pointer");
    if (M == 0) return 0;
    if (grp_in <= 0) { grp_in = M; grp_out = M; grp_off = 0; }
    if (gate_rows <= 0) gate_rows = M;
    hipLaunchKernelGGL(gated_resid_f32_kernel, dim3((unsigned)vs::cdiv64((long long)M * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream_, x, y, (long long)ldy,
                       gate, gate_rows, out, M, C, grp_in, grp_out, grp_off);
    VS_HIP(hipGetLastError());
    return 0;
}

extern "C" int vs_gated_resid_backward_f32(const float *dout, const float *y, int64_t ldy, const float *gate, int32_t gate_rows, float *dy, int64_t lddy,
                                           float *dgate, int32_t M, int32_t C, int32_t grp_in, int32_t grp_out, int32_t grp_off, vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(dout && dy && (!gate || (y && dgate)), "vs_gated_resid_backward_f32: null pointer");
    VS_CHECK(M >= 0 && C > 0 && C % 4 == 0 && C <= 2048 && ldy % 4 == 0 && lddy % 4 == 0, "vs_gated_resid_backward_f32: C (<= 2048), ldy, lddy must be multiples of 4");
    VS_CHECK((((uintptr_t)dout | (uintptr_t)gate | (uintptr_t)dgate | (uintptr_t)y | (uintptr_t)dy) & 15) == 0, "vs_gated_resid_backward_f32: misaligned pointer");
    if (M == 0) return 0;
    if (grp_in <= 0) { grp_in = M; grp_out = M; grp_off = 0; }
    if (!gate || gate_rows <= 0) gate_rows = M;
    const int G = vs::cdiv(M, gate_rows);
    VS_CHECK(G <= 65535, "vs_gated_resid_backward_f32: too many gate groups (%d)", G);
    const int chunks = std::max(1, std::min(1024 / G, vs::cdiv(gate_rows, 4)));
    const int rows_per_chunk = vs::cdiv(gate_rows, chunks);
    dim3 grid(vs::cdiv(gate_rows, rows_per_chunk), G), block(256);
#define VS_GRB2(NV_) hipLaunchKernelGGL((gated_resid_backward_f32_kernel<NV_>), grid, block, 0, stream, dout, y, (long long)ldy, gate, gate_rows, dy, \
                                        (long long)lddy, dgate, M, C, grp_in, grp_out, grp_off, rows_per_chunk)
    if (C <= 256) VS_GRB2(1); else if (C <= 512) VS_GRB2(2); else if (C <= 768) VS_GRB2(3); else if (C <= 1024) VS_GRB2(4); else VS_GRB2(8);
#undef VS_GRB2
    VS_HIP(hipGetLastError());
    return 0;
}

extern "C" int vs_upsample2x_backward_f32_nhwc(const float *dout, float *din, int32_t Nimg, int32_t H, int32_t W, int32_t C, vs_stream_t stream_) {
    VS_CHECK(dout && din, "vs_upsample2x_backward_f32_nhwc: null pointer");
    VS_CHECK(C % 4 == 0 && (((uintptr_t)dout | (uintptr_t)din) & 15) == 0, "vs_upsample2x_backward_f32_nhwc: C %% 4 == 0 and 16-byte alignment required");
    if ((long long)Nimg * H * W * C <= 0) return 0;
    const int rows = Nimg * H, gy = rows < 32768 ? rows : 32768;
    dim3 grid((unsigned)vs::cdiv(W * (C / 4), 256), gy, vs::cdiv(rows, gy)), block(256);
    hipLaunchKernelGGL(upsample2x_backward_f32_kernel, grid, block, 0, (hipStream_t)stream_, dout, din, Nimg, H, W, C);
    VS_HIP(hipGetLastError());
    return 0;
}
