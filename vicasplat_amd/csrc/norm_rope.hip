// Memory-bound token kernels of the ViT blocks for gfx950: LayerNorm (+AdaLN modulation) and RoPE on the packed
// q|k|v projection buffer.  One wavefront per token row, 16-byte vector accesses, wave-shuffle reductions.
//
//  vs_layernorm_mod : y = LN(x; w, b, eps) [* (1 + scale[g]) + shift[g]]   f32 in -> f16/bf16/f32 out
//      reference: nn.LayerNorm(eps=1e-6) backbone_vica.py:370; _modulate :268-273 (scale/shift come from the frame's
//      camera token, AdaLNModulation :194-212), g = row / mod_rows.  Output rows can be re-mapped
//      (row -> (row / grp_in) * grp_out + grp_off + row % grp_in) to interleave camera and image tokens.
//  vs_rope_qk       : in-place rotary embedding of the q and k column blocks of a packed [rows, ld] projection
//      kind 0 : 2-D RoPE (croco/pos_embed.py:112-159 == curope/kernels.cu:39-80), pairs (i, i+16) per 32-wide half,
//               angle = pos * base^(-i/16);   kind 1 : 1-D temporal RoPE with interleaved pairs (2j, 2j+1),
//               angle = t * theta^(-2j/64) (misc/rope_utils.py:133-188, camera tokens);   kind 2 : untouched.
#include "common.h"

namespace {

template <int DT> struct Out;
template <> struct Out<0> {
    using T = float;
    static __device__ __forceinline__ void st4(float *p, float a, float b, float c, float d) { *reinterpret_cast<float4 *>(p) = make_float4(a, b, c, d); }
};
template <> struct Out<1> {
    using T = unsigned short;
    static __device__ __forceinline__ unsigned short cv(float v) { _Float16 h = (_Float16)v; return *reinterpret_cast<unsigned short *>(&h); }
    static __device__ __forceinline__ void st4(unsigned short *p, float a, float b, float c, float d) {
        *reinterpret_cast<uint2 *>(p) = make_uint2(cv(a) | ((unsigned)cv(b) << 16), cv(c) | ((unsigned)cv(d) << 16));
    }
};
template <> struct Out<2> {
    using T = unsigned short;
    static __device__ __forceinline__ unsigned short cv(float v) {
        unsigned u = __float_as_uint(v);
        u += 0x7FFFu + ((u >> 16) & 1u);
        return (unsigned short)(u >> 16);
    }
    static __device__ __forceinline__ void st4(unsigned short *p, float a, float b, float c, float d) {
        *reinterpret_cast<uint2 *>(p) = make_uint2(cv(a) | ((unsigned)cv(b) << 16), cv(c) | ((unsigned)cv(d) << 16));
    }
};

// 3: the packed (hi, lo) form of the split operand class (vs_split_pack_weight layout, scale 2^0; rows are whole 128-byte blocks): the
// LayerNorm output goes straight to a GEMM that then has no conversion to do.  p = address of columns c .. c + 3 in 4-byte units.
template <> struct Out<3> {
    using T = float;
    static __device__ __forceinline__ void st4(float *p, float a, float b, float c, float d) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        const h2 h0 = __builtin_convertvector(f2{a, b}, h2), h1 = __builtin_convertvector(f2{c, d}, h2);
        const h2 l0 = __builtin_convertvector(f2{a - (float)h0.x, b - (float)h0.y}, h2), l1 = __builtin_convertvector(f2{c - (float)h1.x, d - (float)h1.y}, h2);
        const unsigned kk = (unsigned)((reinterpret_cast<uintptr_t>(p) >> 2) & 31u);
        unsigned short *o = reinterpret_cast<unsigned short *>(p - kk) + ((kk & 15u) >> 2) * 8 + (kk >> 4) * 4;
        *reinterpret_cast<uint2 *>(o) = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
        *reinterpret_cast<uint2 *>(o + 32) = make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
    }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

constexpr int kMaxVec = 8;  // C <= 64 lanes * 8 * 4 = 2048

template <int DT>
__global__ void __launch_bounds__(256)
layernorm_mod_kernel(const float *__restrict__ x, long long ldx, const float *__restrict__ w, const float *__restrict__ b,
                     const float *__restrict__ scale, const float *__restrict__ shift, int mod_rows, int mod_ld,
                     typename Out<DT>::T *__restrict__ out, long long ldo, int M, int C, float eps, int grp_in, int grp_out,
                     int grp_off) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float *xr = x + (long long)m * ldx;
    const int nvec = C >> 2;
    float4 v[kMaxVec];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxVec; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nvec) {
            v[i] = *reinterpret_cast<const float4 *>(xr + 4 * idx);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxVec; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nvec) {
            const float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + bb * bb) + (c * c + d * d);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    const long long orow = (long long)(m / grp_in) * grp_out + grp_off + (m % grp_in);
    const float *sc = scale ? scale + (long long)(m / mod_rows) * mod_ld : nullptr;
    const float *sh = shift ? shift + (long long)(m / mod_rows) * mod_ld : nullptr;
#pragma unroll
    for (int i = 0; i < kMaxVec; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nvec) {
            const float4 ww = *reinterpret_cast<const float4 *>(w + 4 * idx);
            const float4 bv = *reinterpret_cast<const float4 *>(b + 4 * idx);
            float y0 = (v[i].x - mean) * rstd * ww.x + bv.x, y1 = (v[i].y - mean) * rstd * ww.y + bv.y;
            float y2 = (v[i].z - mean) * rstd * ww.z + bv.z, y3 = (v[i].w - mean) * rstd * ww.w + bv.w;
            if (sc) {
                const float4 s4 = *reinterpret_cast<const float4 *>(sc + 4 * idx);
                y0 *= 1.0f + s4.x; y1 *= 1.0f + s4.y; y2 *= 1.0f + s4.z; y3 *= 1.0f + s4.w;
            }
            if (sh) {
                const float4 h4 = *reinterpret_cast<const float4 *>(sh + 4 * idx);
                y0 += h4.x; y1 += h4.y; y2 += h4.z; y3 += h4.w;
            }
            Out<DT>::st4(out + orow * ldo + 4 * idx, y0, y1, y2, y3);
        }
    }
}

// C = 256 * NV (1024 and 768: every LayerNorm of the model): a wave takes R consecutive rows and keeps w, b (and the AdaLN scale /
// shift of the rows' frame) in registers across them.  One row per wave re-reads 8-16 KB of parameters from L2 per 4 KB row of x: at
// the bench's row count the kernel was bound by the vector-memory pipe, not by HBM (4.7 TB/s plain, 3.3 TB/s with AdaLN at C = 1024
// against 6.3 achievable; tools/bench_ln.py).  All R rows' loads are issued before the first reduction.
template <int DT, int NV, int R, int ITER>
__global__ void __launch_bounds__(256)
layernorm_rows_kernel(const float *__restrict__ x, long long ldx, const float *__restrict__ w, const float *__restrict__ b,
                      const float *__restrict__ scale, const float *__restrict__ shift, int mod_rows, int mod_ld,
                      typename Out<DT>::T *__restrict__ out, long long ldo, int M, float eps, int grp_in, int grp_out, int grp_off) {
    constexpr int C = 256 * NV;
    const int lane = threadIdx.x & 63;
    const int mw = (blockIdx.x * 4 + (threadIdx.x >> 6)) * (R * ITER);   // the wave's R * ITER consecutive rows
    if (mw >= M) return;
    float4 ww[NV], bv[NV], s4[NV], h4[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        ww[i] = *reinterpret_cast<const float4 *>(w + 4 * (lane + 64 * i));
        bv[i] = *reinterpret_cast<const float4 *>(b + 4 * (lane + 64 * i));
    }
    int grp_loaded = -1;
    if (scale || shift) {   // the first row's frame, fetched with the other parameters (before the reductions, not behind them)
        grp_loaded = mw / mod_rows;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            s4[i] = scale ? *reinterpret_cast<const float4 *>(scale + (long long)grp_loaded * mod_ld + 4 * (lane + 64 * i)) : make_float4(0.f, 0.f, 0.f, 0.f);
            h4[i] = shift ? *reinterpret_cast<const float4 *>(shift + (long long)grp_loaded * mod_ld + 4 * (lane + 64 * i)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    for (int it = 0; it < ITER; ++it) {
    const int m0 = mw + it * R;
    if (m0 >= M) break;
    float4 v[R][NV];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float *xr = x + (long long)min(m0 + r, M - 1) * ldx;
#pragma unroll
        for (int i = 0; i < NV; ++i) v[r][i] = *reinterpret_cast<const float4 *>(xr + 4 * (lane + 64 * i));
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int m = m0 + r;
        if (m >= M) break;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s += (v[r][i].x + v[r][i].y) + (v[r][i].z + v[r][i].w);
        const float mean = wave_sum(s) * (1.0f / (float)C);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float a = v[r][i].x - mean, bb = v[r][i].y - mean, c = v[r][i].z - mean, d = v[r][i].w - mean;
            q += (a * a + bb * bb) + (c * c + d * d);
        }
        const float rstd = rsqrtf(wave_sum(q) * (1.0f / (float)C) + eps);
        if (scale || shift) {
            const int gq = m / mod_rows;
            if (gq != grp_loaded) {   // wave-uniform; the frame changes every mod_rows rows
                grp_loaded = gq;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    s4[i] = scale ? *reinterpret_cast<const float4 *>(scale + (long long)gq * mod_ld + 4 * (lane + 64 * i)) : make_float4(0.f, 0.f, 0.f, 0.f);
                    h4[i] = shift ? *reinterpret_cast<const float4 *>(shift + (long long)gq * mod_ld + 4 * (lane + 64 * i)) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
        const long long orow = (long long)(m / grp_in) * grp_out + grp_off + (m % grp_in);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float y0 = (v[r][i].x - mean) * rstd * ww[i].x + bv[i].x, y1 = (v[r][i].y - mean) * rstd * ww[i].y + bv[i].y;
            float y2 = (v[r][i].z - mean) * rstd * ww[i].z + bv[i].z, y3 = (v[r][i].w - mean) * rstd * ww[i].w + bv[i].w;
            if (scale) { y0 *= 1.0f + s4[i].x; y1 *= 1.0f + s4[i].y; y2 *= 1.0f + s4[i].z; y3 *= 1.0f + s4[i].w; }
            if (shift) { y0 += h4[i].x; y1 += h4[i].y; y2 += h4[i].z; y3 += h4[i].w; }
            Out<DT>::st4(out + orow * ldo + 4 * (lane + 64 * i), y0, y1, y2, y3);
        }
    }
    }
}

template <int DT>
void launch_layernorm(const float *x, long long ldx, const float *w, const float *b, const float *scale, const float *shift, int mod_rows,
                      int mod_ld, typename Out<DT>::T *out, long long ldo, int M, int C, float eps, int grp_in, int grp_out, int grp_off,
                      hipStream_t stream) {
    // plain LayerNorm: one group of R rows per wave (more waves in flight); with AdaLN two groups per wave -- the frame's scale / shift
    // rows are hot L2 lines shared by ~64 waves, and halving those requests is worth more than the lost overlap (tools/bench_ln.py:
    // 70 -> 58 us at 49344 x 1024; plain 51 us either way with one group, 56 with two)
    constexpr int R = 4;
    const dim3 block(256);
    const bool mod = scale || shift;
    const dim3 grid_rows(vs::cdiv(M, 4 * R * (mod ? 2 : 1)));
#define VS_LN_LAUNCH(NV_, IT_) hipLaunchKernelGGL((layernorm_rows_kernel<DT, NV_, R, IT_>), grid_rows, block, 0, stream, x, ldx, w, b, scale, shift, mod_rows, mod_ld, out, ldo, M, eps, grp_in, grp_out, grp_off)
    if (C == 1024) { if (mod) VS_LN_LAUNCH(4, 2); else VS_LN_LAUNCH(4, 1); }
    else if (C == 768) { if (mod) VS_LN_LAUNCH(3, 2); else VS_LN_LAUNCH(3, 1); }
    else
        hipLaunchKernelGGL(layernorm_mod_kernel<DT>, dim3(vs::cdiv(M, 4)), block, 0, stream, x, ldx, w, b, scale, shift, mod_rows, mod_ld, out, ldo, M, C, eps, grp_in, grp_out, grp_off);
#undef VS_LN_LAUNCH
}

template <bool BF16>
__device__ __forceinline__ float ld16(const unsigned short *p) {
    if constexpr (BF16) return __uint_as_float(((unsigned)*p) << 16);
    else return (float)*reinterpret_cast<const _Float16 *>(p);
}
template <bool BF16>
__device__ __forceinline__ void st16(unsigned short *p, float v) {
    if constexpr (BF16) *p = Out<2>::cv(v);
    else *p = Out<1>::cv(v);
}

// one wave per row; lane = (q|k selector) * 32 + pair index; loops over heads. head_dim fixed at 64.  DT 0 f32 (the split / f32 classes'
// packed q|k|v and its gradient), 1 f16, 2 bf16.
template <int DT>
__global__ void __launch_bounds__(256)
rope_qk_kernel(typename Out<DT>::T *__restrict__ buf, long long ld, int rows, int H, int k_col, const int32_t *__restrict__ pos,
               const uint8_t *__restrict__ kind, float base2d, float theta1d, float dir) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int kd = kind ? kind[row] : 0;
    if (kd == 2) return;
    const int p = lane & 31, sel = lane >> 5;
    int iu, iv;
    float ang;
    if (kd == 0) {
        const int half = p >> 4, i = p & 15;
        iu = half * 32 + i; iv = iu + 16;
        ang = (float)pos[2 * row + half] / powf(base2d, (float)i / 16.0f);
    } else {
        iu = 2 * p; iv = iu + 1;
        ang = (float)pos[2 * row] / powf(theta1d, (float)(2 * p) / 64.0f);
    }
    float sn, cs;
    sincosf(ang * dir, &sn, &cs);  // dir = -1: the inverse rotation = the backward pass of the (orthogonal) embedding
    typename Out<DT>::T *r = buf + (long long)row * ld + (sel ? k_col : 0);
    for (int h = 0; h < H; ++h) {
        typename Out<DT>::T *pu = r + h * 64 + iu, *pv = r + h * 64 + iv;
        float u, v;
        if constexpr (DT == 0) { u = *pu; v = *pv; }
        else { u = ld16<DT == 2>(pu); v = ld16<DT == 2>(pv); }
        const float a = u * cs - v * sn, b = v * cs + u * sn;
        if constexpr (DT == 0) { *pu = a; *pv = b; }
        else { st16<DT == 2>(pu, a); st16<DT == 2>(pv, b); }
    }
}

// ---- tiny f32 linear layers of the camera-token path: out[m, n] = sum_k act(x[m, k]) w[n, k] + b[n] in f32 (the intrinsic embedding
// 9 -> 1024, backbone_vica.py:393,535-536; the pose / fov heads ReLU -> Linear(768 -> 8 | 2), vicasplat.py:118-138,179-205; kept in f32
// for pose accuracy).  Latency-bound work of a few thousand outputs: one wave per output, lanes stride K, wave reduction. ----
__global__ void __launch_bounds__(256)
linear_f32_kernel(const float *__restrict__ x, long long ldx, const float *__restrict__ w, long long ldw, const float *__restrict__ b,
                  float *__restrict__ out, long long ldo, int M, int N, int K, int relu_in) {
    const int lane = threadIdx.x & 63;
    const long long o = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= (long long)M * N) return;
    const int m = (int)(o / N), n = (int)(o - (long long)m * N);
    const float *xr = x + m * ldx, *wr = w + n * ldw;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) {
        float v = xr[k];
        if (relu_in) v = fmaxf(v, 0.f);
        s = fmaf(v, wr[k], s);
    }
    s = wave_sum(s);
    if (lane == 0) out[m * ldo + n] = s + (b ? b[n] : 0.f);
}

// silu(x) (f32) -> 16-bit / f32: the AdaLN modulation input SiLU(cam_norm(camera token)) (backbone_vica.py:210-212) as the GEMM operand
template <int DT>
__global__ void __launch_bounds__(256) silu_cast_kernel(const float *__restrict__ x, typename Out<DT>::T *__restrict__ out, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 v = reinterpret_cast<const float4 *>(x)[i];
    auto f = [](float a) { return a / (1.0f + __expf(-a)); };
    Out<DT>::st4(out + 4 * i, f(v.x), f(v.y), f(v.z), f(v.w));
}

}  // namespace

extern "C" int vs_linear_f32(const float *x, int64_t ldx, const float *w, int64_t ldw, const float *bias, float *out, int64_t ldo, int32_t M,
                             int32_t N, int32_t K, int32_t relu_in, vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(x && w && out, "vs_linear_f32: null pointer");
    VS_CHECK(M >= 0 && N > 0 && K > 0, "vs_linear_f32: bad sizes");
    if (M == 0) return 0;
    hipLaunchKernelGGL(linear_f32_kernel, dim3((unsigned)vs::cdiv64((int64_t)M * N, 4)), dim3(256), 0, stream, x, (long long)ldx, w, (long long)ldw,
                       bias, out, (long long)ldo, M, N, K, relu_in);
    VS_HIP(hipGetLastError());
    return 0;
}

extern "C" int vs_silu_cast(const float *x, void *out, int64_t n, int32_t out_dtype, vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(x && out, "vs_silu_cast: null pointer");
    VS_CHECK(n % 4 == 0 && out_dtype >= 0 && out_dtype <= 2, "vs_silu_cast: n must be a multiple of 4, out_dtype 0 f32 / 1 f16 / 2 bf16");
    if (n <= 0) return 0;
    dim3 grid((unsigned)vs::cdiv64(n / 4, 256)), block(256);
    switch (out_dtype) {
        case 0: hipLaunchKernelGGL(silu_cast_kernel<0>, grid, block, 0, stream, x, (float *)out, (long long)(n / 4)); break;
        case 1: hipLaunchKernelGGL(silu_cast_kernel<1>, grid, block, 0, stream, x, (unsigned short *)out, (long long)(n / 4)); break;
        default: hipLaunchKernelGGL(silu_cast_kernel<2>, grid, block, 0, stream, x, (unsigned short *)out, (long long)(n / 4)); break;
    }
    VS_HIP(hipGetLastError());
    return 0;
}

extern "C" int vs_layernorm_mod(const float *x, int64_t ldx, const float *w, const float *b, const float *scale,
                                const float *shift, int32_t mod_rows, int32_t mod_ld, void *out, int64_t ldo,
                                int32_t out_dtype, int32_t M, int32_t C, float eps, int32_t grp_in, int32_t grp_out,
                                int32_t grp_off, vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(x && w && b && out, "vs_layernorm_mod: null pointer");
    VS_CHECK(C > 0 && C % 4 == 0 && C <= 64 * 4 * kMaxVec, "vs_layernorm_mod: C=%d must be a multiple of 4 and <= %d", C, 64 * 4 * kMaxVec);
    VS_CHECK(ldx % 4 == 0 && ldo % 4 == 0, "vs_layernorm_mod: row strides must be multiples of 4 elements");
    VS_CHECK(out_dtype >= 0 && out_dtype <= 3, "vs_layernorm_mod: bad out_dtype %d (0 f32, 1 f16, 2 bf16, 3 packed split)", out_dtype);
    VS_CHECK(out_dtype != 3 || (C % 32 == 0 && ldo % 32 == 0 && ((uintptr_t)out & 127) == 0), "vs_layernorm_mod: a packed output needs C %% 32 == 0, ldo %% 32 == 0 and a 128-byte aligned buffer");
    if (M <= 0) return 0;
    if (grp_in <= 0) { grp_in = M; grp_out = M; grp_off = 0; }
    if (mod_rows <= 0) mod_rows = M;
    if (mod_ld <= 0) mod_ld = C;
    switch (out_dtype) {
        case 0: launch_layernorm<0>(x, ldx, w, b, scale, shift, mod_rows, mod_ld, (float *)out, ldo, M, C, eps, grp_in, grp_out, grp_off, stream); break;
        case 1: launch_layernorm<1>(x, ldx, w, b, scale, shift, mod_rows, mod_ld, (unsigned short *)out, ldo, M, C, eps, grp_in, grp_out, grp_off, stream); break;
        case 3: launch_layernorm<3>(x, ldx, w, b, scale, shift, mod_rows, mod_ld, (float *)out, ldo, M, C, eps, grp_in, grp_out, grp_off, stream); break;
        default: launch_layernorm<2>(x, ldx, w, b, scale, shift, mod_rows, mod_ld, (unsigned short *)out, ldo, M, C, eps, grp_in, grp_out, grp_off, stream); break;
    }
    VS_HIP(hipGetLastError());
    return 0;
}

extern "C" int vs_rope_qk_dir(void *buf, int64_t ld, int32_t rows, int32_t H, int32_t k_col, const int32_t *pos,
                              const uint8_t *kind, float base2d, float theta1d, float dir, int32_t dtype, vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(buf && pos, "vs_rope_qk: null pointer");
    VS_CHECK(dtype >= 0 && dtype <= 2, "vs_rope_qk: dtype must be 0 (f32), 1 (f16) or 2 (bf16)");
    VS_CHECK(dir == 1.0f || dir == -1.0f, "vs_rope_qk_dir: dir must be +1 (forward) or -1 (inverse / backward)");
    if (rows <= 0 || H <= 0) return 0;
    dim3 grid(vs::cdiv(rows, 4)), block(256);
    if (dtype == 2) hipLaunchKernelGGL(rope_qk_kernel<2>, grid, block, 0, stream, (unsigned short *)buf, ld, rows, H, k_col, pos, kind, base2d, theta1d, dir);
    else if (dtype == 1) hipLaunchKernelGGL(rope_qk_kernel<1>, grid, block, 0, stream, (unsigned short *)buf, ld, rows, H, k_col, pos, kind, base2d, theta1d, dir);
    else hipLaunchKernelGGL(rope_qk_kernel<0>, grid, block, 0, stream, (float *)buf, ld, rows, H, k_col, pos, kind, base2d, theta1d, dir);
    VS_HIP(hipGetLastError());
    return 0;
}

extern "C" int vs_rope_qk(void *buf, int64_t ld, int32_t rows, int32_t H, int32_t k_col, const int32_t *pos, const uint8_t *kind,
                          float base2d, float theta1d, int32_t dtype, vs_stream_t stream_) {
    return vs_rope_qk_dir(buf, ld, rows, H, k_col, pos, kind, base2d, theta1d, 1.0f, dtype, stream_);
}
