// 2-D rotary position embedding, in place, for gfx950.
// Replaces curope.rope_2d (/root/reference/src/model/encoder/backbone/croco/curope/curope.cpp:49-69,
// kernels.cu:17-108).  Math: head dim D = [Y half | X half]; in each half Q = D/4 pairs (i, i+Q),
// angle = pos * base^(-i/Q) * fwd;  u' = u cos - v sin, v' = v cos + u sin.
//
// MI355X shape: memory-bound.  One thread owns one (token, head, half, pair-index) quad... no: one thread owns the
// pair (i, i+Q) of one half for ALL heads is what the reference does (it loops h inside the block); here every
// thread handles one pair of one (token, head) so a wave covers 64 consecutive pairs = 2 heads x 2 halves x 16
// pairs (D=64): reads/writes of u are 16-element contiguous runs.  sin/cos are computed once per (token, half, i)
// and reused across heads via the loop over heads inside the thread block (grid = tokens).
#include "common.h"

namespace {

template <typename T> struct Cvt;
template <> struct Cvt<float> {
    static __device__ __forceinline__ float ld(const float *p) { return *p; }
    static __device__ __forceinline__ void st(float *p, float v) { *p = v; }
};
template <> struct Cvt<_Float16> {
    static __device__ __forceinline__ float ld(const _Float16 *p) { return (float)*p; }
    static __device__ __forceinline__ void st(_Float16 *p, float v) { *p = (_Float16)v; }
};
struct bf16_t { unsigned short x; };
template <> struct Cvt<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t *p) { return __uint_as_float(((unsigned)p->x) << 16); }
    static __device__ __forceinline__ void st(bf16_t *p, float v) {
        unsigned u = __float_as_uint(v);
        u += 0x7FFFu + ((u >> 16) & 1u);  // round to nearest even
        p->x = (unsigned short)(u >> 16);
    }
};

// grid = B*N tokens; block = D/2 threads (one per (half, i) pair index), loops over heads.
template <typename T>
__global__ void rope2d_kernel(T *__restrict__ tokens, const int64_t *__restrict__ pos, int N, int H, int D, int64_t sB,
                              int64_t sN, float base, float fwd) {
    const int b = blockIdx.x / N, n = blockIdx.x % N;
    const int Q = D / 4;
    const int t = threadIdx.x;       // 0 .. D/2-1
    const int half = t / Q;          // 0: y, 1: x
    const int i = t % Q;
    const float p = (float)pos[((int64_t)b * N + n) * 2 + half];
    const float inv_freq = fwd / powf(base, (float)i / (float)Q);
    float sn, cs;
    sincosf(p * inv_freq, &sn, &cs);
    T *row = tokens + b * sB + n * sN + half * (D / 2) + i;
    for (int h = 0; h < H; ++h) {
        T *pu = row + (int64_t)h * D;
        T *pv = pu + Q;
        const float u = Cvt<T>::ld(pu), v = Cvt<T>::ld(pv);
        Cvt<T>::st(pu, u * cs - v * sn);
        Cvt<T>::st(pv, v * cs + u * sn);
    }
}

}  // namespace

extern "C" int vs_rope2d(void *tokens, const int64_t *pos, int32_t B, int32_t N, int32_t H, int32_t D, int64_t sB, int64_t sN,
                         float base, float fwd, int32_t dtype, vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(tokens && pos, "vs_rope2d: null pointer");
    VS_CHECK(D > 0 && D % 4 == 0, "vs_rope2d: tokens.size(3) must be a multiple of 4 (got %d)", D);
    VS_CHECK(D / 2 <= 1024, "vs_rope2d: head dim too large");
    VS_CHECK(B >= 0 && N >= 0 && H >= 0, "vs_rope2d: negative size");
    if (B == 0 || N == 0 || H == 0) return 0;
    dim3 grid((unsigned)(B * N)), block((unsigned)(D / 2));
    switch (dtype) {
        case 0: hipLaunchKernelGGL(rope2d_kernel<float>, grid, block, 0, stream, (float *)tokens, pos, N, H, D, sB, sN, base, fwd); break;
        case 1: hipLaunchKernelGGL(rope2d_kernel<_Float16>, grid, block, 0, stream, (_Float16 *)tokens, pos, N, H, D, sB, sN, base, fwd); break;
        case 2: hipLaunchKernelGGL(rope2d_kernel<bf16_t>, grid, block, 0, stream, (bf16_t *)tokens, pos, N, H, D, sB, sN, base, fwd); break;
        default: VS_CHECK(false, "vs_rope2d: unsupported dtype %d", dtype);
    }
    VS_HIP(hipGetLastError());
    return 0;
}
