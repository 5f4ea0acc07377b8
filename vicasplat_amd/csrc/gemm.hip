// MFMA GEMM with fused epilogues for the ViT blocks (gfx950).
//
//   C[m, n] = sum_k A[m, k] * W[n, k]  (+ bias[n])         A: [M,K] 16-bit, W: [N,K] 16-bit (nn.Linear layout)
//
// Replaces the cuBLAS/TF32 nn.Linear calls of the reference on the hot path
// (croco/blocks.py:73-79,94-112; backbone_vica.py:88,101,168-170,124-125,189).  gfx950 has no TF32: operands are
// f16 (10-bit mantissa, the same as TF32) or bf16, accumulation is f32 on v_mfma_f32_16x16x32_{f16,bf16}, and the
// residual stream stays f32 (the epilogue adds into it), which is the TF32-class precision the reference runs at.
//
// Tiling (CDNA4-shaped, not a warp-shaped CUDA tile): 128x128x64 block tile, 256 threads = 4 waves as 2x2, each
// wave owns 64x64 = 4x4 MFMA 16x16 fragments (64 accumulator VGPRs), operands staged global -> VGPR -> LDS with
// 144-byte padded rows (conflict-free ds_read_b128 fragment reads), double-buffered, one barrier pair per K tile.
// blockIdx is remapped so that consecutive tiles of one XCD share A/W panels in that XCD's L2.
//
// Epilogues (all fused, nothing round-trips HBM):
//   0 STORE16   out16[row(m), n] = acc + bias
//   1 GELU16    out16[row(m), n] = gelu_erf(acc + bias)                (croco/blocks.py:60,68: exact GELU)
//   2 RESID32   x32[row(m), n]  += (1 + gate[m / gate_rows, n]) * (acc + bias)   (backbone_vica.py:274-278,302,327,331)
//   3 STORE32   out32[row(m), n] = acc + bias
// row(m) = (m / grp_in) * grp_out + grp_off + m % grp_in lets a GEMM write straight into a larger token buffer
// (e.g. the 257 image tokens of a frame behind the frame's camera token).
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDS_ROW = BK + 8;  // halfs; 144 B row stride

struct GemmArgs {
    const void *A;
    const void *W;
    const float *bias;
    void *out;
    const float *gate;
    int M, N, K;
    int lda, ldw, ldo;
    int grp_in, grp_out, grp_off;
    int gate_rows;  // rows of A per gate vector
    int gate_ld;
    int a_grp_in, a_grp_out, a_grp_off;  // INPUT row map: A row of m = (m / a_grp_in) * a_grp_out + a_grp_off + m % a_grp_in
};

template <bool BF16>
__device__ __forceinline__ f4 mfma(const uint4 &a, const uint4 &b, f4 c) {
    if constexpr (BF16) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf8 *>(&a), *reinterpret_cast<const bf8 *>(&b), c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const half8 *>(&a), *reinterpret_cast<const half8 *>(&b), c, 0, 0, 0);
    }
}

template <bool BF16>
__device__ __forceinline__ unsigned short to16(float v) {
    if constexpr (BF16) {
        unsigned u = __float_as_uint(v);
        u += 0x7FFFu + ((u >> 16) & 1u);
        return (unsigned short)(u >> 16);
    } else {
        _Float16 h = (_Float16)v;
        return *reinterpret_cast<unsigned short *>(&h);
    }
}

template <bool BF16, int EPI>
__global__ void __launch_bounds__(256) gemm_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) unsigned short sA[2][BM * LDS_ROW];
    __shared__ __attribute__((aligned(16))) unsigned short sW[2][BN * LDS_ROW];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;

    // XCD-aware tile order: blocks b, b+8, b+16.. run on the same XCD (observed, speed only) -> give each XCD a
    // contiguous range of tiles so neighbouring tiles (same A panel) hit that XCD's L2.
    const int tiles_n = (g.N + BN - 1) / BN;
    const int tiles_m = (g.M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // staging: tile = 128 rows x 64 halfs = 128 x 8 x 16B; thread t loads rows (t>>3) + 32*i, 16B chunk (t&7)
    const unsigned short *A = reinterpret_cast<const unsigned short *>(g.A);
    const unsigned short *W = reinterpret_cast<const unsigned short *>(g.W);
    const int srow = tid >> 3, schunk = tid & 7;
    size_t oa[4], ow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra_ = min(m0 + srow + 32 * i, g.M - 1);
        const int rw_ = min(n0 + srow + 32 * i, g.N - 1);
        const size_t arow = (size_t)(ra_ / g.a_grp_in) * g.a_grp_out + g.a_grp_off + (ra_ % g.a_grp_in);
        oa[i] = arow * g.lda + schunk * 8;
        ow[i] = (size_t)rw_ * g.ldw + schunk * 8;
    }
    uint4 ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3;  // named registers: arrays here end up in scratch
#define VS_GLOAD(k0)                                                                                                  \
    ra0 = *reinterpret_cast<const uint4 *>(A + oa[0] + (k0)); rw0 = *reinterpret_cast<const uint4 *>(W + ow[0] + (k0)); \
    ra1 = *reinterpret_cast<const uint4 *>(A + oa[1] + (k0)); rw1 = *reinterpret_cast<const uint4 *>(W + ow[1] + (k0)); \
    ra2 = *reinterpret_cast<const uint4 *>(A + oa[2] + (k0)); rw2 = *reinterpret_cast<const uint4 *>(W + ow[2] + (k0)); \
    ra3 = *reinterpret_cast<const uint4 *>(A + oa[3] + (k0)); rw3 = *reinterpret_cast<const uint4 *>(W + ow[3] + (k0));
#define VS_LS1(buf, i, va, vw)                                                                   \
    *reinterpret_cast<uint4 *>(&sA[buf][(srow + 32 * i) * LDS_ROW + schunk * 8]) = va;           \
    *reinterpret_cast<uint4 *>(&sW[buf][(srow + 32 * i) * LDS_ROW + schunk * 8]) = vw;
#define VS_LSTORE(buf) VS_LS1(buf, 0, ra0, rw0) VS_LS1(buf, 1, ra1, rw1) VS_LS1(buf, 2, ra2, rw2) VS_LS1(buf, 3, ra3, rw3)

    f4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fk = (lane >> 4) * 8;
    const int nk = g.K / BK;
    VS_GLOAD(0)
    VS_LSTORE(0)
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) { VS_GLOAD((kt + 1) * BK) }
#pragma unroll
        for (int ks = 0; ks < BK; ks += 32) {
            uint4 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = *reinterpret_cast<const uint4 *>(&sA[buf][(wr * 64 + i * 16 + frow) * LDS_ROW + ks + fk]);
                fb[i] = *reinterpret_cast<const uint4 *>(&sW[buf][(wc * 64 + i * 16 + frow) * LDS_ROW + ks + fk]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma<BF16>(fa[i], fb[j], acc[i][j]);
        }
        if (kt + 1 < nk) {
            VS_LSTORE(buf ^ 1)
            __syncthreads();
        }
    }
#undef VS_GLOAD
#undef VS_LSTORE
#undef VS_LS1

    // epilogue: fragment (i,j): rows m0 + wr*64 + i*16 + (lane>>4)*4 + r, col n0 + wc*64 + j*16 + (lane&15)
    const int ccol = lane & 15, crow = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wc * 64 + j * 16 + ccol;
        if (n >= g.N) continue;
        const float bv = g.bias ? g.bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wr * 64 + i * 16 + crow + r;
                if (m >= g.M) continue;
                const size_t orow = (size_t)(m / g.grp_in) * g.grp_out + g.grp_off + (m % g.grp_in);
                float v = acc[i][j][r] + bv;
                if constexpr (EPI == 0) {
                    reinterpret_cast<unsigned short *>(g.out)[orow * g.ldo + n] = to16<BF16>(v);
                } else if constexpr (EPI == 1) {
                    v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                    reinterpret_cast<unsigned short *>(g.out)[orow * g.ldo + n] = to16<BF16>(v);
                } else if constexpr (EPI == 2) {
                    float *o = reinterpret_cast<float *>(g.out) + orow * g.ldo + n;
                    const float gt = g.gate ? g.gate[(size_t)(m / g.gate_rows) * g.gate_ld + n] : 0.0f;
                    *o = *o + (1.0f + gt) * v;
                } else {
                    reinterpret_cast<float *>(g.out)[orow * g.ldo + n] = v;
                }
            }
        }
    }
}

template <bool BF16>
int launch(const GemmArgs &g, int epi, hipStream_t stream) {
    const int nwg = vs::cdiv(g.M, BM) * vs::cdiv(g.N, BN);
    dim3 grid(nwg), block(256);
    switch (epi) {
        case 0: hipLaunchKernelGGL((gemm_kernel<BF16, 0>), grid, block, 0, stream, g); break;
        case 1: hipLaunchKernelGGL((gemm_kernel<BF16, 1>), grid, block, 0, stream, g); break;
        case 2: hipLaunchKernelGGL((gemm_kernel<BF16, 2>), grid, block, 0, stream, g); break;
        case 3: hipLaunchKernelGGL((gemm_kernel<BF16, 3>), grid, block, 0, stream, g); break;
        default: vs::set_error("vs_gemm_bias_act: unknown epilogue %d", epi); return -1;
    }
    return 0;
}

}  // namespace

extern "C" int vs_gemm_bias_act(const void *A, const void *W, const float *bias, void *out, const float *gate, int32_t M,
                                int32_t N, int32_t K, int32_t lda, int32_t ldw, int32_t ldo, int32_t epilogue, int32_t dtype,
                                int32_t grp_in, int32_t grp_out, int32_t grp_off, int32_t gate_rows, int32_t gate_ld,
                                int32_t a_grp_in, int32_t a_grp_out, int32_t a_grp_off, vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(A && W && out, "vs_gemm_bias_act: null pointer");
    VS_CHECK(M >= 0 && N > 0 && K > 0, "vs_gemm_bias_act: bad sizes M=%d N=%d K=%d", M, N, K);
    VS_CHECK(K % BK == 0, "vs_gemm_bias_act: K=%d must be a multiple of %d", K, BK);
    VS_CHECK(lda % 8 == 0 && ldw % 8 == 0, "vs_gemm_bias_act: lda/ldw must be multiples of 8 elements (16-byte rows)");
    VS_CHECK(dtype == 1 || dtype == 2, "vs_gemm_bias_act: dtype must be 1 (f16) or 2 (bf16)");
    VS_CHECK((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0,
             "vs_gemm_bias_act: A and W must be 16-byte aligned");
    if (M == 0) return 0;
    GemmArgs g;
    g.A = A; g.W = W; g.bias = bias; g.out = out; g.gate = gate;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldo = ldo;
    g.grp_in = grp_in > 0 ? grp_in : (M > 0 ? M : 1);
    g.grp_out = grp_out > 0 ? grp_out : g.grp_in;
    g.grp_off = grp_off;
    g.gate_rows = gate_rows > 0 ? gate_rows : (M > 0 ? M : 1);
    g.gate_ld = gate_ld > 0 ? gate_ld : N;
    g.a_grp_in = a_grp_in > 0 ? a_grp_in : (M > 0 ? M : 1);
    g.a_grp_out = a_grp_out > 0 ? a_grp_out : g.a_grp_in;
    g.a_grp_off = a_grp_off;
    const int rc = dtype == 2 ? launch<true>(g, epilogue, stream) : launch<false>(g, epilogue, stream);
    if (rc) return rc;
    VS_HIP(hipGetLastError());
    return 0;
}
