// Shared pieces of the GEMM / implicit-GEMM kernels: argument block, MFMA wrappers, LDS-DMA helper, fused epilogue.
#pragma once
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int BN = 128;  // BM = 32*MI (MI = 16-row fragments per wave in M: 4 -> 128 rows, 8 -> 256 rows)

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
    int m_lo;  // first logical row this launch covers (rows [m_lo, M)); tail launches of a split GEMM start past 0
    int a_grp_in, a_grp_out, a_grp_off;  // INPUT row map: A row of m = (m / a_grp_in) * a_grp_out + a_grp_off + m % a_grp_in
    // window-GEMM extras (7x7 RGB stem, gemm_kernel only): every a_sup_in row groups skip a_sup_extra more A rows (image
    // padding rows), and k-step kt reads its 32-wide slice at element offset kt * a_kstride (next image row), not kt * 32
    int a_sup_in, a_sup_extra, a_kstride;
    // epilogue 4 (STORE16 + RoPE on the q and k column blocks of a packed qkv projection, head_dim 64): per OUTPUT row
    // pos[2] and kind (0: 2-D pairs (i, i+16) per 32-half with pos[0]/pos[1], 1: 1-D interleaved pairs with pos[0], 2: none)
    const int32_t *rope_pos;
    const uint8_t *rope_kind;
    int rope_C;  // columns [0, C) = q, [C, 2C) = k, rest untouched
    float rope_l2base, rope_l2theta;  // log2 of the 2-D base / 1-D theta
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

template <bool BF16>
__device__ __forceinline__ float from16(unsigned short h) {
    if constexpr (BF16) return __uint_as_float(((unsigned)h) << 16);
    else return (float)*reinterpret_cast<_Float16 *>(&h);
}

// max(x, 0) on two packed 16-bit floats (f16 or bf16): clear every half whose sign bit is set
__device__ __forceinline__ unsigned relu2(unsigned x) {
    const unsigned m = ((x >> 15) & 0x00010001u) * 0xFFFFu;
    return x & ~m;
}

// exact-erf GELU (croco/blocks.py:60,68 uses nn.GELU()): erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below the
// 16-bit output's rounding) on v_rcp/v_exp instead of libm's branchy erff, because in a one-workgroup-per-CU kernel the
// epilogue is not hidden behind another workgroup's MFMAs.
__device__ __forceinline__ float gelu_erf(float v) {
    const float x = fabsf(v) * 0.70710678118654752440f;
    const float t = __frcp_rn(fmaf(0.3275911f, x, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = 1.0f - p * t * __expf(-x * x);  // erf(|v|/sqrt2)
    return 0.5f * v * (1.0f + copysignf(e, v));
}

// cos/sin of an angle given in radians on the hardware v_cos/v_sin (argument in revolutions; |angle| stays < 2^8 rev)
__device__ __forceinline__ void sincos_hw(float ang, float &sn, float &cs) {
    const float rev = ang * 0.15915494309189535f;
    sn = __builtin_amdgcn_sinf(rev);
    cs = __builtin_amdgcn_cosf(rev);
}

// XOR swizzle of the 16-byte chunk index (0..3) inside a 64-byte LDS row, keyed on (row >> 2) & 3, chosen so that the
// four 16-lane service groups of ds_read_b128 each touch 16 distinct 16-byte slots of a 256-byte bank row.
__device__ __forceinline__ int swz4(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }  // f = {0,2,3,1}

// LDS-DMA issued from inline asm: hipcc's waitcnt insertion does not see it, so the counted s_waitcnt vmcnt(N) placed by
// hand below are the only waits (with the builtin it drains vmcnt(0) before the first ds_read of every step, which
// serialises the pipeline).  lds_off must be wave-uniform (it goes to M0).
__device__ __forceinline__ void glds16(const void *gp, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_off), "v"(gp) : "memory");
}

// The epilogue patches are private to a wave: ordering its own LDS writes before its own reads needs no workgroup
// barrier (LDS operations of one wave execute in order) -- only a compiler/memory-model fence at wavefront scope.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- epilogue shared by the tile kernels.  A lane's accumulators are 4 rows x 1 column per fragment: storing them
// directly means 2-byte (or 4-byte) scattered stores.  Instead every wave transposes one 16 x 64 slab at a time through
// a private LDS patch (`scratch`: the operand tiles, dead after the last barrier; 16*68*4 B per wave) and writes /
// updates full 16-byte row chunks.  mw0 / nbase = first output row / column of the wave's (16*MI) x 64 tile. ----
template <bool BF16, int EPI, int MI>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs &g, f4 (&acc)[MI][4], int mw0, int nbase, void *scratch, int wid,
                                              int lane) {
    const int ccol = lane & 15, crow = (lane >> 4) * 4;
    float bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = nbase + j * 16 + ccol;
        bv[j] = (g.bias && n < g.N) ? g.bias[n] : 0.0f;
    }
    if constexpr (EPI == 0 || EPI == 1 || EPI == 4) {
        constexpr int PR = 64 + 8;  // halfs per patch row (144 B: 16-byte aligned, conflict-light)
        unsigned short *patch = reinterpret_cast<unsigned short *>(scratch) + wid * (16 * PR);
        const bool vec_ok = (g.ldo % 8 == 0) && ((reinterpret_cast<uintptr_t>(g.out) & 15) == 0) && (nbase + 64 <= g.N);
        // RoPE (EPI 4): a wave's 64 columns are one head; in the accumulator layout the 2-D pair (c, c+16) of a 32-half is
        // (fragment 2h, fragment 2h+1) of the SAME lane, the 1-D pair (2p, 2p+1) is the neighbouring lane.
        [[maybe_unused]] bool rope_on = false;
        [[maybe_unused]] float inv2d = 0.f;
        if constexpr (EPI == 4) {
            rope_on = nbase < 2 * g.rope_C && nbase + 64 <= g.N;
            inv2d = __builtin_amdgcn_exp2f(-(float)ccol * (1.0f / 16.0f) * g.rope_l2base);
        }
        [[maybe_unused]] int2 rope_p[4] = {};
        [[maybe_unused]] int rope_k[4] = {};
        [[maybe_unused]] auto rope_fetch = [&](int i_) {  // pos / kind of the 4 output rows this lane holds in slab i_
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = min(mw0 + i_ * 16 + crow + r, g.M - 1);
                const size_t orow = (size_t)(m / g.grp_in) * g.grp_out + g.grp_off + (m % g.grp_in);
                rope_p[r] = *reinterpret_cast<const int2 *>(g.rope_pos + 2 * orow);
                rope_k[r] = g.rope_kind ? (int)g.rope_kind[orow] : 0;
            }
        };
        if constexpr (EPI == 4) {
            if (rope_on) rope_fetch(0);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            float vv[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    vv[j][r] = acc[i][j][r] + bv[j];
                    if constexpr (EPI == 1) vv[j][r] = gelu_erf(vv[j][r]);
                }
            if constexpr (EPI == 4) {
                if (rope_on) {
                    // this slab's row table entries were fetched while the previous slab was processed
                    int2 pcur[4]; int kcur[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { pcur[r] = rope_p[r]; kcur[r] = rope_k[r]; }
                    if (i + 1 < MI) rope_fetch(i + 1);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kd = kcur[r];
                        const float p0 = (float)pcur[r].x, p1 = (float)pcur[r].y;
                        if (kd == 0) {
                            float s0, c0, s1, c1;
                            sincos_hw(p0 * inv2d, s0, c0);
                            sincos_hw(p1 * inv2d, s1, c1);
                            const float u0 = vv[0][r], w0 = vv[1][r], u1 = vv[2][r], w1 = vv[3][r];
                            vv[0][r] = u0 * c0 - w0 * s0; vv[1][r] = w0 * c0 + u0 * s0;
                            vv[2][r] = u1 * c1 - w1 * s1; vv[3][r] = w1 * c1 + u1 * s1;
                        }
                        // camera-token rows (1 in 258): interleaved pairs live in lanes (2p, 2p+1); whole wave joins the swap
                        if (__builtin_amdgcn_ballot_w64(kd == 1) != 0) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float other = __shfl_xor(vv[j][r], 1, 64);
                                if (kd == 1) {
                                    float sn, cs;
                                    sincos_hw(p0 * __builtin_amdgcn_exp2f(-(float)((j * 16 + ccol) >> 1) * (1.0f / 32.0f) * g.rope_l2theta), sn, cs);
                                    vv[j][r] = (ccol & 1) ? vv[j][r] * cs + other * sn : vv[j][r] * cs - other * sn;
                                }
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) patch[(crow + r) * PR + j * 16 + ccol] = to16<BF16>(vv[j][r]);
            wave_lds_sync();
            // 16 rows x 8 chunks of 8 halfs = 128 chunks, 2 per lane
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int q = lane + 64 * c, prow = q >> 3, pch = q & 7;
                const int m = mw0 + i * 16 + prow;
                if (m < g.M) {
                    const size_t orow = (size_t)(m / g.grp_in) * g.grp_out + g.grp_off + (m % g.grp_in);
                    unsigned short *dst = reinterpret_cast<unsigned short *>(g.out) + orow * g.ldo + nbase + pch * 8;
                    const uint4 val = *reinterpret_cast<const uint4 *>(&patch[prow * PR + pch * 8]);
                    if (vec_ok) {
                        *reinterpret_cast<uint4 *>(dst) = val;
                    } else {
                        const unsigned short *hv = reinterpret_cast<const unsigned short *>(&val);
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (nbase + pch * 8 + e < g.N) dst[e] = hv[e];
                    }
                }
            }
            wave_lds_sync();
        }
    } else {
        constexpr int PR = 64 + 4;  // floats per patch row (272 B)
        float *patch = reinterpret_cast<float *>(scratch) + wid * (16 * PR);
        const bool vec_ok = (g.ldo % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.out) & 15) == 0) && (nbase + 64 <= g.N);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mw0 + i * 16 + crow + r;
                const float *gp = nullptr;
                if (EPI == 2 && g.gate && m < g.M) gp = g.gate + (size_t)(m / g.gate_rows) * g.gate_ld;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = nbase + j * 16 + ccol;
                    float v = acc[i][j][r] + bv[j];
                    if (gp && n < g.N) v *= 1.0f + gp[n];
                    patch[(crow + r) * PR + j * 16 + ccol] = v;
                }
            }
            wave_lds_sync();
            // 16 rows x 16 chunks of 4 floats = 256 chunks, 4 per lane
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int q = lane + 64 * c, prow = q >> 4, pch = q & 15;
                const int m = mw0 + i * 16 + prow;
                if (m < g.M) {
                    const size_t orow = (size_t)(m / g.grp_in) * g.grp_out + g.grp_off + (m % g.grp_in);
                    float *dst = reinterpret_cast<float *>(g.out) + orow * g.ldo + nbase + pch * 4;
                    const float4 val = *reinterpret_cast<const float4 *>(&patch[prow * PR + pch * 4]);
                    if (vec_ok) {
                        if constexpr (EPI == 2) {
                            float4 o = *reinterpret_cast<float4 *>(dst);
                            o.x += val.x; o.y += val.y; o.z += val.z; o.w += val.w;
                            *reinterpret_cast<float4 *>(dst) = o;
                        } else {
                            *reinterpret_cast<float4 *>(dst) = val;
                        }
                    } else {
                        const float *fv = reinterpret_cast<const float *>(&val);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (nbase + pch * 4 + e < g.N) {
                                if constexpr (EPI == 2) dst[e] += fv[e]; else dst[e] = fv[e];
                            }
                    }
                }
            }
            wave_lds_sync();
        }
    }
}

}  // namespace
