// Shared pieces of the GEMM / implicit-GEMM kernels: argument block, MFMA wrappers, LDS-DMA helper, fused epilogue.
#pragma once
#include "common.h"

#include <type_traits>

namespace {
typedef unsigned u2v __attribute__((ext_vector_type(2)));

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
    const float *resid;  // epilogue 2: source of the residual stream (same row layout as out); null = out itself (in-place update)
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
    // tap-fused weight gradient (gemm_kernel, epilogue 2): ntaps > 0 -> the launch is ntaps GEMMs sharing A; tap t reads the
    // W operand shifted by tap_shift[t] elements and adds into out + t * tap_out_stride; workgroup order is (k-slice, tap, tile)
    // so that the taps of one K slice run together and their re-reads of A / X hit the caches instead of HBM
    int ntaps;
    long long tap_out_stride;
    int tap_shift[9];
    int ksplit;  // > 1 (epilogue 2 only): blockIdx.y-th of ksplit equal K ranges, summed into out with f32 atomics
    // weight-gradient launches with a workspace: K slice s (tap t) STORES its partial tile to partials[((s * ntaps + t) * M + m) * N + n]
    // (plain coalesced stores; splitk_reduce_kernel sums the slices) instead of meeting the other slices through atomics --
    // f32 atomics sustain only ~0.3 TB/s on this part, a tenth of the plain store rate.  The kernels then run the epilogue with
    // ksplit = -1 ("store").
    float *partials;
    // element offset of K slice s within A / W: s * a_slice_stride (0: the slices are consecutive column ranges of one matrix,
    // i.e. K / ksplit).  Weight-gradient operands are stored slice-blocked, [slice][channel][slice length], so that the rows a
    // workgroup walks are a few KB apart instead of the whole reduction length (4 M pixels = 8 MB: one TLB entry per row and
    // K step otherwise)
    long long a_slice_stride, w_slice_stride;
    int k_valid;  // reduction-major (TN) weight gradient: real number of reduction rows; rows in [k_valid, K) read as zeros
    int conv_H, conv_W;  // reduction-major 3x3-conv weight gradient: image size (reduction row = pixel n*H*W + y*W + x of an NHWC tensor)
    // epilogue 4 (STORE16 + RoPE on the q and k column blocks of a packed qkv projection, head_dim 64): per OUTPUT row
    // pos[2] and kind (0: 2-D pairs (i, i+16) per 32-half with pos[0]/pos[1], 1: 1-D interleaved pairs with pos[0], 2: none)
    const int32_t *rope_pos;
    const uint8_t *rope_kind;
    int rope_C;  // columns [0, C) = q, [C, 2C) = k, rest untouched
    float rope_l2base, rope_l2theta;  // log2 of the 2-D base / 1-D theta
    int stagger;  // experiment: first-round workgroups of gemm256_kernel sleep (bid % 8) * stagger * ~4 us before starting
    int row_band; // gemm256_kernel: > 0 = tiles are walked in bands of row_band row tiles, column tile slowest inside a band (L2: see gemm256_kernel)
    int tap_on_a;     // tap-fused weight gradient: the tap shift moves the A operand instead of W (split class: a packed W cannot be shifted)
    int out_packed;   // split operands, epilogues 0 / 1 / 3: the output is written in the packed (hi, lo) form (the A operand of the next GEMM)
    int a_packed;     // split operands: A is ALREADY in the packed (hi, lo) form of vs_split_pack_weight (scale 2^0): the kernels skip the conversion
    float acc_scale;  // split operands (kDtSplit): the packed weights carry a power-of-two scale 2^e; the epilogue multiplies the accumulators by 2^-e
};

// Operand dtype of the MFMA kernels (template parameter `BF16` of every kernel below: the name predates the third value).
//   0 f16, 1 bf16: v_mfma_f32_16x16x32_{f16,bf16}, a fragment register quad = 8 consecutive k values of one tile row.
//   2 f32 (kDtF32): the reference-precision path (fp32 weights and activations; gfx950 has no TF32 -- SURVEY 7-5 / DESIGN 2).
//     The SAME kernels run it: an f32 operand array is addressed as an array of 2-byte units with doubled strides / K, so
//     a staged 128-byte (64-byte) LDS row holds 32 (16) floats instead of 64 (32) halves, a fragment register quad is 4
//     consecutive k, and one 16x16x32 MFMA becomes four v_mfma_f32_16x16x4_f32 (exact f32 products and sums == an fmaf chain;
//     64 FLOP/clk/SIMD = 1/16 of the 16-bit rate).  Lane (row = l & 15, group = l >> 4) feeds float s of its quad to step s,
//     i.e. k = 4 * group + s for BOTH operands -- any bijection of k onto (step, group) is a valid summation order.
//   3 split (kDtSplit): f32-class results at 1/3 of the 16-bit matrix rate (SURVEY 7-5 "fp32 MFMA or 3x split").  Activations
//     stay f32 in HBM and LDS (same 2-byte-unit addressing as kDtF32); a lane converts its A fragment -- 8 floats -- in registers to
//     hi = rne16(x) and lo = rne16(x - hi) (the difference is exact in f32, so hi + lo carries x to 2^-23), the weights are packed
//     ONCE on the host side of the ABI (vs_split_pack_weight) as [hi 32 halves | lo 32 halves] per block of 32 k, scaled by a power
//     of two so that lo stays a normal f16, and the product is three v_mfma_f32_16x16x32_f16: lo_w hi_a + hi_w lo_a + hi_w hi_a
//     (the dropped lo lo term is 2^-22 relative).  Within a block the packed weights are ordered so that chunk g (8 halves) holds
//     k = {4g..4g+3, 16+4g..16+4g+3}: exactly the floats lane group g reads from an f32 A row (chunks g and 4+g).
constexpr int kDtF32 = 2;
constexpr int kDtSplit = 3;
// operand classes whose activations / outputs are f32 arrays (addressed in 2-byte units by the kernels)
__host__ __device__ constexpr bool is_f32io(int dt) { return dt == kDtF32 || dt == kDtSplit; }

__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// 8 floats (the two 16-byte chunks a lane reads from an f32 A row) -> hi / lo f16x8 MFMA operands, in place (f0 <- hi, f1 <- lo).
// Two forms with identical results (round to nearest even; x - float(hi) is exact in f32):
//   split8      compiler-visible instructions only.  Its results may feed an MFMA in the next issue slot, and the VALU-write -> MFMA-read
//               wait state is something only the compiler's hazard recogniser inserts -- it does not look inside inline asm (found the
//               hard way: an inline-asm v_max_f32 in front of the f32 MFMAs left one fragment stale in 1/64 of the outputs).
//   split8_lds  x - float(hi) on v_fma_mix_f32 (f16 source operand, f32 result: one instruction instead of convert + subtract), 16 VALU
//               per 8 floats; for results that go to LDS or memory, never straight into an MFMA.
typedef _Float16 half2v_ __attribute__((ext_vector_type(2)));
typedef float float2v_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split8(uint4 &f0, uint4 &f1) {
    const float x[8] = {__uint_as_float(f0.x), __uint_as_float(f0.y), __uint_as_float(f0.z), __uint_as_float(f0.w),
                        __uint_as_float(f1.x), __uint_as_float(f1.y), __uint_as_float(f1.z), __uint_as_float(f1.w)};
    unsigned h[4], l[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const half2v_ hh = __builtin_convertvector(float2v_{x[2 * p], x[2 * p + 1]}, half2v_);      // v_cvt_pk_f16_f32
        const half2v_ ll = __builtin_convertvector(float2v_{x[2 * p] - (float)hh.x, x[2 * p + 1] - (float)hh.y}, half2v_);
        h[p] = __builtin_bit_cast(unsigned, hh);
        l[p] = __builtin_bit_cast(unsigned, ll);
    }
    f0 = make_uint4(h[0], h[1], h[2], h[3]);
    f1 = make_uint4(l[0], l[1], l[2], l[3]);
}
__device__ __forceinline__ void split8_lds(uint4 &f0, uint4 &f1) {
    const float x[8] = {__uint_as_float(f0.x), __uint_as_float(f0.y), __uint_as_float(f0.z), __uint_as_float(f0.w),
                        __uint_as_float(f1.x), __uint_as_float(f1.y), __uint_as_float(f1.z), __uint_as_float(f1.w)};
    unsigned h[4], l[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        h[p] = cvt_pk_f16(x[2 * p], x[2 * p + 1]);
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h[p]), "v"(x[2 * p]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h[p]), "v"(x[2 * p + 1]));
        l[p] = cvt_pk_f16(r0, r1);
    }
    f0 = make_uint4(h[0], h[1], h[2], h[3]);
    f1 = make_uint4(l[0], l[1], l[2], l[3]);
}
// Four consecutive columns n .. n + 3 (n % 4 == 0) of an f32 activation row, written in the packed (hi, lo) form of vs_split_pack_weight
// (scale 2^0): per block of 32 columns 32 hi halves then 32 lo halves, chunk g = columns {4g..4g+3, 16+4g..16+4g+3}.  `row` = start of the
// row in 4-byte units (rows are whole 128-byte blocks).  What lets a producer hand the next GEMM an operand it need not convert.
__device__ __forceinline__ void store_split4(float *row, int n, float a, float b, float c, float d) {
    uint4 f0 = make_uint4(__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d)), f1 = f0;
    // (split8_lds on a duplicated quad: lanes of f1 are ignored)
    unsigned h[2], l[2];
    h[0] = cvt_pk_f16(a, b); h[1] = cvt_pk_f16(c, d);
    float r0, r1, r2, r3;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h[0]), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h[0]), "v"(b));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(h[1]), "v"(c));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(h[1]), "v"(d));
    l[0] = cvt_pk_f16(r0, r1); l[1] = cvt_pk_f16(r2, r3);
    (void)f0; (void)f1;
    const int kk = n & 31;
    unsigned short *o = reinterpret_cast<unsigned short *>(row + (n & ~31)) + ((kk & 15) >> 2) * 8 + (kk >> 4) * 4;
    *reinterpret_cast<uint2 *>(o) = make_uint2(h[0], h[1]);
    *reinterpret_cast<uint2 *>(o + 32) = make_uint2(l[0], l[1]);
}
// Columns n .. n + 3 and n + 16 .. n + 19 (n % 32 < 16, n % 4 == 0) of a row: together they are ONE 16-byte chunk of hi halves and one of lo
// halves (chunk g = (n & 15) >> 2 holds exactly these eight columns), so the pair leaves as two 16-byte stores -- the same number of store
// instructions as the f32 row (round 4: store_split4 per fragment was two 8-byte stores each, and the GEMM store tail is issue-bound).
__device__ __forceinline__ void store_split8(float *row, int n, const float (&a)[4], const float (&b)[4]) {
    unsigned h[4], l[4];
    float r[8];
    h[0] = cvt_pk_f16(a[0], a[1]); h[1] = cvt_pk_f16(a[2], a[3]); h[2] = cvt_pk_f16(b[0], b[1]); h[3] = cvt_pk_f16(b[2], b[3]);
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[0]) : "v"(h[0]), "v"(a[0]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[1]) : "v"(h[0]), "v"(a[1]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[2]) : "v"(h[1]), "v"(a[2]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[3]) : "v"(h[1]), "v"(a[3]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[4]) : "v"(h[2]), "v"(b[0]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[5]) : "v"(h[2]), "v"(b[1]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[6]) : "v"(h[3]), "v"(b[2]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[7]) : "v"(h[3]), "v"(b[3]));
    l[0] = cvt_pk_f16(r[0], r[1]); l[1] = cvt_pk_f16(r[2], r[3]); l[2] = cvt_pk_f16(r[4], r[5]); l[3] = cvt_pk_f16(r[6], r[7]);
    unsigned short *o = reinterpret_cast<unsigned short *>(row + (n & ~31)) + ((n & 15) >> 2) * 8;
    *reinterpret_cast<uint4 *>(o) = make_uint4(h[0], h[1], h[2], h[3]);
    *reinterpret_cast<uint4 *>(o + 32) = make_uint4(l[0], l[1], l[2], l[3]);
}
// max(x, 0) of an f32 value whose result goes to LDS / memory (one VALU; inline asm: see split8 for why not in front of an MFMA)
__device__ __forceinline__ unsigned relu_f32_lds(unsigned x) {
    unsigned r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}

template <int BF16>
__device__ __forceinline__ f4 mfma(const uint4 &a, const uint4 &b, f4 c) {
    if constexpr (BF16 == kDtF32) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    } else if constexpr (BF16 == 1) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf8 *>(&a), *reinterpret_cast<const bf8 *>(&b), c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const half8 *>(&a), *reinterpret_cast<const half8 *>(&b), c, 0, 0, 0);
    }
}

// One K step of a (W fragment pair, A fragment pair): two consecutive 32-wide MFMA steps for the 16-bit / f32 operand classes; for
// split operands (b0, b1) = (hi, lo) of the packed weights and (a0, a1) = (hi, lo) of the converted activations: three products.
template <int BF16>
__device__ __forceinline__ f4 mma2(const uint4 &b0, const uint4 &b1, const uint4 &a0, const uint4 &a1, f4 c) {
    if constexpr (BF16 == kDtSplit) {
        c = mfma<0>(b1, a0, c);
        c = mfma<0>(b0, a1, c);
        return mfma<0>(b0, a0, c);
    } else {
        c = mfma<BF16>(b0, a0, c);
        return mfma<BF16>(b1, a1, c);
    }
}

// two floats -> one packed pair of 16-bit values (low half = a), round to nearest even: ONE instruction on gfx950 (v_cvt_pk_f16_f32 /
// v_cvt_pk_bf16_f32) where two scalar converts plus the shift / or (and, for bf16, the integer rounding sequence) were 4-10
template <int BF16>
__device__ __forceinline__ unsigned pack16x2(float a, float b) {
    static_assert(!is_f32io(BF16), "f32 operands are stored as floats");
    unsigned r;
    if constexpr (BF16 == 1) asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    else asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int BF16>
__device__ __forceinline__ unsigned short to16(float v) {
    static_assert(!is_f32io(BF16), "f32 operands are stored as floats, not through to16");
    if constexpr (BF16 == 1) {
        unsigned u = __float_as_uint(v);
        u += 0x7FFFu + ((u >> 16) & 1u);
        return (unsigned short)(u >> 16);
    } else {
        _Float16 h = (_Float16)v;
        return *reinterpret_cast<unsigned short *>(&h);
    }
}

template <int BF16>
__device__ __forceinline__ float from16(unsigned short h) {
    static_assert(!is_f32io(BF16), "f32 operands are read as floats, not through from16");
    if constexpr (BF16 == 1) return __uint_as_float(((unsigned)h) << 16);
    else return (float)*reinterpret_cast<_Float16 *>(&h);
}

// max(x, 0) on two packed 16-bit floats (f16 or bf16): clear every half whose sign bit is set
__device__ __forceinline__ unsigned relu2(unsigned x) {
    const unsigned m = ((x >> 15) & 0x00010001u) * 0xFFFFu;
    return x & ~m;
}
// the same on one fragment register of operand dtype DT (f32: one float per register)
template <int DT>
__device__ __forceinline__ unsigned relu_reg(unsigned x) {
    if constexpr (is_f32io(DT)) {
        return x & ~(unsigned)((int)x >> 31);   // (compiler-visible on purpose: the result feeds an MFMA, see split8)
    } else if constexpr (DT == 0) {
        // f16: ONE packed max per register (relu2 is shift / and / quarter-rate 32-bit multiply / and-not: ~7 issue slots, and the
        // implicit-GEMM main loop applies this to 32 A-fragment registers per K-tile beside 64 MFMAs of a one-wave-per-SIMD kernel).
        // (-0.0 and NaN inputs: max(-0.0, +0.0) = +0.0 like the bit trick; a NaN activation is a NaN either way downstream.)
        unsigned r;
        asm("v_pk_max_f16 %0, %1, 0" : "=v"(r) : "v"(x));
        return r;
    } else {
        // bf16: sign masks of both halves with one packed arithmetic shift, then and-not
        unsigned m;   // (the shift count comes from a register: a VOP3P inline constant only reaches the LOW half)
        asm("v_pk_ashrrev_i16 %0, %2, %1" : "=v"(m) : "v"(x), "v"(0x000F000Fu));
        return x & ~m;
    }
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

// two at a time on the packed-f32 VALU forms (v_pk_mul_f32 / v_pk_fma_f32: two lanes' worth of polynomial per issue slot; rcp / exp stay
// scalar).  The GELU epilogue of a one-workgroup-per-CU kernel is pure exposed VALU time (measured: +10 us on a 35 us tile).
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2v gelu_erf2(f2v v) {
    f2v x;
    x.x = fabsf(v.x); x.y = fabsf(v.y);
    x = x * 0.70710678118654752440f;
    f2v t = x * 0.3275911f + 1.0f;
    t.x = __frcp_rn(t.x); t.y = __frcp_rn(t.y);
    f2v p = t * 1.061405429f + (-1.453152027f);
    p = p * t + 1.421413741f;
    p = p * t + (-0.284496736f);
    p = p * t + 0.254829592f;
    const f2v nx2 = x * x * (-1.4426950408889634f);     // -x^2 * log2(e): exp2 below
    f2v e2;
    e2.x = __builtin_amdgcn_exp2f(nx2.x); e2.y = __builtin_amdgcn_exp2f(nx2.y);
    f2v e = 1.0f - p * t * e2;                           // erf(|v| / sqrt 2)
    e.x = copysignf(e.x, v.x); e.y = copysignf(e.y, v.y);
    return v * 0.5f * (e + 1.0f);
}

// d/dv [v Phi(v)] = Phi(v) + v phi(v) on the same Abramowitz-Stegun erf and the SAME exponential (exp(-v^2 / 2) is both the erf's tail factor
// and the density): epilogue 5 of the split class -- dX of the MLP's second linear multiplied by GELU'(pre-activation) where it is produced.
__device__ __forceinline__ f2v dgelu_erf2(f2v v) {
    f2v x;
    x.x = fabsf(v.x); x.y = fabsf(v.y);
    x = x * 0.70710678118654752440f;
    f2v t = x * 0.3275911f + 1.0f;
    t.x = __frcp_rn(t.x); t.y = __frcp_rn(t.y);
    f2v p = t * 1.061405429f + (-1.453152027f);
    p = p * t + 1.421413741f;
    p = p * t + (-0.284496736f);
    p = p * t + 0.254829592f;
    const f2v nx2 = x * x * (-1.4426950408889634f);
    f2v e2;
    e2.x = __builtin_amdgcn_exp2f(nx2.x); e2.y = __builtin_amdgcn_exp2f(nx2.y);     // exp(-v^2 / 2)
    f2v e = 1.0f - p * t * e2;
    e.x = copysignf(e.x, v.x); e.y = copysignf(e.y, v.y);
    return (e + 1.0f) * 0.5f + v * (e2 * 0.3989422804014327f);
}

// GELU for the 16-bit epilogues without transcendentals: erf(x / sqrt 2) ~ t * P(t^2), t = clamp(x, -4.2, 4.2) / 4.2, P of degree 7
// (minimax fit of the GELU error with P(1) = 1 so that the clamp is seamless; tools/fit_gelu_poly.py).  |gelu_poly - gelu_erf| <= 8.1e-5
// absolute in f32 evaluation -- a sixth of an f16 ulp at 1 -- for 13 packed issue slots per two elements instead of ~30 (the rcp and
// exp2 of the erf form are quarter rate, and the GELU epilogue of a one-workgroup-per-CU kernel is exposed VALU time).  The f32 outputs
// (EPI_STORE32-class paths, linear_f32) keep gelu_erf.
__device__ __forceinline__ f2v gelu_poly2(f2v v) {
    constexpr float X = 4.2f;
    f2v t;
    t.x = __builtin_amdgcn_fmed3f(v.x, -X, X); t.y = __builtin_amdgcn_fmed3f(v.y, -X, X);
    t = t * (1.0f / X);
    const f2v u = t * t;
    f2v p = u * (-4.02584553f) + 20.0985432f;
    p = p * u + (-43.586319f);
    p = p * u + 54.4050636f;
    p = p * u + (-43.8327904f);
    p = p * u + 24.3039417f;
    p = p * u + (-9.70970726f);
    p = p * u + 3.34711337f;
    const f2v e = p * t, h = v * 0.5f;
    return h * e + h;
}
__device__ __forceinline__ float gelu_poly(float v) { return gelu_poly2(f2v{v, v}).x; }

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

// ---- epilogue shared by the tile kernels.  The main loops issue mfma(W fragment, A fragment): the accumulator tile is
// C^T, so a lane holds FOUR CONSECUTIVE OUTPUT COLUMNS of one row per fragment (row m = lane & 15, columns
// (lane >> 4) * 4 + 0..3) -- an 8-byte (16-bit) or 16-byte (f32) store per fragment straight from registers, no LDS
// transpose and no barrier; RoPE pairs (c, c+16) are fragments j / j+1 of the same lane, interleaved pairs (2p, 2p+1) are
// neighbouring registers.  mw0 / nbase = first output row / column of the wave's (16*MI) x 64 tile. ----
template <int BF16, int EPI, int MI>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs &g, f4 (&acc)[MI][4], int mw0, int nbase, void *lds_scratch, int, int lane) {
    // (opaque copy of the lane id: keeps the compiler from hoisting the epilogue's per-lane constants -- bias, RoPE frequencies --
    //  above the main loop, where they cost a spill that is reloaded inside it)
    asm volatile("" : "+v"(lane));
    if constexpr (BF16 == kDtSplit) {   // undo the power-of-two scale of the packed weights (exact)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] *= g.acc_scale;
    }
    const int mrow = lane & 15, c4 = (lane >> 4) * 4;
    const bool full_n = nbase + 64 <= g.N;
    float bv[4][4];
    if (g.bias && full_n && (reinterpret_cast<uintptr_t>(g.bias) & 15) == 0) {   // (wave-uniform) 4 x 16-byte loads
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 t = *reinterpret_cast<const float4 *>(g.bias + nbase + j * 16 + c4);
            bv[j][0] = t.x; bv[j][1] = t.y; bv[j][2] = t.z; bv[j][3] = t.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nbase + j * 16 + c4 + r;
                bv[j][r] = (g.bias && n < g.N) ? g.bias[n] : 0.0f;
            }
    }
    [[maybe_unused]] float inv2d[4] = {0.f, 0.f, 0.f, 0.f};
    [[maybe_unused]] bool rope_on = false;
    // 2-D RoPE angles take few distinct values (position 0..gh x 16 frequencies): the workgroup tabulates (sin, cos) for positions
    // 0..kRopeTabPos-1 once in the (now free) LDS ring -- 2 v_sin / v_cos per thread instead of 64 + 64 per lane in the interior-tile
    // path -- with the SAME expression the direct evaluation uses, so the results are bit-identical; larger positions fall back to it.
    constexpr int kRopeTabPos = 64;
    [[maybe_unused]] float2 *const rtab = reinterpret_cast<float2 *>(lds_scratch);
    if constexpr (EPI == 4) {
        rope_on = nbase < 2 * g.rope_C && full_n;
#pragma unroll
        for (int r = 0; r < 4; ++r) inv2d[r] = __builtin_amdgcn_exp2f(-(float)(c4 + r) * (1.0f / 16.0f) * g.rope_l2base);
        for (int e = threadIdx.x; e < kRopeTabPos * 16; e += blockDim.x) {
            float sn, cs;
            sincos_hw((float)(e >> 4) * __builtin_amdgcn_exp2f(-(float)(e & 15) * (1.0f / 16.0f) * g.rope_l2base), sn, cs);
            rtab[e] = make_float2(sn, cs);
        }
        __syncthreads();
    }
    constexpr bool OUT16 = (EPI == 0 || EPI == 1 || EPI == 4) && !is_f32io(BF16);   // f32 operands: every epilogue stores floats
    const bool vec_ok = full_n && (OUT16 ? (g.ldo % 4 == 0 && (reinterpret_cast<uintptr_t>(g.out) & 7) == 0)
                                         : (g.ldo % 4 == 0 && (reinterpret_cast<uintptr_t>(g.out) & 15) == 0));
    // ---- fast path: interior wave tile (every row and column valid, vector-aligned output).  All conditions are wave-uniform,
    // so this is straight-line code: the generic path below branches per element and ends up as one load -> wait -> store chain
    // per fragment (measured on the f32 residual epilogue: 17 us of a 52 us tile, 4 loads in flight per wave).  Here every
    // residual / gate load of a batch of 4 row fragments is issued before the first use (16 x 16 B in flight per lane). ----
    {
        const bool plain_resid = EPI != 2 || (g.ksplit <= 1 && g.ksplit >= 0);
        const bool res_ok = (EPI != 2 && EPI != 5) || !g.resid || (reinterpret_cast<uintptr_t>(g.resid) & 15) == 0;
        const bool gate_ok = EPI != 2 || !g.gate || (g.gate_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(g.gate) & 15) == 0);
        if (vec_ok && mw0 + 16 * MI <= g.M && plain_resid && res_ok && gate_ok) {
            size_t orow[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = mw0 + i * 16 + mrow;
                orow[i] = (size_t)(m / g.grp_in) * g.grp_out + g.grp_off + (m % g.grp_in);
            }
            [[maybe_unused]] int rk[MI];
            [[maybe_unused]] int2 rp[MI];
            if constexpr (EPI == 4) {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    rk[i] = (rope_on && g.rope_kind) ? (int)g.rope_kind[orow[i]] : 0;
                    rp[i] = rope_on ? *reinterpret_cast<const int2 *>(g.rope_pos + 2 * orow[i]) : make_int2(0, 0);
                }
            }
            if constexpr (EPI == 2) {
                float *const outp = reinterpret_cast<float *>(g.out);
                const float *const srcp = g.resid ? g.resid : outp;
                // The gate of a row depends on the row only through its group (frame) m / gate_rows.  With gate_rows >= 16 * MI the lane's
                // MI rows (16 apart) fall into at most two groups: their gate vectors are loaded ONCE (8 x 16 B) and selected per row.
                // (Per-row gate loads: every workgroup of the launch hammers the same few KB of the [groups, N] array -- measured +24 us
                // on a 160 us GEMM.)  Fewer rows per group than that: per-row loads.
                const int m_first = mw0 + mrow;
                const bool two_groups = g.gate && g.gate_rows >= 16 * MI;
                const int f0 = g.gate ? m_first / g.gate_rows : 0;
                [[maybe_unused]] float4 gA[4], gB[4];
                if (two_groups) {
                    const int f1 = (m_first + 16 * (MI - 1)) / g.gate_rows;
                    const float *pa_ = g.gate + (size_t)f0 * g.gate_ld + nbase + c4, *pb_ = g.gate + (size_t)f1 * g.gate_ld + nbase + c4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { gA[j] = *reinterpret_cast<const float4 *>(pa_ + j * 16); gB[j] = *reinterpret_cast<const float4 *>(pb_ + j * 16); }
                }
                // 2 row fragments per batch: 8 residual loads (32 registers) in flight beside the 128 accumulator and the 32 cached gate
                // registers (4 per batch spills)
                constexpr int IB = MI >= 2 ? 2 : 1;
#pragma unroll
                for (int i0 = 0; i0 < MI; i0 += IB) {
                    float4 rs[IB][4];
#pragma unroll
                    for (int ii = 0; ii < IB; ++ii)
#pragma unroll
                        for (int j = 0; j < 4; ++j) rs[ii][j] = *reinterpret_cast<const float4 *>(srcp + orow[i0 + ii] * g.ldo + nbase + c4 + j * 16);
#pragma unroll
                    for (int ii = 0; ii < IB; ++ii) {
                        const int m = m_first + (i0 + ii) * 16;
                        const bool first = g.gate && (m / g.gate_rows == f0);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float4 val = make_float4(acc[i0 + ii][j][0] + bv[j][0], acc[i0 + ii][j][1] + bv[j][1], acc[i0 + ii][j][2] + bv[j][2],
                                                     acc[i0 + ii][j][3] + bv[j][3]);
                            if (g.gate) {
                                float4 gt;
                                if (two_groups) {   // bitwise select (a ?: on the arrays makes hipcc index them through scratch memory)
                                    const unsigned mk = first ? 0xffffffffu : 0u;
                                    auto sel = [mk](float a_, float b_) { return __uint_as_float((__float_as_uint(a_) & mk) | (__float_as_uint(b_) & ~mk)); };
                                    gt = make_float4(sel(gA[j].x, gB[j].x), sel(gA[j].y, gB[j].y), sel(gA[j].z, gB[j].z), sel(gA[j].w, gB[j].w));
                                } else {
                                    gt = *reinterpret_cast<const float4 *>(g.gate + (size_t)(m / g.gate_rows) * g.gate_ld + nbase + c4 + j * 16);
                                }
                                val.x *= 1.0f + gt.x; val.y *= 1.0f + gt.y; val.z *= 1.0f + gt.z; val.w *= 1.0f + gt.w;
                            }
                            float4 o = rs[ii][j];
                            o.x += val.x; o.y += val.y; o.z += val.z; o.w += val.w;
                            *reinterpret_cast<float4 *>(outp + orow[i0 + ii] * g.ldo + nbase + c4 + j * 16) = o;
                        }
                    }
                }
                return;
            } else {
                [[maybe_unused]] uint2 pkprev[4];
                [[maybe_unused]] const bool wide16 = OUT16 && MI % 2 == 0 && g.ldo % 8 == 0 && (reinterpret_cast<uintptr_t>(g.out) & 15) == 0;
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    float v[4][4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 4; r += 2) {
                            if constexpr (EPI == 1) {
                                const f2v zin = f2v{acc[i][j][r] + bv[j][r], acc[i][j][r + 1] + bv[j][r + 1]};
                                const f2v gl = is_f32io(BF16) ? gelu_erf2(zin) : gelu_poly2(zin);   // f32 operands = reference-precision path: exact erf
                                v[j][r] = gl.x; v[j][r + 1] = gl.y;
                            } else {
                                v[j][r] = acc[i][j][r] + bv[j][r];
                                v[j][r + 1] = acc[i][j][r + 1] + bv[j][r + 1];
                            }
                        }
                    if constexpr (EPI == 5) {     // x GELU'(z), z = g.resid (f32, the layout of out): four 16-byte loads in flight per row fragment
                        const float *zp = g.resid + orow[i] * g.ldo + nbase + c4;
                        float4 z[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) z[j] = *reinterpret_cast<const float4 *>(zp + j * 16);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const f2v d0 = dgelu_erf2(f2v{z[j].x, z[j].y}), d1 = dgelu_erf2(f2v{z[j].z, z[j].w});
                            v[j][0] *= d0.x; v[j][1] *= d0.y; v[j][2] *= d1.x; v[j][3] *= d1.y;
                        }
                    }
                    if constexpr (EPI == 4) {
                        if (rope_on) {
                            if (rk[i] == 0) {
#pragma unroll
                                for (int h = 0; h < 2; ++h) {
                                    const int pi = h == 0 ? rp[i].x : rp[i].y;
                                    const float p = (float)pi;
                                    float4 sc01, sc23;   // (sin, cos) of frequencies c4, c4+1 | c4+2, c4+3
                                    const bool tab = (unsigned)pi < (unsigned)kRopeTabPos;
                                    if (tab) {
                                        sc01 = *reinterpret_cast<const float4 *>(rtab + pi * 16 + c4);
                                        sc23 = *reinterpret_cast<const float4 *>(rtab + pi * 16 + c4 + 2);
                                    }
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        float sn, cs;
                                        if (tab) {
                                            sn = r == 0 ? sc01.x : r == 1 ? sc01.z : r == 2 ? sc23.x : sc23.z;
                                            cs = r == 0 ? sc01.y : r == 1 ? sc01.w : r == 2 ? sc23.y : sc23.w;
                                        } else {
                                            sincos_hw(p * inv2d[r], sn, cs);
                                        }
                                        const float u = v[2 * h][r], w = v[2 * h + 1][r];
                                        v[2 * h][r] = u * cs - w * sn;
                                        v[2 * h + 1][r] = w * cs + u * sn;
                                    }
                                }
                            } else if (rk[i] == 1) {
                                const float p = (float)rp[i].x;
#pragma unroll
                                for (int j = 0; j < 4; ++j)
#pragma unroll
                                    for (int r = 0; r < 4; r += 2) {
                                        float sn, cs;
                                        sincos_hw(p * __builtin_amdgcn_exp2f(-(float)((j * 16 + c4 + r) >> 1) * (1.0f / 32.0f) * g.rope_l2theta), sn, cs);
                                        const float u = v[j][r], w = v[j][r + 1];
                                        v[j][r] = u * cs - w * sn;
                                        v[j][r + 1] = w * cs + u * sn;
                                    }
                            }
                        }
                    }
                    if constexpr (OUT16) {
                        constexpr int D16 = is_f32io(BF16) ? 0 : BF16;
                        uint2 pk[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            pk[j].x = pack16x2<D16>(v[j][0], v[j][1]);
                            pk[j].y = pack16x2<D16>(v[j][2], v[j][3]);
                        }
                        if (wide16) {
                            // 16-byte stores: the epilogue is store-ISSUE-bound (4.4 us of a 24 us K = 1024 tile with 8-byte stores, measured
                            // with cycle stamps), and a lane's 4 columns are 8 bytes.  Row fragments are taken in pairs: v_permlane16_swap
                            // exchanges the odd 16-lane rows of fragment i with the even rows of fragment i + 1, after which an even-row lane
                            // holds 8 consecutive columns of fragment i and an odd-row lane 8 consecutive columns of fragment i + 1.
                            if ((i & 1) == 0) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) pkprev[j] = pk[j];
                            } else {
                                const bool odd = (lane >> 4) & 1;
                                unsigned short *dst = reinterpret_cast<unsigned short *>(g.out) + (odd ? orow[i] : orow[i - 1]) * g.ldo + nbase + (c4 & ~4);
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    const u2v sx = __builtin_amdgcn_permlane16_swap(pkprev[j].x, pk[j].x, false, false);
                                    const u2v sy = __builtin_amdgcn_permlane16_swap(pkprev[j].y, pk[j].y, false, false);
                                    *reinterpret_cast<uint4 *>(dst + j * 16) = make_uint4(sx.x, sy.x, sx.y, sy.y);
                                }
                            }
                        } else {
                            unsigned short *dst = reinterpret_cast<unsigned short *>(g.out) + orow[i] * g.ldo + nbase + c4;
#pragma unroll
                            for (int j = 0; j < 4; ++j) *reinterpret_cast<uint2 *>(dst + j * 16) = pk[j];
                        }
                    } else {
                        if (BF16 == kDtSplit && g.out_packed) {
                            float *rowp = reinterpret_cast<float *>(g.out) + orow[i] * g.ldo;   // (nbase % 64 == 0: fragments j, j + 1 share a chunk)
#pragma unroll
                            for (int j = 0; j < 4; j += 2) store_split8(rowp, nbase + c4 + j * 16, v[j], v[j + 1]);
                        } else {
                            float *dst = reinterpret_cast<float *>(g.out) + orow[i] * g.ldo + nbase + c4;
#pragma unroll
                            for (int j = 0; j < 4; ++j) *reinterpret_cast<float4 *>(dst + j * 16) = make_float4(v[j][0], v[j][1], v[j][2], v[j][3]);
                        }
                    }
                }
                return;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = mw0 + i * 16 + mrow;
        const bool valid = m < g.M;
        const int mc = valid ? m : g.M - 1;
        const size_t orow = (size_t)(mc / g.grp_in) * g.grp_out + g.grp_off + (mc % g.grp_in);
        float v[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[j][r] = acc[i][j][r] + bv[j][r];
                if constexpr (EPI == 1) v[j][r] = is_f32io(BF16) ? gelu_erf(v[j][r]) : gelu_poly(v[j][r]);   // same function as the wide path: results do not depend on the tile path
            }
        if constexpr (EPI == 5) {
            if (valid) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        const int n = nbase + j * 16 + c4 + r;
                        const float z0 = n < g.N ? g.resid[orow * g.ldo + n] : 0.f, z1 = n + 1 < g.N ? g.resid[orow * g.ldo + n + 1] : 0.f;
                        const f2v dd = dgelu_erf2(f2v{z0, z1});      // (the pair form of the wide path: results do not depend on the tile path)
                        v[j][r] *= dd.x; v[j][r + 1] *= dd.y;
                    }
            }
        }
        if constexpr (EPI == 4) {
            if (rope_on) {
                const int kd = g.rope_kind ? (int)g.rope_kind[orow] : 0;
                const int2 pp = *reinterpret_cast<const int2 *>(g.rope_pos + 2 * orow);
                if (kd == 0) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float p = (float)(h == 0 ? pp.x : pp.y);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float sn, cs;
                            sincos_hw(p * inv2d[r], sn, cs);
                            const float u = v[2 * h][r], w = v[2 * h + 1][r];
                            v[2 * h][r] = u * cs - w * sn;
                            v[2 * h + 1][r] = w * cs + u * sn;
                        }
                    }
                } else if (kd == 1) {  // temporal rope of a camera-token row: interleaved pairs (2p, 2p+1) = registers (r, r+1)
                    const float p = (float)pp.x;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 4; r += 2) {
                            float sn, cs;
                            sincos_hw(p * __builtin_amdgcn_exp2f(-(float)((j * 16 + c4 + r) >> 1) * (1.0f / 32.0f) * g.rope_l2theta), sn, cs);
                            const float u = v[j][r], w = v[j][r + 1];
                            v[j][r] = u * cs - w * sn;
                            v[j][r + 1] = w * cs + u * sn;
                        }
                }
            }
        }
        if (!valid) continue;
        if constexpr (OUT16) {
            constexpr int D16 = is_f32io(BF16) ? 0 : BF16;   // (never instantiated for f32: keeps to16<> well-formed)
            unsigned short *dst = reinterpret_cast<unsigned short *>(g.out) + orow * g.ldo + nbase + c4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (vec_ok) {
                    uint2 pk;
                    pk.x = pack16x2<D16>(v[j][0], v[j][1]);
                    pk.y = pack16x2<D16>(v[j][2], v[j][3]);
                    *reinterpret_cast<uint2 *>(dst + j * 16) = pk;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (nbase + j * 16 + c4 + r < g.N) dst[j * 16 + r] = to16<D16>(v[j][r]);
                }
            }
        } else {
            float *dst = reinterpret_cast<float *>(g.out) + orow * g.ldo + nbase + c4;
            const float *gp = nullptr;
            if (EPI == 2 && g.gate) gp = g.gate + (size_t)(m / g.gate_rows) * g.gate_ld + nbase + c4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (vec_ok) {
                    float4 val = make_float4(v[j][0], v[j][1], v[j][2], v[j][3]);
                    if constexpr (EPI == 2) {
                        if (gp) {
                            const bool g_ok = (reinterpret_cast<uintptr_t>(gp + j * 16) & 15) == 0;
                            float gt[4];
                            if (g_ok) { const float4 t = *reinterpret_cast<const float4 *>(gp + j * 16); gt[0] = t.x; gt[1] = t.y; gt[2] = t.z; gt[3] = t.w; }
                            else { gt[0] = gp[j * 16]; gt[1] = gp[j * 16 + 1]; gt[2] = gp[j * 16 + 2]; gt[3] = gp[j * 16 + 3]; }
                            val.x *= 1.0f + gt[0]; val.y *= 1.0f + gt[1]; val.z *= 1.0f + gt[2]; val.w *= 1.0f + gt[3];
                        }
                        if (g.ksplit < 0) {  // partial tile of a weight-gradient K slice: plain store into the workspace
                            *reinterpret_cast<float4 *>(dst + j * 16) = val;
                        } else if (g.ksplit > 1) {  // partial sums of a split-K tail launch meet in memory
                            unsafeAtomicAdd(dst + j * 16 + 0, val.x); unsafeAtomicAdd(dst + j * 16 + 1, val.y);
                            unsafeAtomicAdd(dst + j * 16 + 2, val.z); unsafeAtomicAdd(dst + j * 16 + 3, val.w);
                        } else {
                            const float *src = g.resid ? g.resid + (dst - reinterpret_cast<float *>(g.out)) : dst;
                            float4 o = *reinterpret_cast<const float4 *>(src + j * 16);
                            o.x += val.x; o.y += val.y; o.z += val.z; o.w += val.w;
                            *reinterpret_cast<float4 *>(dst + j * 16) = o;
                        }
                    } else {
                        if (BF16 == kDtSplit && g.out_packed) store_split4(reinterpret_cast<float *>(g.out) + orow * g.ldo, nbase + c4 + j * 16, val.x, val.y, val.z, val.w);
                        else *reinterpret_cast<float4 *>(dst + j * 16) = val;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = nbase + j * 16 + c4 + r;
                        if (n < g.N) {
                            float x = v[j][r];
                            if (EPI == 2 && gp) x *= 1.0f + gp[j * 16 + r];
                            if constexpr (EPI == 2) {
                                if (g.ksplit < 0) dst[j * 16 + r] = x;
                                else if (g.ksplit > 1) unsafeAtomicAdd(dst + j * 16 + r, x);
                                else dst[j * 16 + r] = (g.resid ? g.resid[(dst - reinterpret_cast<float *>(g.out)) + j * 16 + r] : dst[j * 16 + r]) + x;
                            } else {
                                dst[j * 16 + r] = x;
                            }
                        }
                    }
                }
            }
        }
    }
}

}  // namespace
