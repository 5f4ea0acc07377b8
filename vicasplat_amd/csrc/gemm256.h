// 256x256x64 GEMM tile kernel for the big ViT-L GEMMs: 8 waves (2 in M x 4 in N, 128x64 outputs each), operands staged
// HBM/L2 -> LDS with global_load_lds_dwordx4 into a two-K-tile ring, and a phase-interleaved K loop in which the two
// wave groups (waves 0-3 / 4-7: one wave of each per SIMD) run half a phase apart, so that while one group feeds the
// MFMA pipe the other issues its ds_reads and the next LDS-DMA.  Same fused epilogues as gemm_kernel.
//
// Staging units.  A K-tile (64 wide) is staged as four 16 KiB units of 128 rows x 128 B, cut by REGISTER sub-tile
// rather than by tile half: A_h = rows {wr*128 + h*64 + 0..63 : wr=0,1}, B_h = W rows {wc*64 + h*32 + 0..31 : wc=0..3}
// (h = 0,1), so unit A_h / B_h is exactly what every wave reads for its h-th A / B register sub-tile.  One unit is one
// LDS-DMA round: 512 lanes x 16 B x 2.  Rows are 128 B; the 16-byte chunk index is XORed with (row>>1)&7 on the global
// source side and again on the read, which makes every 16-lane service group of ds_read_b128 hit 16 distinct slots.
//
// Phases.  Per K-tile a wave does 4 phases of 16 MFMAs (one 64x32 output quadrant x K=64):
//     ph1: read B_0, A_0 -> (A0,B0)   ph2: read B_1 -> (A0,B1)   ph3: read A_1 -> (A1,B1)   ph4: (A1,B0)
// and every phase re-fills ONE unit, as soon as it is legal (>= 2 phases after its last ds_read, because the other
// wave group trails by one barrier) and as early as possible:
//     ph1: B_1 of tile kt+1   ph2: A_1 of kt+1   ph3: A_0 of kt+2   ph4: B_0 of kt+2        (ring slot = tile & 1)
// so 4 units (64 KiB) are always in flight.  A unit is waited for (counted vmcnt, never 0 in the steady state) in the
// phase BEFORE the one that reads it, ahead of that phase's first barrier: LDS-DMA data is ordered for another wave's
// ds_read only by the issuer's vmcnt wait followed by a barrier both have passed.
#pragma once
#include "gemm_common.h"

namespace {

constexpr unsigned kUnitBytes256 = 128 * 128;           // one staged unit: 128 rows x 128 B
constexpr unsigned kLdsBytes256 = 2 * 4 * kUnitBytes256;  // [ring slot][A0 A1 B0 B1] = 128 KiB

// Staged-unit row q (0..127) of wave `wid`, LDS-DMA round j, lane: which tile row it holds and which 16-byte chunk of
// the source row this lane must fetch so that the data lands XOR-swizzled.
__device__ __forceinline__ int unit_row256(int wid, int j, int lane) { return wid * 16 + j * 8 + (lane >> 3); }
__device__ __forceinline__ int unit_src_chunk256(int q, int lane) { return (lane & 7) ^ ((q >> 1) & 7); }
__device__ __forceinline__ int unit_a_tile_row256(int q, int h) { return (q >> 6) * 128 + h * 64 + (q & 63); }
__device__ __forceinline__ int unit_b_tile_row256(int q, int h) { return (q >> 5) * 64 + h * 32 + (q & 31); }

// The K loop.  `st.stage(u, kt, lds_byte_offset)` issues this wave's two global_load_lds_dwordx4 for unit u (0 A_0, 1 A_1,
// 2 B_0, 3 B_1) of K-tile kt; RELU_A clamps the A fragments at zero on their way to the MFMA (conv: activation-before-conv).
//
// TN = true: the operands are REDUCTION-MAJOR in memory (element (k, m) at k * ld + m: a weight gradient dW = dY^T X read
// straight from dY [tokens, N] and X [tokens, K], no transposed copies).  A staged unit is then 64 reduction rows x 256 B (the
// unit's 128 tile rows as two / four contiguous column segments per reduction row), and a fragment is gathered with the LDS
// transpose read ds_read_b64_tr_b16: lane t of a 16-lane group supplies the address of B[t>>2][(t&3)*4 .. +3] of a
// [4 reduction rows][16 tile rows] block and receives column t, i.e. 4 consecutive k of its tile row (probed on hardware:
// tools/probe/tr_read.hip).  Two such reads are the 8 k-values of one MFMA operand.  The 16-byte chunk index is XORed with
// (reduction row & 3) << 1, so that the four rows of a block land in four different 8-bank windows.
template <int BF16, bool RELU_A, class Stager, bool TN = false>
__device__ __forceinline__ void mainloop256(Stager &st, const int KT, f4 (&acc)[8][4], const unsigned char *smem, const int lane,
                                            const int wid) {
    static_assert(BF16 != kDtSplit, "split operands run mainloop256_split");
    constexpr unsigned UNITB = kUnitBytes256;
    const int wr = wid >> 2, wc = wid & 3;
    typedef void __attribute__((address_space(3))) *lptr_t;
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    const unsigned lds_w = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wid * 2048u);
#define VS_STAGE(u_, kt_, d_) st.stage(u_, kt_, lds_w + (unsigned)(((d_) * 4 + (u_)) * UNITB));

    const int frow = lane & 15, fg = lane >> 4;
    const unsigned rd0 = (unsigned)(frow * 128 + (((0 + fg) ^ (frow >> 1)) << 4));
    const unsigned rd1 = (unsigned)(frow * 128 + (((4 + fg) ^ (frow >> 1)) << 4));
    const unsigned char *rdA = smem + wr * (64 * 128);
    const unsigned char *rdB = smem + wc * (32 * 128);
    // TN: per-lane byte offsets of the transpose reads inside a unit (see above); t = lane in its 16-lane group, kg = group
    [[maybe_unused]] unsigned tnA[4], tnB[2];
    if constexpr (TN) {
        const int t = lane & 15, kg = lane >> 4, tq = t >> 2, c1 = (t & 3) >> 1;
        const unsigned rowb = (unsigned)((kg * 8 + tq) * 256 + (t & 1) * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) tnA[i] = rowb + (unsigned)(((wr * 8 + i * 2 + c1) ^ (tq << 1)) << 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) tnB[j] = rowb + (unsigned)(((wc * 4 + j * 2 + c1) ^ (tq << 1)) << 4);
    }
    typedef short tr4 __attribute__((ext_vector_type(4)));
    typedef tr4 __attribute__((address_space(3))) *trp_t;
    auto tr8 = [&](const unsigned char *p) -> uint4 {   // 8 k-values of one tile row: reduction rows +0..3 and +4..7
        const tr4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(const_cast<unsigned char *>(p)));
        const tr4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(const_cast<unsigned char *>(p + 4 * 256)));
        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        return make_uint4(l2.x, l2.y, h2.x, h2.y);
    };
    uint4 fa[4][2], fb[2][2][2];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

#define VS_RD_A(h_, d_)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                              \
        if constexpr (TN) {                                                                                      \
            fa[i][0] = tr8(smem + ((d_) * 4 + (h_)) * UNITB + tnA[i]);                                           \
            fa[i][1] = tr8(smem + ((d_) * 4 + (h_)) * UNITB + tnA[i] + 32 * 256);                                \
        } else {                                                                                                 \
            fa[i][0] = *reinterpret_cast<const uint4 *>(rdA + ((d_) * 4 + (h_)) * UNITB + i * 2048 + rd0);       \
            fa[i][1] = *reinterpret_cast<const uint4 *>(rdA + ((d_) * 4 + (h_)) * UNITB + i * 2048 + rd1);       \
        }                                                                                                        \
    }
#define VS_RD_B(h_, d_)                                                                                          \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                              \
        if constexpr (TN) {                                                                                      \
            fb[h_][j][0] = tr8(smem + ((d_) * 4 + 2 + (h_)) * UNITB + tnB[j]);                                   \
            fb[h_][j][1] = tr8(smem + ((d_) * 4 + 2 + (h_)) * UNITB + tnB[j] + 32 * 256);                        \
        } else {                                                                                                 \
            fb[h_][j][0] = *reinterpret_cast<const uint4 *>(rdB + ((d_) * 4 + 2 + (h_)) * UNITB + j * 2048 + rd0); \
            fb[h_][j][1] = *reinterpret_cast<const uint4 *>(rdB + ((d_) * 4 + 2 + (h_)) * UNITB + j * 2048 + rd1); \
        }                                                                                                        \
    }
#define VS_RELU_A()                                                                                              \
    if constexpr (RELU_A) {                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                            \
            _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                                   \
                fa[i][s_].x = relu_reg<BF16>(fa[i][s_].x); fa[i][s_].y = relu_reg<BF16>(fa[i][s_].y);            \
                fa[i][s_].z = relu_reg<BF16>(fa[i][s_].z); fa[i][s_].w = relu_reg<BF16>(fa[i][s_].w);            \
            }                                                                                                    \
    }
#define VS_MM(ha_, hb_)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                            \
            acc[(ha_) * 4 + i][(hb_) * 2 + j] = mma2<BF16>(fb[hb_][j][0], fb[hb_][j][1], fa[i][0], fa[i][1], acc[(ha_) * 4 + i][(hb_) * 2 + j]);
#define VS_BAR()                                  \
    {                                             \
        __builtin_amdgcn_sched_barrier(0);        \
        __builtin_amdgcn_s_barrier();             \
        __builtin_amdgcn_sched_barrier(0);        \
    }
#define VS_COMPUTE(ha_, hb_, relu_)               \
    {                                             \
        VS_BAR()                                  \
        __builtin_amdgcn_s_setprio(1);            \
        if (relu_) VS_RELU_A()                    \
        VS_MM(ha_, hb_)                           \
        __builtin_amdgcn_s_setprio(0);            \
        VS_BAR()                                  \
    }
#define VS_WAIT(n_) asm volatile("s_waitcnt vmcnt(" #n_ ")" ::: "memory");
    // MODE 0: steady state (tiles kt+1 and kt+2 exist), 1: kt == KT-2, 2: kt == KT-1
#define VS_KTILE(kt_, d_, MODE_)                                                  \
    {                                                                             \
        VS_RD_B(0, d_) VS_RD_A(0, d_)                                             \
        if (MODE_ <= 1) { VS_STAGE(3, (kt_) + 1, (d_) ^ 1) VS_WAIT(8) } else { VS_WAIT(2) } \
        VS_COMPUTE(0, 0, true)                                                    \
        VS_RD_B(1, d_)                                                            \
        if (MODE_ <= 1) { VS_STAGE(1, (kt_) + 1, (d_) ^ 1) VS_WAIT(8) } else { VS_WAIT(0) } \
        VS_COMPUTE(0, 1, false)                                                   \
        VS_RD_A(1, d_)                                                            \
        if (MODE_ == 0) VS_STAGE(0, (kt_) + 2, d_)                                \
        VS_COMPUTE(1, 1, true)                                                    \
        if (MODE_ == 0) { VS_STAGE(2, (kt_) + 2, d_) VS_WAIT(8) } else if (MODE_ == 1) { VS_WAIT(4) } \
        VS_COMPUTE(1, 0, false)                                                   \
    }

    // KT is even and >= 2 (checked by the launchers)
    VS_STAGE(0, 0, 0) VS_STAGE(2, 0, 0) VS_STAGE(3, 0, 0) VS_STAGE(1, 0, 0) VS_STAGE(0, 1, 1) VS_STAGE(2, 1, 1)
    VS_WAIT(8)
    VS_BAR()
    if (wr == 1) VS_BAR()  // the second wave group trails the first by one barrier from here on
    for (int kt = 0; kt + 2 < KT; kt += 2) {
        VS_KTILE(kt, 0, 0)
        VS_KTILE(kt + 1, 1, 0)
    }
    VS_KTILE(KT - 2, 0, 1)
    VS_KTILE(KT - 1, 1, 2)
    if (wr == 0) VS_BAR()
#undef VS_KTILE
#undef VS_WAIT
#undef VS_COMPUTE
#undef VS_BAR
#undef VS_MM
#undef VS_RELU_A
#undef VS_RD_B
#undef VS_RD_A
#undef VS_STAGE
    __syncthreads();  // every wave is done with the ring before an epilogue reuses it
}

// ---- split operands (gemm_common.h, kDtSplit) on the same tile, ring, units and phases.  The A units arrive as f32 rows (LDS-DMA
// cannot convert) and are converted IN PLACE in LDS, once per workgroup, to the (hi, lo) f16 form the MFMAs take: a thread reads the two
// 16-byte chunks (c, c + 4) of a row -- 8 floats, exactly the pair a lane group reads as a fragment -- and writes hi back over chunk c
// and lo over chunk c + 4 (16 VALU; the per-wave conversion of every fragment read it replaces cost 4x that, in the waves that feed the
// matrix pipe: measured 0.37 -> of the 16-bit peak on the ViT-L GEMMs).  Rows 64 g .. 64 g + 63 of an A unit are staged, converted
// and read by wave group g alone, one item per thread and unit, in the two light read segments of a K-tile:
//     R0: read B_0, A_0 | M0 (A0,B0) | R1: read B_1; stage A_0, B_0 of kt+2; CONVERT A_1 of kt | M1 (A0,B1)
//     R2: read A_1; stage B_1 of kt+2 | M2 (A1,B1) | R3: stage A_1 of kt+2; CONVERT A_0 of kt+1 | M3 (A1,B0)
// Issue order per wave and tile: A_0, B_0, B_1, A_1.  A unit must have LANDED one barrier before its conversion: W2 (B_1, A_1 of kt:
// four younger units in flight, vmcnt 8) closes the barrier interval of group 0's M0 / group 1's R0, W1 (A_0, B_0 of kt+1: five
// younger units, vmcnt 10) that of group 0's M2 / group 1's R2 -- the groups run one barrier apart, so the same interval is a compute
// segment for one and a read segment for the other.  The ReLU of an implicit-GEMM convolution is applied during the conversion.
// A_PACKED: the A units arrive ALREADY converted (the packed (hi, lo) image of vs_split_pack_weight, whose 128-byte block of 32 k is exactly
// the post-conversion LDS row: hi chunks 0..3 | lo chunks 0..3): no conversion segment, the loop is VALU-free like the 16-bit one.
// A_TN (round 4, the weight gradient dW = dY^T X without a transposed copy of dY): the A operand is REDUCTION-MAJOR f32 in memory (element
// (k, m) at k * lda + m).  An A unit is then [wave group g][32 k][64 tile rows x 4 B]: wave w of group g stages k rows 8 (w & 3) .. + 7 of
// its group's half (still its own 2 KiB of the unit) and converts exactly those: a lane takes 8 floats = 8 consecutive tile rows of one k
// (16-byte slots 4i + 2c, 4i + 2c + 1 of the 16-slot row; i = 16-row fragment, c = its half) and writes hi to slot 4i + c, lo to slot
// 4i + 2 + c -- the two halves of a fragment swap slots inside ONE wave, whose loads all return before its stores issue.  A fragment is
// gathered by the LDS transpose read (mainloop256, TN): lane t of a 16-lane group supplies the 8 bytes of tile rows 4 (t & 3) .. + 3 at
// k = kb + (t >> 2) and receives tile row t at k = kb .. kb + 3; kb = 4 kg and 16 + 4 kg, the k order of chunk kg of the packed W operand
// (gemm_common.h).  Slots are XORed with (k & 3) << 1 on the global-source side and on every LDS access (four k rows of a transpose read in
// four different 8-bank windows; the XOR is even, so slot pairs stay pairs).  Conversions, waits and barriers are those of the NT loop.
template <bool RELU_A, class Stager, bool A_PACKED = false, bool A_TN = false>
__device__ __forceinline__ void mainloop256_split(Stager &st, const int KT, f4 (&acc)[8][4], unsigned char *smem, const int lane, const int wid) {
    static_assert(!(RELU_A && A_PACKED), "a packed A operand carries its ReLU already");
    static_assert(!(A_TN && A_PACKED), "the reduction-major A operand is plain f32");
    constexpr unsigned UNITB = kUnitBytes256;
    constexpr int BF16 = kDtSplit;
    const int wr = wid >> 2, wc = wid & 3;
    typedef void __attribute__((address_space(3))) *lptr_t;
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    const unsigned lds_w = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wid * 2048u);
#define VS_STAGE(u_, kt_, d_) st.stage(u_, kt_, lds_w + (unsigned)(((d_) * 4 + (u_)) * UNITB));
    const int frow = lane & 15, fg = lane >> 4;
    const unsigned rd0 = (unsigned)(frow * 128 + (((0 + fg) ^ (frow >> 1)) << 4));
    const unsigned rd1 = (unsigned)(frow * 128 + (((4 + fg) ^ (frow >> 1)) << 4));
    const unsigned char *rdA = smem + wr * (64 * 128);
    const unsigned char *rdB = smem + wc * (32 * 128);
    // this thread's conversion item inside an A unit: unit row wr*64 + (wid&3)*16 + frow, chunk pair (fg, fg + 4) -- the fragment read pattern
    unsigned char *cvp = smem + (wr * 64 + (wid & 3) * 16) * 128;
    // ... with the 16 rows of the wave taken in the order even rows, then odd rows: the in-place ds_write_b128 is serviced in groups of 8
    // CONTIGUOUS lanes under a 32-bank modulus, where rows 2k and 2k + 1 share their chunk slot (PMC, round 3: 18 % of the LDS cycles of
    // the kernel were bank conflicts with the fragment order); the ds_read_b128 groups stay conflict-free under this order too
    const int crow = 2 * (frow & 7) + (frow >> 3);
    unsigned cv0 = (unsigned)(crow * 128 + (((0 + fg) ^ (crow >> 1)) << 4));     // conversion loads ...
    unsigned cv1 = (unsigned)(crow * 128 + (((4 + fg) ^ (crow >> 1)) << 4));
    unsigned cw0 = cv0, cw1 = cv1;                                              // ... and stores (in place)
    [[maybe_unused]] unsigned tnA[4];
    if constexpr (A_TN) {
        // conversion item of this lane inside the wave's own 2 KiB (= wid * 2048 = cvp): k row lane >> 3 of the wave's 8, fragment half lane & 7
        const int rr = lane >> 3, pi = lane & 7, ci = pi >> 1, cc = pi & 1, sw = (rr & 3) << 1;
        cv0 = (unsigned)(rr * 256 + (((4 * ci + 2 * cc) ^ sw) << 4));
        cv1 = cv0 + 16u;
        cw0 = (unsigned)(rr * 256 + (((4 * ci + cc) ^ sw) << 4));
        cw1 = cw0 ^ 32u;
        // transpose reads: t = lane in its 16-lane group, kg = group
        const int t = lane & 15, kg = lane >> 4, tq = t >> 2, c1 = (t & 3) >> 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) tnA[i] = (unsigned)(wr * 8192 + (4 * kg + tq) * 256 + (((4 * i + c1) ^ (tq << 1)) << 4) + (t & 1) * 8);
    }
    typedef short tr4 __attribute__((ext_vector_type(4)));
    typedef tr4 __attribute__((address_space(3))) *trp_t;
    [[maybe_unused]] auto tr8 = [&](const unsigned char *p) -> uint4 {   // k = kb .. kb + 3 and kb + 16 .. kb + 19 of one tile row
        const tr4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(const_cast<unsigned char *>(p)));
        const tr4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(const_cast<unsigned char *>(p + 16 * 256)));
        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        return make_uint4(l2.x, l2.y, h2.x, h2.y);
    };
    uint4 fa[4][2], fb[2][2][2];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

#define VS_RD_A(h_, d_)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                              \
        if constexpr (A_TN) {                                                                                    \
            fa[i][0] = tr8(smem + ((d_) * 4 + (h_)) * UNITB + tnA[i]);                                           \
            fa[i][1] = tr8(smem + ((d_) * 4 + (h_)) * UNITB + (tnA[i] ^ 32u));                                   \
        } else {                                                                                                 \
            fa[i][0] = *reinterpret_cast<const uint4 *>(rdA + ((d_) * 4 + (h_)) * UNITB + i * 2048 + rd0);       \
            fa[i][1] = *reinterpret_cast<const uint4 *>(rdA + ((d_) * 4 + (h_)) * UNITB + i * 2048 + rd1);       \
        }                                                                                                        \
    }
#define VS_RD_B(h_, d_)                                                                                          \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                              \
        fb[h_][j][0] = *reinterpret_cast<const uint4 *>(rdB + ((d_) * 4 + 2 + (h_)) * UNITB + j * 2048 + rd0);   \
        fb[h_][j][1] = *reinterpret_cast<const uint4 *>(rdB + ((d_) * 4 + 2 + (h_)) * UNITB + j * 2048 + rd1);   \
    }
    // the conversion of one item is cut in two: its LDS loads are issued at the start of a read segment, the VALU work and the stores at
    // its end, and the wave only waits for the stores (lgkmcnt) at the END of the following compute segment -- the converted rows are
    // read two barriers later, and a wait inside the read segment made it longer than the 24 MFMAs it runs beside
    [[maybe_unused]] uint4 cx0, cx1;
#define VS_CVT_LD(h_, d_)                                                                                        \
    if constexpr (!A_PACKED) {                                                                                   \
        cx0 = *reinterpret_cast<const uint4 *>(cvp + ((d_) * 4 + (h_)) * UNITB + cv0);                           \
        cx1 = *reinterpret_cast<const uint4 *>(cvp + ((d_) * 4 + (h_)) * UNITB + cv1);                           \
    }
#define VS_CVT_ST(h_, d_)                                                                                        \
    if constexpr (!A_PACKED) {                                                                                   \
        if constexpr (RELU_A) {                                                                                  \
            cx0.x = relu_f32_lds(cx0.x); cx0.y = relu_f32_lds(cx0.y); cx0.z = relu_f32_lds(cx0.z); cx0.w = relu_f32_lds(cx0.w); \
            cx1.x = relu_f32_lds(cx1.x); cx1.y = relu_f32_lds(cx1.y); cx1.z = relu_f32_lds(cx1.z); cx1.w = relu_f32_lds(cx1.w); \
        }                                                                                                        \
        split8_lds(cx0, cx1);                                                                                    \
        *reinterpret_cast<uint4 *>(cvp + ((d_) * 4 + (h_)) * UNITB + cw0) = cx0;                                 \
        *reinterpret_cast<uint4 *>(cvp + ((d_) * 4 + (h_)) * UNITB + cw1) = cx1;                                 \
    }
#define VS_LGK0 if constexpr (!A_PACKED) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* conversion stores are in LDS before the barrier that publishes them */
#define VS_MM(ha_, hb_)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                            \
            acc[(ha_) * 4 + i][(hb_) * 2 + j] = mma2<BF16>(fb[hb_][j][0], fb[hb_][j][1], fa[i][0], fa[i][1], acc[(ha_) * 4 + i][(hb_) * 2 + j]);
#define VS_BAR()                                  \
    {                                             \
        asm volatile("" ::: "memory");   /* (compiler fence: LDS accesses of the in-place conversion stay on their side) */ \
        __builtin_amdgcn_sched_barrier(0);        \
        __builtin_amdgcn_s_barrier();             \
        __builtin_amdgcn_sched_barrier(0);        \
        asm volatile("" ::: "memory");            \
    }
#define VS_WAIT(n_) asm volatile("s_waitcnt vmcnt(" #n_ ")" ::: "memory");
    // one compute segment; wr_wait_ (a VS_WAIT or nothing) is executed at its END by wave group 0 -- for group 1 the same barrier
    // interval is the preceding read segment, where the caller places the same wait
#define VS_COMPUTE(ha_, hb_, g0_wait_, all_wait_) \
    {                                             \
        VS_BAR()                                  \
        __builtin_amdgcn_s_setprio(1);            \
        VS_MM(ha_, hb_)                           \
        __builtin_amdgcn_s_setprio(0);            \
        if (wr == 0) { g0_wait_ }                 \
        all_wait_                                 \
        VS_BAR()                                  \
    }
    // MODE 0: steady state (tiles kt+1 and kt+2 exist), 1: kt == KT-2, 2: kt == KT-1
#define VS_KTILE(kt_, d_, MODE_)                                                  \
    {                                                                             \
        VS_RD_B(0, d_) VS_RD_A(0, d_)                                             \
        if (wr == 1) { if (MODE_ <= 1) { VS_WAIT(8) } else { VS_WAIT(0) } }       \
        if (MODE_ <= 1) { VS_COMPUTE(0, 0, VS_WAIT(8), ) } else { VS_COMPUTE(0, 0, VS_WAIT(0), ) } \
        VS_CVT_LD(1, d_)                                                          \
        VS_RD_B(1, d_)                                                            \
        if (MODE_ == 0) { VS_STAGE(0, (kt_) + 2, d_) VS_STAGE(2, (kt_) + 2, d_) } \
        VS_CVT_ST(1, d_)                                                          \
        VS_COMPUTE(0, 1, , VS_LGK0)                                               \
        VS_RD_A(1, d_)                                                            \
        if (MODE_ == 0) VS_STAGE(3, (kt_) + 2, d_)                                \
        if (wr == 1) { if (MODE_ == 0) { VS_WAIT(10) } else if (MODE_ == 1) { VS_WAIT(4) } } \
        if (MODE_ == 0) { VS_COMPUTE(1, 1, VS_WAIT(10), ) } else if (MODE_ == 1) { VS_COMPUTE(1, 1, VS_WAIT(4), ) } else { VS_COMPUTE(1, 1, , ) } \
        if (MODE_ <= 1) VS_CVT_LD(0, (d_) ^ 1)                                    \
        if (MODE_ == 0) VS_STAGE(1, (kt_) + 2, d_)                                \
        if (MODE_ <= 1) VS_CVT_ST(0, (d_) ^ 1)                                    \
        VS_COMPUTE(1, 0, , VS_LGK0)                                               \
    }

    // KT is even and >= 2 (checked by the launchers).  Prologue: both tiles' units in the steady-state order, A_0 of tile 0 converted by
    // its owners before the first fragment read.
    VS_STAGE(0, 0, 0) VS_STAGE(2, 0, 0) VS_STAGE(3, 0, 0) VS_STAGE(1, 0, 0) VS_STAGE(0, 1, 1) VS_STAGE(2, 1, 1) VS_STAGE(3, 1, 1) VS_STAGE(1, 1, 1)
    VS_WAIT(12)
    VS_BAR()
    VS_CVT_LD(0, 0)
    VS_CVT_ST(0, 0)
    VS_LGK0
    VS_BAR()
    if (wr == 1) VS_BAR()  // the second wave group trails the first by one barrier from here on
    for (int kt = 0; kt + 2 < KT; kt += 2) {
        VS_KTILE(kt, 0, 0)
        VS_KTILE(kt + 1, 1, 0)
    }
    VS_KTILE(KT - 2, 0, 1)
    VS_KTILE(KT - 1, 1, 2)
    if (wr == 0) VS_BAR()
#undef VS_KTILE
#undef VS_WAIT
#undef VS_COMPUTE
#undef VS_BAR
#undef VS_MM
#undef VS_CVT_LD
#undef VS_CVT_ST
#undef VS_LGK0
#undef VS_RD_B
#undef VS_RD_A
#undef VS_STAGE
    __syncthreads();  // every wave is done with the ring before an epilogue reuses it
}

// ---- 256 x 128 tile on the split main loop (the Cout = 128 convolutions of the pts3d head; round 3).  8 waves as 4 (M) x 2 (N), wave
// tile 64 x 64 (acc[4][4]); a K-tile of 32 k is THREE 16 KiB units: A_h = tile rows {wm*64 + h*32 + 0..31 : wm = 0..3} (h = 0, 1) and B =
// the 128 weight rows; ring of 2 slots x 3 units = 96 KiB.  Wave group g = wid >> 2 owns tile rows 128 g .. 128 g + 127, i.e. unit rows
// 64 g .. 64 g + 63 of both A units: it stages, converts (in place, as mainloop256_split) and reads them alone; the B unit is shared.
// Two phases of 24 MFMAs per K-tile and wave, the groups one barrier apart:
//     R0: convert A_1 of kt (own rows); read B, A_0; stage A_1 of kt+1 | M0: (A0, B) | R1: convert A_0 of kt+1; read A_1; stage A_0, B of
//     kt+2 | M1: (A1, B)
// Issue order per wave: .. R0(kt): A_1(kt+1) | R1(kt): A_0(kt+2), B(kt+2) ..  Waits (own pieces only for the A units -- a wave converts
// the rows it staged itself; the shared B unit is waited for one barrier before its first read): start of R0: A_1(kt) (four younger
// pieces in flight), start of R1: A_0(kt+1) and B(kt+1) (two younger pieces).
// A_PACKED (round 4): the A units arrive already in the packed (hi, lo) form (the producer wrote it: vs_upsample2x_nhwc relu_add + 16) -- the
// conversion loads / VALU / stores drop out, the waits and barriers stay (a unit is read one barrier after the wait that covers its pieces).
template <bool RELU_A, class Stager, bool A_PACKED = false>
__device__ __forceinline__ void mainloop256x128_split(Stager &st, const int KT, f4 (&acc)[4][4], unsigned char *smem, const int lane, const int wid) {
    static_assert(!(RELU_A && A_PACKED), "a packed A operand carries its ReLU already");
    constexpr unsigned UNITB = kUnitBytes256;
    constexpr int BF16 = kDtSplit;
    const int grp = wid >> 2, wl = wid & 3;
    const int wm = grp * 2 + (wl >> 1), wc = wl & 1;
    typedef void __attribute__((address_space(3))) *lptr_t;
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    const unsigned lds_w = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wid * 2048u);
#define VS_STAGE(u_, kt_, d_) st.stage(u_, kt_, lds_w + (unsigned)(((d_) * 3 + (u_)) * UNITB));
    const int frow = lane & 15, fg = lane >> 4;
    const unsigned rd0 = (unsigned)(frow * 128 + (((0 + fg) ^ (frow >> 1)) << 4));
    const unsigned rd1 = (unsigned)(frow * 128 + (((4 + fg) ^ (frow >> 1)) << 4));
    const unsigned char *rdA = smem + wm * (32 * 128);
    const unsigned char *rdB = smem + 2 * UNITB + wc * (64 * 128);
    unsigned char *cvp = smem + (wid * 16) * 128;
    const int crow = 2 * (frow & 7) + (frow >> 3);
    const unsigned cv0 = (unsigned)(crow * 128 + (((0 + fg) ^ (crow >> 1)) << 4));
    const unsigned cv1 = (unsigned)(crow * 128 + (((4 + fg) ^ (crow >> 1)) << 4));
    uint4 fa[2][2], fb[4][2], cx0, cx1;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
#define VS_RD_A(h_, d_)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                              \
        fa[i][0] = *reinterpret_cast<const uint4 *>(rdA + ((d_) * 3 + (h_)) * UNITB + i * 2048 + rd0);           \
        fa[i][1] = *reinterpret_cast<const uint4 *>(rdA + ((d_) * 3 + (h_)) * UNITB + i * 2048 + rd1);           \
    }
#define VS_RD_B(d_)                                                                                              \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                              \
        fb[j][0] = *reinterpret_cast<const uint4 *>(rdB + (d_) * 3 * UNITB + j * 2048 + rd0);                    \
        fb[j][1] = *reinterpret_cast<const uint4 *>(rdB + (d_) * 3 * UNITB + j * 2048 + rd1);                    \
    }
#define VS_CVT_LD(h_, d_)                                                                                        \
    if constexpr (!A_PACKED) {                                                                                   \
        cx0 = *reinterpret_cast<const uint4 *>(cvp + ((d_) * 3 + (h_)) * UNITB + cv0);                           \
        cx1 = *reinterpret_cast<const uint4 *>(cvp + ((d_) * 3 + (h_)) * UNITB + cv1);                           \
    }
#define VS_CVT_ST(h_, d_)                                                                                        \
    if constexpr (!A_PACKED) {                                                                                   \
        if constexpr (RELU_A) {                                                                                  \
            cx0.x = relu_f32_lds(cx0.x); cx0.y = relu_f32_lds(cx0.y); cx0.z = relu_f32_lds(cx0.z); cx0.w = relu_f32_lds(cx0.w); \
            cx1.x = relu_f32_lds(cx1.x); cx1.y = relu_f32_lds(cx1.y); cx1.z = relu_f32_lds(cx1.z); cx1.w = relu_f32_lds(cx1.w); \
        }                                                                                                        \
        split8_lds(cx0, cx1);                                                                                    \
        *reinterpret_cast<uint4 *>(cvp + ((d_) * 3 + (h_)) * UNITB + cv0) = cx0;                                 \
        *reinterpret_cast<uint4 *>(cvp + ((d_) * 3 + (h_)) * UNITB + cv1) = cx1;                                 \
    }
#define VS_MM(ha_)                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                            \
            acc[(ha_) * 2 + i][j] = mma2<BF16>(fb[j][0], fb[j][1], fa[i][0], fa[i][1], acc[(ha_) * 2 + i][j]);
#define VS_BAR()                                  \
    {                                             \
        asm volatile("" ::: "memory");            \
        __builtin_amdgcn_sched_barrier(0);        \
        __builtin_amdgcn_s_barrier();             \
        __builtin_amdgcn_sched_barrier(0);        \
        asm volatile("" ::: "memory");            \
    }
#define VS_WAIT(n_) asm volatile("s_waitcnt vmcnt(" #n_ ")" ::: "memory");
#define VS_COMPUTE(ha_)                           \
    {                                             \
        VS_BAR()                                  \
        __builtin_amdgcn_s_setprio(1);            \
        VS_MM(ha_)                                \
        __builtin_amdgcn_s_setprio(0);            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* conversion stores are in LDS before the barrier that publishes them */ \
        VS_BAR()                                  \
    }
    // MODE 0: steady state (tiles kt+1 and kt+2 exist), 1: kt == KT-2, 2: kt == KT-1
#define VS_KTILE(kt_, d_, MODE_)                                                  \
    {                                                                             \
        if (MODE_ <= 1) { VS_WAIT(4) } else { VS_WAIT(0) }                        \
        VS_CVT_LD(1, d_)                                                          \
        VS_RD_B(d_) VS_RD_A(0, d_)                                                \
        if (MODE_ <= 1) VS_STAGE(1, (kt_) + 1, (d_) ^ 1)                          \
        VS_CVT_ST(1, d_)                                                          \
        VS_COMPUTE(0)                                                             \
        if (MODE_ <= 1) { VS_WAIT(2) VS_CVT_LD(0, (d_) ^ 1) }                     \
        VS_RD_A(1, d_)                                                            \
        if (MODE_ == 0) { VS_STAGE(0, (kt_) + 2, d_) VS_STAGE(2, (kt_) + 2, d_) } \
        if (MODE_ <= 1) VS_CVT_ST(0, (d_) ^ 1)                                    \
        VS_COMPUTE(1)                                                             \
    }
    // KT even and >= 2.  Prologue: A_0(0), B(0) | A_1(0) | A_0(1), B(1) -- the issue order of the steady state
    VS_STAGE(0, 0, 0) VS_STAGE(2, 0, 0) VS_STAGE(1, 0, 0) VS_STAGE(0, 1, 1) VS_STAGE(2, 1, 1)
    VS_WAIT(6)
    VS_CVT_LD(0, 0)
    VS_CVT_ST(0, 0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    VS_BAR()
    if (grp == 1) VS_BAR()
    for (int kt = 0; kt + 2 < KT; kt += 2) {
        VS_KTILE(kt, 0, 0)
        VS_KTILE(kt + 1, 1, 0)
    }
    VS_KTILE(KT - 2, 0, 1)
    VS_KTILE(KT - 1, 1, 2)
    if (grp == 0) VS_BAR()
#undef VS_KTILE
#undef VS_WAIT
#undef VS_COMPUTE
#undef VS_BAR
#undef VS_MM
#undef VS_CVT_LD
#undef VS_CVT_ST
#undef VS_RD_B
#undef VS_RD_A
#undef VS_STAGE
    __syncthreads();
}

struct GemmStager256 {
    const unsigned short *pu[4][2];  // [A0 A1 B0 B1][round]
    __device__ __forceinline__ void stage(int u, int kt, unsigned lds) const {
        glds16(pu[u][0] + kt * 64, lds);
        glds16(pu[u][1] + kt * 64, lds + 1024u);
    }
};

template <int BF16, int EPI, bool APACK = false>
__global__ void __launch_bounds__(512, 1) gemm256_kernel(const GemmArgs g) {
    constexpr int BM2 = 256, BN2 = 256;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[kLdsBytes256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;

    const int tiles_n = (g.N + BN2 - 1) / BN2;
    const int tiles_m = (g.M - g.m_lo + BM2 - 1) / BM2;
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tm = bid / tiles_n, tn = bid % tiles_n;
    if (g.row_band > 0 && tiles_n > 4) {
        // bands of row_band row tiles, column tile slowest inside a band: the ~32 workgroups an XCD runs at a time then cover row_band row
        // tiles x (32 / row_band) column tiles instead of 2 x 16 -- fewer distinct W panels in flight per XCD, each A panel shared less
        const int per_band = g.row_band * tiles_n, band = bid / per_band, rows = min(g.row_band, tiles_m - band * g.row_band);
        const int r = bid - band * per_band;
        tn = r / rows; tm = band * g.row_band + (r - tn * rows);
    }
    const int m0 = g.m_lo + tm * BM2, n0 = tn * BN2;
    if (g.stagger > 0 && blockIdx.x < 256) {
        const int n = (int)((blockIdx.x >> 3) & 7) * g.stagger;
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
    }

    const unsigned short *A = reinterpret_cast<const unsigned short *>(g.A);
    const unsigned short *W = reinterpret_cast<const unsigned short *>(g.W);
    GemmStager256 st;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = unit_row256(wid, j, lane);
        const int src_chunk = unit_src_chunk256(q, lane);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ra_ = min(m0 + unit_a_tile_row256(q, h), g.M - 1);
            const size_t arow = (size_t)(ra_ / g.a_grp_in) * g.a_grp_out + g.a_grp_off + (ra_ % g.a_grp_in);
            st.pu[h][j] = A + arow * g.lda + src_chunk * 8;
            const int rw_ = min(n0 + unit_b_tile_row256(q, h), g.N - 1);
            st.pu[2 + h][j] = W + (size_t)rw_ * g.ldw + src_chunk * 8;
        }
    }
    f4 acc[8][4];
    if constexpr (BF16 == kDtSplit) mainloop256_split<false, GemmStager256, APACK>(st, g.K / 64, acc, smem, lane, wid);
    else mainloop256<BF16, false>(st, g.K / 64, acc, smem, lane, wid);
    gemm_epilogue<BF16, EPI, 8>(g, acc, m0 + wr * 128, n0 + wc * 64, smem, wid, lane);
}

// Logical workgroup id of the split-K / tap-fused kernels.  The hardware hands consecutive blockIdx.x to consecutive XCDs (8 on MI355X, each
// with its own 4 MB L2), so "tile fastest inside a K slice" put the workgroups that read the same rows of A and W on EIGHT different L2s:
// the counters showed the split-class weight gradients fetching 151 GB (linears) + 122 GB (3x3 convolutions) per 8-scene training step for
// ~55 GB of operands, at 4-5.5 TB/s.  With g.row_band != 0 (the weight-gradient entries set it: VS_WGRAD_XCD, default 1) consecutive LOGICAL
// ids share an XCD (gemm256_kernel's remap): the nine taps / the tiles of one K slice meet in one L2.
__device__ __forceinline__ int splitk_logical_block(const GemmArgs &g) {
    const int b = blockIdx.x;
    if (g.row_band == 0) return b;
    const int n = gridDim.x, q = n >> 3, r = n & 7, xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Split-K / tap-fused variant for weight gradients: out32[tap][M,N] += A[M, Kslice] (W + shift[tap])[N, Kslice]^T through f32
// atomics, or (g.partials) per-slice partial tiles for splitk_reduce_kernel.  blockIdx.x = (k-slice, tap, tile), tile fastest: the workgroups of one K slice run together, so A and the (up to
// nine, overlapping) shifted views of W of that slice are shared through L2.  K / 64 / ksplit must be even and >= 2.
template <int BF16>
__global__ void __launch_bounds__(512, 1) gemm256_splitk_kernel(const GemmArgs g_in) {
    constexpr int BM2 = 256, BN2 = 256;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[kLdsBytes256];
    GemmArgs g = g_in;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;

    const int tiles_n = (g.N + BN2 - 1) / BN2;
    const int tiles = ((g.M + BM2 - 1) / BM2) * tiles_n;
    const int ntaps = g.ntaps > 0 ? g.ntaps : 1;
    const int lb = splitk_logical_block(g);
    const int ksp = lb / (tiles * ntaps);
    const int rem = lb - ksp * tiles * ntaps;
    const int tap = rem / tiles, bid = rem - tap * tiles;
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * BM2, n0 = tn * BN2;
    const int KT = g.K / 64 / g.ksplit;

    const unsigned short *A = reinterpret_cast<const unsigned short *>(g.A) + ksp * (g.a_slice_stride ? g.a_slice_stride : (long long)KT * 64);
    const unsigned short *W = reinterpret_cast<const unsigned short *>(g.W) + ksp * (g.w_slice_stride ? g.w_slice_stride : (long long)KT * 64);
    if (g.ntaps > 0) {
        if (g.tap_on_a) A += g_in.tap_shift[tap];
        else W += g_in.tap_shift[tap];  // index the kernarg: a dynamically indexed local copy would live in scratch
        g.out = reinterpret_cast<float *>(g.out) + (long long)tap * g.tap_out_stride;
    }
    g.ksplit = 2;  // epilogue: "partial sums meet through atomics" ...
    if (g.partials) {  // ... or are stored to the workspace for splitk_reduce_kernel
        g.out = g.partials + ((long long)(ksp * ntaps + tap) * g.M) * g.N;
        g.ldo = g.N;
        g.ksplit = -1;
    }
    GemmStager256 st;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = unit_row256(wid, j, lane);
        const int src_chunk = unit_src_chunk256(q, lane);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ra_ = min(m0 + unit_a_tile_row256(q, h), g.M - 1);
            st.pu[h][j] = A + (size_t)ra_ * g.lda + src_chunk * 8;
            const int rw_ = min(n0 + unit_b_tile_row256(q, h), g.N - 1);
            st.pu[2 + h][j] = W + (size_t)rw_ * g.ldw + src_chunk * 8;
        }
    }
    f4 acc[8][4];
    if constexpr (BF16 == kDtSplit) mainloop256_split<false>(st, KT, acc, smem, lane, wid);   // (split class: A f32, W packed; K counts 2-byte units)
    else mainloop256<BF16, false>(st, KT, acc, smem, lane, wid);
    g.bias = nullptr;
    g.gate = nullptr;
    gemm_epilogue<BF16, 2, 8>(g, acc, m0 + wr * 128, n0 + wc * 64, smem, wid, lane);
}

// ---- reduction-major (TN) weight gradient: out32[M, N] (+)= sum_k A[k, m] W[k, n], A = dY [tokens, M], W = X [tokens, N] as they
// are in memory.  Same tiling, K slicing and epilogue as gemm256_splitk_kernel; only the staging (column segments of 64
// reduction rows, rows >= k_valid from a zero page) and the fragment reads (transpose reads) differ. ----
__device__ __attribute__((aligned(256))) unsigned short vs_zero_row256[128] = {0};

struct GemmStagerTN {
    const unsigned short *pu[4][2];  // [A0 A1 B0 B1][round]: this lane's 16 bytes in reduction row rrow[round] of K-tile 0
    long long kst[2];                // element stride of one K-tile (64 reduction rows) in A / W
    int rrow[2], klim;               // the lane's reduction row inside a K-tile per round; rows >= klim (slice-relative) are zero
    __device__ __forceinline__ void stage(int u, int kt, unsigned lds) const {
        const unsigned short *z = vs_zero_row256;
        const long long off = kt * kst[u >> 1];
        glds16((kt * 64 + rrow[0] < klim && pu[u][0]) ? pu[u][0] + off : z, lds);
        glds16((kt * 64 + rrow[1] < klim && pu[u][1]) ? pu[u][1] + off : z, lds + 1024u);
    }
};

template <int BF16>
__global__ void __launch_bounds__(512, 1) gemm256_tn_splitk_kernel(const GemmArgs g_in) {
    constexpr int BM2 = 256, BN2 = 256;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[kLdsBytes256];
    GemmArgs g = g_in;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int tiles_n = (g.N + BN2 - 1) / BN2;
    const int tiles = ((g.M + BM2 - 1) / BM2) * tiles_n;
    const int lb = splitk_logical_block(g);
    const int ksp = lb / tiles;
    const int bid = lb - ksp * tiles;
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * BM2, n0 = tn * BN2;
    const int KT = g.K / 64 / g.ksplit;
    const long long k0 = (long long)ksp * KT * 64;       // first reduction row of this slice

    const unsigned short *A = reinterpret_cast<const unsigned short *>(g.A) + k0 * g.lda;
    const unsigned short *W = reinterpret_cast<const unsigned short *>(g.W) + k0 * g.ldw;
    GemmStagerTN st;
    st.kst[0] = 64LL * g.lda; st.kst[1] = 64LL * g.ldw;
    st.klim = (int)max(0LL, min((long long)KT * 64, (long long)g.k_valid - k0));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (wid * 2 + j) * 4 + (lane >> 4);       // reduction row inside the K-tile that this lane's 16 bytes belong to
        const int c = (lane & 15) ^ ((r & 3) << 1);           // logical 16-byte chunk of the unit row that must land at slot lane & 15
        st.rrow[j] = r;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // 16-byte column chunks that start beyond the row's storage (outputs that are not whole 256-tiles: skinny layers) come
            // from the zero page; chunks inside [M, lda) hold whatever pads the row -- their outputs are masked by the epilogue
            const int ca = m0 + (c >> 3) * 128 + h * 64 + (c & 7) * 8;       // A_h: tile rows wr*128 + h*64 + 0..63
            const int cw = n0 + (c >> 2) * 64 + h * 32 + (c & 3) * 8;        // B_h: tile cols wc*64 + h*32 + 0..31
            st.pu[h][j] = ca + 8 <= g.lda ? A + (long long)r * g.lda + ca : nullptr;
            st.pu[2 + h][j] = cw + 8 <= g.ldw ? W + (long long)r * g.ldw + cw : nullptr;
        }
    }
    f4 acc[8][4];
    mainloop256<BF16, false, GemmStagerTN, true>(st, KT, acc, smem, lane, wid);
    g.bias = nullptr;
    g.gate = nullptr;
    g.ksplit = 2;
    if (g.partials) {
        g.out = g.partials + ((long long)ksp * g.M) * g.N;
        g.ldo = g.N;
        g.ksplit = -1;
    }
    gemm_epilogue<BF16, 2, 8>(g, acc, m0 + wr * 128, n0 + wc * 64, smem, wid, lane);
}

// ---- split-class weight gradient with the A operand reduction-major (mainloop256_split, A_TN): out32[M, N] (+)= sum_k A[k, m] Wp[n, k],
// A = dY [tokens, M] f32 as it is in memory, Wp = the packed transposed X (vs_transpose_pack_split) [N, tokens].  Tiling, K slicing and
// epilogue of gemm256_splitk_kernel; a K-tile is 32 tokens.  g.K = padded reduction length in tokens, g.k_valid = rows of A that exist. ----
struct SplitStagerATN {
    const float *pa[2][2];            // [A unit h][round]: this lane's 16 bytes in k row rrow[round] of K-tile 0 (nullptr: beyond the row)
    const unsigned short *pw[2][2];   // [B unit h][round]
    long long kstA;                   // floats per K-tile of A (32 rows)
    int rrow[2], klim;
    __device__ __forceinline__ void stage(int u, int kt, unsigned lds) const {
        if (u < 2) {
            const unsigned short *z = vs_zero_row256;
            glds16((kt * 32 + rrow[0] < klim && pa[u][0]) ? (const void *)(pa[u][0] + kt * kstA) : (const void *)z, lds);
            glds16((kt * 32 + rrow[1] < klim && pa[u][1]) ? (const void *)(pa[u][1] + kt * kstA) : (const void *)z, lds + 1024u);
        } else {
            glds16(pw[u - 2][0] + kt * 64, lds);
            glds16(pw[u - 2][1] + kt * 64, lds + 1024u);
        }
    }
};

__global__ void __launch_bounds__(512, 1) gemm256_split_atn_splitk_kernel(const GemmArgs g_in) {
    constexpr int BM2 = 256, BN2 = 256;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[kLdsBytes256];
    GemmArgs g = g_in;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int tiles_n = (g.N + BN2 - 1) / BN2;
    const int tiles = ((g.M + BM2 - 1) / BM2) * tiles_n;
    const int lb = splitk_logical_block(g);
    const int ksp = lb / tiles;
    const int bid = lb - ksp * tiles;
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * BM2, n0 = tn * BN2;
    const int KT = g.K / 32 / g.ksplit;                  // K-tiles of 32 tokens in this slice (even, >= 2: checked by the entry)
    const long long k0 = (long long)ksp * KT * 32;

    const float *A = reinterpret_cast<const float *>(g.A) + k0 * g.lda;
    const unsigned short *W = reinterpret_cast<const unsigned short *>(g.W) + k0 * 2;     // packed: 64 two-byte units per 32 tokens
    SplitStagerATN st;
    st.kstA = 32LL * g.lda;
    st.klim = (int)max(0LL, min((long long)KT * 32, (long long)g.k_valid - k0));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        {   // A: k row and 16-byte slot of this lane inside its wave's 2 KiB; the slot holds logical chunk slot ^ ((k & 3) << 1)
            const int rr = lane >> 4, r = (wid & 3) * 8 + j * 4 + rr;
            const int c = (lane & 15) ^ (rr << 1);
            st.rrow[j] = r;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ca = m0 + wr * 128 + h * 64 + c * 4;
                st.pa[h][j] = ca + 4 <= g.lda ? A + (long long)r * g.lda + ca : nullptr;
            }
        }
        const int q = unit_row256(wid, j, lane);
        const int src_chunk = unit_src_chunk256(q, lane);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rw_ = min(n0 + unit_b_tile_row256(q, h), g.N - 1);
            st.pw[h][j] = W + (size_t)rw_ * g.ldw + src_chunk * 8;
        }
    }
    f4 acc[8][4];
    mainloop256_split<false, SplitStagerATN, false, true>(st, KT, acc, smem, lane, wid);
    g.bias = nullptr;
    g.gate = nullptr;
    g.ksplit = 2;
    if (g.partials) {
        g.out = g.partials + ((long long)ksp * g.M) * g.N;
        g.ldo = g.N;
        g.ksplit = -1;
    }
    gemm_epilogue<kDtSplit, 2, 8>(g, acc, m0 + wr * 128, n0 + wc * 64, smem, wid, lane);
}

// ---- split-class weight gradient of a 3x3 convolution (stride 1, pad 1) with X read as it is: out32[tap][ci][co] (+)= sum over pixels p of
// act(X)[p + shift(tap)][ci] * dY[p][co]; X [pixels, Cin] f32 NHWC (the reduction-major A operand: tap shift = pixel-row shift, zero page
// outside the image, per lane with an incremental (y, x) counter as ConvWgradStagerTN), Wp = vs_transpose_pack_split(dY) [Cout, Ppad].
// blockIdx.x = (k-slice, tap, tile), tile fastest: the nine taps of a slice share X and dY through L2.  Cin, Cout multiples of 256. ----
struct ConvSplitStagerATN {
    const float *pa[2][2];            // [A unit h][round]: tap-shifted, K-tile 0
    const unsigned short *pw[2][2];
    long long kstA;
    int rrow[2], klim;
    int py[2][2], px[2][2];           // [A unit h][round]: (y, x) of the lane's (unshifted) pixel at the unit's NEXT staging call
    int ty, tx, H, W, q32, r32;       // tap offset in {-1, 0, 1}^2; 32 = q32 * W + r32
    __device__ __forceinline__ void stage(int u, int kt, unsigned lds) {
        if (u < 2) {
            const unsigned short *z = vs_zero_row256;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int y = py[u][j] + ty, x = px[u][j] + tx;
                const bool ok = kt * 32 + rrow[j] < klim && pa[u][j] && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
                glds16(ok ? (const void *)(pa[u][j] + kt * kstA) : (const void *)z, lds + 1024u * j);
                int nx = px[u][j] + r32, ny = py[u][j] + q32;
                if (nx >= W) { nx -= W; ++ny; }
                while (ny >= H) ny -= H;
                px[u][j] = nx; py[u][j] = ny;
            }
        } else {
            glds16(pw[u - 2][0] + kt * 64, lds);
            glds16(pw[u - 2][1] + kt * 64, lds + 1024u);
        }
    }
};

template <bool RELU_A>
__global__ void __launch_bounds__(512, 1) conv3x3_wgrad_split_atn_kernel(const GemmArgs g_in) {
    constexpr int BM2 = 256, BN2 = 256;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[kLdsBytes256];
    GemmArgs g = g_in;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int tiles_n = g.N / BN2;
    const int tiles = (g.M / BM2) * tiles_n;
    const int lb = splitk_logical_block(g);
    const int ksp = lb / (tiles * 9);
    const int rem = lb - ksp * tiles * 9;
    const int tap = rem / tiles, bid = rem - tap * tiles;
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * BM2, n0 = tn * BN2;
    const int KT = g.K / 32 / g.ksplit;
    const long long k0 = (long long)ksp * KT * 32;

    ConvSplitStagerATN st;
    st.ty = tap / 3 - 1; st.tx = tap - (tap / 3) * 3 - 1;
    st.H = g.conv_H; st.W = g.conv_W; st.q32 = 32 / g.conv_W; st.r32 = 32 % g.conv_W;
    const long long shift = (long long)st.ty * g.conv_W + st.tx;
    const float *A = reinterpret_cast<const float *>(g.A) + (k0 + shift) * g.lda;
    const unsigned short *W = reinterpret_cast<const unsigned short *>(g.W) + k0 * 2;
    st.kstA = 32LL * g.lda;
    st.klim = (int)max(0LL, min((long long)KT * 32, (long long)g.k_valid - k0));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        {
            const int rr = lane >> 4, r = (wid & 3) * 8 + j * 4 + rr;
            const int c = (lane & 15) ^ (rr << 1);
            st.rrow[j] = r;
            const int rem_p = (int)((k0 + r) % ((long long)g.conv_H * g.conv_W));
            const int y0 = rem_p / g.conv_W, x0 = rem_p - y0 * g.conv_W;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ca = m0 + wr * 128 + h * 64 + c * 4;
                st.pa[h][j] = ca + 4 <= g.lda ? A + (long long)r * g.lda + ca : nullptr;
                st.py[h][j] = y0; st.px[h][j] = x0;
            }
        }
        const int q = unit_row256(wid, j, lane);
        const int src_chunk = unit_src_chunk256(q, lane);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rw_ = min(n0 + unit_b_tile_row256(q, h), g.N - 1);
            st.pw[h][j] = W + (size_t)rw_ * g.ldw + src_chunk * 8;
        }
    }
    f4 acc[8][4];
    mainloop256_split<RELU_A, ConvSplitStagerATN, false, true>(st, KT, acc, smem, lane, wid);
    g.bias = nullptr;
    g.gate = nullptr;
    g.out = reinterpret_cast<float *>(g.out) + (long long)tap * g.tap_out_stride;
    g.ksplit = 2;
    if (g.partials) {
        g.out = g.partials + ((long long)(ksp * 9 + tap) * g.M) * g.N;
        g.ldo = g.N;
        g.ksplit = -1;
    }
    gemm_epilogue<kDtSplit, 2, 8>(g, acc, m0 + wr * 128, n0 + wc * 64, smem, wid, lane);
}

// ---- reduction-major weight gradient of a 3x3 convolution (stride 1, pad 1): out32[tap][ci][co] (+)= sum over pixels p of
// act(X)[p + shift(tap)][ci] * dY[p][co], X [pixels, Cin] and dY [pixels, Cout] NHWC as they are -- no zero-bordered, transposed
// copies.  The A operand (X) of tap (ty, tx) is the pixel row shifted by (ty-1)*W + (tx-1); where that pixel falls outside the
// image the lane reads the zero page instead (the forward conv's scheme, conv.hip), tracked per lane with an incremental
// (y, x) counter: every staging call of a unit advances its pixel by one K-tile of 64 rows.  ReLU on X = RELU_A of the main loop.
struct ConvWgradStagerTN {
    const unsigned short *pa[2][2], *pw[2][2];  // [unit h][round]: this lane's 16 bytes for K-tile 0 (A already tap-shifted)
    long long kstA, kstW;
    int rrow[2], klim;
    int py[2][2], px[2][2];                      // [A unit h][round]: (y, x) of the lane's pixel at the unit's NEXT staging call
    int dyy[2][2], dxx[2][2];                    // [A unit h][round]: the tap offset in {-1, 0, 1}^2 of the lane's tile row (tap groups)
    int H, W, q64, r64;                          // 64 = q64 * W + r64
    __device__ __forceinline__ void stage(int u, int kt, unsigned lds) {
        const unsigned short *z = vs_zero_row256;
        if (u < 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int y = py[u][j] + dyy[u][j], x = px[u][j] + dxx[u][j];
                const bool ok = kt * 64 + rrow[j] < klim && pa[u][j] && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
                glds16(ok ? pa[u][j] + kt * kstA : z, lds + 1024u * j);
                int nx = px[u][j] + r64, ny = py[u][j] + q64;   // advance this unit's pixel by 64
                if (nx >= W) { nx -= W; ++ny; }
                while (ny >= H) ny -= H;
                px[u][j] = nx; py[u][j] = ny;
            }
        } else {
            const long long off = kt * kstW;
            glds16((kt * 64 + rrow[0] < klim && pw[u - 2][0]) ? pw[u - 2][0] + off : z, lds);
            glds16((kt * 64 + rrow[1] < klim && pw[u - 2][1]) ? pw[u - 2][1] + off : z, lds + 1024u);
        }
    }
};

template <int BF16, bool RELU_A>
__global__ void __launch_bounds__(512, 1) conv3x3_wgrad_tn_kernel(const GemmArgs g_in) {
    constexpr int BM2 = 256, BN2 = 256;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[kLdsBytes256];
    GemmArgs g = g_in;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    // tap groups: with Cin <= 128 (a divisor of 256) the 256 tile rows hold G = 256 / Cin taps side by side -- out[tap][ci][co] is
    // contiguous over (tap, ci), so rows m = tap_in_group * Cin + ci of the tile ARE rows of out + group * G * Cin * Cout
    const int Cin = g.M;
    const int G = (Cin <= 128 && 256 % Cin == 0) ? 256 / Cin : 1;
    const int ngroups = (9 + G - 1) / G;
    const int tiles_n = (g.N + BN2 - 1) / BN2;
    const int tiles = (G > 1 ? 1 : (Cin + BM2 - 1) / BM2) * tiles_n;
    const int lb = splitk_logical_block(g);
    const int ksp = lb / (tiles * ngroups);        // (k-slice, tap group, tile): the taps of a slice share X and dY through L2
    const int rem = lb - ksp * tiles * ngroups;
    const int grp = rem / tiles, bid = rem - grp * tiles;
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * BM2, n0 = tn * BN2;
    const int KT = g.K / 64 / g.ksplit;
    const long long k0 = (long long)ksp * KT * 64;
    const int taps_here = min(G, 9 - grp * G);

    const unsigned short *A = reinterpret_cast<const unsigned short *>(g.A) + k0 * g.lda;
    const unsigned short *W = reinterpret_cast<const unsigned short *>(g.W) + k0 * g.ldw;
    ConvWgradStagerTN st;
    st.kstA = 64LL * g.lda; st.kstW = 64LL * g.ldw;
    st.klim = (int)max(0LL, min((long long)KT * 64, (long long)g.k_valid - k0));
    st.H = g.conv_H; st.W = g.conv_W; st.q64 = 64 / g.conv_W; st.r64 = 64 % g.conv_W;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (wid * 2 + j) * 4 + (lane >> 4);
        const int c = (lane & 15) ^ ((r & 3) << 1);
        st.rrow[j] = r;
        const long long p = k0 + r;                          // the lane's pixel at K-tile 0
        const int rem_p = (int)(p % ((long long)g.conv_H * g.conv_W));
        const int y0 = rem_p / g.conv_W, x0 = rem_p - y0 * g.conv_W;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ca = m0 + (c >> 3) * 128 + h * 64 + (c & 7) * 8;   // tile row = (tap in group, input channel)
            const int cw = n0 + (c >> 2) * 64 + h * 32 + (c & 3) * 8;
            const int tl = G > 1 ? ca / Cin : 0, ci = G > 1 ? ca - tl * Cin : ca;
            const int tap = grp * G + tl;
            const int ty = tap / 3 - 1, tx = tap - (tap / 3) * 3 - 1;
            st.dyy[h][j] = ty; st.dxx[h][j] = tx;
            const long long shift = (long long)ty * g.conv_W + tx;
            st.pa[h][j] = (tl < taps_here && ci + 8 <= g.lda) ? A + (shift + r) * g.lda + ci : nullptr;
            st.pw[h][j] = cw + 8 <= g.ldw ? W + (long long)r * g.ldw + cw : nullptr;
            st.py[h][j] = y0; st.px[h][j] = x0;
        }
    }
    f4 acc[8][4];
    mainloop256<BF16, RELU_A, ConvWgradStagerTN, true>(st, KT, acc, smem, lane, wid);
    g.bias = nullptr;
    g.gate = nullptr;
    g.out = reinterpret_cast<float *>(g.out) + (long long)grp * G * g.tap_out_stride;
    g.ksplit = 2;
    if (g.partials) {
        g.out = g.partials + ((long long)(ksp * 9 + grp * G) * Cin) * g.N;
        g.ldo = g.N;
        g.ksplit = -1;
    }
    if (G > 1) { g.M = taps_here * Cin; g.grp_in = g.M; g.grp_out = g.M; }   // rows of the group's taps, contiguous in out
    gemm_epilogue<BF16, 2, 8>(g, acc, m0 + wr * 128, n0 + wc * 64, smem, wid, lane);
}

}  // namespace
