// DPT-head convolutions for gfx950: 3x3 (stride 1, pad 1) implicit-GEMM on MFMA + bilinear x2 upsampling.
// the two DPT heads are 42 % of the forward FLOPs and one layer (conv3 256->256 @256^2,
// heads/dpt_block.py:338) is 18 % of a frame.  Replaces nn.Conv2d(k=3,p=1) / ResidualConvUnit / F.interpolate(scale 2,
// bilinear, align_corners=True) of heads/dpt_block.py:79-218,316-343 on NHWC 16-bit activations.
//
// conv3x3 as GEMM:  M = N*H*W output pixels, N = Cout, K = 9*Cin ordered (tap, cin); A[m, (t,c)] = in[n, y+dy, x+dx, c].
// With NHWC activations the 64-channel slice of one tap is 128 contiguous bytes of the shifted pixel, i.e. exactly one
// A-tile row of the GEMM kernel (gemm.hip): the same global_load_lds_dwordx4 + XOR-swizzle staging applies, only the
// per-row source pointer changes with the tap, and out-of-image taps read a 128-byte zero page.  No im2col buffer ever
// exists.  Fused: ReLU on the input (ResidualConvUnit applies the activation BEFORE each conv), bias, residual add,
// ReLU on the output.
#include "gemm256.h"

#include <cstdlib>

__device__ __attribute__((aligned(128))) unsigned short vs_zero_page[64] = {0};

namespace {

struct ConvArgs {
    const unsigned short *in;   // [N,H,W,Cin]
    const unsigned short *w;    // [Cout, 3, 3, Cin]
    const float *bias;          // [Cout] or null
    const unsigned short *res;  // [N,H,W,Cout] or null
    unsigned short *out;        // [N,H,W,Cout]
    int Nimg, H, W, Cin, Cout;  // H, W: OUTPUT size
    int relu_in, relu_out;
    int Hin, Win, stride;
    // fused head (vs_conv3x3_head1x1_nhwc): out2[pixel, 0..C2) = W2 relu(conv3x3(in) + bias) + bias2, the 3x3 result never leaves the
    // workgroup.  w2: [C2pad, Cout] 16-bit (rows >= C2 zero), bias2: [C2pad] f32, out2 row stride ld2 (>= C2pad for the MFMA form)
    const unsigned short *w2;
    const float *bias2;
    unsigned short *out2;
    int C2, C2pad, ld2;
    float acc_scale;            // split operands: 2^-e of the packed weights (gemm_common.h, kDtSplit)
    float acc_scale2;           // split operands, fused head: 2^-e of the packed w2
    int a_packed;               // split operands: `in` is already the packed (hi, lo) image (ops.split_act; any ReLU applied by its producer)
    const float *res2;          // f32 activations: a SECOND residual [N,H,W,Cout] added with `res` (the FeatureFusionBlock's x + rcu(skip), dpt_block.py:196-208)
    int korder;                 // 256-tile kernels: 0 = K-tiles tap-major (tap, channel block), 1 = channel-block-major (see ConvStager256)
};

// K-tile order of the 256-tile implicit-GEMM kernels (VS_CONV_KORDER, default 1).  Tap-major, the three kx taps of one image row re-read the
// same lines kpt K-tiles apart -- x 32 resident workgroups per XCD that is more than the 4 MB L2 holds, and the counters showed the fused
// Gaussian head fetching 124 GB per step for a 12.9 GB input.  Channel-block-major, the nine taps of one 64-channel block are consecutive
// K-tiles: the kx re-reads are one K-tile apart.  Only the ORDER of the reduction changes (f32 rounding), not the operands.
static int conv_korder() {
    static const int v = [] { const char *e = getenv("VS_CONV_KORDER"); return e ? atoi(e) : 1; }();
    return v;
}

// epilogue shared by the conv kernels (accumulators are C^T, see gemm_common.h: a lane holds 4 consecutive output
// channels of one pixel per fragment): + bias, + residual (8-byte load, added in f32), ReLU, round to 16 bit, one 8-byte
// store per fragment -- no LDS round trip.  mw0 / nbase = first output pixel / channel of the wave's tile.
template <int BF16, int MI>
__device__ __forceinline__ void conv_epilogue(const ConvArgs &g, f4 (&acc)[MI][4], int mw0, int nbase, int M, void *, int, int lane) {
    const int mrow = lane & 15, c4 = (lane >> 4) * 4;
    if constexpr (BF16 == kDtSplit) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] *= g.acc_scale;
    }
    float bv[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = nbase + j * 16 + c4 + r;
            bv[j][r] = (g.bias && n < g.Cout) ? g.bias[n] : 0.0f;
        }
    const bool vec_ok = (g.Cout % 4 == 0) && (nbase + 64 <= g.Cout) && ((reinterpret_cast<uintptr_t>(g.out) & 7) == 0) &&
                        (!g.res || (reinterpret_cast<uintptr_t>(g.res) & 7) == 0);
    // interior wave tile, 16-bit output: 16-byte stores -- row fragments in pairs, v_permlane16_swap gives an even-row lane 8 consecutive
    // columns of fragment i and an odd-row lane 8 consecutive columns of fragment i + 1 (as in gemm_epilogue: the tile's store tail is
    // issue-bound, 8-byte stores cost twice the instructions)
    if constexpr (!is_f32io(BF16) && MI % 2 == 0) {
        if (vec_ok && g.Cout % 8 == 0 && (reinterpret_cast<uintptr_t>(g.out) & 15) == 0 && mw0 + 16 * MI <= M) {
            constexpr int D16 = BF16;
            typedef unsigned u2v_ __attribute__((ext_vector_type(2)));
            const bool odd = (lane >> 4) & 1;
            uint2 prev[4];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const size_t o = (size_t)(mw0 + i * 16 + mrow) * g.Cout + nbase + c4;
                uint2 pk[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bv[j][r];
                    if (g.res) {
                        const uint2 rv = *reinterpret_cast<const uint2 *>(g.res + o + j * 16);
                        const float r0 = from16<D16>((unsigned short)(rv.x & 0xffffu)), r1 = from16<D16>((unsigned short)(rv.x >> 16));
                        const float r2 = from16<D16>((unsigned short)(rv.y & 0xffffu)), r3 = from16<D16>((unsigned short)(rv.y >> 16));
                        if (g.relu_out == 2) {
                            v[0] = r0 > 0.f ? v[0] : 0.f; v[1] = r1 > 0.f ? v[1] : 0.f; v[2] = r2 > 0.f ? v[2] : 0.f; v[3] = r3 > 0.f ? v[3] : 0.f;
                        } else {
                            v[0] += r0; v[1] += r1; v[2] += r2; v[3] += r3;
                        }
                    }
                    if (g.relu_out == 1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
                    }
                    pk[j].x = pack16x2<D16>(v[0], v[1]);
                    pk[j].y = pack16x2<D16>(v[2], v[3]);
                }
                if ((i & 1) == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) prev[j] = pk[j];
                } else {
                    unsigned short *dst = g.out + (size_t)(mw0 + (odd ? i : i - 1) * 16 + mrow) * g.Cout + nbase + (c4 & ~4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const u2v_ sx = __builtin_amdgcn_permlane16_swap(prev[j].x, pk[j].x, false, false);
                        const u2v_ sy = __builtin_amdgcn_permlane16_swap(prev[j].y, pk[j].y, false, false);
                        *reinterpret_cast<uint4 *>(dst + j * 16) = make_uint4(sx.x, sy.x, sx.y, sy.y);
                    }
                }
            }
            return;
        }
    }
    // f32 activations (f32 / split operand classes), interior wave tile: 16-byte residual loads and output stores, straight-line
    if constexpr (is_f32io(BF16)) {
        if (g.Cout % 4 == 0 && nbase + 64 <= g.Cout && mw0 + 16 * MI <= M &&
            ((reinterpret_cast<uintptr_t>(g.out) | reinterpret_cast<uintptr_t>(g.res) | reinterpret_cast<uintptr_t>(g.res2)) & 15) == 0) {
            const float *res32 = reinterpret_cast<const float *>(g.res);
            float *out32 = reinterpret_cast<float *>(g.out);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const size_t o = (size_t)(mw0 + i * 16 + mrow) * g.Cout + nbase + c4;
                float4 rv[4];
                if (res32) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) rv[j] = *reinterpret_cast<const float4 *>(res32 + o + j * 16);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bv[j][r];
                    if (res32) {
                        const float rr[4] = {rv[j].x, rv[j].y, rv[j].z, rv[j].w};
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = g.relu_out == 2 ? (rr[r] > 0.f ? v[r] : 0.f) : v[r] + rr[r];
                    }
                    if (g.res2) {
                        const float4 r2 = *reinterpret_cast<const float4 *>(g.res2 + o + j * 16);
                        v[0] += r2.x; v[1] += r2.y; v[2] += r2.z; v[3] += r2.w;
                    }
                    if (g.relu_out == 1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
                    }
                    *reinterpret_cast<float4 *>(out32 + o + j * 16) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = mw0 + i * 16 + mrow;
        if (m >= M) continue;
        const size_t o = (size_t)m * g.Cout + nbase + c4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bv[j][r];
            if constexpr (is_f32io(BF16)) {   // f32 activations: residual / output are float arrays with the same element indices
                const float *res32 = reinterpret_cast<const float *>(g.res);
                float *out32 = reinterpret_cast<float *>(g.out);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (nbase + j * 16 + c4 + r < g.Cout) {
                        float x = v[r];
                        if (res32) {
                            const float rr = res32[o + j * 16 + r];
                            x = g.relu_out == 2 ? (rr > 0.f ? x : 0.f) : x + rr;
                        }
                        if (g.res2) x += g.res2[o + j * 16 + r];
                        if (g.relu_out == 1) x = fmaxf(x, 0.0f);
                        v[r] = x;
                    }
                if (nbase + j * 16 + c4 + 3 < g.Cout && g.Cout % 4 == 0) {
                    *reinterpret_cast<float4 *>(out32 + o + j * 16) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (nbase + j * 16 + c4 + r < g.Cout) out32[o + j * 16 + r] = v[r];
                }
                continue;
            }
            constexpr int D16 = is_f32io(BF16) ? 0 : BF16;
            if (vec_ok) {
                if (g.res) {
                    const uint2 rv = *reinterpret_cast<const uint2 *>(g.res + o + j * 16);
                    const float r0 = from16<D16>((unsigned short)(rv.x & 0xffffu)), r1 = from16<D16>((unsigned short)(rv.x >> 16));
                    const float r2 = from16<D16>((unsigned short)(rv.y & 0xffffu)), r3 = from16<D16>((unsigned short)(rv.y >> 16));
                    if (g.relu_out == 2) {  // data gradient of a conv behind a ReLU: keep it where the ReLU's input was positive
                        v[0] = r0 > 0.f ? v[0] : 0.f; v[1] = r1 > 0.f ? v[1] : 0.f; v[2] = r2 > 0.f ? v[2] : 0.f; v[3] = r3 > 0.f ? v[3] : 0.f;
                    } else {
                        v[0] += r0; v[1] += r1; v[2] += r2; v[3] += r3;
                    }
                }
                if (g.relu_out == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
                }
                uint2 pk;
                pk.x = pack16x2<D16>(v[0], v[1]);
                pk.y = pack16x2<D16>(v[2], v[3]);
                *reinterpret_cast<uint2 *>(g.out + o + j * 16) = pk;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (nbase + j * 16 + c4 + r < g.Cout) {
                        float x = v[r];
                        if (g.res) {
                            const float rr = from16<D16>(g.res[o + j * 16 + r]);
                            x = g.relu_out == 2 ? (rr > 0.f ? x : 0.f) : x + rr;
                        }
                        if (g.relu_out == 1) x = fmaxf(x, 0.0f);
                        g.out[o + j * 16 + r] = to16<D16>(x);
                    }
            }
        }
    }
}

// ---- fused 1x1 head behind a 3x3 convolution with Cout = 256 (the Gaussian-parameter head: conv3(256->256, no bias) -> ReLU ->
// conv1(256->83), dpt_block.py:335-343).  The [256 pixels x 256 channels] tile of the 3x3 result is multiplied by W2 [C2pad <= 96, 256]
// on the MFMA inside the workgroup: it saves the write + read of the 256-channel activation at full resolution (2 x 6.4 GB per
// 24-scene step).  Round 2, after cycle stamps (tools notes in profiles/round2_pmc_gemm256.md: of a 63 us tile the second GEMM took 11.8 us
// and the staging 5.7 -- the W2 fragments came from L2 one k-step ahead of their use, a latency per step):
//   * W2 (48 KiB) is brought into LDS ONCE per workgroup by LDS-DMA, issued right after the K loop so that it lands under the staging
//     arithmetic; rows of 512 B, 16-byte chunk index XOR (row & 31) applied on the source side (LDS-DMA writes lanes linearly);
//   * the 3x3 result goes to LDS as 16-bit values (the rounding the unfused path applies when it stores the activation) in TWO channel
//     halves of 64 KiB (rows of 256 B, chunk XOR (row & 15)): the waves that own channels 0..127 stage, all eight waves run k-steps
//     0..3 of the second GEMM (wave w: pixel rows 32 w .. 32 w + 31, 2 x NF fragments), then the other half -- W2 and a half tile
//     fit the 160 KiB together, a full tile (128 KiB) did not;
//   * the [256 x C2pad] result leaves as 16-byte stores (v_permlane16_swap pairs the two row fragments of a wave).
template <int BF16, int NF>
__device__ __forceinline__ void conv_head1x1_epilogue256(const ConvArgs &g, f4 (&acc)[8][4], int m0, int wr, int wc, unsigned char *smem, int wid,
                                                         int lane) {
    typedef unsigned u2v_ __attribute__((ext_vector_type(2)));
    const int mrow = lane & 15, grp = lane >> 4;
    unsigned char *sX = smem;                 // [256 rows][256 B]
    unsigned char *sW = smem + 64 * 1024;     // [NF * 16 rows][512 B]
    {   // W2 -> LDS: 16-byte piece p = round * 512 + tid holds (row p >> 5, LDS slot p & 31) = global chunk (slot ^ (row & 31))
        typedef void __attribute__((address_space(3))) *lptr_t;
        const unsigned lds_w = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)sW + (unsigned)wid * 1024u);
        constexpr int kRounds = (NF * 16 * 32 + 511) / 512;   // NF * 16 rows x 32 pieces, 512 lanes per round
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            const int p = r * 512 + wid * 64 + lane;
            const int row = min(p >> 5, NF * 16 - 1), slot = p & 31;
            if (r * 512 + wid * 64 < NF * 16 * 32)   // (wave-uniform: NF * 16 * 32 is a multiple of 64)
                glds16(g.w2 + (size_t)row * g.Cout + ((slot ^ (row & 31)) << 3), lds_w + (unsigned)r * 8192u);
        }
    }
    f4 acc2[2][NF];
#pragma unroll
    for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
        for (int n = 0; n < NF; ++n) acc2[a_][n] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half == 1) __syncthreads();   // every wave is done reading the first half of X
        if ((wc >> 1) == half) {          // this wave's 64 channels belong to the half: relu(acc + bias) -> 16 bit -> LDS
            const bool relu = g.relu_out == 1;
            float4 bv[4];                 // the lane's 16 bias values, loaded once (they do not depend on the row fragment)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                bv[j] = g.bias ? *reinterpret_cast<const float4 *>(g.bias + wc * 64 + j * 16 + grp * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = wr * 128 + i * 16 + mrow;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = wc * 64 + j * 16 + grp * 4, lc = col & 127;
                    uint2 pk;   // (+ bias) -> 16 bit -> ReLU on the packed pairs (max(round(x), 0) == round(max(x, 0)))
                    pk.x = pack16x2<BF16>(acc[i][j][0] + bv[j].x, acc[i][j][1] + bv[j].y);
                    pk.y = pack16x2<BF16>(acc[i][j][2] + bv[j].z, acc[i][j][3] + bv[j].w);
                    if (relu) { pk.x = relu_reg<BF16>(pk.x); pk.y = relu_reg<BF16>(pk.y); }
                    const int chunk = (lc >> 3) ^ (row & 15);
                    *reinterpret_cast<uint2 *>(sX + row * 256 + chunk * 16 + (lc & 7) * 2) = pk;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's W2 pieces have landed (first half: under the staging above)
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 wf[NF];
#pragma unroll
            for (int n = 0; n < NF; ++n) {
                const int row = n * 16 + mrow;
                wf[n] = *reinterpret_cast<const uint4 *>(sW + row * 512 + ((((half * 4 + ks) * 4 + grp) ^ (row & 31)) << 4));
            }
#pragma unroll
            for (int a_ = 0; a_ < 2; ++a_) {
                const int row = wid * 32 + a_ * 16 + mrow;
                const uint4 xf = *reinterpret_cast<const uint4 *>(sX + row * 256 + (((ks * 4 + grp) ^ (row & 15)) << 4));
#pragma unroll
                for (int n = 0; n < NF; ++n) acc2[a_][n] = mfma<BF16>(wf[n], xf, acc2[a_][n]);
            }
        }
    }
    // C^T layout again: lane holds 4 consecutive output channels (n * 16 + grp * 4 ..) of pixel row (wid * 32 + a * 16 + mrow); the two
    // row fragments are paired so that an even-grp lane stores 8 channels of fragment 0 and an odd-grp lane 8 of fragment 1
    const bool odd = grp & 1;
    const size_t m = (size_t)m0 + wid * 32 + (odd ? 16 : 0) + mrow;
    unsigned short *dst = g.out2 + m * g.ld2 + (grp & ~1) * 4;
    const bool wide = g.ld2 % 8 == 0 && (reinterpret_cast<uintptr_t>(g.out2) & 15) == 0;
#pragma unroll
    for (int n = 0; n < NF; ++n) {
        const float4 b2 = *reinterpret_cast<const float4 *>(g.bias2 + n * 16 + grp * 4);
        uint2 pk[2];
#pragma unroll
        for (int a_ = 0; a_ < 2; ++a_) {
            pk[a_].x = pack16x2<BF16>(acc2[a_][n][0] + b2.x, acc2[a_][n][1] + b2.y);
            pk[a_].y = pack16x2<BF16>(acc2[a_][n][2] + b2.z, acc2[a_][n][3] + b2.w);
        }
        if (wide) {
            const u2v_ sx = __builtin_amdgcn_permlane16_swap(pk[0].x, pk[1].x, false, false);
            const u2v_ sy = __builtin_amdgcn_permlane16_swap(pk[0].y, pk[1].y, false, false);
            *reinterpret_cast<uint4 *>(dst + n * 16) = make_uint4(sx.x, sy.x, sx.y, sy.y);
        } else {
#pragma unroll
            for (int a_ = 0; a_ < 2; ++a_)
                *reinterpret_cast<uint2 *>(g.out2 + ((size_t)m0 + wid * 32 + a_ * 16 + mrow) * g.ld2 + grp * 4 + n * 16) = pk[a_];
        }
    }
}

// ---- the same fusion for SPLIT operands (round 3): conv3(256 -> 256) -> ReLU -> conv1(256 -> C2pad <= 96) of the Gaussian-parameter head
// at f32-class precision.  The [256 px x 256 ch] tile of the 3x3 result is f32 in the accumulators; the second GEMM needs it as (hi, lo)
// f16 operands -- 2 x 128 KiB for the whole tile -- so it runs in FOUR K-quarters of 64 channels: the two waves that own the quarter's
// channels (wave column wc = quarter) write relu(acc 2^-e + bias) as hi / lo images of [256 rows][64 halves] (2 x 32 KiB, the packed
// weights' k order inside each block of 32, chunk index XOR (row >> 1) & 7), every thread brings the quarter's slice of the packed W2
// ([C2pad rows][256 B], chunk XOR row & 15) from L2, one barrier, then all eight waves run the quarter's 2 k-steps x 3 MFMAs for their 32
// pixel rows x C2pad outputs.  72 MFMAs per wave and quarter: +2 % on the tile; the 256-channel f32 activation at 256^2 (12.9 GB per
// 24-scene step, written and read) and the HBM-bound 1x1 GEMM behind it (5.2 ms) are gone.
template <int NF>
__device__ __forceinline__ void conv_head1x1_split_epilogue256(const ConvArgs &g, f4 (&acc)[8][4], int m0, int wr, int wc, unsigned char *smem, int wid,
                                                               int lane) {
    const int mrow = lane & 15, grp = lane >> 4, tid = wid * 64 + lane;
    unsigned char *sXh = smem, *sXl = smem + 32 * 1024, *sW = smem + 64 * 1024;      // [256][128 B] | [256][128 B] | [NF*16][256 B]
    f4 acc2[2][NF];
#pragma unroll
    for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
        for (int n = 0; n < NF; ++n) acc2[a_][n] = f4{0.f, 0.f, 0.f, 0.f};
    const bool relu = g.relu_out == 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q > 0) __syncthreads();                       // every wave is done with the previous quarter's X and W2 slice
        {   // W2 slice: rows 0..NF*16-1, bytes [q*256, q*256 + 256) of each packed row (row stride = Cout 4-byte units)
            constexpr int kPieces = NF * 16 * 16;         // 16-byte pieces
#pragma unroll
            for (int r = 0; r < (kPieces + 511) / 512; ++r) {
                const int p = r * 512 + tid;
                if (p < kPieces) {
                    const int row = p >> 4, ch = p & 15;
                    const uint4 v = *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(g.w2) + (size_t)row * (g.Cout * 4) + q * 256 + ch * 16);
                    *reinterpret_cast<uint4 *>(sW + row * 256 + ((ch ^ (row & 15)) << 4)) = v;
                }
            }
        }
        if (wc == q) {                                    // this wave's 64 channels are the quarter: relu(acc * s + bias) -> (hi, lo) -> LDS
            float4 bv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                bv[j] = g.bias ? *reinterpret_cast<const float4 *>(g.bias + wc * 64 + j * 16 + grp * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = wr * 128 + i * 16 + mrow;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v0 = acc[i][j][0] * g.acc_scale + bv[j].x, v1 = acc[i][j][1] * g.acc_scale + bv[j].y;
                    float v2 = acc[i][j][2] * g.acc_scale + bv[j].z, v3 = acc[i][j][3] * g.acc_scale + bv[j].w;
                    if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                    const unsigned h0 = cvt_pk_f16(v0, v1), h1 = cvt_pk_f16(v2, v3);
                    float r0, r1, r2, r3;
                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h0), "v"(v0));
                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h0), "v"(v1));
                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(h1), "v"(v2));
                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(h1), "v"(v3));
                    const unsigned l0 = cvt_pk_f16(r0, r1), l1 = cvt_pk_f16(r2, r3);
                    // channels j*16 + grp*4 + 0..3 of the quarter: block j >> 1, chunk grp, first / second half of the chunk (packed k order)
                    const int chunk = ((j >> 1) * 4 + grp) ^ ((row >> 1) & 7);
                    const int off = row * 128 + chunk * 16 + (j & 1) * 8;
                    *reinterpret_cast<uint2 *>(sXh + off) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2 *>(sXl + off) = make_uint2(l0, l1);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 xh[2], xl[2];
#pragma unroll
            for (int a_ = 0; a_ < 2; ++a_) {
                const int row = wid * 32 + a_ * 16 + mrow;
                const int o = row * 128 + (((ks * 4 + grp) ^ ((row >> 1) & 7)) << 4);
                xh[a_] = *reinterpret_cast<const uint4 *>(sXh + o); xl[a_] = *reinterpret_cast<const uint4 *>(sXl + o);
            }
#pragma unroll
            for (int n = 0; n < NF; ++n) {      // (W2 fragments one output block at a time: all NF of them beside the 128 accumulators spill)
                const int row = n * 16 + mrow;
                const uint4 wh = *reinterpret_cast<const uint4 *>(sW + row * 256 + (((ks * 8 + grp) ^ (row & 15)) << 4));
                const uint4 wl = *reinterpret_cast<const uint4 *>(sW + row * 256 + (((ks * 8 + 4 + grp) ^ (row & 15)) << 4));
#pragma unroll
                for (int a_ = 0; a_ < 2; ++a_) acc2[a_][n] = mma2<kDtSplit>(wh, wl, xh[a_], xl[a_], acc2[a_][n]);
            }
        }
    }
    // C^T layout: lane holds output channels n*16 + grp*4 + 0..3 of pixel row wid*32 + a*16 + mrow -> 16-byte f32 stores
    float *out2 = reinterpret_cast<float *>(g.out2);
#pragma unroll
    for (int n = 0; n < NF; ++n) {
        const float4 b2 = *reinterpret_cast<const float4 *>(g.bias2 + n * 16 + grp * 4);
#pragma unroll
        for (int a_ = 0; a_ < 2; ++a_) {
            const size_t m = (size_t)m0 + wid * 32 + a_ * 16 + mrow;
            *reinterpret_cast<float4 *>(out2 + m * g.ld2 + n * 16 + grp * 4) =
                make_float4(acc2[a_][n][0] * g.acc_scale2 + b2.x, acc2[a_][n][1] * g.acc_scale2 + b2.y, acc2[a_][n][2] * g.acc_scale2 + b2.z,
                            acc2[a_][n][3] * g.acc_scale2 + b2.w);
        }
    }
}

// ---- 256 x 256 x 64 implicit-GEMM variant on the phase-interleaved main loop of gemm256.h (Cout tile 256, one tap x 64
// input channels per K-tile; needs Cin = 64 << cshift).  Only the staging differs from the GEMM: per staged row the source
// is the tap-shifted pixel's 128-byte channel slice, or the zero page outside the image. ----
struct ConvStager256 {
    // Per staged A row (this lane's 16 bytes of it): address of the CENTRE tap's channel slice and a 9-bit mask of the taps that fall
    // inside the image, both fixed for the whole K loop; a stage call then needs one scalar offset (tap shift x Cin + channel block),
    // one bit test and one select per row instead of re-deriving (y, x), the bounds and the 64-bit address from packed coordinates
    // (the implicit-GEMM loop had 4x the VALU instructions of the plain GEMM loop between the same 64 MFMAs per K-tile).
    const unsigned short *pa[2][2];  // [A_h][round]
    const unsigned short *pz[2];     // [round] this lane's 16 bytes of the zero page
    unsigned vmask[2][2];            // [A_h][round] bit t: tap t is inside the image (0 for rows past M)
    const unsigned short *pw[2][2];  // [B_h][round]
    int Cin, Win, kpt, kinv;         // kpt = K-tiles per tap (Cin / 64, any value <= 64: round 5 -- was a power of two), kinv = ceil(2^16 / kpt)
    int chmajor;                     // K-tile order: 0 (tap, channel block), 1 (channel block, tap) -- late round 5, see conv_korder()
    __device__ __forceinline__ void stage(int u, int kt, unsigned lds) const {
        if (u < 2) {
            int tap, kc;
            if (chmajor) { kc = (kt * 7282) >> 16; tap = kt - kc * 9; }     // kt / 9, exact for kt < 576
            else { tap = (kt * kinv) >> 16; kc = kt - tap * kpt; }          // kt / kpt, exact for kt < 9 * kpt (checked for every kpt <= 64)
            const int ty = (tap * 11) >> 5;                      // tap / 3 for tap in 0..8
            const int dy = ty - 1, dx = tap - ty * 3 - 1;
            const long long off = (long long)(dy * Win + dx) * Cin + kc * 64;   // wave-uniform
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bool ok = (vmask[u][j] >> tap) & 1u;
                glds16(ok ? pa[u][j] + off : pz[j], lds + j * 1024u);
            }
        } else {
            int wk = kt;
            if (chmajor) { const int kc = (kt * 7282) >> 16; wk = (kt - kc * 9) * kpt + kc; }     // the weight rows stay [tap][channel]
            glds16(pw[u - 2][0] + wk * 64, lds);
            glds16(pw[u - 2][1] + wk * 64, lds + 1024u);
        }
    }
};

template <int BF16, bool RELU_IN, int FUSE_NF = 0, bool APACK = false>
__global__ void __launch_bounds__(512, 1) conv3x3_256_kernel(const ConvArgs g, const int kpt) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[kLdsBytes256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int HW = g.H * g.W;
    const int M = g.Nimg * HW;
    const int K = 9 * g.Cin;
    const int tiles_n = (g.Cout + 255) / 256;
    const int tiles_m = (M + 255) / 256;
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * 256, n0 = tn * 256;

    ConvStager256 st;
    st.Cin = g.Cin; st.Win = g.Win; st.kpt = kpt; st.kinv = (65536 + kpt - 1) / kpt; st.chmajor = g.korder;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = unit_row256(wid, j, lane);
        const int chunk = unit_src_chunk256(q, lane);
        st.pz[j] = vs_zero_page + chunk * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = m0 + unit_a_tile_row256(q, h);
            const int p = m < M ? m : 0;
            const int nimg = p / HW, rem = p - nimg * HW;
            const int y = (rem / g.W) * g.stride, x = (rem % g.W) * g.stride;
            st.pa[h][j] = g.in + ((size_t)((nimg * g.Hin + y) * g.Win + x) * g.Cin + chunk * 8);
            unsigned vm = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                if (m < M && (unsigned)yy < (unsigned)g.Hin && (unsigned)xx < (unsigned)g.Win) vm |= 1u << t;
            }
            st.vmask[h][j] = vm;
            const int rw_ = min(n0 + unit_b_tile_row256(q, h), g.Cout - 1);
            st.pw[h][j] = g.w + (size_t)rw_ * K + chunk * 8;
        }
    }
    f4 acc[8][4];
    if constexpr (BF16 == kDtSplit) mainloop256_split<RELU_IN, ConvStager256, APACK>(st, K / 64, acc, smem, lane, wid);
    else mainloop256<BF16, RELU_IN>(st, K / 64, acc, smem, lane, wid);
    if constexpr (FUSE_NF > 0 && BF16 == kDtSplit) conv_head1x1_split_epilogue256<FUSE_NF>(g, acc, m0, wr, wc, smem, wid, lane);
    else if constexpr (FUSE_NF > 0) conv_head1x1_epilogue256<BF16, FUSE_NF>(g, acc, m0, wr, wc, smem, wid, lane);
    else conv_epilogue<BF16, 8>(g, acc, m0 + wr * 128, n0 + wc * 64, M, smem, wid, lane);
}

// ---- 256 x 128 tile on the split main loop (gemm256.h, mainloop256x128_split): the Cout = 128 layers of the pts3d head (conv 256 -> 128
// at 128^2, conv 128 -> 128 at 256^2 with the fused 128 -> 3 head).  The 4-wave pair-step kernel ran them at 0.30 of the split ceiling with
// ONE workgroup of 4 waves per CU (96 KiB of LDS): one wave per SIMD cannot overlap its own conversion VALU with its MFMAs. ----
struct ConvStager256x128 {
    const unsigned short *pa[2][2];  // [A_h][round]
    const unsigned short *pz[2];
    unsigned vmask[2][2];
    const unsigned short *pw[2];     // [round] the B unit: 128 weight rows
    int Cin, Win, kpt, kinv;         // kpt = K-tiles per tap (Cin / 64, any value <= 64: round 5 -- was a power of two), kinv = ceil(2^16 / kpt)
    int chmajor;                     // K-tile order: 0 (tap, channel block), 1 (channel block, tap) -- late round 5, see conv_korder()
    __device__ __forceinline__ void stage(int u, int kt, unsigned lds) const {
        if (u < 2) {
            int tap, kc;
            if (chmajor) { kc = (kt * 7282) >> 16; tap = kt - kc * 9; }     // kt / 9, exact for kt < 576
            else { tap = (kt * kinv) >> 16; kc = kt - tap * kpt; }          // kt / kpt, exact for kt < 9 * kpt (checked for every kpt <= 64)
            const int ty = (tap * 11) >> 5;
            const int dy = ty - 1, dx = tap - ty * 3 - 1;
            const long long off = (long long)(dy * Win + dx) * Cin + kc * 64;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bool ok = (vmask[u][j] >> tap) & 1u;
                glds16(ok ? pa[u][j] + off : pz[j], lds + j * 1024u);
            }
        } else {
            int wk = kt;
            if (chmajor) { const int kc = (kt * 7282) >> 16; wk = (kt - kc * 9) * kpt + kc; }
            glds16(pw[0] + wk * 64, lds);
            glds16(pw[1] + wk * 64, lds + 1024u);
        }
    }
};

template <int BF16, int MI>
__device__ __forceinline__ void conv_head_dot_epilogue(const ConvArgs &g, f4 (&acc)[MI][4], int m0, int wr, int wc, float *red, int lane);

template <bool RELU_IN, bool FUSE_DOT, bool A_PACKED = false>
__global__ void __launch_bounds__(512, 1) conv3x3_256x128_split_kernel(const ConvArgs g, const int kpt) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * 3 * kUnitBytes256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wid >> 2) * 2 + ((wid & 3) >> 1), wc = wid & 1;
    const int HW = g.H * g.W;
    const int M = g.Nimg * HW;
    const int K = 9 * g.Cin;
    const int tiles_n = (g.Cout + 127) / 128;
    const int nwg = ((M + 255) / 256) * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * 256, n0 = tn * 128;
    ConvStager256x128 st;
    st.Cin = g.Cin; st.Win = g.Win; st.kpt = kpt; st.kinv = (65536 + kpt - 1) / kpt; st.chmajor = g.korder;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = unit_row256(wid, j, lane);                 // unit row 0..127 this lane stages
        const int chunk = unit_src_chunk256(q, lane);
        st.pz[j] = vs_zero_page + chunk * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = m0 + (q >> 5) * 64 + h * 32 + (q & 31);   // A_h: tile rows {wm*64 + h*32 + 0..31}
            const int p = m < M ? m : 0;
            const int nimg = p / HW, rem = p - nimg * HW;
            const int y = (rem / g.W) * g.stride, x = (rem % g.W) * g.stride;
            st.pa[h][j] = g.in + ((size_t)((nimg * g.Hin + y) * g.Win + x) * g.Cin + chunk * 8);
            unsigned vm = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                if (m < M && (unsigned)yy < (unsigned)g.Hin && (unsigned)xx < (unsigned)g.Win) vm |= 1u << t;
            }
            st.vmask[h][j] = vm;
        }
        const int rw_ = min(n0 + q, g.Cout - 1);
        st.pw[j] = g.w + (size_t)rw_ * K + chunk * 8;
    }
    f4 acc[4][4];
    mainloop256x128_split<RELU_IN, ConvStager256x128, A_PACKED>(st, K / 64, acc, smem, lane, wid);
    if constexpr (FUSE_DOT) conv_head_dot_epilogue<kDtSplit, 4>(g, acc, m0, wm, wc, reinterpret_cast<float *>(smem), lane);
    else conv_epilogue<kDtSplit, 4>(g, acc, m0 + wm * 64, n0 + wc * 64, M, smem, wid, lane);
}

// ---- fused 1x1 head with <= 4 outputs behind a 3x3 convolution whose Cout fits one 128-column tile (the pts3d head: conv3(128->128)
// -> ReLU -> conv1(128->3), dpt_block.py:316-333): per pixel three 128-long dot products on the VALU.  A lane owns 16 channels of a pixel
// row per fragment; partial sums are reduced over the 4 lane groups of the wave (xor 16 / 32) and over the two waves that share a row
// (wc = 0 / 1) through LDS; the activation is rounded to 16 bit first, as the unfused path stores it.  Output [pixels, ld2 >= 4] 16-bit.
template <int BF16, int MI>
__device__ __forceinline__ void conv_head_dot_epilogue(const ConvArgs &g, f4 (&acc)[MI][4], int m0, int wr, int wc, float *red, int lane) {
    // f32 activations (split operand class): w2 is an f32 [C2, Cout] array, nothing is rounded to 16 bit, the output row is 4 floats
    constexpr bool F32 = is_f32io(BF16);
    constexpr int D16 = F32 ? 0 : BF16;
    const int mrow = lane & 15, grp = lane >> 4;
    if constexpr (BF16 == kDtSplit) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] *= g.acc_scale;
    }
    float w2v[4][4][4], bv[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = wc * 64 + j * 16 + grp * 4 + r;
            bv[j][r] = g.bias ? g.bias[col] : 0.0f;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                if constexpr (F32) w2v[o][j][r] = o < g.C2 ? reinterpret_cast<const float *>(g.w2)[(size_t)o * g.Cout + col] : 0.0f;
                else w2v[o][j][r] = o < g.C2 ? from16<D16>(g.w2[(size_t)o * g.Cout + col]) : 0.0f;
            }
        }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        float p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[i][j][r] + bv[j][r];
                if (g.relu_out == 1) v = fmaxf(v, 0.0f);
                if constexpr (!F32) v = from16<D16>(to16<D16>(v));
#pragma unroll
                for (int o = 0; o < 4; ++o) p[o] = fmaf(v, w2v[o][j][r], p[o]);
            }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            p[o] += __shfl_xor(p[o], 16, 64);
            p[o] += __shfl_xor(p[o], 32, 64);
        }
        const int row = wr * (16 * MI) + i * 16 + mrow;
        if (wc == 1 && grp == 0) *reinterpret_cast<float4 *>(red + row * 4) = make_float4(p[0], p[1], p[2], p[3]);
        acc[i][0] = f4{p[0], p[1], p[2], p[3]};
    }
    __syncthreads();
    if (wc == 0 && grp == 0) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = wr * (16 * MI) + i * 16 + mrow;
            const float4 q = *reinterpret_cast<const float4 *>(red + row * 4);
            const float o0 = acc[i][0][0] + q.x + g.bias2[0], o1 = acc[i][0][1] + q.y + g.bias2[1];
            const float o2 = acc[i][0][2] + q.z + g.bias2[2], o3 = acc[i][0][3] + q.w + g.bias2[3];
            if constexpr (F32) {
                *reinterpret_cast<float4 *>(reinterpret_cast<float *>(g.out2) + ((size_t)m0 + row) * g.ld2) = make_float4(o0, o1, o2, o3);
            } else {
                uint2 pk;
                pk.x = pack16x2<D16>(o0, o1);
                pk.y = pack16x2<D16>(o2, o3);
                *reinterpret_cast<uint2 *>(g.out2 + ((size_t)m0 + row) * g.ld2) = pk;
            }
        }
    }
}

template <int BF16, int MI, bool FUSE_DOT = false>
__global__ void __launch_bounds__(256, (MI == 8 || BF16 == kDtSplit) ? 2 : 3) conv3x3_kernel(const ConvArgs g) {
    constexpr int BM = 32 * MI;
    constexpr bool SPLIT = BF16 == kDtSplit;
    constexpr int NS = SPLIT ? 4 : 3;
    constexpr int GL = MI / 2 + 2;
    // same 3-stage LDS ring / counted-vmcnt pipeline as gemm.hip (32-channel stages: one tap x 32 input channels)
    __shared__ __attribute__((aligned(1024))) unsigned short smem[NS * (BM + BN) * 32];
    unsigned short *const sA = smem, *const sW = smem + NS * BM * 32;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int HW = g.H * g.W;
    const int M = g.Nimg * HW;
    const int K = 9 * g.Cin;
    const int tiles_n = (g.Cout + BN - 1) / BN;
    const int tiles_m = (M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // n-fastest tile order: the (few) Cout tiles of one pixel tile run back to back and share the A panel in L2
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int rho = lane >> 2, gchunk = (lane & 3) ^ swz4(rho);
    // per staged A row (piece = 16 rows x 64 B): linear input pixel index and packed (y, x); rows beyond M are invalid
    int pix[MI / 2];
    unsigned yx[MI / 2];
#pragma unroll
    for (int i = 0; i < MI / 2; ++i) {
        const int m = m0 + (wid * (MI / 2) + i) * 16 + rho;
        const int p = m < M ? m : 0;
        const int nimg = p / HW, rem = p - nimg * HW;
        const int y = (rem / g.W) * g.stride, x = (rem % g.W) * g.stride;  // centre tap in INPUT coordinates
        pix[i] = (nimg * g.Hin + y) * g.Win + x;
        yx[i] = m < M ? ((unsigned)y << 16) | (unsigned)x : 0x7fff0000u;
    }
    const unsigned short *pw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rw_ = min(n0 + (wid * 2 + i) * 16 + rho, g.Cout - 1);
        pw[i] = g.w + (size_t)rw_ * K + gchunk * 8;
    }
    typedef void __attribute__((address_space(3))) *lptr_t;
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    const unsigned ldsA = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(wid * (MI / 2)) * 1024u);
    const unsigned ldsW = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(NS * BM * 32 * 2) + (unsigned)(wid * 2) * 1024u);

    f4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fg = lane >> 4;
    const int st_per_tap = g.Cin / 32;
    const int nk = K / 32;
    // stage kt = (tap, 32-channel slice); issue order is sequential so (tap, kc) are tracked incrementally
    int s_tap = 0, s_kc = 0;
#define VS_STAGE(slot_)                                                                                              \
    {                                                                                                                \
        const int dy = s_tap / 3 - 1, dx = s_tap - (s_tap / 3) * 3 - 1;                                              \
        const int shift = dy * g.Win + dx;                                                                           \
        const int cin0 = s_kc * 32;                                                                                  \
        _Pragma("unroll") for (int i = 0; i < MI / 2; ++i) {                                                         \
            const int y = (int)(yx[i] >> 16) + dy, x = (int)(yx[i] & 0xffffu) + dx;                                  \
            const bool ok = (unsigned)y < (unsigned)g.Hin && (unsigned)x < (unsigned)g.Win;                          \
            const unsigned short *src = ok ? g.in + ((size_t)(pix[i] + shift) * g.Cin + cin0 + gchunk * 8)          \
                                           : vs_zero_page + gchunk * 8;                                              \
            glds16(src, ldsA + (unsigned)((slot_) * (BM * 32 * 2) + i * 1024));                                      \
        }                                                                                                            \
        const int kglob = (s_tap * st_per_tap + s_kc) * 32;                                                          \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                \
            glds16(pw[i] + kglob, ldsW + (unsigned)((slot_) * (BN * 32 * 2) + i * 1024));                            \
        if (++s_kc == st_per_tap) { s_kc = 0; ++s_tap; }                                                             \
    }
#define VS_STEP(kt_, slot_, nslot_)                                                                         \
    {                                                                                                       \
        if ((kt_) + 1 < nk) {                                                                               \
            if constexpr (GL == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                         \
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                           \
        } else {                                                                                            \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                \
        }                                                                                                   \
        __builtin_amdgcn_s_barrier();                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        if ((kt_) + 2 < nk) VS_STAGE(nslot_)                                                                \
        const unsigned short *cA = sA + (slot_) * (BM * 32), *cW = sW + (slot_) * (BN * 32);                \
        uint4 fb[4];                                                                                        \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                     \
            const int rb_ = wc * 64 + j * 16 + frow;                                                        \
            fb[j] = *reinterpret_cast<const uint4 *>(&cW[rb_ * 32 + ((fg ^ swz4(rb_)) << 3)]);              \
        }                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < MI; ++i) {                                                    \
            const int ra_ = wr * (16 * MI) + i * 16 + frow;                                                 \
            uint4 fa = *reinterpret_cast<const uint4 *>(&cA[ra_ * 32 + ((fg ^ swz4(ra_)) << 3)]);           \
            if (g.relu_in) { fa.x = relu_reg<BF16>(fa.x); fa.y = relu_reg<BF16>(fa.y); fa.z = relu_reg<BF16>(fa.z); fa.w = relu_reg<BF16>(fa.w); } \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = mfma<BF16>(fb[j], fa, acc[i][j]);     \
        }                                                                                                   \
    }
    if constexpr (SPLIT) {
        // split operands: stages in pairs = one 128-byte block of 32 input channels (f32 activations: floats 0..15 / 16..31; packed
        // weights: hi / lo halves), two pairs of ring slots, one barrier per pair (as gemm_kernel)
        const int np = nk >> 1;
        VS_STAGE(0)
        VS_STAGE(1)
        for (int p = 0; p < np; ++p) {
            const int s0 = (p & 1) * 2;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (p + 1 < np) {
                VS_STAGE(s0 ^ 2)
                VS_STAGE((s0 ^ 2) + 1)
            }
            const unsigned short *cA0 = sA + s0 * (BM * 32), *cA1 = cA0 + BM * 32, *cW0 = sW + s0 * (BN * 32), *cW1 = cW0 + BN * 32;
            uint4 fbh[4], fbl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rb_ = wc * 64 + j * 16 + frow;
                fbh[j] = *reinterpret_cast<const uint4 *>(&cW0[rb_ * 32 + ((fg ^ swz4(rb_)) << 3)]);
                fbl[j] = *reinterpret_cast<const uint4 *>(&cW1[rb_ * 32 + ((fg ^ swz4(rb_)) << 3)]);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int ra_ = wr * (16 * MI) + i * 16 + frow;
                uint4 fa0 = *reinterpret_cast<const uint4 *>(&cA0[ra_ * 32 + ((fg ^ swz4(ra_)) << 3)]);
                uint4 fa1 = *reinterpret_cast<const uint4 *>(&cA1[ra_ * 32 + ((fg ^ swz4(ra_)) << 3)]);
                if (g.relu_in) {
                    fa0.x = relu_reg<BF16>(fa0.x); fa0.y = relu_reg<BF16>(fa0.y); fa0.z = relu_reg<BF16>(fa0.z); fa0.w = relu_reg<BF16>(fa0.w);
                    fa1.x = relu_reg<BF16>(fa1.x); fa1.y = relu_reg<BF16>(fa1.y); fa1.z = relu_reg<BF16>(fa1.z); fa1.w = relu_reg<BF16>(fa1.w);
                }
                split8(fa0, fa1);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mma2<BF16>(fbh[j], fbl[j], fa0, fa1, acc[i][j]);
            }
        }
    } else {
        VS_STAGE(0)
        if (nk > 1) VS_STAGE(1)
        for (int kt = 0; kt < nk; kt += 3) {
            VS_STEP(kt, 0, 2)
            if (kt + 1 < nk) VS_STEP(kt + 1, 1, 0)
            if (kt + 2 < nk) VS_STEP(kt + 2, 2, 1)
        }
    }
#undef VS_STEP
#undef VS_STAGE
    __syncthreads();

    if constexpr (FUSE_DOT) conv_head_dot_epilogue<BF16, MI>(g, acc, m0, wr, wc, reinterpret_cast<float *>(smem), lane);
    else conv_epilogue<BF16, MI>(g, acc, m0 + wr * (16 * MI), n0 + wc * 64, M, smem, wid, lane);
}

// ---- bilinear x2, align_corners=True, NHWC 16-bit; optional fused "+ add" (gs head: up2(trunk) + image features) ----
// grid = (ceil(Wo * C/8 / 256), Nimg * Ho): the row quantities (image, source rows, ly) are block-uniform and the per-thread index math
// is 32-bit (one thread per (output pixel, 8 channels); the flat 64-bit div/mod chain of the first version was ~200 VALU per 16 bytes
// of output -- the kernel ran VALU-bound at 3.6 TB/s).  Source index = dst * (in-1)/(out-1) with the ratio formed once in f32, as
// PyTorch's area_pixel_compute_source_index does for align_corners=True (and as the backward kernel below does).
// One thread per 2x2 OUTPUT block and 8 channels: the four outputs draw on a 3x3 source neighbourhood (consecutive outputs move the
// source position by < 1/2, so their floor indices differ by 0 or 1), i.e. 9 sixteen-byte loads per 4 outputs instead of 16 -- one
// output per thread ran at 4.3 TB/s with the vector-memory pipe, not HBM, as the limit (5 memory instructions per 16 bytes written).
// Every output is still lerp_y(lerp_x(.)) of its own four neighbours with its own weights, in f32: results are those of the
// one-output-per-thread form.
template <bool BF16>
__global__ void __launch_bounds__(256)
upsample2x_kernel(const unsigned short *__restrict__ in, const unsigned short *__restrict__ add, unsigned short *__restrict__ out,
                  int Nimg, int H, int W, int C, int relu_add) {
    const int Ho = 2 * H, Wo = 2 * W, c8 = C >> 3;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= W * c8) return;
    const int xb = e / c8, cc = e - xb * c8;
    const int rowb = blockIdx.z * gridDim.y + blockIdx.y;   // (image, output row pair); grid.y is capped at 32768, larger batches spill into grid.z
    if (rowb >= Nimg * H) return;
    const int n = rowb / H, yb = rowb - n * H;
    const float ry = (float)(H - 1) / (float)(Ho - 1), rx = (float)(W - 1) / (float)(Wo - 1);
    // block-uniform row quantities
    const float sya = (float)(2 * yb) * ry, syb = (float)(2 * yb + 1) * ry;
    const int y0a = min((int)sya, H - 1), y0b = min((int)syb, H - 1);
    const float lya = sya - (float)y0a, lyb = syb - (float)y0b;
    const int dyb = y0b - y0a;                               // 0 or 1
    const float sxa = (float)(2 * xb) * rx, sxb = (float)(2 * xb + 1) * rx;
    const int x0a = min((int)sxa, W - 1), x0b = min((int)sxb, W - 1);
    const float lxa = sxa - (float)x0a, lxb = sxb - (float)x0b;
    const bool dxb = x0b != x0a;                             // per lane
    const int xs1 = min(x0a + 1, W - 1), xs2 = min(x0a + 2, W - 1);
    const size_t base = (size_t)n * H * W;
    float ha[3][8], hb[3][8];   // horizontal lerps of the three source rows at the two output columns
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const size_t rowoff = (base + (size_t)min(y0a + r, H - 1) * W) * C + cc * 8;
        const uint4 s0 = *reinterpret_cast<const uint4 *>(in + rowoff + (size_t)x0a * C);
        const uint4 s1 = *reinterpret_cast<const uint4 *>(in + rowoff + (size_t)xs1 * C);
        const uint4 s2 = *reinterpret_cast<const uint4 *>(in + rowoff + (size_t)xs2 * C);
        const unsigned w0[4] = {s0.x, s0.y, s0.z, s0.w}, w1[4] = {s1.x, s1.y, s1.z, s1.w}, w2[4] = {s2.x, s2.y, s2.z, s2.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const float f0 = from16<BF16>((unsigned short)(hh ? w0[k] >> 16 : w0[k] & 0xffff));
                const float f1 = from16<BF16>((unsigned short)(hh ? w1[k] >> 16 : w1[k] & 0xffff));
                const float f2 = from16<BF16>((unsigned short)(hh ? w2[k] >> 16 : w2[k] & 0xffff));
                ha[r][2 * k + hh] = f0 * (1.f - lxa) + f1 * lxa;
                const float g0 = dxb ? f1 : f0, g1 = dxb ? f2 : f1;
                hb[r][2 * k + hh] = g0 * (1.f - lxb) + g1 * lxb;
            }
        }
    }
    const size_t oa = (((size_t)n * Ho + 2 * yb) * Wo + 2 * xb) * C + cc * 8, ob = oa + (size_t)Wo * C;
    auto emit = [&](const float (&top)[8], const float (&bot)[8], float ly, size_t o) {
        uint4 av = make_uint4(0, 0, 0, 0);
        if (add) {
            av = *reinterpret_cast<const uint4 *>(add + o);
            if (relu_add) { av.x = relu2(av.x); av.y = relu2(av.y); av.z = relu2(av.z); av.w = relu2(av.w); }
        }
        const unsigned aa[4] = {av.x, av.y, av.z, av.w};
        unsigned r4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float lo = top[2 * k] * (1.f - ly) + bot[2 * k] * ly + (add ? from16<BF16>((unsigned short)(aa[k] & 0xffff)) : 0.f);
            const float hi = top[2 * k + 1] * (1.f - ly) + bot[2 * k + 1] * ly + (add ? from16<BF16>((unsigned short)(aa[k] >> 16)) : 0.f);
            r4[k] = pack16x2<BF16>(lo, hi);
        }
        *reinterpret_cast<uint4 *>(out + o) = make_uint4(r4[0], r4[1], r4[2], r4[3]);
    };
    emit(ha[0], ha[1], lya, oa);
    emit(hb[0], hb[1], lya, oa + C);
    if (dyb) {   // block-uniform
        emit(ha[1], ha[2], lyb, ob);
        emit(hb[1], hb[2], lyb, ob + C);
    } else {
        emit(ha[0], ha[1], lyb, ob);
        emit(hb[0], hb[1], lyb, ob + C);
    }
}

// f32 variant (reference-precision path): one thread per (output pixel, 4 channels), same grid and index math
__global__ void __launch_bounds__(256)
upsample2x_f32_kernel(const float *__restrict__ in, const float *__restrict__ add, float *__restrict__ out, int Nimg, int H, int W, int C,
                      int relu_add) {
    const int Ho = 2 * H, Wo = 2 * W, c4 = C >> 2;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= Wo * c4) return;
    const int xo = e / c4, cc = e - xo * c4;
    const int row = blockIdx.z * gridDim.y + blockIdx.y;   // (grid.y is capped at 32768 rows; larger batches spill into grid.z)
    if (row >= Nimg * Ho) return;
    const int n = row / Ho, yo = row - n * Ho;
    const float ry = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, rx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const float sy = (float)yo * ry, sx = (float)xo * rx;
    const int y0 = min((int)sy, H - 1), x0 = min((int)sx, W - 1);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const size_t base = (size_t)n * H * W;
    const float4 v00 = *reinterpret_cast<const float4 *>(in + ((base + (size_t)y0 * W + x0) * C + cc * 4));
    const float4 v01 = *reinterpret_cast<const float4 *>(in + ((base + (size_t)y0 * W + x1) * C + cc * 4));
    const float4 v10 = *reinterpret_cast<const float4 *>(in + ((base + (size_t)y1 * W + x0) * C + cc * 4));
    const float4 v11 = *reinterpret_cast<const float4 *>(in + ((base + (size_t)y1 * W + x1) * C + cc * 4));
    const size_t o = (((size_t)n * Ho + yo) * Wo + xo) * C + cc * 4;
    float4 av = make_float4(0.f, 0.f, 0.f, 0.f);
    if (add) {
        av = *reinterpret_cast<const float4 *>(add + o);
        if (relu_add) { av.x = fmaxf(av.x, 0.f); av.y = fmaxf(av.y, 0.f); av.z = fmaxf(av.z, 0.f); av.w = fmaxf(av.w, 0.f); }
    }
    auto mix = [&](float a, float b, float c, float d, float e) {
        const float t = a * (1.f - lx) + b * lx, bt = c * (1.f - lx) + d * lx;
        return t * (1.f - ly) + bt * ly + e;
    };
    *reinterpret_cast<float4 *>(out + o) = make_float4(mix(v00.x, v01.x, v10.x, v11.x, av.x), mix(v00.y, v01.y, v10.y, v11.y, av.y),
                                                       mix(v00.z, v01.z, v10.z, v11.z, av.z), mix(v00.w, v01.w, v10.w, v11.w, av.w));
}

// f32, one thread per 2x2 OUTPUT block and 4 channels (the scheme of upsample2x_kernel: 9 sixteen-byte loads per 4 outputs instead of 16;
// every output is lerp_y(lerp_x(.)) of its own four neighbours with its own weights: the results of the one-output-per-thread form)
__global__ void __launch_bounds__(256)
upsample2x_f32_block_kernel(const float *__restrict__ in, const float *__restrict__ add, float *__restrict__ out, int Nimg, int H, int W, int C,
                            int relu_add, int pack_out) {
    const int Ho = 2 * H, Wo = 2 * W, c4 = C >> 2;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= W * c4) return;
    const int xb = e / c4, cc = e - xb * c4;
    const int rowb = blockIdx.z * gridDim.y + blockIdx.y;
    if (rowb >= Nimg * H) return;
    const int n = rowb / H, yb = rowb - n * H;
    const float ry = (float)(H - 1) / (float)(Ho - 1), rx = (float)(W - 1) / (float)(Wo - 1);
    const float sya = (float)(2 * yb) * ry, syb = (float)(2 * yb + 1) * ry;
    const int y0a = min((int)sya, H - 1), y0b = min((int)syb, H - 1);
    const float lya = sya - (float)y0a, lyb = syb - (float)y0b;
    const int dyb = y0b - y0a;
    const float sxa = (float)(2 * xb) * rx, sxb = (float)(2 * xb + 1) * rx;
    const int x0a = min((int)sxa, W - 1), x0b = min((int)sxb, W - 1);
    const float lxa = sxa - (float)x0a, lxb = sxb - (float)x0b;
    const bool dxb = x0b != x0a;
    const int xs1 = min(x0a + 1, W - 1), xs2 = min(x0a + 2, W - 1);
    const size_t base = (size_t)n * H * W;
    float ha[3][4], hb[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const size_t rowoff = (base + (size_t)min(y0a + r, H - 1) * W) * C + cc * 4;
        const float4 s0 = *reinterpret_cast<const float4 *>(in + rowoff + (size_t)x0a * C);
        const float4 s1 = *reinterpret_cast<const float4 *>(in + rowoff + (size_t)xs1 * C);
        const float4 s2 = *reinterpret_cast<const float4 *>(in + rowoff + (size_t)xs2 * C);
        const float f0[4] = {s0.x, s0.y, s0.z, s0.w}, f1[4] = {s1.x, s1.y, s1.z, s1.w}, f2[4] = {s2.x, s2.y, s2.z, s2.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ha[r][k] = f0[k] * (1.f - lxa) + f1[k] * lxa;
            const float g0 = dxb ? f1[k] : f0[k], g1 = dxb ? f2[k] : f1[k];
            hb[r][k] = g0 * (1.f - lxb) + g1 * lxb;
        }
    }
    const size_t oa = (((size_t)n * Ho + 2 * yb) * Wo + 2 * xb) * C + cc * 4, ob = oa + (size_t)Wo * C;
    auto emit = [&](const float (&top)[4], const float (&bot)[4], float ly, size_t o) {
        float4 av = make_float4(0.f, 0.f, 0.f, 0.f);
        if (add) {
            av = *reinterpret_cast<const float4 *>(add + o);
            if (relu_add) { av.x = fmaxf(av.x, 0.f); av.y = fmaxf(av.y, 0.f); av.z = fmaxf(av.z, 0.f); av.w = fmaxf(av.w, 0.f); }
        }
        const float4 r = make_float4(top[0] * (1.f - ly) + bot[0] * ly + av.x, top[1] * (1.f - ly) + bot[1] * ly + av.y,
                                     top[2] * (1.f - ly) + bot[2] * ly + av.z, top[3] * (1.f - ly) + bot[3] * ly + av.w);
        if (pack_out) store_split4(out + (o - cc * 4), cc * 4, r.x, r.y, r.z, r.w);   // packed (hi, lo) pixel row: the next conv's operand as it is
        else *reinterpret_cast<float4 *>(out + o) = r;
    };
    emit(ha[0], ha[1], lya, oa);
    emit(hb[0], hb[1], lya, oa + C);
    if (dyb) {
        emit(ha[1], ha[2], lyb, ob);
        emit(hb[1], hb[2], lyb, ob + C);
    } else {
        emit(ha[0], ha[1], lyb, ob);
        emit(hb[0], hb[1], lyb, ob + C);
    }
}

// ---- backward of the bilinear x2 (align_corners=True): gather form, one thread per (input pixel, 8 channels).  Input row y
// receives from the output rows whose two source rows include y: sy = yo * (H-1)/(2H-1) in [y-1, y+1), i.e. yo within
// [2y-3, 2y+3]; the weight of an output row for y is (y0 == y)(1 - ly) + (y1 == y) ly, which is what the forward used. ----
template <bool BF16>
__global__ void __launch_bounds__(256)
upsample2x_backward_kernel(const unsigned short *__restrict__ dout, unsigned short *__restrict__ din, int Nimg, int H, int W, int C) {
    const int Ho = 2 * H, Wo = 2 * W, c8 = C >> 3;
    const int e = blockIdx.x * 256 + threadIdx.x;          // grid = (ceil(W * C/8 / 256), rows of din), as the forward
    if (e >= W * c8) return;
    const int x = e / c8, cc = e - x * c8;
    const int row = blockIdx.z * gridDim.y + blockIdx.y;
    if (row >= Nimg * H) return;
    const int n = row / H, y = row - n * H;
    const float ry = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, rx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
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
            const uint4 v = *reinterpret_cast<const uint4 *>(dout + ((((size_t)n * Ho + yo) * Wo + xo) * C + cc * 8));
            const unsigned vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc[2 * k] += wgt * from16<BF16>((unsigned short)(vv[k] & 0xffffu));
                acc[2 * k + 1] += wgt * from16<BF16>((unsigned short)(vv[k] >> 16));
            }
        }
    }
    unsigned r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = pack16x2<BF16>(acc[2 * k], acc[2 * k + 1]);
    *reinterpret_cast<uint4 *>(din + ((((size_t)n * H + y) * W + x) * C + cc * 8)) = make_uint4(r[0], r[1], r[2], r[3]);
}

}  // namespace

static int conv3x3_entry(const void *in, const void *w, const float *bias, const void *residual, void *out, int32_t Nimg,
                         int32_t Hin, int32_t Win, int32_t Cin, int32_t Cout, int32_t stride, int32_t relu_in, int32_t relu_out,
                         int32_t dtype, float acc_scale, vs_stream_t stream_, const float *residual2 = nullptr) {
    const int H = (Hin - 1) / (stride > 0 ? stride : 1) + 1, W = (Win - 1) / (stride > 0 ? stride : 1) + 1;  // k=3, pad=1
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(in && w && out, "vs_conv3x3_nhwc: null pointer");
    VS_CHECK(Nimg >= 0 && Hin > 0 && Win > 0 && Cin > 0 && Cout > 0 && (stride == 1 || stride == 2), "vs_conv3x3_nhwc: bad sizes");
    VS_CHECK(dtype == 1 || dtype == 2 || dtype == 3 || dtype == 4, "vs_conv3x3_nhwc: dtype must be 1 (f16), 2 (bf16), 3 (f32) or 4 (split)");
    if (dtype == 3 || dtype == 4) {   // f32 activations (and f32 / packed-split weights): the kernels address them in 2-byte units (gemm_common.h)
        VS_CHECK(Cin % 16 == 0, "vs_conv3x3_nhwc: Cin=%d must be a multiple of 16 for f32 (pad the channels)", Cin);
        Cin *= 2;
    }
    VS_CHECK(Cin % 32 == 0, "vs_conv3x3_nhwc: Cin=%d must be a multiple of 32 (pad the channels)", Cin);
    VS_CHECK(Hin < 32767 && Win < 65536 && (long long)Nimg * Hin * Win < 2147483647LL, "vs_conv3x3_nhwc: image too large");
    VS_CHECK(relu_out >= 0 && relu_out <= 2 && (relu_out != 2 || residual), "vs_conv3x3_nhwc: relu_out must be 0, 1 or 2 (2 = mask by `residual`, which must be given)");
    VS_CHECK(((uintptr_t)in & 15) == 0 && ((uintptr_t)w & 15) == 0, "vs_conv3x3_nhwc: 16-byte alignment required");
    if (Nimg == 0) return 0;
    const int in_packed = (relu_in & 16) ? 1 : 0;      // + 16 (split class): `in` is the packed (hi, lo) image, its ReLU already applied by the producer
    relu_in &= ~16;
    VS_CHECK(!in_packed || (dtype == 4 && relu_in == 0), "vs_conv3x3_split_nhwc: a packed input is a split-class operand that carries its ReLU already");
    ConvArgs g{(const unsigned short *)in, (const unsigned short *)w, bias, (const unsigned short *)residual, (unsigned short *)out,
               Nimg, H, W, Cin, Cout, relu_in, relu_out, Hin, Win, stride, nullptr, nullptr, nullptr, 0, 0, 0, acc_scale, 1.f, in_packed, residual2};
    g.korder = conv_korder();
    VS_CHECK(!residual2 || ((dtype == 3 || dtype == 4) && relu_out != 2), "vs_conv3x3_nhwc: a second residual needs f32 activations (dtype 3 / 4) and no mask epilogue");
    const long long M = (long long)Nimg * H * W;
    static const int force = [] { const char *e = getenv("VS_CONV_MI"); return e ? atoi(e) : 0; }();
    int cshift = -1;
    for (int sft = 0; sft < 4; ++sft)
        if (Cin == (64 << sft)) cshift = sft;
    // round 5: the 256 x 256 tile kernel takes any Cin that is a whole number of 64-unit K-tiles per tap (96 -> 128, 192, 384, 768 channels of
    // the DPT reassemble / layer_rn convolutions and the stride-2 768 -> 768 convolution used to run on the 4-wave kernels at 170-270 TF/s)
    const int kpt_any = (Cin % 64 == 0 && Cin / 64 <= 64) ? Cin / 64 : -1;
    const long long t256 = vs::cdiv64(M, 256) * vs::cdiv(Cout, 256);
    static const int t256_min_env = [] { const char *e = getenv("VS_CONV_T256_MIN"); return e ? atoi(e) : 0; }();
    // >= 150 tiles (round 5; was 224): a 59 %-full round of 256 x 256 tiles beats the 4-wave kernel's one and a half rounds at two workgroups
    // per CU in the split class (16 x 16 maps of the bench step: 0.21 -> 0.17 ms per convolution); 8 x 8 maps (48 tiles) stay where they were
    const long long t256_min = t256_min_env > 0 ? t256_min_env : (dtype == 4 ? 150 : 224);
    if (force != 8 && force != 4 && dtype == 4 && cshift < 0 && kpt_any > 0 && Cout % 256 == 0 && (9 * kpt_any) % 2 == 0 && t256 >= t256_min && !g.a_packed) {
        dim3 grid((unsigned)t256), block(512);
        if (relu_in) hipLaunchKernelGGL((conv3x3_256_kernel<kDtSplit, true>), grid, block, 0, stream, g, kpt_any);
        else hipLaunchKernelGGL((conv3x3_256_kernel<kDtSplit, false>), grid, block, 0, stream, g, kpt_any);
        VS_HIP(hipGetLastError());
        return 0;
    }
    if (force != 8 && force != 4 && cshift >= 0 && Cout % 256 == 0 && (9 * Cin / 64) % 2 == 0 && t256 >= t256_min) {
        dim3 grid((unsigned)t256), block(512);
        if (dtype == 4) {
            if (g.a_packed) hipLaunchKernelGGL((conv3x3_256_kernel<kDtSplit, false, 0, true>), grid, block, 0, stream, g, 1 << cshift);
            else if (relu_in) hipLaunchKernelGGL((conv3x3_256_kernel<kDtSplit, true>), grid, block, 0, stream, g, 1 << cshift);
            else hipLaunchKernelGGL((conv3x3_256_kernel<kDtSplit, false>), grid, block, 0, stream, g, 1 << cshift);
        } else if (dtype == 3) {
            if (relu_in) hipLaunchKernelGGL((conv3x3_256_kernel<kDtF32, true>), grid, block, 0, stream, g, 1 << cshift);
            else hipLaunchKernelGGL((conv3x3_256_kernel<kDtF32, false>), grid, block, 0, stream, g, 1 << cshift);
        } else if (dtype == 2) {
            if (relu_in) hipLaunchKernelGGL((conv3x3_256_kernel<true, true>), grid, block, 0, stream, g, 1 << cshift);
            else hipLaunchKernelGGL((conv3x3_256_kernel<true, false>), grid, block, 0, stream, g, 1 << cshift);
        } else {
            if (relu_in) hipLaunchKernelGGL((conv3x3_256_kernel<false, true>), grid, block, 0, stream, g, 1 << cshift);
            else hipLaunchKernelGGL((conv3x3_256_kernel<false, false>), grid, block, 0, stream, g, 1 << cshift);
        }
        VS_HIP(hipGetLastError());
        return 0;
    }
    const long long big = vs::cdiv64(M, 256) * vs::cdiv(Cout, BN);
    if (dtype == 4) {
        VS_CHECK(Cin % 64 == 0, "vs_conv3x3_split_nhwc: Cin must be a multiple of 32");
        static const int no128 = [] { const char *e = getenv("VS_CONV_SPLIT_NO256X128"); return e ? atoi(e) : 0; }();
        if (!no128 && cshift >= 0 && Cout % 128 == 0 && Cout % 256 != 0 && (9 * Cin / 64) % 2 == 0 && big >= 224) {   // (Cout = 256 maps too small for the 256 x 256 kernel: the 4-wave kernel is 8 % faster there, measured)
            dim3 grid((unsigned)(vs::cdiv64(M, 256) * (Cout / 128))), block(512);
            if (g.a_packed) hipLaunchKernelGGL((conv3x3_256x128_split_kernel<false, false, true>), grid, block, 0, stream, g, 1 << cshift);
            else if (relu_in) hipLaunchKernelGGL((conv3x3_256x128_split_kernel<true, false>), grid, block, 0, stream, g, 1 << cshift);
            else hipLaunchKernelGGL((conv3x3_256x128_split_kernel<false, false>), grid, block, 0, stream, g, 1 << cshift);
            VS_HIP(hipGetLastError());
            return 0;
        }
        VS_CHECK(!g.a_packed, "vs_conv3x3_split_nhwc: a packed input is taken by the 256 x 256 and 256 x 128 tile kernels only (this shape runs on the 4-wave kernel)");
        static const int smi = [] { const char *e = getenv("VS_CONV_SPLIT_MI"); return e ? atoi(e) : 0; }();
        // round 6, small batches (one 8-view scene: 16 x 16 / 32 x 32 maps = 2 048 / 8 192 pixels x 256 channels = 32 / 128 tiles of 128 x 128 for
        // 256 CUs): 64-row tiles (MI = 2, the same kernel) double the workgroups.  VS_CONV_MI2=0 for the A/B.
        static const int mi2 = [] { const char *e = getenv("VS_CONV_MI2"); return e ? atoi(e) : 1; }();
        const long long t4 = vs::cdiv64(M, 128) * vs::cdiv(Cout, BN);
        if ((big >= 512 && smi != 4) || smi == 8) hipLaunchKernelGGL((conv3x3_kernel<kDtSplit, 8>), dim3((unsigned)big), dim3(256), 0, stream, g);
        else if (mi2 && smi == 0 && t4 <= 192 && M > 64) hipLaunchKernelGGL((conv3x3_kernel<kDtSplit, 2>), dim3((unsigned)(vs::cdiv64(M, 64) * vs::cdiv(Cout, BN))), dim3(256), 0, stream, g);
        else hipLaunchKernelGGL((conv3x3_kernel<kDtSplit, 4>), dim3((unsigned)t4), dim3(256), 0, stream, g);
    } else if (dtype == 3) {
        hipLaunchKernelGGL((conv3x3_kernel<kDtF32, 4>), dim3((unsigned)(vs::cdiv64(M, 128) * vs::cdiv(Cout, BN))), dim3(256), 0, stream, g);
    } else if ((big >= 256 || force == 8) && force != 4) {
        dim3 grid((unsigned)big);
        if (dtype == 2) hipLaunchKernelGGL((conv3x3_kernel<true, 8>), grid, dim3(256), 0, stream, g);
        else hipLaunchKernelGGL((conv3x3_kernel<false, 8>), grid, dim3(256), 0, stream, g);
    } else {
        dim3 grid((unsigned)(vs::cdiv64(M, 128) * vs::cdiv(Cout, BN)));
        if (dtype == 2) hipLaunchKernelGGL((conv3x3_kernel<true, 4>), grid, dim3(256), 0, stream, g);
        else hipLaunchKernelGGL((conv3x3_kernel<false, 4>), grid, dim3(256), 0, stream, g);
    }
    VS_HIP(hipGetLastError());
    return 0;
}

extern "C" int vs_conv3x3_nhwc(const void *in, const void *w, const float *bias, const void *residual, void *out, int32_t Nimg,
                               int32_t Hin, int32_t Win, int32_t Cin, int32_t Cout, int32_t stride, int32_t relu_in, int32_t relu_out,
                               int32_t dtype, vs_stream_t stream_) {
    VS_CHECK(dtype != 4, "vs_conv3x3_nhwc: split operands go through vs_conv3x3_split_nhwc (they need the weights' scale)");
    return conv3x3_entry(in, w, bias, residual, out, Nimg, Hin, Win, Cin, Cout, stride, relu_in, relu_out, dtype, 1.f, stream_);
}

// 3x3 convolution on split operands (gemm_common.h, kDtSplit): in / residual / out f32 NHWC, wp = vs_split_pack_weight image of the
// [Cout, 9 * Cin] f32 weight (tap-major, channel-minor; Cin a multiple of 32), acc_scale = 2^-scale_exp.  Otherwise as vs_conv3x3_nhwc.
extern "C" int vs_conv3x3_split_nhwc(const float *in, const void *wp, float acc_scale, const float *bias, const float *residual, float *out,
                                     int32_t Nimg, int32_t Hin, int32_t Win, int32_t Cin, int32_t Cout, int32_t stride, int32_t relu_in,
                                     int32_t relu_out, vs_stream_t stream_) {
    VS_CHECK(acc_scale > 0.f && Cin % 32 == 0, "vs_conv3x3_split_nhwc: acc_scale must be positive and Cin a multiple of 32");
    return conv3x3_entry(in, wp, bias, residual, out, Nimg, Hin, Win, Cin, Cout, stride, relu_in, relu_out, 4, acc_scale, stream_);
}

// vs_conv3x3_split_nhwc with a SECOND residual: out = conv(in) + bias + residual + residual2 (relu_out applies after both) -- the
// FeatureFusionBlock's `x + ResidualConvUnit(skip)` (dpt_block.py:196-208) folded into the unit's last convolution.
extern "C" int vs_conv3x3_split_res2_nhwc(const float *in, const void *wp, float acc_scale, const float *bias, const float *residual,
                                          const float *residual2, float *out, int32_t Nimg, int32_t Hin, int32_t Win, int32_t Cin, int32_t Cout,
                                          int32_t stride, int32_t relu_in, int32_t relu_out, vs_stream_t stream_) {
    VS_CHECK(acc_scale > 0.f && Cin % 32 == 0, "vs_conv3x3_split_res2_nhwc: acc_scale must be positive and Cin a multiple of 32");
    return conv3x3_entry(in, wp, bias, residual, out, Nimg, Hin, Win, Cin, Cout, stride, relu_in, relu_out, 4, acc_scale, stream_, residual2);
}

/* out2[pixel, 0..C2) = W2 relu_out(conv3x3(in) + bias) + bias2 in ONE kernel (the 3x3 result stays on chip): the last two layers of
 * both DPT heads (dpt_block.py:316-343).  Two forms: Cout = 256 and C2pad in {16,..,96} (multiple of 16; w2 [C2pad, 256], bias2 [C2pad],
 * ld2 >= C2pad; MFMA through LDS), or Cout = 128 and C2 <= 4 (w2 [C2, 128], bias2 [4], ld2 >= 4; VALU dot products).  stride 1,
 * N*H*W a multiple of 256, dtype 1 f16 / 2 bf16. */
extern "C" int vs_conv3x3_head1x1_nhwc(const void *in, const void *w, const float *bias, const void *w2, const float *bias2, void *out2,
                                       int32_t Nimg, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t C2, int32_t C2pad, int32_t ld2,
                                       int32_t relu_in, int32_t relu_out, int32_t dtype, vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(in && w && w2 && bias2 && out2, "vs_conv3x3_head1x1_nhwc: null pointer");
    VS_CHECK(dtype == 1 || dtype == 2, "vs_conv3x3_head1x1_nhwc: dtype must be 1 (f16) or 2 (bf16)");
    VS_CHECK(Nimg > 0 && H > 0 && W > 0 && H < 32767 && W < 65536, "vs_conv3x3_head1x1_nhwc: bad sizes");
    const long long M = (long long)Nimg * H * W;
    VS_CHECK(M % 256 == 0 && M < 2147483647LL, "vs_conv3x3_head1x1_nhwc: N*H*W must be a multiple of 256");
    VS_CHECK(((uintptr_t)in & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)w2 & 15) == 0 && ((uintptr_t)out2 & 7) == 0 &&
             ((uintptr_t)bias2 & 15) == 0, "vs_conv3x3_head1x1_nhwc: alignment");
    VS_CHECK(relu_out == 0 || relu_out == 1, "vs_conv3x3_head1x1_nhwc: relu_out must be 0 or 1");
    ConvArgs g{(const unsigned short *)in, (const unsigned short *)w, bias, nullptr, nullptr, Nimg, H, W, Cin, Cout, relu_in, relu_out, H, W, 1,
               (const unsigned short *)w2, bias2, (unsigned short *)out2, C2, C2pad, ld2, 1.f, 1.f};
    g.korder = conv_korder();
    if (Cout == 256) {
        int cshift = -1;
        for (int sft = 0; sft < 4; ++sft)
            if (Cin == (64 << sft)) cshift = sft;
        VS_CHECK(cshift >= 0 && (9 * Cin / 64) % 2 == 0, "vs_conv3x3_head1x1_nhwc: Cin=%d not supported with Cout=256", Cin);
        VS_CHECK(C2pad % 16 == 0 && C2pad >= 16 && C2pad <= 96 && C2 <= C2pad && ld2 >= C2pad && ld2 % 4 == 0 && !relu_in,
                 "vs_conv3x3_head1x1_nhwc: need C2pad in 16..96 (multiple of 16), ld2 >= C2pad, no relu_in (C2pad=%d ld2=%d)", C2pad, ld2);
        dim3 grid((unsigned)(M / 256)), block(512);
#define VS_FUSE(NF_)                                                                                                        \
        if (dtype == 2) hipLaunchKernelGGL((conv3x3_256_kernel<1, false, NF_>), grid, block, 0, stream, g, 1 << cshift);         \
        else hipLaunchKernelGGL((conv3x3_256_kernel<0, false, NF_>), grid, block, 0, stream, g, 1 << cshift);
        switch (C2pad / 16) {
            case 1: VS_FUSE(1) break;
            case 2: VS_FUSE(2) break;
            case 3: VS_FUSE(3) break;
            case 4: VS_FUSE(4) break;
            case 5: VS_FUSE(5) break;
            default: VS_FUSE(6) break;
        }
#undef VS_FUSE
    } else {
        VS_CHECK(Cout == 128 && C2 >= 1 && C2 <= 4 && ld2 >= 4 && ld2 % 4 == 0 && Cin % 32 == 0,
                 "vs_conv3x3_head1x1_nhwc: the dot-product form needs Cout = 128, C2 <= 4, ld2 >= 4 (Cout=%d C2=%d)", Cout, C2);
        dim3 grid((unsigned)(M / 256));
        if (dtype == 2) hipLaunchKernelGGL((conv3x3_kernel<1, 8, true>), grid, dim3(256), 0, stream, g);
        else hipLaunchKernelGGL((conv3x3_kernel<0, 8, true>), grid, dim3(256), 0, stream, g);
    }
    VS_HIP(hipGetLastError());
    return 0;
}

// conv3(Cin -> 128) -> ReLU -> conv1(128 -> C2 <= 3) of the pts3d head on split operands, one kernel (the dot-product form of
// vs_conv3x3_head1x1_nhwc): in f32 NHWC, wp packed [128, 9 * Cin], w2 f32 [C2, 128], bias2 f32 [4], out2 f32 [N*H*W, ld2 >= 4, ld2 % 4 == 0];
// N*H*W a multiple of 256, Cin a multiple of 32.
extern "C" int vs_conv3x3_head_dot_split_nhwc(const float *in, const void *wp, float acc_scale, const float *bias, const float *w2, const float *bias2,
                                              float *out2, int32_t Nimg, int32_t H, int32_t W, int32_t Cin, int32_t C2, int32_t ld2, int32_t relu_in,
                                              int32_t relu_out, vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int in_packed = (relu_out & 16) ? 1 : 0;     // + 16: `in` is the packed (hi, lo) image of the f32 tensor (vs_upsample2x_nhwc ... + 16), ABI 5
    relu_out &= ~16;
    VS_CHECK(in && wp && w2 && bias2 && out2 && acc_scale > 0.f, "vs_conv3x3_head_dot_split_nhwc: null pointer / bad scale");
    const long long M = (long long)Nimg * H * W;
    VS_CHECK(Nimg > 0 && H > 0 && W > 0 && H < 32767 && W < 65536 && M % 256 == 0 && M < 2147483647LL, "vs_conv3x3_head_dot_split_nhwc: N*H*W must be a positive multiple of 256");
    VS_CHECK(Cin % 32 == 0 && C2 >= 1 && C2 <= 4 && ld2 >= 4 && ld2 % 4 == 0 && (relu_out == 0 || relu_out == 1), "vs_conv3x3_head_dot_split_nhwc: Cin %% 32, C2 <= 4, ld2 >= 4");
    VS_CHECK((((uintptr_t)in | (uintptr_t)wp | (uintptr_t)out2 | (uintptr_t)bias2) & 15) == 0, "vs_conv3x3_head_dot_split_nhwc: 16-byte alignment required");
    ConvArgs g{(const unsigned short *)in, (const unsigned short *)wp, bias, nullptr, nullptr, Nimg, H, W, 2 * Cin, 128, relu_in, relu_out, H, W, 1,
               (const unsigned short *)w2, bias2, (unsigned short *)out2, C2, C2, ld2, acc_scale, 1.f};
    g.korder = conv_korder();
    int cshift = -1;
    for (int sft = 0; sft < 4; ++sft)
        if (2 * Cin == (64 << sft)) cshift = sft;
    static const int no128 = [] { const char *e = getenv("VS_CONV_SPLIT_NO256X128"); return e ? atoi(e) : 0; }();
    const bool tile128 = !no128 && cshift >= 0 && (9 * 2 * Cin / 64) % 2 == 0 && !relu_in;
    VS_CHECK(!in_packed || tile128, "vs_conv3x3_head_dot_split_nhwc: a packed input needs the 256 x 128 tile kernel (Cin in {32, 64, 128, 256}, no relu_in)");
    if (tile128 && in_packed)
        hipLaunchKernelGGL((conv3x3_256x128_split_kernel<false, true, true>), dim3((unsigned)(M / 256)), dim3(512), 0, stream, g, 1 << cshift);
    else if (tile128)
        hipLaunchKernelGGL((conv3x3_256x128_split_kernel<false, true>), dim3((unsigned)(M / 256)), dim3(512), 0, stream, g, 1 << cshift);
    else
        hipLaunchKernelGGL((conv3x3_kernel<kDtSplit, 8, true>), dim3((unsigned)(M / 256)), dim3(256), 0, stream, g);
    VS_HIP(hipGetLastError());
    return 0;
}

// conv3x3(Cin -> 256) -> relu_out -> conv1x1(256 -> C2 <= C2pad <= 96) in one kernel on split operands (the MFMA form of
// vs_conv3x3_head1x1_nhwc): in f32 NHWC, wp packed [256, 9 * Cin], w2p = vs_split_pack_weight image of the [C2pad, 256] f32 weight (rows >= C2
// zero), bias2 f32 [C2pad], out2 f32 [N*H*W, ld2 >= C2pad, ld2 % 4 == 0]; N*H*W a multiple of 256, Cin in {32, 64, 128, 256}.
extern "C" int vs_conv3x3_head1x1_split_nhwc(const float *in, const void *wp, float acc_scale, const float *bias, const void *w2p, float acc_scale2,
                                             const float *bias2, float *out2, int32_t Nimg, int32_t H, int32_t W, int32_t Cin, int32_t C2, int32_t C2pad,
                                             int32_t ld2, int32_t relu_out, vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int in_packed = (relu_out & 16) ? 1 : 0;     // + 16: `in` is the packed (hi, lo) image of the f32 tensor (vs_upsample2x_nhwc relu_add + 16, ...)
    relu_out &= ~16;
    VS_CHECK(in && wp && w2p && bias2 && out2 && acc_scale > 0.f && acc_scale2 > 0.f, "vs_conv3x3_head1x1_split_nhwc: null pointer / bad scale");
    const long long M = (long long)Nimg * H * W;
    VS_CHECK(Nimg > 0 && H > 0 && W > 0 && H < 32767 && W < 65536 && M % 256 == 0 && M < 2147483647LL, "vs_conv3x3_head1x1_split_nhwc: N*H*W must be a positive multiple of 256");
    int cshift = -1;
    for (int sft = 0; sft < 4; ++sft)
        if (2 * Cin == (64 << sft)) cshift = sft;
    VS_CHECK(cshift >= 0 && (9 * 2 * Cin / 64) % 2 == 0, "vs_conv3x3_head1x1_split_nhwc: Cin=%d not supported", Cin);
    VS_CHECK(C2pad % 16 == 0 && C2pad >= 16 && C2pad <= 96 && C2 <= C2pad && ld2 >= C2pad && ld2 % 4 == 0 && (relu_out == 0 || relu_out == 1),
             "vs_conv3x3_head1x1_split_nhwc: need C2pad in 16..96 (multiple of 16), ld2 >= C2pad (C2pad=%d ld2=%d)", C2pad, ld2);
    VS_CHECK((((uintptr_t)in | (uintptr_t)wp | (uintptr_t)w2p | (uintptr_t)out2 | (uintptr_t)bias2) & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0),
             "vs_conv3x3_head1x1_split_nhwc: 16-byte alignment required");
    ConvArgs g{(const unsigned short *)in, (const unsigned short *)wp, bias, nullptr, nullptr, Nimg, H, W, 2 * Cin, 256, 0, relu_out, H, W, 1,
               (const unsigned short *)w2p, bias2, (unsigned short *)out2, C2, C2pad, ld2, acc_scale, acc_scale2};
    g.korder = conv_korder();
    dim3 grid((unsigned)(M / 256)), block(512);
    g.a_packed = in_packed;
#define VS_HEAD(NF_) { if (in_packed) hipLaunchKernelGGL((conv3x3_256_kernel<kDtSplit, false, NF_, true>), grid, block, 0, stream, g, 1 << cshift); \
                       else hipLaunchKernelGGL((conv3x3_256_kernel<kDtSplit, false, NF_>), grid, block, 0, stream, g, 1 << cshift); }
    switch (C2pad / 16) {
        case 1: VS_HEAD(1) break;
        case 2: VS_HEAD(2) break;
        case 3: VS_HEAD(3) break;
        case 4: VS_HEAD(4) break;
        case 5: VS_HEAD(5) break;
        default: VS_HEAD(6) break;
    }
#undef VS_HEAD
    VS_HIP(hipGetLastError());
    return 0;
}

extern "C" int vs_upsample2x_nhwc(const void *in, const void *add, void *out, int32_t Nimg, int32_t H, int32_t W, int32_t C,
                                  int32_t relu_add, int32_t dtype, vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(in && out, "vs_upsample2x_nhwc: null pointer");
    VS_CHECK(dtype == 1 || dtype == 2 || dtype == 3, "vs_upsample2x_nhwc: dtype must be 1 (f16), 2 (bf16) or 3 (f32)");
    const int pack_out = (relu_add & 16) ? 1 : 0;      // + 16 (f32 only): write the packed (hi, lo) form of the split class (C % 32 == 0, 128-byte aligned out)
    relu_add &= ~16;
    VS_CHECK(!pack_out || (dtype == 3 && C % 32 == 0 && ((uintptr_t)out & 127) == 0), "vs_upsample2x_nhwc: a packed output needs dtype 3, C %% 32 == 0, 128-byte alignment");
    if (dtype == 3) {
        VS_CHECK(C % 4 == 0, "vs_upsample2x_nhwc: C=%d must be a multiple of 4", C);
        if ((long long)Nimg * H * W * C <= 0) return 0;
        static const int oldk = [] { const char *e = getenv("VS_UPSAMPLE_F32_OLD"); return e ? atoi(e) : 0; }();
        if (H >= 2 && W >= 2 && !oldk) {
            const int rows = Nimg * H, gy = rows < 32768 ? rows : 32768;
            hipLaunchKernelGGL(upsample2x_f32_block_kernel, dim3((unsigned)vs::cdiv(W * (C / 4), 256), gy, vs::cdiv(rows, gy)), dim3(256), 0, stream,
                               (const float *)in, (const float *)add, (float *)out, Nimg, H, W, C, relu_add, pack_out);
            VS_HIP(hipGetLastError());
            return 0;
        }
        VS_CHECK(!pack_out, "vs_upsample2x_nhwc: the packed output needs H, W >= 2");
        const int rows = Nimg * 2 * H, gy = rows < 32768 ? rows : 32768;
        hipLaunchKernelGGL(upsample2x_f32_kernel, dim3((unsigned)vs::cdiv(2 * W * (C / 4), 256), gy, vs::cdiv(rows, gy)), dim3(256), 0, stream,
                           (const float *)in, (const float *)add, (float *)out, Nimg, H, W, C, relu_add);
        VS_HIP(hipGetLastError());
        return 0;
    }
    VS_CHECK(C % 8 == 0, "vs_upsample2x_nhwc: C=%d must be a multiple of 8", C);
    if ((long long)Nimg * H * W * C <= 0) return 0;
    const int rows = Nimg * H, gy = rows < 32768 ? rows : 32768;   // one thread per 2x2 output block and 8 channels
    dim3 grid((unsigned)vs::cdiv(W * (C / 8), 256), gy, vs::cdiv(rows, gy)), block(256);
    if (dtype == 2) hipLaunchKernelGGL(upsample2x_kernel<true>, grid, block, 0, stream, (const unsigned short *)in, (const unsigned short *)add, (unsigned short *)out, Nimg, H, W, C, relu_add);
    else hipLaunchKernelGGL(upsample2x_kernel<false>, grid, block, 0, stream, (const unsigned short *)in, (const unsigned short *)add, (unsigned short *)out, Nimg, H, W, C, relu_add);
    VS_HIP(hipGetLastError());
    return 0;
}

extern "C" int vs_upsample2x_backward_nhwc(const void *dout, void *din, int32_t Nimg, int32_t H, int32_t W, int32_t C, int32_t dtype,
                                           vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(dout && din, "vs_upsample2x_backward_nhwc: null pointer");
    VS_CHECK(C % 8 == 0, "vs_upsample2x_backward_nhwc: C=%d must be a multiple of 8", C);
    VS_CHECK(dtype == 1 || dtype == 2, "vs_upsample2x_backward_nhwc: dtype must be 1 (f16) or 2 (bf16)");
    if ((long long)Nimg * H * W * C <= 0) return 0;
    const int rows = Nimg * H, gy = rows < 32768 ? rows : 32768;
    dim3 grid((unsigned)vs::cdiv(W * (C / 8), 256), gy, vs::cdiv(rows, gy)), block(256);
    if (dtype == 2) hipLaunchKernelGGL(upsample2x_backward_kernel<true>, grid, block, 0, stream, (const unsigned short *)dout, (unsigned short *)din, Nimg, H, W, C);
    else hipLaunchKernelGGL(upsample2x_backward_kernel<false>, grid, block, 0, stream, (const unsigned short *)dout, (unsigned short *)din, Nimg, H, W, C);
    VS_HIP(hipGetLastError());
    return 0;
}
