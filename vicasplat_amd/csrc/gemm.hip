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
// wave owns 64x64 = 4x4 MFMA 16x16 fragments (64 accumulator VGPRs).  Operands go HBM -> LDS directly with
// global_load_lds_dwordx4 (no VGPR round trip), XOR-swizzled for conflict-free ds_read_b128 fragment reads; 32 KiB of
// LDS per workgroup keeps 4 workgroups (16 waves) resident per CU, whose interleaving hides the load latency.
// blockIdx is remapped so that consecutive tiles of one XCD share A/W panels in that XCD's L2.
//
// Epilogues (all fused, nothing round-trips HBM):
//   0 STORE16   out16[row(m), n] = acc + bias
//   1 GELU16    out16[row(m), n] = gelu_erf(acc + bias)                (croco/blocks.py:60,68: exact GELU)
//   2 RESID32   x32[row(m), n]  += (1 + gate[m / gate_rows, n]) * (acc + bias)   (backbone_vica.py:274-278,302,327,331)
//   3 STORE32   out32[row(m), n] = acc + bias
// row(m) = (m / grp_in) * grp_out + grp_off + m % grp_in lets a GEMM write straight into a larger token buffer
// (e.g. the 257 image tokens of a frame behind the frame's camera token).
#include "gemm256.h"

#include <cstdlib>

namespace {

template <int BF16, int EPI, int MI>
__global__ void __launch_bounds__(256, (MI == 8 || BF16 == kDtSplit) ? 2 : 3) gemm_kernel(const GemmArgs g_in) {
    GemmArgs g = g_in;
    int ksp = (EPI == 2 && g.ksplit > 1) ? (int)blockIdx.y : 0;  // split-K slice (tail launches of the f32 residual epilogue)
    int tap = -1, lin_tile = -1;
    if (EPI == 2 && g.ntaps > 0) {  // tap-fused weight gradient: blockIdx.x = (k-slice, tap, tile)
        const int tiles = ((g.M - g.m_lo + 32 * MI - 1) / (32 * MI)) * ((g.N + BN - 1) / BN);
        ksp = blockIdx.x / (tiles * g.ntaps);
        const int rem = blockIdx.x - ksp * tiles * g.ntaps;
        tap = rem / tiles;
        lin_tile = rem - tap * tiles;
        if (g.tap_on_a) g.A = reinterpret_cast<const unsigned short *>(g.A) + g_in.tap_shift[tap];
        else g.W = reinterpret_cast<const unsigned short *>(g.W) + g_in.tap_shift[tap];  // (indexing the kernarg, not the copy: a
                                                                                          //  dynamically indexed local struct lives in scratch)
        g.out = reinterpret_cast<float *>(g.out) + (long long)tap * g.tap_out_stride;
    }
    if (ksp > 0) g.bias = nullptr;                                      // the bias belongs to slice 0
    const int ksplit_n = g.ksplit;
    if (EPI == 2 && g.partials) {  // weight-gradient slice: store the partial tile to the workspace (see GemmArgs::partials)
        const int nt = g.ntaps > 0 ? g.ntaps : 1;
        g.out = g.partials + ((long long)(ksp * nt + (tap > 0 ? tap : 0)) * g.M) * g.N;
        g.ldo = g.N;
        g.ksplit = -1;
    }
    constexpr int BM = 32 * MI;
    constexpr bool SPLIT = BF16 == kDtSplit;
    constexpr int NS = SPLIT ? 4 : 3;           // LDS ring depth (split operands: two PAIRS of stages, see below)
    constexpr int GL = MI / 2 + 2;              // global_load_lds per wave per stage (A: BM/16/4, W: 128/16/4)
    // K is consumed in 32-wide stages through a 3-deep LDS ring filled by global_load_lds_dwordx4 (LDS address = wave-uniform
    // base + lane*16, so every stage is lane-linear: rows of 64 B, chunk index XOR-swizzled on the GLOBAL source address and
    // again on the fragment read).  Two stages are always in flight behind the one being multiplied: counted s_waitcnt
    // vmcnt(GL) + raw s_barrier, never a full drain inside the loop.
    // ONE __shared__ object: with a second one hipcc inserts s_waitcnt vmcnt(0) before the first ds_read of every k-step
    // of a global_load_lds pipeline (cdna_hip_programming.md, "three .s-level traps").
    __shared__ __attribute__((aligned(1024))) unsigned short smem[NS * (BM + BN) * 32];
    unsigned short *const sA = smem, *const sW = smem + NS * BM * 32;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;

    // XCD-aware tile order: blocks b, b+8, b+16.. run on the same XCD (observed, speed only) -> give each XCD a
    // contiguous range of tiles so neighbouring tiles (same A panel) hit that XCD's L2.
    const int tiles_n = (g.N + BN - 1) / BN;
    const int tiles_m = (g.M - g.m_lo + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    if (lin_tile >= 0) {
        bid = lin_tile;
    } else {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = g.m_lo + tm * BM, n0 = tn * BN;

    // staging: one global_load_lds piece = 16 tile rows x 64 B; lane l lands at (row p*16 + (l>>2), slot l&3) and therefore
    // fetches global chunk (l&3) ^ swz4(row).  Wave w owns A pieces w*MI/2 .. and W pieces 2w, 2w+1.
    const unsigned short *A = reinterpret_cast<const unsigned short *>(g.A);
    const unsigned short *W = reinterpret_cast<const unsigned short *>(g.W);
    const int rho = lane >> 2, gchunk = (lane & 3) ^ swz4(rho);
    const unsigned short *pa[MI / 2], *pw[2];
#pragma unroll
    for (int i = 0; i < MI / 2; ++i) {
        const int ra_ = min(m0 + (wid * (MI / 2) + i) * 16 + rho, g.M - 1);
        const int rg_ = ra_ / g.a_grp_in;
        const size_t arow = (size_t)rg_ * g.a_grp_out + g.a_grp_off + (ra_ % g.a_grp_in) + (size_t)(rg_ / g.a_sup_in) * g.a_sup_extra;
        pa[i] = A + arow * g.lda + gchunk * 8;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rw_ = min(n0 + (wid * 2 + i) * 16 + rho, g.N - 1);
        pw[i] = W + (size_t)rw_ * g.ldw + gchunk * 8;
    }
    typedef void __attribute__((address_space(3))) *lptr_t;

    f4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fg = lane >> 4;
    const int nk = (EPI == 2 && ksplit_n > 1) ? g.K / 32 / ksplit_n : g.K / 32;
    if (ksp > 0) {
#pragma unroll
        for (int i = 0; i < MI / 2; ++i) pa[i] += ksp * (g.a_slice_stride ? g.a_slice_stride : (long long)nk * g.a_kstride);
#pragma unroll
        for (int i = 0; i < 2; ++i) pw[i] += ksp * (g.w_slice_stride ? g.w_slice_stride : (long long)nk * 32);
    }
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    const unsigned ldsA = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(wid * (MI / 2)) * 1024u);
    const unsigned ldsW = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(NS * BM * 32 * 2) + (unsigned)(wid * 2) * 1024u);
#define VS_STAGE(kt_, slot_)                                                                                   \
    {                                                                                                          \
        const int k0_ = (kt_) * 32;                                                                            \
        const long long ka_ = (long long)(kt_) * g.a_kstride;                                                  \
        _Pragma("unroll") for (int i = 0; i < MI / 2; ++i)                                                     \
            glds16(pa[i] + ka_, ldsA + (unsigned)((slot_) * (BM * 32 * 2) + i * 1024));                        \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                          \
            glds16(pw[i] + k0_, ldsW + (unsigned)((slot_) * (BN * 32 * 2) + i * 1024));                        \
    }
    // one pipeline step with COMPILE-TIME ring slots (so that hipcc can prove the DMA target and the fragment reads do
    // not alias and keeps the counted vmcnt instead of draining the queue before the first ds_read)
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
        if ((kt_) + 2 < nk) VS_STAGE((kt_) + 2, nslot_)                                                     \
        const unsigned short *cA = sA + (slot_) * (BM * 32), *cW = sW + (slot_) * (BN * 32);                \
        uint4 fb[4];                                                                                        \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                     \
            const int rb_ = wc * 64 + j * 16 + frow;                                                        \
            fb[j] = *reinterpret_cast<const uint4 *>(&cW[rb_ * 32 + ((fg ^ swz4(rb_)) << 3)]);              \
        }                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < MI; ++i) {                                                    \
            const int ra_ = wr * (16 * MI) + i * 16 + frow;                                                 \
            const uint4 fa = *reinterpret_cast<const uint4 *>(&cA[ra_ * 32 + ((fg ^ swz4(ra_)) << 3)]);     \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = mfma<BF16>(fb[j], fa, acc[i][j]);     \
        }                                                                                                   \
    }
    if constexpr (SPLIT) {
        // split operands: stages are consumed in PAIRS (2p, 2p + 1) = the two 64-byte halves of a 128-byte block of 32 k -- floats 0..15 /
        // 16..31 of an f32 A row, hi / lo halves of a packed W row -- double-buffered as two pairs of ring slots.  One barrier per pair:
        // pair p has landed (its loads are the only ones outstanding) and every wave is done with pair p - 1, whose slots take pair p + 1.
        const int np = nk >> 1;
        VS_STAGE(0, 0)
        VS_STAGE(1, 1)
        for (int p = 0; p < np; ++p) {
            const int s0 = (p & 1) * 2;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (p + 1 < np) {
                VS_STAGE(2 * p + 2, s0 ^ 2)
                VS_STAGE(2 * p + 3, (s0 ^ 2) + 1)
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
                if (!g.a_packed) split8(fa0, fa1);     // (wave-uniform; a packed A block is already [hi | lo])
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mma2<BF16>(fbh[j], fbl[j], fa0, fa1, acc[i][j]);
            }
        }
    } else {
        VS_STAGE(0, 0)
        if (nk > 1) VS_STAGE(1, 1)
        for (int kt = 0; kt < nk; kt += 3) {
            VS_STEP(kt, 0, 2)
            if (kt + 1 < nk) VS_STEP(kt + 1, 1, 0)
            if (kt + 2 < nk) VS_STEP(kt + 2, 2, 1)
        }
    }
#undef VS_STEP
#undef VS_STAGE
    __syncthreads();  // all waves done with the ring before the epilogue reuses it

    gemm_epilogue<BF16, EPI, MI>(g, acc, m0 + wr * (16 * MI), n0 + wc * 64, smem, wid, lane);
}

// ---- small-M path (camera-token GEMMs: M = B*T rows; and the <= 64 leftover rows of a split GEMM).  Latency bound:
// one workgroup per 16 output columns, its NW waves split K into contiguous ranges and walk them 4 k-steps at a time
// with all 20 fragment loads of a batch in flight (A is tiny and L2-resident, W is read exactly once per launch);
// partial sums meet in LDS, then the same fused epilogues. ----
template <int BF16, int EPI, int NW, int MF>
__global__ void __launch_bounds__(64 * NW) gemm_smallm_kernel(const GemmArgs g) {
    __shared__ float red[NW][MF][256];  // [wave][m-frag][16x16]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n0 = blockIdx.x * 16;
    const int mlo = g.m_lo + (int)blockIdx.y * 16 * MF;     // (blockIdx.y: row groups of 16 * MF rows -- the <= 256-row tails of the big GEMMs)
    const int frow = lane & 15, fg = lane >> 4;
    const unsigned short *A = reinterpret_cast<const unsigned short *>(g.A);
    const unsigned short *W = reinterpret_cast<const unsigned short *>(g.W);
    // MF 16-row fragments cover rows [mlo, mlo + 16*MF); rows past M are clamped duplicates whose results are dropped
    const unsigned short *pa[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const int ra_ = min(mlo + i * 16 + frow, g.M - 1);
        const size_t arow = (size_t)(ra_ / g.a_grp_in) * g.a_grp_out + g.a_grp_off + (ra_ % g.a_grp_in);
        pa[i] = A + arow * g.lda + fg * 8;
    }
    const unsigned short *pw = W + (size_t)min(n0 + frow, g.N - 1) * g.ldw + fg * 8;
    f4 acc[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    if constexpr (BF16 == kDtSplit) {
        // split operands: k-steps in pairs = one 128-byte block of 32 k (f32 A: floats 4 fg.. and 16 + 4 fg..; packed W: hi / lo chunk fg)
        const int kpairs = g.K / 64;
        const int per = (kpairs + NW - 1) / NW;
        const int kp0 = wid * per, kp1 = min(kp0 + per, kpairs);
        constexpr int U = MF == 4 ? 2 : 4;
        int kp = kp0;
        for (; kp + U <= kp1; kp += U) {
            uint4 fb[U][2], fa[U][MF][2];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                fb[u][0] = *reinterpret_cast<const uint4 *>(pw + (kp + u) * 64);
                fb[u][1] = *reinterpret_cast<const uint4 *>(pw + (kp + u) * 64 + 32);
#pragma unroll
                for (int i = 0; i < MF; ++i) {
                    fa[u][i][0] = *reinterpret_cast<const uint4 *>(pa[i] + (kp + u) * 64);
                    fa[u][i][1] = *reinterpret_cast<const uint4 *>(pa[i] + (kp + u) * 64 + 32);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < MF; ++i) {
                    if (!g.a_packed) split8(fa[u][i][0], fa[u][i][1]);      // (a packed row holds hi chunk fg / lo chunk fg at the same two addresses)
                    acc[i] = mma2<BF16>(fa[u][i][0], fa[u][i][1], fb[u][0], fb[u][1], acc[i]);
                }
        }
        for (; kp < kp1; ++kp) {
            const uint4 fb0 = *reinterpret_cast<const uint4 *>(pw + kp * 64), fb1 = *reinterpret_cast<const uint4 *>(pw + kp * 64 + 32);
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                uint4 fa0 = *reinterpret_cast<const uint4 *>(pa[i] + kp * 64), fa1 = *reinterpret_cast<const uint4 *>(pa[i] + kp * 64 + 32);
                if (!g.a_packed) split8(fa0, fa1);
                acc[i] = mma2<BF16>(fa0, fa1, fb0, fb1, acc[i]);
            }
        }
    }
    const int ksteps = BF16 == kDtSplit ? 0 : g.K / 32;
    const int per = (ksteps + NW - 1) / NW;
    const int ks0 = wid * per, ks1 = min(ks0 + per, ksteps);
    constexpr int U = MF == 4 ? 4 : 8;  // k-steps per batch: (MF + 1) * U 16-byte loads in flight per lane
    int ks = ks0;
    for (; ks + U <= ks1; ks += U) {
        uint4 fb[U], fa[U][MF];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            fb[u] = *reinterpret_cast<const uint4 *>(pw + (ks + u) * 32);
#pragma unroll
            for (int i = 0; i < MF; ++i) fa[u][i] = *reinterpret_cast<const uint4 *>(pa[i] + (ks + u) * 32);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < MF; ++i) acc[i] = mfma<BF16>(fa[u][i], fb[u], acc[i]);
    }
    for (; ks < ks1; ++ks) {
        const uint4 fb = *reinterpret_cast<const uint4 *>(pw + ks * 32);
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const uint4 fa = *reinterpret_cast<const uint4 *>(pa[i] + ks * 32);
            acc[i] = mfma<BF16>(fa, fb, acc[i]);
        }
    }
    // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + r
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wid][i][(fg * 4 + r) * 16 + frow] = acc[i][r];
    __syncthreads();
    for (int e = tid; e < MF * 256; e += 64 * NW) {
        const int i = e >> 8, rc = e & 255, m = mlo + i * 16 + (rc >> 4), n = n0 + (rc & 15);
        if (m >= g.M || n >= g.N) continue;
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w][i][rc];
        if constexpr (BF16 == kDtSplit) v *= g.acc_scale;
        if (g.bias) v += g.bias[n];
        const size_t orow = (size_t)(m / g.grp_in) * g.grp_out + g.grp_off + (m % g.grp_in);
        if constexpr (EPI == 0 && !is_f32io(BF16)) {
            reinterpret_cast<unsigned short *>(g.out)[orow * g.ldo + n] = to16<BF16>(v);
        } else if constexpr (EPI == 1 && !is_f32io(BF16)) {
            v = gelu_poly(v);
            reinterpret_cast<unsigned short *>(g.out)[orow * g.ldo + n] = to16<BF16>(v);
        } else if constexpr (EPI == 1) {     // f32 operands: GELU, f32 store
            reinterpret_cast<float *>(g.out)[orow * g.ldo + n] = gelu_erf(v);
        } else if constexpr (EPI == 2) {
            float *o = reinterpret_cast<float *>(g.out) + orow * g.ldo + n;
            const float gt = g.gate ? g.gate[(size_t)(m / g.gate_rows) * g.gate_ld + n] : 0.0f;
            *o = (g.resid ? g.resid[orow * g.ldo + n] : *o) + (1.0f + gt) * v;
        } else {
            reinterpret_cast<float *>(g.out)[orow * g.ldo + n] = v;
        }
    }
}

// ---- skinny path, split class (round 5): M - m_lo <= 256 rows -- the 192-row tails of the big ViT GEMMs (M = frames x 257 never fills the
// last 256-row tile) and the decoder's camera-token GEMMs (M = frames).  Those launches were LATENCY bound, not bandwidth or MFMA bound:
// a handful of 128 x 128 tiles each walking all of K one 32-wide stage per barrier (32-90 us for 5-15 us of work; 11 ms of the 257 ms
// step), or -- on the weight-streaming kernel above -- 16 output columns per workgroup, i.e. A re-read N / 16 times through L2.
// Here a workgroup owns a 64 x 64 output tile; its eight waves split K into contiguous ranges and each keeps the WHOLE tile in
// registers (16 accumulator fragments), reading its operand fragments straight from global memory (packed rows are fragment-ready:
// 16 B hi + 16 B lo per lane and 32-k block), two 32-k blocks = 32 x 16-byte loads in flight per lane, no LDS staging and no barrier in
// the K loop.  The partial tiles meet once in LDS; wave w then runs the shared epilogue on row fragment w, so every fused epilogue (GELU,
// RoPE, packed output, gated f32 residual) is the tile kernels' code.  Epilogue 2 may also split K over blockIdx.z (atomics, as the tile
// kernels' tails do) when N is too narrow to give the chip enough workgroups.
template <int BF16, int EPI>
__global__ void __launch_bounds__(512) gemm_skinny_kernel(const GemmArgs g_in) {
    constexpr int NW = 8;                                                // waves = K slices of the workgroup (two per SIMD: twice the loads in flight)
    __shared__ __attribute__((aligned(16))) float red[4][4][16][64];   // [wave pair][row fragment][j * 4 + r][lane]: 64 KiB
    __shared__ __attribute__((aligned(16))) float2 rope_tab[64 * 16];  // gemm_epilogue<., 4>'s (sin, cos) table
    GemmArgs g = g_in;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * 64, m0 = g.m_lo + (int)blockIdx.y * 64;
    const int ksp = (int)blockIdx.z, nks = (int)gridDim.z;
    const int frow = lane & 15, fg = lane >> 4;
    const unsigned short *A = reinterpret_cast<const unsigned short *>(g.A);
    const unsigned short *W = reinterpret_cast<const unsigned short *>(g.W);
    const unsigned short *pa[4], *pw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra_ = min(m0 + i * 16 + frow, g.M - 1);           // rows past M: clamped duplicates, dropped by the epilogue
        const size_t arow = (size_t)(ra_ / g.a_grp_in) * g.a_grp_out + g.a_grp_off + (ra_ % g.a_grp_in);
        pa[i] = A + arow * g.lda + fg * 8;
        pw[i] = W + (size_t)min(n0 + i * 16 + frow, g.N - 1) * g.ldw + fg * 8;
    }
    f4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    // 32-k blocks (128 bytes of a packed or f32 row) of this workgroup's K slice, cut into NW contiguous wave ranges
    const int kb_all = g.K / 64, kb_per_wg = (kb_all + nks - 1) / nks;
    const int kb_lo = ksp * kb_per_wg, kb_hi = min(kb_all, kb_lo + kb_per_wg);
    const int per = (kb_hi - kb_lo + NW - 1) / NW;
    const int kb0 = kb_lo + wid * per, kb1 = min(kb_hi, kb0 + per);
    const bool apk = BF16 != kDtSplit || g.a_packed != 0;    // (16-bit classes: rows are fragment-ready as they are; a block = two 32-wide k-steps)
    int kb = kb0;
    for (; kb + 2 <= kb1; kb += 2) {
        uint4 fa[2][4][2], fb[2][4][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fb[u][i][0] = *reinterpret_cast<const uint4 *>(pw[i] + (kb + u) * 64);
                fb[u][i][1] = *reinterpret_cast<const uint4 *>(pw[i] + (kb + u) * 64 + 32);
                fa[u][i][0] = *reinterpret_cast<const uint4 *>(pa[i] + (kb + u) * 64);
                fa[u][i][1] = *reinterpret_cast<const uint4 *>(pa[i] + (kb + u) * 64 + 32);
            }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (!apk) split8(fa[u][i][0], fa[u][i][1]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mma2<BF16>(fb[u][j][0], fb[u][j][1], fa[u][i][0], fa[u][i][1], acc[i][j]);
            }
    }
    for (; kb < kb1; ++kb) {
        uint4 fa[4][2], fb[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fb[i][0] = *reinterpret_cast<const uint4 *>(pw[i] + kb * 64);
            fb[i][1] = *reinterpret_cast<const uint4 *>(pw[i] + kb * 64 + 32);
            fa[i][0] = *reinterpret_cast<const uint4 *>(pa[i] + kb * 64);
            fa[i][1] = *reinterpret_cast<const uint4 *>(pa[i] + kb * 64 + 32);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!apk) split8(fa[i][0], fa[i][1]);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mma2<BF16>(fb[j][0], fb[j][1], fa[i][0], fa[i][1], acc[i][j]);
        }
    }
    // ---- the eight K slices meet in two steps through one 64 KiB image: waves 4..7 hand their tiles to waves 0..3, which add them to
    // their own and publish the four pair sums; wave w < 4 then sums row fragment w (waves 4..7 run the epilogue on no rows: its
    // workgroup barrier -- the RoPE table -- needs every wave)
    if (wid >= 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wid - 4][i][j * 4 + r][lane] = acc[i][j][r];
    }
    __syncthreads();
    if (wid < 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wid][i][j * 4 + r][lane] += acc[i][j][r];
    }
    __syncthreads();
    f4 sum[1][4];
    const int wf = wid & 3;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            sum[0][j][r] = (red[0][wf][j * 4 + r][lane] + red[1][wf][j * 4 + r][lane]) + (red[2][wf][j * 4 + r][lane] + red[3][wf][j * 4 + r][lane]);
    if (ksp > 0) g.bias = nullptr;                      // split-K over workgroups: the bias belongs to slice 0
    g.ksplit = nks;                                     // > 1: the epilogue adds its (gated) partial sums into out with f32 atomics
    gemm_epilogue<BF16, EPI, 1>(g, sum, wid < 4 ? m0 + wid * 16 : g.M, n0, rope_tab, wid, lane);
}

template <int BF16>
int launch_skinny(const GemmArgs &g, int epi, hipStream_t stream) {
    const int rows = g.M - g.m_lo, gy = vs::cdiv(rows, 64), gx = vs::cdiv(g.N, 64);
    // epilogue 2 with the residual already in `out`: split K over workgroups while the grid is small and a slice keeps >= 8 blocks per wave
    int ks = 1;
    if (epi == 2 && !g.resid && !vs::deterministic())
        while (ks < 8 && gx * gy * ks * 2 <= 384 && g.K / 64 / (ks * 2) >= 32) ks *= 2;
    dim3 grid(gx, gy, ks), block(512);
    switch (epi) {
        case 0: hipLaunchKernelGGL((gemm_skinny_kernel<BF16, 0>), grid, block, 0, stream, g); break;
        case 1: hipLaunchKernelGGL((gemm_skinny_kernel<BF16, 1>), grid, block, 0, stream, g); break;
        case 2: hipLaunchKernelGGL((gemm_skinny_kernel<BF16, 2>), grid, block, 0, stream, g); break;
        case 3: hipLaunchKernelGGL((gemm_skinny_kernel<BF16, 3>), grid, block, 0, stream, g); break;
        case 4: hipLaunchKernelGGL((gemm_skinny_kernel<BF16, 4>), grid, block, 0, stream, g); break;
        case 5: if constexpr (BF16 == kDtSplit) { hipLaunchKernelGGL((gemm_skinny_kernel<BF16, 5>), grid, block, 0, stream, g); break; }
                vs::set_error("epilogue 5 (x GELU') is a split-class epilogue"); return -1;
        default: vs::set_error("vs_gemm_bias_act: unknown epilogue %d", epi); return -1;
    }
    return 0;
}

template <int BF16, int NW, int MF>
int launch_smallm_nw(const GemmArgs &g, int epi, hipStream_t stream) {
    dim3 grid(vs::cdiv(g.N, 16), vs::cdiv(g.M - g.m_lo, 16 * MF)), block(64 * NW);
    switch (epi) {
        case 0: hipLaunchKernelGGL((gemm_smallm_kernel<BF16, 0, NW, MF>), grid, block, 0, stream, g); break;
        case 1: hipLaunchKernelGGL((gemm_smallm_kernel<BF16, 1, NW, MF>), grid, block, 0, stream, g); break;
        case 2: hipLaunchKernelGGL((gemm_smallm_kernel<BF16, 2, NW, MF>), grid, block, 0, stream, g); break;
        case 3: hipLaunchKernelGGL((gemm_smallm_kernel<BF16, 3, NW, MF>), grid, block, 0, stream, g); break;
        default: vs::set_error("vs_gemm_bias_act: unknown epilogue %d", epi); return -1;
    }
    return 0;
}

template <int BF16>
int launch_smallm(const GemmArgs &g, int epi, hipStream_t stream) {
    // 16 waves split K when it is long (each walks <= 8 k-steps of 4096), 8 otherwise; 1 or 4 row fragments
    const bool one = g.M - g.m_lo <= 16, deep = g.K / 32 >= 64;
    if (one) return deep ? launch_smallm_nw<BF16, 16, 1>(g, epi, stream) : launch_smallm_nw<BF16, 8, 1>(g, epi, stream);
    return deep ? launch_smallm_nw<BF16, 16, 4>(g, epi, stream) : launch_smallm_nw<BF16, 8, 4>(g, epi, stream);
}

template <int BF16, int MI>
int launch_mi(const GemmArgs &g, int epi, hipStream_t stream) {
    const int nwg = vs::cdiv(g.M - g.m_lo, 32 * MI) * vs::cdiv(g.N, BN);
    dim3 grid(nwg, epi == 2 && g.ksplit > 1 ? g.ksplit : 1), block(256);
    if (epi == 2 && g.ntaps > 0) grid = dim3((unsigned)nwg * g.ntaps * g.ksplit, 1);
    switch (epi) {
        case 0: hipLaunchKernelGGL((gemm_kernel<BF16, 0, MI>), grid, block, 0, stream, g); break;
        case 1: hipLaunchKernelGGL((gemm_kernel<BF16, 1, MI>), grid, block, 0, stream, g); break;
        case 2: hipLaunchKernelGGL((gemm_kernel<BF16, 2, MI>), grid, block, 0, stream, g); break;
        case 3: hipLaunchKernelGGL((gemm_kernel<BF16, 3, MI>), grid, block, 0, stream, g); break;
        case 4: hipLaunchKernelGGL((gemm_kernel<BF16, 4, MI>), grid, block, 0, stream, g); break;
        case 5: if constexpr (BF16 == kDtSplit) { hipLaunchKernelGGL((gemm_kernel<BF16, 5, MI>), grid, block, 0, stream, g); break; }
                vs::set_error("epilogue 5 (x GELU') is a split-class epilogue"); return -1;
        default: vs::set_error("vs_gemm_bias_act: unknown epilogue %d", epi); return -1;
    }
    return 0;
}

// 7x7 RGB stem as a window GEMM on the 256x256 kernel's main loop: a K-tile (64) is TWO kernel rows of 32 halfs (21 used), i.e. the
// two 64-byte halves of a staged A row come from two consecutive padded image rows; K = 8 rows x 32 (the eighth row of the packed
// weights is zero -- its image data is finite and costs nothing).  Same staging units, swizzle and epilogue as gemm256_kernel.
struct WindowStager256 {
    const unsigned short *pa[2][2], *pw[2][2];
    long long kstep;   // A elements per K-tile = 2 padded image rows
    __device__ __forceinline__ void stage(int u, int kt, unsigned lds) const {
        if (u < 2) {
            glds16(pa[u][0] + kt * kstep, lds);
            glds16(pa[u][1] + kt * kstep, lds + 1024u);
        } else {
            glds16(pw[u - 2][0] + kt * 64, lds);
            glds16(pw[u - 2][1] + kt * 64, lds + 1024u);
        }
    }
};

// Epilogue of the split-class 7x7 stem fused with the Gaussian-parameter head's "upsample + add" (dpt_gs_head.py:142-150; round 4):
//     out[pixel] = packed( bilinear_x2(trunk)[pixel] + relu(conv7x7(image)[pixel] + bias) )
// i.e. what vs_upsample2x_nhwc(trunk, add = stem, relu_add, packed) computed from the stem's f32 output -- 12.9 GB written by this kernel and
// read again by that one per 24-scene step, plus the trunk read; here the stem's 256 x 256 f32 map never reaches HBM.  g.gate = the
// low-resolution trunk [Nimg, conv_H, conv_W, N] f32 (NHWC), the output image is (2 conv_H) x (2 conv_W).  The interpolation is the
// expression of upsample2x_f32_block_kernel (align_corners = True: source = destination * (Hs - 1) / (H - 1), taps clamped to the last
// row / column), evaluated per output; a wave's 128 pixels x 64 channels touch a 2 x 65 x 64 slab of the trunk (33 KB: vector-L1 resident),
// so the four taps of an output cost L1 hits, not L2 traffic.
// Two column fragments (j, j + 1: one 16-byte hi chunk and one lo chunk of the packed row) of one pixel from their four taps
__device__ __forceinline__ void stem_upadd_pair(const GemmArgs &g, const f4 &acc0, const f4 &acc1, const float (&bv0)[4], const float (&bv1)[4],
                                                const float4 (&t00)[2], const float4 (&t01)[2], const float4 (&t10)[2], const float4 (&t11)[2], float lx,
                                                float ly, float *rowp, int ncol) {
    float v[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float a00[4] = {t00[j].x, t00[j].y, t00[j].z, t00[j].w}, a01[4] = {t01[j].x, t01[j].y, t01[j].z, t01[j].w};
        const float a10[4] = {t10[j].x, t10[j].y, t10[j].z, t10[j].w}, a11[4] = {t11[j].x, t11[j].y, t11[j].z, t11[j].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float top = a00[r] * (1.f - lx) + a01[r] * lx, bot = a10[r] * (1.f - lx) + a11[r] * lx;
            const float st = fmaxf((j ? acc1[r] : acc0[r]) * g.acc_scale + (j ? bv1[r] : bv0[r]), 0.f);
            v[j][r] = top * (1.f - ly) + bot * ly + st;
        }
    }
    store_split8(rowp, ncol, v[0], v[1]);
}

// The taps are gathered straight from global memory: a wave's 128 pixels x 64 channels touch a 2 x 65 x 64 slab of the trunk (33 KB), eight
// waves thrash the 32 KB vector L1, so the epilogue runs at L2 latency: +3.8 ms per 24-scene step on this 6 ms kernel against the 5.9 ms of
// the stand-alone upsample-add launch it replaces (and 25 GB less HBM traffic).  Measured and dropped: staging the tile's two source rows in
// the released LDS ring in two channel halves (LDS-DMA, swizzled, conflict-free gathers) -- each trunk element then crosses L2 -> CU once per
// tile, but the two halves serialise (DMA, barrier, half the waves gather) and the kernel was 0.6 ms SLOWER than the global gather.
__device__ __forceinline__ void stem_upadd_epilogue(const GemmArgs &g, f4 (&acc)[8][4], int m0, int wr, int wc, unsigned char *smem, int tid, int lane) {
    asm volatile("" : "+v"(lane));
    const int mrow = lane & 15, c4 = (lane >> 4) * 4;
    const int mw0 = m0 + wr * 128, nbase = wc * 64;       // (N == 256: one column tile)
    const float *__restrict__ trunk = g.gate;
    const int Hs = g.conv_H, Ws = g.conv_W, H = 2 * Hs, W = 2 * Ws, C = g.N;
    const float ry = (float)(Hs - 1) / (float)(H - 1), rx = (float)(Ws - 1) / (float)(W - 1);
    float bv[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 t = *reinterpret_cast<const float4 *>(g.bias + nbase + j * 16 + c4);
        bv[j][0] = t.x; bv[j][1] = t.y; bv[j][2] = t.z; bv[j][3] = t.w;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = mw0 + i * 16 + mrow;
        if (m >= g.M) continue;
        const int n = m / (H * W), rem = m - n * (H * W), yo = rem / W, xo = rem - yo * W;
        const float sy = (float)yo * ry, sx = (float)xo * rx;
        const int y0 = min((int)sy, Hs - 1), x0 = min((int)sx, Ws - 1);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        const int y1 = min(y0 + 1, Hs - 1), x1 = min(x0 + 1, Ws - 1);
        const float *r0 = trunk + ((size_t)(n * Hs + y0) * Ws) * C + nbase + c4, *r1 = trunk + ((size_t)(n * Hs + y1) * Ws) * C + nbase + c4;
        float *rowp = reinterpret_cast<float *>(g.out) + (size_t)m * g.ldo;
#pragma unroll
        for (int jp = 0; jp < 4; jp += 2) {
            float4 t00[2], t01[2], t10[2], t11[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                t00[j] = *reinterpret_cast<const float4 *>(r0 + (size_t)x0 * C + (jp + j) * 16);
                t01[j] = *reinterpret_cast<const float4 *>(r0 + (size_t)x1 * C + (jp + j) * 16);
                t10[j] = *reinterpret_cast<const float4 *>(r1 + (size_t)x0 * C + (jp + j) * 16);
                t11[j] = *reinterpret_cast<const float4 *>(r1 + (size_t)x1 * C + (jp + j) * 16);
            }
            stem_upadd_pair(g, acc[i][jp], acc[i][jp + 1], bv[jp], bv[jp + 1], t00, t01, t10, t11, lx, ly, rowp, nbase + c4 + jp * 16);
        }
    }
}

template <int BF16, bool UPADD = false>
__global__ void __launch_bounds__(512, 1) conv7x7_256_kernel(const GemmArgs g) {
    constexpr int BM2 = 256, BN2 = 256;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[kLdsBytes256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int tiles_n = (g.N + BN2 - 1) / BN2;
    const int nwg = ((g.M + BM2 - 1) / BM2) * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * BM2, n0 = tn * BN2;
    const unsigned short *A = reinterpret_cast<const unsigned short *>(g.A);
    const unsigned short *W = reinterpret_cast<const unsigned short *>(g.W);
    WindowStager256 st;
    // split operands: the image is f32, a 128-byte staged row is ONE kernel row (32 floats, 21 used) and K = 8 rows = 8 K-tiles
    constexpr bool SPLIT = BF16 == kDtSplit;
    st.kstep = SPLIT ? (long long)g.a_kstride : 2ll * g.a_kstride;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = unit_row256(wid, j, lane);
        const int c = unit_src_chunk256(q, lane);   // 16-byte chunk of the 128-byte staged row: 0..3 kernel row 2kt, 4..7 kernel row 2kt + 1
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ra_ = min(m0 + unit_a_tile_row256(q, h), g.M - 1);
            const int rg_ = ra_ / g.a_grp_in;
            const size_t arow = (size_t)rg_ * g.a_grp_out + (ra_ % g.a_grp_in) + (size_t)(rg_ / g.a_sup_in) * g.a_sup_extra;
            st.pa[h][j] = SPLIT ? A + arow * g.lda + c * 8 : A + arow * g.lda + (size_t)(c >> 2) * g.a_kstride + (c & 3) * 8;
            const int rw_ = min(n0 + unit_b_tile_row256(q, h), g.N - 1);
            st.pw[h][j] = W + (size_t)rw_ * g.ldw + c * 8;
        }
    }
    f4 acc[8][4];
    if constexpr (SPLIT) mainloop256_split<false>(st, 8, acc, smem, lane, wid);
    else mainloop256<BF16, false>(st, 4, acc, smem, lane, wid);
    if constexpr (SPLIT && UPADD) {
        {               // fused "+ bilinear x2 of the trunk, ReLU, packed output": vs_conv7x7_rgb_split_up_nhwc (its own instantiation:
                        // the tap registers of this epilogue beside the 128 accumulators must not cost the plain stem kernel anything)
            stem_upadd_epilogue(g, acc, m0, wr, wc, smem, tid, lane);      // (Cout == 256 is required by the entry: n0 == 0)
            return;
        }
    }
    gemm_epilogue<BF16, 0, 8>(g, acc, m0 + wr * 128, n0 + wc * 64, smem, wid, lane);
}

template <int BF16>
int launch_256(const GemmArgs &g, int epi, hipStream_t stream) {
    const int nwg = vs::cdiv(g.M - g.m_lo, 256) * vs::cdiv(g.N, 256);
    dim3 grid(nwg), block(512);
    if constexpr (BF16 == kDtSplit) {
        if (g.a_packed) {
            switch (epi) {
                case 0: hipLaunchKernelGGL((gemm256_kernel<BF16, 0, true>), grid, block, 0, stream, g); break;
                case 1: hipLaunchKernelGGL((gemm256_kernel<BF16, 1, true>), grid, block, 0, stream, g); break;
                case 2: hipLaunchKernelGGL((gemm256_kernel<BF16, 2, true>), grid, block, 0, stream, g); break;
                case 3: hipLaunchKernelGGL((gemm256_kernel<BF16, 3, true>), grid, block, 0, stream, g); break;
                case 4: hipLaunchKernelGGL((gemm256_kernel<BF16, 4, true>), grid, block, 0, stream, g); break;
                default: vs::set_error("vs_gemm_bias_act: unknown epilogue %d", epi); return -1;
            }
            return 0;
        }
    }
    switch (epi) {
        case 0: hipLaunchKernelGGL((gemm256_kernel<BF16, 0>), grid, block, 0, stream, g); break;
        case 1: hipLaunchKernelGGL((gemm256_kernel<BF16, 1>), grid, block, 0, stream, g); break;
        case 2: hipLaunchKernelGGL((gemm256_kernel<BF16, 2>), grid, block, 0, stream, g); break;
        case 3: hipLaunchKernelGGL((gemm256_kernel<BF16, 3>), grid, block, 0, stream, g); break;
        case 4: hipLaunchKernelGGL((gemm256_kernel<BF16, 4>), grid, block, 0, stream, g); break;
        case 5: if constexpr (BF16 == kDtSplit) { hipLaunchKernelGGL((gemm256_kernel<BF16, 5>), grid, block, 0, stream, g); break; }
                vs::set_error("epilogue 5 (x GELU') is a split-class epilogue"); return -1;
        default: vs::set_error("vs_gemm_bias_act: unknown epilogue %d", epi); return -1;
    }
    return 0;
}

// Rows [g.M - rem, g.M) of a split GEMM: <= 64 rows on the weight-streaming kernel, <= 128 on one row of 128x128 tiles.
template <int BF16>
int launch_tail(const GemmArgs &g, int rem, int epi, hipStream_t stream) {
    GemmArgs t = g;
    t.m_lo = g.M - rem;
    // a tail is a handful of tiles walking all of K serially: for the f32 residual epilogue split K over up to 8 workgroups
    // per tile (>= 8 k-steps each) and let the partial sums meet through f32 atomics
    static const int no_ksplit = [] { const char *e = getenv("VS_GEMM_NO_KSPLIT"); return e ? atoi(e) : 0; }();
    if (epi == 2 && rem > 64 && !no_ksplit && !g.resid && !vs::deterministic()) {   // (split-K partial sums meet in out: it must already hold the residual)
        const int tiles = vs::cdiv(rem, 128) * vs::cdiv(g.N, BN);
        int ks = 1;
        while (ks < 8 && (g.K / 32) % (ks * 2) == 0 && g.K / 32 / (ks * 2) >= 8 && tiles * ks * 2 <= 512) ks *= 2;
        t.ksplit = ks;
    }
    // (the RoPE epilogue pairs columns 16 apart and a packed output is written in 32-column blocks: only the tile kernels hold those in one
    // workgroup).  Round 4: the weight-streaming kernel takes row groups of 64 (blockIdx.y) and packed A rows, so VS_GEMM_TAIL_SMALLM=256 sends
    // the 192-row tails of the bench step to it (plain stores of whole-K sums instead of the K-split tiles' f32 atomics).  Measured same-box:
    // 246.6 / 248.3 vs 247.6 / 247.7 ms per step -- no gain (A is re-read by each of the N / 16 column workgroups); the default stays 64.
    static const int tail_rows = [] { const char *e = getenv("VS_GEMM_TAIL_SMALLM"); return e ? atoi(e) : 64; }();
    // round 5: tails of 65 .. 256 rows of the split class run on the skinny kernel (every epilogue, packed A / packed output included)
    static const int skinny = [] { const char *e = getenv("VS_GEMM_SKINNY"); return e ? atoi(e) : 1; }();
    if constexpr (BF16 == kDtSplit || BF16 == 0 || BF16 == 1) {   // (split class and, since late round 5, the 16-bit classes)
        // (<= 64 rows stay on the weight-streaming kernel where it applies; the RoPE epilogue and packed outputs, which it does not have, come here)
        if (skinny && rem <= 256 && g.K % 64 == 0 && (rem > 64 || skinny == 2 || epi == 4 || epi == 5 || g.out_packed) && (BF16 == kDtSplit || (!g.a_packed && !g.out_packed))) {
            t.ksplit = 1;
            return launch_skinny<BF16>(t, epi, stream);
        }
    }
    if (rem <= (BF16 == kDtSplit ? tail_rows : 64) && epi != 4 && epi != 5 && !g.out_packed && (BF16 == kDtSplit || !g.a_packed)) {
        t.ksplit = 1;
        return launch_smallm<BF16>(t, epi, stream);
    }
    return launch_mi<BF16, 4>(t, epi, stream);
}

// reference-precision path (exact f32 MFMA at 1/16 of the 16-bit rate): the matrix pipe, not the tile schedule, sets the time, so
// there is one route per shape class and no tail splitting: 256x256 tiles when K allows (K in 2-byte units: a multiple of
// 128 = 64 floats), 128x128 tiles otherwise, the weight-streaming kernel for the camera-token GEMMs
int launch_f32(const GemmArgs &g, int epi, hipStream_t stream) {
    if (g.M <= 64 && epi != 4) return launch_smallm<kDtF32>(g, epi, stream);
    if (g.K % 128 == 0) return launch_256<kDtF32>(g, epi, stream);
    return launch_mi<kDtF32, 4>(g, epi, stream);
}

template <int BF16>
int launch(const GemmArgs &g, int epi, hipStream_t stream) {
    // VS_GEMM_MI = 4 | 8 | 16 forces the 128x128 | 256x128 | 256x256 kernel (benchmarks, tests).
    static const int force = [] { const char *e = getenv("VS_GEMM_MI"); return e ? atoi(e) : 0; }();
    if (g.M <= 64 && force == 0 && epi != 4 && epi != 5 && !g.out_packed && (BF16 == kDtSplit || !g.a_packed)) return launch_smallm<BF16>(g, epi, stream);
    if constexpr (BF16 == kDtSplit || BF16 == 0 || BF16 == 1) {   // camera-token GEMMs and other launches of <= 256 rows: the skinny kernel (round 5)
        static const int skinny = [] { const char *e = getenv("VS_GEMM_SKINNY"); return e ? atoi(e) : 1; }();
        if (skinny && force == 0 && g.M - g.m_lo <= 256 && g.K % 64 == 0 && g.ntaps == 0 && !g.partials && g.ksplit <= 1 && g.a_sup_extra == 0 && g.a_kstride == 32 &&
            (BF16 == kDtSplit || (!g.a_packed && !g.out_packed)))
            return launch_skinny<BF16>(g, epi, stream);
    }
    if (force == 8) return launch_mi<BF16, 8>(g, epi, stream);
    if (force == 4) return launch_mi<BF16, 4>(g, epi, stream);
    // 256x256 tiles run one 8-wave workgroup per CU, i.e. in rounds of 256 tiles, and a partly filled round costs as much
    // as a full one (M = frames * 257 tokens never divides).  So the big tiles only get as many rows as fill whole rounds;
    // the remaining rows are finished by a launch of small tiles (128x128, 3 workgroups per CU) or, for <= 64 rows, the
    // weight-streaming kernel -- a short round instead of a mostly idle long one.
    if (g.K % 128 == 0 && g.M >= 256) {
        const int tn = vs::cdiv(g.N, 256), mt = g.M / 256;
        const long long full = (long long)mt * tn, rounds_all = (vs::cdiv(g.M, 256) * (long long)tn + 255) / 256;
        int mt_main = mt;
        if (full > 256 && full % 256 != 0) mt_main = (int)((full / 256) * 256 / tn);  // whole rounds only
        const int rows_main = mt_main * 256, rem = g.M - rows_main;
        // cost in units of one 256x256 round; a 128x128 round (768 tiles) is about 0.31 of it
        const long long tiles4 = (long long)vs::cdiv(rem, 128) * vs::cdiv(g.N, 128);
        const double cost_split = (double)(((long long)mt_main * tn + 255) / 256) + (rem > 0 ? 0.31 * (double)((tiles4 + 767) / 768) : 0.0);
        const long long tiles_one = vs::cdiv(g.M, 256) * (long long)tn;
        // a last round that is >= 95 % full is not worth a second launch (e.g. 31 eight-view scenes: 249 row tiles -> 3.89 / 11.67 /
        // 15.56 rounds): one launch, no tail
        const bool nearly_full = tiles_one * 100 >= rounds_all * 256 * 95;
        const bool split = !nearly_full && rem > 0 && mt_main > 0 && cost_split < (double)rounds_all - 0.05;
        // (a round that is >= 70 % full still beats the 256x128 tiles: 8 scenes x 257 tokens x 768 columns = 195 tiles, 385 vs 391 ms per
        // split-class training step)
        if (force == 16 || split || tiles_one * 100 >= rounds_all * 256 * 70) {
            GemmArgs main_g = g;
            if (split) main_g.M = rows_main;
            int rc = launch_256<BF16>(main_g, epi, stream);
            if (rc || !split) return rc;
            return launch_tail<BF16>(g, rem, epi, stream);
        }
    }
    // 256x128 tiles (half the W-panel traffic and LDS-DMA issue per flop of 128x128) whenever there are enough of them
    // to fill 256 CUs x 2 resident workgroups; 128x128 otherwise.
    const int tiles_n = vs::cdiv(g.N, BN);
    const long long t8 = (long long)vs::cdiv(g.M, 256) * tiles_n;
    const int mi = t8 >= 256 ? 8 : 4, bm = 32 * mi, slots = 256 * (mi == 8 ? 2 : 3);
    const int rem = g.M % bm;
    const long long full = (long long)(g.M / bm) * tiles_n;
    // a partly filled extra round costs a whole one.  Round 6: the matrix pipes of a CU are shared by its resident workgroups, so a "round"
    // of MFMA-bound tiles is 256 tiles (one per CU), not `slots` -- one 8-view scene's fc1 (2 056 rows x 4 096 columns) is 288 tiles of
    // 256 x 128, i.e. TWO rounds with the second 12 % full; peeling the 8 leftover rows makes it one (110 -> 70 us).
    const bool split = full > 0 && rem > 0 && rem <= bm / 2 &&
                       ((full + tiles_n + slots - 1) / slots > (full + slots - 1) / slots || (full + tiles_n + 255) / 256 > (full + 255) / 256);
    GemmArgs main_g = g;
    if (split) main_g.M = g.M - rem;
    // Round 6, small batches (one 8-view scene = 2 056 rows): the residual epilogue's GEMMs (attention projection, fc2: N = 1024 / 768) are
    // 136 / 102 tiles of 128 x 128 -- half the chip for one pass over K.  64-row tiles (MI = 2: the same kernel, each wave a 32 x 64 slab)
    // double the workgroups: VS_GEMM_MI2=0 for the A/B.
    if constexpr (BF16 == kDtSplit) {
        static const int mi2 = [] { const char *e = getenv("VS_GEMM_MI2"); return e ? atoi(e) : 1; }();
        const long long tiles4 = (long long)vs::cdiv(main_g.M - main_g.m_lo, 128) * tiles_n;
        if (mi2 && mi == 4 && epi == 2 && g.ntaps == 0 && g.ksplit <= 1 && !g.partials && tiles4 <= 192 && main_g.M - main_g.m_lo > 64) {
            const long long nwg = (long long)vs::cdiv(main_g.M - main_g.m_lo, 64) * tiles_n;
            hipLaunchKernelGGL((gemm_kernel<kDtSplit, 2, 2>), dim3((unsigned)nwg, 1), dim3(256), 0, stream, main_g);
            if (!split) return 0;
            return launch_tail<BF16>(g, rem, epi, stream);
        }
    }
    // (Round 6, measured and not kept: cutting K over blockIdx.y for the 136-tile residual-epilogue GEMMs of a ONE-scene batch -- 2 056 rows,
    // N = 1024: half the chip for one pass over K -- until the grid fills the chip.  The partial sums meet through f32 atomics and the
    // epilogue's atomic traffic costs more than the idle CUs: B = 1 encoder 22.2 -> 26.2 ms, B = 2 29.7 -> 35.3 ms.)
    int rc = mi == 8 ? launch_mi<BF16, 8>(main_g, epi, stream) : launch_mi<BF16, 4>(main_g, epi, stream);
    if (rc || !split) return rc;
    return launch_tail<BF16>(g, rem, epi, stream);
}

// XCD-aware logical block ids in the split-K / tap-fused weight-gradient kernels (gemm256.h splitk_logical_block): VS_WGRAD_XCD=0 restores the
// plain blockIdx order for an A/B.
static int wgrad_xcd() {
    static const int v = [] { const char *e = getenv("VS_WGRAD_XCD"); return e ? atoi(e) : 1; }();
    return v;
}

// out[t][m, n] += sum_s partials[((s * ntaps + t) * M + m) * N + n]: second stage of a weight-gradient GEMM with a workspace.
template <int VEC>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float *__restrict__ ws, float *__restrict__ out, int M, int N, int ntaps,
                                                            int ks, long long ldo, long long tap_out_stride, int accumulate) {
    const int Nv = N / VEC;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long per_tap = (long long)M * Nv;
    if (idx >= per_tap * ntaps) return;
    const int t = (int)(idx / per_tap);
    const long long r = idx - t * per_tap;
    const int m = (int)(r / Nv), n = (int)(r - (long long)m * Nv) * VEC;
    const long long slice = (long long)ntaps * M * N;
    const float *p = ws + ((long long)t * M + m) * N + n;
    float *o = out + t * tap_out_stride + (long long)m * ldo + n;
    if constexpr (VEC == 4) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int s = 0; s < ks; ++s) {
            const float4 v = *reinterpret_cast<const float4 *>(p + s * slice);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        if (accumulate) {
            const float4 c = *reinterpret_cast<float4 *>(o);
            a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w;
        }
        *reinterpret_cast<float4 *>(o) = a;
    } else {
        float a = accumulate ? *o : 0.f;
        for (int s = 0; s < ks; ++s) a += p[s * slice];
        *o = a;
    }
}

// out[n, m] (+)= sum_s ws[(s * M + m) * N + n]: the second stage writing the TRANSPOSE of the GEMM's [M, N] result (32 x 32 tiles through LDS) --
// for a weight gradient computed as dW^T = X^T dY (vs_gemm_wgrad_split_atn, transpose_out).  M, N multiples of 32.
__global__ void __launch_bounds__(256) splitk_reduce_t_kernel(const float *__restrict__ ws, float *__restrict__ out, int M, int N, int ks, long long ldo,
                                                              int accumulate) {
    __shared__ float t[32][33];
    const int tiles_n = N >> 5;
    const int m0 = (int)(blockIdx.x / tiles_n) * 32, n0 = (int)(blockIdx.x % tiles_n) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long long slice = (long long)M * N;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float *p = ws + (long long)(m0 + ty + 8 * r) * N + n0 + tx;
        float a = 0.f;
        for (int s_ = 0; s_ < ks; ++s_) a += p[s_ * slice];
        t[ty + 8 * r][tx] = a;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float *o = out + (long long)(n0 + ty + 8 * r) * ldo + m0 + tx;
        const float a = t[tx][ty + 8 * r];
        *o = accumulate ? *o + a : a;
    }
}

// Weight-gradient launches (split-K, optionally tap-fused): 256x256 tiles on the phase-interleaved main loop when the
// output is made of whole 256-tiles and every K slice is an even number (>= 2) of 64-wide K tiles; 128x128 tiles otherwise.
// With a workspace of ksplit * ntaps * M * N floats the slices store partial tiles and a second kernel sums them; without
// one they meet through f32 atomics.
template <int BF16>
int launch_wgrad(GemmArgs &g, int ksplit, float *ws, long long ws_bytes, int accumulate, hipStream_t stream) {
    static const int no256 = [] { const char *e = getenv("VS_WGRAD_NO256"); return e ? atoi(e) : 0; }();
    const int ntaps = g.ntaps > 0 ? g.ntaps : 1;
    const bool big = !no256 && g.M % 256 == 0 && g.N % 256 == 0 && g.K % (128 * ksplit) == 0;
    const int slices = big ? ksplit : (ksplit > 1 ? ksplit : 2);  // gemm_kernel always runs >= 2 slices in this mode
    const long long need = (long long)slices * ntaps * g.M * g.N * (long long)sizeof(float);
    g.partials = nullptr;
    if (ws) {
        if (ws_bytes < need) { vs::set_error("weight-gradient GEMM: workspace of %lld bytes given, %lld needed", ws_bytes, need); return -1; }
        if ((reinterpret_cast<uintptr_t>(ws) & 15) != 0) { vs::set_error("weight-gradient GEMM: workspace must be 16-byte aligned"); return -1; }
        g.partials = ws;
    }
    int rc = 0;
    if (big) {
        g.ksplit = ksplit;
        const long long nwg = (long long)(g.M / 256) * (g.N / 256) * ntaps * ksplit;
        if (nwg > 0x7fffffffLL) { vs::set_error("weight-gradient GEMM: grid too large"); return -1; }
        hipLaunchKernelGGL((gemm256_splitk_kernel<BF16>), dim3((unsigned)nwg), dim3(512), 0, stream, g);
    } else {
        g.ksplit = slices;
        if (slices != ksplit && (g.a_slice_stride || g.w_slice_stride)) { vs::set_error("weight-gradient GEMM: slice-blocked operands need ksplit >= 2 on the 128x128 tiling"); return -1; }
        if (g.K % (32 * slices) != 0) { vs::set_error("weight-gradient GEMM: K=%d must be a multiple of %d", g.K, 32 * slices); return -1; }
        if (BF16 == kDtSplit && g.K % (64 * slices) != 0) { vs::set_error("weight-gradient GEMM (split operands): K=%d floats must be a multiple of %d (whole 32-float blocks per slice)", g.K / 2, 32 * slices); return -1; }
        rc = launch_mi<BF16, 4>(g, 2, stream);
    }
    if (rc || !ws) return rc;
    const bool v4 = g.N % 4 == 0 && g.ldo % 4 == 0 && g.tap_out_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(g.out) & 15) == 0;
    const long long items = (long long)ntaps * g.M * (v4 ? g.N / 4 : g.N);
    const dim3 grid((unsigned)((items + 255) / 256));
    if (v4) hipLaunchKernelGGL(splitk_reduce_kernel<4>, grid, dim3(256), 0, stream, ws, (float *)g.out, g.M, g.N, ntaps, slices, (long long)g.ldo, g.tap_out_stride, accumulate);
    else hipLaunchKernelGGL(splitk_reduce_kernel<1>, grid, dim3(256), 0, stream, ws, (float *)g.out, g.M, g.N, ntaps, slices, (long long)g.ldo, g.tap_out_stride, accumulate);
    return 0;
}

}  // namespace

namespace {
int gemm_entry(const char *fn, const void *A, const void *W, const float *bias, void *out, const float *gate, const float *resid, int32_t M, int32_t N,
               int32_t K, int32_t lda, int32_t ldw, int32_t ldo, int32_t epilogue, int32_t dtype, int32_t grp_in, int32_t grp_out,
               int32_t grp_off, int32_t gate_rows, int32_t gate_ld, int32_t a_grp_in, int32_t a_grp_out, int32_t a_grp_off,
               const int32_t *rope_pos, const uint8_t *rope_kind, int32_t rope_C, float base2d, float theta1d, hipStream_t stream,
               float acc_scale = 1.f, int a_packed = 0, int out_packed = 0) {
    VS_CHECK(A && W && out, "%s: null pointer", fn);
    VS_CHECK(M >= 0 && N > 0 && K > 0, "%s: bad sizes M=%d N=%d K=%d", fn, M, N, K);
    VS_CHECK(dtype == 1 || dtype == 2 || dtype == 3 || dtype == 4, "%s: dtype must be 1 (f16), 2 (bf16), 3 (f32) or 4 (split)", fn);
    if (dtype == 3 || dtype == 4) {   // f32 operands are addressed in 2-byte units by the kernels (gemm_common.h, kDtF32 / kDtSplit): K, lda, ldw double
        VS_CHECK(K % 32 == 0, "%s: K=%d must be a multiple of 32 for f32 operands", fn, K);
        VS_CHECK(lda % 4 == 0 && ldw % 4 == 0, "%s: lda/ldw must be multiples of 4 floats (16-byte rows)", fn);
        K *= 2; lda *= 2; ldw *= 2;
    }
    VS_CHECK(K % 64 == 0, "%s: K=%d must be a multiple of 64", fn, K);
    VS_CHECK(lda % 8 == 0 && ldw % 8 == 0, "%s: lda/ldw must be multiples of 8 elements (16-byte rows)", fn);
    VS_CHECK((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0,
             "%s: A and W must be 16-byte aligned", fn);
    if (M == 0) return 0;
    GemmArgs g;
    g.A = A; g.W = W; g.bias = bias; g.out = out; g.gate = gate; g.resid = resid;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldo = ldo;
    g.grp_in = grp_in > 0 ? grp_in : (M > 0 ? M : 1);
    g.grp_out = grp_out > 0 ? grp_out : g.grp_in;
    g.grp_off = grp_off;
    g.gate_rows = gate_rows > 0 ? gate_rows : (M > 0 ? M : 1);
    g.gate_ld = gate_ld > 0 ? gate_ld : N;
    g.a_grp_in = a_grp_in > 0 ? a_grp_in : (M > 0 ? M : 1);
    g.a_grp_out = a_grp_out > 0 ? a_grp_out : g.a_grp_in;
    g.a_grp_off = a_grp_off;
    g.m_lo = 0;
    g.a_sup_in = 0x7fffffff; g.a_sup_extra = 0; g.a_kstride = 32; g.ksplit = 1; g.ntaps = 0; g.tap_out_stride = 0; g.partials = nullptr; g.a_slice_stride = 0; g.w_slice_stride = 0; g.k_valid = 0; g.conv_H = 0; g.conv_W = 0;
    g.rope_pos = rope_pos; g.rope_kind = rope_kind; g.rope_C = rope_C;
    { static const int stg = [] { const char *e = getenv("VS_GEMM_STAGGER"); return e ? atoi(e) : 0; }(); g.stagger = stg; }
    { static const int rb = [] { const char *e = getenv("VS_GEMM_ROW_BAND"); return e ? atoi(e) : 0; }(); g.row_band = rb; }
    g.rope_l2base = base2d > 0.f ? log2f(base2d) : 0.f;
    g.rope_l2theta = theta1d > 0.f ? log2f(theta1d) : 0.f;
    g.acc_scale = acc_scale;
    g.a_packed = a_packed;
    g.out_packed = out_packed;
    g.tap_on_a = 0;
    // split operands (kDtSplit) take the 16-bit classes' routing: whole rounds of 256 x 256 tiles + a tail launch (K counts 2-byte units
    // of the f32 rows; every stage pair / K-tile of the kernels is one 128-byte block of 32 k)
    const int rc = dtype == 4 ? launch<kDtSplit>(g, epilogue, stream)
                 : dtype == 3 ? launch_f32(g, epilogue, stream) : dtype == 2 ? launch<1>(g, epilogue, stream) : launch<0>(g, epilogue, stream);
    if (rc) return rc;
    VS_HIP(hipGetLastError());
    return 0;
}
}  // namespace

extern "C" int vs_gemm_bias_act(const void *A, const void *W, const float *bias, void *out, const float *gate, int32_t M,
                                int32_t N, int32_t K, int32_t lda, int32_t ldw, int32_t ldo, int32_t epilogue, int32_t dtype,
                                int32_t grp_in, int32_t grp_out, int32_t grp_off, int32_t gate_rows, int32_t gate_ld,
                                int32_t a_grp_in, int32_t a_grp_out, int32_t a_grp_off, vs_stream_t stream_) {
    VS_CHECK(epilogue >= 0 && epilogue <= 3, "vs_gemm_bias_act: unknown epilogue %d", epilogue);
    return gemm_entry("vs_gemm_bias_act", A, W, bias, out, gate, nullptr, M, N, K, lda, ldw, ldo, epilogue, dtype, grp_in, grp_out, grp_off,
                      gate_rows, gate_ld, a_grp_in, a_grp_out, a_grp_off, nullptr, nullptr, 0, 0.f, 0.f, (hipStream_t)stream_);
}

// out32[row(m), n] = resid[row(m), n] + (1 + gate[m / gate_rows, n]) * (A W^T + bias): epilogue 2 of vs_gemm_bias_act with the residual
// stream read from a second buffer (same layout as out), so a caller that must keep the old stream (training: the LayerNorm input
// is needed by the backward) does not clone it first.
extern "C" int vs_gemm_resid(const void *A, const void *W, const float *bias, const float *resid, float *out, const float *gate, int32_t M,
                             int32_t N, int32_t K, int32_t lda, int32_t ldw, int32_t ldo, int32_t dtype, int32_t gate_rows, int32_t gate_ld,
                             vs_stream_t stream_) {
    VS_CHECK(resid, "vs_gemm_resid: null resid");
    return gemm_entry("vs_gemm_resid", A, W, bias, out, gate, resid, M, N, K, lda, ldw, ldo, 2, dtype, 0, 0, 0, gate_rows, gate_ld, 0, 0, 0,
                      nullptr, nullptr, 0, 0.f, 0.f, (hipStream_t)stream_);
}

// Packed q|k|v projection with the rotary embedding of q and k applied in the epilogue (head_dim 64): replaces
// nn.Linear + RoPE2D / temporal RoPE of croco/blocks.py:94-104 and backbone_vica.py:95-118 in one pass over the output.
extern "C" int vs_gemm_qkv_rope(const void *A, const void *W, const float *bias, void *out, int32_t M, int32_t N, int32_t K,
                                int32_t lda, int32_t ldw, int32_t ldo, int32_t dtype, int32_t grp_in, int32_t grp_out,
                                int32_t grp_off, int32_t a_grp_in, int32_t a_grp_out, int32_t a_grp_off, const int32_t *pos,
                                const uint8_t *kind, int32_t C, float base2d, float theta1d, vs_stream_t stream_) {
    VS_CHECK(pos, "vs_gemm_qkv_rope: null pos");
    VS_CHECK(C > 0 && C % 64 == 0 && N >= 2 * C && N % 64 == 0, "vs_gemm_qkv_rope: need C %% 64 == 0, N %% 64 == 0, N >= 2C (C=%d N=%d)", C, N);
    VS_CHECK(base2d > 0.f && theta1d > 0.f, "vs_gemm_qkv_rope: base2d and theta1d must be positive");
    return gemm_entry("vs_gemm_qkv_rope", A, W, bias, out, nullptr, nullptr, M, N, K, lda, ldw, ldo, 4, dtype, grp_in, grp_out, grp_off, 0, 0,
                      a_grp_in, a_grp_out, a_grp_off, pos, kind, C, base2d, theta1d, (hipStream_t)stream_);
}

// ---- split operands (dtype code 4; gemm_common.h, kDtSplit): f32-class GEMM at a third of the 16-bit matrix rate ----
// W [N, K] f32 (row stride ldw floats, K a multiple of 32) -> packed [N, K] 4-byte units: per block of 32 k, 32 halves hi then 32 halves
// lo of w * 2^scale_exp, hi = rne16, lo = rne16(w 2^e - hi); inside a block chunk g (8 halves) holds k = {4g..4g+3, 16+4g..16+4g+3}.
namespace {
__global__ void __launch_bounds__(256) split_pack_kernel(const float *__restrict__ w, long long ldw, unsigned short *__restrict__ out, long long ldo,
                                                         int N, int K, float scale) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;       // one thread per (row, 4 consecutive k)
    const int kq = K / 4;
    if (idx >= (long long)N * kq) return;
    const int n = (int)(idx / kq), k4 = (int)(idx - (long long)n * kq) * 4;
    const float4 v = *reinterpret_cast<const float4 *>(w + n * ldw + k4);
    const float x[4] = {v.x * scale, v.y * scale, v.z * scale, v.w * scale};
    unsigned short h[4], l[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const _Float16 hh = (_Float16)x[r];
        const _Float16 ll = (_Float16)(x[r] - (float)hh);
        h[r] = __builtin_bit_cast(unsigned short, hh);
        l[r] = __builtin_bit_cast(unsigned short, ll);
    }
    const int blk = k4 >> 5, kk = k4 & 31;                                  // kk in {0, 4, .., 28}
    const int g = (kk & 15) >> 2, pos = g * 8 + (kk >= 16 ? 4 : 0);         // chunk g, first / second half of the chunk
    unsigned short *o = out + n * ldo + blk * 64 + pos;
    *reinterpret_cast<uint2 *>(o) = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
    *reinterpret_cast<uint2 *>(o + 32) = make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
}
}  // namespace

extern "C" int vs_split_pack_weight(const float *w, int64_t ldw, void *out, int64_t ldo, int32_t N, int32_t K, int32_t scale_exp,
                                    vs_stream_t stream_) {
    VS_CHECK(w && out, "vs_split_pack_weight: null pointer");
    VS_CHECK(N > 0 && K > 0 && K % 32 == 0 && ldw % 4 == 0 && ldo % 4 == 0, "vs_split_pack_weight: K must be a multiple of 32, row strides of 4 (N=%d K=%d)", N, K);
    VS_CHECK((((uintptr_t)w | (uintptr_t)out) & 15) == 0, "vs_split_pack_weight: 16-byte alignment required");
    VS_CHECK(scale_exp >= -30 && scale_exp <= 30, "vs_split_pack_weight: scale_exp out of range");
    const long long items = (long long)N * (K / 4);
    hipLaunchKernelGGL(split_pack_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, w, (long long)ldw,
                       (unsigned short *)out, 2LL * ldo, N, K, ldexpf(1.0f, scale_exp));
    VS_HIP(hipGetLastError());
    return 0;
}

// out = epilogue(acc_scale * (A Wp^T) + bias): A [M, K] f32 activations, Wp = vs_split_pack_weight image of the f32 weight (acc_scale =
// 2^-scale_exp), every output f32.  Epilogues 0 / 3 store, 1 exact-erf GELU, 2 gated residual update (resid null: in place), 5 x GELU'(resid), 4 packed
// q|k|v with RoPE (pos / kind / C / bases as vs_gemm_qkv_rope).  Row maps, gate and strides (in floats) as vs_gemm_bias_act.
extern "C" int vs_gemm_split(const float *A, const void *Wp, float acc_scale, const float *bias, float *out, const float *gate, const float *resid,
                             int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldw, int32_t ldo, int32_t epilogue, int32_t grp_in,
                             int32_t grp_out, int32_t grp_off, int32_t gate_rows, int32_t gate_ld, int32_t a_grp_in, int32_t a_grp_out,
                             int32_t a_grp_off, const int32_t *pos, const uint8_t *kind, int32_t C, float base2d, float theta1d,
                             vs_stream_t stream_) {
    const int out_packed = (epilogue & 16) ? 1 : 0;      // + 16: packed (hi, lo) output (epilogues 0 / 1 / 3: the A operand of vs_gemm_split_packed; 4: q | k | v for vs_attention dtype 4 + 32)
    epilogue &= ~16;
    VS_CHECK(epilogue >= 0 && epilogue <= 5, "vs_gemm_split: unknown epilogue %d", epilogue);
    VS_CHECK(epilogue != 5 || (resid && !bias && !gate && !out_packed && grp_in == 0 && K % 64 == 0), "vs_gemm_split: epilogue 5 (x GELU'(z)) needs z in `resid` (the layout of out), no bias / gate / row map / packed output, K %% 64 == 0");
    VS_CHECK(!out_packed || ((epilogue == 0 || epilogue == 1 || epilogue == 3 || epilogue == 4) && N % 64 == 0 && ldo % 32 == 0 && ((uintptr_t)out & 127) == 0),
             "vs_gemm_split: a packed output needs epilogue 0 / 1 / 3 / 4, N %% 64 == 0, ldo %% 32 == 0 and a 128-byte aligned buffer");
    VS_CHECK(epilogue != 4 || (pos && C > 0 && C % 64 == 0 && N >= 2 * C && N % 64 == 0 && base2d > 0.f && theta1d > 0.f),
             "vs_gemm_split: the RoPE epilogue needs pos, C %% 64 == 0, N >= 2C, positive bases");
    VS_CHECK(acc_scale > 0.f, "vs_gemm_split: acc_scale must be positive");
    return gemm_entry("vs_gemm_split", A, Wp, bias, out, gate, resid, M, N, K, lda, ldw, ldo, epilogue, 4, grp_in, grp_out, grp_off, gate_rows,
                      gate_ld, a_grp_in, a_grp_out, a_grp_off, epilogue == 4 ? pos : nullptr, epilogue == 4 ? kind : nullptr, C, base2d, theta1d,
                      (hipStream_t)stream_, acc_scale, 0, out_packed);
}

// vs_gemm_split with the A operand ALREADY packed (Ap = the vs_split_pack_weight image, scale_exp 0, of the f32 activation matrix; lda in
// 4-byte units as for the f32 matrix): the kernels skip the in-loop conversion -- what a producer that writes the packed form buys.
extern "C" int vs_gemm_split_packed(const void *Ap, const void *Wp, float acc_scale, const float *bias, float *out, const float *gate, const float *resid,
                                    int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldw, int32_t ldo, int32_t epilogue, int32_t grp_in,
                                    int32_t grp_out, int32_t grp_off, int32_t gate_rows, int32_t gate_ld, int32_t a_grp_in, int32_t a_grp_out,
                                    int32_t a_grp_off, const int32_t *pos, const uint8_t *kind, int32_t C, float base2d, float theta1d,
                                    vs_stream_t stream_) {
    const int out_packed = (epilogue & 16) ? 1 : 0;      // + 16: packed (hi, lo) output (epilogues 0 / 1 / 3: the A operand of vs_gemm_split_packed; 4: q | k | v for vs_attention dtype 4 + 32)
    epilogue &= ~16;
    VS_CHECK(epilogue >= 0 && epilogue <= 4, "vs_gemm_split_packed: unknown epilogue %d", epilogue);
    VS_CHECK(!out_packed || ((epilogue == 0 || epilogue == 1 || epilogue == 3 || epilogue == 4) && N % 64 == 0 && ldo % 32 == 0 && ((uintptr_t)out & 127) == 0),
             "vs_gemm_split_packed: a packed output needs epilogue 0 / 1 / 3 / 4, N %% 64 == 0, ldo %% 32 == 0 and a 128-byte aligned buffer");
    VS_CHECK(epilogue != 4 || (pos && C > 0 && C % 64 == 0 && N >= 2 * C && N % 64 == 0 && base2d > 0.f && theta1d > 0.f),
             "vs_gemm_split_packed: the RoPE epilogue needs pos, C %% 64 == 0, N >= 2C, positive bases");
    VS_CHECK(acc_scale > 0.f, "vs_gemm_split_packed: acc_scale must be positive");
    return gemm_entry("vs_gemm_split_packed", Ap, Wp, bias, out, gate, resid, M, N, K, lda, ldw, ldo, epilogue, 4, grp_in, grp_out, grp_off, gate_rows,
                      gate_ld, a_grp_in, a_grp_out, a_grp_off, epilogue == 4 ? pos : nullptr, epilogue == 4 ? kind : nullptr, C, base2d, theta1d,
                      (hipStream_t)stream_, acc_scale, 1, out_packed);
}

// Weight-gradient GEMM: out32[t][M,N] += A[M,K] (W + shift[t])[N,K]^T for t < max(ntaps, 1), the K range cut into `ksplit` slices
// that run as separate workgroups: long thin reductions (M, N = channels, K = millions of pixels or tokens).  ntaps = 0: one
// plain GEMM (shifts ignored); 1..9: the taps of a 3x3 convolution -- A = dY^T [Cout, pixels], W = X^T [Cin, pixels]
// (zero-bordered), shift[t] = the tap's pixel offset, out + t * tap_out_stride = that tap's [M, N] gradient.  A and W may start
// at any 2-byte aligned address (LDS-DMA staging), which is what lets the taps be shifted views.  workspace: null (slices meet
// through f32 atomics) or >= slices * max(ntaps, 1) * M * N floats (slices store partial tiles, a second kernel sums them and,
// with accumulate == 0, overwrites out instead of adding to it).
// a_slice_stride / w_slice_stride: 0 = K slice s is columns [s K/ksplit, (s+1) K/ksplit) of A / W; > 0 = slice-blocked operands,
// slice s is columns [0, K/ksplit) of the matrix at A + s * a_slice_stride (elements), see vs_transpose16_ex.
extern "C" int vs_gemm_wgrad(const void *A, const void *W, float *out, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldw,
                             int32_t ldo, int64_t a_slice_stride, int64_t w_slice_stride, int64_t tap_out_stride, const int32_t *shifts,
                             int32_t ntaps, int32_t ksplit, int32_t dtype, void *workspace, int64_t workspace_bytes, int32_t accumulate,
                             vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(A && W && out, "vs_gemm_wgrad: null pointer");
    VS_CHECK(M > 0 && N > 0 && K > 0 && ksplit >= 1 && ksplit <= 65535, "vs_gemm_wgrad: bad sizes M=%d N=%d K=%d ksplit=%d", M, N, K, ksplit);
    VS_CHECK(ntaps >= 0 && ntaps <= 9 && (ntaps == 0 || shifts), "vs_gemm_wgrad: 0 <= ntaps <= 9, and shifts when ntaps > 0");
    VS_CHECK(K % (32 * ksplit) == 0, "vs_gemm_wgrad: K=%d must be a multiple of 32 * ksplit (%d)", K, 32 * ksplit);
    VS_CHECK(dtype == 1 || dtype == 2 || dtype == 4, "vs_gemm_wgrad: dtype must be 1 (f16), 2 (bf16) or 4 (split: A f32, W packed by vs_split_pack_weight / vs_transpose_pack_split)");
    if (dtype == 4) {   // split operands: f32 rows addressed in 2-byte units (gemm_common.h, kDtSplit); tap shifts (in floats) move the f32 A
                        // operand -- a packed W cannot be shifted: A = zero-bordered X^T with a readable halo of |shift| floats on both sides
        VS_CHECK(lda % 4 == 0 && ldw % 4 == 0 && (((uintptr_t)A | (uintptr_t)W) & 15) == 0, "vs_gemm_wgrad: split operands need 16-byte aligned rows");
        K *= 2; lda *= 2; ldw *= 2; a_slice_stride *= 2; w_slice_stride *= 2;
    }
    GemmArgs g;
    g.A = A; g.W = W; g.bias = nullptr; g.out = out; g.gate = nullptr; g.resid = nullptr;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldo = ldo;
    g.grp_in = M; g.grp_out = M; g.grp_off = 0; g.gate_rows = M; g.gate_ld = N;
    g.a_grp_in = M; g.a_grp_out = M; g.a_grp_off = 0; g.m_lo = 0;
    g.a_sup_in = 0x7fffffff; g.a_sup_extra = 0; g.a_kstride = 32;
    g.rope_pos = nullptr; g.rope_kind = nullptr; g.rope_C = 0; g.rope_l2base = 0.f; g.rope_l2theta = 0.f; g.stagger = 0; g.row_band = wgrad_xcd(); g.acc_scale = 1.f; g.a_packed = 0; g.out_packed = 0; g.tap_on_a = 0;
    g.ntaps = ntaps; g.tap_out_stride = ntaps > 0 ? tap_out_stride : 0;
    g.a_slice_stride = a_slice_stride; g.w_slice_stride = w_slice_stride; g.k_valid = 0; g.conv_H = 0; g.conv_W = 0;
    VS_CHECK(a_slice_stride >= 0 && w_slice_stride >= 0, "vs_gemm_wgrad: negative slice stride");
    for (int t = 0; t < 9; ++t) g.tap_shift[t] = t < ntaps ? shifts[t] * (dtype == 4 ? 2 : 1) : 0;
    g.tap_on_a = dtype == 4 ? 1 : 0;
    VS_CHECK(accumulate || workspace, "vs_gemm_wgrad: accumulate = 0 (overwrite out) needs a workspace; the atomics path can only add");
    const int rc = dtype == 4 ? launch_wgrad<kDtSplit>(g, ksplit, (float *)workspace, workspace_bytes, accumulate, stream)
                 : dtype == 2 ? launch_wgrad<1>(g, ksplit, (float *)workspace, workspace_bytes, accumulate, stream)
                              : launch_wgrad<0>(g, ksplit, (float *)workspace, workspace_bytes, accumulate, stream);
    if (rc) return rc;
    VS_HIP(hipGetLastError());
    return 0;
}

// Weight gradient from REDUCTION-MAJOR operands: out32[M, N] (+)= sum over k < Kred of A[k, m] W[k, n] with A [Kred, M] and
// W [Kred, N] row-major 16-bit -- dW = dY^T X straight from dY [tokens, out features] and X [tokens, in features], without the
// transposed copies vs_gemm_wgrad needs.  lda / ldw multiples of 8 (rows may be padded: lda >= M), A / W 16-byte aligned; M and N
// need not be tile multiples (256 x 256 tiles; columns beyond the row storage come from a zero page).  The reduction is cut into ksplit slices of an even number (>= 2) of 64-row K tiles
// (the last slice is zero-filled past Kred by the kernel); workspace / accumulate as in vs_gemm_wgrad.
extern "C" int vs_gemm_wgrad_tn(const void *A, const void *W, float *out, int32_t M, int32_t N, int32_t Kred, int32_t lda, int32_t ldw,
                                int32_t ldo, int32_t ksplit, int32_t dtype, void *workspace, int64_t workspace_bytes, int32_t accumulate,
                                vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(A && W && out, "vs_gemm_wgrad_tn: null pointer");
    VS_CHECK(M > 0 && N > 0 && Kred > 0 && ksplit >= 1 && ksplit <= 65535, "vs_gemm_wgrad_tn: bad sizes");
    VS_CHECK(lda % 8 == 0 && ldw % 8 == 0 && lda >= M && ldw >= N, "vs_gemm_wgrad_tn: lda / ldw must be multiples of 8 and cover the rows");
    VS_CHECK((((uintptr_t)A | (uintptr_t)W) & 15) == 0, "vs_gemm_wgrad_tn: A and W must be 16-byte aligned");
    VS_CHECK(dtype == 1 || dtype == 2, "vs_gemm_wgrad_tn: dtype must be 1 (f16) or 2 (bf16)");
    VS_CHECK(accumulate || workspace, "vs_gemm_wgrad_tn: accumulate = 0 (overwrite out) needs a workspace");
    const int unit = 128 * ksplit;
    const long long Kpad = ((long long)Kred + unit - 1) / unit * unit;
    VS_CHECK(Kpad < 2147483647LL, "vs_gemm_wgrad_tn: reduction too long");
    GemmArgs g;
    g.A = A; g.W = W; g.bias = nullptr; g.out = out; g.gate = nullptr; g.resid = nullptr;
    g.M = M; g.N = N; g.K = (int)Kpad; g.lda = lda; g.ldw = ldw; g.ldo = ldo;
    g.grp_in = M; g.grp_out = M; g.grp_off = 0; g.gate_rows = M; g.gate_ld = N;
    g.a_grp_in = M; g.a_grp_out = M; g.a_grp_off = 0; g.m_lo = 0;
    g.a_sup_in = 0x7fffffff; g.a_sup_extra = 0; g.a_kstride = 32;
    g.rope_pos = nullptr; g.rope_kind = nullptr; g.rope_C = 0; g.rope_l2base = 0.f; g.rope_l2theta = 0.f; g.stagger = 0; g.row_band = wgrad_xcd(); g.acc_scale = 1.f; g.a_packed = 0; g.out_packed = 0; g.tap_on_a = 0;
    g.ntaps = 0; g.tap_out_stride = 0; g.a_slice_stride = 0; g.w_slice_stride = 0;
    g.ksplit = ksplit; g.k_valid = Kred; g.partials = nullptr; g.conv_H = 0; g.conv_W = 0;
    const long long need = (long long)ksplit * M * N * (long long)sizeof(float);
    if (workspace) {
        VS_CHECK(workspace_bytes >= need, "vs_gemm_wgrad_tn: workspace of %lld bytes given, %lld needed", (long long)workspace_bytes, need);
        VS_CHECK(((uintptr_t)workspace & 15) == 0, "vs_gemm_wgrad_tn: workspace must be 16-byte aligned");
        g.partials = (float *)workspace;
    }
    const long long nwg = (long long)vs::cdiv(M, 256) * vs::cdiv(N, 256) * ksplit;
    VS_CHECK(nwg <= 0x7fffffffLL, "vs_gemm_wgrad_tn: grid too large");
    if (dtype == 2) hipLaunchKernelGGL((gemm256_tn_splitk_kernel<true>), dim3((unsigned)nwg), dim3(512), 0, stream, g);
    else hipLaunchKernelGGL((gemm256_tn_splitk_kernel<false>), dim3((unsigned)nwg), dim3(512), 0, stream, g);
    if (workspace) {
        const bool v4 = N % 4 == 0 && ldo % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
        const long long items = (long long)M * (v4 ? N / 4 : N);
        const dim3 grid((unsigned)((items + 255) / 256));
        if (v4) hipLaunchKernelGGL(splitk_reduce_kernel<4>, grid, dim3(256), 0, stream, (const float *)workspace, out, M, N, 1, ksplit, (long long)ldo, 0LL, accumulate);
        else hipLaunchKernelGGL(splitk_reduce_kernel<1>, grid, dim3(256), 0, stream, (const float *)workspace, out, M, N, 1, ksplit, (long long)ldo, 0LL, accumulate);
    }
    VS_HIP(hipGetLastError());
    return 0;
}

// Split-class weight gradient with dY read as it is in memory: out32[M, N] (+)= sum_{k < Kred} A[k, m] Wp[n, k]; A [Kred, M] f32 (row stride lda
// floats), Wp [N, Kpad] the packed (hi, lo) transposed X of vs_transpose_pack_split (row stride ldw 4-byte units, zero beyond Kred).  M, N multiples
// of 256, Kpad a multiple of 64 * ksplit (an even number of 32-token K tiles per slice); rows of A in [Kred, Kpad) are not read.
// transpose_out: out is [N, M] (row stride ldo >= M) = the transpose of the product, written by the second stage (workspace required).
extern "C" int vs_gemm_wgrad_split_atn(const float *A, const void *Wp, float *out, int32_t M, int32_t N, int32_t Kred, int32_t Kpad, int32_t lda,
                                       int32_t ldw, int32_t ldo, int32_t ksplit, int32_t transpose_out, void *workspace, int64_t workspace_bytes,
                                       int32_t accumulate, vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(A && Wp && out, "vs_gemm_wgrad_split_atn: null pointer");
    VS_CHECK(!transpose_out || workspace, "vs_gemm_wgrad_split_atn: transpose_out needs a workspace (the transposing second stage)");
    VS_CHECK(M > 0 && N > 0 && Kred > 0 && Kpad >= Kred && ksplit >= 1 && ksplit <= 65535, "vs_gemm_wgrad_split_atn: bad sizes");
    VS_CHECK(M % 256 == 0 && N % 256 == 0, "vs_gemm_wgrad_split_atn: M=%d and N=%d must be multiples of 256", M, N);
    VS_CHECK(Kpad % (64 * ksplit) == 0, "vs_gemm_wgrad_split_atn: Kpad=%d must be a multiple of 64 * ksplit (ksplit=%d)", Kpad, ksplit);
    VS_CHECK(lda % 4 == 0 && lda >= M && ldw % 4 == 0 && ldw >= Kpad && ldo >= (transpose_out ? M : N), "vs_gemm_wgrad_split_atn: lda / ldw must be multiples of 4 and cover the rows");
    VS_CHECK((((uintptr_t)A | (uintptr_t)Wp) & 15) == 0, "vs_gemm_wgrad_split_atn: A and Wp must be 16-byte aligned");
    VS_CHECK(accumulate || workspace, "vs_gemm_wgrad_split_atn: accumulate = 0 (overwrite out) needs a workspace");
    GemmArgs g;
    g.A = A; g.W = Wp; g.bias = nullptr; g.out = out; g.gate = nullptr; g.resid = nullptr;
    g.M = M; g.N = N; g.K = Kpad; g.lda = lda; g.ldw = 2 * ldw; g.ldo = ldo;
    g.grp_in = M; g.grp_out = M; g.grp_off = 0; g.gate_rows = M; g.gate_ld = N;
    g.a_grp_in = M; g.a_grp_out = M; g.a_grp_off = 0; g.m_lo = 0;
    g.a_sup_in = 0x7fffffff; g.a_sup_extra = 0; g.a_kstride = 32;
    g.rope_pos = nullptr; g.rope_kind = nullptr; g.rope_C = 0; g.rope_l2base = 0.f; g.rope_l2theta = 0.f; g.stagger = 0; g.row_band = wgrad_xcd(); g.acc_scale = 1.f; g.a_packed = 0; g.out_packed = 0; g.tap_on_a = 0;
    g.ntaps = 0; g.tap_out_stride = 0; g.a_slice_stride = 0; g.w_slice_stride = 0;
    g.ksplit = ksplit; g.k_valid = Kred; g.partials = nullptr; g.conv_H = 0; g.conv_W = 0;
    const long long need = (long long)ksplit * M * N * (long long)sizeof(float);
    if (workspace) {
        VS_CHECK(workspace_bytes >= need, "vs_gemm_wgrad_split_atn: workspace of %lld bytes given, %lld needed", (long long)workspace_bytes, need);
        VS_CHECK(((uintptr_t)workspace & 15) == 0, "vs_gemm_wgrad_split_atn: workspace must be 16-byte aligned");
        g.partials = (float *)workspace;
    }
    const long long nwg = (long long)(M / 256) * (N / 256) * ksplit;
    VS_CHECK(nwg <= 0x7fffffffLL, "vs_gemm_wgrad_split_atn: grid too large");
    hipLaunchKernelGGL(gemm256_split_atn_splitk_kernel, dim3((unsigned)nwg), dim3(512), 0, stream, g);
    if (transpose_out) {
        hipLaunchKernelGGL(splitk_reduce_t_kernel, dim3((unsigned)((M / 32) * (N / 32))), dim3(256), 0, stream, (const float *)workspace, out, M, N, ksplit,
                           (long long)ldo, accumulate);
    } else if (workspace) {
        const bool v4 = ldo % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
        const long long items = (long long)M * (v4 ? N / 4 : N);
        const dim3 grid((unsigned)((items + 255) / 256));
        if (v4) hipLaunchKernelGGL(splitk_reduce_kernel<4>, grid, dim3(256), 0, stream, (const float *)workspace, out, M, N, 1, ksplit, (long long)ldo, 0LL, accumulate);
        else hipLaunchKernelGGL(splitk_reduce_kernel<1>, grid, dim3(256), 0, stream, (const float *)workspace, out, M, N, 1, ksplit, (long long)ldo, 0LL, accumulate);
    }
    VS_HIP(hipGetLastError());
    return 0;
}

// Split-class weight gradient of nn.Conv2d(k=3, s=1, p=1) with X read as it is: out32[tap][ci][co] (+)= sum over pixels p of act(x)[p + tap offset][ci]
// dy[p][co] (zero outside the image); x [Nimg,H,W,Cin] f32 NHWC contiguous, dyTp = vs_transpose_pack_split(dy [pixels, Cout]) [Cout, Ppad] (row stride ldw
// 4-byte units, zero beyond the pixels), out [9, Cin, Cout].  Cin, Cout multiples of 256, Ppad a multiple of 64 * ksplit.
extern "C" int vs_conv3x3_wgrad_split_atn(const float *x, const void *dyTp, float *out, int32_t Nimg, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                                          int32_t Ppad, int32_t ldw, int32_t relu_in, int32_t ksplit, void *workspace, int64_t workspace_bytes,
                                          int32_t accumulate, vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(x && dyTp && out, "vs_conv3x3_wgrad_split_atn: null pointer");
    VS_CHECK(Nimg > 0 && H > 0 && W > 0 && ksplit >= 1 && ksplit <= 65535, "vs_conv3x3_wgrad_split_atn: bad sizes");
    VS_CHECK(Cin > 0 && Cout > 0 && Cin % 256 == 0 && Cout % 256 == 0, "vs_conv3x3_wgrad_split_atn: Cin=%d and Cout=%d must be multiples of 256", Cin, Cout);
    const long long P = (long long)Nimg * H * W;
    VS_CHECK(P < 2147483647LL && Ppad >= P && Ppad % (64 * ksplit) == 0 && ldw % 4 == 0 && ldw >= Ppad, "vs_conv3x3_wgrad_split_atn: Ppad=%d must cover the pixels and be a multiple of 64 * ksplit", Ppad);
    VS_CHECK((((uintptr_t)x | (uintptr_t)dyTp) & 15) == 0, "vs_conv3x3_wgrad_split_atn: x and dyTp must be 16-byte aligned");
    VS_CHECK(accumulate || workspace, "vs_conv3x3_wgrad_split_atn: accumulate = 0 (overwrite out) needs a workspace");
    GemmArgs g;
    g.A = x; g.W = dyTp; g.bias = nullptr; g.out = out; g.gate = nullptr; g.resid = nullptr;
    g.M = Cin; g.N = Cout; g.K = Ppad; g.lda = Cin; g.ldw = 2 * ldw; g.ldo = Cout;
    g.grp_in = Cin; g.grp_out = Cin; g.grp_off = 0; g.gate_rows = Cin; g.gate_ld = Cout;
    g.a_grp_in = Cin; g.a_grp_out = Cin; g.a_grp_off = 0; g.m_lo = 0;
    g.a_sup_in = 0x7fffffff; g.a_sup_extra = 0; g.a_kstride = 32;
    g.rope_pos = nullptr; g.rope_kind = nullptr; g.rope_C = 0; g.rope_l2base = 0.f; g.rope_l2theta = 0.f; g.stagger = 0; g.row_band = wgrad_xcd(); g.acc_scale = 1.f; g.a_packed = 0; g.out_packed = 0; g.tap_on_a = 1;
    g.ntaps = 9; g.tap_out_stride = (long long)Cin * Cout; g.a_slice_stride = 0; g.w_slice_stride = 0;
    g.ksplit = ksplit; g.k_valid = (int)P; g.partials = nullptr; g.conv_H = H; g.conv_W = W;
    const long long need = (long long)ksplit * 9 * Cin * Cout * (long long)sizeof(float);
    if (workspace) {
        VS_CHECK(workspace_bytes >= need, "vs_conv3x3_wgrad_split_atn: workspace of %lld bytes given, %lld needed", (long long)workspace_bytes, need);
        VS_CHECK(((uintptr_t)workspace & 15) == 0, "vs_conv3x3_wgrad_split_atn: workspace must be 16-byte aligned");
        g.partials = (float *)workspace;
    }
    const long long nwg = (long long)(Cin / 256) * (Cout / 256) * 9 * ksplit;
    VS_CHECK(nwg <= 0x7fffffffLL, "vs_conv3x3_wgrad_split_atn: grid too large");
    if (relu_in) hipLaunchKernelGGL(conv3x3_wgrad_split_atn_kernel<true>, dim3((unsigned)nwg), dim3(512), 0, stream, g);
    else hipLaunchKernelGGL(conv3x3_wgrad_split_atn_kernel<false>, dim3((unsigned)nwg), dim3(512), 0, stream, g);
    if (workspace) {
        const bool v4 = (reinterpret_cast<uintptr_t>(out) & 15) == 0;
        const long long items = 9LL * Cin * (v4 ? Cout / 4 : Cout);
        const dim3 grid((unsigned)((items + 255) / 256));
        if (v4) hipLaunchKernelGGL(splitk_reduce_kernel<4>, grid, dim3(256), 0, stream, (const float *)workspace, out, Cin, Cout, 9, ksplit, (long long)Cout, g.tap_out_stride, accumulate);
        else hipLaunchKernelGGL(splitk_reduce_kernel<1>, grid, dim3(256), 0, stream, (const float *)workspace, out, Cin, Cout, 9, ksplit, (long long)Cout, g.tap_out_stride, accumulate);
    }
    VS_HIP(hipGetLastError());
    return 0;
}

// Weight gradient of nn.Conv2d(k=3, s=1, p=1) from the NHWC tensors as they are: out32[tap][ci][co] (+)= sum over pixels of
// act(x)[pixel + tap offset][ci] * dy[pixel][co] (zero outside the image), x [Nimg,H,W,Cin], dy [Nimg,H,W,Cout] contiguous 16-bit,
// out [9, Cin, Cout] (tap = ky*3 + kx; transpose the last two axes for the module's [Cout, Cin, ky, kx]).  relu_in applies the
// ResidualConvUnit's activation-before-conv to x on the fly.  Cin, Cout multiples of 8; 256 x 256 tiles, so efficient for channel
// counts that are multiples of 256.  ksplit slices of the pixel range; workspace (>= ksplit * 9 * Cin * Cout floats) / accumulate
// as in vs_gemm_wgrad.
extern "C" int vs_conv3x3_wgrad_tn(const void *x, const void *dy, float *out, int32_t Nimg, int32_t H, int32_t W, int32_t Cin,
                                   int32_t Cout, int32_t relu_in, int32_t ksplit, int32_t dtype, void *workspace,
                                   int64_t workspace_bytes, int32_t accumulate, vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(x && dy && out, "vs_conv3x3_wgrad_tn: null pointer");
    VS_CHECK(Nimg > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && ksplit >= 1 && ksplit <= 65535, "vs_conv3x3_wgrad_tn: bad sizes");
    VS_CHECK(Cin % 8 == 0 && Cout % 8 == 0, "vs_conv3x3_wgrad_tn: Cin=%d and Cout=%d must be multiples of 8", Cin, Cout);
    VS_CHECK((((uintptr_t)x | (uintptr_t)dy) & 15) == 0, "vs_conv3x3_wgrad_tn: x and dy must be 16-byte aligned");
    VS_CHECK(dtype == 1 || dtype == 2, "vs_conv3x3_wgrad_tn: dtype must be 1 (f16) or 2 (bf16)");
    VS_CHECK(accumulate || workspace, "vs_conv3x3_wgrad_tn: accumulate = 0 (overwrite out) needs a workspace");
    const long long P = (long long)Nimg * H * W;
    const int unit = 128 * ksplit;
    const long long Kpad = (P + unit - 1) / unit * unit;
    VS_CHECK(Kpad < 2147483647LL, "vs_conv3x3_wgrad_tn: too many pixels");
    GemmArgs g;
    g.A = x; g.W = dy; g.bias = nullptr; g.out = out; g.gate = nullptr; g.resid = nullptr;
    g.M = Cin; g.N = Cout; g.K = (int)Kpad; g.lda = Cin; g.ldw = Cout; g.ldo = Cout;
    g.grp_in = Cin; g.grp_out = Cin; g.grp_off = 0; g.gate_rows = Cin; g.gate_ld = Cout;
    g.a_grp_in = Cin; g.a_grp_out = Cin; g.a_grp_off = 0; g.m_lo = 0;
    g.a_sup_in = 0x7fffffff; g.a_sup_extra = 0; g.a_kstride = 32;
    g.rope_pos = nullptr; g.rope_kind = nullptr; g.rope_C = 0; g.rope_l2base = 0.f; g.rope_l2theta = 0.f; g.stagger = 0; g.row_band = wgrad_xcd(); g.acc_scale = 1.f; g.a_packed = 0; g.out_packed = 0; g.tap_on_a = 0;
    g.ntaps = 9; g.tap_out_stride = (long long)Cin * Cout; g.a_slice_stride = 0; g.w_slice_stride = 0;
    g.ksplit = ksplit; g.k_valid = (int)P; g.partials = nullptr; g.conv_H = H; g.conv_W = W;
    const long long need = (long long)ksplit * 9 * Cin * Cout * (long long)sizeof(float);
    if (workspace) {
        VS_CHECK(workspace_bytes >= need, "vs_conv3x3_wgrad_tn: workspace of %lld bytes given, %lld needed", (long long)workspace_bytes, need);
        VS_CHECK(((uintptr_t)workspace & 15) == 0, "vs_conv3x3_wgrad_tn: workspace must be 16-byte aligned");
        g.partials = (float *)workspace;
    }
    const int G = (Cin <= 128 && 256 % Cin == 0) ? 256 / Cin : 1;   // taps that share one 256-row tile (conv3x3_wgrad_tn_kernel)
    const long long nwg = (long long)(G > 1 ? 1 : vs::cdiv(Cin, 256)) * vs::cdiv(Cout, 256) * vs::cdiv(9, G) * ksplit;
    VS_CHECK(nwg <= 0x7fffffffLL, "vs_conv3x3_wgrad_tn: grid too large");
    const dim3 grid((unsigned)nwg), block(512);
    if (dtype == 2) {
        if (relu_in) hipLaunchKernelGGL((conv3x3_wgrad_tn_kernel<true, true>), grid, block, 0, stream, g);
        else hipLaunchKernelGGL((conv3x3_wgrad_tn_kernel<true, false>), grid, block, 0, stream, g);
    } else {
        if (relu_in) hipLaunchKernelGGL((conv3x3_wgrad_tn_kernel<false, true>), grid, block, 0, stream, g);
        else hipLaunchKernelGGL((conv3x3_wgrad_tn_kernel<false, false>), grid, block, 0, stream, g);
    }
    if (workspace) {
        const bool v4 = Cout % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
        const long long items = 9LL * Cin * (v4 ? Cout / 4 : Cout);
        const dim3 rg((unsigned)((items + 255) / 256));
        if (v4) hipLaunchKernelGGL(splitk_reduce_kernel<4>, rg, dim3(256), 0, stream, (const float *)workspace, out, Cin, Cout, 9, ksplit, (long long)Cout, (long long)Cin * Cout, accumulate);
        else hipLaunchKernelGGL(splitk_reduce_kernel<1>, rg, dim3(256), 0, stream, (const float *)workspace, out, Cin, Cout, 9, ksplit, (long long)Cout, (long long)Cin * Cout, accumulate);
    }
    VS_HIP(hipGetLastError());
    return 0;
}

// The two atomics-only forms of vs_gemm_wgrad (no workspace).
extern "C" int vs_gemm_splitk_accumulate(const void *A, const void *W, float *out, int32_t M, int32_t N, int32_t K, int32_t lda,
                                         int32_t ldw, int32_t ldo, int32_t ksplit, int32_t dtype, vs_stream_t stream_) {
    return vs_gemm_wgrad(A, W, out, M, N, K, lda, ldw, ldo, 0, 0, 0, nullptr, 0, ksplit, dtype, nullptr, 0, 1, stream_);
}

extern "C" int vs_gemm_taps_accumulate(const void *A, const void *W, float *out, int32_t M, int32_t N, int32_t K, int32_t lda,
                                       int32_t ldw, int32_t ldo, int64_t tap_out_stride, const int32_t *shifts, int32_t ntaps,
                                       int32_t ksplit, int32_t dtype, vs_stream_t stream_) {
    VS_CHECK(ntaps >= 1, "vs_gemm_taps_accumulate: ntaps must be >= 1");
    return vs_gemm_wgrad(A, W, out, M, N, K, lda, ldw, ldo, 0, 0, tap_out_stride, shifts, ntaps, ksplit, dtype, nullptr, 0, 1, stream_);
}

// 7x7 stride-1 pad-3 convolution of an RGB image (the gs head's input_merger, heads/dpt_gs_head.py:112-118) as a window
// GEMM on gemm_kernel: with a zero-padded NHWC image [Nimg, Hp, Wp, 3] the 7 horizontal taps x 3 channels of one kernel
// row are 21 CONTIGUOUS halfs starting at padded pixel (y+dy, x), so k-step dy of output pixel (y, x) is the 32-half slice
// at that address (LDS-DMA takes any 2-byte-aligned source; the 11 trailing halfs are finite image data multiplied by zero
// weights).  M = Nimg*H*W pixels, N = Cout, K = 7 x 32 (the packed weights carry an eighth, all-zero kernel row: Cout >= 256 runs on
// the 256x256 main loop with K = 8 x 32, whose last row reads padded image row y + 7: the image buffer must extend one padded row
// + 64 halfs behind its last pixel); no im2col buffer, bias fused.
extern "C" int vs_conv7x7_rgb_nhwc(const void *in_padded, const void *w, const float *bias, void *out, int32_t Nimg, int32_t H,
                                   int32_t W, int32_t Hp, int32_t Wp, int32_t Cout, int32_t dtype, vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(in_padded && w && out, "vs_conv7x7_rgb_nhwc: null pointer");
    VS_CHECK(Nimg >= 0 && H > 0 && W > 0 && Cout > 0, "vs_conv7x7_rgb_nhwc: bad sizes");
    VS_CHECK(Hp >= H + 6 && Wp >= W + 6, "vs_conv7x7_rgb_nhwc: padded image must be at least (H+6) x (W+6), got %d x %d", Hp, Wp);
    VS_CHECK(dtype == 1 || dtype == 2, "vs_conv7x7_rgb_nhwc: dtype must be 1 (f16) or 2 (bf16)");
    VS_CHECK((long long)Nimg * H * W < 2147483647LL && (long long)Nimg * Hp * Wp * 3 < 2147483647LL, "vs_conv7x7_rgb_nhwc: too large");
    VS_CHECK((reinterpret_cast<uintptr_t>(w) & 15) == 0, "vs_conv7x7_rgb_nhwc: w must be 16-byte aligned");
    if (Nimg == 0) return 0;
    GemmArgs g;
    g.A = in_padded; g.W = w; g.bias = bias; g.out = out; g.gate = nullptr; g.resid = nullptr;
    g.M = Nimg * H * W; g.N = Cout; g.K = 7 * 32;
    g.lda = 3; g.ldw = 8 * 32; g.ldo = Cout;
    g.grp_in = g.M; g.grp_out = g.M; g.grp_off = 0;
    g.gate_rows = g.M; g.gate_ld = Cout;
    g.m_lo = 0;
    g.a_grp_in = W; g.a_grp_out = Wp; g.a_grp_off = 0;  // pixel (row r, x) -> padded pixel r * Wp + x ...
    g.a_sup_in = H; g.a_sup_extra = (Hp - H) * Wp;      // ... plus the padding rows of the images before it
    g.a_kstride = Wp * 3;                               // next kernel row = next padded image row
    g.ksplit = 1; g.ntaps = 0; g.tap_out_stride = 0; g.partials = nullptr; g.a_slice_stride = 0; g.w_slice_stride = 0; g.k_valid = 0; g.conv_H = 0; g.conv_W = 0;
    g.rope_pos = nullptr; g.rope_kind = nullptr; g.rope_C = 0; g.rope_l2base = 0.f; g.rope_l2theta = 0.f; g.stagger = 0; g.row_band = wgrad_xcd(); g.acc_scale = 1.f; g.a_packed = 0; g.out_packed = 0; g.tap_on_a = 0;
    static const int no256 = [] { const char *e = getenv("VS_STEM_NO256"); return e ? atoi(e) : 0; }();
    if (Cout % 256 == 0 && g.M >= 256 && !no256) {
        const int nwg = vs::cdiv(g.M, 256) * (Cout / 256);
        if (dtype == 2) hipLaunchKernelGGL(conv7x7_256_kernel<1>, dim3(nwg), dim3(512), 0, stream, g);
        else hipLaunchKernelGGL(conv7x7_256_kernel<0>, dim3(nwg), dim3(512), 0, stream, g);
        VS_HIP(hipGetLastError());
        return 0;
    }
    const int rc = dtype == 2 ? launch_mi<true, 8>(g, 0, stream) : launch_mi<false, 8>(g, 0, stream);
    if (rc) return rc;
    VS_HIP(hipGetLastError());
    return 0;
}

// The 7x7 RGB stem on split operands (gemm_common.h, kDtSplit): in_padded is the f32 zero-bordered NHWC image [Nimg, Hp, Wp, 3] (plus one
// spare padded row and 64 spare floats behind it), wp the vs_split_pack_weight image of the [Cout, 8 * 32] f32 weight (kernel row dy at
// columns dy * 32 + dx * 3 + c, the rest zero), out f32 [Nimg, H, W, Cout]; Cout a multiple of 256.
extern "C" int vs_conv7x7_rgb_split_nhwc(const float *in_padded, const void *wp, float acc_scale, const float *bias, float *out, int32_t Nimg,
                                         int32_t H, int32_t W, int32_t Hp, int32_t Wp, int32_t Cout, vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(in_padded && wp && out && acc_scale > 0.f, "vs_conv7x7_rgb_split_nhwc: null pointer / bad scale");
    VS_CHECK(Nimg >= 0 && H > 0 && W > 0 && Cout > 0 && Cout % 256 == 0, "vs_conv7x7_rgb_split_nhwc: bad sizes (Cout must be a multiple of 256)");
    VS_CHECK(Hp >= H + 6 && Wp >= W + 6, "vs_conv7x7_rgb_split_nhwc: padded image must be at least (H+6) x (W+6), got %d x %d", Hp, Wp);
    VS_CHECK((long long)Nimg * H * W < 2147483647LL && (long long)Nimg * Hp * Wp * 6 < 2147483647LL, "vs_conv7x7_rgb_split_nhwc: too large");
    VS_CHECK((reinterpret_cast<uintptr_t>(wp) & 15) == 0 && (reinterpret_cast<uintptr_t>(in_padded) & 3) == 0, "vs_conv7x7_rgb_split_nhwc: alignment");
    if (Nimg == 0) return 0;
    GemmArgs g;
    g.A = in_padded; g.W = wp; g.bias = bias; g.out = out; g.gate = nullptr; g.resid = nullptr;
    g.M = Nimg * H * W; g.N = Cout; g.K = 8 * 64;         // (2-byte units of the f32 rows)
    g.lda = 6; g.ldw = 8 * 64; g.ldo = Cout;
    g.grp_in = g.M; g.grp_out = g.M; g.grp_off = 0;
    g.gate_rows = g.M; g.gate_ld = Cout;
    g.m_lo = 0;
    g.a_grp_in = W; g.a_grp_out = Wp; g.a_grp_off = 0;
    g.a_sup_in = H; g.a_sup_extra = (Hp - H) * Wp;
    g.a_kstride = Wp * 6;
    g.ksplit = 1; g.ntaps = 0; g.tap_out_stride = 0; g.partials = nullptr; g.a_slice_stride = 0; g.w_slice_stride = 0; g.k_valid = 0; g.conv_H = 0; g.conv_W = 0;
    g.rope_pos = nullptr; g.rope_kind = nullptr; g.rope_C = 0; g.rope_l2base = 0.f; g.rope_l2theta = 0.f; g.stagger = 0; g.row_band = wgrad_xcd(); g.acc_scale = acc_scale; g.a_packed = 0; g.out_packed = 0; g.tap_on_a = 0;
    hipLaunchKernelGGL(conv7x7_256_kernel<kDtSplit>, dim3(vs::cdiv(g.M, 256) * (Cout / 256)), dim3(512), 0, stream, g);
    VS_HIP(hipGetLastError());
    return 0;
}

// The same stem with the Gaussian-parameter head's upsample-add fused into its epilogue (stem_upadd_epilogue): out_packed [Nimg, H, W, Cout] in the
// packed (hi, lo) form = bilinear_x2(up_src [Nimg, H/2, W/2, Cout] f32) + relu(conv7x7 + bias).  H, W even; bias required.
extern "C" int vs_conv7x7_rgb_split_up_nhwc(const float *in_padded, const void *wp, float acc_scale, const float *bias, const float *up_src, void *out_,
                                            int32_t Nimg, int32_t H, int32_t W, int32_t Hp, int32_t Wp, int32_t Cout, vs_stream_t stream_) {
    float *out = (float *)out_;
    VS_CHECK(up_src && bias && H % 2 == 0 && W % 2 == 0 && H >= 4 && W >= 4, "vs_conv7x7_rgb_split_up_nhwc: up_src, bias, even H, W >= 4 required");
    VS_CHECK(Cout == 256, "vs_conv7x7_rgb_split_up_nhwc: Cout must be 256 (one column tile: the epilogue owns whole pixel rows)");
    VS_CHECK(((uintptr_t)out & 127) == 0 && ((uintptr_t)up_src & 15) == 0 && ((uintptr_t)bias & 15) == 0, "vs_conv7x7_rgb_split_up_nhwc: alignment (out 128 B, up_src / bias 16 B)");
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(in_padded && wp && out && acc_scale > 0.f, "vs_conv7x7_rgb_split_up_nhwc: null pointer / bad scale");
    VS_CHECK(Nimg >= 0 && H > 0 && W > 0 && Cout > 0 && Cout % 256 == 0, "vs_conv7x7_rgb_split_up_nhwc: bad sizes (Cout must be a multiple of 256)");
    VS_CHECK(Hp >= H + 6 && Wp >= W + 6, "vs_conv7x7_rgb_split_up_nhwc: padded image must be at least (H+6) x (W+6), got %d x %d", Hp, Wp);
    VS_CHECK((long long)Nimg * H * W < 2147483647LL && (long long)Nimg * Hp * Wp * 6 < 2147483647LL, "vs_conv7x7_rgb_split_up_nhwc: too large");
    VS_CHECK((reinterpret_cast<uintptr_t>(wp) & 15) == 0 && (reinterpret_cast<uintptr_t>(in_padded) & 3) == 0, "vs_conv7x7_rgb_split_up_nhwc: alignment");
    if (Nimg == 0) return 0;
    GemmArgs g;
    g.A = in_padded; g.W = wp; g.bias = bias; g.out = out; g.gate = up_src; g.resid = nullptr;
    g.M = Nimg * H * W; g.N = Cout; g.K = 8 * 64;         // (2-byte units of the f32 rows)
    g.lda = 6; g.ldw = 8 * 64; g.ldo = Cout;
    g.grp_in = g.M; g.grp_out = g.M; g.grp_off = 0;
    g.gate_rows = g.M; g.gate_ld = Cout;
    g.m_lo = 0;
    g.a_grp_in = W; g.a_grp_out = Wp; g.a_grp_off = 0;
    g.a_sup_in = H; g.a_sup_extra = (Hp - H) * Wp;
    g.a_kstride = Wp * 6;
    g.ksplit = 1; g.ntaps = 0; g.tap_out_stride = 0; g.partials = nullptr; g.a_slice_stride = 0; g.w_slice_stride = 0; g.k_valid = 0; g.conv_H = H / 2; g.conv_W = W / 2;
    g.rope_pos = nullptr; g.rope_kind = nullptr; g.rope_C = 0; g.rope_l2base = 0.f; g.rope_l2theta = 0.f; g.stagger = 0; g.row_band = wgrad_xcd(); g.acc_scale = acc_scale; g.a_packed = 0; g.out_packed = 0; g.tap_on_a = 0;
    hipLaunchKernelGGL((conv7x7_256_kernel<kDtSplit, true>), dim3(vs::cdiv(g.M, 256) * (Cout / 256)), dim3(512), 0, stream, g);
    VS_HIP(hipGetLastError());
    return 0;
}
