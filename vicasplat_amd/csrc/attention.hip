// Fused multi-head attention forward (head_dim 64) for gfx950: softmax(Q K^T * scale) V without materialising the
// score matrix.  One kernel serves the three attention shapes of VicaSplat:
//   * frame encoder        croco/blocks.py:94-112        257 queries x 257 keys per (frame, head)
//   * video/camera         backbone_vica.py:76-126        T*258 queries x T*258 keys; camera-token queries see the
//                                                         key PREFIX of frames <= t (the blocked-causal mask of
//                                                         :585-593 is exactly a per-query key-prefix length)
//   * cross-neighbour      backbone_vica.py:152-191       257 queries x keys of frames t-1 and t+1: expressed as two
//                                                         key SEGMENTS per batch item, gathered by row index -- no
//                                                         roll / cat copies of K and V
// Q, K, V are read in place from the packed projection output ([row, 3*H*64] with q | k | v column blocks): no
// head transposes.  Output is token-major [row, H*64], directly the A operand of the projection GEMM.
//
// CDNA4 mapping: 256 threads = 4 waves, 16 query rows per wave (64 per workgroup), 64-key tiles staged in LDS
// (K row-major, V transposed so that MFMA B fragments are two ds_read_b64).  Scores are computed TRANSPOSED
// (S^T = K Q^T with v_mfma_f32_16x16x32) so that every lane owns ONE query column: the online-softmax row
// reductions are 15 in-register max/adds + 2 cross-lane steps, and the f32->16-bit P fragments are already in
// the A-operand layout of the P.V MFMA (keys permuted consistently on the V side).
#include "common.h"

#include <cstdlib>

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int HD = 64;       // head dim
constexpr int KB = 64;       // keys per tile
constexpr int KROW = HD + 8; // halfs, K tile row stride (144 B)

struct AttnArgs {
    const unsigned short *q, *k, *v;
    unsigned short *out;
    const int32_t *kv_seg;   // [nbatch,4] base0,len0,base1,len1 (rows) or null
    const int32_t *q_kvlen;  // [nbatch*Lq] or null
    int nbatch, H, Lq, Lk;
    long long q_batch_rows, k_batch_rows;
    int ldq, ldk, ldv, ldo;
    float scale_log2e;
    float *lse;  // optional [rows, H] f32: log2-domain logsumexp of the scaled scores (saved for the backward pass)
};

template <bool BF16>
__device__ __forceinline__ f4 mfma(const uint4 &a, const uint4 &b, f4 c) {
    if constexpr (BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf8 *>(&a), *reinterpret_cast<const bf8 *>(&b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const half8 *>(&a), *reinterpret_cast<const half8 *>(&b), c, 0, 0, 0);
}

template <bool BF16>
__device__ __forceinline__ unsigned pack2(float a, float b) {
    // one packed convert (round to nearest even) instead of two converts + an OR: the softmax / dS path is VALU-bound
    unsigned r;
    if constexpr (BF16) asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    else asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// The same convert for a value that goes STRAIGHT INTO AN MFMA: compiler-visible instructions, so that the hazard recogniser inserts the
// VALU-write -> MFMA-read wait states (it does not look inside inline asm).  Round 4: attention_sp_kernel's third query group came out
// 5e-5 off with the asm form whenever the scheduler placed a P V MFMA right behind the convert; the older kernels used the asm form for
// their P fragments too and were only protected by what happened to be scheduled in between.
template <bool BF16>
__device__ __forceinline__ unsigned pack2v(float a, float b) {
    typedef float f2p_ __attribute__((ext_vector_type(2)));
    if constexpr (BF16) {
        typedef __bf16 b2p_ __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(unsigned, __builtin_convertvector(f2p_{a, b}, b2p_));
    } else {
        typedef _Float16 h2p_ __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(unsigned, __builtin_convertvector(f2p_{a, b}, h2p_));
    }
}

template <bool BF16>
__device__ __forceinline__ unsigned short to16(float v) {
    return (unsigned short)(pack2<BF16>(v, 0.f) & 0xFFFFu);
}

// ---- online-softmax step of one 16-query group over one 64-key tile.  The softmax is the VALU bottleneck of both kernels (16 scores
// per lane and group against 16 MFMAs = 256 matrix cycles; v_exp_f32 is quarter rate, 16 of them are 256 issue cycles on their own), so
// everything around the exponentials is kept to the fewest issue slots: bare v_max3_f32, the two cross-row reductions through
// v_permlane16_swap / v_permlane32_swap (VALU, no ds_bpermute round trip through the LDS queue and its lgkmcnt(0) drain), scale-and-
// subtract and the row sum on the packed-f32 forms (v_pk_fma_f32 / v_pk_add_f32; measured equal to the scalar forms here).  Scores are
// finite or -inf, never NaN. ----
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float vmax2(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// the value of the lane 16 (32) positions away, i.e. in the neighbouring 16-lane row (32-lane half): a swap of two copies of x hands
// every lane its own value and its partner's, in either order (tools/probe/permlane_swap.hip).  Written as inline asm on two distinct
// registers: the builtin called with the same value twice compiles to code that reads ONE of the two results twice (ROCm 7.2), and the
// s_nop covers the VALU-write -> permlane-read hazard the compiler cannot see inside an asm block.
__device__ __forceinline__ void swap16(float &a, float &b) { asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap32(float &a, float &b) { asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float rows_max(float x) {
    float a = x, b = x;
    swap16(a, b);
    a = vmax2(a, b); b = a;
    swap32(a, b);
    return vmax2(a, b);
}
__device__ __forceinline__ float rows_sum(float x) {
    float a = x, b = x;
    swap16(a, b);
    a += b; b = a;
    swap32(a, b);
    return a + b;
}

// st: S^T fragments (keys nb*16 + g*4 + r of this lane's query), masked entries -inf.  Updates the running max / sum, rescales the
// transposed O accumulator when some query's maximum moved, and returns P in the A-operand layout of the P V MFMA.
template <bool BF16>
__device__ __forceinline__ void softmax_tile(f4 (&st)[4], float scale_log2e, float &m_run, float &l_run, f4 (&o)[4], uint4 (&pf)[2]) {
    // scores to the log2 domain first (packed multiplies).  It also makes the MFMA results' first reader an instruction the compiler's
    // hazard recogniser sees (an inline-asm v_max3 reading them directly gets no MFMA -> VALU wait states: NaNs from in-flight
    // registers), and lets fmaxf() compile to bare v_max3_f32: arithmetic results are known canonical, MFMA outputs are not.
    // the maximum is taken on the raw MFMA outputs and scaled once: this file is compiled with -fno-honor-nans (Makefile), without which
    // fmaxf() on values the compiler cannot prove canonical costs an extra v_max per operand under IEEE mode (16 per tile; measured
    // -4 % on the whole kernel family).  The MFMA results' first readers stay compiler-visible instructions: an inline-asm v_max3
    // reading them gets no MFMA -> VALU wait states from the hazard recogniser (tried: NaNs from in-flight registers).
    float mx = fmaxf(fmaxf(st[0][0], st[0][1]), st[0][2]);
    mx = fmaxf(fmaxf(mx, st[0][3]), st[1][0]);
    mx = fmaxf(fmaxf(mx, st[1][1]), st[1][2]);
    mx = fmaxf(fmaxf(mx, st[1][3]), st[2][0]);
    mx = fmaxf(fmaxf(mx, st[2][1]), st[2][2]);
    mx = fmaxf(fmaxf(mx, st[2][3]), st[3][0]);
    mx = fmaxf(fmaxf(mx, st[3][1]), st[3][2]);
    mx = fmaxf(mx, st[3][3]);
    mx = rows_max(mx) * scale_log2e;
    const float m_new = vmax2(m_run, mx);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;
    const f2v sc = f2v{scale_log2e, scale_log2e}, neg_m = f2v{-m_use, -m_use};
    f2v acc = f2v{0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f2v x = f2v{st[nb][2 * h], st[nb][2 * h + 1]} * sc + neg_m;
            f2v p;
            p.x = __builtin_amdgcn_exp2f(x.x);
            p.y = __builtin_amdgcn_exp2f(x.y);
            st[nb][2 * h] = p.x; st[nb][2 * h + 1] = p.y;
            acc += p;
        }
    const float rs = rows_sum(acc.x + acc.y);
    if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {  // wave-uniform: some query's running max moved
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        l_run *= alpha;
#pragma unroll
        for (int db = 0; db < 4; ++db) o[db] *= alpha;   // (O is held transposed: this lane's registers are all its own query's)
        m_run = m_new;
    }
    l_run += rs;
    // (pack2v: compiler-visible converts -- these registers are MFMA operands, see pack2v)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        pf[ks].x = pack2v<BF16>(st[2 * ks][0], st[2 * ks][1]);
        pf[ks].y = pack2v<BF16>(st[2 * ks][2], st[2 * ks][3]);
        pf[ks].z = pack2v<BF16>(st[2 * ks + 1][0], st[2 * ks + 1][1]);
        pf[ks].w = pack2v<BF16>(st[2 * ks + 1][2], st[2 * ks + 1][3]);
    }
}

// QG = 16-query MFMA groups per wave (1 -> 64 queries per workgroup, 2 -> 128): more MFMAs per staged K/V tile.
// 4 consecutive ROWS (keys) of one column (d) of a row-major [keys][KROW] LDS tile through the transpose read ds_read_b64_tr_b16
// (semantics probed in tools/probe/tr_read.hip, as in attention_bwd.hip): lane t of a 16-lane group supplies
// &tile[row0 + (t >> 2)][col0 + (t & 3) * 4] and receives tile[row0 .. row0 + 3][col0 + t].  With it the P V product reads V as it is
// staged (16-byte row copies): no transposed V image, whose staging cost eight 4-byte LDS stores and ~16 bit operations per thread and
// tile (cycle stamps: 24 % of a frame-encoder workgroup's time was K/V staging).
typedef short tr4v_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 v_rows4(const unsigned short *tile, int row0, int col0, int t) {
    typedef tr4v_t __attribute__((address_space(3))) *trp_t;
    const unsigned short *p = tile + (row0 + (t >> 2)) * KROW + col0 + (t & 3) * 4;
    const tr4v_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(const_cast<unsigned short *>(p)));
    return __builtin_bit_cast(uint2, v);
}

template <int PITCH>
__device__ __forceinline__ uint2 v_rows4p(const unsigned short *tile, int row0, int col0, int t) {   // v_rows4 on rows of PITCH halves
    typedef tr4v_t __attribute__((address_space(3))) *trp_t;
    const unsigned short *p = tile + (row0 + (t >> 2)) * PITCH + col0 + (t & 3) * 4;
    const tr4v_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(const_cast<unsigned short *>(p)));
    return __builtin_bit_cast(uint2, v);
}

// O is accumulated TRANSPOSED (O^T = V^T P^T: the MFMA's operands swapped), so a lane holds its own query's output: registers
// o[db][r] = O[query][db*16 + g*4 + r].  The running-max rescale and the final 1/l then need no cross-lane traffic, and the row leaves
// as 16-byte stores: v_permlane16_swap gives an even-g lane 8 consecutive d of block db and an odd-g lane 8 of block db + 1 (the
// first version stored 2 bytes per lane and instruction: 16 store instructions per 16 queries, issue-bound).
template <bool BF16>
__device__ __forceinline__ void store_o_rows(const f4 (&o)[4], float l, int q, int Lq, unsigned short *out, long long row0, int ldo, int col0, int g) {
    typedef unsigned u2v_ __attribute__((ext_vector_type(2)));
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    uint2 pk[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        pk[db].x = pack2<BF16>(o[db][0] * inv, o[db][1] * inv);
        pk[db].y = pack2<BF16>(o[db][2] * inv, o[db][3] * inv);
    }
    unsigned short *op = out + (row0 + min(q, Lq - 1)) * ldo + col0 + (g & ~1) * 4;
    const bool odd = g & 1;
#pragma unroll
    for (int d2 = 0; d2 < 4; d2 += 2) {
        const u2v_ sx = __builtin_amdgcn_permlane16_swap(pk[d2].x, pk[d2 + 1].x, false, false);
        const u2v_ sy = __builtin_amdgcn_permlane16_swap(pk[d2].y, pk[d2 + 1].y, false, false);
        if (q < Lq) *reinterpret_cast<uint4 *>(op + (d2 + (odd ? 1 : 0)) * 16) = make_uint4(sx.x, sy.x, sx.y, sy.y);
    }
}

template <bool BF16, int QG, int WPE>
__global__ void __launch_bounds__(256, WPE) attention_kernel(const AttnArgs a) {
    constexpr int QBLK = 64 * QG;
    __shared__ __attribute__((aligned(16))) unsigned short sK2[2][KB * KROW];   // two-tile ring: one barrier per tile
    __shared__ __attribute__((aligned(16))) unsigned short sV2[2][KB * KROW];    // row-major, read through ds_read_b64_tr_b16
    __shared__ int s_maxlen;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QBLK;

    int base0, len0, base1, len1;
    if (a.kv_seg) {
        base0 = a.kv_seg[4 * b + 0]; len0 = a.kv_seg[4 * b + 1]; base1 = a.kv_seg[4 * b + 2]; len1 = a.kv_seg[4 * b + 3];
    } else {
        base0 = (int)(b * a.k_batch_rows); len0 = a.Lk; base1 = 0; len1 = 0;
    }
    const int Lk = len0 + len1;

    // ---- this lane's queries (columns of S^T): one per 16-query group ----
    int my_len[QG];
    uint4 qf[QG][2];
    int wave_len = 0;
#pragma unroll
    for (int u = 0; u < QG; ++u) {
        const int qi = q0 + (wid * QG + u) * 16 + c16;
        const bool qvalid = qi < a.Lq;
        const long long qrow = b * a.q_batch_rows + (qvalid ? qi : a.Lq - 1);
        int ml = Lk;
        if (a.q_kvlen && qvalid) ml = min(Lk, a.q_kvlen[(long long)b * a.Lq + qi]);
        if (!qvalid) ml = 0;
        my_len[u] = ml;
        wave_len = max(wave_len, ml);
        const unsigned short *qp = a.q + qrow * a.ldq + h * HD + g * 8;
        qf[u][0] = *reinterpret_cast<const uint4 *>(qp);
        qf[u][1] = *reinterpret_cast<const uint4 *>(qp + 32);
    }
#pragma unroll
    for (int o_ = 32; o_ > 0; o_ >>= 1) wave_len = max(wave_len, __shfl_xor(wave_len, o_, 64));  // wave-uniform
    int wave_minlen[QG];  // smallest key limit among the group's queries: tiles entirely below it need no masking
#pragma unroll
    for (int u = 0; u < QG; ++u) {
        int mn = my_len[u];
#pragma unroll
        for (int o_ = 32; o_ > 0; o_ >>= 1) mn = min(mn, __shfl_xor(mn, o_, 64));
        wave_minlen[u] = mn;
    }
    if (tid == 0) s_maxlen = 0;
    __syncthreads();
    if (lane == 0) atomicMax(&s_maxlen, wave_len);
    __syncthreads();
    const int maxlen = s_maxlen;

    f4 o[QG][4];
    float m_run[QG], l_run[QG];
#pragma unroll
    for (int u = 0; u < QG; ++u) {
        m_run[u] = -INFINITY; l_run[u] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[u][i] = f4{0.f, 0.f, 0.f, 0.f};
    }

    // staging roles
    const int k_key = tid >> 2, k_chunk = (tid & 3) * 16;  // K and V: one key row, 2 x 16B of it

    auto key_row = [&](int j) -> long long {
        j = min(j, Lk - 1);
        return j < len0 ? (long long)base0 + j : (long long)base1 + (j - len0);
    };
    uint4 pk0, pk1, pva, pvb;  // register prefetch of the NEXT tile (hides the global latency behind the MFMAs)
    auto gload = [&](int kt) {
        const unsigned short *kp = a.k + key_row(kt + k_key) * a.ldk + h * HD + k_chunk;
        pk0 = *reinterpret_cast<const uint4 *>(kp);
        pk1 = *reinterpret_cast<const uint4 *>(kp + 8);
        const unsigned short *vp = a.v + key_row(kt + k_key) * a.ldv + h * HD + k_chunk;
        pva = *reinterpret_cast<const uint4 *>(vp);
        pvb = *reinterpret_cast<const uint4 *>(vp + 8);
    };
    auto lds_store = [&](int buf) {  // K and V tiles row-major, from the prefetched registers
        unsigned short *sK = sK2[buf], *sV = sV2[buf];
        *reinterpret_cast<uint4 *>(&sK[k_key * KROW + k_chunk]) = pk0;
        *reinterpret_cast<uint4 *>(&sK[k_key * KROW + k_chunk + 8]) = pk1;
        *reinterpret_cast<uint4 *>(&sV[k_key * KROW + k_chunk]) = pva;
        *reinterpret_cast<uint4 *>(&sV[k_key * KROW + k_chunk + 8]) = pvb;
    };
    if (maxlen > 0) {
        gload(0);
        lds_store(0);
        if (KB < maxlen) gload(KB);
    }
    __syncthreads();

    for (int kt = 0, it = 0; kt < maxlen; kt += KB, ++it) {
        const unsigned short *sK = sK2[it & 1], *sV = sV2[it & 1];
        // tile kt is in ring slot it&1 (ordered by the barrier that closed the previous iteration); the other slot was
        // last read during iteration it-1, so tile kt+KB can be written into it now and the loads for kt+2KB issued.
        if (kt + KB < maxlen) {
            lds_store((it + 1) & 1);
            if (kt + 2 * KB < maxlen) gload(kt + 2 * KB);
        }
        if (kt >= wave_len) { __syncthreads(); continue; }  // nothing visible to this wave's queries in this tile
        // K fragments are shared by the wave's query groups
        uint4 kf[4][2];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                kf[nb][ks] = *reinterpret_cast<const uint4 *>(&sK[(nb * 16 + c16) * KROW + ks * 32 + g * 8]);
        uint4 pf[QG][2];
#pragma unroll
        for (int u = 0; u < QG; ++u) {
            f4 st[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                st[nb] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) st[nb] = mfma<BF16>(kf[nb][ks], qf[u][ks], st[nb]);
            }
            // the prefix mask is applied only on the tile that straddles a query's key limit (one VGPR limit against constants)
            if (kt + KB > wave_minlen[u]) {
                const int lim = my_len[u] - kt - g * 4;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (nb * 16 + r >= lim) st[nb][r] = -INFINITY;
            }
            softmax_tile<BF16>(st, a.scale_log2e, m_run[u], l_run[u], o[u], pf[u]);
        }
        // O += P V ; V fragments shared by the query groups
#pragma unroll
        for (int db = 0; db < 4; ++db) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint2 lo = v_rows4(sV, (2 * ks) * 16 + g * 4, db * 16, c16);      // V[keys g*4..+3 of sub-block 2ks][d = db*16 + c16]
                const uint2 hi = v_rows4(sV, (2 * ks + 1) * 16 + g * 4, db * 16, c16);
                const uint4 vf = make_uint4(lo.x, lo.y, hi.x, hi.y);
#pragma unroll
                for (int u = 0; u < QG; ++u) o[u][db] = mfma<BF16>(vf, pf[u][ks], o[u][db]);   // O^T += V^T P^T: lane = query, registers = 4 consecutive d
            }
        }
        __syncthreads();
    }

    // ---- epilogue: O^T: query = this lane's c16, d = db*16 + g*4 + r ----
#pragma unroll
    for (int u = 0; u < QG; ++u) {
        store_o_rows<BF16>(o[u], l_run[u], q0 + (wid * QG + u) * 16 + c16, a.Lq, a.out, b * a.q_batch_rows, a.ldo, h * HD, g);
        if (a.lse && g == 0) {
            const int qo = q0 + (wid * QG + u) * 16 + c16;
            if (qo < a.Lq) a.lse[(b * a.q_batch_rows + qo) * a.H + h] = l_run[u] > 0.f ? m_run[u] + log2f(l_run[u]) : -INFINITY;
        }
    }
}

// ---- resident variant for short key sets (the frame encoder: 257 x 257 per (frame, head), no mask, no segments).
// The tiled kernel re-stages every K/V tile once per 64-query workgroup (5 x for 257 queries) with two barriers per
// tile, and pays full tiles for the 257th key and the 257th query.  Here ONE workgroup of 8 waves owns a (frame, head):
// all keys and values go to LDS once (both row-major, 76.5 KiB for 272 padded keys -> two workgroups per CU),
// one barrier, then every wave walks its 16-query groups over the key tiles with no further synchronisation; a partial
// last tile only runs the 16-key sub-blocks that hold keys. ----
constexpr int kResMaxKeys = 320;

template <bool BF16>
__global__ void __launch_bounds__(512, 2) attention_res_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_res[];
    const int Lk = a.Lk;
    const int Lkp = (Lk + 15) & ~15;      // keys padded to the 16-key MFMA sub-block
    unsigned short *sK = smem_res;                  // [Lkp][KROW]
    unsigned short *sV = smem_res + Lkp * KROW;     // [Lkp][KROW], row-major (rows >= Lk zero)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y;
    const long long kbase = b * a.k_batch_rows;

    // ---- stage all keys / values: thread -> (key, 16-half chunk) for K, (key pair, 8-dim chunk) for V.  Every global
    // load of the workgroup is issued before the first LDS store (up to 3 rounds x 4 x 16 B per thread in flight). ----
    constexpr int NR = (kResMaxKeys * 4 + 511) / 512;  // rounds of 512 threads over (key, 16-half chunk) items
    uint4 rk[NR][2], rv[NR][2];
    const int nK = Lkp * 4;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int idx = tid + r * 512;
        rk[r][0] = rk[r][1] = rv[r][0] = rv[r][1] = make_uint4(0, 0, 0, 0);
        if (idx < nK) {
            const int key = idx >> 2, ch = (idx & 3) * 16;
            if (key < Lk) {
                const unsigned short *kp = a.k + (kbase + key) * a.ldk + h * HD + ch;
                const unsigned short *vp = a.v + (kbase + key) * a.ldv + h * HD + ch;
                rk[r][0] = *reinterpret_cast<const uint4 *>(kp);
                rk[r][1] = *reinterpret_cast<const uint4 *>(kp + 8);
                rv[r][0] = *reinterpret_cast<const uint4 *>(vp);
                rv[r][1] = *reinterpret_cast<const uint4 *>(vp + 8);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int idx = tid + r * 512;
        if (idx < nK) {
            const int key = idx >> 2, ch = (idx & 3) * 16;
            *reinterpret_cast<uint4 *>(&sK[key * KROW + ch]) = rk[r][0];
            *reinterpret_cast<uint4 *>(&sK[key * KROW + ch + 8]) = rk[r][1];
            *reinterpret_cast<uint4 *>(&sV[key * KROW + ch]) = rv[r][0];
            *reinterpret_cast<uint4 *>(&sV[key * KROW + ch + 8]) = rv[r][1];
        }
    }
    __syncthreads();

    const int ngroups = (a.Lq + 15) >> 4;
    for (int grp = wid; grp < ngroups; grp += 8) {
        const int qi = grp * 16 + c16;
        const bool qvalid = qi < a.Lq;
        const long long qrow = b * a.q_batch_rows + (qvalid ? qi : a.Lq - 1);
        const unsigned short *qp = a.q + qrow * a.ldq + h * HD + g * 8;
        const uint4 qf0 = *reinterpret_cast<const uint4 *>(qp), qf1 = *reinterpret_cast<const uint4 *>(qp + 32);
        f4 o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = f4{0.f, 0.f, 0.f, 0.f};
        float m_run = -INFINITY, l_run = 0.f;
        int kt = 0;
        for (; kt + KB <= Lk; kt += KB) {   // full tiles: straight-line code, no sub-block conditions
            f4 st[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const unsigned short *kr = &sK[(kt + nb * 16 + c16) * KROW + g * 8];
                st[nb] = mfma<BF16>(*reinterpret_cast<const uint4 *>(kr), qf0, f4{0.f, 0.f, 0.f, 0.f});
                st[nb] = mfma<BF16>(*reinterpret_cast<const uint4 *>(kr + 32), qf1, st[nb]);
            }
            uint4 pf[2];
            softmax_tile<BF16>(st, a.scale_log2e, m_run, l_run, o, pf);
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const uint2 lo = v_rows4(sV, kt + (2 * ks) * 16 + g * 4, db * 16, c16);
                    const uint2 hi = v_rows4(sV, kt + (2 * ks + 1) * 16 + g * 4, db * 16, c16);
                    o[db] = mfma<BF16>(make_uint4(lo.x, lo.y, hi.x, hi.y), pf[ks], o[db]);   // O^T += V^T P^T
                }
        }
        if (kt < Lk) {   // the partial last tile: only the 16-key sub-blocks that hold keys (wave-uniform conditions)
            const int nbmax = (Lk - kt + 15) >> 4;
            f4 st[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                st[nb] = f4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                if (nb < nbmax) {
                    const unsigned short *kr = &sK[(kt + nb * 16 + c16) * KROW + g * 8];
                    f4 acc = mfma<BF16>(*reinterpret_cast<const uint4 *>(kr), qf0, f4{0.f, 0.f, 0.f, 0.f});
                    st[nb] = mfma<BF16>(*reinterpret_cast<const uint4 *>(kr + 32), qf1, acc);
                }
            }
            const int lim = Lk - kt - g * 4;   // keys past Lk are padding
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (nb * 16 + r >= lim) st[nb][r] = -INFINITY;
            uint4 pf[2];
            softmax_tile<BF16>(st, a.scale_log2e, m_run, l_run, o, pf);
            const int ksmax = (nbmax + 1) >> 1;  // 32-key MFMA steps that hold keys (V rows beyond Lk are zero)
#pragma unroll
            for (int db = 0; db < 4; ++db) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if (ks < ksmax) {
                        const uint2 lo = v_rows4(sV, kt + (2 * ks) * 16 + g * 4, db * 16, c16);
                        uint2 hi = make_uint2(0, 0);
                        if (2 * ks + 1 < nbmax) hi = v_rows4(sV, kt + (2 * ks + 1) * 16 + g * 4, db * 16, c16);
                        o[db] = mfma<BF16>(make_uint4(lo.x, lo.y, hi.x, hi.y), pf[ks], o[db]);
                    }
                }
            }
        }
        store_o_rows<BF16>(o, l_run, qi, a.Lq, a.out, b * a.q_batch_rows, a.ldo, h * HD, g);
        if (a.lse && g == 0 && qvalid) a.lse[(b * a.q_batch_rows + qi) * a.H + h] = l_run > 0.f ? m_run + log2f(l_run) : -INFINITY;
    }
}

// ---- split-operand attention (dtype code 4; gemm_common.h, kDtSplit): q | k | v and the output are f32 arrays, every matrix product
// is three f16 MFMAs on (hi, lo) pairs -- hi = rne16(x), lo = rne16(x - hi) -- with f32 accumulation: f32-class results at a third of
// the 16-bit matrix rate instead of the 1/16 of the exact-f32 MFMA.  Same structure, masks and segments as attention_kernel: the K / V
// tiles are converted once when they are staged (two LDS images each), Q when it is loaded, P in registers after the exponentials. ----
struct AttnArgsSplit {
    const float *q, *k, *v;
    float *out;
    const int32_t *kv_seg, *q_kvlen;
    int nbatch, H, Lq, Lk;
    long long q_batch_rows, k_batch_rows;
    int ldq, ldk, ldv, ldo;
    float scale_log2e;
    float *lse;
    int out_packed;   // write O in the packed (hi, lo) form of the split class (the A operand of the projection GEMM: vs_gemm_split_packed)
    int xcd;          // 1: XCD-aware (tile, head, batch) ids: the query tiles of one (batch, head) read its K / V behind ONE L2 (attention_bwd.hip block_bhx)
};

// four consecutive columns n .. n + 3 of an f32 row written in the packed (hi, lo) layout of vs_split_pack_weight (gemm_common.h, store_split4)
__device__ __forceinline__ void store_split4_attn(float *row, int n, float a, float b, float c, float d) {
    typedef _Float16 h2_ __attribute__((ext_vector_type(2)));
    typedef float f2_ __attribute__((ext_vector_type(2)));
    const h2_ h0 = __builtin_convertvector(f2_{a, b}, h2_), h1 = __builtin_convertvector(f2_{c, d}, h2_);
    const h2_ l0 = __builtin_convertvector(f2_{a - (float)h0.x, b - (float)h0.y}, h2_), l1 = __builtin_convertvector(f2_{c - (float)h1.x, d - (float)h1.y}, h2_);
    const int kk = n & 31;
    unsigned short *o = reinterpret_cast<unsigned short *>(row + (n & ~31)) + ((kk & 15) >> 2) * 8 + (kk >> 4) * 4;
    *reinterpret_cast<uint2 *>(o) = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
    *reinterpret_cast<uint2 *>(o + 32) = make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
}

__device__ __forceinline__ unsigned cvt_pk_f16s(float a, float b) { return pack2<false>(a, b); }
// 8 floats -> 8 hi halves + 8 lo halves (in order); x - float(hi) on v_fma_mix_f32 (f16 source operand, exact f32 result)
__device__ __forceinline__ void split8s(const float4 &a, const float4 &b, uint4 &hi, uint4 &lo) {
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unsigned h[4], l[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        h[p] = cvt_pk_f16s(x[2 * p], x[2 * p + 1]);
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h[p]), "v"(x[2 * p]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h[p]), "v"(x[2 * p + 1]));
        l[p] = cvt_pk_f16s(r0, r1);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}
// c += b a over (hi, lo) pairs: lo_b hi_a + hi_b lo_a + hi_b hi_a
__device__ __forceinline__ f4 mma3(const uint4 &bh, const uint4 &bl, const uint4 &ah, const uint4 &al, f4 c) {
    c = mfma<false>(bl, ah, c);
    c = mfma<false>(bh, al, c);
    return mfma<false>(bh, ah, c);
}

template <int QG>
__global__ void __launch_bounds__(256, 2) attention_split_kernel(const AttnArgsSplit a) {
    constexpr int QBLK = 64 * QG;
    __shared__ __attribute__((aligned(16))) unsigned short sKh2[2][KB * KROW], sKl2[2][KB * KROW];   // two-tile ring, hi / lo images
    // V tiles keep the K pitch (144 B).  A 160-byte pitch (the four rows of a transpose read in four disjoint 8-bank windows) was measured:
    // the bank-conflict counter drops 42 -> 38 %, the stores become 2-way conflicted, the kernel time is unchanged (+-2 %).
    constexpr int VROW = KROW;
    __shared__ __attribute__((aligned(16))) unsigned short sVh2[2][KB * VROW], sVl2[2][KB * VROW];
    __shared__ int s_maxlen;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    int b = blockIdx.z, h = blockIdx.y, bx_ = blockIdx.x;
    if (a.xcd) {      // consecutive workgroups are dealt to consecutive XCDs: make consecutive LOGICAL ids share one (gemm256_kernel's remap)
        const int gx = gridDim.x, gy = gridDim.y, n = gx * gy * gridDim.z;
        const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const int q_ = n >> 3, r_ = n & 7, xc = lin & 7, idx = lin >> 3;
        const int l = (xc < r_ ? xc * (q_ + 1) : r_ * (q_ + 1) + (xc - r_) * q_) + idx;
        bx_ = l % gx; const int t_ = l / gx; h = t_ % gy; b = t_ / gy;
    }
    const int q0 = bx_ * QBLK;

    int base0, len0, base1, len1;
    if (a.kv_seg) {
        base0 = a.kv_seg[4 * b + 0]; len0 = a.kv_seg[4 * b + 1]; base1 = a.kv_seg[4 * b + 2]; len1 = a.kv_seg[4 * b + 3];
    } else {
        base0 = (int)(b * a.k_batch_rows); len0 = a.Lk; base1 = 0; len1 = 0;
    }
    const int Lk = len0 + len1;

    int my_len[QG];
    uint4 qh[QG][2], ql[QG][2];
    int wave_len = 0;
#pragma unroll
    for (int u = 0; u < QG; ++u) {
        const int qi = q0 + (wid * QG + u) * 16 + c16;
        const bool qvalid = qi < a.Lq;
        const long long qrow = b * a.q_batch_rows + (qvalid ? qi : a.Lq - 1);
        int ml = Lk;
        if (a.q_kvlen && qvalid) ml = min(Lk, a.q_kvlen[(long long)b * a.Lq + qi]);
        if (!qvalid) ml = 0;
        my_len[u] = ml;
        wave_len = max(wave_len, ml);
        const float *qp = a.q + qrow * a.ldq + h * HD + g * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            split8s(*reinterpret_cast<const float4 *>(qp + ks * 32), *reinterpret_cast<const float4 *>(qp + ks * 32 + 4), qh[u][ks], ql[u][ks]);
    }
#pragma unroll
    for (int o_ = 32; o_ > 0; o_ >>= 1) wave_len = max(wave_len, __shfl_xor(wave_len, o_, 64));
    int wave_minlen[QG];
#pragma unroll
    for (int u = 0; u < QG; ++u) {
        int mn = my_len[u];
#pragma unroll
        for (int o_ = 32; o_ > 0; o_ >>= 1) mn = min(mn, __shfl_xor(mn, o_, 64));
        wave_minlen[u] = mn;
    }
    if (tid == 0) s_maxlen = 0;
    __syncthreads();
    if (lane == 0) atomicMax(&s_maxlen, wave_len);
    __syncthreads();
    const int maxlen = s_maxlen;

    f4 o[QG][4];
    float m_run[QG], l_run[QG];
#pragma unroll
    for (int u = 0; u < QG; ++u) {
        m_run[u] = -INFINITY; l_run[u] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[u][i] = f4{0.f, 0.f, 0.f, 0.f};
    }

    const int k_key = tid >> 2, k_chunk = (tid & 3) * 16;  // staging: one key row, 16 floats of K and of V per thread
    auto key_row = [&](int j) -> long long {
        j = min(j, Lk - 1);
        return j < len0 ? (long long)base0 + j : (long long)base1 + (j - len0);
    };
    float4 pk[4], pv[4];  // register prefetch of the NEXT tile
    auto gload = [&](int kt) {
        const long long r = key_row(kt + k_key);
        const float *kp = a.k + r * a.ldk + h * HD + k_chunk, *vp = a.v + r * a.ldv + h * HD + k_chunk;
#pragma unroll
        for (int i = 0; i < 4; ++i) { pk[i] = *reinterpret_cast<const float4 *>(kp + 4 * i); pv[i] = *reinterpret_cast<const float4 *>(vp + 4 * i); }
    };
    auto lds_store = [&](int buf) {
        uint4 h0, l0, h1, l1;
        split8s(pk[0], pk[1], h0, l0); split8s(pk[2], pk[3], h1, l1);
        *reinterpret_cast<uint4 *>(&sKh2[buf][k_key * KROW + k_chunk]) = h0; *reinterpret_cast<uint4 *>(&sKh2[buf][k_key * KROW + k_chunk + 8]) = h1;
        *reinterpret_cast<uint4 *>(&sKl2[buf][k_key * KROW + k_chunk]) = l0; *reinterpret_cast<uint4 *>(&sKl2[buf][k_key * KROW + k_chunk + 8]) = l1;
        split8s(pv[0], pv[1], h0, l0); split8s(pv[2], pv[3], h1, l1);
        *reinterpret_cast<uint4 *>(&sVh2[buf][k_key * VROW + k_chunk]) = h0; *reinterpret_cast<uint4 *>(&sVh2[buf][k_key * VROW + k_chunk + 8]) = h1;
        *reinterpret_cast<uint4 *>(&sVl2[buf][k_key * VROW + k_chunk]) = l0; *reinterpret_cast<uint4 *>(&sVl2[buf][k_key * VROW + k_chunk + 8]) = l1;
    };
    if (maxlen > 0) {
        gload(0);
        lds_store(0);
        if (KB < maxlen) gload(KB);
    }
    __syncthreads();

    for (int kt = 0, it = 0; kt < maxlen; kt += KB, ++it) {
        const unsigned short *sKh = sKh2[it & 1], *sKl = sKl2[it & 1], *sVh = sVh2[it & 1], *sVl = sVl2[it & 1];
        if (kt + KB < maxlen) {
            lds_store((it + 1) & 1);
            if (kt + 2 * KB < maxlen) gload(kt + 2 * KB);
        }
        if (kt >= wave_len) { __syncthreads(); continue; }
        uint4 pfh[QG][2], pfl[QG][2];
        // S^T for all QG query groups of the wave from ONE read of each K fragment (the LDS pipe is this kernel's busiest unit: PMC, 47 % of
        // the cycles + 20 % bank conflicts on the 2064-key video attention; re-reading the fragments per group was a third of its reads)
        f4 stq[QG][4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
            for (int u = 0; u < QG; ++u) stq[u][nb] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int off = (nb * 16 + c16) * KROW + ks * 32 + g * 8;
                const uint4 kh_ = *reinterpret_cast<const uint4 *>(&sKh[off]), kl_ = *reinterpret_cast<const uint4 *>(&sKl[off]);
#pragma unroll
                for (int u = 0; u < QG; ++u) stq[u][nb] = mma3(kh_, kl_, qh[u][ks], ql[u][ks], stq[u][nb]);
            }
        }
#pragma unroll
        for (int u = 0; u < QG; ++u) {
            f4 (&st)[4] = stq[u];
            if (kt + KB > wave_minlen[u]) {
                const int lim = my_len[u] - kt - g * 4;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (nb * 16 + r >= lim) st[nb][r] = -INFINITY;
            }
            softmax_tile<false>(st, a.scale_log2e, m_run[u], l_run[u], o[u], pfh[u]);   // st now holds P (f32), pfh its rne16
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const unsigned hh[4] = {pfh[u][ks].x, pfh[u][ks].y, pfh[u][ks].z, pfh[u][ks].w};
                unsigned ll[4];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float p0 = st[2 * ks + (w >> 1)][(w & 1) * 2], p1 = st[2 * ks + (w >> 1)][(w & 1) * 2 + 1];
                    // compiler-visible instructions only: the result feeds the P V MFMAs, and the VALU-write -> MFMA-read wait state
                    // is inserted by the compiler's hazard recogniser, which does not look inside inline asm
                    typedef _Float16 h2_ __attribute__((ext_vector_type(2)));
                    typedef float f2_ __attribute__((ext_vector_type(2)));
                    const h2_ hv = __builtin_bit_cast(h2_, hh[w]);
                    ll[w] = __builtin_bit_cast(unsigned, __builtin_convertvector(f2_{p0 - (float)hv.x, p1 - (float)hv.y}, h2_));
                }
                pfl[u][ks] = make_uint4(ll[0], ll[1], ll[2], ll[3]);
            }
        }
#pragma unroll
        for (int db = 0; db < 4; ++db) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint2 h0 = v_rows4p<VROW>(sVh, (2 * ks) * 16 + g * 4, db * 16, c16), h1 = v_rows4p<VROW>(sVh, (2 * ks + 1) * 16 + g * 4, db * 16, c16);
                const uint2 l0 = v_rows4p<VROW>(sVl, (2 * ks) * 16 + g * 4, db * 16, c16), l1 = v_rows4p<VROW>(sVl, (2 * ks + 1) * 16 + g * 4, db * 16, c16);
                const uint4 vh = make_uint4(h0.x, h0.y, h1.x, h1.y), vl = make_uint4(l0.x, l0.y, l1.x, l1.y);
#pragma unroll
                for (int u = 0; u < QG; ++u) o[u][db] = mma3(vh, vl, pfh[u][ks], pfl[u][ks], o[u][db]);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: O^T: query = this lane's c16, d = db*16 + g*4 + r -> one 16-byte f32 store per (query, db) ----
#pragma unroll
    for (int u = 0; u < QG; ++u) {
        const int qo = q0 + (wid * QG + u) * 16 + c16;
        if (qo >= a.Lq) continue;
        const float inv = l_run[u] > 0.f ? 1.0f / l_run[u] : 0.f;
        float *op = a.out + (b * a.q_batch_rows + qo) * a.ldo + h * HD + g * 4;
        if (a.out_packed) {
            float *rowp = a.out + (b * a.q_batch_rows + qo) * a.ldo;
#pragma unroll
            for (int db = 0; db < 4; ++db)
                store_split4_attn(rowp, h * HD + db * 16 + g * 4, o[u][db][0] * inv, o[u][db][1] * inv, o[u][db][2] * inv, o[u][db][3] * inv);
        } else {
#pragma unroll
            for (int db = 0; db < 4; ++db)
                *reinterpret_cast<float4 *>(op + db * 16) = make_float4(o[u][db][0] * inv, o[u][db][1] * inv, o[u][db][2] * inv, o[u][db][3] * inv);
        }
        if (a.lse && g == 0) a.lse[(b * a.q_batch_rows + qo) * a.H + h] = l_run[u] > 0.f ? m_run[u] + log2f(l_run[u]) : -INFINITY;
    }
}


// ---- split-operand attention on PACKED q | k | v (round 4; dtype code 4 + 32) ------------------------------------------------------------
// The qkv projection's RoPE epilogue writes q | k | v in the packed (hi, lo) form of the split class (vs_gemm_split(_packed), epilogue
// 4 + 16: per block of 32 columns 32 hi halves then 32 lo halves, gemm_common.h store_split4), so a head's row of q, k or v is 256 bytes =
// [hi 0..31 | lo 0..31 | hi 32..63 | lo 32..63] (positions; the column order inside a block is the packed permutation, the same for q and
// k, so dot products need no un-permutation, and V's column position IS the output's packed position).  What that buys here:
//   * staging is plain LDS-DMA (global_load_lds_dwordx4): no conversion VALU, no staging registers, the next tile in flight under the
//     current tile's MFMAs, one barrier per tile.  attention_split_kernel converted every K / V tile again in EVERY workgroup that
//     staged it (17 x for the 2064-key video attention);
//   * LDS read traffic per FLOP is HALVED: PMC on attention_split_kernel showed the LDS pipe as its busiest unit (47 % + 20 % conflict
//     cycles) because each of the four waves read the whole K and V tile (hi and lo) for only 32 queries.  Here the four waves of a
//     workgroup are 2 query halves x 2 KEY halves: wave (qh, kh) owns 64 queries (four 16-query MFMA groups) and keys kh*32 .. kh*32+31
//     of every 64-key tile, so it reads half a tile for twice the queries, at the same 2 workgroups x 4 waves per CU and the same MFMA
//     count per wave and tile (96).  Each wave runs its own online softmax over its keys; the two key halves of a query meet ONCE, after
//     the last tile, through LDS (m, l, O of the kh = 1 wave; flash-decoding style merge);
//   * the grid is linear and XCD-aware: all query tiles of one (batch, head) are consecutive logical blocks mapped to ONE XCD
//     (gemm256.h's remap), so its K / V (1 MB for the video attention) is fetched into one L2 instead of eight.
// LDS tiles are unpadded 256-byte rows; 16-byte chunk index XOR (key & 15) for K (conflict-free ds_read_b128 over 16 keys), 32-byte
// chunk index XOR (key & 7) for V (the four rows of a ds_read_b64_tr_b16 in four different 32-byte windows); the swizzle is applied on
// the GLOBAL side of the DMA.  Rows past the end of the key list are duplicates of the last key, masked by the length logic.
struct AttnArgsSP {
    const unsigned char *q, *k, *v;   // packed rows; strides in bytes
    float *out;
    const int32_t *kv_seg, *q_kvlen;
    int nbatch, H, Lq, Lk, nqt;
    long long q_batch_rows, k_batch_rows;
    long long ldq_b, ldk_b, ldv_b;
    int ldo;
    float scale_log2e;
    float *lse;
    int out_packed;
};

typedef void __attribute__((address_space(3))) *lds_ptr_sp_t;
__device__ __forceinline__ void glds16_sp(const void *gp, unsigned lds_off) {   // (inline asm: the compiler adds no waits of its own)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_off), "v"(gp) : "memory");
}

// online-softmax step of one 16-query group over this wave's 32 keys of a tile (softmax_tile for two 16-key blocks).  VALU work does NOT
// hide under MFMAs on gfx950 (tools/probe/mfma_valu_overlap.hip: an MFMA-only wave and an FMA-only wave sharing a SIMD take the SUM of
// their times; only transcendentals overlap), so every instruction here is paid for in full:
//   * l_run is a PER-LANE partial sum (the lane's own keys): the maximum is made uniform over the four lanes of a query column every
//     tile, so the rescale factor is uniform too and the cross-lane row sum can wait for the end of the kernel (one rows_sum per query
//     group instead of one per tile);
//   * MASKED = false: every score is finite, the running maximum is finite after the first tile: no -inf guard.
template <bool MASKED, bool PLO = true>
__device__ __forceinline__ void softmax_half(f4 (&st)[2], float scale_log2e, float &m_run, float &l_run, f4 (&o)[4], uint4 &pf) {
    float mx = fmaxf(fmaxf(st[0][0], st[0][1]), st[0][2]);
    mx = fmaxf(fmaxf(mx, st[0][3]), st[1][0]);
    mx = fmaxf(fmaxf(mx, st[1][1]), st[1][2]);
    mx = fmaxf(mx, st[1][3]);
    mx = rows_max(mx) * scale_log2e;
    const float m_new = vmax2(m_run, mx);
    const float m_use = (MASKED && m_new == -INFINITY) ? 0.f : m_new;
    const f2v sc = f2v{scale_log2e, scale_log2e}, neg_m = f2v{-m_use, -m_use};
    f2v acc = f2v{0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f2v x = f2v{st[nb][2 * h], st[nb][2 * h + 1]} * sc + neg_m;
            f2v p;
            p.x = __builtin_amdgcn_exp2f(x.x);
            p.y = __builtin_amdgcn_exp2f(x.y);
            st[nb][2 * h] = p.x; st[nb][2 * h + 1] = p.y;
            acc += p;
        }
    if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        l_run *= alpha;
#pragma unroll
        for (int db = 0; db < 4; ++db) o[db] *= alpha;
        m_run = m_new;
    }
    if constexpr (PLO) l_run += acc.x + acc.y;
    // COMPILER-VISIBLE converts: pf feeds the P V MFMAs, and the VALU-write -> MFMA-read wait states come from the compiler's hazard
    // recogniser, which does not look inside inline asm (pack2<> is asm: with it the third query group's rows came out 5e-5 off whenever the
    // scheduler placed its first P V MFMA right behind the convert -- the bug class of gemm_common.h split8 / relu_f32_lds, met again here)
    typedef _Float16 h2s_ __attribute__((ext_vector_type(2)));
    typedef float f2s_ __attribute__((ext_vector_type(2)));
    pf.x = __builtin_bit_cast(unsigned, __builtin_convertvector(f2s_{st[0][0], st[0][1]}, h2s_));
    pf.y = __builtin_bit_cast(unsigned, __builtin_convertvector(f2s_{st[0][2], st[0][3]}, h2s_));
    pf.z = __builtin_bit_cast(unsigned, __builtin_convertvector(f2s_{st[1][0], st[1][1]}, h2s_));
    pf.w = __builtin_bit_cast(unsigned, __builtin_convertvector(f2s_{st[1][2], st[1][3]}, h2s_));
    if constexpr (!PLO) {
        // P as one f16: the row sum is taken over the ROUNDED weights the P V product uses, so the output is an exact weighted mean of V
        // under weights p (1 + e), |e| <= 2^-11 -- the rounding no longer shifts the mean, only the spread around it
        const h2s_ one2 = h2s_{(_Float16)1.0f, (_Float16)1.0f};
        float a2 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2s_, pf.x), one2, 0.f, false);
        a2 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2s_, pf.y), one2, a2, false);
        a2 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2s_, pf.z), one2, a2, false);
        l_run += __builtin_amdgcn_fdot2(__builtin_bit_cast(h2s_, pf.w), one2, a2, false);
    }
}

// One tile of one wave: its 32 keys (rows kw .. kw + 31 of the staged tile) against its 64 queries.  Straight-line code: ALLQ (all four
// query groups hold real queries) and MASKED (some query of the wave does not see all 32 keys) are compile-time, so the 96 MFMAs, the four
// softmax steps and the 24 LDS reads of a tile form ONE basic block the scheduler can interleave.  kofs / vofs: per-lane byte offsets inside a
// K / V tile (swizzle applied), loop invariant; the ring slot is a compile-time constant, so every LDS address is register + immediate.
template <int NG, bool ALLQ, bool MASKED, bool PLO>
__device__ __forceinline__ void sp_tile(const unsigned char *sK, const unsigned char *sV, const int (&kofs)[2][2], const int (&vofs)[8], int nact,
                                        const int (&my_len)[NG], int kbase, int g, float scale_log2e, const uint4 (&qfh)[NG][2], const uint4 (&qfl)[NG][2],
                                        float (&m_run)[NG], float (&l_run)[NG], f4 (&o)[NG][4]) {
    f4 st[NG][2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
        for (int u = 0; u < NG; ++u) st[u][nb] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint4 kh_ = *reinterpret_cast<const uint4 *>(sK + kofs[0][ks] + nb * 4096);
            const uint4 kl_ = *reinterpret_cast<const uint4 *>(sK + kofs[1][ks] + nb * 4096);
#pragma unroll
            for (int u = 0; u < NG; ++u)
                if (ALLQ || u < nact) st[u][nb] = mma3(kh_, kl_, qfh[u][ks], qfl[u][ks], st[u][nb]);
        }
    }
    uint4 pfh[NG], pfl[NG];
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        if (!ALLQ && u >= nact) continue;
        if (MASKED) {
            const int lim = my_len[u] - kbase - g * 4;          // keys kbase + nb*16 + g*4 + r are visible while nb*16 + r < lim
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (nb * 16 + r >= lim) st[u][nb][r] = -INFINITY;
        }
        softmax_half<MASKED, PLO>(st[u], scale_log2e, m_run[u], l_run[u], o[u], pfh[u]);     // st now holds P (f32), pfh its rne16
        const unsigned hh[4] = {pfh[u].x, pfh[u].y, pfh[u].z, pfh[u].w};
        unsigned ll[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int w = 0; w < (PLO ? 4 : 0); ++w) {
            // lo = rne16(p - float(hi)): the difference on v_fma_mix_f32 (f16 source operand, exact f32 result: one instruction instead of a
            // convert and a subtract), the rounding on a COMPILER-VISIBLE convert -- the value that feeds the P V MFMAs must come out of an
            // instruction the hazard recogniser sees (VALU write -> MFMA read wait states); asm -> VALU needs no software wait
            typedef _Float16 h2_ __attribute__((ext_vector_type(2)));
            typedef float f2_ __attribute__((ext_vector_type(2)));
            const float p0 = st[u][w >> 1][(w & 1) * 2], p1 = st[u][w >> 1][(w & 1) * 2 + 1];
            float r0, r1;
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hh[w]), "v"(p0));
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hh[w]), "v"(p1));
            ll[w] = __builtin_bit_cast(unsigned, __builtin_convertvector(f2_{r0, r1}, h2_));
        }
        pfl[u] = make_uint4(ll[0], ll[1], ll[2], ll[3]);
    }
    // ---- O^T += V^T P^T over the wave's 32 keys: V fragments through the transpose read, once for the four query groups.
    // db = 16-position block of the head's packed row: (db >> 1) = 32-column block, (db & 1) = its first / second 16 positions
    typedef tr4v_t __attribute__((address_space(3))) *trp_t;
    auto trd = [&](int ofs) -> uint2 {
        return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(const_cast<unsigned char *>(sV + ofs))));
    };
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        const int c32h = (db >> 1) * 4 + (db & 1), c32l = c32h + 2;           // 32-byte chunk of the hi / lo positions
        const uint2 h0 = trd(vofs[c32h]), h1 = trd(vofs[c32h] + 4096), l0 = trd(vofs[c32l]), l1 = trd(vofs[c32l] + 4096);
        const uint4 vh = make_uint4(h0.x, h0.y, h1.x, h1.y), vl = make_uint4(l0.x, l0.y, l1.x, l1.y);
#pragma unroll
        for (int u = 0; u < NG; ++u)
            if (ALLQ || u < nact) {
                if constexpr (PLO) {
                    o[u][db] = mma3(vh, vl, pfh[u], pfl[u], o[u][db]);
                } else {      // P as ONE f16 (p in [0, 1]: its lo half carries <= 2^-12 p): V_lo P_hi + V_hi P_hi, two MFMAs instead of three
                    o[u][db] = mfma<false>(vl, pfh[u], o[u][db]);
                    o[u][db] = mfma<false>(vh, pfh[u], o[u][db]);
                }
            }
    }
}

template <int NG, bool PLO = true>
__global__ void __launch_bounds__(256, 2) attention_sp_kernel(const AttnArgsSP a) {
    constexpr int QW = 16 * NG;                                               // queries per wave (NG MFMA groups of 16), 2 * QW per workgroup
    constexpr int TILE_B = KB * 256;                                          // one K (or V) tile: 64 keys x 256 bytes
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2][2][TILE_B];  // [ring slot][K | V]
    __shared__ int s_maxlen;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c16 = lane & 15;
    const int qh = wid >> 1, kh = wid & 1;
    // ---- XCD-aware linear block id -> (batch, head, query tile): consecutive logical ids (one (batch, head)'s query tiles) share an XCD
    const int nwg = a.nqt * a.H * a.nbatch;
    int bid = blockIdx.x;
    {
        const int qn = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (qn + 1) : r * (qn + 1) + (xcd - r) * qn) + idx;
    }
    const int qt = bid % a.nqt, h = (bid / a.nqt) % a.H, b = bid / (a.nqt * a.H);
    const int q0 = qt * (2 * QW) + qh * QW;
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_sp_t)&smem[0][0][0];

    int base0, len0, base1, len1;
    if (a.kv_seg) {
        base0 = a.kv_seg[4 * b + 0]; len0 = a.kv_seg[4 * b + 1]; base1 = a.kv_seg[4 * b + 2]; len1 = a.kv_seg[4 * b + 3];
    } else {
        base0 = (int)(b * a.k_batch_rows); len0 = a.Lk; base1 = 0; len1 = 0;
    }
    const int Lk = len0 + len1;

    // ---- Q fragments: straight from the packed rows (hi chunk g, lo chunk g of the head's two 32-column blocks)
    int my_len[NG];
    uint4 qfh[NG][2], qfl[NG][2];
    int wave_len = 0, wave_min = 0x7fffffff, nact = 0;
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        const int qi = q0 + u * 16 + c16;
        const bool qvalid = qi < a.Lq;
        const long long qrow = b * a.q_batch_rows + (qvalid ? qi : a.Lq - 1);
        int ml = Lk;
        if (a.q_kvlen && qvalid) ml = min(Lk, a.q_kvlen[(long long)b * a.Lq + qi]);
        if (!qvalid) ml = 0;
        my_len[u] = ml;
        wave_len = max(wave_len, ml);
        wave_min = min(wave_min, ml);
        if (q0 + u * 16 < a.Lq) nact = u + 1;            // (wave-uniform: groups 0 .. nact-1 hold at least one real query)
        const unsigned char *qp = a.q + qrow * a.ldq_b + h * 256 + g * 16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qfh[u][ks] = *reinterpret_cast<const uint4 *>(qp + ks * 128);
            qfl[u][ks] = *reinterpret_cast<const uint4 *>(qp + ks * 128 + 64);
        }
    }
    // The Q fragments must have LANDED, as far as the compiler's wait-count tracking is concerned, before the tile loop: the loop's DMA is
    // issued from inline asm the compiler does not see, so any s_waitcnt vmcnt(N) it inserts inside the loop for THESE loads (it places the
    // wait at the first use, i.e. at the first MFMAs of the loop body) also waits for the next tile's DMA -- which made the prefetch
    // synchronous (ISA: vmcnt(7) .. vmcnt(0) in front of the S = K Q^T MFMAs).  An asm that reads the registers forces the wait here.
#pragma unroll
    for (int u = 0; u < NG; ++u)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            asm volatile("" ::"v"(qfh[u][ks].x), "v"(qfh[u][ks].y), "v"(qfh[u][ks].z), "v"(qfh[u][ks].w), "v"(qfl[u][ks].x), "v"(qfl[u][ks].y),
                         "v"(qfl[u][ks].z), "v"(qfl[u][ks].w));
#pragma unroll
    for (int o_ = 32; o_ > 0; o_ >>= 1) {
        wave_len = max(wave_len, __shfl_xor(wave_len, o_, 64));
        wave_min = min(wave_min, __shfl_xor(wave_min, o_, 64));
    }
    wave_len = __builtin_amdgcn_readfirstlane(wave_len);
    wave_min = __builtin_amdgcn_readfirstlane(wave_min);     // the shortest key list among the wave's 64 query slots (0 with a missing query)
    int maxlen = Lk;                                         // keys the workgroup has to stage: all of them unless every query has a shorter prefix
    if (a.q_kvlen) {                                         // (kernel-uniform)
        if (tid == 0) s_maxlen = 0;
        __syncthreads();
        if (lane == 0) atomicMax(&s_maxlen, wave_len);
        __syncthreads();
        maxlen = s_maxlen;
    }

    f4 o[NG][4];
    float m_run[NG], l_run[NG];
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        m_run[u] = -INFINITY; l_run[u] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[u][i] = f4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- staging: per tile 16 pieces of 1 KiB (4 key rows) for K and for V; wave w issues pieces w, w + 4, w + 8, w + 12 of each.  Lane l of
    // a piece = (row l >> 4, LDS slot l & 15); (row & 15) == (w * 4 + (l >> 4)) & 15 for every piece of this wave, so the swizzled source
    // chunk is a per-thread constant
    const int r0 = wid * 4 + (lane >> 4);
    const int kchunk = (lane & 15) ^ (r0 & 15);
    const int vchunk = ((((lane & 15) >> 1) ^ (r0 & 7)) << 1) | (lane & 1);
    const unsigned char *kbase = a.k + h * 256 + kchunk * 16, *vbase = a.v + h * 256 + vchunk * 16;
    auto key_row = [&](int j) -> long long {
        j = min(j, Lk - 1);
        return j < len0 ? (long long)base0 + j : (long long)base1 + (j - len0);
    };
    auto issue = [&](int kt, int slot) {
        const unsigned ldsk = lds0 + (unsigned)(slot * 2) * TILE_B + (unsigned)wid * 1024u, ldsv = ldsk + TILE_B;
        long long rowbase = -1;                  // whole tile inside one segment (and inside the list): rows are base + r
        if (kt + KB <= len0) rowbase = (long long)base0 + kt;
        else if (kt >= len0 && kt + KB <= Lk) rowbase = (long long)base1 + (kt - len0);
        if (rowbase >= 0) {
            const unsigned char *kp = kbase + (rowbase + r0) * a.ldk_b, *vp = vbase + (rowbase + r0) * a.ldv_b;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                glds16_sp(kp + (long long)j * 16 * a.ldk_b, __builtin_amdgcn_readfirstlane(ldsk + (unsigned)j * 4096u));
                glds16_sp(vp + (long long)j * 16 * a.ldv_b, __builtin_amdgcn_readfirstlane(ldsv + (unsigned)j * 4096u));
            }
        } else {                                 // a tile that straddles the segment boundary or the end of the list: per-row lookup
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long long r = key_row(kt + j * 16 + r0);
                glds16_sp(kbase + r * a.ldk_b, __builtin_amdgcn_readfirstlane(ldsk + (unsigned)j * 4096u));
                glds16_sp(vbase + r * a.ldv_b, __builtin_amdgcn_readfirstlane(ldsv + (unsigned)j * 4096u));
            }
        }
    };
    if (maxlen > 0) issue(0, 0);

    // ---- loop-invariant LDS offsets of this lane's fragments (bytes inside a K / V tile, swizzle applied)
    const int kw = kh * 32;                       // this wave's keys inside a tile
    int kofs[2][2], vofs[8];
#pragma unroll
    for (int hl = 0; hl < 2; ++hl)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) kofs[hl][ks] = (kw + c16) * 256 + (((ks * 8 + hl * 4 + g) ^ c16) << 4);     // (+ nb * 4096)
    {
        const int row = kw + g * 4 + (c16 >> 2), inb = (c16 & 3) * 8;
#pragma unroll
        for (int c = 0; c < 8; ++c) vofs[c] = row * 256 + ((c ^ (row & 7)) << 5) + inb;                                // (+ 4096 for the keys + 16)
    }
    const bool allq = nact == NG;
    auto tile = [&](int kt, const unsigned char *sK, const unsigned char *sV) {
        const int kb = kt + kw;
        if (kb >= wave_len) return;                          // none of this wave's queries sees any of its keys of this tile
        const bool masked = kb + 32 > wave_min;              // (wave-uniform)
        if (allq) {
            if (masked) sp_tile<NG, true, true, PLO>(sK, sV, kofs, vofs, nact, my_len, kb, g, a.scale_log2e, qfh, qfl, m_run, l_run, o);
            else sp_tile<NG, true, false, PLO>(sK, sV, kofs, vofs, nact, my_len, kb, g, a.scale_log2e, qfh, qfl, m_run, l_run, o);
        } else {
            sp_tile<NG, false, true, PLO>(sK, sV, kofs, vofs, nact, my_len, kb, g, a.scale_log2e, qfh, qfl, m_run, l_run, o);
        }
    };
    for (int kt = 0; kt < maxlen; kt += 2 * KB) {            // two tiles per trip: the ring slot is a compile-time constant
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of tile kt have landed ...
        __syncthreads();                                       // ... and everybody's; all waves are done with the other slot
        if (kt + KB < maxlen) issue(kt + KB, 1);
        tile(kt, &smem[0][0][0], &smem[0][1][0]);
        if (kt + KB >= maxlen) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 2 * KB < maxlen) issue(kt + 2 * KB, 0);
        tile(kt + KB, &smem[1][0][0], &smem[1][1][0]);
    }

    // ---- merge the two key halves of every query (once): the kh = 1 wave hands (m, l, O) to its kh = 0 partner through LDS
#pragma unroll
    for (int u = 0; u < NG; ++u) l_run[u] = rows_sum(l_run[u]);       // the per-lane partial sums of a query column meet here, once
    __syncthreads();
    float *xch = reinterpret_cast<float *>(&smem[0][0][0]) + qh * (18 * 64 * 4);      // [u][m | l | 16 x O][lane]: 18 KiB per query half
    if (kh == 1) {
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            float *x = xch + u * 18 * 64 + lane;
            x[0] = m_run[u]; x[64] = l_run[u];
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int r = 0; r < 4; ++r) x[(2 + db * 4 + r) * 64] = o[u][db][r];
        }
    }
    __syncthreads();
    if (kh == 1) return;
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        const int qo = q0 + u * 16 + c16;
        const float *x = xch + u * 18 * 64 + lane;
        const float m1 = x[0], l1 = x[64];
        const float m = fmaxf(m_run[u], m1);
        const float mu = m == -INFINITY ? 0.f : m;
        const float a0 = __builtin_amdgcn_exp2f(m_run[u] - mu), a1 = __builtin_amdgcn_exp2f(m1 - mu);
        const float l = l_run[u] * a0 + l1 * a1;
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        if (qo >= a.Lq) continue;
        float *rowp = a.out + (b * a.q_batch_rows + qo) * a.ldo;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            float v4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v4[r] = (o[u][db][r] * a0 + x[(2 + db * 4 + r) * 64] * a1) * inv;
            // positions (db & 1) * 16 + g * 4 + r of block (db >> 1) = columns {0, 16}[g & 1] + 4 * ((db & 1) * 2 + (g >> 1)) + r
            const int col = h * HD + (db >> 1) * 32 + ((g & 1) ? 16 : 0) + 4 * ((db & 1) * 2 + (g >> 1));
            if (a.out_packed) store_split4_attn(rowp, col, v4[0], v4[1], v4[2], v4[3]);
            else *reinterpret_cast<float4 *>(rowp + col) = make_float4(v4[0], v4[1], v4[2], v4[3]);
        }
        if (a.lse && g == 0) a.lse[(b * a.q_batch_rows + qo) * a.H + h] = l > 0.f ? m + log2f(l) : -INFINITY;
    }
}

// (Measured and not kept, round 5: folding a segment's single remainder key (257 = 4 tiles + 1 key; 257 + 257 = 8 aligned tiles + 2 keys) into
//  the initial online-softmax state on the VALU -- m = s, l = 1, O = v from 48 v_dot2_f32_f16 per query group -- instead of a fifth / ninth
//  tile: correct (4.8e-7 of float64), and 386 vs 392 us on the frame encoder's shape, 415 vs 408 on the cross-neighbour shape: nothing.  A
//  workgroup of these shapes spends ~2 us of its ~21 us in MFMAs; the rest is the dependent chain Q load -> first DMA -> barrier -> ...
//  -> merge -> store at 8 waves per CU, which a shorter key loop does not shorten.
//  Also measured, kept as an OPT-IN switch (VS_ATTN_PLO=0): P as ONE f16 in the P V product -- two MFMAs instead of three, no lo split --
//  with the row sum taken over the ROUNDED weights (v_dot2_f32_f16 against ones), so that the output stays an exact weighted mean of V under
//  weights p (1 + e), |e| <= 2^-11: 25.95 -> 22.95 ms per step.  With that normalisation the whole encoder moves from <= 1.3e-5 to <= 2.5e-5
//  of the reference's f64 goldens (bar 2e-4), the render PSNR against the oracle chain from 73-79 to 71.5-73.7 dB, the poses from 1.4e-6 to
//  3.7e-6 (bar 2e-5) -- without it (row sum over the exact p): 9e-5, 58.5 dB, 1.4e-5.  It is not the default because the OPERATOR is then
//  f16-class on its own (1e-4 of float64 on random inputs against 5e-7; tests/test_split_path_gpu.py holds the class to 6e-6 per operator).)
// (Measured and not kept, late round 5: the first tile's DMA issued BEFORE the Q loads -- a 257-key workgroup otherwise starts with two dependent
//  memory round trips, Q then tile 0.  Bit-identical, and over 4 x 60 launches per shape same-process (tools/ab_attn_early.py at that commit):
//  encoder 327.6 vs 329.0 us, video 1057.6 vs 1058.9, cross-neighbour 374.2 vs 375.4 -- nothing: with two workgroups per CU the other
//  workgroup's tiles cover the start-up chain; the CU's time is the per-tile work itself.)
// (Measured and not kept, round 3: a resident variant -- one 8-wave workgroup per (frame, head), all K / V converted once into 153 KiB of
// LDS -- runs the frame encoder's 257 x 257 attention at the same 28 ms per step as this tiled kernel: the time is the per-group softmax /
// split VALU work and the 3 x MFMAs, not the re-staging.)

}  // namespace

// ---- reference-precision attention (f32 q | k | v and output, exact f32 MFMA v_mfma_f32_16x16x4_f32; DESIGN 2 "f32 path") ----
// Same semantics and indexing as attention_kernel (key segments, per-query key-prefix lengths, log2-domain logsumexp), a plain
// structure: 4 waves x 16 queries per workgroup, 32-key tiles of K and V staged synchronously in LDS (rows padded to 68 floats:
// conflict-free float4 / float fragment reads), S^T = K Q^T so that a lane owns one query column, online softmax in f32, O += P V
// with the scores used as the A operand as they are.  Summation index maps: QK^T step (j, s) takes d = 16 j + 4 g + s for both
// operands (g = lane >> 4); P V step (nb, r) takes key = 16 nb + 4 g + r.
struct AttnArgsF32 {
    const float *q, *k, *v;
    float *out;
    const int32_t *kv_seg, *q_kvlen;
    int nbatch, H, Lq, Lk;
    long long q_batch_rows, k_batch_rows;
    int ldq, ldk, ldv, ldo;
    float scale_log2e;
    float *lse;
};

__global__ void __launch_bounds__(256) attention_f32_kernel(const AttnArgsF32 a) {
    constexpr int TK = 32, ROW = HD + 4;
    __shared__ __attribute__((aligned(16))) float sK[TK * ROW];
    __shared__ __attribute__((aligned(16))) float sV[TK * ROW];
    __shared__ int s_maxlen;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 64;
    int base0, len0, base1, len1;
    if (a.kv_seg) {
        base0 = a.kv_seg[4 * b + 0]; len0 = a.kv_seg[4 * b + 1]; base1 = a.kv_seg[4 * b + 2]; len1 = a.kv_seg[4 * b + 3];
    } else {
        base0 = (int)(b * a.k_batch_rows); len0 = a.Lk; base1 = 0; len1 = 0;
    }
    const int Lk = len0 + len1;
    const int qi = q0 + wid * 16 + c16;
    const bool qvalid = qi < a.Lq;
    const long long qrow = b * a.q_batch_rows + (qvalid ? qi : a.Lq - 1);
    int my_len = Lk;
    if (a.q_kvlen && qvalid) my_len = min(Lk, a.q_kvlen[(long long)b * a.Lq + qi]);
    if (!qvalid) my_len = 0;
    float4 qf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) qf[j] = *reinterpret_cast<const float4 *>(a.q + qrow * a.ldq + h * HD + 16 * j + 4 * g);
    int wave_len = my_len;
#pragma unroll
    for (int o_ = 32; o_ > 0; o_ >>= 1) wave_len = max(wave_len, __shfl_xor(wave_len, o_, 64));
    if (tid == 0) s_maxlen = 0;
    __syncthreads();
    if (lane == 0) atomicMax(&s_maxlen, wave_len);
    __syncthreads();
    const int maxlen = s_maxlen;

    f4 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = f4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    auto key_row = [&](int j) -> long long {
        j = min(j, Lk - 1);
        return j < len0 ? (long long)base0 + j : (long long)base1 + (j - len0);
    };
    const int s_key = tid >> 3, s_c = (tid & 7) * 8;   // staging: one key row, 8 floats of K and of V per thread
    for (int kt = 0; kt < maxlen; kt += TK) {
        {
            const long long r = key_row(kt + s_key);
            const float *kp = a.k + r * a.ldk + h * HD + s_c, *vp = a.v + r * a.ldv + h * HD + s_c;
            const float4 k0 = *reinterpret_cast<const float4 *>(kp), k1 = *reinterpret_cast<const float4 *>(kp + 4);
            const float4 v0 = *reinterpret_cast<const float4 *>(vp), v1 = *reinterpret_cast<const float4 *>(vp + 4);
            *reinterpret_cast<float4 *>(&sK[s_key * ROW + s_c]) = k0; *reinterpret_cast<float4 *>(&sK[s_key * ROW + s_c + 4]) = k1;
            *reinterpret_cast<float4 *>(&sV[s_key * ROW + s_c]) = v0; *reinterpret_cast<float4 *>(&sV[s_key * ROW + s_c + 4]) = v1;
        }
        __syncthreads();
        if (kt < wave_len) {
            f4 st[2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                st[nb] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 kf = *reinterpret_cast<const float4 *>(&sK[(nb * 16 + c16) * ROW + 16 * j + 4 * g]);
                    st[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[j].x, st[nb], 0, 0, 0);
                    st[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[j].y, st[nb], 0, 0, 0);
                    st[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[j].z, st[nb], 0, 0, 0);
                    st[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[j].w, st[nb], 0, 0, 0);
                }
            }
            // lane holds S^T[key = kt + 16 nb + 4 g + r][query c16]; mask, then the online softmax of this query column
            float mx = -INFINITY;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sc = (kt + nb * 16 + g * 4 + r < my_len) ? st[nb][r] * a.scale_log2e : -INFINITY;
                    st[nb][r] = sc;
                    mx = fmaxf(mx, sc);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = m_new == -INFINITY ? 1.f : exp2f(m_run - m_new);
            float ps = 0.f;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = m_new == -INFINITY ? 0.f : exp2f(st[nb][r] - m_new);
                    st[nb][r] = p;
                    ps += p;
                }
            ps += __shfl_xor(ps, 16, 64);
            ps += __shfl_xor(ps, 32, 64);
            l_run = l_run * alpha + ps;
            m_run = m_new;
            // O rows are queries g*4 + r: fetch their rescale factors from the lanes that own those query columns
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float al = __shfl(alpha, g * 4 + r, 64);
#pragma unroll
                for (int db = 0; db < 4; ++db) o[db][r] *= al;
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float *vr = &sV[(nb * 16 + g * 4 + r) * ROW + c16];
#pragma unroll
                    for (int db = 0; db < 4; ++db) o[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(st[nb][r], vr[db * 16], o[db], 0, 0, 0);
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float lr = __shfl(l_run, g * 4 + r, 64);
        const int qo = q0 + wid * 16 + g * 4 + r;
        if (qo >= a.Lq) continue;
        const float inv = lr > 0.f ? 1.0f / lr : 0.f;
        float *op = a.out + (b * a.q_batch_rows + qo) * a.ldo + h * HD + c16;
#pragma unroll
        for (int db = 0; db < 4; ++db) op[db * 16] = o[db][r] * inv;
    }
    if (a.lse && g == 0 && qvalid) a.lse[(b * a.q_batch_rows + qi) * a.H + h] = l_run > 0.f ? m_run + log2f(l_run) : -INFINITY;
}

extern "C" int vs_attention_lse(const void *q, const void *k, const void *v, void *out, int32_t nbatch, int32_t H, int32_t Lq,
                                int32_t Lk, int64_t q_batch_rows, int64_t k_batch_rows, int32_t ldq, int32_t ldk, int32_t ldv,
                                int32_t ldo, const int32_t *kv_seg, const int32_t *q_kvlen, float scale, int32_t dtype, float *lse,
                                vs_stream_t stream_);

extern "C" int vs_attention(const void *q, const void *k, const void *v, void *out, int32_t nbatch, int32_t H, int32_t Lq,
                            int32_t Lk, int64_t q_batch_rows, int64_t k_batch_rows, int32_t ldq, int32_t ldk, int32_t ldv,
                            int32_t ldo, const int32_t *kv_seg, const int32_t *q_kvlen, float scale, int32_t dtype,
                            vs_stream_t stream_) {
    return vs_attention_lse(q, k, v, out, nbatch, H, Lq, Lk, q_batch_rows, k_batch_rows, ldq, ldk, ldv, ldo, kv_seg, q_kvlen, scale, dtype,
                            nullptr, stream_);
}

extern "C" int vs_attention_lse(const void *q, const void *k, const void *v, void *out, int32_t nbatch, int32_t H, int32_t Lq,
                                int32_t Lk, int64_t q_batch_rows, int64_t k_batch_rows, int32_t ldq, int32_t ldk, int32_t ldv,
                                int32_t ldo, const int32_t *kv_seg, const int32_t *q_kvlen, float scale, int32_t dtype, float *lse,
                                vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(q && k && v && out, "vs_attention: null pointer");
    VS_CHECK(nbatch >= 0 && H > 0 && Lq >= 0, "vs_attention: bad sizes");
    VS_CHECK(kv_seg || Lk > 0, "vs_attention: Lk must be positive when kv_seg is null");
    VS_CHECK(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0, "vs_attention: row strides must be multiples of 8 elements");
    VS_CHECK(((uintptr_t)q & 15) == 0 && ((uintptr_t)k & 15) == 0 && ((uintptr_t)v & 15) == 0, "vs_attention: 16-byte alignment required");
    // split class (4): + 16 = the output in the packed (hi, lo) form (ldo % 32 == 0, 128-byte aligned); + 32 = q | k | v ALREADY packed
    // (the qkv GEMM's RoPE epilogue wrote them so: vs_gemm_split(_packed) epilogue 4 + 16) -> attention_sp_kernel
    const int out_packed = (dtype == 20 || dtype == 52) ? 1 : 0, in_packed = (dtype == 36 || dtype == 52) ? 1 : 0;
    if (out_packed || in_packed) dtype = 4;
    VS_CHECK(dtype == 1 || dtype == 2 || dtype == 3 || dtype == 4, "vs_attention: dtype must be 1 (f16), 2 (bf16), 3 (f32) or 4 (split: f32 data, 3 x f16 MFMA; + 16 packed output, + 32 packed q | k | v)");
    VS_CHECK(!out_packed || (ldo % 32 == 0 && ((uintptr_t)out & 127) == 0), "vs_attention: a packed output needs ldo %% 32 == 0 and a 128-byte aligned buffer");
    VS_CHECK(H <= 65535 && nbatch <= 65535, "vs_attention: grid too large");
    if (nbatch == 0 || Lq == 0) return 0;
    if (dtype == 4) {
        VS_CHECK(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)out & 15) == 0, "vs_attention: split operands need 16-byte aligned f32 rows");
        if (in_packed) {
            VS_CHECK(ldq % 32 == 0 && ldk % 32 == 0 && ldv % 32 == 0 && (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 127) == 0,
                     "vs_attention: packed q | k | v need row strides that are multiples of 32 (4-byte units) and 128-byte aligned head-0 columns");
            AttnArgsSP f;
            f.q = (const unsigned char *)q; f.k = (const unsigned char *)k; f.v = (const unsigned char *)v; f.out = (float *)out; f.kv_seg = kv_seg; f.q_kvlen = q_kvlen;
            static const int ng = [] { const char *e = getenv("VS_ATTN_SP_NG"); return e ? atoi(e) : 3; }();
            f.nbatch = nbatch; f.H = H; f.Lq = Lq; f.Lk = Lk; f.nqt = vs::cdiv(Lq, 32 * ng); f.q_batch_rows = q_batch_rows; f.k_batch_rows = k_batch_rows;
            f.ldq_b = 4LL * ldq; f.ldk_b = 4LL * ldk; f.ldv_b = 4LL * ldv; f.ldo = ldo; f.scale_log2e = scale * 1.4426950408889634f; f.lse = lse;
            f.out_packed = out_packed;
            const long long nwg = (long long)f.nqt * H * nbatch;
            VS_CHECK(nwg < (1LL << 31), "vs_attention: grid too large");
            static const int plo = [] { const char *e = getenv("VS_ATTN_PLO"); return e ? atoi(e) : 1; }();
            if (ng == 4) hipLaunchKernelGGL(attention_sp_kernel<4>, dim3((unsigned)nwg), dim3(256), 0, stream, f);
            else if (ng == 2) hipLaunchKernelGGL(attention_sp_kernel<2>, dim3((unsigned)nwg), dim3(256), 0, stream, f);
            else if (plo == 0 || (plo == 2 && q_kvlen) || (plo == 3 && !q_kvlen)) hipLaunchKernelGGL((attention_sp_kernel<3, false>), dim3((unsigned)nwg), dim3(256), 0, stream, f);
            else hipLaunchKernelGGL(attention_sp_kernel<3>, dim3((unsigned)nwg), dim3(256), 0, stream, f);
            VS_HIP(hipGetLastError());
            return 0;
        }
        AttnArgsSplit f;
        { static const int x_ = [] { const char *e = getenv("VS_ATTN_SPLIT_XCD"); return e ? atoi(e) : 1; }(); f.xcd = x_; }
        f.q = (const float *)q; f.k = (const float *)k; f.v = (const float *)v; f.out = (float *)out; f.kv_seg = kv_seg; f.q_kvlen = q_kvlen;
        f.nbatch = nbatch; f.H = H; f.Lq = Lq; f.Lk = Lk; f.q_batch_rows = q_batch_rows; f.k_batch_rows = k_batch_rows;
        f.ldq = ldq; f.ldk = ldk; f.ldv = ldv; f.ldo = ldo; f.scale_log2e = scale * 1.4426950408889634f; f.lse = lse; f.out_packed = out_packed;
        const int Lk_eff = kv_seg ? 2 * Lk : Lk;
        static const int force_qg4 = [] { const char *e = getenv("VS_ATTN_SPLIT_QG"); return e ? atoi(e) : 0; }();
        bool big = (long long)vs::cdiv(Lq, 128) * H * nbatch >= 512 && Lk_eff > 256;
        if (force_qg4) big = force_qg4 == 2;
        if (big) hipLaunchKernelGGL((attention_split_kernel<2>), dim3(vs::cdiv(Lq, 128), H, nbatch), dim3(256), 0, stream, f);
        else hipLaunchKernelGGL((attention_split_kernel<1>), dim3(vs::cdiv(Lq, 64), H, nbatch), dim3(256), 0, stream, f);
        VS_HIP(hipGetLastError());
        return 0;
    }
    if (dtype == 3) {
        AttnArgsF32 f;
        f.q = (const float *)q; f.k = (const float *)k; f.v = (const float *)v; f.out = (float *)out; f.kv_seg = kv_seg; f.q_kvlen = q_kvlen;
        f.nbatch = nbatch; f.H = H; f.Lq = Lq; f.Lk = Lk; f.q_batch_rows = q_batch_rows; f.k_batch_rows = k_batch_rows;
        f.ldq = ldq; f.ldk = ldk; f.ldv = ldv; f.ldo = ldo; f.scale_log2e = scale * 1.4426950408889634f; f.lse = lse;
        hipLaunchKernelGGL(attention_f32_kernel, dim3(vs::cdiv(Lq, 64), H, nbatch), dim3(256), 0, stream, f);
        VS_HIP(hipGetLastError());
        return 0;
    }
    AttnArgs a;
    a.q = (const unsigned short *)q; a.k = (const unsigned short *)k; a.v = (const unsigned short *)v;
    a.out = (unsigned short *)out; a.kv_seg = kv_seg; a.q_kvlen = q_kvlen;
    a.nbatch = nbatch; a.H = H; a.Lq = Lq; a.Lk = Lk; a.q_batch_rows = q_batch_rows; a.k_batch_rows = k_batch_rows;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.scale_log2e = scale * 1.4426950408889634f;
    a.lse = lse;
    // 128 queries per workgroup when that still leaves >= 2 workgroups per CU; else 64
    static const int force_qg = [] { const char *e = getenv("VS_ATTN_QG"); return e ? atoi(e) : 0; }();
    static const int force_wpe = [] { const char *e = getenv("VS_ATTN_WPE"); return e ? atoi(e) : 0; }();
    static const int force_res = [] { const char *e = getenv("VS_ATTN_RES"); return e ? atoi(e) : -1; }();
    if (!kv_seg && !q_kvlen && Lk <= kResMaxKeys && Lq <= 3 * 8 * 16 && force_res != 0) {
        const int Lkp = (Lk + 15) & ~15;
        const size_t lds = (size_t)(2 * Lkp * KROW) * sizeof(unsigned short);   // K and V, row-major
        dim3 grid(1, H, nbatch);
        if (dtype == 2) {
            VS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&attention_res_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((attention_res_kernel<true>), grid, dim3(512), lds, stream, a);
        } else {
            VS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&attention_res_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((attention_res_kernel<false>), grid, dim3(512), lds, stream, a);
        }
        VS_HIP(hipGetLastError());
        return 0;
    }
    // 128 queries per workgroup only for long key sequences (more MFMAs per staged tile); the 257/516-key shapes are
    // latency bound and run faster with 64-query workgroups at higher occupancy
    const int Lk_eff = kv_seg ? 2 * Lk : Lk;
    bool big = (long long)vs::cdiv(Lq, 128) * H * nbatch >= 512 && Lk_eff > 1024;
    if (force_qg) big = force_qg == 2;
    dim3 block(256);
#define VS_LAUNCH(QG_, WPE_)                                                                                   \
    {                                                                                                          \
        dim3 grid(vs::cdiv(Lq, 64 * QG_), H, nbatch);                                                          \
        if (dtype == 2) hipLaunchKernelGGL((attention_kernel<true, QG_, WPE_>), grid, block, 0, stream, a);    \
        else hipLaunchKernelGGL((attention_kernel<false, QG_, WPE_>), grid, block, 0, stream, a);              \
    }
    if (big) {
        if (force_wpe == 2) VS_LAUNCH(2, 2) else VS_LAUNCH(2, 3)
    } else {
        if (force_wpe == 2) VS_LAUNCH(1, 2) else VS_LAUNCH(1, 4)
    }
#undef VS_LAUNCH
    VS_HIP(hipGetLastError());
    return 0;
}
