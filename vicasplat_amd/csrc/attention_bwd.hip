// Attention backward (head dim 64) for the three attention shapes of attention.hip: same q | k | v addressing, key-prefix
// mask and two-segment key gather.  Backward of the encoder's attention in the training step (autograd.AttentionFn), parity-tested against torch
// autograd.  Flash-style: nothing of size Lq x Lk is stored; the forward saves the log2-domain logsumexp L (vs_attention_lse).
//
//   delta_i = sum_d dO_i O_i                                   attn_delta_kernel      (8 lanes per (row, head))
//   P = exp2(S * scale*log2e - L),  dP = dO V^T,  dS = P o (dP - delta) * scale
//   dQ_i = sum_j dS_ij K_j                                      attn_bwd_dq_kernel     (workgroup = 64 queries, loops key tiles;
//                                                                                        lane = query, as in the forward)
//   dV_j = sum_i P_ij dO_i,   dK_j = sum_i dS_ij Q_i            attn_bwd_dkv_kernel    (workgroup = 64 keys, loops query tiles;
//                                                                                        lane = key); f32 atomics into dK / dV
//                                                               because with key segments a K/V row serves several batch items;
//                                                               without segments 16-bit stores in place (vs_attention_backward16)
// S and dP are recomputed in both kernels (7 instead of 5 MFMA products per (i, j) tile, no atomics on dQ).
#include "common.h"

#include <cstdlib>

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int HD = 64, TB = 64;
// LDS transpose read (ds_read_b64_tr_b16, tools/probe/tr_read.hip): lane t of a 16-lane group supplies the address of 4 consecutive
// halfs of row (row0 + (t >> 2)) at column col0 + (t & 3) * 4 and receives tile[row0 .. row0 + 3][col0 + t].  With it the "k along
// rows" MFMA operands (dO^T, Q^T, K^T) are gathered from the ROW-MAJOR tiles: no transposed LDS images, no packing pass (tr_rows4_sw
// below, on the swizzled 128-byte rows the LDS-DMA staging writes).
typedef short tr4_t __attribute__((ext_vector_type(4)));

struct AttnBwdArgs {
    const unsigned short *q, *k, *v, *o, *dout;
    const float *lse;
    float *delta;
    unsigned short *dq;
    float *dk, *dv;
    unsigned short *dk16, *dv16;   // direct mode (no key segments: a K/V row has one owner): 16-bit stores, no atomics, no zero fill
    const int32_t *kv_seg, *q_kvlen;
    int nbatch, H, Lq, Lk;
    long long q_batch_rows, k_batch_rows;
    int ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
    float scale, scale_log2e;
    // split operand class (dtype 4): q / k / v / dout above are the HI 16-bit images of the f32 tensors, these the LO images (same strides);
    // o32 / dout32 the f32 tensors themselves (delta), dq32 / dk / dv the f32 outputs (dk / dv: atomics with key segments, plain stores without)
    const unsigned short *q_lo, *k_lo, *v_lo, *dout_lo;
    const float *o32, *dout32;
    float *dq32;
    int ldo32, lddo32, lddq32, kv_direct;
    int xcd;     // 1: XCD-aware block ids (block_bhx): the tiles of one (batch, head) run behind ONE L2
};

// (tile, head, batch) of this workgroup.  The hardware deals consecutive workgroups (x fastest) to consecutive XCDs, which puts the tiles of one
// (batch, head) -- they all read that head's whole K / V (dq kernels) or Q / dO (dk|dv kernels) -- behind eight different L2s (counters: 76 GB
// fetched per 8-scene split training step for ~38 GB of operands).  With a.xcd consecutive LOGICAL ids share an XCD (gemm256_kernel's remap).
__device__ __forceinline__ void block_bhx(const AttnBwdArgs &a, int &b, int &h, int &x) {
    if (!a.xcd) { b = blockIdx.z; h = blockIdx.y; x = blockIdx.x; return; }
    const int gx = gridDim.x, gy = gridDim.y, n = gx * gy * gridDim.z;
    const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const int q = n >> 3, r = n & 7, xc = lin & 7, idx = lin >> 3;
    const int l = (xc < r ? xc * (q + 1) : r * (q + 1) + (xc - r) * q) + idx;
    x = l % gx; const int t = l / gx; h = t % gy; b = t / gy;
}

template <bool BF16>
__device__ __forceinline__ f4 mfma(const uint4 &a, const uint4 &b, f4 c) {
    if constexpr (BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf8 *>(&a), *reinterpret_cast<const bf8 *>(&b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const half8 *>(&a), *reinterpret_cast<const half8 *>(&b), c, 0, 0, 0);
}
template <bool BF16>
__device__ __forceinline__ float ld16(unsigned short h) {
    if constexpr (BF16) return __uint_as_float(((unsigned)h) << 16);
    else return (float)*reinterpret_cast<_Float16 *>(&h);
}
template <bool BF16>
__device__ __forceinline__ unsigned pack2(float a, float b) {
    // one packed convert (round to nearest even) instead of two converts + an OR: the softmax / dS path is VALU-bound
    unsigned r;
    if constexpr (BF16) asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    else asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// The same convert for a value that goes straight into an MFMA: compiler-visible instructions, so that the hazard recogniser inserts the
// VALU-write -> MFMA-read wait states (it does not look inside inline asm; attention.hip, pack2v).
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

// 8 lanes per (row, head): each lane multiplies 8 elements (one 16-byte load of O and of dO), three DPP steps sum the 8 lanes.
template <bool BF16>
__global__ void __launch_bounds__(256)
attn_delta_kernel(const AttnBwdArgs a, long long rows) {
    const long long item = ((long long)blockIdx.x * 256 + threadIdx.x) >> 3;  // (row, head)
    const int sub = threadIdx.x & 7;
    float v = 0.f;
    const bool live = item < rows * a.H;
    if (live) {
        const long long row = item / a.H;
        const int h = (int)(item % a.H);
        const uint4 ov = *reinterpret_cast<const uint4 *>(a.o + row * a.ldo + h * HD + sub * 8);
        const uint4 dv = *reinterpret_cast<const uint4 *>(a.dout + row * a.lddo + h * HD + sub * 8);
        const unsigned ow[4] = {ov.x, ov.y, ov.z, ov.w}, dw[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            v += ld16<BF16>((unsigned short)(ow[k] & 0xffffu)) * ld16<BF16>((unsigned short)(dw[k] & 0xffffu)) +
                 ld16<BF16>((unsigned short)(ow[k] >> 16)) * ld16<BF16>((unsigned short)(dw[k] >> 16));
    }
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    if (live && sub == 0) a.delta[item] = v;
}

struct KeyList {
    int base0, len0, base1, len1, Lk;
    __device__ __forceinline__ long long row(int j) const {
        j = min(j, Lk - 1);
        return j < len0 ? (long long)base0 + j : (long long)base1 + (j - len0);
    }
};
__device__ __forceinline__ KeyList key_list(const AttnBwdArgs &a, int b) {
    KeyList kl;
    if (a.kv_seg) {
        kl.base0 = a.kv_seg[4 * b + 0]; kl.len0 = a.kv_seg[4 * b + 1]; kl.base1 = a.kv_seg[4 * b + 2]; kl.len1 = a.kv_seg[4 * b + 3];
    } else {
        kl.base0 = (int)(b * a.k_batch_rows); kl.len0 = a.Lk; kl.base1 = 0; kl.len1 = 0;
    }
    kl.Lk = kl.len0 + kl.len1;
    return kl;
}

// ---- asynchronous staging (round 2): the 64 x 64 tiles are filled by LDS-DMA (global_load_lds_dwordx4: no staging registers, so
// no occupancy cost -- a register-prefetch variant lost more to occupancy than it won) into a two-slot ring, the NEXT tile in flight
// while the current one is multiplied, one barrier per tile.  The synchronous version (load -> LDS store -> barrier -> compute ->
// barrier) spent ~3.3 us per tile and workgroup against ~0.7 us of MFMA + VALU work: latency-bound with 3-4 workgroups per CU.
// A DMA instruction writes 1 KiB contiguous (lane l -> base + 16 l), so rows are unpadded (128 B) and the 16-byte chunk index is XOR-
// swizzled with (row >> 1) & 7 on the GLOBAL side: the sixteen rows of a ds_read_b128 fragment read then fall into sixteen different
// 16-byte slots of the 256-byte bank row, and the four rows of a transpose read into different windows.  Rows past the end of the
// list are duplicates of the last row (a DMA cannot zero-fill): their scores are finite and their probabilities are forced to 0 by
// the existing masks (L = +inf for missing queries; key >= my_len for missing keys: a tile with a missing key is never "full").
typedef void __attribute__((address_space(3))) *lds_ptr_t;
__device__ __forceinline__ void glds16(const void *gp, unsigned lds_off) {   // (inline asm: the compiler adds no waits of its own)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_off), "v"(gp) : "memory");
}
__device__ __forceinline__ int sw_off(int row, int chunk) { return row * HD + ((chunk ^ ((row >> 1) & 7)) << 3); }   // halfs
__device__ __forceinline__ uint2 tr_rows4_sw(const unsigned short *tile, int row0, int col0, int t) {
    typedef tr4_t __attribute__((address_space(3))) *trp_t;
    const int row = row0 + (t >> 2), col = col0 + (t & 3) * 4;
    const unsigned short *p = tile + sw_off(row, col >> 3) + (col & 7);
    const tr4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(const_cast<unsigned short *>(p)));
    return __builtin_bit_cast(uint2, v);
}
// 64 rows x 128 B: wave w issues pieces 2w, 2w + 1 (8 rows each); lane -> (row = piece * 8 + lane / 8, slot = lane % 8)
template <class RowFn>
__device__ __forceinline__ void dma_tile(const unsigned short *src, int ld, int col0, RowFn row_of, unsigned lds_base, int tid) {
    const int w = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int piece = w * 2 + j, row = piece * 8 + (lane >> 3), gchunk = (lane & 7) ^ ((row >> 1) & 7);
        glds16(src + row_of(row) * ld + col0 + gchunk * 8, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)piece * 1024u));
    }
}

// ---- dQ: workgroup = 64 * NG queries (4 waves x NG groups of 16; group u of wave w = rows q0 + (4u + w) * 16 ..), lane = (query c16, key
// group g); loops over 64-key tiles.  NG = 2 (round 4): a K / V fragment read from LDS feeds the MFMAs of BOTH query groups -- with one group
// per wave a tile costs a wave 24 LDS fragment reads for 24 MFMAs and the kernel was bound by the LDS pipe, not by the matrix pipe.  A
// group whose 16 rows lie beyond Lq is skipped (wave-uniform); VS_ATTN_BWD_NG=1 restores one group per wave. ----
template <bool BF16, int NG>
__global__ void __launch_bounds__(256, 2)
attn_bwd_dq_kernel(const AttnBwdArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned short smem[2][2][TB * HD];   // [ring slot][K | V]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    int b, h, bx_;
    block_bhx(a, b, h, bx_);
    const int q0 = bx_ * (64 * NG);
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)&smem[0][0][0];
    const KeyList kl = key_list(a, b);
    bool qvalid[NG];
    long long qrow[NG];
    int my_len[NG], nact = 0;
    uint4 qf[NG][2], dof[NG][2];
    float L[NG], D[NG];
    f4 dq[NG][4];
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        const int qi = q0 + (u * 4 + wid) * 16 + c16;
        qvalid[u] = qi < a.Lq;
        if (q0 + (u * 4 + wid) * 16 < a.Lq) nact = u + 1;          // (wave-uniform: groups 0 .. nact - 1 hold at least one real query)
        qrow[u] = b * a.q_batch_rows + (qvalid[u] ? qi : a.Lq - 1);
        my_len[u] = qvalid[u] ? kl.Lk : 0;
        if (a.q_kvlen && qvalid[u]) my_len[u] = min(kl.Lk, a.q_kvlen[(long long)b * a.Lq + qi]);
        // B-operand fragments (rows = queries): Q and dO
        const unsigned short *qp = a.q + qrow[u] * a.ldq + h * HD + g * 8, *dop = a.dout + qrow[u] * a.lddo + h * HD + g * 8;
        qf[u][0] = *reinterpret_cast<const uint4 *>(qp); qf[u][1] = *reinterpret_cast<const uint4 *>(qp + 32);
        dof[u][0] = *reinterpret_cast<const uint4 *>(dop); dof[u][1] = *reinterpret_cast<const uint4 *>(dop + 32);
        if (!qvalid[u]) dof[u][0] = dof[u][1] = make_uint4(0, 0, 0, 0);
        L[u] = qvalid[u] ? a.lse[qrow[u] * a.H + h] : INFINITY;
        D[u] = qvalid[u] ? a.delta[qrow[u] * a.H + h] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) dq[u][i] = f4{0.f, 0.f, 0.f, 0.f};
    }
    auto issue = [&](int kt, int slot) {
        auto rk = [&](int r) { return kl.row(kt + r); };   // (clamped to the last key)
        dma_tile(a.k, a.ldk, h * HD, rk, lds0 + (unsigned)(slot * 2) * (TB * HD * 2), tid);
        dma_tile(a.v, a.ldv, h * HD, rk, lds0 + (unsigned)(slot * 2 + 1) * (TB * HD * 2), tid);
    };
    if (kl.Lk > 0) issue(0, 0);
    for (int kt = 0, it = 0; kt < kl.Lk; kt += TB, ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile kt have landed ...
        __syncthreads();                                     // ... and everybody's; all waves are done with the other slot
        if (kt + TB < kl.Lk) issue(kt + TB, (it + 1) & 1);
        if (nact == 0) continue;                             // (no real query in this wave: it only stages)
        const unsigned short *sK = smem[it & 1][0], *sV = smem[it & 1][1];
        bool short_row = false;
#pragma unroll
        for (int u = 0; u < NG; ++u) short_row = short_row || (u < nact && my_len[u] < kt + TB);
        const bool tile_full = __builtin_amdgcn_ballot_w64(short_row) == 0ull;
        uint4 dsf[NG][2];
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {                  // key blocks 2 ks2, 2 ks2 + 1 -> the ks2-th k-step of dQ += dS K
            f4 ds[NG][2];
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2) {
                const int nb = 2 * ks2 + n2;
                f4 s[NG], dp[NG];
#pragma unroll
                for (int u = 0; u < NG; ++u) s[u] = dp[u] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const uint4 kf = *reinterpret_cast<const uint4 *>(&sK[sw_off(nb * 16 + c16, ks * 4 + g)]);
                    const uint4 vf = *reinterpret_cast<const uint4 *>(&sV[sw_off(nb * 16 + c16, ks * 4 + g)]);
#pragma unroll
                    for (int u = 0; u < NG; ++u)
                        if (u < nact) {
                            s[u] = mfma<BF16>(kf, qf[u][ks], s[u]);      // S^T[key][query]
                            dp[u] = mfma<BF16>(vf, dof[u][ks], dp[u]);   // dP^T[key][query]
                        }
                }
#pragma unroll
                for (int u = 0; u < NG; ++u) {
                    if (u >= nact) continue;
                    if (tile_full) {   // every key of the tile is visible to every query of the wave: no mask arithmetic
#pragma unroll
                        for (int r = 0; r < 4; ++r) ds[u][n2][r] = __builtin_amdgcn_exp2f(fmaf(s[u][r], a.scale_log2e, -L[u])) * (dp[u][r] - D[u]) * a.scale;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int key = kt + nb * 16 + g * 4 + r;
                            const float p = key < my_len[u] ? __builtin_amdgcn_exp2f(fmaf(s[u][r], a.scale_log2e, -L[u])) : 0.f;
                            ds[u][n2][r] = p * (dp[u][r] - D[u]) * a.scale;
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                if (u >= nact) continue;
                dsf[u][ks2].x = pack2v<BF16>(ds[u][0][0], ds[u][0][1]);
                dsf[u][ks2].y = pack2v<BF16>(ds[u][0][2], ds[u][0][3]);
                dsf[u][ks2].z = pack2v<BF16>(ds[u][1][0], ds[u][1][1]);
                dsf[u][ks2].w = pack2v<BF16>(ds[u][1][2], ds[u][1][3]);
            }
        }
        // dQ += dS K : A = dS (rows = query, k = keys in the permuted order of dsf), B = K^T rows d
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint2 lo = tr_rows4_sw(sK, (2 * ks) * 16 + g * 4, db * 16, c16);        // K[keys g*4..+3 of block 2ks][d = db*16 + c16]
                const uint2 hi = tr_rows4_sw(sK, (2 * ks + 1) * 16 + g * 4, db * 16, c16);
                const uint4 kt4 = make_uint4(lo.x, lo.y, hi.x, hi.y);
#pragma unroll
                for (int u = 0; u < NG; ++u)
                    if (u < nact) dq[u][db] = mfma<BF16>(kt4, dsf[u][ks], dq[u][db]);   // dQ^T += K^T dS^T: lane = query, registers = 4 consecutive d
            }
    }
    // dQ^T: query = this lane's c16, d = db*16 + g*4 + r; 16-byte stores (v_permlane16_swap: an even-g lane takes 8 consecutive d of
    // block db, an odd-g lane 8 of block db + 1), as the forward's store_o_rows
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        if (u >= nact) continue;
        typedef unsigned u2v_ __attribute__((ext_vector_type(2)));
        uint2 pk[4];
#pragma unroll
        for (int db = 0; db < 4; ++db) { pk[db].x = pack2<BF16>(dq[u][db][0], dq[u][db][1]); pk[db].y = pack2<BF16>(dq[u][db][2], dq[u][db][3]); }
        unsigned short *op = a.dq + qrow[u] * a.lddq + h * HD + (g & ~1) * 4;
        const bool odd = g & 1;
#pragma unroll
        for (int d2 = 0; d2 < 4; d2 += 2) {
            const u2v_ sx = __builtin_amdgcn_permlane16_swap(pk[d2].x, pk[d2 + 1].x, false, false);
            const u2v_ sy = __builtin_amdgcn_permlane16_swap(pk[d2].y, pk[d2 + 1].y, false, false);
            if (qvalid[u]) *reinterpret_cast<uint4 *>(op + (d2 + (odd ? 1 : 0)) * 16) = make_uint4(sx.x, sy.x, sx.y, sy.y);
        }
    }
}

// ---- dK, dV: workgroup = 64 * NG keys of the batch item's key list (4 waves x NG groups of 16, interleaved as in the dQ kernel), lane =
// (key c16, query group g); loops over 64-query tiles: a Q / dO fragment read feeds the MFMAs of all the wave's key groups ----
template <bool BF16, int NG>
__global__ void __launch_bounds__(256, 2)
attn_bwd_dkv_kernel(const AttnBwdArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned short smem[2][2][TB * HD];   // [ring slot][Q | dO]
    __shared__ __attribute__((aligned(16))) float sL2[2][TB], sD2[2][TB];
    __shared__ __attribute__((aligned(16))) int sLen2[2][TB];
    __shared__ int s_minlen2[2];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    int b, h, bx_;
    block_bhx(a, b, h, bx_);
    const int kt0 = bx_ * (64 * NG);
    const KeyList kl = key_list(a, b);
    if (kt0 >= kl.Lk) return;
    int kj[NG], kg0[NG], nact = 0;              // position of this lane's key in the key list; first position of the group
    bool kvalid[NG];
    uint4 kf[NG][2], vf[NG][2];
    f4 dk[NG][4], dv[NG][4];
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        kg0[u] = kt0 + (u * 4 + wid) * 16;
        kj[u] = kg0[u] + c16;
        kvalid[u] = kj[u] < kl.Lk;
        if (kg0[u] < kl.Lk) nact = u + 1;
        const long long krow = kl.row(kj[u]);
        // B-operand fragments (rows = keys): K and V
        const unsigned short *kp = a.k + krow * a.ldk + h * HD + g * 8, *vp = a.v + krow * a.ldv + h * HD + g * 8;
        kf[u][0] = *reinterpret_cast<const uint4 *>(kp); kf[u][1] = *reinterpret_cast<const uint4 *>(kp + 32);
        vf[u][0] = *reinterpret_cast<const uint4 *>(vp); vf[u][1] = *reinterpret_cast<const uint4 *>(vp + 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) dk[u][i] = dv[u][i] = f4{0.f, 0.f, 0.f, 0.f};
    }
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)&smem[0][0][0];
    auto issue = [&](int qt, int slot) {
        auto rq = [&](int r) { return b * a.q_batch_rows + min(qt + r, a.Lq - 1); };
        dma_tile(a.q, a.ldq, h * HD, rq, lds0 + (unsigned)(slot * 2) * (TB * HD * 2), tid);
        dma_tile(a.dout, a.lddo, h * HD, rq, lds0 + (unsigned)(slot * 2 + 1) * (TB * HD * 2), tid);
    };
    // per-query scalars of a tile (wave 0): logsumexp (+inf for missing queries: p = 0), delta, key limit, and the tile's smallest limit
    float aL = 0.f, aD = 0.f; int aN = 0;
    auto aux_load = [&](int qt) {
        const int qi = qt + tid;
        const bool ok = qi < a.Lq;
        const long long row = b * a.q_batch_rows + (ok ? qi : a.Lq - 1);
        aL = ok ? a.lse[row * a.H + h] : INFINITY;
        aD = ok ? a.delta[row * a.H + h] : 0.f;
        aN = !ok ? 0 : (a.q_kvlen ? min(kl.Lk, a.q_kvlen[(long long)b * a.Lq + qi]) : kl.Lk);
    };
    auto aux_store = [&](int slot) {   // (tid < 64 is exactly wave 0)
        sL2[slot][tid] = aL; sD2[slot][tid] = aD; sLen2[slot][tid] = aN;
        int mn = aN;
#pragma unroll
        for (int o_ = 32; o_ > 0; o_ >>= 1) mn = min(mn, __shfl_xor(mn, o_, 64));
        if (tid == 0) s_minlen2[slot] = mn;
    };
    if (a.Lq > 0) {
        issue(0, 0);
        if (tid < TB) { aux_load(0); aux_store(0); }
    }
    for (int qt = 0, it = 0; qt < a.Lq; qt += TB, ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile qt have landed ...
        __syncthreads();                                     // ... and everybody's (and wave 0's scalars); the other slot is free
        const bool more = qt + TB < a.Lq;
        if (more) {
            issue(qt + TB, (it + 1) & 1);
            if (tid < TB) aux_load(qt + TB);
        }
        const int slot = it & 1;
        const unsigned short *sQ = smem[slot][0], *sDO = smem[slot][1];
        const float *sL = sL2[slot], *sD = sD2[slot];
        const int *sLen = sLen2[slot];
        const int s_minlen = s_minlen2[slot];
        if (nact > 0) {
            uint4 pf[NG][2], dsf[NG][2];
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {              // query blocks 2 ks2, 2 ks2 + 1 -> the ks2-th k-step of dV / dK
                f4 p[NG][2], ds[NG][2];
#pragma unroll
                for (int n2 = 0; n2 < 2; ++n2) {
                    const int qb = 2 * ks2 + n2;
                    f4 s[NG], dp[NG];
#pragma unroll
                    for (int u = 0; u < NG; ++u) s[u] = dp[u] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const uint4 qa = *reinterpret_cast<const uint4 *>(&sQ[sw_off(qb * 16 + c16, ks * 4 + g)]);
                        const uint4 da = *reinterpret_cast<const uint4 *>(&sDO[sw_off(qb * 16 + c16, ks * 4 + g)]);
#pragma unroll
                        for (int u = 0; u < NG; ++u)
                            if (u < nact) {
                                s[u] = mfma<BF16>(qa, kf[u][ks], s[u]);     // S[query][key]: lane & 15 = key, registers = queries g*4 + r
                                dp[u] = mfma<BF16>(da, vf[u][ks], dp[u]);   // dP[query][key]
                            }
                    }
                    const float4 L4 = *reinterpret_cast<const float4 *>(&sL[qb * 16 + g * 4]), D4 = *reinterpret_cast<const float4 *>(&sD[qb * 16 + g * 4]);
                    const float Lr[4] = {L4.x, L4.y, L4.z, L4.w}, Dr[4] = {D4.x, D4.y, D4.z, D4.w};
                    const int4 N4 = *reinterpret_cast<const int4 *>(&sLen[qb * 16 + g * 4]);
                    const int Nr[4] = {N4.x, N4.y, N4.z, N4.w};
#pragma unroll
                    for (int u = 0; u < NG; ++u) {
                        if (u >= nact) continue;
                        // all 16 keys of the group are real and visible to all 64 queries of the tile: no mask arithmetic
                        const bool tile_full = kg0[u] + 16 <= min(kl.Lk, s_minlen);
                        if (tile_full) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float pv = __builtin_amdgcn_exp2f(fmaf(s[u][r], a.scale_log2e, -Lr[r]));
                                p[u][n2][r] = pv;
                                ds[u][n2][r] = pv * (dp[u][r] - Dr[r]) * a.scale;
                            }
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float pv = (kvalid[u] && kj[u] < Nr[r]) ? __builtin_amdgcn_exp2f(fmaf(s[u][r], a.scale_log2e, -Lr[r])) : 0.f;
                                p[u][n2][r] = pv;
                                ds[u][n2][r] = pv * (dp[u][r] - Dr[r]) * a.scale;
                            }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < NG; ++u) {
                    if (u >= nact) continue;
                    pf[u][ks2].x = pack2v<BF16>(p[u][0][0], p[u][0][1]);   pf[u][ks2].y = pack2v<BF16>(p[u][0][2], p[u][0][3]);
                    pf[u][ks2].z = pack2v<BF16>(p[u][1][0], p[u][1][1]);   pf[u][ks2].w = pack2v<BF16>(p[u][1][2], p[u][1][3]);
                    dsf[u][ks2].x = pack2v<BF16>(ds[u][0][0], ds[u][0][1]); dsf[u][ks2].y = pack2v<BF16>(ds[u][0][2], ds[u][0][3]);
                    dsf[u][ks2].z = pack2v<BF16>(ds[u][1][0], ds[u][1][1]); dsf[u][ks2].w = pack2v<BF16>(ds[u][1][2], ds[u][1][3]);
                }
            }
            // dV += P^T dO, dK += dS^T Q : A = P^T / dS^T (rows = key, k = queries in the packed order), B = dO^T / Q^T rows d
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const uint2 dlo = tr_rows4_sw(sDO, (2 * ks) * 16 + g * 4, db * 16, c16), dhi = tr_rows4_sw(sDO, (2 * ks + 1) * 16 + g * 4, db * 16, c16);
                    const uint2 qlo = tr_rows4_sw(sQ, (2 * ks) * 16 + g * 4, db * 16, c16), qhi = tr_rows4_sw(sQ, (2 * ks + 1) * 16 + g * 4, db * 16, c16);
                    const uint4 dO4 = make_uint4(dlo.x, dlo.y, dhi.x, dhi.y), q4 = make_uint4(qlo.x, qlo.y, qhi.x, qhi.y);
#pragma unroll
                    for (int u = 0; u < NG; ++u)
                        if (u < nact) {
                            dv[u][db] = mfma<BF16>(pf[u][ks], dO4, dv[u][db]);
                            dk[u][db] = mfma<BF16>(dsf[u][ks], q4, dk[u][db]);
                        }
                }
        }
        if (more && tid < TB) aux_store((it + 1) & 1);   // (read after the next barrier; last read two barriers ago)
    }
    // dK / dV rows key = g*4 + r (of the group's 16), cols d = db*16 + c16
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        if (u >= nact) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kpos = kg0[u] + g * 4 + r;
            if (kpos >= kl.Lk) continue;
            const long long row = kl.row(kpos);
            if (a.dk16) {
                unsigned short *pk = a.dk16 + row * a.lddk + h * HD + c16, *pv = a.dv16 + row * a.lddv + h * HD + c16;
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    pk[db * 16] = (unsigned short)(pack2<BF16>(dk[u][db][r], 0.f) & 0xffffu);
                    pv[db * 16] = (unsigned short)(pack2<BF16>(dv[u][db][r], 0.f) & 0xffffu);
                }
                continue;
            }
            float *pk = a.dk + row * a.lddk + h * HD + c16, *pv = a.dv + row * a.lddv + h * HD + c16;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                unsafeAtomicAdd(pk + db * 16, dk[u][db][r]);
                unsafeAtomicAdd(pv + db * 16, dv[u][db][r]);
            }
        }
    }
}

// ================= split operand class (dtype 4): the reference-precision backward =================
// Every tensor is f32 in HBM; the MFMA operands are (hi, lo) f16 pairs, hi = rne16(x), lo = rne16(x - hi), and every product is three
// v_mfma_f32_16x16x32_f16 (lo hi + hi lo + hi hi, f32 accumulate; gemm_common.h kDtSplit).  Q, K, V and dO arrive as separate hi / lo
// 16-bit images (vs_split16: one elementwise pass), so the tiles keep the 16-bit layout of the kernels above -- LDS-DMA staging,
// swizzled rows, transpose reads -- twice; P and dS are split in registers.  Gradients may be small: the caller keeps them in the f16
// range with a power-of-two loss scale (exact in f32), lo then degrades gracefully into f16 subnormals (absolute floor 2^-25).
typedef _Float16 half2b_ __attribute__((ext_vector_type(2)));
typedef float float2b_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2(float x, float y, unsigned &h, unsigned &l) {   // compiler-visible: the results feed MFMAs
    const half2b_ hh = __builtin_convertvector(float2b_{x, y}, half2b_);
    const half2b_ ll = __builtin_convertvector(float2b_{x - (float)hh.x, y - (float)hh.y}, half2b_);
    h = __builtin_bit_cast(unsigned, hh);
    l = __builtin_bit_cast(unsigned, ll);
}
__device__ __forceinline__ f4 mma3(const uint4 &ah, const uint4 &al, const uint4 &bh, const uint4 &bl, f4 c) {
    c = mfma<false>(al, bh, c);
    c = mfma<false>(ah, bl, c);
    return mfma<false>(ah, bh, c);
}
// f32 [rows 4-float groups] -> four split pairs of one fragment register quad
__device__ __forceinline__ void split_frag(const f4 &a, const f4 &b, uint4 &h, uint4 &l, int) {
    split2(a[0], a[1], h.x, l.x); split2(a[2], a[3], h.y, l.y); split2(b[0], b[1], h.z, l.z); split2(b[2], b[3], h.w, l.w);
}

__global__ void __launch_bounds__(256)
attn_delta_f32_kernel(const AttnBwdArgs a, long long rows) {
    const long long item = ((long long)blockIdx.x * 256 + threadIdx.x) >> 3;  // (row, head)
    const int sub = threadIdx.x & 7;
    float v = 0.f;
    const bool live = item < rows * a.H;
    if (live) {
        const long long row = item / a.H;
        const int h = (int)(item % a.H);
        const float *op = a.o32 + row * a.ldo32 + h * HD + sub * 8, *dp = a.dout32 + row * a.lddo32 + h * HD + sub * 8;
        const float4 o0 = *reinterpret_cast<const float4 *>(op), o1 = *reinterpret_cast<const float4 *>(op + 4);
        const float4 d0 = *reinterpret_cast<const float4 *>(dp), d1 = *reinterpret_cast<const float4 *>(dp + 4);
        v = o0.x * d0.x + o0.y * d0.y + o0.z * d0.z + o0.w * d0.w + o1.x * d1.x + o1.y * d1.y + o1.z * d1.z + o1.w * d1.w;
    }
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    if (live && sub == 0) a.delta[item] = v;
}

// NG row groups of 16 per wave (round 4), as the 16-bit kernels above: one K / V (Q / dO) fragment read from LDS feeds the 3-MFMA products of
// all the wave's groups.
template <int NG>
__global__ void __launch_bounds__(256, 2)
attn_bwd_dq_split_kernel(const AttnBwdArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned short smem[2][4][TB * HD];   // [ring slot][K hi | K lo | V hi | V lo]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    int b, h, bx_;
    block_bhx(a, b, h, bx_);
    const int q0 = bx_ * (64 * NG);
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)&smem[0][0][0];
    const KeyList kl = key_list(a, b);
    bool qvalid[NG];
    long long qrow[NG];
    int my_len[NG], nact = 0;
    uint4 qh[NG][2], ql[NG][2], doh[NG][2], dol[NG][2];
    float L[NG], D[NG];
    f4 dq[NG][4];
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        const int qi = q0 + (u * 4 + wid) * 16 + c16;
        qvalid[u] = qi < a.Lq;
        if (q0 + (u * 4 + wid) * 16 < a.Lq) nact = u + 1;
        qrow[u] = b * a.q_batch_rows + (qvalid[u] ? qi : a.Lq - 1);
        my_len[u] = qvalid[u] ? kl.Lk : 0;
        if (a.q_kvlen && qvalid[u]) my_len[u] = min(kl.Lk, a.q_kvlen[(long long)b * a.Lq + qi]);
        const long long qo = qrow[u] * a.ldq + h * HD + g * 8, dO = qrow[u] * a.lddo + h * HD + g * 8;
        qh[u][0] = *reinterpret_cast<const uint4 *>(a.q + qo); qh[u][1] = *reinterpret_cast<const uint4 *>(a.q + qo + 32);
        ql[u][0] = *reinterpret_cast<const uint4 *>(a.q_lo + qo); ql[u][1] = *reinterpret_cast<const uint4 *>(a.q_lo + qo + 32);
        doh[u][0] = *reinterpret_cast<const uint4 *>(a.dout + dO); doh[u][1] = *reinterpret_cast<const uint4 *>(a.dout + dO + 32);
        dol[u][0] = *reinterpret_cast<const uint4 *>(a.dout_lo + dO); dol[u][1] = *reinterpret_cast<const uint4 *>(a.dout_lo + dO + 32);
        if (!qvalid[u]) doh[u][0] = doh[u][1] = dol[u][0] = dol[u][1] = make_uint4(0, 0, 0, 0);
        L[u] = qvalid[u] ? a.lse[qrow[u] * a.H + h] : INFINITY;
        D[u] = qvalid[u] ? a.delta[qrow[u] * a.H + h] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) dq[u][i] = f4{0.f, 0.f, 0.f, 0.f};
    }
    auto issue = [&](int kt, int slot) {
        auto rk = [&](int r) { return kl.row(kt + r); };
        const unsigned base = lds0 + (unsigned)(slot * 4) * (TB * HD * 2);
        dma_tile(a.k, a.ldk, h * HD, rk, base, tid);
        dma_tile(a.k_lo, a.ldk, h * HD, rk, base + (TB * HD * 2), tid);
        dma_tile(a.v, a.ldv, h * HD, rk, base + 2 * (TB * HD * 2), tid);
        dma_tile(a.v_lo, a.ldv, h * HD, rk, base + 3 * (TB * HD * 2), tid);
    };
    if (kl.Lk > 0) issue(0, 0);
    for (int kt = 0, it = 0; kt < kl.Lk; kt += TB, ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + TB < kl.Lk) issue(kt + TB, (it + 1) & 1);
        if (nact == 0) continue;                             // (no real query in this wave: it only stages)
        const unsigned short *sKh = smem[it & 1][0], *sKl = smem[it & 1][1], *sVh = smem[it & 1][2], *sVl = smem[it & 1][3];
        uint4 dsh[NG][2], dsl[NG][2];
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
            f4 ds[NG][2];
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2) {
                const int nb = 2 * ks2 + n2;
                f4 s[NG], dp[NG];
#pragma unroll
                for (int u = 0; u < NG; ++u) s[u] = dp[u] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int off = sw_off(nb * 16 + c16, ks * 4 + g);
                    const uint4 kfh = *reinterpret_cast<const uint4 *>(&sKh[off]), kfl = *reinterpret_cast<const uint4 *>(&sKl[off]);
                    const uint4 vfh = *reinterpret_cast<const uint4 *>(&sVh[off]), vfl = *reinterpret_cast<const uint4 *>(&sVl[off]);
#pragma unroll
                    for (int u = 0; u < NG; ++u)
                        if (u < nact) {
                            s[u] = mma3(kfh, kfl, qh[u][ks], ql[u][ks], s[u]);        // S^T[key][query]
                            dp[u] = mma3(vfh, vfl, doh[u][ks], dol[u][ks], dp[u]);    // dP^T[key][query]
                        }
                }
#pragma unroll
                for (int u = 0; u < NG; ++u) {
                    if (u >= nact) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kt + nb * 16 + g * 4 + r;
                        const float p = key < my_len[u] ? __builtin_amdgcn_exp2f(fmaf(s[u][r], a.scale_log2e, -L[u])) : 0.f;
                        ds[u][n2][r] = p * (dp[u][r] - D[u]) * a.scale;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < NG; ++u)
                if (u < nact) split_frag(ds[u][0], ds[u][1], dsh[u][ks2], dsl[u][ks2], 0);
        }
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint2 hlo = tr_rows4_sw(sKh, (2 * ks) * 16 + g * 4, db * 16, c16), hhi = tr_rows4_sw(sKh, (2 * ks + 1) * 16 + g * 4, db * 16, c16);
                const uint2 llo = tr_rows4_sw(sKl, (2 * ks) * 16 + g * 4, db * 16, c16), lhi = tr_rows4_sw(sKl, (2 * ks + 1) * 16 + g * 4, db * 16, c16);
                const uint4 kth = make_uint4(hlo.x, hlo.y, hhi.x, hhi.y), ktl = make_uint4(llo.x, llo.y, lhi.x, lhi.y);
#pragma unroll
                for (int u = 0; u < NG; ++u)
                    if (u < nact) dq[u][db] = mma3(kth, ktl, dsh[u][ks], dsl[u][ks], dq[u][db]);
            }
    }
    // dQ^T: query = this lane's c16, d = db*16 + g*4 + r: one 16-byte f32 store per block
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        if (u >= nact || !qvalid[u]) continue;
        float *op = a.dq32 + qrow[u] * a.lddq32 + h * HD + g * 4;
#pragma unroll
        for (int db = 0; db < 4; ++db) *reinterpret_cast<float4 *>(op + db * 16) = make_float4(dq[u][db][0], dq[u][db][1], dq[u][db][2], dq[u][db][3]);
    }
}

template <int NG>
__global__ void __launch_bounds__(256, 2)
attn_bwd_dkv_split_kernel(const AttnBwdArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned short smem[2][4][TB * HD];   // [ring slot][Q hi | Q lo | dO hi | dO lo]
    __shared__ __attribute__((aligned(16))) float sL2[2][TB], sD2[2][TB];
    __shared__ __attribute__((aligned(16))) int sLen2[2][TB];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    int b, h, bx_;
    block_bhx(a, b, h, bx_);
    const int kt0 = bx_ * (64 * NG);
    const KeyList kl = key_list(a, b);
    if (kt0 >= kl.Lk) return;
    int kj[NG], kg0[NG], nact = 0;
    bool kvalid[NG];
    uint4 kh[NG][2], klo[NG][2], vh[NG][2], vl[NG][2];
    f4 dk[NG][4], dv[NG][4];
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        kg0[u] = kt0 + (u * 4 + wid) * 16;
        kj[u] = kg0[u] + c16;
        kvalid[u] = kj[u] < kl.Lk;
        if (kg0[u] < kl.Lk) nact = u + 1;
        const long long krow = kl.row(kj[u]);
        const long long ko = krow * a.ldk + h * HD + g * 8, vo = krow * a.ldv + h * HD + g * 8;
        kh[u][0] = *reinterpret_cast<const uint4 *>(a.k + ko); kh[u][1] = *reinterpret_cast<const uint4 *>(a.k + ko + 32);
        klo[u][0] = *reinterpret_cast<const uint4 *>(a.k_lo + ko); klo[u][1] = *reinterpret_cast<const uint4 *>(a.k_lo + ko + 32);
        vh[u][0] = *reinterpret_cast<const uint4 *>(a.v + vo); vh[u][1] = *reinterpret_cast<const uint4 *>(a.v + vo + 32);
        vl[u][0] = *reinterpret_cast<const uint4 *>(a.v_lo + vo); vl[u][1] = *reinterpret_cast<const uint4 *>(a.v_lo + vo + 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) dk[u][i] = dv[u][i] = f4{0.f, 0.f, 0.f, 0.f};
    }
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)&smem[0][0][0];
    auto issue = [&](int qt, int slot) {
        auto rq = [&](int r) { return b * a.q_batch_rows + min(qt + r, a.Lq - 1); };
        const unsigned base = lds0 + (unsigned)(slot * 4) * (TB * HD * 2);
        dma_tile(a.q, a.ldq, h * HD, rq, base, tid);
        dma_tile(a.q_lo, a.ldq, h * HD, rq, base + (TB * HD * 2), tid);
        dma_tile(a.dout, a.lddo, h * HD, rq, base + 2 * (TB * HD * 2), tid);
        dma_tile(a.dout_lo, a.lddo, h * HD, rq, base + 3 * (TB * HD * 2), tid);
    };
    float aL = 0.f, aD = 0.f; int aN = 0;
    auto aux_load = [&](int qt) {
        const int qi = qt + tid;
        const bool ok = qi < a.Lq;
        const long long row = b * a.q_batch_rows + (ok ? qi : a.Lq - 1);
        aL = ok ? a.lse[row * a.H + h] : INFINITY;
        aD = ok ? a.delta[row * a.H + h] : 0.f;
        aN = !ok ? 0 : (a.q_kvlen ? min(kl.Lk, a.q_kvlen[(long long)b * a.Lq + qi]) : kl.Lk);
    };
    auto aux_store = [&](int slot) { sL2[slot][tid] = aL; sD2[slot][tid] = aD; sLen2[slot][tid] = aN; };
    if (a.Lq > 0) {
        issue(0, 0);
        if (tid < TB) { aux_load(0); aux_store(0); }
    }
    for (int qt = 0, it = 0; qt < a.Lq; qt += TB, ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const bool more = qt + TB < a.Lq;
        if (more) {
            issue(qt + TB, (it + 1) & 1);
            if (tid < TB) aux_load(qt + TB);
        }
        const int slot = it & 1;
        const unsigned short *sQh = smem[slot][0], *sQl = smem[slot][1], *sDh = smem[slot][2], *sDl = smem[slot][3];
        const float *sL = sL2[slot], *sD = sD2[slot];
        const int *sLen = sLen2[slot];
        if (nact > 0) {
            uint4 pfh[NG][2], pfl[NG][2], dsh[NG][2], dsl[NG][2];
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                f4 p[NG][2], ds[NG][2];
#pragma unroll
                for (int n2 = 0; n2 < 2; ++n2) {
                    const int qb = 2 * ks2 + n2;
                    f4 s[NG], dp[NG];
#pragma unroll
                    for (int u = 0; u < NG; ++u) s[u] = dp[u] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const int off = sw_off(qb * 16 + c16, ks * 4 + g);
                        const uint4 qah = *reinterpret_cast<const uint4 *>(&sQh[off]), qal = *reinterpret_cast<const uint4 *>(&sQl[off]);
                        const uint4 dah = *reinterpret_cast<const uint4 *>(&sDh[off]), dal = *reinterpret_cast<const uint4 *>(&sDl[off]);
#pragma unroll
                        for (int u = 0; u < NG; ++u)
                            if (u < nact) {
                                s[u] = mma3(qah, qal, kh[u][ks], klo[u][ks], s[u]);     // S[query][key]: lane & 15 = key, registers = queries g*4 + r
                                dp[u] = mma3(dah, dal, vh[u][ks], vl[u][ks], dp[u]);    // dP[query][key]
                            }
                    }
                    const float4 L4 = *reinterpret_cast<const float4 *>(&sL[qb * 16 + g * 4]), D4 = *reinterpret_cast<const float4 *>(&sD[qb * 16 + g * 4]);
                    const int4 N4 = *reinterpret_cast<const int4 *>(&sLen[qb * 16 + g * 4]);
                    const float Lr[4] = {L4.x, L4.y, L4.z, L4.w}, Dr[4] = {D4.x, D4.y, D4.z, D4.w};
                    const int Nr[4] = {N4.x, N4.y, N4.z, N4.w};
#pragma unroll
                    for (int u = 0; u < NG; ++u) {
                        if (u >= nact) continue;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float pv = (kvalid[u] && kj[u] < Nr[r]) ? __builtin_amdgcn_exp2f(fmaf(s[u][r], a.scale_log2e, -Lr[r])) : 0.f;
                            p[u][n2][r] = pv;
                            ds[u][n2][r] = pv * (dp[u][r] - Dr[r]) * a.scale;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < NG; ++u) {
                    if (u >= nact) continue;
                    split_frag(p[u][0], p[u][1], pfh[u][ks2], pfl[u][ks2], 0);
                    split_frag(ds[u][0], ds[u][1], dsh[u][ks2], dsl[u][ks2], 0);
                }
            }
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int r0 = (2 * ks) * 16 + g * 4, r1 = (2 * ks + 1) * 16 + g * 4;
                    const uint2 dh0 = tr_rows4_sw(sDh, r0, db * 16, c16), dh1 = tr_rows4_sw(sDh, r1, db * 16, c16);
                    const uint2 dl0 = tr_rows4_sw(sDl, r0, db * 16, c16), dl1 = tr_rows4_sw(sDl, r1, db * 16, c16);
                    const uint2 qh0 = tr_rows4_sw(sQh, r0, db * 16, c16), qh1 = tr_rows4_sw(sQh, r1, db * 16, c16);
                    const uint2 ql0 = tr_rows4_sw(sQl, r0, db * 16, c16), ql1 = tr_rows4_sw(sQl, r1, db * 16, c16);
                    const uint4 d4h = make_uint4(dh0.x, dh0.y, dh1.x, dh1.y), d4l = make_uint4(dl0.x, dl0.y, dl1.x, dl1.y);
                    const uint4 q4h = make_uint4(qh0.x, qh0.y, qh1.x, qh1.y), q4l = make_uint4(ql0.x, ql0.y, ql1.x, ql1.y);
#pragma unroll
                    for (int u = 0; u < NG; ++u)
                        if (u < nact) {
                            dv[u][db] = mma3(pfh[u][ks], pfl[u][ks], d4h, d4l, dv[u][db]);
                            dk[u][db] = mma3(dsh[u][ks], dsl[u][ks], q4h, q4l, dk[u][db]);
                        }
                }
        }
        if (more && tid < TB) aux_store((it + 1) & 1);
    }
    // dK / dV rows key = g*4 + r (of the group's 16), cols d = db*16 + c16
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        if (u >= nact) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kpos = kg0[u] + g * 4 + r;
            if (kpos >= kl.Lk) continue;
            const long long row = kl.row(kpos);
            float *pk = a.dk + row * a.lddk + h * HD + c16, *pv = a.dv + row * a.lddv + h * HD + c16;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                if (a.kv_direct) { pk[db * 16] = dk[u][db][r]; pv[db * 16] = dv[u][db][r]; }
                else { unsafeAtomicAdd(pk + db * 16, dk[u][db][r]); unsafeAtomicAdd(pv + db * 16, dv[u][db][r]); }
            }
        }
    }
}

}  // namespace

namespace {
int attention_backward_impl(const void *q, const void *k, const void *v, const void *o, const void *dout, const float *lse,
                            float *delta, void *dq, float *dk, float *dv, void *dk16, void *dv16, int32_t nbatch, int32_t H, int32_t Lq, int32_t Lk,
                            int64_t q_batch_rows, int64_t k_batch_rows, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                            int32_t lddo, int32_t lddq, int32_t lddk, int32_t lddv, const int32_t *kv_seg,
                            const int32_t *q_kvlen, int32_t max_keys, float scale, int32_t dtype, vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(q && k && v && o && dout && lse && delta && dq && ((dk && dv) || (dk16 && dv16)), "vs_attention_backward: null pointer");
    VS_CHECK(!(dk16 && kv_seg), "vs_attention_backward16: key segments share K/V rows between batch items: use the f32 (atomic) entry");
    VS_CHECK(nbatch >= 0 && H > 0 && Lq >= 0, "vs_attention_backward: bad sizes");
    VS_CHECK(dtype == 1 || dtype == 2, "vs_attention_backward: dtype must be 1 (f16) or 2 (bf16)");
    VS_CHECK(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && lddo % 8 == 0 && ldo % 8 == 0, "vs_attention_backward: row strides must be multiples of 8 elements");
    VS_CHECK((((uintptr_t)o | (uintptr_t)dout) & 15) == 0, "vs_attention_backward: o and dout must be 16-byte aligned");
    VS_CHECK(kv_seg ? max_keys > 0 : Lk > 0, "vs_attention_backward: Lk (or max_keys with kv_seg) must be positive");
    if (nbatch == 0 || Lq == 0) return 0;
    AttnBwdArgs a;
    { static const int x_ = [] { const char *e = getenv("VS_ATTN_BWD_XCD"); return e ? atoi(e) : 1; }(); a.xcd = x_; }
    a.q = (const unsigned short *)q; a.k = (const unsigned short *)k; a.v = (const unsigned short *)v;
    a.o = (const unsigned short *)o; a.dout = (const unsigned short *)dout; a.lse = lse; a.delta = delta;
    a.dq = (unsigned short *)dq; a.dk = dk; a.dv = dv; a.dk16 = (unsigned short *)dk16; a.dv16 = (unsigned short *)dv16; a.kv_seg = kv_seg; a.q_kvlen = q_kvlen;
    a.nbatch = nbatch; a.H = H; a.Lq = Lq; a.Lk = Lk; a.q_batch_rows = q_batch_rows; a.k_batch_rows = k_batch_rows;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.lddo = lddo; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
    a.scale = scale; a.scale_log2e = scale * 1.4426950408889634f;
    const long long rows = (long long)(nbatch - 1) * q_batch_rows + Lq;
    const int keys = kv_seg ? max_keys : Lk;
    dim3 block(256);
    static const int ng = [] { const char *e = getenv("VS_ATTN_BWD_NG"); return e && atoi(e) == 1 ? 1 : 2; }();   // row groups of 16 per wave
    const dim3 gq(vs::cdiv(Lq, 64 * ng), H, nbatch), gk(vs::cdiv(keys, 64 * ng), H, nbatch);
    if (dtype == 2) {
        hipLaunchKernelGGL(attn_delta_kernel<true>, dim3((unsigned)vs::cdiv64(rows * H, 32)), block, 0, stream, a, rows);
        if (ng == 1) {
            hipLaunchKernelGGL((attn_bwd_dq_kernel<true, 1>), gq, block, 0, stream, a);
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<true, 1>), gk, block, 0, stream, a);
        } else {
            hipLaunchKernelGGL((attn_bwd_dq_kernel<true, 2>), gq, block, 0, stream, a);
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<true, 2>), gk, block, 0, stream, a);
        }
    } else {
        hipLaunchKernelGGL(attn_delta_kernel<false>, dim3((unsigned)vs::cdiv64(rows * H, 32)), block, 0, stream, a, rows);
        if (ng == 1) {
            hipLaunchKernelGGL((attn_bwd_dq_kernel<false, 1>), gq, block, 0, stream, a);
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<false, 1>), gk, block, 0, stream, a);
        } else {
            hipLaunchKernelGGL((attn_bwd_dq_kernel<false, 2>), gq, block, 0, stream, a);
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<false, 2>), gk, block, 0, stream, a);
        }
    }
    VS_HIP(hipGetLastError());
    return 0;
}
}  // namespace

extern "C" int vs_attention_backward(const void *q, const void *k, const void *v, const void *o, const void *dout, const float *lse,
                                     float *delta, void *dq, float *dk, float *dv, int32_t nbatch, int32_t H, int32_t Lq, int32_t Lk,
                                     int64_t q_batch_rows, int64_t k_batch_rows, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                                     int32_t lddo, int32_t lddq, int32_t lddk, int32_t lddv, const int32_t *kv_seg,
                                     const int32_t *q_kvlen, int32_t max_keys, float scale, int32_t dtype, vs_stream_t stream_) {
    return attention_backward_impl(q, k, v, o, dout, lse, delta, dq, dk, dv, nullptr, nullptr, nbatch, H, Lq, Lk, q_batch_rows, k_batch_rows, ldq, ldk, ldv,
                                   ldo, lddo, lddq, lddk, lddv, kv_seg, q_kvlen, max_keys, scale, dtype, stream_);
}

extern "C" int vs_attention_backward16(const void *q, const void *k, const void *v, const void *o, const void *dout, const float *lse,
                                       float *delta, void *dq, void *dk, void *dv, int32_t nbatch, int32_t H, int32_t Lq, int32_t Lk,
                                       int64_t q_batch_rows, int64_t k_batch_rows, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                                       int32_t lddo, int32_t lddq, int32_t lddk, int32_t lddv, const int32_t *q_kvlen, float scale,
                                       int32_t dtype, vs_stream_t stream_) {
    return attention_backward_impl(q, k, v, o, dout, lse, delta, dq, nullptr, nullptr, dk, dv, nbatch, H, Lq, Lk, q_batch_rows, k_batch_rows, ldq, ldk,
                                   ldv, ldo, lddo, lddq, lddk, lddv, nullptr, q_kvlen, 0, scale, dtype, stream_);
}

// Backward of vs_attention(_lse) in the split operand class (dtype 4): every tensor f32.  q_hi / q_lo ... are the 16-bit (hi, lo) images of
// q, k, v, dout written by vs_split16 (row strides ldq / ldk / ldv / lddo in 16-bit elements, shared by hi and lo); o, dout the f32
// tensors (delta); dq f32 [rows, lddq] written; dk, dv f32 indexed by key row: written (kv_seg null: every K/V row has one owner) or
// ADDED to with f32 atomics (kv_seg: zero them first).  Three f16 MFMAs per product, f32 accumulate: the reference-precision backward
// of F.scaled_dot_product_attention (croco/blocks.py:106-110).
extern "C" int vs_attention_backward_split(const void *q_hi, const void *q_lo, const void *k_hi, const void *k_lo, const void *v_hi,
                                           const void *v_lo, const void *do_hi, const void *do_lo, const float *o, const float *dout,
                                           const float *lse, float *delta, float *dq, float *dk, float *dv, int32_t nbatch, int32_t H,
                                           int32_t Lq, int32_t Lk, int64_t q_batch_rows, int64_t k_batch_rows, int32_t ldq, int32_t ldk,
                                           int32_t ldv, int32_t lddo, int32_t ldo32, int32_t lddo32, int32_t lddq, int32_t lddk, int32_t lddv,
                                           const int32_t *kv_seg, const int32_t *q_kvlen, int32_t max_keys, float scale, vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(q_hi && q_lo && k_hi && k_lo && v_hi && v_lo && do_hi && do_lo && o && dout && lse && delta && dq && dk && dv,
             "vs_attention_backward_split: null pointer");
    VS_CHECK(nbatch >= 0 && H > 0 && Lq >= 0, "vs_attention_backward_split: bad sizes");
    VS_CHECK(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && lddo % 8 == 0, "vs_attention_backward_split: 16-bit row strides must be multiples of 8");
    VS_CHECK(ldo32 % 4 == 0 && lddo32 % 4 == 0 && lddq % 4 == 0, "vs_attention_backward_split: f32 row strides must be multiples of 4");
    VS_CHECK((((uintptr_t)q_hi | (uintptr_t)q_lo | (uintptr_t)k_hi | (uintptr_t)k_lo | (uintptr_t)v_hi | (uintptr_t)v_lo | (uintptr_t)do_hi |
               (uintptr_t)do_lo | (uintptr_t)o | (uintptr_t)dout | (uintptr_t)dq) & 15) == 0, "vs_attention_backward_split: 16-byte alignment required");
    VS_CHECK(kv_seg ? max_keys > 0 : Lk > 0, "vs_attention_backward_split: Lk (or max_keys with kv_seg) must be positive");
    if (nbatch == 0 || Lq == 0) return 0;
    AttnBwdArgs a;
    { static const int x_ = [] { const char *e = getenv("VS_ATTN_BWD_XCD"); return e ? atoi(e) : 1; }(); a.xcd = x_; }
    a.q = (const unsigned short *)q_hi; a.k = (const unsigned short *)k_hi; a.v = (const unsigned short *)v_hi; a.dout = (const unsigned short *)do_hi;
    a.q_lo = (const unsigned short *)q_lo; a.k_lo = (const unsigned short *)k_lo; a.v_lo = (const unsigned short *)v_lo; a.dout_lo = (const unsigned short *)do_lo;
    a.o = nullptr; a.o32 = o; a.dout32 = dout; a.lse = lse; a.delta = delta;
    a.dq = nullptr; a.dq32 = dq; a.dk = dk; a.dv = dv; a.dk16 = nullptr; a.dv16 = nullptr; a.kv_seg = kv_seg; a.q_kvlen = q_kvlen;
    a.nbatch = nbatch; a.H = H; a.Lq = Lq; a.Lk = Lk; a.q_batch_rows = q_batch_rows; a.k_batch_rows = k_batch_rows;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = 0; a.lddo = lddo; a.lddq = 0; a.lddk = lddk; a.lddv = lddv;
    a.ldo32 = ldo32; a.lddo32 = lddo32; a.lddq32 = lddq; a.kv_direct = kv_seg ? 0 : 1;
    a.scale = scale; a.scale_log2e = scale * 1.4426950408889634f;
    const long long rows = (long long)(nbatch - 1) * q_batch_rows + Lq;
    const int keys = kv_seg ? max_keys : Lk;
    dim3 block(256);
    hipLaunchKernelGGL(attn_delta_f32_kernel, dim3((unsigned)vs::cdiv64(rows * H, 32)), block, 0, stream, a, rows);
    static const int ngq = [] { const char *e = getenv("VS_ATTN_BWD_NG"); return e && atoi(e) == 1 ? 1 : 2; }();       // row groups of 16 per wave
    // (dK / dV: two key groups need 64 + 64 + 64 registers for the K | V fragments, the accumulators and the (hi, lo) P | dS fragments alone:
    // the NG = 2 instantiation spills 102 VGPRs under the 256-register cap -- one group per wave unless VS_ATTN_BWD_NG_KV=2)
    static const int ngk = [] { const char *e = getenv("VS_ATTN_BWD_NG_KV"); return e && atoi(e) == 2 ? 2 : 1; }();
    if (ngq == 1) hipLaunchKernelGGL(attn_bwd_dq_split_kernel<1>, dim3(vs::cdiv(Lq, 64), H, nbatch), block, 0, stream, a);
    else hipLaunchKernelGGL(attn_bwd_dq_split_kernel<2>, dim3(vs::cdiv(Lq, 128), H, nbatch), block, 0, stream, a);
    if (ngk == 1) hipLaunchKernelGGL(attn_bwd_dkv_split_kernel<1>, dim3(vs::cdiv(keys, 64), H, nbatch), block, 0, stream, a);
    else hipLaunchKernelGGL(attn_bwd_dkv_split_kernel<2>, dim3(vs::cdiv(keys, 128), H, nbatch), block, 0, stream, a);
    VS_HIP(hipGetLastError());
    return 0;
}
