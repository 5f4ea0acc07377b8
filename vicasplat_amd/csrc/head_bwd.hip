// Backward of the LAST 1x1 convolution of a DPT head in the split operand class, fused with the ReLU backward of its input: one pass over
// the full-resolution tensors instead of seven.
//
// Forward (heads/dpt_block.py:316-343, dpt_gs_head.py:120-157): t = relu(conv3x3(...)) [pixels, Cin] -> y = t W^T + b [pixels, Cout]
// (Cout = 3 | 4 for pts3d, 8 + 3 d_sh = 83 for the Gaussian parameters; Cin = 128 | 256).  Backward, for dy [pixels, Cout]:
//     dt[p, c] = (t[p, c] > 0) * sum_n dy[p, n] W[n, c]        dW[n, c] = sum_p dy[p, n] t[p, c]        db[n] = sum_p dy[p, n]
// The operator-by-operator route (autograd.LinearSplitFn + Conv3x3Fn's relu_mask) pads dy to 32 columns, runs the dx GEMM, transposes dy
// and t into packed images for the weight-gradient GEMM and masks dt in a further pass: 10 GB of algorithmic traffic become 35 GB, all on
// 4.2 M-row tensors (8.2 + 2.1 ms of a 366 ms split-class training step for the Gaussian head alone).  Here a persistent workgroup streams
// 32-pixel tiles: dy and t are read ONCE (f32), converted to (hi, lo) f16 pairs into LDS, and both products run from that image --
//     dt^T = W^T dy^T   A = W^T fragments (registers, loaded once per workgroup), B = dy rows (ds_read_b128)          -> float4 stores along c
//     dW  += dy^T t     A = dy^T, B = t: both reduction-major, gathered with the LDS transpose read ds_read_b64_tr_b16 (gemm256.h)
// every product three f16 MFMAs (lo hi + hi lo + hi hi, f32 accumulate) as in gemm_common.h mma2<kDtSplit>.  The ReLU mask is a nibble per
// four channels written beside the converted tile (exact: taken from the f32 value); the column sums ride on two extra MFMAs against a
// fragment of ones.  dW / db leave as per-workgroup partials summed by the caller (deterministic, no atomics).
// HBM-bound: (Cout + 2 Cin) * 4 bytes per pixel.  The 16-bit classes (f16 / bf16 tensors, one MFMA per product) run the same kernel
// without the conversion: (Cout + 2 Cin) * 2 bytes per pixel.
#include "common.h"
#include "gemm_common.h"

namespace {

struct HeadBwdArgs {
    const void *dy, *t;    // f32 (split class) or 16-bit
    const float *w;
    void *dt;
    float *dw_part, *db_part;
    long long ntiles;      // 32-pixel tiles
    int cout, ldy, relu;   // ldy: row stride of dy in elements (>= cout; rows padded for alignment are skipped, not read as data)
    float w_scale, inv_scale;
};

__device__ __forceinline__ void split4(const float4 x, uint2 &h, uint2 &l) {   // results go to LDS (inline asm: see gemm_common.h split8_lds)
    h.x = cvt_pk_f16(x.x, x.y); h.y = cvt_pk_f16(x.z, x.w);
    float r0, r1, r2, r3;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h.x), "v"(x.x));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h.x), "v"(x.y));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(h.y), "v"(x.z));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(h.y), "v"(x.w));
    l.x = cvt_pk_f16(r0, r1); l.y = cvt_pk_f16(r2, r3);
}

// positive (ReLU passes its gradient) for a 16-bit float of either format: sign clear and not zero
__device__ __forceinline__ unsigned pos16(unsigned h) { return ((h & 0x8000u) == 0u && (h & 0x7fffu) != 0u) ? 1u : 0u; }

// DT: kDtSplit (f32 tensors, three f16 MFMAs per product), 0 (f16) or 1 (bf16: 16-bit tensors, one MFMA per product)
// CIN: input channels (128 | 256); KS: 32-wide k-steps covering Cout in the dt product; MT: 16-row tiles covering Cout in the dW product
template <int DT, int CIN, int KS, int MT>
__global__ void __launch_bounds__(512) head1x1_bwd_kernel(const HeadBwdArgs a) {
    constexpr bool SPLIT = DT == kDtSplit;
    constexpr int R = 32;                       // pixels per tile
    constexpr int CW = CIN / 8, CT = CW / 16;   // channels per wave, 16-channel tiles per wave
    constexpr int TP = CIN * 2 + 32;            // byte pitch of a t row (hi or lo image): the four rows of a transpose read in four bank windows
    constexpr int DP = KS * 64 + 16;            // byte pitch of a dy row
    constexpr int NT = SPLIT ? CIN / 64 : CIN / 128;      // 16-byte pieces of t per thread and tile
    constexpr int ND = SPLIT ? 2 : 1;                     // 16-byte pieces of dy per thread and tile (ldy <= 128 f32 | 16-bit elements)
    constexpr int EPC = SPLIT ? 4 : 8;                    // elements per 16-byte piece
    __shared__ __attribute__((aligned(16))) unsigned char sTH[R * TP], sTL[SPLIT ? R * TP : 16], sDH[R * DP], sDL[SPLIT ? R * DP : 16];
    __shared__ __attribute__((aligned(4))) unsigned char sMask[R * (CIN / 4)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, g = lane >> 4;
    const int wbase = wid * CW;
    const int cout = a.cout, ldy = a.ldy;

    // ---- W^T fragments (A operand of dt^T = W^T dy^T): row c = wbase + ct*16 + l16, k = n = ks*32 + g*8 + 0..7 (split: scaled by 2^e) ----
    uint4 wh[CT][KS];
    [[maybe_unused]] uint4 wl[CT][KS];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int c = wbase + ct * 16 + l16;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = ks * 32 + g * 8 + j;
                v[j] = n < cout ? a.w[(long long)n * CIN + c] * a.w_scale : 0.f;
            }
            if constexpr (SPLIT) {
                uint4 f0 = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
                uint4 f1 = make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7]));
                split8(f0, f1);
                wh[ct][ks] = f0; wl[ct][ks] = f1;
            } else {
                wh[ct][ks] = make_uint4(pack16x2<DT>(v[0], v[1]), pack16x2<DT>(v[2], v[3]), pack16x2<DT>(v[4], v[5]), pack16x2<DT>(v[6], v[7]));
            }
        }
    // the dy images' columns >= Cout stay zero for the whole kernel (the conversion only writes n < Cout)
    for (int i = tid; i < R * DP / 4; i += 512) {
        reinterpret_cast<unsigned *>(sDH)[i] = 0u;
        if constexpr (SPLIT) reinterpret_cast<unsigned *>(sDL)[i] = 0u;
    }
    f4 dw[MT][CT], dbacc = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) dw[mt][ct] = f4{0.f, 0.f, 0.f, 0.f};
    const unsigned one2 = DT == 1 ? 0x3F803F80u : 0x3C003C00u;
    const uint4 ones = make_uint4(one2, one2, one2, one2);
    __syncthreads();

    typedef short tr4 __attribute__((ext_vector_type(4)));
    typedef tr4 __attribute__((address_space(3))) *trp_t;
    auto tr8 = [&](const unsigned char *p, int pitch) -> uint4 {   // reduction rows 8g .. 8g+3 and 8g+4 .. 8g+7 of one column
        const tr4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(const_cast<unsigned char *>(p)));
        const tr4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(const_cast<unsigned char *>(p + 4 * pitch)));
        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        return make_uint4(l2.x, l2.y, h2.x, h2.y);
    };
    const int trow = 8 * g + (l16 >> 2), tcol = (l16 & 3) * 4;      // this lane's address inside a [4 rows][16 columns] block

    const int ndy = R * ldy / EPC;               // 16-byte pieces of a dy tile (R * ldy elements, contiguous, 16-byte aligned)
    uint4 pt[NT], pd[ND];
    auto ldnt = [](const void *p) -> uint4 {     // read-once streams: nontemporal
        const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p));
        return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
    };
    auto prefetch = [&](long long tile) {
        const uint4 *tp = reinterpret_cast<const uint4 *>(a.t) + tile * (R * CIN / EPC);
#pragma unroll
        for (int j = 0; j < NT; ++j) pt[j] = ldnt(tp + tid + 512 * j);
        const uint4 *dp = reinterpret_cast<const uint4 *>(a.dy) + tile * ndy;
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const int q = tid + 512 * j;
            pd[j] = q < ndy ? ldnt(dp + q) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    long long tile = blockIdx.x;
    if (tile < a.ntiles) prefetch(tile);
    for (; tile < a.ntiles; tile += gridDim.x) {
        // ---- the prefetched tile -> LDS images (split: converted to (hi, lo) halves) + the ReLU mask, a nibble per four channels ----
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int f = tid + 512 * j;
            if constexpr (SPLIT) {
                const int p = f / (CIN / 4), c4 = f % (CIN / 4);
                const float4 x = make_float4(__uint_as_float(pt[j].x), __uint_as_float(pt[j].y), __uint_as_float(pt[j].z), __uint_as_float(pt[j].w));
                uint2 h, l;
                split4(x, h, l);
                *reinterpret_cast<uint2 *>(sTH + p * TP + c4 * 8) = h;
                *reinterpret_cast<uint2 *>(sTL + p * TP + c4 * 8) = l;
                sMask[p * (CIN / 4) + c4] = (unsigned char)((x.x > 0.f ? 1 : 0) | (x.y > 0.f ? 2 : 0) | (x.z > 0.f ? 4 : 0) | (x.w > 0.f ? 8 : 0));
            } else {
                const int p = f / (CIN / 8), c8 = f % (CIN / 8);
                *reinterpret_cast<uint4 *>(sTH + p * TP + c8 * 16) = pt[j];
                const unsigned m0 = pos16(pt[j].x & 0xffffu) | (pos16(pt[j].x >> 16) << 1) | (pos16(pt[j].y & 0xffffu) << 2) | (pos16(pt[j].y >> 16) << 3);
                const unsigned m1 = pos16(pt[j].z & 0xffffu) | (pos16(pt[j].z >> 16) << 1) | (pos16(pt[j].w & 0xffffu) << 2) | (pos16(pt[j].w >> 16) << 3);
                *reinterpret_cast<unsigned short *>(sMask + p * (CIN / 4) + c8 * 2) = (unsigned short)(m0 | (m1 << 8));
            }
        }
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const int q = tid + 512 * j;
            if (q < ndy) {
                int e = EPC * q, p = e / ldy, n = e - p * ldy;
                unsigned hh[EPC];
                [[maybe_unused]] unsigned ll[EPC];
                if constexpr (SPLIT) {
                    const float4 x = make_float4(__uint_as_float(pd[j].x), __uint_as_float(pd[j].y), __uint_as_float(pd[j].z), __uint_as_float(pd[j].w));
                    uint2 h, l;
                    split4(x, h, l);
                    hh[0] = h.x & 0xffffu; hh[1] = h.x >> 16; hh[2] = h.y & 0xffffu; hh[3] = h.y >> 16;
                    ll[0] = l.x & 0xffffu; ll[1] = l.x >> 16; ll[2] = l.y & 0xffffu; ll[3] = l.y >> 16;
                } else {
                    hh[0] = pd[j].x & 0xffffu; hh[1] = pd[j].x >> 16; hh[2] = pd[j].y & 0xffffu; hh[3] = pd[j].y >> 16;
                    hh[4] = pd[j].z & 0xffffu; hh[5] = pd[j].z >> 16; hh[6] = pd[j].w & 0xffffu; hh[7] = pd[j].w >> 16;
                }
#pragma unroll
                for (int i = 0; i < EPC; ++i) {
                    if (n < cout) {
                        *reinterpret_cast<unsigned short *>(sDH + p * DP + n * 2) = (unsigned short)hh[i];
                        if constexpr (SPLIT) *reinterpret_cast<unsigned short *>(sDL + p * DP + n * 2) = (unsigned short)ll[i];
                    }
                    if (++n == ldy) { n = 0; ++p; }
                }
            }
        }
        __syncthreads();
        const long long next = tile + gridDim.x;
        if (next < a.ntiles) prefetch(next);

        // ---- dt^T = W^T dy^T: per 16-pixel tile the dy fragments are read once for the wave's CT channel tiles ----
#pragma unroll
        for (int ptile = 0; ptile < R / 16; ++ptile) {
            f4 acc[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[ct] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int off = (ptile * 16 + l16) * DP + ks * 64 + g * 16;
                const uint4 bh = *reinterpret_cast<const uint4 *>(sDH + off);
                if constexpr (SPLIT) {
                    const uint4 bl = *reinterpret_cast<const uint4 *>(sDL + off);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        acc[ct] = mfma<0>(wl[ct][ks], bh, acc[ct]);
                        acc[ct] = mfma<0>(wh[ct][ks], bl, acc[ct]);
                        acc[ct] = mfma<0>(wh[ct][ks], bh, acc[ct]);
                    }
                } else {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) acc[ct] = mfma<DT>(wh[ct][ks], bh, acc[ct]);
                }
            }
            const int p = ptile * 16 + l16;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = wbase + ct * 16 + 4 * g;
                const unsigned m = a.relu ? sMask[p * (CIN / 4) + (c >> 2)] : 15u;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = ((m >> r) & 1u) ? acc[ct][r] * a.inv_scale : 0.f;
                const long long idx = (tile * R + p) * (long long)CIN + c;
                if constexpr (SPLIT) *reinterpret_cast<float4 *>(reinterpret_cast<float *>(a.dt) + idx) = make_float4(o[0], o[1], o[2], o[3]);
                else *reinterpret_cast<uint2 *>(reinterpret_cast<unsigned short *>(a.dt) + idx) = make_uint2(pack16x2<DT>(o[0], o[1]), pack16x2<DT>(o[2], o[3]));
            }
        }
        // ---- dW += dy^T t (one 32-pixel k-step); db += dy^T 1 on the wave that owns the row tile ----
        uint4 th[CT];
        [[maybe_unused]] uint4 tl[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int off = trow * TP + (wbase + ct * 16 + tcol) * 2;
            th[ct] = tr8(sTH + off, TP);
            if constexpr (SPLIT) tl[ct] = tr8(sTL + off, TP);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int off = trow * DP + (mt * 16 + tcol) * 2;
            const uint4 ah = tr8(sDH + off, DP);
            if constexpr (SPLIT) {
                const uint4 al = tr8(sDL + off, DP);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    dw[mt][ct] = mfma<0>(al, th[ct], dw[mt][ct]);
                    dw[mt][ct] = mfma<0>(ah, tl[ct], dw[mt][ct]);
                    dw[mt][ct] = mfma<0>(ah, th[ct], dw[mt][ct]);
                }
            } else {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) dw[mt][ct] = mfma<DT>(ah, th[ct], dw[mt][ct]);
            }
        }
        if (wid < MT) {      // column sums of dy: row tile `wid` of dy^T against a fragment of ones
            const int off = trow * DP + (wid * 16 + tcol) * 2;
            const uint4 ah = tr8(sDH + off, DP);
            if constexpr (SPLIT) dbacc = mfma<0>(tr8(sDL + off, DP), ones, dbacc);
            dbacc = mfma<SPLIT ? 0 : DT>(ah, ones, dbacc);
        }
        __syncthreads();
    }
    // ---- per-workgroup partial sums: dw_part [wg][MT*16][CIN], db_part [wg][MT*16] ----
    float *dwp = a.dw_part + (long long)blockIdx.x * (MT * 16) * CIN;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) dwp[(long long)(mt * 16 + 4 * g + r) * CIN + wbase + ct * 16 + l16] = dw[mt][ct][r];
    if (wid < MT && l16 == 0) {
        float *dbp = a.db_part + (long long)blockIdx.x * (MT * 16) + wid * 16 + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) dbp[r] = dbacc[r];
    }
}

template <int DT>
int launch_head_bwd(const HeadBwdArgs &a, int Cin, bool wide, int nwg, hipStream_t s) {
    dim3 grid((unsigned)nwg), block(512);
    if (Cin == 256 && wide) hipLaunchKernelGGL((head1x1_bwd_kernel<DT, 256, 3, 6>), grid, block, 0, s, a);
    else if (Cin == 256) hipLaunchKernelGGL((head1x1_bwd_kernel<DT, 256, 1, 1>), grid, block, 0, s, a);
    else if (wide) hipLaunchKernelGGL((head1x1_bwd_kernel<DT, 128, 3, 6>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((head1x1_bwd_kernel<DT, 128, 1, 1>), grid, block, 0, s, a);
    VS_HIP(hipGetLastError());
    return 0;
}

int head_bwd_entry(const char *who, const void *dy, int64_t ldy, const void *t, const float *w, int w_scale_exp, void *dt, float *dw_part, float *db_part,
                   int64_t P, int Cin, int Cout, int relu, int nwg, int dtype, vs_stream_t stream) {
    VS_CHECK(dy && t && w && dt && dw_part && db_part, "%s: null pointer", who);
    VS_CHECK(P > 0 && P % 32 == 0, "%s: P=%lld must be a positive multiple of 32", who, (long long)P);
    VS_CHECK(Cin == 128 || Cin == 256, "%s: Cin=%d must be 128 or 256", who, Cin);
    VS_CHECK(Cout >= 1 && Cout <= 96, "%s: Cout=%d must be in 1..96", who, Cout);
    VS_CHECK(ldy >= Cout && ldy <= 128, "%s: ldy=%lld must be in Cout..128", who, (long long)ldy);
    VS_CHECK(nwg >= 1 && nwg <= 65535, "%s: nwg=%d out of range", who, nwg);
    VS_CHECK(w_scale_exp >= -60 && w_scale_exp <= 60, "%s: w_scale_exp=%d out of range", who, w_scale_exp);
    VS_CHECK((((uintptr_t)dy | (uintptr_t)t | (uintptr_t)dt) & 15) == 0, "%s: dy, t and dt must be 16-byte aligned", who);
    HeadBwdArgs a;
    a.dy = dy; a.t = t; a.w = w; a.dt = dt; a.dw_part = dw_part; a.db_part = db_part;
    a.ntiles = P / 32; a.cout = Cout; a.ldy = (int)ldy; a.relu = relu;
    a.w_scale = ldexpf(1.f, w_scale_exp); a.inv_scale = ldexpf(1.f, -w_scale_exp);
    const bool wide = Cout > 16;     // 3 k-steps / 6 row tiles (<= 96 outputs) or 1 / 1 (<= 16)
    hipStream_t s = (hipStream_t)stream;
    if (dtype == 4) return launch_head_bwd<kDtSplit>(a, Cin, wide, nwg, s);
    if (dtype == 1) return launch_head_bwd<0>(a, Cin, wide, nwg, s);
    return launch_head_bwd<1>(a, Cin, wide, nwg, s);
}

}  // namespace

extern "C" int vs_head1x1_backward_split(const float *dy, int64_t ldy, const float *t, const float *w, int32_t w_scale_exp, float *dt, float *dw_part,
                                         float *db_part, int64_t P, int32_t Cin, int32_t Cout, int32_t relu, int32_t nwg, vs_stream_t stream) {
    return head_bwd_entry("vs_head1x1_backward_split", dy, ldy, t, w, w_scale_exp, dt, dw_part, db_part, P, Cin, Cout, relu, nwg, 4, stream);
}

extern "C" int vs_head1x1_backward16(const void *dy, int64_t ldy, const void *t, const float *w, void *dt, float *dw_part, float *db_part, int64_t P,
                                     int32_t Cin, int32_t Cout, int32_t relu, int32_t nwg, int32_t dtype, vs_stream_t stream) {
    VS_CHECK(dtype == 1 || dtype == 2, "vs_head1x1_backward16: dtype must be 1 (f16) or 2 (bf16)");
    return head_bwd_entry("vs_head1x1_backward16", dy, ldy, t, w, 0, dt, dw_part, db_part, P, Cin, Cout, relu, nwg, dtype, stream);
}
