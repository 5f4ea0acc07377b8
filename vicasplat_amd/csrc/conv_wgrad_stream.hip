// Weight gradient of a 3x3 convolution (stride 1, pad 1) in the split operand class for the NARROW layers of the DPT heads (Cin, Cout
// multiples of 64 that are not both multiples of 256: the pts3d head's 256 -> 128 and 128 -> 128 convolutions, heads/dpt_block.py:316-343),
// as ONE streaming pass over X and dY as they lie in memory (NHWC f32):
//     dW[tap = (ky, kx)][ci][co] = sum over images n and pixels (y, x) of act(X)[n, y + ky - 1, x + kx - 1, ci] * dY[n, y, x, co]
// The tile route for these shapes (ops.conv3x3_backward_split, non-ATN branch) writes a zero-bordered transposed copy of X and a transposed
// packed copy of dY (two passes over 2.1 GB each at 64 x 256 x 256 x 128) and then runs nine 128 x 128 output tiles per K slice on the 4-wave
// GEMM kernel at 0.19 of the matrix peak: 11.0 + 4.5 ms of a 348 ms split-class training step for the two layers.
// Here a workgroup owns a 64 (ci) x 64 (co) block of all nine taps (72 accumulator registers per lane) and walks DOWN a 32-pixel-wide column
// strip of one image: per step it loads ONE new row of X (34 pixels with the halo, 64 channels) and one row of dY (32 pixels, 64 channels),
// converts them to (hi, lo) f16 images in LDS (X in a three-row ring), and every tap is a reduction over the 32 pixels with BOTH operands
// gathered reduction-major by the LDS transpose read (ds_read_b64_tr_b16, as gemm256.h / head_bwd.hip) -- the tap shift is a row offset of
// that read, the zero padding is zeros in the halo.  Every product is three f16 MFMAs (mma2<kDtSplit>'s order).  The bias gradient rides on
// two extra MFMAs against a fragment of ones in the ci-block-0 workgroups.  Persistent workgroups; dW / db leave as per-worker partials the
// caller sums (deterministic).  Traffic: X is read Cout / 64 times and dY Cin / 64 times (through L2: the blocks of one worker share an XCD).
#include "common.h"
#include "gemm_common.h"

namespace {

struct ConvWgradStreamArgs {
    const float *x, *dy;
    float *dw_part, *db_part;    // [workers][9][Cin][Cout], [workers][Cout]
    int N, H, W, Cin, Cout, relu, workers, nb_co;     // nb_co = Cout / 64 (blocks per worker = (Cin / 64) * nb_co)
};

__device__ __forceinline__ void split4s(const float4 x, uint2 &h, uint2 &l) {   // results go to LDS (inline asm: see gemm_common.h split8_lds)
    h.x = cvt_pk_f16(x.x, x.y); h.y = cvt_pk_f16(x.z, x.w);
    float r0, r1, r2, r3;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h.x), "v"(x.x));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h.x), "v"(x.y));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(h.y), "v"(x.z));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(h.y), "v"(x.w));
    l.x = cvt_pk_f16(r0, r1); l.y = cvt_pk_f16(r2, r3);
}

__global__ void __launch_bounds__(512) conv3x3_wgrad_stream_kernel(const ConvWgradStreamArgs a) {
    constexpr int PW = 32;                 // pixels per step (one row segment)
    constexpr int PIT = 64 * 2 + 16;       // byte pitch of one pixel's 64 channels (hi or lo image): the four rows of a transpose read in four bank windows
    constexpr int XROW = (PW + 2) * PIT;   // one ring row of X: 34 pixels (halo left and right)
    __shared__ __attribute__((aligned(16))) unsigned char sXH[3 * XROW], sXL[3 * XROW], sDH[PW * PIT], sDL[PW * PIT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, g = lane >> 4;
    // block id -> (worker, block): the blocks of one worker are 8 ids apart (same XCD: they read the same X / dY rows through one L2)
    const int nblk = (a.Cin >> 6) * a.nb_co;
    const int bid = blockIdx.x;
    const int wlo = bid & 7, rest = bid >> 3;
    const int blk = rest % nblk, worker = (rest / nblk) * 8 + wlo;
    if (worker >= a.workers) return;
    const int ci0 = (blk / a.nb_co) * 64, co0 = (blk % a.nb_co) * 64;
    const int cit = wid & 3, cot = (wid >> 2) * 2;          // this wave: ci tile cit, co tiles cot, cot + 1 (of the block's 4 x 4)
    const bool do_db = ci0 == 0 && cit == 0;

    f4 acc[9][2], dbacc[2];
#pragma unroll
    for (int t = 0; t < 9; ++t) { acc[t][0] = f4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = f4{0.f, 0.f, 0.f, 0.f}; }
    dbacc[0] = dbacc[1] = f4{0.f, 0.f, 0.f, 0.f};
    const uint4 ones = make_uint4(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u);

    typedef short tr4 __attribute__((ext_vector_type(4)));
    typedef tr4 __attribute__((address_space(3))) *trp_t;
    auto tr8 = [&](const unsigned char *p) -> uint4 {   // reduction rows (pixels) 8g .. 8g+3 and 8g+4 .. 8g+7 of one channel column
        const tr4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(const_cast<unsigned char *>(p)));
        const tr4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(const_cast<unsigned char *>(p + 4 * PIT)));
        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        return make_uint4(l2.x, l2.y, h2.x, h2.y);
    };
    const int trow = 8 * g + (l16 >> 2), tcol = (l16 & 3) * 4;      // this lane's address inside a [4 pixels][16 channels] block
    const int offA = trow * PIT + (cit * 16 + tcol) * 2;           // + kx * PIT + ring row
    const int offB = trow * PIT + (cot * 16 + tcol) * 2;           // second co tile: + 32

    // staging roles: piece = (pixel, 4 channels); X: pieces 0 .. 543 (34 pixels), dY: pieces 0 .. 511
    const int spx = tid >> 4, sc4 = tid & 15;
    const int strips = a.W / PW, nitems = a.N * strips;
    const long long rowX = (long long)a.W * a.Cin, rowD = (long long)a.W * a.Cout;
    auto ldnt = [](const float *p) -> float4 { return *reinterpret_cast<const float4 *>(p); };   // (plain loads: sibling blocks re-read these rows from L2)
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int item = worker; item < nitems; item += a.workers) {
        const int n = item / strips, x0 = (item - n * strips) * PW;
        const float *xb = a.x + ((long long)n * a.H * a.W) * a.Cin + ci0 + sc4 * 4;
        const float *db = a.dy + ((long long)n * a.H * a.W + x0 + spx) * a.Cout + co0 + sc4 * 4;
        // X pieces of this thread: pixel x0 - 1 + spx (all threads) and x0 + 31 + spx (threads of pixels 0, 1: ring pixels 32, 33)
        const int xa = x0 - 1 + spx, xe = x0 + 31 + spx;
        const bool oka = xa >= 0, oke = spx < 2 && xe < a.W;
        auto load_x = [&](int yy, float4 &va, float4 &ve) {
            va = oka ? ldnt(xb + yy * rowX + (long long)xa * a.Cin) : z4;
            ve = oke ? ldnt(xb + yy * rowX + (long long)xe * a.Cin) : z4;
        };
        auto store_x = [&](int slot, float4 va, float4 ve) {
            if (a.relu) {
                va.x = fmaxf(va.x, 0.f); va.y = fmaxf(va.y, 0.f); va.z = fmaxf(va.z, 0.f); va.w = fmaxf(va.w, 0.f);
                ve.x = fmaxf(ve.x, 0.f); ve.y = fmaxf(ve.y, 0.f); ve.z = fmaxf(ve.z, 0.f); ve.w = fmaxf(ve.w, 0.f);
            }
            uint2 h, l;
            split4s(va, h, l);
            *reinterpret_cast<uint2 *>(sXH + slot * XROW + spx * PIT + sc4 * 8) = h;
            *reinterpret_cast<uint2 *>(sXL + slot * XROW + spx * PIT + sc4 * 8) = l;
            if (spx < 2) {
                split4s(ve, h, l);
                *reinterpret_cast<uint2 *>(sXH + slot * XROW + (32 + spx) * PIT + sc4 * 8) = h;
                *reinterpret_cast<uint2 *>(sXL + slot * XROW + (32 + spx) * PIT + sc4 * 8) = l;
            }
        };
        // ---- prologue: ring slot 2 (row -1) = zeros, slot 0 = row 0; registers: row 1 of X, row 0 of dY ----
        float4 xa4, xe4, d4;
        load_x(0, xa4, xe4);
        __syncthreads();                       // the previous item's last step is done with the ring
        store_x(2, z4, z4);
        store_x(0, xa4, xe4);
        if (a.H > 1) load_x(1, xa4, xe4); else { xa4 = z4; xe4 = z4; }
        d4 = ldnt(db);
        for (int y = 0; y < a.H; ++y) {
            // ---- registers -> LDS: X row y + 1 (zeros below the image) into slot (y + 1) % 3, dY row y ----
            const int s_up = (y + 2) % 3, s_mid = y % 3, s_dn = (y + 1) % 3;     // rows y - 1, y, y + 1
            store_x(s_dn, xa4, xe4);
            {
                uint2 h, l;
                split4s(d4, h, l);
                *reinterpret_cast<uint2 *>(sDH + spx * PIT + sc4 * 8) = h;
                *reinterpret_cast<uint2 *>(sDL + spx * PIT + sc4 * 8) = l;
            }
            __syncthreads();
            // ---- prefetch: X row y + 2, dY row y + 1 ----
            if (y + 2 < a.H) load_x(y + 2, xa4, xe4); else { xa4 = z4; xe4 = z4; }
            if (y + 1 < a.H) d4 = ldnt(db + (y + 1) * rowD);
            // ---- nine taps over the 32 pixels ----
            const uint4 bh0 = tr8(sDH + offB), bl0 = tr8(sDL + offB), bh1 = tr8(sDH + offB + 32), bl1 = tr8(sDL + offB + 32);
            const int slots[3] = {s_up, s_mid, s_dn};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int rb = slots[ky] * XROW + offA;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const uint4 ah = tr8(sXH + rb + kx * PIT), al = tr8(sXL + rb + kx * PIT);
                    f4 &c0 = acc[ky * 3 + kx][0], &c1 = acc[ky * 3 + kx][1];
                    c0 = mfma<0>(al, bh0, c0); c0 = mfma<0>(ah, bl0, c0); c0 = mfma<0>(ah, bh0, c0);
                    c1 = mfma<0>(al, bh1, c1); c1 = mfma<0>(ah, bl1, c1); c1 = mfma<0>(ah, bh1, c1);
                }
            }
            if (do_db) {      // column sums of dY (bias gradient): ones^T dY for this wave's two co tiles
                dbacc[0] = mfma<0>(ones, bl0, dbacc[0]); dbacc[0] = mfma<0>(ones, bh0, dbacc[0]);
                dbacc[1] = mfma<0>(ones, bl1, dbacc[1]); dbacc[1] = mfma<0>(ones, bh1, dbacc[1]);
            }
            __syncthreads();
        }
    }
    // ---- per-worker partials: dw_part[worker][tap][ci][co], db_part[worker][co] ----
    float *dwp = a.dw_part + (long long)worker * 9 * a.Cin * a.Cout;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                dwp[((long long)t * a.Cin + ci0 + cit * 16 + 4 * g + r) * a.Cout + co0 + (cot + j) * 16 + l16] = acc[t][j][r];
    if (do_db && g == 0) {     // every row of ones^T dY holds the same sums: lanes of row group 0, register 0
        float *dbp = a.db_part + (long long)worker * a.Cout + co0 + cot * 16 + l16;
        dbp[0] = dbacc[0][0];
        dbp[16] = dbacc[1][0];
    }
}

}  // namespace

extern "C" int vs_conv3x3_wgrad_split_stream(const float *x, const float *dy, float *dw_part, float *db_part, int32_t N, int32_t H, int32_t W,
                                             int32_t Cin, int32_t Cout, int32_t relu_in, int32_t workers, vs_stream_t stream) {
    VS_CHECK(x && dy && dw_part && db_part, "vs_conv3x3_wgrad_split_stream: null pointer");
    VS_CHECK(N > 0 && H > 0 && W > 0 && W % 32 == 0, "vs_conv3x3_wgrad_split_stream: W=%d must be a positive multiple of 32", W);
    VS_CHECK(Cin > 0 && Cout > 0 && Cin % 64 == 0 && Cout % 64 == 0, "vs_conv3x3_wgrad_split_stream: Cin=%d and Cout=%d must be multiples of 64", Cin, Cout);
    VS_CHECK(workers >= 1 && workers % 8 == 0, "vs_conv3x3_wgrad_split_stream: workers=%d must be a positive multiple of 8", workers);
    VS_CHECK((((uintptr_t)x | (uintptr_t)dy) & 15) == 0, "vs_conv3x3_wgrad_split_stream: x and dy must be 16-byte aligned");
    ConvWgradStreamArgs a;
    a.x = x; a.dy = dy; a.dw_part = dw_part; a.db_part = db_part;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.relu = relu_in; a.workers = workers; a.nb_co = Cout / 64;
    const long long nwg = (long long)workers * (Cin / 64) * (Cout / 64);
    VS_CHECK(nwg <= 0x7fffffffLL, "vs_conv3x3_wgrad_split_stream: grid too large");
    hipLaunchKernelGGL(conv3x3_wgrad_stream_kernel, dim3((unsigned)nwg), dim3(512), 0, (hipStream_t)stream, a);
    VS_HIP(hipGetLastError());
    return 0;
}
