// The Gaussian-parameter head's 7x7 RGB stem fused with its "upsample + add" (dpt_gs_head.py:112-118,142-150), split operand class, as a
// STREAMING kernel:   out[pixel] = packed( bilinear_x2(trunk)[pixel] + relu(conv7x7(image)[pixel] + bias) )      (Cout = 256)
// The tile route (gemm.hip conv7x7_256_kernel<split, UPADD>) runs this on the 256 x 256 GEMM main loop: one kernel row (21 of 32 staged floats)
// per K-tile, K = 256 for 147 taps, one 8-wave workgroup per CU whose epilogue (four global tap gathers per output at L2 latency, 262 KB of
// stores) cannot overlap the next tile's main loop: 9.4 ms per 24-scene step for 16 GB of traffic and 2.8 PFLOP of MFMA work (0.12 of peak).
// Here a persistent workgroup walks DOWN a 32-pixel-wide column strip of one frame.  The image (3 channels) lives in LDS as a 16-row ring of
// (hi, lo) f16 images -- four copies shifted by 0..3 pixels, 37 KB in all, refreshed four rows at a time -- and the reduction index is ordered (channel, kx) x ky: the eight
// k-values of an MFMA lane are the SAME column of eight consecutive image rows, i.e. exactly what the LDS transpose read
// (ds_read_b64_tr_b16) delivers for 16 consecutive pixels; the kx shift picks the copy that keeps the read 8-byte aligned; ky = 7 is a
// zero weight.  K = 21 x 8 = 168 -> six 32-wide steps (147 useful of 192).  W fragments stay in registers for the whole kernel (a wave owns
// 32 output channels = one 32-column block of the packed row); per 32-pixel step a wave issues 72 MFMAs and 48 transpose reads, then adds
// the bilinear taps (the expression of upsample2x_f32_block_kernel / stem_upadd_pair) and writes 16-byte hi and lo chunks (store_split8).
// HBM-bound on the 12.9 GB of packed output + 3.2 GB of trunk per 24-scene step.
#include "common.h"
#include "gemm_common.h"

#include <cstdlib>

namespace {

struct StemStreamArgs {
    const float *img;      // zero-bordered NHWC frames [N, Hp, Wp, 3] (ops.pad_rgb_nhwc: 3 pixels before, >= 3 after)
    const float *w;        // [256, 3, 7, 7] f32 (the module's parameter)
    const float *bias;     // [256]
    const float *trunk;    // [N, H/2, W/2, 256] f32
    float *out;            // packed rows [N*H*W][256 x 4 bytes]
    int N, H, W, Hp, Wp;
    float w_scale, inv_scale;
};

constexpr int kIP = 96;                 // byte pitch of one ring row of one (copy, channel) image: 48 halves
constexpr int kIS = 16 * kIP;           // one (copy, channel) image: 16 ring slots (two batches of 8 rows)
constexpr int kImgBytes = 2 * 4 * 3 * kIS;   // (hi | lo) x 4 shifted copies x 3 channels

template <int PT>     // 16-pixel tiles per step: the strip is 16 * PT pixels wide
__global__ void __launch_bounds__(512) stem_up_stream_kernel(const StemStreamArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char sImg[kImgBytes];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, g = lane >> 4;
    const int chb = wid * 32;                 // this wave's 32 output channels (one block of the packed row)

    // ---- W fragments: row = channel chb + ct*16 + l16; k-block g of step ks = combo q = ks*4 + g = (c, kx), its 8 k = ky 0..7 (ky 7: zero) ----
    uint4 wh[2][6], wl[2][6];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            const int q = ks * 4 + g, c = q / 7, kx = q - c * 7;
            const float *wp = a.w + (long long)(chb + ct * 16 + l16) * 147 + c * 49 + kx;
            float v[8];
#pragma unroll
            for (int ky = 0; ky < 8; ++ky) v[ky] = (q < 21 && ky < 7) ? wp[ky * 7] * a.w_scale : 0.f;
            uint4 f0 = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
            uint4 f1 = make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7]));
            split8(f0, f1);
            wh[ct][ks] = f0; wl[ct][ks] = f1;
        }
    float bv[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const float4 t = *reinterpret_cast<const float4 *>(a.bias + chb + ct * 16 + 4 * g);
        bv[ct][0] = t.x; bv[ct][1] = t.y; bv[ct][2] = t.z; bv[ct][3] = t.w;
    }
    // ---- per-lane image offsets of the transpose reads: combo (c, kx) of step ks -> copy s = kx & 3, column block (kx & ~3) + (l16 & 3) * 4 ----
    int ioff[6];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
        int q = ks * 4 + g;
        if (q >= 21) q = 0;                   // zero weights: any finite data
        const int c = q / 7, kx = q - c * 7;
        ioff[ks] = ((kx & 3) * 3 + c) * kIS + ((kx & ~3) + (l16 & 3) * 4) * 2;
    }
    const int trq = l16 >> 2;                 // ring row inside a 4-row transpose block
    typedef short tr4 __attribute__((ext_vector_type(4)));
    typedef tr4 __attribute__((address_space(3))) *trp_t;
    auto trd = [&](const unsigned char *p) -> uint2 {
        return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(const_cast<unsigned char *>(p))));
    };

    // staging role: thread < 132 owns (pixel column px = tid / 3 of the strip's 44, channel c = tid % 3) of every image row
    const int spx = tid / 3, sc = tid - spx * 3;
    const bool stager = tid < 132;
    auto store_px = [&](int slot, float v) {          // one pixel value -> (hi, lo) halves in the four shifted copies
        const unsigned h = cvt_pk_f16(v, 0.f) & 0xffffu;
        const _Float16 hf = __builtin_bit_cast(_Float16, (unsigned short)h);
        const unsigned l = cvt_pk_f16(v - (float)hf, 0.f) & 0xffffu;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int j = spx - s;
            if (j >= 0 && j < 48) {
                *reinterpret_cast<unsigned short *>(sImg + (s * 3 + sc) * kIS + slot * kIP + j * 2) = (unsigned short)h;
                *reinterpret_cast<unsigned short *>(sImg + kImgBytes / 2 + (s * 3 + sc) * kIS + slot * kIP + j * 2) = (unsigned short)l;
            }
        }
    };
    for (int i = tid; i < kImgBytes / 4; i += 512) reinterpret_cast<unsigned *>(sImg)[i] = 0u;      // (columns 44 .. 47 of every row stay zero)

    const int Hs = a.H >> 1, Ws = a.W >> 1, C = 256;
    const float ry = (float)(Hs - 1) / (float)(a.H - 1), rx = (float)(Ws - 1) / (float)(a.W - 1);
    const int strips = a.W / (16 * PT), nitems = a.N * strips;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int n = item / strips, x0s = (item - n * strips) * (16 * PT);
        const float *ib = a.img + ((long long)n * a.Hp * a.Wp + x0s) * 3 + tid;        // row r: + r * Wp * 3
        const bool colok = stager && x0s + spx < a.Wp;
        auto load_row = [&](int r) -> float { return (colok && r < a.Hp) ? ib[(long long)r * a.Wp * 3] : 0.f; };
        // ---- prologue: padded rows 0 .. 11 into their slots, rows 12 .. 15 into registers ----
        __syncthreads();                       // the previous item is done with the ring
        float nx[4];
        if (stager) {
#pragma unroll 1
            for (int r0_ = 0; r0_ < 12; r0_ += 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) nx[r] = load_row(r0_ + r);
#pragma unroll
                for (int r = 0; r < 4; ++r) store_px(r0_ + r, nx[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) nx[r] = load_row(12 + r);
        }
        // the x taps of this lane's two pixels (ptile 0 / 1) do not change down the strip
        int tx0[PT], tx1[PT];
        float tlx[PT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const float sx = (float)(x0s + pt * 16 + l16) * rx;
            tx0[pt] = min((int)sx, Ws - 1);
            tlx[pt] = sx - (float)tx0[pt];
            tx1[pt] = min(tx0[pt] + 1, Ws - 1);
        }
        float4 tpA[2][4], tpB[2][4];
        auto load_taps = [&](float4 (&t)[2][4], const float *q0, const float *q1, int pt) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                t[ct][0] = *reinterpret_cast<const float4 *>(q0 + (long long)tx0[pt] * C + ct * 16);
                t[ct][1] = *reinterpret_cast<const float4 *>(q0 + (long long)tx1[pt] * C + ct * 16);
                t[ct][2] = *reinterpret_cast<const float4 *>(q1 + (long long)tx0[pt] * C + ct * 16);
                t[ct][3] = *reinterpret_cast<const float4 *>(q1 + (long long)tx1[pt] * C + ct * 16);
            }
        };
        const float *r0 = a.trunk + ((long long)(n * Hs) * Ws) * C + chb + 4 * g;      // output row 0: source rows 0 and min(1, Hs - 1), ly = 0
        const float *r1 = a.trunk + ((long long)(n * Hs + min(1, Hs - 1)) * Ws) * C + chb + 4 * g;
        float ly_c = 0.f;
        bool same_prev = false;                 // this output row's two source rows = the previous row's (every other row at scale 2)
        load_taps(tpA, r0, r1, 0);
        for (int yo = 0; yo < a.H; ++yo) {
            // the ring is refreshed four rows at a time: steps 4b .. 4b + 3 read padded rows 4b .. 4b + 11; the rows stored here (4b + 8 .. 4b + 11) take
            // the slots of rows last read two batches ago, so ONE barrier per four steps orders everything (one per step: the kernel ran at the
            // latency of its slowest wave every 32 pixels, 6.7 ms per 24-scene step)
            if ((yo & 3) == 0) {
                if (yo > 0 && stager) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) store_px((yo + 8 + r) & 15, nx[r]);
                }
                __syncthreads();
                if (stager) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) nx[r] = load_row(yo + 12 + r);
                }
            }
            const float ly = ly_c;
            const int so0 = ((yo + trq) & 15) * kIP, so1 = ((yo + 4 + trq) & 15) * kIP;
            // the tap rows of the NEXT output row (its first tile's taps are requested under this row's last tile)
            const float sy_n = (float)(yo + 1) * ry;
            const int y0_n = min((int)sy_n, Hs - 1), y1_n = min(y0_n + 1, Hs - 1);
            const float *r0_n = a.trunk + ((long long)(n * Hs + y0_n) * Ws) * C + chb + 4 * g, *r1_n = a.trunk + ((long long)(n * Hs + y1_n) * Ws) * C + chb + 4 * g;
            const bool same_next = r0_n == r0 && r1_n == r1;
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                // ---- software pipeline: the bilinear taps of the NEXT tile are requested before this tile's MFMAs (8 x 16 bytes per lane in flight
                // for a whole tile time; requested and consumed inside one tile they cost 2.1 of 6.6 ms, and every wait for them also waited for
                // the previous tile's stores: vmcnt counts in order) ----
                float4 (&tc)[2][4] = ((PT * 0 + pt) & 1) ? tpB : tpA;      // this tile's taps
                float4 (&tn)[2][4] = ((PT * 0 + pt) & 1) ? tpA : tpB;      // the next tile's
                // (source rows that do not change from one output row to the next leave both tiles' taps where they are: half the gathers)
                if (pt + 1 < PT) { if (!same_prev) load_taps(tn, r0, r1, pt + 1); }
                else if (yo + 1 < a.H && !same_next) load_taps(tn, r0_n, r1_n, 0);
                // ---- conv: C[ch][px] over six k-steps; a patch fragment (B operand) is read once for the wave's two channel tiles ----
                f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int ks = 0; ks < 6; ++ks) {
                    const unsigned char *p = sImg + ioff[ks] + pt * 32;
                    const uint2 h0 = trd(p + so0), h1 = trd(p + so1), l0 = trd(p + kImgBytes / 2 + so0), l1 = trd(p + kImgBytes / 2 + so1);
                    const uint4 bh = make_uint4(h0.x, h0.y, h1.x, h1.y), bl = make_uint4(l0.x, l0.y, l1.x, l1.y);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        acc[ct] = mfma<0>(wl[ct][ks], bh, acc[ct]);
                        acc[ct] = mfma<0>(wh[ct][ks], bl, acc[ct]);
                        acc[ct] = mfma<0>(wh[ct][ks], bh, acc[ct]);
                    }
                }
                // ---- epilogue: + bias, ReLU, + bilinear x2 of the trunk (align_corners = True), packed (hi, lo) store ----
                const float lx = tlx[pt];
                float v[2][4];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const float4 t00 = tc[ct][0], t01 = tc[ct][1], t10 = tc[ct][2], t11 = tc[ct][3];
                    const float a00[4] = {t00.x, t00.y, t00.z, t00.w}, a01[4] = {t01.x, t01.y, t01.z, t01.w};
                    const float a10[4] = {t10.x, t10.y, t10.z, t10.w}, a11[4] = {t11.x, t11.y, t11.z, t11.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float top = a00[r] * (1.f - lx) + a01[r] * lx, bot = a10[r] * (1.f - lx) + a11[r] * lx;
                        const float st = fmaxf(acc[ct][r] * a.inv_scale + bv[ct][r], 0.f);
                        v[ct][r] = top * (1.f - ly) + bot * ly + st;
                    }
                }
                float *rowp = a.out + ((long long)(n * a.H + yo) * a.W + x0s + pt * 16 + l16) * C;
                store_split8(rowp, chb + 4 * g, v[0], v[1]);
            }
            if (PT == 1 && !same_next) {      // (one tile per row: the next row's taps were loaded into the second buffer)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int k = 0; k < 4; ++k) tpA[ct][k] = tpB[ct][k];
            }
            r0 = r0_n; r1 = r1_n; same_prev = same_next;
            ly_c = sy_n - (float)y0_n;
        }
    }
}

}  // namespace

extern "C" int vs_stem7x7_up_split_stream(const float *img_padded, const float *w, int32_t w_scale_exp, const float *bias, const float *trunk, void *out,
                                          int32_t N, int32_t H, int32_t W, int32_t Hp, int32_t Wp, int32_t Cout, int32_t nwg, vs_stream_t stream) {
    VS_CHECK(img_padded && w && bias && trunk && out, "vs_stem7x7_up_split_stream: null pointer");
    VS_CHECK(N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 32 == 0, "vs_stem7x7_up_split_stream: H=%d must be even and W=%d a multiple of 32", H, W);
    VS_CHECK(Cout == 256, "vs_stem7x7_up_split_stream: Cout=%d must be 256", Cout);
    VS_CHECK(Hp >= H + 6 && Wp >= W + 6, "vs_stem7x7_up_split_stream: padded image must be at least (H+6) x (W+6), got %d x %d", Hp, Wp);
    VS_CHECK(nwg >= 1 && nwg <= 65535 && w_scale_exp >= -60 && w_scale_exp <= 60, "vs_stem7x7_up_split_stream: nwg / w_scale_exp out of range");
    VS_CHECK((((uintptr_t)out | (uintptr_t)trunk | (uintptr_t)bias) & 15) == 0, "vs_stem7x7_up_split_stream: out, trunk and bias must be 16-byte aligned");
    VS_CHECK((long long)N * H * W < 2147483647LL, "vs_stem7x7_up_split_stream: too large");
    StemStreamArgs a;
    a.img = img_padded; a.w = w; a.bias = bias; a.trunk = trunk; a.out = (float *)out;
    a.N = N; a.H = H; a.W = W; a.Hp = Hp; a.Wp = Wp;
    a.w_scale = ldexpf(1.f, w_scale_exp); a.inv_scale = ldexpf(1.f, -w_scale_exp);
    static const int pt = [] { const char *e = getenv("VS_STEM_PT"); return e ? atoi(e) : 2; }();
    if (pt == 1) hipLaunchKernelGGL(stem_up_stream_kernel<1>, dim3((unsigned)nwg), dim3(512), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(stem_up_stream_kernel<2>, dim3((unsigned)nwg), dim3(512), 0, (hipStream_t)stream, a);
    VS_HIP(hipGetLastError());
    return 0;
}
