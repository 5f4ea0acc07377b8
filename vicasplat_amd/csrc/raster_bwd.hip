// Tile-based Gaussian rasterizer, BACKWARD pass (incl. camera-twist gradients), hand-written for gfx950.
//
// Replaces diff_gaussian_rasterization._C.rasterize_gaussians_backward behind the autograd Function VicaSplat calls at
// /root/reference/src/model/decoder/cuda_splatting.py:226-235 (semantics: SURVEY.md Appendix B.5; mathematics pinned
// by oracle/raster_ref.c against PyTorch autograd + finite differences, tests/test_raster_oracle.py).
//
//   K1 render_backward_kernel : per (camera, tile) back-to-front replay from final_T / n_contrib; gradients w.r.t. the
//        screen-space mean (NDC units), conic (true partials), opacity, colour and depth of every (camera, Gaussian)
//        are summed over the wave's 8x8 pixels in registers (six DPP adds per component) and leave the wave as ONE f32
//        atomic per component -- a pixel-sized Gaussian is seen by tens of lanes of a wave, and per-lane atomics (the
//        upstream scheme) made this kernel 8x the forward.  Same 4-wave quadrant layout and wave-uniform footprint
//        cull as the forward, so a wave only touches the Gaussians that can reach its pixels.
//   K2 preprocess_backward_kernel : thread = Gaussian of a scene, loops over the scene's cameras and sums the
//        per-camera contributions in registers -> ONE plain store per output element (no atomics over views), plus a
//        block-reduced atomic for the per-camera twist gradient dL/dtau = (rho, theta), T_cw' = Exp(tau) T_cw.
#include "common.h"

#include <cstdlib>

namespace {

using vs::kGeomFloats;
using vs::kTile;

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                       -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                       0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                       -0.5900435899266435f};

// per-(camera,Gaussian) gradient record written by K1: mean2D.xy | conic.xyz | opacity | rgb | depth
constexpr int kG = 10;

// sum over the 4 lanes of a 2x2 pixel block (lanes 4k .. 4k+3), valid in all four: two quad-permute DPP adds
__device__ __forceinline__ float quad_total(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    return v;
}

// Same tile / quadrant / 2x2-block layout as the forward's render_kernel (raster_fwd.hip K5): per round of 64 staged entries every
// lane tests one entry's footprint against the quadrant's four block columns and rows, a block's list is column-mask & row-mask,
// and the block's 4 lanes walk it from the BACK (per-lane clz) -- one trip of the loop replays up to sixteen different Gaussians.
// Gradients of a (tile, entry) are accumulated in LDS (per batch of 256 entries: [256][10+1] floats; a block adds its 4-lane sums with
// ds_add_f32) and leave the workgroup once per batch as plain global atomics on the non-zero components: a pixel-sized Gaussian is
// seen by several blocks of several waves, and the per-(wave, entry) 64-lane reductions + global atomics of the previous version
// (and the per-entry scalar loop around them) were VALU time -- this kernel was 62 ms of the 24-scene training step.
__global__ void __launch_bounds__(256)
render_backward_kernel(int P, int W, int H, const int2 *__restrict__ ranges, const uint32_t *__restrict__ point_list,
                       const float *__restrict__ geom, const float *__restrict__ background, const float *__restrict__ final_T,
                       const int32_t *__restrict__ n_contrib, const float *__restrict__ dL_dpix, const float *__restrict__ dL_dpixdepth,
                       float *__restrict__ grec) {
    constexpr int NT = 256;
    __shared__ float4 sq0[NT], sq1[NT], sq2[NT];
    __shared__ uint32_t sid[NT];
    __shared__ float sgr[NT][kG + 1];   // [entry][component], 11-float rows: block adds and the flush below both spread over the banks
    __shared__ int s_max;
    const int gx = (W + kTile - 1) / kTile;
    const int tiles = gridDim.x;
    // (camera, tile) from the forward's longest-first launch order, stored behind the ranges (raster_fwd.hip, tile_scan_kernel)
    const int t_lin = reinterpret_cast<const int32_t *>(ranges + (size_t)gridDim.x * gridDim.y)[(size_t)blockIdx.y * gridDim.x + blockIdx.x];
    const int c = t_lin / tiles;
    const int tile = t_lin - c * tiles;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tile_x = tile % gx, tile_y = tile / gx;
    const int qx0 = tile_x * kTile + (wid & 1) * 8, qy0 = tile_y * kTile + (wid >> 1) * 8;
    const int sb = lane >> 2, bxi = sb & 3, byi = sb >> 2;     // 2x2 pixel block of this lane inside the wave's 8x8 quadrant
    const int pxi = qx0 + bxi * 2 + (lane & 1), pyi = qy0 + byi * 2 + ((lane >> 1) & 1);
    const float pixfx = (float)pxi, pixfy = (float)pyi;
    const float bcx = (float)qx0 + 0.5f, bcy = (float)qy0 + 0.5f;  // centre of block column / row 0; 2 px apart, half size 0.5 px
    const int2 rg = ranges[(size_t)c * tiles + tile];
    const float4 *__restrict__ g4 = reinterpret_cast<const float4 *>(geom + (size_t)c * P * kGeomFloats);
    const uint32_t *__restrict__ plist = point_list + rg.x;
    const bool inside = pxi < W && pyi < H;
    const size_t HW = (size_t)H * W;
    const size_t pix = (size_t)pyi * W + pxi;

    const int last = inside ? n_contrib[c * HW + pix] : 0;
    const float T_final = inside ? final_T[c * HW + pix] : 0.f;
    float T = T_final;
    float dLr = 0.f, dLg = 0.f, dLb = 0.f, dLd = 0.f;
    if (inside) {
        dLr = dL_dpix[(c * 3 + 0) * HW + pix]; dLg = dL_dpix[(c * 3 + 1) * HW + pix]; dLb = dL_dpix[(c * 3 + 2) * HW + pix];
        if (dL_dpixdepth) dLd = dL_dpixdepth[c * HW + pix];
    }
    const float bg_dot = background[3 * c] * dLr + background[3 * c + 1] * dLg + background[3 * c + 2] * dLb;
    float acc_r = 0.f, acc_g = 0.f, acc_b = 0.f, acc_d = 0.f, last_r = 0.f, last_g = 0.f, last_b = 0.f, last_d = 0.f, last_alpha = 0.f;
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;

    if (tid == 0) s_max = 0;
    __syncthreads();
    atomicMax(&s_max, last);
    __syncthreads();
    const int nmax = s_max;  // entries [0, nmax) of the tile list can contribute to some pixel of this tile
    int wave_max = last;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wave_max = max(wave_max, __shfl_xor(wave_max, o, 64));

    float *__restrict__ gr = grec + (size_t)c * P * kG;
    const int nb = (nmax + NT - 1) / NT;
    // register pipeline of the gather: records of the batch staged next, ids of the one after (batches run back to front)
    uint32_t g_cur = 0, g_nxt = 0;
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
    if (nb > 0 && (nb - 1) * NT + tid < nmax) g_cur = plist[(nb - 1) * NT + tid];
    if (nb > 1) g_nxt = plist[(nb - 2) * NT + tid];
    if (nb > 0 && (nb - 1) * NT + tid < nmax) {
        r0 = g4[(size_t)g_cur * 3 + 0]; r1 = g4[(size_t)g_cur * 3 + 1]; r2 = g4[(size_t)g_cur * 3 + 2];
    }
    // walk the list back to front in batches of NT: batch b covers positions [b*NT, min((b+1)*NT, nmax))
    for (int b = nb - 1; b >= 0; --b) {
        const int p0 = b * NT;
        const int cnt = min(NT, nmax - p0);
        __syncthreads();   // the previous batch's flush has read sgr / sid
        if (tid < cnt) {
            sq0[tid] = r0; sq1[tid] = r1; sq2[tid] = r2;
            sid[tid] = g_cur;
        }
#pragma unroll
        for (int e = 0; e < kG; ++e) sgr[tid][e] = 0.f;
        __syncthreads();
        g_cur = g_nxt;
        if (b > 0) {   // (batches below the last one are full)
            r0 = g4[(size_t)g_cur * 3 + 0]; r1 = g4[(size_t)g_cur * 3 + 1]; r2 = g4[(size_t)g_cur * 3 + 2];
        }
        if (b > 1) g_nxt = plist[(b - 2) * NT + tid];
        if (p0 < wave_max) {   // something in this batch can reach the wave's pixels
            for (int j0 = (cnt - 1) & ~63; j0 >= 0; j0 -= 64) {
                if (p0 + j0 >= wave_max) continue;
                const int je = j0 + lane;
                float tx = 0.f, ty = 0.f, ex = -1.f, ey = -1.f;
                if (je < cnt && p0 + je < wave_max) {
                    const float4 t = sq0[je];
                    tx = t.x - bcx; ty = t.y - bcy; ex = t.z + 0.5f; ey = t.w + 0.5f;
                }
                const unsigned long long mx0 = __ballot(fabsf(tx) <= ex), mx1 = __ballot(fabsf(tx - 2.0f) <= ex),
                                         mx2 = __ballot(fabsf(tx - 4.0f) <= ex), mx3 = __ballot(fabsf(tx - 6.0f) <= ex);
                const unsigned long long my0 = __ballot(fabsf(ty) <= ey), my1 = __ballot(fabsf(ty - 2.0f) <= ey),
                                         my2 = __ballot(fabsf(ty - 4.0f) <= ey), my3 = __ballot(fabsf(ty - 6.0f) <= ey);
                unsigned long long mine = ((bxi & 2) ? ((bxi & 1) ? mx3 : mx2) : ((bxi & 1) ? mx1 : mx0)) &
                                          ((byi & 2) ? ((byi & 1) ? my3 : my2) : ((byi & 1) ? my1 : my0));
                while (__any(mine != 0ull)) {
                    const bool blk = mine != 0ull;
                    const int jj = blk ? 63 - __builtin_clzll(mine) : 0;
                    mine &= ~(1ull << jj);
                    const int j = j0 + jj;
                    const int posn = p0 + j;  // 0-based position; upstream's `contributor` = posn + 1
                    const float4 q0 = sq0[j];
                    const float4 q1 = sq1[j];
                    const float dx = q0.x - pixfx, dy = q0.y - pixfy;
                    const float power = -0.5f * (q1.x * dx * dx + q1.z * dy * dy) - q1.y * dx * dy;
                    const float G = __expf(fminf(power, 0.0f));
                    const float alpha = fminf(0.99f, q1.w * G);
                    const bool act = blk && posn < last && power <= 0.0f && alpha >= 1.0f / 255.0f;
                    const unsigned long long am = __ballot(act);
                    if (am == 0ull) continue;
                    const float4 q2 = sq2[j];
                    float v[kG];
#pragma unroll
                    for (int e = 0; e < kG; ++e) v[e] = 0.f;
                    if (act) {
                        T = T / (1.0f - alpha);
                        const float dch = alpha * T;
                        float dL_dalpha = 0.f;
                        acc_r = last_alpha * last_r + (1.0f - last_alpha) * acc_r; last_r = q2.x; dL_dalpha += (q2.x - acc_r) * dLr;
                        acc_g = last_alpha * last_g + (1.0f - last_alpha) * acc_g; last_g = q2.y; dL_dalpha += (q2.y - acc_g) * dLg;
                        acc_b = last_alpha * last_b + (1.0f - last_alpha) * acc_b; last_b = q2.z; dL_dalpha += (q2.z - acc_b) * dLb;
                        acc_d = last_alpha * last_d + (1.0f - last_alpha) * acc_d; last_d = q2.w; dL_dalpha += (q2.w - acc_d) * dLd;
                        dL_dalpha *= T;
                        last_alpha = alpha;
                        dL_dalpha += (-T_final / (1.0f - alpha)) * bg_dot;
                        const float dL_dG = q1.w * dL_dalpha;
                        const float gdx = G * dx, gdy = G * dy;
                        v[0] = dL_dG * (-gdx * q1.x - gdy * q1.y) * ddelx_dx;
                        v[1] = dL_dG * (-gdy * q1.z - gdx * q1.y) * ddely_dy;
                        v[2] = -0.5f * gdx * dx * dL_dG;
                        v[3] = -1.0f * gdx * dy * dL_dG;
                        v[4] = -0.5f * gdy * dy * dL_dG;
                        v[5] = G * dL_dalpha;
                        v[6] = dch * dLr;
                        v[7] = dch * dLg;
                        v[8] = dch * dLb;
                        v[9] = dch * dLd;
                    }
#pragma unroll
                    for (int e = 0; e < kG; ++e) v[e] = quad_total(v[e]);
                    // one lane per block that has an active pixel adds the block's sums to the batch record of ITS entry
                    if ((lane & 3) == 0 && ((am >> (lane & 60)) & 0xFull) != 0ull) {
#pragma unroll
                        for (int e = 0; e < kG; ++e) atomicAdd(&sgr[j][e], v[e]);
                    }
                }
            }
        }
        __syncthreads();   // every wave's LDS adds of this batch are done
        // flush: lane = (entry slot 0..5, component 0..9), so the ten atomics of one record are ten adjacent lanes of one instruction
        // on ten adjacent addresses (one or two cache lines per record), zeros skipped
        {
            const int sub = lane / kG, e = lane - sub * kG;
            for (int k = 0; k < 64; k += 6) {
                const int j = wid * 64 + k + sub;
                if (lane < 6 * kG && k + sub < 64 && j < cnt) {
                    const float x = sgr[j][e];
                    if (x != 0.0f) atomicAdd(gr + (size_t)sid[j] * kG + e, x);
                }
            }
        }
    }
}

// column / row mask of block b out of the eight ballots of a footprint test
__device__ __forceinline__ unsigned long long sel8(int b, unsigned long long m0, unsigned long long m1, unsigned long long m2,
                                                   unsigned long long m3, unsigned long long m4, unsigned long long m5,
                                                   unsigned long long m6, unsigned long long m7) {
    const unsigned long long a0 = (b & 1) ? m1 : m0, a1 = (b & 1) ? m3 : m2, a2 = (b & 1) ? m5 : m4, a3 = (b & 1) ? m7 : m6;
    const unsigned long long c0 = (b & 2) ? a1 : a0, c1 = (b & 2) ? a3 : a2;
    return (b & 4) ? c1 : c0;
}

// Round 6, the route of every differentiated call: SEGMENT-PARALLEL replay, lane = one 2x2 pixel block of the tile (8 x 8 blocks; the block's
// four pixels are replayed by the same lane one after the other, so a trip of the survivor walk serves up to 64 (block, entry) pairs and no
// cross-lane sum exists).  The whole-list kernel above walks a tile's list in one piece, back to front, from final_T; PMC on the bench scene
// showed it bound by the LDS pipeline, not by arithmetic (ds_add_f32 retires about a lane per cycle: 85 % LDS-busy, VALU issue 0.32,
// 19.1 ms per 288 views against the forward's 4.0).  With VS_BUF_CHECKPOINT (the forward's blending state in front of every 512th entry,
// raster_fwd.hip) a wave replays ONE 512-entry segment of ONE tile, FRONT to back, from the state the forward had there:
//     T_i, D_i = sum_{k<=i} w_k (c_k . dL)   (w = alpha T: the forward's own recurrence)
//     dL/dalpha_i = T_i (c_i . dL) - (out . dL - D_i) / (1 - alpha_i)        (out = rendered pixel incl. background: the suffix sum + T_final bg)
// so the per-pixel state is T, D and four constants instead of the nine running values of the back-to-front form (a trip is ~45 instead of
// ~70 VALU per pixel, one v_rcp_f32 serves the division: the backward is float-tolerant, nothing here feeds an integer decision), and the
// work items are (tile, segment) pairs of at most eight rounds each.
template <bool DEPTH>
__global__ void __launch_bounds__(64)
render_backward_seg_kernel(int P, int W, int H, int tiles, const int2 *__restrict__ ranges, const uint32_t *__restrict__ point_list,
                           const float *__restrict__ geom, const float *__restrict__ ckpt, const int2 *__restrict__ cktab,
                           const int32_t *__restrict__ n_contrib, const float *__restrict__ out_color, const float *__restrict__ out_depth,
                           const float *__restrict__ dL_dpix, const float *__restrict__ dL_dpixdepth, float *__restrict__ grec) {
    // Accumulation WITHOUT LDS atomics (PMC: with one ds_add_f32 per sum and (block, entry) pair this kernel spent ~85 % of its time in the
    // LDS pipeline -- the instruction retires about one lane per cycle, and every trip issued nine of them: 15.8 ms per 288 views).  The
    // blocks an entry covers are a RECTANGLE of the 8 x 8 block grid (the footprint test is separable), so block (bx, by) has a rank inside
    // it: each (block, entry) pair owns a slot of a staging area -- entry j's slots start at the exclusive scan of the areas -- written with
    // plain 8-byte stores (9-10 sums + an epoch tag: a pair that was pruned or had no active pixel writes nothing and reads as zero), and
    // lane j, the OWNER of entry j, then adds its slots in registers.  The owners' sums go through the staging area once more so that one
    // global atomic instruction covers six records x ten adjacent floats (straight from the owners' registers an instruction touched 64
    // different cache lines and the kernel waited on the atomic path: 12.7 ms per launch).  Batches whose pairs outnumber the staging area
    // are cut into chunks of consecutive entries.
    constexpr int NT = 64;
    constexpr int SW = DEPTH ? 12 : 10;           // words per slot: sums, tag (, pad): a whole number of 8-byte stores
    constexpr int CAP = DEPTH ? 184 : 224;        // slots (>= 64: the largest rectangle)
    __shared__ float4 sq0[NT], sq1[NT], sq2[NT];
    __shared__ uint32_t sinfo[NT];
    __shared__ __attribute__((aligned(8))) float stage[CAP * SW];
    const int2 slot = cktab[blockIdx.x];
    if (slot.x < 0) return;
    const int t_lin = slot.x, seg = slot.y;
    const int gx = (W + kTile - 1) / kTile;
    const int c = t_lin / tiles;
    const int tile = t_lin - c * tiles;
    const int lane = threadIdx.x;
    const int tile_x = tile % gx, tile_y = tile / gx;
    const int bx = lane & 7, by = lane >> 3;
    const int x0 = tile_x * kTile, y0 = tile_y * kTile;
    const float pfx[2] = {(float)(x0 + 2 * bx), (float)(x0 + 2 * bx + 1)}, pfy[2] = {(float)(y0 + 2 * by), (float)(y0 + 2 * by + 1)};
    const float bcx = (float)x0 + 0.5f, bcy = (float)y0 + 0.5f;
    const size_t HW = (size_t)H * W;
    int last[4], lmax = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int px = x0 + 2 * bx + (p & 1), py = y0 + 2 * by + (p >> 1);
        last[p] = (px < W && py < H) ? n_contrib[c * HW + (size_t)py * W + px] : 0;
        lmax = max(lmax, last[p]);
    }
    int nmax = lmax;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o, 64));
    const int s0 = seg * vs::kCkSeg;
    if (s0 >= nmax) return;                       // every pixel of the tile was done before this segment (its checkpoint was never written)
    const int s1 = min(s0 + vs::kCkSeg, nmax);
    const int2 rg = ranges[t_lin];
    const float4 *__restrict__ g4 = reinterpret_cast<const float4 *>(geom + (size_t)c * P * kGeomFloats);
    const uint32_t *__restrict__ plist = point_list + rg.x + s0;
    const int n = s1 - s0;                        // entries of this segment that can act on some pixel

    // first batch's ids and records go out before the pixel data
    uint32_t g_cur = lane < n ? plist[lane] : 0u, g_nxt = 64 + lane < n ? plist[64 + lane] : 0u;
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
    if (lane < n) { r0 = g4[(size_t)g_cur * 3 + 0]; r1 = g4[(size_t)g_cur * 3 + 1]; r2 = g4[(size_t)g_cur * 3 + 2]; }

    float T[4], D[4], F[4], dLr[4], dLg[4], dLb[4], dLd[4];
    {
        float4 cT = make_float4(1.f, 1.f, 1.f, 1.f), cR = make_float4(0.f, 0.f, 0.f, 0.f), cG = cR, cB = cR, cD = cR;
        if (seg > 0) {
            const float4 *ck = reinterpret_cast<const float4 *>(ckpt + (size_t)blockIdx.x * vs::kCkFloats);
            cT = ck[lane]; cR = ck[64 + lane]; cG = ck[128 + lane]; cB = ck[192 + lane];
            if (DEPTH) cD = ck[256 + lane];
        }
        const float t4[4] = {cT.x, cT.y, cT.z, cT.w}, r4[4] = {cR.x, cR.y, cR.z, cR.w}, g_4[4] = {cG.x, cG.y, cG.z, cG.w},
                    b4[4] = {cB.x, cB.y, cB.z, cB.w}, d4[4] = {cD.x, cD.y, cD.z, cD.w};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int px = x0 + 2 * bx + (p & 1), py = y0 + 2 * by + (p >> 1);
            const bool inside = px < W && py < H;
            const size_t pix = (size_t)py * W + px;
            dLr[p] = dLg[p] = dLb[p] = dLd[p] = 0.f;
            float o_r = 0.f, o_g = 0.f, o_b = 0.f, o_d = 0.f;
            if (inside) {
                dLr[p] = dL_dpix[(c * 3 + 0) * HW + pix]; dLg[p] = dL_dpix[(c * 3 + 1) * HW + pix]; dLb[p] = dL_dpix[(c * 3 + 2) * HW + pix];
                o_r = out_color[(c * 3 + 0) * HW + pix]; o_g = out_color[(c * 3 + 1) * HW + pix]; o_b = out_color[(c * 3 + 2) * HW + pix];
                if (DEPTH) { dLd[p] = dL_dpixdepth[c * HW + pix]; o_d = out_depth[c * HW + pix]; }
            }
            T[p] = t4[p];
            D[p] = r4[p] * dLr[p] + g_4[p] * dLg[p] + b4[p] * dLb[p];
            F[p] = o_r * dLr[p] + o_g * dLg[p] + o_b * dLb[p];
            if (DEPTH) { D[p] += d4[p] * dLd[p]; F[p] += o_d * dLd[p]; }
        }
    }
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
    float *__restrict__ gr = grec + (size_t)c * P * kG;
    for (int k = lane; k < CAP; k += 64) stage[k * SW + (DEPTH ? 10 : 9)] = 0.f;     // tags: epochs start at 1
    int epoch = 0;

    for (int base = 0; base < n; base += NT) {
        const int cnt = min(NT, n - base);
        __syncthreads();   // the previous batch's owners have read the staging area / sinfo
        if (lane < cnt) { sq0[lane] = r0; sq1[lane] = r1; sq2[lane] = r2; }
        const uint32_t my_id = g_cur;
        const float4 r0_own = r0;
        g_cur = g_nxt;
        if (base + NT + lane < n) { r0 = g4[(size_t)g_cur * 3 + 0]; r1 = g4[(size_t)g_cur * 3 + 1]; r2 = g4[(size_t)g_cur * 3 + 2]; }
        if (base + 2 * NT + lane < n) g_nxt = plist[base + 2 * NT + lane];
        float tx = 0.f, ty = 0.f, ex = -1.f, ey = -1.f;
        if (lane < cnt) { tx = r0_own.x - bcx; ty = r0_own.y - bcy; ex = r0_own.z + 0.5f; ey = r0_own.w + 0.5f; }
        const bool c0 = fabsf(tx) <= ex, c1 = fabsf(tx - 2.0f) <= ex, c2 = fabsf(tx - 4.0f) <= ex, c3 = fabsf(tx - 6.0f) <= ex,
                   c4 = fabsf(tx - 8.0f) <= ex, c5 = fabsf(tx - 10.0f) <= ex, c6 = fabsf(tx - 12.0f) <= ex, c7 = fabsf(tx - 14.0f) <= ex;
        const bool w0 = fabsf(ty) <= ey, w1 = fabsf(ty - 2.0f) <= ey, w2 = fabsf(ty - 4.0f) <= ey, w3 = fabsf(ty - 6.0f) <= ey,
                   w4 = fabsf(ty - 8.0f) <= ey, w5 = fabsf(ty - 10.0f) <= ey, w6 = fabsf(ty - 12.0f) <= ey, w7 = fabsf(ty - 14.0f) <= ey;
        const unsigned long long mx0 = __ballot(c0), mx1 = __ballot(c1), mx2 = __ballot(c2), mx3 = __ballot(c3),
                                 mx4 = __ballot(c4), mx5 = __ballot(c5), mx6 = __ballot(c6), mx7 = __ballot(c7);
        const unsigned long long my0 = __ballot(w0), my1 = __ballot(w1), my2 = __ballot(w2), my3 = __ballot(w3),
                                 my4 = __ballot(w4), my5 = __ballot(w5), my6 = __ballot(w6), my7 = __ballot(w7);
        // this lane's ENTRY: the rectangle of blocks it covers (the same predicates as the ballots: the two views cannot disagree)
        const unsigned cb = (c0 ? 1u : 0u) | (c1 ? 2u : 0u) | (c2 ? 4u : 0u) | (c3 ? 8u : 0u) | (c4 ? 16u : 0u) | (c5 ? 32u : 0u) | (c6 ? 64u : 0u) | (c7 ? 128u : 0u);
        const unsigned rb = (w0 ? 1u : 0u) | (w1 ? 2u : 0u) | (w2 ? 4u : 0u) | (w3 ? 8u : 0u) | (w4 ? 16u : 0u) | (w5 ? 32u : 0u) | (w6 ? 64u : 0u) | (w7 ? 128u : 0u);
        const int nx = __popc(cb), ny = __popc(rb);
        const int area = nx * ny;
        const int cx0 = cb ? __builtin_ctz(cb) : 0, cy0 = rb ? __builtin_ctz(rb) : 0;
        int off_end = area;                       // inclusive scan over the lanes -> [off, off_end) = this entry's slots
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(off_end, o, 64);
            if (lane >= o) off_end += y;
        }
        const int off = off_end - area;
        sinfo[lane] = (uint32_t)cx0 | ((uint32_t)nx << 3) | ((uint32_t)cy0 << 7) | ((uint32_t)off << 10);
        __syncthreads();
        unsigned long long mine = sel8(bx, mx0, mx1, mx2, mx3, mx4, mx5, mx6, mx7) & sel8(by, my0, my1, my2, my3, my4, my5, my6, my7);
        {   // entries at or behind the last contributor of all four pixels of this block cannot act
            const int lim = lmax - (s0 + base);
            mine = lim >= 64 ? mine : (lim <= 0 ? 0ull : (mine & ((1ull << lim) - 1ull)));
        }
        float acc[10];
#pragma unroll
        for (int e = 0; e < 10; ++e) acc[e] = 0.f;
        int lo = 0;
        while (lo < cnt) {
            // chunk [lo, hi): the longest run of entries from lo whose slots fit the staging area
            const int off_lo = __builtin_amdgcn_readlane(off, __builtin_amdgcn_readfirstlane(lo));
            const unsigned long long fit = __ballot(lane >= lo && off_end - off_lo <= CAP);
            const unsigned long long rest = ~(fit >> lo);
            const int hi = rest ? min(64, lo + __builtin_ctzll(rest)) : 64;
            const unsigned long long cmask = (hi >= 64 ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull);
            ++epoch;
            const float tagf = __int_as_float(epoch);
            unsigned long long mc = mine & cmask;
        while (__any(mc != 0ull)) {
            const bool blk = mc != 0ull;
            const int j = blk ? __builtin_ctzll(mc) : 0;
            mc &= mc - 1ull;
            const int posn = s0 + base + j;   // 0-based position in the tile's list; upstream's `contributor` = posn + 1
            const float4 q0 = sq0[j];
            const float4 q1 = sq1[j];
            const float4 q2 = sq2[j];
            float A = 0.f, B = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, v5 = 0.f, v6 = 0.f, v7 = 0.f, v8 = 0.f, v9 = 0.f;
            bool any_act = false;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float dx = q0.x - pfx[p & 1], dy = q0.y - pfy[p >> 1];
                const float power = -0.5f * (q1.x * dx * dx + q1.z * dy * dy) - q1.y * dx * dy;
                const float G = __expf(fminf(power, 0.0f));
                const float alpha = fminf(0.99f, q1.w * G);
                const bool act = blk && posn < last[p] && power <= 0.0f && alpha >= 1.0f / 255.0f;
                any_act = any_act || act;
                const float w = act ? alpha * T[p] : 0.0f;            // the pixel's blending weight of this entry (0: nothing below changes)
                float cd = q2.x * dLr[p] + q2.y * dLg[p] + q2.z * dLb[p];
                if (DEPTH) cd += q2.w * dLd[p];
                D[p] += w * cd;
                const float rinv = __builtin_amdgcn_rcpf(1.0f - alpha);
                const float dL_dalpha = act ? T[p] * cd - (F[p] - D[p]) * rinv : 0.0f;
                T[p] -= w;                                             // T (1 - alpha)
                const float dL_dG = q1.w * dL_dalpha;
                const float a_ = dL_dG * (G * dx), b_ = dL_dG * (G * dy);
                A += a_; B += b_;
                s2 += a_ * dx; s3 += a_ * dy; s4 += b_ * dy;
                v5 += G * dL_dalpha;
                v6 += w * dLr[p]; v7 += w * dLg[p]; v8 += w * dLb[p];
                if (DEPTH) v9 += w * dLd[p];
            }
            if (any_act) {
                const uint32_t info = sinfo[j];
                const int slot = (int)(info >> 10) - off_lo + (by - (int)((info >> 7) & 7u)) * (int)((info >> 3) & 15u) + (bx - (int)(info & 7u));
                float2 *rec = reinterpret_cast<float2 *>(stage + slot * SW);
                rec[0] = make_float2(-(A * q1.x + B * q1.y) * ddelx_dx, -(B * q1.z + A * q1.y) * ddely_dy);
                rec[1] = make_float2(-0.5f * s2, -s3);
                rec[2] = make_float2(-0.5f * s4, v5);
                rec[3] = make_float2(v6, v7);
                if (DEPTH) { rec[4] = make_float2(v8, v9); rec[5] = make_float2(tagf, 0.f); }
                else rec[4] = make_float2(v8, tagf);
            }
        }
            __syncthreads();   // the chunk's stores are visible to the owners
            if (lane >= lo && lane < hi) {
                for (int sI = 0; sI < area; ++sI) {
                    const float2 *rec = reinterpret_cast<const float2 *>(stage + (off - off_lo + sI) * SW);
                    const float2 a0 = rec[0], a1 = rec[1], a2 = rec[2], a3 = rec[3], a4 = rec[4];
                    float2 a5 = make_float2(0.f, 0.f);
                    if (DEPTH) a5 = rec[5];
                    const bool ok = __float_as_int(DEPTH ? a5.x : a4.y) == epoch;
                    if (ok) {
                        acc[0] += a0.x; acc[1] += a0.y; acc[2] += a1.x; acc[3] += a1.y; acc[4] += a2.x; acc[5] += a2.y; acc[6] += a3.x; acc[7] += a3.y;
                        acc[8] += a4.x;
                        if (DEPTH) acc[9] += a4.y;
                    }
                }
            }
            __syncthreads();   // the owners are done with the staging area before the next chunk's stores
            lo = hi;
        }
        // one set of global atomics per (segment, entry).  Through the staging area once more, so that the ten atomics of one record are ten
        // adjacent lanes of one instruction (lane = (entry slot 0..5, component)): straight from the owners' registers every instruction
        // touched 64 different records -- 64 cache lines -- and the kernel waited on the atomic path instead of the LDS (12.7 ms per launch)
        sinfo[lane] = my_id;
#pragma unroll
        for (int e = 0; e < kG; ++e) stage[lane * SW + e] = (e < (DEPTH ? 10 : 9)) ? acc[e] : 0.f;
        __syncthreads();
        {
            const int sub = lane / kG, e = lane - sub * kG;
            for (int k = 0; k < cnt; k += 6) {
                const int j = k + sub;
                if (lane < 6 * kG && j < cnt) {
                    const float x = stage[j * SW + e];
                    if (x != 0.0f) atomicAdd(gr + (size_t)sinfo[j] * kG + e, x);
                }
            }
        }
    }
}

// K2, round 6.  Same mathematics as rounds 2-5 (one thread per Gaussian of a scene, the scene's cameras summed in registers, plain stores),
// re-cut for the machine: the round-5 kernel held 249 VGPRs and 59 KB of LDS (two waves per SIMD) and waited on two dependent global round
// trips per camera (radius -> gradient record): 0.20 of the VALU issue rate, 9.4 ms per 288 views.
//   * the view-direction gradient of the colour no longer forms dRGB/d(dir) per channel (3 x 45 coefficient reads, ~270 FLOPs): the
//     channel sum is taken FIRST, w_k = sum_ch sh[k][ch] g[ch] (45 FMAs, every coefficient read once from LDS), then
//     d(dir) = sum_k grad(basis_k) w_k (45 FMAs on polynomials shared by the channels);
//   * software pipeline over the cameras: the radius of camera k + 2 and the gradient record / clamp mask of camera k + 1 are in
//     flight while camera k is computed;
//   * the camera-twist sums leave per wave (shuffle reduction + one atomic per wave and component) instead of twelve block barriers per camera;
//   * the SH gradient (75 floats per Gaussian at a 300-byte stride: 64 cache lines per store instruction, 9.2 GB written for 5.6 GB of
//     results) leaves through the LDS image of the coefficients, which is dead by then: coalesced dword stores.
__global__ void __launch_bounds__(256)
preprocess_backward_kernel(const VsRasterIn in, const int32_t *__restrict__ radii, const uint8_t *__restrict__ clamped,
                           const float *__restrict__ grec, float *__restrict__ dL_dmeans3D, float *__restrict__ dL_dcov3D,
                           float *__restrict__ dL_dshs, float *__restrict__ dL_dcolors_precomp, float *__restrict__ dL_dopac,
                           float *__restrict__ dL_dmeans2D, float *__restrict__ dL_dtau) {
    const int s = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int P = in.P;
    const bool live = i < P;
    const size_t gi = (size_t)s * P + (live ? i : 0);
    const int W = in.width, H = in.height;
    const bool has_sh = in.colors_precomp == nullptr;
    const bool rgb_major = (in.flags & VS_RASTER_SH_RGB_MAJOR) != 0;
    const int M = in.sh_coeffs;
    const int deg = in.sh_degree;
    const int ncoef = deg >= 3 ? 16 : (deg + 1) * (deg + 1);
    const int lane = threadIdx.x & 63;

    float px = 0.f, py = 0.f, pz = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f, c4 = 0.f, c5 = 0.f;
    // the Gaussian's SH coefficients of bands 1..3 live in LDS, one column per thread (45 x 256 floats)
    __shared__ float s_sh[45][256];
#define SHV(k_, ch_) s_sh[((k_) - 1) * 3 + (ch_)][threadIdx.x]
#pragma unroll
    for (int k = 0; k < 45; ++k) s_sh[k][threadIdx.x] = 0.f;
    if (live) {
        px = in.means3D[3 * gi]; py = in.means3D[3 * gi + 1]; pz = in.means3D[3 * gi + 2];
        if (in.flags & VS_RASTER_COV_3X3) {
            const float *cv = in.cov3D + 9 * gi;
            c0 = cv[0]; c1 = cv[1]; c2 = cv[2]; c3 = cv[4]; c4 = cv[5]; c5 = cv[8];
        } else {
            const float *cv = in.cov3D + 6 * gi;
            c0 = cv[0]; c1 = cv[1]; c2 = cv[2]; c3 = cv[3]; c4 = cv[4]; c5 = cv[5];
        }
        if (has_sh) {
            const float *shp = in.shs + gi * (size_t)M * 3;
#pragma unroll
            for (int k = 1; k < 16; ++k)
#pragma unroll
                for (int ch = 0; ch < 3; ++ch)
                    if (k < ncoef) SHV(k, ch) = rgb_major ? shp[ch * M + k] : shp[3 * k + ch];
        }
    }
    const float S[3][3] = {{c0, c1, c2}, {c1, c3, c4}, {c2, c4, c5}};

    float g_mean[3] = {0.f, 0.f, 0.f}, g_cov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, g_op = 0.f, g_cp[3] = {0.f, 0.f, 0.f};

    // the scene's cameras, ascending (built per block by wave 0 with ballot compaction, as in the forward's preprocess_kernel)
    constexpr int kCamChunk = 512;
    __shared__ int cams[kCamChunk];
    __shared__ int ncam_s;
    constexpr int kParCams = 32, kParStride = 40;
    __shared__ __attribute__((aligned(16))) float cpar[kParCams * kParStride];
    // the camera list of chunk [cbase, cbase + kCamChunk) and the staged parameters of its first kParCams cameras (block-wide: two barriers)
    auto build_cams = [&](int cbase) -> int {
      __syncthreads();
      if (threadIdx.x < 64) {
          int n = 0;
          const int cend = min(in.num_cameras, cbase + kCamChunk);
          for (int cc0 = cbase; cc0 < cend; cc0 += 64) {
              const int c = cc0 + lane;
              const bool mine = c < cend && (in.cam_scene ? in.cam_scene[c] : (c % in.num_scenes)) == s;
              const unsigned long long m = __ballot(mine);
              if (mine) cams[n + __popcll(m & ((1ull << lane) - 1ull))] = c;
              n += __popcll(m);
          }
          if (lane == 0) ncam_s = n;
      }
      __syncthreads();
      const int ncam = ncam_s;
      // (the kernel argument is a struct of pointers: the compiler would fetch the camera parameters with vector loads in every iteration)
      const int nst = min(ncam, kParCams);
      for (int e = threadIdx.x; e < nst * kParStride; e += 256) {
          const int kk = e / kParStride, q = e - kk * kParStride, cc = cams[kk];
          float v = 0.f;
          if (q < 16) v = in.viewmatrix[16 * cc + q];
          else if (q < 32) v = in.projmatrix[16 * cc + q - 16];
          else if (q < 35) v = in.campos[3 * cc + q - 32];
          else if (q < 37) v = in.tanfov[2 * cc + q - 35];
          cpar[e] = v;
      }
      __syncthreads();
      return ncam;
    };
    for (int cbase = 0; cbase < in.num_cameras; cbase += kCamChunk) {
      const int ncam = build_cams(cbase);
      // ---- TWO passes over the cameras (the second one below, after this loop), so that the register file holds either the geometry path's state or the colour path's 48 SH
      // accumulators, not both (251 VGPRs -> two waves per SIMD when they shared one loop).  Each pass is software-pipelined: the radius
      // of camera k + 2 and its part of the gradient record of camera k + 1 are in flight while camera k is computed. ----
      // pass A: 2-D covariance, projected mean, depth -> g_mean, g_cov, g_op (and their part of the twist gradient)
      {
      int rad0 = 0, rad1 = 0;
      float2 ra0 = make_float2(0.f, 0.f), rb0 = ra0, rc0 = ra0, re0 = ra0;
      if (live && ncam > 0) rad0 = radii[(size_t)cams[0] * P + i];
      if (live && ncam > 1) rad1 = radii[(size_t)cams[1] * P + i];
      if (rad0 > 0) {
          const float2 *r2 = reinterpret_cast<const float2 *>(grec + ((size_t)cams[0] * P + i) * kG);
          ra0 = r2[0]; rb0 = r2[1]; rc0 = r2[2]; re0 = r2[4];
      }
      for (int kc = 0; kc < ncam; ++kc) {
        const int c = __builtin_amdgcn_readfirstlane(cams[kc]);   // wave-uniform
        int rad2 = 0;
        if (live && kc + 2 < ncam) rad2 = radii[(size_t)cams[kc + 2] * P + i];
        float2 ra1 = make_float2(0.f, 0.f), rb1 = ra1, rc1 = ra1, re1 = ra1;
        if (rad1 > 0) {   // (rad1 > 0 implies live and kc + 1 < ncam)
            const float2 *r2 = reinterpret_cast<const float2 *>(grec + ((size_t)cams[kc + 1] * P + i) * kG);
            ra1 = r2[0]; rb1 = r2[1]; rc1 = r2[2]; re1 = r2[4];
        }
        float tau[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (rad0 > 0) {
            float vm[16], pm[16], tanfovx, tanfovy;
            if (kc < kParCams) {                                  // (wave-uniform) LDS broadcast reads
                const float4 *q4 = reinterpret_cast<const float4 *>(cpar + kc * kParStride);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 a4 = q4[j], b4 = q4[4 + j];
                    vm[4 * j] = a4.x; vm[4 * j + 1] = a4.y; vm[4 * j + 2] = a4.z; vm[4 * j + 3] = a4.w;
                    pm[4 * j] = b4.x; pm[4 * j + 1] = b4.y; pm[4 * j + 2] = b4.z; pm[4 * j + 3] = b4.w;
                }
                const float4 c4_ = q4[8], d4 = q4[9];
                tanfovx = c4_.w; tanfovy = d4.x;
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) { vm[j] = in.viewmatrix[16 * c + j]; pm[j] = in.projmatrix[16 * c + j]; }
                tanfovx = in.tanfov[2 * c]; tanfovy = in.tanfov[2 * c + 1];
            }
            const float fx = (float)W / (2.0f * tanfovx), fy = (float)H / (2.0f * tanfovy);
            const float g2x = ra0.x, g2y = ra0.y, gA = rb0.x, gB = rb0.y, gC = rc0.x;
            g_op += rc0.y;
            const float gdep = re0.y;
            if (dL_dmeans2D) { const size_t ci = (size_t)c * P + i; dL_dmeans2D[2 * ci] = g2x; dL_dmeans2D[2 * ci + 1] = g2y; }
            const float vx = vm[0] * px + vm[4] * py + vm[8] * pz + vm[12];
            const float vy = vm[1] * px + vm[5] * py + vm[9] * pz + vm[13];
            const float vz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
            float gpc[3] = {0.f, 0.f, 0.f};
            float gR[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
            // ---- 2-D covariance path ----
            {
                const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
                const float rz = 1.0f / vz;
                const float txtz = vx * rz, tytz = vy * rz;
                const float tx = fminf(limx, fmaxf(-limx, txtz)) * vz, ty = fminf(limy, fmaxf(-limy, tytz)) * vz;
                const float xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
                const float ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
                const float tz2 = rz * rz, tz3 = tz2 * rz;
                const float J00 = fx * rz, J02 = -(fx * tx) * tz2, J11 = fy * rz, J12 = -(fy * ty) * tz2;
                float M0[3], M1[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    M0[k] = J00 * vm[4 * k + 0] + J02 * vm[4 * k + 2];
                    M1[k] = J11 * vm[4 * k + 1] + J12 * vm[4 * k + 2];
                }
                float t0[3], t1[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    t0[k] = S[k][0] * M0[0] + S[k][1] * M0[1] + S[k][2] * M0[2];
                    t1[k] = S[k][0] * M1[0] + S[k][1] * M1[1] + S[k][2] * M1[2];
                }
                const float a = M0[0] * t0[0] + M0[1] * t0[1] + M0[2] * t0[2] + 0.3f;
                const float b = M0[0] * t1[0] + M0[1] * t1[1] + M0[2] * t1[2];
                const float cc = M1[0] * t1[0] + M1[1] * t1[1] + M1[2] * t1[2] + 0.3f;
                const float det = a * cc - b * b;
                const float d2inv = 1.0f / (det * det + 0.0000001f);
                float ga = 0.f, gb = 0.f, gc = 0.f;
                if (d2inv != 0.0f) {
                    ga = d2inv * (-cc * cc * gA + b * cc * gB - b * b * gC);
                    gc = d2inv * (-b * b * gA + a * b * gB - a * a * gC);
                    gb = d2inv * (2.0f * b * cc * gA - (det + 2.0f * b * b) * gB + 2.0f * a * b * gC);
                }
                g_cov[0] += M0[0] * M0[0] * ga + M0[0] * M1[0] * gb + M1[0] * M1[0] * gc;
                g_cov[3] += M0[1] * M0[1] * ga + M0[1] * M1[1] * gb + M1[1] * M1[1] * gc;
                g_cov[5] += M0[2] * M0[2] * ga + M0[2] * M1[2] * gb + M1[2] * M1[2] * gc;
                g_cov[1] += 2.f * M0[0] * M0[1] * ga + (M0[0] * M1[1] + M0[1] * M1[0]) * gb + 2.f * M1[0] * M1[1] * gc;
                g_cov[2] += 2.f * M0[0] * M0[2] * ga + (M0[0] * M1[2] + M0[2] * M1[0]) * gb + 2.f * M1[0] * M1[2] * gc;
                g_cov[4] += 2.f * M0[1] * M0[2] * ga + (M0[1] * M1[2] + M0[2] * M1[1]) * gb + 2.f * M1[1] * M1[2] * gc;
                float gM0[3], gM1[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    gM0[k] = 2.f * ga * t0[k] + gb * t1[k];
                    gM1[k] = 2.f * gc * t1[k] + gb * t0[k];
                }
                float gJ00 = 0.f, gJ02 = 0.f, gJ11 = 0.f, gJ12 = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    gJ00 += gM0[k] * vm[4 * k + 0]; gJ02 += gM0[k] * vm[4 * k + 2];
                    gJ11 += gM1[k] * vm[4 * k + 1]; gJ12 += gM1[k] * vm[4 * k + 2];
                    gR[0][k] += J00 * gM0[k];
                    gR[1][k] += J11 * gM1[k];
                    gR[2][k] += J02 * gM0[k] + J12 * gM1[k];
                }
                gpc[0] += xmul * (-fx * tz2) * gJ02;
                gpc[1] += ymul * (-fy * tz2) * gJ12;
                gpc[2] += -fx * tz2 * gJ00 - fy * tz2 * gJ11 + (2.f * fx * tx) * tz3 * gJ02 + (2.f * fy * ty) * tz3 * gJ12;
            }
            // ---- projected mean path ----
            {
                const float hx = pm[0] * px + pm[4] * py + pm[8] * pz + pm[12];
                const float hy = pm[1] * px + pm[5] * py + pm[9] * pz + pm[13];
                const float hw = pm[3] * px + pm[7] * py + pm[11] * pz + pm[15];
                const float m_w = 1.0f / (hw + 0.0000001f);
                const float mul1 = hx * m_w * m_w, mul2 = hy * m_w * m_w;
                float gw[3];
                gw[0] = (pm[0] * m_w - pm[3] * mul1) * g2x + (pm[1] * m_w - pm[3] * mul2) * g2y;
                gw[1] = (pm[4] * m_w - pm[7] * mul1) * g2x + (pm[5] * m_w - pm[7] * mul2) * g2y;
                gw[2] = (pm[8] * m_w - pm[11] * mul1) * g2x + (pm[9] * m_w - pm[11] * mul2) * g2y;
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) gpc[rr] += vm[0 + rr] * gw[0] + vm[4 + rr] * gw[1] + vm[8 + rr] * gw[2];
            }
            // ---- depth path ----
            gpc[2] += gdep;
            // ---- assemble: p_C = R p + t ----
#pragma unroll
            for (int k = 0; k < 3; ++k) g_mean[k] += vm[4 * k + 0] * gpc[0] + vm[4 * k + 1] * gpc[1] + vm[4 * k + 2] * gpc[2];
            tau[0] += gpc[0]; tau[1] += gpc[1]; tau[2] += gpc[2];
            tau[3] += vy * gpc[2] - vz * gpc[1];
            tau[4] += vz * gpc[0] - vx * gpc[2];
            tau[5] += vx * gpc[1] - vy * gpc[0];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float r0_ = vm[4 * j + 0], r1_ = vm[4 * j + 1], r2_ = vm[4 * j + 2];
                tau[3] += r1_ * gR[2][j] - r2_ * gR[1][j];
                tau[4] += r2_ * gR[0][j] - r0_ * gR[2][j];
                tau[5] += r0_ * gR[1][j] - r1_ * gR[0][j];
            }
        }
        if (dL_dtau) {   // per wave: shuffle reduction, one atomic per component
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                float v = tau[k];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                if (lane == 0 && v != 0.f) atomicAdd(&dL_dtau[6 * c + k], v);
            }
        }
        rad0 = rad1; rad1 = rad2;
        ra0 = ra1; rb0 = rb1; rc0 = rc1; re0 = re1;
      }
      }
    }
    float g_sh[16][3];
#pragma unroll
    for (int k = 0; k < 16; ++k) g_sh[k][0] = g_sh[k][1] = g_sh[k][2] = 0.f;
    for (int cbase = 0; cbase < in.num_cameras; cbase += kCamChunk) {
      const int ncam = in.num_cameras > kCamChunk ? build_cams(cbase) : ncam_s;      // (one chunk: the list and the parameters of pass A stand)
      // pass B: colour -> g_sh (or g_cp), the view-direction part of g_mean and of the twist's translation
      {
      int rad0 = 0, rad1 = 0;
      float2 rd0 = make_float2(0.f, 0.f), re0 = rd0;
      uint32_t cl0 = 0;
      if (live && ncam > 0) rad0 = radii[(size_t)cams[0] * P + i];
      if (live && ncam > 1) rad1 = radii[(size_t)cams[1] * P + i];
      if (rad0 > 0) {
          const size_t ci0 = (size_t)cams[0] * P + i;
          const float2 *r2 = reinterpret_cast<const float2 *>(grec + ci0 * kG);
          rd0 = r2[3]; re0 = r2[4];
          if (has_sh) cl0 = clamped[ci0];
      }
      for (int kc = 0; kc < ncam; ++kc) {
        const int c = __builtin_amdgcn_readfirstlane(cams[kc]);
        int rad2 = 0;
        if (live && kc + 2 < ncam) rad2 = radii[(size_t)cams[kc + 2] * P + i];
        float2 rd1 = make_float2(0.f, 0.f), re1 = rd1;
        uint32_t cl1 = 0;
        if (rad1 > 0) {
            const size_t ci1 = (size_t)cams[kc + 1] * P + i;
            const float2 *r2 = reinterpret_cast<const float2 *>(grec + ci1 * kG);
            rd1 = r2[3]; re1 = r2[4];
            if (has_sh) cl1 = clamped[ci1];
        }
        float tau[3] = {0.f, 0.f, 0.f};
        if (rad0 > 0) {
            const float gcol[3] = {rd0.x, rd0.y, re0.x};
            if (!has_sh) {
                g_cp[0] += gcol[0]; g_cp[1] += gcol[1]; g_cp[2] += gcol[2];
            } else {
                float cp[3];
                if (kc < kParCams) {
                    const float4 c4_ = reinterpret_cast<const float4 *>(cpar + kc * kParStride)[8];
                    cp[0] = c4_.x; cp[1] = c4_.y; cp[2] = c4_.z;
                } else {
                    cp[0] = in.campos[3 * c]; cp[1] = in.campos[3 * c + 1]; cp[2] = in.campos[3 * c + 2];
                }
                const float dxo = px - cp[0], dyo = py - cp[1], dzo = pz - cp[2];
                const float rlen = 1.0f / sqrtf(dxo * dxo + dyo * dyo + dzo * dzo);
                const float x = dxo * rlen, y = dyo * rlen, z = dzo * rlen;
                float g[3];
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) g[ch] = (cl0 >> ch) & 1u ? 0.f : gcol[ch];
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                float ddx = 0.f, ddy = 0.f, ddz = 0.f;
                // w_k = sum over the channels of sh[k][ch] g[ch]: the colour gradient seen by coefficient row k (every coefficient read once)
#define WK(k_) (SHV(k_, 0) * g[0] + SHV(k_, 1) * g[1] + SHV(k_, 2) * g[2])
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) g_sh[0][ch] += SH_C0 * g[ch];
                if (deg > 0) {
                    const float b1 = -SH_C1 * y, b2 = SH_C1 * z, b3 = -SH_C1 * x;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) { g_sh[1][ch] += b1 * g[ch]; g_sh[2][ch] += b2 * g[ch]; g_sh[3][ch] += b3 * g[ch]; }
                    const float w1 = WK(1), w2 = WK(2), w3 = WK(3);
                    ddx = -SH_C1 * w3; ddy = -SH_C1 * w1; ddz = SH_C1 * w2;
                    if (deg > 1) {
                        const float b4 = SH_C2[0] * xy, b5 = SH_C2[1] * yz, b6 = SH_C2[2] * (2.0f * zz - xx - yy), b7 = SH_C2[3] * xz, b8 = SH_C2[4] * (xx - yy);
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) {
                            g_sh[4][ch] += b4 * g[ch]; g_sh[5][ch] += b5 * g[ch]; g_sh[6][ch] += b6 * g[ch]; g_sh[7][ch] += b7 * g[ch]; g_sh[8][ch] += b8 * g[ch];
                        }
                        const float w4 = WK(4), w5 = WK(5), w6 = WK(6), w7 = WK(7), w8 = WK(8);
                        ddx += SH_C2[0] * y * w4 - 2.0f * SH_C2[2] * x * w6 + SH_C2[3] * z * w7 + 2.0f * SH_C2[4] * x * w8;
                        ddy += SH_C2[0] * x * w4 + SH_C2[1] * z * w5 - 2.0f * SH_C2[2] * y * w6 - 2.0f * SH_C2[4] * y * w8;
                        ddz += SH_C2[1] * y * w5 + 4.0f * SH_C2[2] * z * w6 + SH_C2[3] * x * w7;
                        if (deg > 2) {
                            const float b9 = SH_C3[0] * y * (3.0f * xx - yy), b10 = SH_C3[1] * xy * z, b11 = SH_C3[2] * y * (4.0f * zz - xx - yy),
                                        b12 = SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy), b13 = SH_C3[4] * x * (4.0f * zz - xx - yy),
                                        b14 = SH_C3[5] * z * (xx - yy), b15 = SH_C3[6] * x * (xx - 3.0f * yy);
#pragma unroll
                            for (int ch = 0; ch < 3; ++ch) {
                                g_sh[9][ch] += b9 * g[ch]; g_sh[10][ch] += b10 * g[ch]; g_sh[11][ch] += b11 * g[ch]; g_sh[12][ch] += b12 * g[ch];
                                g_sh[13][ch] += b13 * g[ch]; g_sh[14][ch] += b14 * g[ch]; g_sh[15][ch] += b15 * g[ch];
                            }
                            const float w9 = WK(9), w10 = WK(10), w11 = WK(11), w12 = WK(12), w13 = WK(13), w14 = WK(14), w15 = WK(15);
                            ddx += SH_C3[0] * w9 * 6.0f * xy + SH_C3[1] * w10 * yz + SH_C3[2] * w11 * -2.0f * xy + SH_C3[3] * w12 * -6.0f * xz +
                                   SH_C3[4] * w13 * (-3.0f * xx + 4.0f * zz - yy) + SH_C3[5] * w14 * 2.0f * xz + SH_C3[6] * w15 * 3.0f * (xx - yy);
                            ddy += SH_C3[0] * w9 * 3.0f * (xx - yy) + SH_C3[1] * w10 * xz + SH_C3[2] * w11 * (-3.0f * yy + 4.0f * zz - xx) +
                                   SH_C3[3] * w12 * -6.0f * yz + SH_C3[4] * w13 * -2.0f * xy + SH_C3[5] * w14 * -2.0f * yz + SH_C3[6] * w15 * -6.0f * xy;
                            ddz += SH_C3[1] * w10 * xy + SH_C3[2] * w11 * 8.0f * yz + SH_C3[3] * w12 * 3.0f * (2.0f * zz - xx - yy) +
                                   SH_C3[4] * w13 * 8.0f * xz + SH_C3[5] * w14 * (xx - yy);
                        }
                    }
                }
#undef WK
                const float dot = x * ddx + y * ddy + z * ddz;
                const float gd0 = (ddx - x * dot) * rlen, gd1 = (ddy - y * dot) * rlen, gd2 = (ddz - z * dot) * rlen;
                g_mean[0] += gd0; g_mean[1] += gd1; g_mean[2] += gd2;
                if (dL_dtau) {
                    float r9[9];
                    if (kc < kParCams) {
                        const float4 *q4 = reinterpret_cast<const float4 *>(cpar + kc * kParStride);
                        const float4 a0 = q4[0], a1 = q4[1], a2 = q4[2];
                        r9[0] = a0.x; r9[1] = a0.y; r9[2] = a0.z; r9[3] = a1.x; r9[4] = a1.y; r9[5] = a1.z; r9[6] = a2.x; r9[7] = a2.y; r9[8] = a2.z;
                    } else {
#pragma unroll
                        for (int j = 0; j < 3; ++j) { r9[3 * j] = in.viewmatrix[16 * c + 4 * j]; r9[3 * j + 1] = in.viewmatrix[16 * c + 4 * j + 1]; r9[3 * j + 2] = in.viewmatrix[16 * c + 4 * j + 2]; }
                    }
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr) tau[rr] = r9[rr] * gd0 + r9[3 + rr] * gd1 + r9[6 + rr] * gd2;
                }
            }
        }
        if (dL_dtau && has_sh) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float v = tau[k];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                if (lane == 0 && v != 0.f) atomicAdd(&dL_dtau[6 * c + k], v);
            }
        }
        rad0 = rad1; rad1 = rad2;
        rd0 = rd1; re0 = re1; cl0 = cl1;
      }
      }
    }
    if (live) {
        dL_dmeans3D[3 * gi] = g_mean[0]; dL_dmeans3D[3 * gi + 1] = g_mean[1]; dL_dmeans3D[3 * gi + 2] = g_mean[2];
        if (in.flags & VS_RASTER_COV_3X3) {   // the caller differentiates the symmetric 3x3 layout: off-diagonal partials split in halves
            float *o = dL_dcov3D + 9 * gi;
            o[0] = g_cov[0]; o[4] = g_cov[3]; o[8] = g_cov[5];
            o[1] = o[3] = 0.5f * g_cov[1]; o[2] = o[6] = 0.5f * g_cov[2]; o[5] = o[7] = 0.5f * g_cov[4];
        } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) dL_dcov3D[6 * gi + k] = g_cov[k];
        }
        dL_dopac[gi] = g_op;
        if (!has_sh && dL_dcolors_precomp) {
            dL_dcolors_precomp[3 * gi] = g_cp[0]; dL_dcolors_precomp[3 * gi + 1] = g_cp[1]; dL_dcolors_precomp[3 * gi + 2] = g_cp[2];
        }
    }
    if (has_sh && dL_dshs) {
        // the SH gradient leaves through the (now dead) LDS image of the coefficients: a half block's 128 x 3M floats are one contiguous
        // run of the output, written as coalesced dwords
        float *stage = &s_sh[0][0];
        const int row = 3 * M;
        if (128 * row <= 45 * 256) {
#pragma unroll 1
            for (int h = 0; h < 2; ++h) {
                __syncthreads();   // the camera loop's last reads of s_sh / the previous half's copy are done
                if ((threadIdx.x >> 7) == h) {
                    float *o = stage + (threadIdx.x & 127) * row;
                    for (int k = 16; k < M; ++k)      // bands the rasterizer never reads: zero gradient
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) o[rgb_major ? ch * M + k : 3 * k + ch] = 0.f;
#pragma unroll
                    for (int k = 0; k < 16; ++k)
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch)
                            if (k < M) o[rgb_major ? ch * M + k : 3 * k + ch] = g_sh[k][ch];
                }
                __syncthreads();
                const int first = blockIdx.x * 256 + h * 128;                 // first Gaussian of this half
                const int nlive = min(128, P - first);
                if (nlive > 0) {
                    float *dst = dL_dshs + ((size_t)s * P + first) * (size_t)row;
                    for (int e = threadIdx.x; e < nlive * row; e += 256) dst[e] = stage[e];
                }
            }
        } else if (live) {      // (more than 30 coefficients per channel in memory: direct stores)
            float *o = dL_dshs + gi * (size_t)row;
            for (int k = 0; k < M; ++k)
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    float v = 0.f;
#pragma unroll
                    for (int kk = 0; kk < 16; ++kk) v = (kk == k) ? g_sh[kk][ch] : v;
                    if (rgb_major) o[ch * M + k] = v; else o[3 * k + ch] = v;
                }
        }
    }
}

}  // namespace

extern "C" int vs_raster_backward(const VsRasterIn *in, const VsRasterOut *saved, const VsRasterGrads *g, VsAllocFn alloc,
                                  void *actx, vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(in && saved && g && alloc, "vs_raster_backward: null argument");
    VS_CHECK(g->dL_dcolor && g->dL_dmeans3D && g->dL_dcov3D && g->dL_dopacities, "vs_raster_backward: null gradient pointer");
    const int C = in->num_cameras, P = in->P, W = in->width, H = in->height;
    if (P == 0) return 0;
    const int gx = vs::cdiv(W, kTile), gy = vs::cdiv(H, kTile), tiles = gx * gy;
    const float *geom = (const float *)saved->buffers[VS_BUF_GEOM];
    const uint8_t *clamped = (const uint8_t *)saved->buffers[VS_BUF_CLAMPED];
    const int2 *ranges = (const int2 *)saved->buffers[VS_BUF_TILE_RANGES];
    const uint32_t *point_list = (const uint32_t *)saved->buffers[VS_BUF_POINT_LIST];
    const float *final_T = (const float *)saved->buffers[VS_BUF_FINAL_T];
    const int32_t *n_contrib = (const int32_t *)saved->buffers[VS_BUF_N_CONTRIB];
    VS_CHECK(geom && clamped && ranges && point_list && final_T && n_contrib && saved->radii,
             "vs_raster_backward: forward state missing (buffers[] of the matching vs_raster_forward must be alive)");
    const size_t rec_bytes = (size_t)C * P * kG * sizeof(float);
    float *grec = (float *)alloc(actx, VS_BUF_MISC, rec_bytes);
    VS_CHECK(grec, "vs_raster_backward: allocator returned null");
    VS_HIP(hipMemsetAsync(grec, 0, rec_bytes, stream));
    if (g->dL_dtau) VS_HIP(hipMemsetAsync(g->dL_dtau, 0, (size_t)C * 6 * sizeof(float), stream));
    if (g->dL_dmeans2D) VS_HIP(hipMemsetAsync(g->dL_dmeans2D, 0, (size_t)C * P * 2 * sizeof(float), stream));
    if (saved->num_rendered > 0) {
        // one wave per tile (lane = 2x2 block) when there are enough tiles to fill the chip with single waves, as the forward's render
        // kernel; four waves per tile (a quadrant each) for small calls.  VS_RBWD_WAVES = 1 | 4 forces one (read per call: the tests run both).
        const char *fw = getenv("VS_RBWD_WAVES");
        const float *ckpt = (const float *)saved->buffers[VS_BUF_CHECKPOINT];
        // segment-parallel replay whenever the forward saved checkpoints (every differentiated call of this package); the whole-list kernel
        // (four waves per tile) for a caller without VS_RASTER_SAVE_FOR_BACKWARD.  VS_RBWD_WAVES=4 forces the latter (read per call: tests).
        const bool seg = ckpt && !(fw && atoi(fw) == 4);
        if (seg) {
            VS_CHECK(saved->color && (!g->dL_ddepth || saved->depth), "vs_raster_backward: the checkpoint route reads the rendered colour / depth of the forward");
            const size_t ck_slots = (size_t)(saved->num_rendered >> vs::kCkShift) + (size_t)tiles * C;
            const int2 *cktab = reinterpret_cast<const int2 *>(ckpt + ck_slots * vs::kCkFloats);
            if (g->dL_ddepth)
                hipLaunchKernelGGL((render_backward_seg_kernel<true>), dim3((unsigned)ck_slots), dim3(64), 0, stream, P, W, H, tiles, ranges, point_list, geom,
                                   ckpt, cktab, n_contrib, saved->color, saved->depth, g->dL_dcolor, g->dL_ddepth, grec);
            else
                hipLaunchKernelGGL((render_backward_seg_kernel<false>), dim3((unsigned)ck_slots), dim3(64), 0, stream, P, W, H, tiles, ranges, point_list, geom,
                                   ckpt, cktab, n_contrib, saved->color, saved->depth, g->dL_dcolor, g->dL_ddepth, grec);
        } else {
            hipLaunchKernelGGL(render_backward_kernel, dim3(tiles, C), dim3(256), 0, stream, P, W, H, ranges, point_list, geom,
                               in->background, final_T, n_contrib, g->dL_dcolor, g->dL_ddepth, grec);
        }
    }
    hipLaunchKernelGGL(preprocess_backward_kernel, dim3(vs::cdiv(P, 256), in->num_scenes), dim3(256), 0, stream, *in, saved->radii,
                       clamped, grec, g->dL_dmeans3D, g->dL_dcov3D, g->dL_dshs, g->dL_dcolors_precomp, g->dL_dopacities,
                       g->dL_dmeans2D, g->dL_dtau);
    VS_HIP(hipGetLastError());
    return 0;
}
