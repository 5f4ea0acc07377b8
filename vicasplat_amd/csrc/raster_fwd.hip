// Tile-based 3D Gaussian splatting rasterizer, forward pass, hand-written for gfx950 (MI355X).
//
// Replaces diff_gaussian_rasterization._C.rasterize_gaussians as called from
// /root/reference/src/model/decoder/cuda_splatting.py:207-235 (semantics: SURVEY.md Appendix B.1-B.4).
// NOT a translation of upstream's pipeline.  MI355X-first differences:
//   * ONE batched launch set for all cameras of all scenes (grid.y = camera); Gaussians are shared by the cameras
//     of a scene instead of being replicated per view (decoder_splatting_cuda.py:86-89 copies them Vt times);
//   * binning without a global 64-bit radix sort: per-(camera,tile) counts -> one tiny scan gives the tile ranges
//     directly -> atomic-cursor scatter -> one workgroup per tile sorts its own segment in LDS.  The sort key is
//     (depth_bits << 32 | gaussian_index): a total order, so the result equals upstream's stable sort of
//     (tile | depth) keys emitted in Gaussian order, independent of the scatter order;
//   * per-Gaussian render attributes are packed into one 48-byte record (3 x dwordx4 gathers per staged entry);
//   * tan(fov) etc. are read from device memory: no per-view .item() host sync (cuda_splatting.py:210-211).
//
// Arithmetic contract (shared with oracle/raster_ref.c): everything that feeds an integer decision (cull, radius,
// tile rectangle, depth key) is computed in float32 with FMA contraction OFF in the operation order below, so
// radii / tile ranges / sorted ids are bit-identical to the CPU oracle.
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

__device__ __forceinline__ int f2i(float x) {
    x = fminf(fmaxf(x, -1.0e6f), 1.0e6f);
    return (int)x;
}

// ---------------------------------------------------------------------------------------------------------
// K1: preprocess.  grid = (ceil(P/256), S), block = 256.  A thread owns ONE Gaussian of scene s: it loads the
// Gaussian's 58 input floats once and then walks over the cameras of that scene (the 232 input bytes per Gaussian
// come from HBM once per scene, not once per view).  The scene's camera list is built once per block (wave 0, ballot
// compaction, ascending) so that the loop only visits its own cameras.  Per camera the block bins its (Gaussian,tile)
// instances in an LDS histogram and flushes one global atomic per non-empty bin, instead of one contended global atomic
// per instance (measured: wave-aggregated global atomics are 2x slower here -- every block of a scene hits the same few
// counters at the same time).  Two histograms alternate between cameras and the flush leaves its bins zeroed, so a camera
// costs ONE barrier (accumulate | flush) instead of three (zero | accumulate | flush).
// ---------------------------------------------------------------------------------------------------------
constexpr int kHistTiles = 4096;  // LDS histogram capacity (bins are launch-sized, up to 2 x 16 KiB in K1); larger tile grids use global atomics directly
constexpr int kCamChunk = 2048;   // cameras scanned per list build (8 KiB of LDS)

struct __attribute__((packed, aligned(4))) f3_t { float x, y, z; };

__global__ void __launch_bounds__(256)
preprocess_kernel(const VsRasterIn in, float *__restrict__ geom, ushort4 *__restrict__ rect, uint8_t *__restrict__ clamped,
                  int32_t *__restrict__ radii, float *__restrict__ depth, int32_t *__restrict__ tile_count) {
#pragma clang fp contract(off)
    __shared__ int cams[kCamChunk];
    // round 5: the parameters of the scene's first kParCams cameras (view 16 | proj 16 | campos 3 | tanfov 2 | pad: 40 floats each), staged
    // once per block.  The kernel argument is a struct of pointers, so the compiler cannot prove that the loop's stores leave the camera
    // arrays alone and fetched them with VECTOR loads at the top of every camera iteration (PMC: 5 scalar-memory instructions per wave, 43 %
    // of the wave cycles waiting); an LDS broadcast read costs a tenth of that latency.  Same values, same arithmetic: bit-identical.
    constexpr int kParCams = 32, kParStride = 40;
    __shared__ __attribute__((aligned(16))) float cpar[kParCams * kParStride];
    extern __shared__ int hist_dyn[];   // 2 x tiles bins (launch-sized: 2 KiB for a 256 x 256 image instead of 2 x 16 KiB for the 4096-tile capacity)
    __shared__ int ncam_s;
    const int s = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int P = in.P;
    const bool live = i < P;
    const size_t gi = (size_t)s * P + (live ? i : 0);
    const int W = in.width, H = in.height;
    const int gx = (W + kTile - 1) / kTile, gy = (H + kTile - 1) / kTile;
    const int tiles = gx * gy;
    const int lane = threadIdx.x & 63;
    const bool use_lds = tiles <= kHistTiles;
    if (use_lds)
        for (int t = threadIdx.x; t < 2 * tiles; t += 256) hist_dyn[t] = 0;   // (ordered by the camera-list barrier below)
    int hb = 0;

    // ---- per-Gaussian inputs, loaded once ----
    float px = 0.f, py = 0.f, pz = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f, c4 = 0.f, c5 = 0.f, opac = 0.f;
    float sh[16][3];
    const bool has_sh = in.colors_precomp == nullptr;
    const int deg = in.sh_degree;
    if (live) {
        const f3_t m = *reinterpret_cast<const f3_t *>(in.means3D + 3 * gi);
        px = m.x; py = m.y; pz = m.z;
        if (in.flags & VS_RASTER_COV_3X3) {
            const float *__restrict__ cv = in.cov3D + 9 * gi;
            c0 = cv[0]; c1 = cv[1]; c2 = cv[2]; c3 = cv[4]; c4 = cv[5]; c5 = cv[8];
        } else {
            const f3_t a = *reinterpret_cast<const f3_t *>(in.cov3D + 6 * gi);
            const f3_t b = *reinterpret_cast<const f3_t *>(in.cov3D + 6 * gi + 3);
            c0 = a.x; c1 = a.y; c2 = a.z; c3 = b.x; c4 = b.y; c5 = b.z;
        }
        opac = in.opacities[gi];
        if (has_sh) {
            const float *__restrict__ shp = in.shs + gi * (size_t)in.sh_coeffs * 3;
            const int ncoef = deg >= 3 ? 16 : (deg + 1) * (deg + 1);
            if (in.flags & VS_RASTER_SH_RGB_MAJOR) {
                const int M = in.sh_coeffs;
#pragma unroll
                for (int k = 0; k < 16; ++k)
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) sh[k][ch] = k < ncoef ? shp[ch * M + k] : 0.f;
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    if (k < ncoef) {
                        const f3_t v = *reinterpret_cast<const f3_t *>(shp + 3 * k);
                        sh[k][0] = v.x; sh[k][1] = v.y; sh[k][2] = v.z;
                    } else {
                        sh[k][0] = sh[k][1] = sh[k][2] = 0.f;
                    }
                }
            }
        } else {
            const f3_t v = *reinterpret_cast<const f3_t *>(in.colors_precomp + 3 * gi);
            sh[0][0] = v.x; sh[0][1] = v.y; sh[0][2] = v.z;
        }
    }
    const float S[3][3] = {{c0, c1, c2}, {c1, c3, c4}, {c2, c4, c5}};

    for (int cbase = 0; cbase < in.num_cameras; cbase += kCamChunk) {
      // cameras of this scene in [cbase, cbase + kCamChunk), ascending
      if (cbase > 0) __syncthreads();
      if (threadIdx.x < 64) {
          int n = 0;
          const int cend = min(in.num_cameras, cbase + kCamChunk);
          for (int c0 = cbase; c0 < cend; c0 += 64) {
              const int c = c0 + lane;
              const bool mine = c < cend && (in.cam_scene ? in.cam_scene[c] : (c % in.num_scenes)) == s;
              const unsigned long long m = __ballot(mine);
              if (mine) cams[n + __popcll(m & ((1ull << lane) - 1ull))] = c;
              n += __popcll(m);
          }
          if (lane == 0) ncam_s = n;
      }
      __syncthreads();
      const int ncam = ncam_s;
      {   // stage the camera parameters (element e of camera slot kk: one thread each, strided)
          const int nst = min(ncam, kParCams);
          for (int e = threadIdx.x; e < nst * kParStride; e += 256) {
              const int kk = e / kParStride, q = e - kk * kParStride, cc = cams[kk];
              float v = 0.f;
              if (q < 16) v = in.viewmatrix[16 * cc + q];
              else if (q < 32) v = in.projmatrix[16 * cc + q - 16];
              else if (q < 35) v = in.campos[3 * cc + q - 32];
              else if (q < 37) v = in.tanfov[2 * cc + q - 35];
              else if (q == 37) v = (float)W / (2.0f * in.tanfov[2 * cc]);          // focal_x, focal_y: the same expression the loop used per lane
              else if (q == 38) v = (float)H / (2.0f * in.tanfov[2 * cc + 1]);
              cpar[e] = v;
          }
          __syncthreads();
      }
      for (int k = 0; k < ncam; ++k) {
        const int c = __builtin_amdgcn_readfirstlane(cams[k]);   // wave-uniform
        const size_t ci = (size_t)c * P + i;
        bool visible = false;
        int rminx = 0, rminy = 0, rmaxx = 0, rmaxy = 0;
        if (live) {
            float vm[16], pm[16], cp[3], tanfovx, tanfovy, focal_x, focal_y;
            if (k < kParCams) {                                  // (wave-uniform) LDS broadcast reads, 16 bytes each
                const float4 *q4 = reinterpret_cast<const float4 *>(cpar + k * kParStride);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 a4 = q4[j], b4 = q4[4 + j];
                    vm[4 * j] = a4.x; vm[4 * j + 1] = a4.y; vm[4 * j + 2] = a4.z; vm[4 * j + 3] = a4.w;
                    pm[4 * j] = b4.x; pm[4 * j + 1] = b4.y; pm[4 * j + 2] = b4.z; pm[4 * j + 3] = b4.w;
                }
                const float4 c4 = q4[8], d4 = q4[9];
                cp[0] = c4.x; cp[1] = c4.y; cp[2] = c4.z; tanfovx = c4.w; tanfovy = d4.x; focal_x = d4.y; focal_y = d4.z;
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) { vm[j] = in.viewmatrix[16 * c + j]; pm[j] = in.projmatrix[16 * c + j]; }
                cp[0] = in.campos[3 * c]; cp[1] = in.campos[3 * c + 1]; cp[2] = in.campos[3 * c + 2];
                tanfovx = in.tanfov[2 * c]; tanfovy = in.tanfov[2 * c + 1];
                focal_x = (float)W / (2.0f * tanfovx); focal_y = (float)H / (2.0f * tanfovy);
            }
            int radius_i = 0;
            uint32_t clamp_bits = 0;
            do {
                const float vx = vm[0] * px + vm[4] * py + vm[8] * pz + vm[12];
                const float vy = vm[1] * px + vm[5] * py + vm[9] * pz + vm[13];
                const float vz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
                if (!(vz > 0.2f)) break;
                const float hx = pm[0] * px + pm[4] * py + pm[8] * pz + pm[12];
                const float hy = pm[1] * px + pm[5] * py + pm[9] * pz + pm[13];
                const float hw = pm[3] * px + pm[7] * py + pm[11] * pz + pm[15];
                const float p_w = 1.0f / (hw + 0.0000001f);
                const float projx = hx * p_w, projy = hy * p_w;

                const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
                const float txtz = vx / vz, tytz = vy / vz;
                const float tx = fminf(limx, fmaxf(-limx, txtz)) * vz;
                const float ty = fminf(limy, fmaxf(-limy, tytz)) * vz;
                const float tz = vz;
                const float J00 = focal_x / tz, J02 = -(focal_x * tx) / (tz * tz);
                const float J11 = focal_y / tz, J12 = -(focal_y * ty) / (tz * tz);
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
                if (det == 0.0f) break;
                const float det_inv = 1.0f / det;
                const float conx = cc * det_inv, cony = -b * det_inv, conz = a * det_inv;
                const float mid = 0.5f * (a + cc);
                const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
                const float lambda1 = mid + sq, lambda2 = mid - sq;
                const float my_radius = ceilf(3.0f * sqrtf(fmaxf(lambda1, lambda2)));
                const float pixx = ((projx + 1.0f) * (float)W - 1.0f) * 0.5f;
                const float pixy = ((projy + 1.0f) * (float)H - 1.0f) * 0.5f;
                rminx = min(gx, max(0, f2i((pixx - my_radius) / (float)kTile)));
                rminy = min(gy, max(0, f2i((pixy - my_radius) / (float)kTile)));
                rmaxx = min(gx, max(0, f2i((pixx + my_radius + (float)(kTile - 1)) / (float)kTile)));
                rmaxy = min(gy, max(0, f2i((pixy + my_radius + (float)(kTile - 1)) / (float)kTile)));
                if ((rmaxx - rminx) * (rmaxy - rminy) <= 0) break;

                float rgb[3];
                if (!has_sh) {
                    rgb[0] = sh[0][0]; rgb[1] = sh[0][1]; rgb[2] = sh[0][2];
                } else {
                    const float dx = px - cp[0], dy = py - cp[1], dz = pz - cp[2];
                    // (v_rsq_f32 + three multiplies instead of an IEEE sqrt and three IEEE divisions -- ~45 VALU of this kernel's ~600 per
                    //  (Gaussian, camera): the view direction only feeds the COLOUR, which is float-tolerant (1 ulp of the direction = 1e-7 of
                    //  the colour; tests bound it at 2e-5); everything that feeds an integer decision or the conic keeps the exact forms)
                    const float rlen = __builtin_amdgcn_rsqf(dx * dx + dy * dy + dz * dz);
                    const float x = dx * rlen, y = dy * rlen, z = dz * rlen;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        float r = SH_C0 * sh[0][ch];
                        if (deg > 0) {
                            r = r - SH_C1 * y * sh[1][ch] + SH_C1 * z * sh[2][ch] - SH_C1 * x * sh[3][ch];
                            if (deg > 1) {
                                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                                r = r + SH_C2[0] * xy * sh[4][ch] + SH_C2[1] * yz * sh[5][ch] +
                                    SH_C2[2] * (2.0f * zz - xx - yy) * sh[6][ch] + SH_C2[3] * xz * sh[7][ch] +
                                    SH_C2[4] * (xx - yy) * sh[8][ch];
                                if (deg > 2) {
                                    r = r + SH_C3[0] * y * (3.0f * xx - yy) * sh[9][ch] + SH_C3[1] * xy * z * sh[10][ch] +
                                        SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[11][ch] +
                                        SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[12][ch] +
                                        SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[13][ch] +
                                        SH_C3[5] * z * (xx - yy) * sh[14][ch] + SH_C3[6] * x * (xx - 3.0f * yy) * sh[15][ch];
                                }
                            }
                        }
                        r += 0.5f;
                        if (r < 0.0f) clamp_bits |= (1u << ch);
                        rgb[ch] = fmaxf(r, 0.0f);
                    }
                }
                // conservative footprint of {alpha >= 1/255}: alpha = o*exp(power) >= 1/255  <=>  1/2 d^T Q d <= tau with
                // tau = ln(255 o), Q = conic = inverse of the dilated 2-D covariance (a, b, cc): the ellipse's axis-aligned
                // half extents are sqrt(2 tau a), sqrt(2 tau cc).  A pixel outside it is skipped by the render loop's
                // alpha < 1/255 test anyway, so culling with (1% + 0.05 px) margin never changes a result.
                // (hardware log2 / sqrt: their 1-ulp error is five orders of magnitude inside the margin)
                const float tau = __builtin_amdgcn_logf(255.0f * opac) * 0.6931471805599453f;
                float ext_x = -1.0f, ext_y = -1.0f;
                if (tau > 0.0f) {
                    ext_x = __builtin_amdgcn_sqrtf(2.0f * tau * a) * 1.01f + 0.05f;
                    ext_y = __builtin_amdgcn_sqrtf(2.0f * tau * cc) * 1.01f + 0.05f;
                }
                float4 *g4 = reinterpret_cast<float4 *>(geom + ci * kGeomFloats);
                g4[0] = make_float4(pixx, pixy, ext_x, ext_y);
                g4[1] = make_float4(conx, cony, conz, opac);
                g4[2] = make_float4(rgb[0], rgb[1], rgb[2], vz);
                depth[ci] = vz;
                radius_i = f2i(my_radius);
                visible = true;
            } while (false);
            radii[ci] = radius_i;
            rect[ci] = visible ? make_ushort4((unsigned short)rminx, (unsigned short)rminy, (unsigned short)rmaxx, (unsigned short)rmaxy)
                               : make_ushort4(0, 0, 0, 0);
            clamped[ci] = (uint8_t)clamp_bits;
        }
        int32_t *tc = tile_count + (size_t)c * tiles;
        int *hist = hist_dyn + hb * tiles;
        if (visible) {
            for (int y = rminy; y < rmaxy; ++y)
                for (int x = rminx; x < rmaxx; ++x) {
                    if (use_lds) atomicAdd(&hist[y * gx + x], 1);
                    else atomicAdd(&tc[y * gx + x], 1);
                }
        }
        if (use_lds) {
            __syncthreads();   // this camera's bins are complete; the other histogram (next camera) was zeroed by its last flush
            for (int t = threadIdx.x; t < tiles; t += 256) {
                const int v = hist[t];
                if (v) { atomicAdd(&tc[t], v); hist[t] = 0; }
            }
            hb ^= 1;
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------------------
// K2: exclusive scan of the C*tiles counts -> tile ranges; zeroes the counters (re-used as scatter cursors).
// Single workgroup (n is a few thousand).  misc[0] = R (int64), misc[1] = max tile population.
// ---------------------------------------------------------------------------------------------------------
// It also writes the LAUNCH ORDER of the per-(camera, tile) workgroups behind the ranges: tile ids sorted by descending size class
// (floor(log2(population)), 32 classes).  The lists are heavy-tailed (median 845 entries, p99 18 k, max 23 k on the bench scene) and
// workgroups are dispatched in grid order: in camera-major order a 20 k-entry tile that starts late is the tail of the kernel (list
// scheduling of the measured populations: 1.03-1.12 x the balanced time for 512-2048 resident workgroups, longest-first 1.00;
// tools/tile_stats.py).  The order only changes which workgroup takes which tile.
constexpr int kOrderClasses = 32;
__global__ void __launch_bounds__(1024) tile_scan_kernel(int32_t *__restrict__ count, int2 *__restrict__ ranges, int n,
                                                          long long *__restrict__ misc, long long capacity) {
    __shared__ long long wave_tot[16];
    __shared__ long long carry_s;
    __shared__ int max_s[16];
    __shared__ int cls_cnt[kOrderClasses];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    int32_t *const order = reinterpret_cast<int32_t *>(ranges + n);
    if (tid < kOrderClasses) cls_cnt[tid] = 0;
    __syncthreads();
    for (int idx = tid; idx < n; idx += 1024) atomicAdd(&cls_cnt[31 - __clz(count[idx] | 1)], 1);   // (counts are still intact here)
    __syncthreads();
    if (tid == 0) {   // first slot of each class, largest class first
        int run = 0;
        for (int k = kOrderClasses - 1; k >= 0; --k) { const int c_ = cls_cnt[k]; cls_cnt[k] = run; run += c_; }
    }
    __syncthreads();
    for (int idx = tid; idx < n; idx += 1024) order[atomicAdd(&cls_cnt[31 - __clz(count[idx] | 1)], 1)] = idx;
    __syncthreads();
    if (tid == 0) carry_s = 0;
    int local_max = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int idx = base + tid;
        const int v = idx < n ? count[idx] : 0;
        local_max = max(local_max, v);
        long long x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            long long y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wave_tot[wid] = x;
        __syncthreads();
        long long woff = 0;
        for (int w = 0; w < wid; ++w) woff += wave_tot[w];
        const long long carry = carry_s;
        const long long incl = carry + woff + x;
        if (idx < n) {
            ranges[idx] = make_int2((int)(incl - v), (int)incl);
            count[idx] = 0;
        }
        __syncthreads();
        if (tid == 1023) carry_s = incl;
        __syncthreads();
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local_max = max(local_max, __shfl_xor(local_max, o, 64));
    if (lane == 0) max_s[wid] = local_max;
    __syncthreads();
    if (tid == 0) {
        int m = 0;
        for (int w = 0; w < 16; ++w) m = max(m, max_s[w]);
        misc[0] = carry_s;
        misc[1] = m;
        misc[2] = (capacity > 0 && carry_s > capacity) ? 1 : 0;
    }
    // capacity mode (no host round trip sized the buffers): more instances than the buffers hold -> every tile list becomes empty,
    // the scatter / sort / render kernels then touch nothing and the caller finds misc[2] set
    if (capacity > 0 && carry_s > capacity)
        for (int idx = tid; idx < n; idx += 1024) ranges[idx] = make_int2(0, 0);
}

// ---------------------------------------------------------------------------------------------------------
// K3: scatter (depth_bits<<32 | gaussian) keys into the per-(camera,tile) segments.  grid = (ceil(P/256), C).
// Block-aggregated: instances are counted in an LDS histogram, one returning global atomic per non-empty bin
// reserves the block's slots in the tile segment, then each instance takes an LDS-local slot.
// Reads 12 bytes per (camera, Gaussian) pair: the tile rectangle (all-zero = not visible) and the view-space depth that K1
// wrote to its own compact array (the 48-byte geom record holds it too, but at a 48-byte stride: 4x the bytes of this whole kernel).
// ---------------------------------------------------------------------------------------------------------
constexpr int kScatterPer = 4;   // Gaussians per thread: the three block-wide steps (zero, reserve, slot) are paid once per 1024 pairs

__global__ void __launch_bounds__(256)
scatter_kernel(int P, int tiles, int gx, const float *__restrict__ depth, const ushort4 *__restrict__ rect,
               const int2 *__restrict__ ranges, int32_t *__restrict__ cursor, unsigned long long *__restrict__ keys,
               const long long *__restrict__ misc) {
    extern __shared__ int hist_dyn[];   // tiles bins + tiles segment bases (launch-sized)
    int *hist = hist_dyn, *base = hist_dyn + tiles;
    if (misc[2] != 0) return;   // capacity mode, more instances than `keys` holds: the scan emptied every tile list, nothing to place
    const int c = blockIdx.y;
    const bool use_lds = tiles <= kHistTiles;
    const size_t t0 = (size_t)c * tiles;
    ushort4 r[kScatterPer];
    unsigned long long key[kScatterPer];
#pragma unroll
    for (int u = 0; u < kScatterPer; ++u) {
        const int i = (blockIdx.x * kScatterPer + u) * 256 + threadIdx.x;
        const size_t ci = (size_t)c * P + (i < P ? i : 0);
        r[u] = make_ushort4(0, 0, 0, 0);
        if (i < P) r[u] = rect[ci];
        const bool visible = r[u].z > r[u].x && r[u].w > r[u].y;
        if (!visible) r[u] = make_ushort4(0, 0, 0, 0);
        key[u] = 0;
        if (visible) key[u] = ((unsigned long long)__float_as_uint(depth[ci]) << 32) | (unsigned)i;
    }
    if (!use_lds) {
#pragma unroll
        for (int u = 0; u < kScatterPer; ++u)
            for (int y = r[u].y; y < r[u].w; ++y)
                for (int x = r[u].x; x < r[u].z; ++x) {
                    const size_t t = t0 + y * gx + x;
                    const int slot = atomicAdd(&cursor[t], 1);
                    keys[(size_t)ranges[t].x + slot] = key[u];
                }
        return;
    }
    for (int t = threadIdx.x; t < tiles; t += 256) hist[t] = 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kScatterPer; ++u)
        for (int y = r[u].y; y < r[u].w; ++y)
            for (int x = r[u].x; x < r[u].z; ++x) atomicAdd(&hist[y * gx + x], 1);
    __syncthreads();
    for (int t = threadIdx.x; t < tiles; t += 256) {
        const int v = hist[t];
        if (v) {
            base[t] = ranges[t0 + t].x + atomicAdd(&cursor[t0 + t], v);
            hist[t] = 0;
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kScatterPer; ++u)
        for (int y = r[u].y; y < r[u].w; ++y)
            for (int x = r[u].x; x < r[u].z; ++x) {
                const int t = y * gx + x;
                const int slot = atomicAdd(&hist[t], 1);
                keys[(size_t)base[t] + slot] = key[u];
            }
}

// ---------------------------------------------------------------------------------------------------------
// K4: one workgroup per (camera, tile) sorts its key segment; output = sorted Gaussian ids.
//   n <= 1024 : bitonic network in LDS (8 KiB);
//   larger    : workgroup-level stable LSD radix sort, 8-bit digits, ping-pong between the key segment and a scratch
//               segment in global memory (L2-resident).  O(n) per pass, any n; passes whose digit is constant over the
//               segment (high depth-exponent bits, unused index bits) are skipped.  Stable ranking inside a 256-key
//               chunk uses wave ballots (8 per key) + a 4x256 per-wave digit-count table.
// ---------------------------------------------------------------------------------------------------------
constexpr int kSortLds = 1024;
constexpr int kSortWave = 256;    // lists up to this length are sorted by one wave in registers inside tile_sort_kernel
constexpr int kSegShift = 6;      // a tile list [x, y) owns the run-table slots (x >> kSegShift) + t .. (y >> kSegShift) + t
constexpr int kRunKeys = 112;     // published runs are cut every kRunKeys keys (at bucket starts): with ~10 keys per bucket nearly every run is <= 128
                                  // keys, i.e. the 28-stage network on two registers instead of 36 stages on four

template <typename KeyPtr>
__device__ __forceinline__ void bitonic_sort(KeyPtr a, int N) {
    for (int k = 2; k <= N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (N >> 1); t += 256) {
                const int i = 2 * t - (t & (j - 1));
                const int l = i + j;
                const bool up = ((i & k) == 0);
                const unsigned long long x = a[i], y = a[l];
                if ((x > y) == up) { a[i] = y; a[l] = x; }
            }
            __syncthreads();
        }
    }
}

// Bitonic sort of 256 keys held by ONE wave, element e = r * 64 + lane in v[r]: compare distances below 64 are lane
// exchanges (two 32-bit shuffles per key), 64 and 128 are register pairs of the same lane.  No LDS image, no barrier.
// value of lane (l ^ J): DPP permutes for J < 16 (VALU rate, no LDS crossbar traffic), ds_bpermute for 16 and 32
template <int J>
__device__ __forceinline__ unsigned lane_xor(unsigned v) {
    if constexpr (J == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);        // quad_perm [1,0,3,2]
    else if constexpr (J == 2) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    else if constexpr (J == 4) {  // (l ^ 7) ^ 3: row_half_mirror then quad_perm [3,2,1,0]
        const int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);
        return (unsigned)__builtin_amdgcn_update_dpp(0, t, 0x1B, 0xF, 0xF, false);
    } else if constexpr (J == 8) {  // (l ^ 15) ^ 7: row_mirror then row_half_mirror
        const int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);
        return (unsigned)__builtin_amdgcn_update_dpp(0, t, 0x141, 0xF, 0xF, false);
    } else return (unsigned)__shfl_xor((int)v, J, 64);
}
template <int J>
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v) {
    const unsigned lo = lane_xor<J>((unsigned)v), hi = lane_xor<J>((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
template <int NR, int K, int J>
__device__ __forceinline__ void wave_bitonic_step(unsigned long long (&v)[NR], int lane) {
    if constexpr (J >= 64) {
        constexpr int dr = J >> 6;  // 1 or 2: partner register r ^ dr
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if ((r & dr) == 0) {
                const bool up = ((r * 64) & K) == 0;  // lane bits are below 64 <= J < K
                const unsigned long long x = v[r], y = v[r | dr];
                const bool sw = (x > y) == up;
                v[r] = sw ? y : x; v[r | dr] = sw ? x : y;
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int e = r * 64 + lane;
            const unsigned long long x = v[r], y = shfl_xor_u64<J>(x);
            const bool up = (e & K) == 0, lower = (lane & J) == 0;
            v[r] = ((x > y) == (up == lower)) ? y : x;
        }
    }
    if constexpr (J > 1) wave_bitonic_step<NR, K, (J >> 1)>(v, lane);
}
template <int NR, int K>
__device__ __forceinline__ void wave_bitonic_merge(unsigned long long (&v)[NR], int lane) {
    wave_bitonic_step<NR, K, (K >> 1)>(v, lane);
    if constexpr (K < 64 * NR) wave_bitonic_merge<NR, (K << 1)>(v, lane);
}
// 64 * NR keys of one wave (NR = 2: 128 keys, 28 compare-exchange stages on two registers; NR = 4: 256 keys, 36 stages on four)
template <int NR>
__device__ __forceinline__ void wave_bitonic(unsigned long long (&v)[NR], int lane) { wave_bitonic_merge<NR, 2>(v, lane); }

constexpr int kDigitBits = 8;
constexpr int kBins = 1 << kDigitBits;

// Peers of this lane's digit inside the wave (lanes holding the same digit), by 8 ballots.
__device__ __forceinline__ unsigned long long digit_peers(int d, bool valid) {
    unsigned long long m = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < kDigitBits; ++bit) {
        const unsigned long long bb = __ballot((d >> bit) & 1);
        m &= ((d >> bit) & 1) ? bb : ~bb;
    }
    return m;
}

__global__ void __launch_bounds__(256)
tile_sort_kernel(const int2 *__restrict__ ranges, unsigned long long *__restrict__ keys, uint32_t *__restrict__ point_list,
                 unsigned long long *__restrict__ scratch, int2 *__restrict__ segs) {
    // one LDS object, 24.6 KiB: the bitonic image | the bucket starts and cursors; the radix path's per-wave digit
    // tables ([current | next pass][wave][digit]) reuse the bucket arrays, which are dead by then
    constexpr int kBucketsT = 2048;
    __shared__ __attribute__((aligned(16))) unsigned char sort_lds[kSortLds * 8 + (kBucketsT + 4) * 4 + kBucketsT * 4];
    unsigned long long *const skeys = reinterpret_cast<unsigned long long *>(sort_lds);
    int *const bstart = reinterpret_cast<int *>(sort_lds + kSortLds * 8);
    int *const bcur = bstart + kBucketsT + 4;
    int (*const whist)[4][kBins] = reinterpret_cast<int (*)[4][kBins]>(sort_lds + kSortLds * 8);
    static_assert(2 * 4 * kBins * 4 <= (kBucketsT + 4) * 4 + kBucketsT * 4, "digit tables must fit in the bucket arrays");
    __shared__ int wave_tot[4];
    const size_t ntile_all = (size_t)gridDim.x * gridDim.y;
    const size_t t = (size_t)reinterpret_cast<const int32_t *>(ranges + ntile_all)[(size_t)blockIdx.y * gridDim.x + blockIdx.x];   // longest first
    const int2 rg = ranges[t];
    const int n = rg.y - rg.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // this tile's slots in the segment table consumed by segment_sort_kernel (see the bucket path): clear them first
    const size_t seg_base = (size_t)(rg.x >> kSegShift) + t;
    const int seg_slots = (rg.y >> kSegShift) - (rg.x >> kSegShift) + 1;
    if (segs) {
        for (int k = tid; k < seg_slots; k += 256) segs[seg_base + k] = make_int2(0, 0);
        __syncthreads();
    }
    if (n <= 0) return;
    unsigned long long *a = keys + rg.x;
    if (n <= kSortWave) {
        // round 5: a list of <= 256 keys is ONE wave's register sort (the network segment_sort_kernel uses): no LDS image, no barrier
        if (wid != 0) return;
        if (n <= 64) {
            unsigned long long v[1] = {lane < n ? a[lane] : ~0ull};
            wave_bitonic<1>(v, lane);
            if (lane < n) point_list[rg.x + lane] = (uint32_t)v[0];
        } else if (n <= 128) {
            unsigned long long v[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) v[r] = (r * 64 + lane) < n ? a[r * 64 + lane] : ~0ull;
            wave_bitonic<2>(v, lane);
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (r * 64 + lane < n) point_list[rg.x + r * 64 + lane] = (uint32_t)v[r];
        } else {
            unsigned long long v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (r * 64 + lane) < n ? a[r * 64 + lane] : ~0ull;
            wave_bitonic<4>(v, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r * 64 + lane < n) point_list[rg.x + r * 64 + lane] = (uint32_t)v[r];
        }
        return;
    }
    if (!scratch) {     // (a caller without the scratch buffer: not reached from vs_raster_forward, which provides it whenever a list is longer)
        if (n > kSortLds) return;
        int N = 2;
        while (N < n) N <<= 1;
        for (int k = tid; k < N; k += 256) skeys[k] = k < n ? a[k] : ~0ull;
        __syncthreads();
        bitonic_sort(skeys, N);
        for (int k = tid; k < n; k += 256) point_list[rg.x + k] = (uint32_t)skeys[k];
        return;
    }
    unsigned long long *b = scratch + rg.x;
    // ---- every longer list (round 5: from 257 keys on, not only the > 1024-key lists -- the 55-stage block-wide bitonic network with a barrier
    // per stage that lists of 257 .. 1024 keys used to take was most of this kernel's time on the bench step, whose median list is 845
    // keys): ONE order-preserving bucket pass + register / in-LDS sorts.  A radix pass scatters 8-byte keys over the whole
    // tile (every store its own cache line), so passes are what cost: bucket the keys by (depth - min) >> shift into
    // kBuckets ranges (one scattered pass, any order inside a bucket), then sort runs of consecutive buckets of <= 1024
    // keys with the bitonic network on the full 64-bit (depth, index) key and write the index list coalesced.  Tiles
    // whose depths pile up in one bucket (> 512 keys) take the radix path below instead. ----
    {
        constexpr int kBuckets = kBucketsT, kSeg = kSortLds / 2;
        __shared__ unsigned s_mn, s_mx;
        if (tid == 0) { s_mn = ~0u; s_mx = 0u; }
        for (int k = tid; k < kBuckets; k += 256) bcur[k] = 0;
        // Round 5: the depth range of a list longer than 8192 keys comes from the keys a thread holds plus a strided SAMPLE of 256 keys, padded
        // by 1/32 of itself on both sides; keys outside it land in the first / last bucket (the bucket function stays monotone, so the
        // result is the same exact sort -- only the balance of the buckets depends on the sample; a list whose tails pile up takes the
        // fat / clustered routes below like any other piled-up list).  And the keys are read ONCE: a thread keeps its first kKeep keys
        // (tiles up to 256 * kKeep = 8192 entries completely) in registers between the histogram and the scatter pass.  Before: three
        // reads of the list from HBM (min / max, histogram, scatter): 8.7 GB of the forward's 39.8 GB on the bench step.  (Measured and not
        // kept: 16 kept keys and 72 VGPRs for six instead of four workgroups per CU -- 1.92 -> 2.63 ms: more lists in flight than L2 holds.)
        constexpr int kKeep = 32;
        // the key loads go out FIRST (32 per thread in flight under the setup above); the range is then taken from the registers -- exact
        // for lists up to 8192 keys -- plus, for longer lists, the strided sample of the rest: one dependent memory round trip less
        unsigned long long kreg[kKeep];
#pragma unroll
        for (int j = 0; j < kKeep; ++j) {
            const int i = tid + j * 256;
            kreg[j] = i < n ? a[i] : 0ull;
        }
        unsigned mn = ~0u, mx = 0u;
        if (n > kKeep * 256) {
            const unsigned hi = (unsigned)(a[(int)(((long long)tid * n) >> 8)] >> 32);
            mn = mx = hi;
        }
#pragma unroll
        for (int j = 0; j < kKeep; ++j)
            if (tid + j * 256 < n) { const unsigned hi = (unsigned)(kreg[j] >> 32); mn = min(mn, hi); mx = max(mx, hi); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mn = min(mn, (unsigned)__shfl_xor((int)mn, o, 64)); mx = max(mx, (unsigned)__shfl_xor((int)mx, o, 64)); }
        __syncthreads();                       // s_mn / s_mx initialised, bcur zeroed
        if (lane == 0) { atomicMin(&s_mn, mn); atomicMax(&s_mx, mx); }
        __syncthreads();
        {
            const unsigned pad = n > kKeep * 256 ? ((s_mx - s_mn) >> 5) + 1u : 0u;     // (sampled ranges only)
            mn = s_mn > pad ? s_mn - pad : 0u;
            mx = s_mx < 0xffffffffu - pad ? s_mx + pad : 0xffffffffu;
        }
        const unsigned range = mx - mn;
        const int shift = range >= (unsigned)kBuckets ? (32 - __clz(range)) - 11 : 0;
        auto bucket_of = [&](unsigned long long key) -> int {
            const unsigned hi = (unsigned)(key >> 32);
            return hi <= mn ? 0 : (int)min((hi - mn) >> shift, (unsigned)(kBuckets - 1));
        };
#pragma unroll
        for (int j = 0; j < kKeep; ++j)
            if (tid + j * 256 < n) atomicAdd(&bcur[bucket_of(kreg[j])], 1);
        for (int i = tid + kKeep * 256; i < n; i += 256) atomicAdd(&bcur[bucket_of(a[i])], 1);
        __syncthreads();
        int loc[kBuckets / 256], sum = 0, big = 0;
#pragma unroll
        for (int q = 0; q < kBuckets / 256; ++q) { loc[q] = bcur[tid * (kBuckets / 256) + q]; sum += loc[q]; big = max(big, loc[q]); }
        int x = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wave_tot[wid] = x;
        const bool clustered = __syncthreads_or(big > kSeg);
        const bool fat = __syncthreads_or(big > 128);  // some bucket too large for the per-wave register sort
        if (!clustered) {
            int run = x - sum;
            for (int w = 0; w < wid; ++w) run += wave_tot[w];
#pragma unroll
            for (int q = 0; q < kBuckets / 256; ++q) { bstart[tid * (kBuckets / 256) + q] = run; bcur[tid * (kBuckets / 256) + q] = run; run += loc[q]; }
            if (tid == 255) bstart[kBuckets] = n;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < kKeep; ++j)
                if (tid + j * 256 < n) b[atomicAdd(&bcur[bucket_of(kreg[j])], 1)] = kreg[j];
            for (int i = tid + kKeep * 256; i < n; i += 256) {
                const unsigned long long key = a[i];
                b[atomicAdd(&bcur[bucket_of(key)], 1)] = key;
            }
            __syncthreads();
            auto boundary = [&](int k, int seg, int nseg) -> int {  // largest bucket start <= k * seg (tile end for k == nseg)
                if (k >= nseg) return n;
                const int target = k * seg;
                int lo = 0, hi = kBuckets;  // bstart[0] == 0 <= target
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (bstart[mid] <= target) lo = mid; else hi = mid - 1;
                }
                return bstart[lo];
            };
            if (!fat) {
                // every bucket <= 128 keys: runs of consecutive buckets cut every kRunKeys keys are < 256 keys.  They are not
                // sorted here (a 20k-key tile would walk 160 of them on 4 waves while small tiles' CUs idle): publish
                // them, segment_sort_kernel sorts one run per wave across the whole chip.
                constexpr int kW = kRunKeys;
                const int nseg = (n + kW - 1) / kW;  // <= seg_slots
                for (int k = tid; k < nseg; k += 256) {
                    const int s_lo = boundary(k, kW, nseg), m_ = boundary(k + 1, kW, nseg) - s_lo;
                    segs[seg_base + k] = make_int2(rg.x + s_lo, m_);
                }
                return;
            }
            const int nseg = (n + kSeg - 1) / kSeg;
            int s_lo = 0;
            for (int k = 0; k < nseg; ++k) {
                const int s_hi = boundary(k + 1, kSeg, nseg), m_ = s_hi - s_lo;
                if (m_ > 0) {
                    int N = 2;
                    while (N < m_) N <<= 1;
                    for (int q = tid; q < N; q += 256) skeys[q] = q < m_ ? b[s_lo + q] : ~0ull;
                    __syncthreads();
                    bitonic_sort(skeys, N);
                    for (int q = tid; q < m_; q += 256) point_list[rg.x + s_lo + q] = (uint32_t)skeys[q];
                    __syncthreads();
                }
                s_lo = s_hi;
            }
            return;
        }
        __syncthreads();
    }
    // ---- clustered tiles: stable LSD radix sort on the DEPTH half of the key (bits 32..63), four 8-bit passes.  Each wave
    // owns a contiguous quarter of the positions and its own digit counters, so the scatter loop has NO workgroup barrier:
    // a wave ranks its 64 keys with ballots, bumps its private running offsets, and adds every key it places to the
    // histogram of the wave that will own that position in the next pass. ----
    const int quarter = (n + 3) >> 2;
    const int w_lo = wid * quarter, w_hi = min(n, w_lo + quarter);
    auto wave_hist = [&](const unsigned long long *src, int shift, int buf) {  // per-wave histogram of one digit
        for (int i0 = w_lo; i0 < w_hi; i0 += 64) {
            const int i = i0 + lane;
            const bool valid = i < w_hi;
            const int d = valid ? (int)((src[i] >> shift) & (kBins - 1)) : 0;
            const unsigned long long m = digit_peers(d, valid);
            if (valid && (m & ((1ull << lane) - 1ull)) == 0) whist[buf][wid][d] += __popcll(m);
        }
    };
    for (int k = tid; k < 2 * 4 * kBins; k += 256) (&whist[0][0][0])[k] = 0;
    __syncthreads();
    wave_hist(a, 32, 0);
    int cur = 0;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 32 + kDigitBits * pass;
        __syncthreads();  // whist[cur] complete
        const int c0 = whist[cur][0][tid], c1 = whist[cur][1][tid], c2 = whist[cur][2][tid], c3 = whist[cur][3][tid];
        const int tot = c0 + c1 + c2 + c3;
        if (__syncthreads_or(tot == n)) {  // constant digit: nothing moves; only the next digit's histogram is needed
            if (pass < 3) {
                whist[cur][0][tid] = 0; whist[cur][1][tid] = 0; whist[cur][2][tid] = 0; whist[cur][3][tid] = 0;
                __syncthreads();
                wave_hist(a, shift + kDigitBits, cur);
            }
            continue;
        }
        // exclusive scan, digit-major / wave-minor: whist[cur][w][d] becomes wave w's first output slot for digit d
        int x = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wave_tot[wid] = x;
        __syncthreads();
        int run = x - tot;
        for (int w = 0; w < wid; ++w) run += wave_tot[w];
        whist[cur][0][tid] = run; whist[cur][1][tid] = run + c0; whist[cur][2][tid] = run + c0 + c1; whist[cur][3][tid] = run + c0 + c1 + c2;
        const int nxt = cur ^ 1;
        whist[nxt][0][tid] = 0; whist[nxt][1][tid] = 0; whist[nxt][2][tid] = 0; whist[nxt][3][tid] = 0;
        __syncthreads();
        unsigned long long key_next = (w_lo + lane < w_hi) ? a[w_lo + lane] : 0ull;
        for (int i0 = w_lo; i0 < w_hi; i0 += 64) {
            const unsigned long long key = key_next;
            const bool valid = i0 + lane < w_hi;
            if (i0 + 64 + lane < w_hi) key_next = a[i0 + 64 + lane];
            const int d = (int)((key >> shift) & (kBins - 1));
            const unsigned long long m = digit_peers(d, valid);
            const int rank = __popcll(m & ((1ull << lane) - 1ull));
            if (valid) {
                const int base = whist[cur][wid][d];           // all peers read ...
                if (rank == 0) whist[cur][wid][d] = base + __popcll(m);  // ... before their leader moves the offset
                const int off = base + rank;
                b[off] = key;
                if (pass < 3) atomicAdd(&whist[nxt][off / quarter][(int)((key >> (shift + kDigitBits)) & (kBins - 1))], 1);
            }
        }
        cur = nxt;
        unsigned long long *tmp = a; a = b; b = tmp;
    }
    __syncthreads();
    // ---- ties: keys with identical depth bits are still in scatter order; order each run by Gaussian index ----
    for (int i = tid; i < n; i += 256) {
        const unsigned hi = (unsigned)(a[i] >> 32);
        const bool start = (i == 0 || (unsigned)(a[i - 1] >> 32) != hi) && (i + 1 < n) && ((unsigned)(a[i + 1] >> 32) == hi);
        if (start) {
            int e = i + 1;
            while (e < n && (unsigned)(a[e] >> 32) == hi) ++e;
            for (int p = i + 1; p < e; ++p) {  // insertion sort of [i, e)
                const unsigned long long v = a[p];
                int q = p - 1;
                while (q >= i && a[q] > v) { a[q + 1] = a[q]; --q; }
                a[q + 1] = v;
            }
        }
    }
    __syncthreads();
    for (int k = tid; k < n; k += 256) point_list[rg.x + k] = (uint32_t)a[k];
}

// One wave per published run (<= 256 keys of one tile's bucketized list in `scratch`): register bitonic sort on the full
// (depth, index) key, index list written coalesced.
__global__ void __launch_bounds__(256)
segment_sort_kernel(const int2 *__restrict__ segs, int nslots, const unsigned long long *__restrict__ scratch,
                    uint32_t *__restrict__ point_list) {
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= nslots) return;
    const int2 sg = segs[slot];
    if (sg.y <= 0) return;
    if (sg.y <= 64) {
        unsigned long long v[1] = {lane < sg.y ? scratch[(size_t)sg.x + lane] : ~0ull};
        wave_bitonic<1>(v, lane);
        if (lane < sg.y) point_list[(size_t)sg.x + lane] = (uint32_t)v[0];
        return;
    }
    if (sg.y <= 128) {     // (wave-uniform) nearly every published run: the 128-key network is 2.5x less compare-exchange work
        unsigned long long v[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) v[r] = (r * 64 + lane) < sg.y ? scratch[(size_t)sg.x + r * 64 + lane] : ~0ull;
        wave_bitonic<2>(v, lane);
#pragma unroll
        for (int r = 0; r < 2; ++r)
            if (r * 64 + lane < sg.y) point_list[(size_t)sg.x + r * 64 + lane] = (uint32_t)v[r];
        return;
    }
    unsigned long long v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (r * 64 + lane) < sg.y ? scratch[(size_t)sg.x + r * 64 + lane] : ~0ull;
    wave_bitonic<4>(v, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (r * 64 + lane < sg.y) point_list[(size_t)sg.x + r * 64 + lane] = (uint32_t)v[r];
}

// ---------------------------------------------------------------------------------------------------------
// K5: render.  One workgroup (4 waves) per (camera, 16x16 tile); wave w owns the 8x8 pixel quadrant (w&1, w>>1),
// one pixel per lane.  The tile's sorted list is staged through LDS in batches of 256 records (coalesced id read,
// 3 x 16-byte gathers of the packed record).  The gather is software-pipelined through registers: while a batch is
// blended out of LDS, the records of the next batch and the ids of the one after are already in flight, so the two
// dependent memory round trips of a batch overlap with the previous batch's arithmetic instead of preceding it.
// Culling is per 2x2 pixel block: the 4 lanes of a block walk their OWN survivor list.  Per round of 64 staged entries each
// lane tests one entry's conservative {alpha >= 1/255} footprint against the quadrant's four block columns and four block rows
// (eight compares whose lane masks ARE the ballots; a block's list is column-mask & row-mask); a block then steps through the
// set bits of its mask -- per-lane ctz, LDS reads at up to sixteen different addresses per wave -- so one trip of the loop
// blends up to sixteen different Gaussians and the trip count is the LONGEST block list instead of the union over the 8x8
// quadrant (a 3-4 px footprint reaches ~8 of 64 entries per block against ~34 per quadrant).
// The trip is straight-line predicated code, not nested branches.  (Per-PIXEL lists -- 8 + 8 masks, 64 different LDS addresses per read --
// were measured at 18 ms against 7.0 for the blocks: the broadcast reads turn into bank-conflicted gathers.)  Semantics are those of upstream's per-pixel loop: identical skip / stop thresholds,
// `contributor` counts every list entry, so final_T / n_contrib match the oracle.
// ---------------------------------------------------------------------------------------------------------
// NW = waves per tile.  4 (the product path): a wave per 8x8 quadrant.  1 (round 4, the north star's "one wavefront per tile", built for
// the A/B VERDICT r3 asked for): ONE wave walks the four quadrants of the tile in turn -- four pixels per lane, one per quadrant -- so a
// staged record's footprint test is read from LDS once per tile instead of once per quadrant-wave, the batch barriers are wave-local, and
// a tile occupies one wave slot (12 tiles per CU by LDS instead of 3-4 workgroups of four waves).  Same arithmetic, same order per pixel:
// bit-identical output (tests/test_raster_gpu.py runs both).  Measured: see DESIGN 5.
template <bool COUNT_TOUCHED, int NW, int NTB = 256>
__global__ void __launch_bounds__(64 * NW)
render_kernel(int P, int W, int H, const int2 *__restrict__ ranges, const uint32_t *__restrict__ point_list,
              const float *__restrict__ geom, const float *__restrict__ background, float *__restrict__ out_color,
              float *__restrict__ out_depth, float *__restrict__ out_opacity, float *__restrict__ final_T,
              int32_t *__restrict__ n_contrib, int32_t *__restrict__ n_touched, float *__restrict__ ckpt, int2 *__restrict__ cktab) {
    constexpr int NTHR = 64 * NW, NT = NTB, RPT = NT / NTHR, NQ = 4 / NW;   // staged batch: NT records, RPT per thread; NQ quadrants per wave
    static_assert(vs::kCkSeg % NTB == 0, "checkpoints fall on batch boundaries");
    __shared__ float4 sq0[NT], sq1[NT], sq2[NT];
    __shared__ uint32_t sid[NT];
    const int gx = (W + kTile - 1) / kTile;
    const int tiles = gridDim.x;
    // (camera, tile) of this workgroup from the longest-first launch order behind the ranges (tile_scan_kernel)
    const int t_lin = reinterpret_cast<const int32_t *>(ranges + (size_t)gridDim.x * gridDim.y)[(size_t)blockIdx.y * gridDim.x + blockIdx.x];
    const int c = t_lin / tiles;
    const int tile = t_lin - c * tiles;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tile_x = tile % gx, tile_y = tile / gx;
    const int sb = lane >> 2;                                    // this lane's 2x2 pixel block inside an 8x8 quadrant (4 x 4 blocks)
    const int bxi = sb & 3, byi = sb >> 2;
    int pxi[NQ], pyi[NQ];
    float pixfx[NQ], pixfy[NQ], bcx[NQ], bcy[NQ];
    bool inside[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int quad = NW == 4 ? wid : q;
        const int qx0 = tile_x * kTile + (quad & 1) * 8, qy0 = tile_y * kTile + (quad >> 1) * 8;
        pxi[q] = qx0 + bxi * 2 + (lane & 1); pyi[q] = qy0 + byi * 2 + ((lane >> 1) & 1);
        pixfx[q] = (float)pxi[q]; pixfy[q] = (float)pyi[q];
        bcx[q] = (float)qx0 + 0.5f; bcy[q] = (float)qy0 + 0.5f;  // centre of block column / row 0; 2 px apart, half size 0.5 px
        inside[q] = pxi[q] < W && pyi[q] < H;
    }
    const int2 rg = ranges[(size_t)c * tiles + tile];
    const float4 *__restrict__ g4 = reinterpret_cast<const float4 *>(geom + (size_t)c * P * kGeomFloats);
    const uint32_t *__restrict__ plist = point_list + rg.x;
    const int n = rg.y - rg.x;
    // VS_RASTER_SAVE_FOR_BACKWARD (round 6): the blending state of the tile's 256 pixels is stored every kCkSeg list entries, so that the
    // backward can replay the segments of a list INDEPENDENTLY (render_backward_seg_kernel, raster_bwd.hip) instead of walking the whole
    // list in one wave.  Tile t (linear index) owns the slots (x >> kCkShift) + t .. (y >> kCkShift) + t of the checkpoint table -- the
    // ranges are contiguous, so the slots tile the table exactly, as the sort's run table does.  Slot k of the tile = segment k = entries
    // [k * kCkSeg, (k + 1) * kCkSeg); its state is written when the loop reaches it (a tile whose pixels are all done earlier never
    // writes it: the backward reads the last contributors first and never asks for it).
    const size_t ck_base = (size_t)(rg.x >> vs::kCkShift) + (size_t)t_lin;
    if (cktab) {
        const int slots = (rg.y >> vs::kCkShift) - (rg.x >> vs::kCkShift) + 1;
        for (int k = tid; k < slots; k += NTHR) cktab[ck_base + k] = (k * vs::kCkSeg < n) ? make_int2(t_lin, k) : make_int2(-1, 0);
    }

    float T[NQ], Cr[NQ], Cg[NQ], Cb[NQ], Dd[NQ], thr[NQ];
    int last_contrib[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        T[q] = 1.0f; Cr[q] = Cg[q] = Cb[q] = Dd[q] = 0.f; last_contrib[q] = 0;
        thr[q] = inside[q] ? 1.0f / 255.0f : __builtin_inff();   // alpha threshold: 1/255 while live, +inf once done / outside the image
    }

    // pipeline registers: records (and id) of the batch that is staged next, id of the batch after it
    uint32_t g_cur[RPT], g_nxt[RPT];
    float4 r0[RPT], r1[RPT], r2[RPT];
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const int e = u * NTHR + tid;
        g_cur[u] = e < n ? plist[e] : 0u;
        g_nxt[u] = NT + e < n ? plist[NT + e] : 0u;
    }
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        r0[u] = r1[u] = r2[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (u * NTHR + tid < n) {
            r0[u] = g4[(size_t)g_cur[u] * 3 + 0];
            r1[u] = g4[(size_t)g_cur[u] * 3 + 1];
            r2[u] = g4[(size_t)g_cur[u] * 3 + 2];
        }
    }

    int contributor = 0;
    // Upstream's per-pixel step (skip / stop thresholds, operation order) for staged entry j.
    // One trip of a 2x2 block's lanes over staged entry j, as straight-line predicated code.  The kernel is VALU-bound
    // (SQ_ACTIVE_INST_VALU = its whole duration) and hipcc turns divergent branches around loop-carried accumulators into register
    // copies at every nesting level (70 VALU per trip measured; ~40 this way).
    //   thr   alpha threshold of the pixel: 1/255 while live, +inf once done (saturated / outside the image): "live and
    //         alpha >= 1/255" is one compare;
    //   act   this lane's block has an entry in this trip (alpha forced to 0 otherwise);
    //   the accumulators take w = 0 in lanes where upstream's three conditions do not all hold: x + c*0 == x for the finite
    //   colours / depths preprocess writes, so those lanes are unchanged and the others see upstream's operations in its order.
    auto step = [&](int q, int j, bool act) -> bool {
        const float4 q0 = sq0[j];
        const float4 q1 = sq1[j];
        const float4 q2 = sq2[j];
        const float dx = q0.x - pixfx[q], dy = q0.y - pixfy[q];
        const float power = -0.5f * (q1.x * dx * dx + q1.z * dy * dy) - q1.y * dx * dy;
        float alpha = fminf(0.99f, q1.w * __expf(fminf(power, 0.0f)));
        alpha = (act && power <= 0.0f) ? alpha : 0.0f;   // upstream skips power > 0 (a NaN power compares false here too)
        const bool pass = alpha >= thr[q];
        const float test_T = T[q] * (1.0f - alpha);
        const bool go = pass && !(test_T < 0.0001f);
        thr[q] = (pass && !go) ? __builtin_inff() : thr[q];
        const float w = go ? alpha * T[q] : 0.0f;
        Cr[q] += q2.x * w; Cg[q] += q2.y * w; Cb[q] += q2.z * w;
        Dd[q] += q2.w * w;
        T[q] = go ? test_T : T[q];
        last_contrib[q] = go ? contributor + j + 1 : last_contrib[q];
        return go && test_T > 0.5f;
    };
    auto lane_done = [&]() -> bool {
        bool d = true;
#pragma unroll
        for (int q = 0; q < NQ; ++q) d = d && thr[q] > 1.0f;
        return d;
    };

    for (int base = 0; base < n; base += NT) {
        // (also orders the previous batch's LDS reads before this batch's stores)
        if (__syncthreads_count(lane_done()) == NTHR) break;
        if (ckpt && base > 0 && (base & (vs::kCkSeg - 1)) == 0) {   // state in front of entry `base`: [T | Cr | Cg | Cb | D][block (by * 8 + bx)][pixel of the 2x2 block]
            float *ck = ckpt + (ck_base + (size_t)(base >> vs::kCkShift)) * vs::kCkFloats;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int quad = NW == 4 ? wid : q;
                const int idx = ((((quad >> 1) * 4 + byi) * 8) + (quad & 1) * 4 + bxi) * 4 + (lane & 3);
                ck[idx] = T[q]; ck[256 + idx] = Cr[q]; ck[512 + idx] = Cg[q]; ck[768 + idx] = Cb[q]; ck[1024 + idx] = Dd[q];
            }
        }
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            const int e = u * NTHR + tid;
            if (base + e < n) {
                sq0[e] = r0[u]; sq1[e] = r1[u]; sq2[e] = r2[u];
                if (COUNT_TOUCHED) sid[e] = g_cur[u];
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            const int e = u * NTHR + tid;
            g_cur[u] = g_nxt[u];
            if (base + NT + e < n) {
                r0[u] = g4[(size_t)g_cur[u] * 3 + 0];
                r1[u] = g4[(size_t)g_cur[u] * 3 + 1];
                r2[u] = g4[(size_t)g_cur[u] * 3 + 2];
            }
            if (base + 2 * NT + e < n) g_nxt[u] = plist[base + 2 * NT + e];
        }
        const int cnt = min(NT, n - base);
        if (!__all(lane_done())) {
            for (int j0 = 0; j0 < cnt; j0 += 64) {
                const int je = j0 + lane;
                float4 t = make_float4(0.f, 0.f, -1.5f, -1.5f);   // (extent -1: matches no block)
                if (je < cnt) t = sq0[je];                          // read ONCE for all the quadrants of this wave
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    if (NQ > 1 && __all(thr[q] > 1.0f)) continue;  // this quadrant is finished
                    // entry je against the four block columns and the four block rows: a block's mask is column-mask & row-mask
                    const float tx = t.x - bcx[q], ty = t.y - bcy[q], ex = t.z + 0.5f, ey = t.w + 0.5f;
                    const unsigned long long mx0 = __ballot(fabsf(tx) <= ex), mx1 = __ballot(fabsf(tx - 2.0f) <= ex),
                                             mx2 = __ballot(fabsf(tx - 4.0f) <= ex), mx3 = __ballot(fabsf(tx - 6.0f) <= ex);
                    const unsigned long long my0 = __ballot(fabsf(ty) <= ey), my1 = __ballot(fabsf(ty - 2.0f) <= ey),
                                             my2 = __ballot(fabsf(ty - 4.0f) <= ey), my3 = __ballot(fabsf(ty - 6.0f) <= ey);
                    unsigned long long mine = ((bxi & 2) ? ((bxi & 1) ? mx3 : mx2) : ((bxi & 1) ? mx1 : mx0)) &
                                              ((byi & 2) ? ((byi & 1) ? my3 : my2) : ((byi & 1) ? my1 : my0));
                    while (__any(mine != 0ull)) {
                        const bool act = mine != 0ull;
                        const int j = j0 + (act ? __builtin_ctzll(mine) : 0);
                        mine &= mine - 1ull;
                        const bool touched = step(q, j, act);
                        if (COUNT_TOUCHED) {   // the sixteen blocks blend different entries: one count per block
                            const int tot = __popcll(__ballot(touched) & (0xFull << (sb * 4)));
                            if ((lane & 3) == 0 && tot > 0) atomicAdd(&n_touched[(size_t)c * P + sid[j]], tot);
                        }
                    }
                }
                if (__all(lane_done())) break;
            }
        }
        contributor += cnt;
    }

#pragma unroll
    for (int q = 0; q < NQ; ++q)
        if (inside[q]) {
            const float bgr = background[3 * c], bgg = background[3 * c + 1], bgb = background[3 * c + 2];
            const size_t HW = (size_t)H * W;
            const size_t pix = (size_t)pyi[q] * W + pxi[q];
            final_T[c * HW + pix] = T[q];
            n_contrib[c * HW + pix] = last_contrib[q];
            out_color[(c * 3 + 0) * HW + pix] = Cr[q] + T[q] * bgr;
            out_color[(c * 3 + 1) * HW + pix] = Cg[q] + T[q] * bgg;
            out_color[(c * 3 + 2) * HW + pix] = Cb[q] + T[q] * bgb;
            out_depth[c * HW + pix] = Dd[q];
            out_opacity[c * HW + pix] = 1.0f - T[q];
        }
}

}  // namespace

extern "C" int64_t vs_raster_forward(const VsRasterIn *in, VsRasterOut *out, VsAllocFn alloc, void *actx, vs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK(in && out && alloc, "vs_raster_forward: null argument");
    VS_CHECK(in->num_cameras > 0 && in->num_scenes > 0 && in->P >= 0, "vs_raster_forward: bad sizes C=%d S=%d P=%d",
             in->num_cameras, in->num_scenes, in->P);
    VS_CHECK(in->width > 0 && in->height > 0, "vs_raster_forward: bad image size %dx%d", in->width, in->height);
    VS_CHECK(in->P == 0 || (in->shs != nullptr) != (in->colors_precomp != nullptr),
             "vs_raster_forward: exactly one of shs / colors_precomp must be given");
    if (in->shs) {
        const int need = in->sh_degree >= 3 ? 16 : (in->sh_degree + 1) * (in->sh_degree + 1);
        VS_CHECK(in->sh_coeffs >= need, "vs_raster_forward: sh_coeffs=%d too small for sh_degree=%d", in->sh_coeffs, in->sh_degree);
    }
    VS_CHECK(in->P == 0 || (in->means3D && in->cov3D && in->opacities), "vs_raster_forward: null Gaussian pointer");
    VS_CHECK(in->viewmatrix && in->projmatrix && in->campos && in->tanfov && in->background,
             "vs_raster_forward: null camera pointer");
    VS_CHECK(out->color && out->depth && out->opacity && (out->radii || in->P == 0), "vs_raster_forward: null output pointer");
    const int C = in->num_cameras, P = in->P, W = in->width, H = in->height;
    const int gx = vs::cdiv(W, kTile), gy = vs::cdiv(H, kTile), tiles = gx * gy;
    VS_CHECK(gx <= 65535 && gy <= 65535, "vs_raster_forward: image too large");
    const size_t CP = (size_t)C * P;

    for (int k = 0; k < VS_BUF_COUNT; ++k) out->buffers[k] = nullptr;
    auto get = [&](int tag, size_t bytes) -> void * {
        void *p = alloc(actx, tag, bytes ? bytes : 16);
        out->buffers[tag] = p;
        return p;
    };
    float *geom = (float *)get(VS_BUF_GEOM, CP * kGeomFloats * sizeof(float));
    ushort4 *rect = (ushort4 *)get(VS_BUF_RECT, CP * sizeof(ushort4));
    uint8_t *clamped = (uint8_t *)get(VS_BUF_CLAMPED, CP);
    float *depthkey = (float *)get(VS_BUF_DEPTH, CP * sizeof(float));
    int2 *ranges = (int2 *)get(VS_BUF_TILE_RANGES, (size_t)C * tiles * (sizeof(int2) + sizeof(int32_t)));   // ranges, then the launch order (tile_scan_kernel)
    int32_t *cursor = (int32_t *)get(VS_BUF_TILE_CURSOR, (size_t)C * tiles * sizeof(int32_t));
    long long *misc = (long long *)get(VS_BUF_MISC, 4 * sizeof(long long));
    float *final_T = (float *)get(VS_BUF_FINAL_T, (size_t)C * H * W * sizeof(float));
    int32_t *n_contrib = (int32_t *)get(VS_BUF_N_CONTRIB, (size_t)C * H * W * sizeof(int32_t));
    VS_CHECK(geom && rect && clamped && depthkey && ranges && cursor && misc && final_T && n_contrib, "vs_raster_forward: allocator returned null");

    VS_HIP(hipMemsetAsync(cursor, 0, (size_t)C * tiles * sizeof(int32_t), stream));
    VS_HIP(hipMemsetAsync(misc, 0, 4 * sizeof(long long), stream));
    if (out->n_touched) VS_HIP(hipMemsetAsync(out->n_touched, 0, CP * sizeof(int32_t), stream));
    if (P > 0) {
        dim3 grid(vs::cdiv(P, 256), in->num_scenes);
        hipLaunchKernelGGL(preprocess_kernel, grid, dim3(256), tiles <= kHistTiles ? (size_t)2 * tiles * sizeof(int) : 0, stream, *in, geom, rect, clamped, out->radii, depthkey, cursor);
    }
    VS_CHECK(in->capacity >= 0 && in->capacity < 2147483647LL, "vs_raster_forward: capacity %lld out of range", (long long)in->capacity);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, stream, cursor, ranges, C * tiles, misc, (long long)in->capacity);
    long long R, max_tile;
    if (in->capacity > 0) {
        // capacity mode: nothing comes back to the host; every buffer is sized for `capacity` instances and the large-tile sort
        // scratch is always provided (the largest tile is not known here)
        R = in->capacity;
        max_tile = R;
    } else {
        long long host_misc[2] = {0, 0};
        VS_HIP(hipMemcpyAsync(host_misc, misc, sizeof(host_misc), hipMemcpyDeviceToHost, stream));
        VS_HIP(hipStreamSynchronize(stream));
        R = host_misc[0];
        max_tile = host_misc[1];
        VS_CHECK(R >= 0 && R < 2147483647LL, "vs_raster_forward: %lld (Gaussian,tile) instances overflow int32 ranges", R);
    }
    out->num_rendered = R;

    unsigned long long *keys = (unsigned long long *)get(VS_BUF_KEYS, (size_t)R * 8);
    uint32_t *point_list = (uint32_t *)get(VS_BUF_POINT_LIST, (size_t)R * 4);
    unsigned long long *scratch = nullptr;
    // lists longer than one wave's register sort: [R] u64 bucketized keys followed by the run table: tile t owns slots
    // (x_t >> kSegShift) + t .. (y_t >> kSegShift) + t, which tile the table exactly ((R >> kSegShift) + tiles * C slots of {start, length})
    // because the ranges are contiguous
    const long long nslots = (R >> kSegShift) + (long long)tiles * C;
    int2 *segs = nullptr;
    if (max_tile > kSortWave) {
        scratch = (unsigned long long *)get(VS_BUF_SORT_SCRATCH, (size_t)R * 8 + (size_t)nslots * sizeof(int2));
        segs = scratch ? reinterpret_cast<int2 *>(scratch + R) : nullptr;
    }
    VS_CHECK(keys && point_list && (max_tile <= kSortWave || scratch), "vs_raster_forward: allocator returned null");
    // capacity mode: the table is sized for `capacity` instances but tile_sort_kernel only clears the slots of the instances that exist
    if (segs && in->capacity > 0) VS_HIP(hipMemsetAsync(segs, 0, (size_t)nslots * sizeof(int2), stream));
    if (R > 0) {
        dim3 grid(vs::cdiv(P, 256 * kScatterPer), C);
        hipLaunchKernelGGL(scatter_kernel, grid, dim3(256), tiles <= kHistTiles ? (size_t)2 * tiles * sizeof(int) : 0, stream, P, tiles, gx, depthkey, rect, ranges, cursor, keys, misc);
        hipLaunchKernelGGL(tile_sort_kernel, dim3(tiles, C), dim3(256), 0, stream, ranges, keys, point_list, scratch, segs);
        if (segs)
            hipLaunchKernelGGL(segment_sort_kernel, dim3((unsigned)vs::cdiv64(nslots, 4)), dim3(256), 0, stream, segs, (int)nslots,
                               scratch, point_list);
    }
    const bool count = (in->flags & VS_RASTER_COUNT_TOUCHED) && out->n_touched;
    // blending checkpoints for the segment-parallel backward: [slots][5][256] f32 followed by the slot table [slots] {tile, segment}
    float *ckpt = nullptr;
    int2 *cktab = nullptr;
    if ((in->flags & VS_RASTER_SAVE_FOR_BACKWARD) && R > 0) {
        const size_t ck_slots = (size_t)(R >> vs::kCkShift) + (size_t)tiles * C;
        ckpt = (float *)get(VS_BUF_CHECKPOINT, ck_slots * (vs::kCkFloats * sizeof(float) + sizeof(int2)));
        VS_CHECK(ckpt, "vs_raster_forward: allocator returned null");
        cktab = reinterpret_cast<int2 *>(ckpt + ck_slots * vs::kCkFloats);
        VS_HIP(hipMemsetAsync(cktab, 0xFF, ck_slots * sizeof(int2), stream));   // slots no tile owns (capacity mode) stay {-1, -1}
    }
    dim3 rgrid(tiles, C);
    // One wave per tile when there are enough tiles to fill the chip with single waves (>= 16 per CU: the batched bench / training calls),
    // four waves per tile (a quadrant each) for small calls, where a tile's latency matters more than wave slots.  Same results either
    // way (bit-identical); VS_RENDER_WAVES=1 | 4 forces one (A/B: DESIGN 5 -- 5.14 vs 5.20 ms on the 288-view bench step).
    static const int force_waves = [] { const char *e = getenv("VS_RENDER_WAVES"); return e ? atoi(e) : 0; }();
    const int render_waves = force_waves ? force_waves : ((long long)tiles * C >= 4096 ? 1 : 4);
#define VS_RENDER(CNT_, NW_)                                                                                                         \
    hipLaunchKernelGGL((render_kernel<CNT_, NW_>), rgrid, dim3(64 * NW_), 0, stream, P, W, H, ranges, point_list, geom, in->background, \
                       out->color, out->depth, out->opacity, final_T, n_contrib, out->n_touched, ckpt, cktab)
    // one wave per tile: staged batches of 64 records (round 5: 96 VGPRs and 3.3 KiB of LDS per wave -> five waves per SIMD instead of
    // three; the kernel is latency bound -- PMC: VALU issue 0.42, 38 % of the wave cycles waiting -- and a batch is one round of the
    // 64-entry footprint test anyway.  256 / 128 / 64: 5.23 / 4.73 / 4.60 ms on the bench step, bit-identical; VS_RENDER_NT = 256 | 128 for A/B)
    static const int ntb = [] { const char *e = getenv("VS_RENDER_NT"); return e ? atoi(e) : 64; }();
#define VS_RENDER1(CNT_, NT_)                                                                                                         \
    hipLaunchKernelGGL((render_kernel<CNT_, 1, NT_>), rgrid, dim3(64), 0, stream, P, W, H, ranges, point_list, geom, in->background,   \
                       out->color, out->depth, out->opacity, final_T, n_contrib, out->n_touched, ckpt, cktab)
    // (round 6, measured and not kept: lane = one 2x2 pixel block of the tile -- the mapping of the backward's replay kernels, a trip serving 64
    // (block, entry) pairs instead of 16, record reads and mask bookkeeping paid once per four pixels: 4.45 vs 4.35 ms per 288 views, identical
    // images.  The forward's trips are not what bounds it.)
    if (render_waves == 1) {
        if (ntb == 256) { if (count) VS_RENDER1(true, 256); else VS_RENDER1(false, 256); }
        else if (ntb == 128) { if (count) VS_RENDER1(true, 128); else VS_RENDER1(false, 128); }
        else { if (count) VS_RENDER1(true, 64); else VS_RENDER1(false, 64); }
    }
#undef VS_RENDER1
    else { if (count) VS_RENDER(true, 4); else VS_RENDER(false, 4); }
#undef VS_RENDER
    VS_HIP(hipGetLastError());
    return R;
}
