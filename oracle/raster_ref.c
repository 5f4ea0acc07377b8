/*
 * oracle/raster_ref.c -- CPU restatement of the tile-based 3D Gaussian splatting rasterizer
 * (forward + backward, with camera-pose gradients) that VicaSplat calls at
 * /root/reference/src/model/decoder/cuda_splatting.py:207-235.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product path (vicasplat_amd/) never links it.
 *
 * PARITY UNPINNED: the rasterizer is the un-vendored, un-pinned pip dependency
 *   git+https://github.com/rmurai0610/diff-gaussian-rasterization-w-pose.git
 * (/root/reference/requirements.txt:17); no source, test or golden image for it exists under
 * /root/reference.  This file restates the published algorithm (upstream
 * cuda_rasterizer/{forward,backward,rasterizer_impl}.cu, auxiliary.h) and is pinned only by our own
 * known-answer tests, finite differences and a PyTorch-autograd cross-check (tests/test_raster_oracle.py).
 *
 * Arithmetic contract shared with the HIP kernels (vicasplat_amd/csrc/raster_fwd.hip): float32,
 * no FMA contraction (-ffp-contract=off), the exact operation order written below for everything
 * that feeds an integer decision (cull, radius, tile rectangle, depth key).  Under that contract
 * radii / tiles_touched / tile ranges / sorted ids are bit-identical between this file and the GPU.
 *
 * Matrix convention: viewmatrix / projmatrix are 16 floats stored so that
 * element [4*c + r] is M_math[r][c] (column-major), i.e. what cuda_splatting.py:191-194 produces.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TILE 16
#define NEAR_CULL 0.2f

static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

typedef struct {
    int P;            /* Gaussians */
    int D;            /* active SH degree as passed by the caller (4 for VicaSplat); bands > 3 are ignored */
    int M;            /* SH coefficients per colour channel in memory (25) */
    int W, H;
    float tanfovx, tanfovy;
    float scale_modifier; /* unused with cov3D_precomp; kept for signature parity */
    const float *bg;        /* [3] */
    const float *means3D;   /* [P,3] */
    const float *cov3D;     /* [P,6] xx xy xz yy yz zz */
    const float *shs;       /* [P,M,3] or NULL */
    const float *colors_precomp; /* [P,3] or NULL */
    const float *opacities; /* [P] */
    const float *viewmatrix; /* [16] */
    const float *projmatrix; /* [16] */
    const float *campos;     /* [3] */
} RefRasterIn;

/* float -> int exactly as the GPU does it: clamp first so out-of-range / NaN values are defined. */
static inline int f2i(float x) {
    x = fminf(fmaxf(x, -1.0e6f), 1.0e6f);
    return (int)x;
}
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

static inline uint32_t float_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* ---- B.1 per-Gaussian preprocess ------------------------------------------------------------- */
/* Outputs (all [P] unless noted): depths, radii(int), xy[P,2], conic_opacity[P,4], rgb[P,3],
 * clamped[P,3] (uint8), rect[P,4] (min.x,min.y,max.x,max.y), tiles_touched(int). Returns sum of tiles_touched (R). */
static void eval_sh(int deg, const float *sh /* [M,3] */, float dx, float dy, float dz, float *out3, uint8_t *clamped3) {
    float len = sqrtf(dx * dx + dy * dy + dz * dz);
    float x = dx / len, y = dy / len, z = dz / len;
    for (int c = 0; c < 3; ++c) {
        float r = SH_C0 * sh[0 * 3 + c];
        if (deg > 0) {
            r = r - SH_C1 * y * sh[1 * 3 + c] + SH_C1 * z * sh[2 * 3 + c] - SH_C1 * x * sh[3 * 3 + c];
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + SH_C2[0] * xy * sh[4 * 3 + c] + SH_C2[1] * yz * sh[5 * 3 + c] +
                    SH_C2[2] * (2.0f * zz - xx - yy) * sh[6 * 3 + c] + SH_C2[3] * xz * sh[7 * 3 + c] +
                    SH_C2[4] * (xx - yy) * sh[8 * 3 + c];
                if (deg > 2) {
                    r = r + SH_C3[0] * y * (3.0f * xx - yy) * sh[9 * 3 + c] + SH_C3[1] * xy * z * sh[10 * 3 + c] +
                        SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[11 * 3 + c] +
                        SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[12 * 3 + c] +
                        SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[13 * 3 + c] +
                        SH_C3[5] * z * (xx - yy) * sh[14 * 3 + c] + SH_C3[6] * x * (xx - 3.0f * yy) * sh[15 * 3 + c];
                }
            }
        }
        r += 0.5f;
        clamped3[c] = (uint8_t)(r < 0.0f);
        out3[c] = fmaxf(r, 0.0f);
    }
}

long ref_preprocess(const RefRasterIn *in, float *depths, int *radii, float *xy, float *conic_opacity, float *rgb,
                    uint8_t *clamped, int *rect, int *tiles_touched) {
    const int P = in->P, W = in->W, H = in->H;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const float *vm = in->viewmatrix, *pm = in->projmatrix;
    const float focal_x = (float)W / (2.0f * in->tanfovx), focal_y = (float)H / (2.0f * in->tanfovy);
    long R = 0;
    for (int i = 0; i < P; ++i) {
        radii[i] = 0; tiles_touched[i] = 0; depths[i] = 0.f;
        xy[2 * i] = xy[2 * i + 1] = 0.f;
        for (int k = 0; k < 4; ++k) { conic_opacity[4 * i + k] = 0.f; rect[4 * i + k] = 0; }
        for (int k = 0; k < 3; ++k) { rgb[3 * i + k] = 0.f; clamped[3 * i + k] = 0; }
        const float px = in->means3D[3 * i], py = in->means3D[3 * i + 1], pz = in->means3D[3 * i + 2];
        /* view-space point (transformPoint4x3) */
        const float vx = vm[0] * px + vm[4] * py + vm[8] * pz + vm[12];
        const float vy = vm[1] * px + vm[5] * py + vm[9] * pz + vm[13];
        const float vz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
        if (!(vz > NEAR_CULL)) continue; /* cull p_view.z <= 0.2 (also NaN) */
        /* clip-space point (transformPoint4x4) */
        const float hx = pm[0] * px + pm[4] * py + pm[8] * pz + pm[12];
        const float hy = pm[1] * px + pm[5] * py + pm[9] * pz + pm[13];
        const float hw = pm[3] * px + pm[7] * py + pm[11] * pz + pm[15];
        const float p_w = 1.0f / (hw + 0.0000001f);
        const float projx = hx * p_w, projy = hy * p_w;
        /* EWA 2-D covariance */
        const float limx = 1.3f * in->tanfovx, limy = 1.3f * in->tanfovy;
        const float txtz = vx / vz, tytz = vy / vz;
        const float tx = fminf(limx, fmaxf(-limx, txtz)) * vz;
        const float ty = fminf(limy, fmaxf(-limy, tytz)) * vz;
        const float tz = vz;
        const float J00 = focal_x / tz, J02 = -(focal_x * tx) / (tz * tz);
        const float J11 = focal_y / tz, J12 = -(focal_y * ty) / (tz * tz);
        float M0[3], M1[3];
        for (int c = 0; c < 3; ++c) { /* Rv[r][c] = vm[4*c + r] */
            M0[c] = J00 * vm[4 * c + 0] + J02 * vm[4 * c + 2];
            M1[c] = J11 * vm[4 * c + 1] + J12 * vm[4 * c + 2];
        }
        const float *cv = in->cov3D + 6 * i;
        const float S[3][3] = {{cv[0], cv[1], cv[2]}, {cv[1], cv[3], cv[4]}, {cv[2], cv[4], cv[5]}};
        float t0[3], t1[3];
        for (int k = 0; k < 3; ++k) {
            t0[k] = S[k][0] * M0[0] + S[k][1] * M0[1] + S[k][2] * M0[2];
            t1[k] = S[k][0] * M1[0] + S[k][1] * M1[1] + S[k][2] * M1[2];
        }
        const float a = M0[0] * t0[0] + M0[1] * t0[1] + M0[2] * t0[2] + 0.3f;
        const float b = M0[0] * t1[0] + M0[1] * t1[1] + M0[2] * t1[2];
        const float c = M1[0] * t1[0] + M1[1] * t1[1] + M1[2] * t1[2] + 0.3f;
        const float det = a * c - b * b;
        if (det == 0.0f) continue;
        const float det_inv = 1.0f / det;
        const float conx = c * det_inv, cony = -b * det_inv, conz = a * det_inv;
        const float mid = 0.5f * (a + c);
        const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
        const float lambda1 = mid + sq, lambda2 = mid - sq;
        const float my_radius = ceilf(3.0f * sqrtf(fmaxf(lambda1, lambda2)));
        const float pixx = ((projx + 1.0f) * (float)W - 1.0f) * 0.5f;
        const float pixy = ((projy + 1.0f) * (float)H - 1.0f) * 0.5f;
        const int rminx = imin(gx, imax(0, f2i((pixx - my_radius) / (float)TILE)));
        const int rminy = imin(gy, imax(0, f2i((pixy - my_radius) / (float)TILE)));
        const int rmaxx = imin(gx, imax(0, f2i((pixx + my_radius + (float)(TILE - 1)) / (float)TILE)));
        const int rmaxy = imin(gy, imax(0, f2i((pixy + my_radius + (float)(TILE - 1)) / (float)TILE)));
        const int area = (rmaxx - rminx) * (rmaxy - rminy);
        if (area <= 0) continue;
        if (in->colors_precomp) {
            for (int k = 0; k < 3; ++k) rgb[3 * i + k] = in->colors_precomp[3 * i + k];
        } else {
            eval_sh(in->D, in->shs + (size_t)i * in->M * 3, px - in->campos[0], py - in->campos[1],
                    pz - in->campos[2], rgb + 3 * i, clamped + 3 * i);
        }
        depths[i] = vz;
        radii[i] = f2i(my_radius);
        xy[2 * i] = pixx; xy[2 * i + 1] = pixy;
        conic_opacity[4 * i + 0] = conx; conic_opacity[4 * i + 1] = cony; conic_opacity[4 * i + 2] = conz;
        conic_opacity[4 * i + 3] = in->opacities[i];
        rect[4 * i + 0] = rminx; rect[4 * i + 1] = rminy; rect[4 * i + 2] = rmaxx; rect[4 * i + 3] = rmaxy;
        tiles_touched[i] = area;
        R += area;
    }
    return R;
}

/* ---- B.2 binning: (tile | depth) keys, stable sort, tile ranges ------------------------------- */
typedef struct { uint64_t key; uint32_t id; } KV;
static int kv_cmp(const void *a, const void *b) {
    const KV *x = (const KV *)a, *y = (const KV *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0); /* stable: emission order == Gaussian index order */
}
/* point_list[R] (sorted Gaussian ids), ranges[2*tiles] */
int ref_bin(int P, int W, int H, const float *depths, const int *rect, const int *tiles_touched, long R,
            uint32_t *point_list, uint64_t *point_keys /* may be NULL */, int *ranges) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    KV *kv = (KV *)malloc(sizeof(KV) * (size_t)(R > 0 ? R : 1));
    if (!kv) return -1;
    long off = 0;
    for (int i = 0; i < P; ++i) {
        if (tiles_touched[i] == 0) continue;
        for (int y = rect[4 * i + 1]; y < rect[4 * i + 3]; ++y)
            for (int x = rect[4 * i + 0]; x < rect[4 * i + 2]; ++x) {
                uint64_t key = (uint64_t)(y * gx + x);
                key <<= 32;
                key |= (uint64_t)float_bits(depths[i]);
                kv[off].key = key; kv[off].id = (uint32_t)i; ++off;
            }
    }
    if (off != R) { free(kv); return -2; }
    qsort(kv, (size_t)R, sizeof(KV), kv_cmp);
    for (int t = 0; t < gx * gy; ++t) ranges[2 * t] = ranges[2 * t + 1] = 0;
    for (long r = 0; r < R; ++r) {
        point_list[r] = kv[r].id;
        if (point_keys) point_keys[r] = kv[r].key;
        int tile = (int)(kv[r].key >> 32);
        if (r == 0) ranges[2 * tile] = 0;
        else {
            int prev = (int)(kv[r - 1].key >> 32);
            if (prev != tile) { ranges[2 * prev + 1] = (int)r; ranges[2 * tile] = (int)r; }
        }
        if (r == R - 1) ranges[2 * tile + 1] = (int)R;
    }
    free(kv);
    return 0;
}

/* ---- B.3 render ------------------------------------------------------------------------------- */
void ref_render(int W, int H, const float *bg, const int *ranges, const uint32_t *point_list, const float *xy,
                const float *conic_opacity, const float *rgb, const float *depths, float *out_color /*[3,H,W]*/,
                float *out_depth /*[H,W]*/, float *out_opacity /*[H,W]*/, float *final_T, int *n_contrib,
                int *n_touched /*[P], pre-zeroed by caller*/) {
    const int gx = (W + TILE - 1) / TILE;
    for (int py = 0; py < H; ++py)
        for (int px = 0; px < W; ++px) {
            const int tile = (py / TILE) * gx + (px / TILE);
            const int r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            const float pixfx = (float)px, pixfy = (float)py;
            float T = 1.0f, C[3] = {0, 0, 0}, Dd = 0.0f;
            int contributor = 0, last_contributor = 0;
            for (int r = r0; r < r1; ++r) {
                ++contributor;
                const uint32_t g = point_list[r];
                const float dx = xy[2 * g] - pixfx, dy = xy[2 * g + 1] - pixfy;
                const float *co = conic_opacity + 4 * g;
                const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0.0f) continue;
                const float alpha = fminf(0.99f, co[3] * expf(power));
                if (alpha < 1.0f / 255.0f) continue;
                const float test_T = T * (1.0f - alpha);
                if (test_T < 0.0001f) break; /* this Gaussian is NOT composited */
                const float w = alpha * T;
                for (int ch = 0; ch < 3; ++ch) C[ch] += rgb[3 * g + ch] * w;
                Dd += depths[g] * w;
                if (test_T > 0.5f) n_touched[g] += 1;
                T = test_T;
                last_contributor = contributor;
            }
            const int pix = py * W + px;
            final_T[pix] = T; n_contrib[pix] = last_contributor;
            for (int ch = 0; ch < 3; ++ch) out_color[ch * H * W + pix] = C[ch] + T * bg[ch];
            out_depth[pix] = Dd;
            out_opacity[pix] = 1.0f - T;
        }
}

/* ---- B.5 backward: render ---------------------------------------------------------------------- */
/* Accumulates (+=) into dL_dmean2D[P,2] (gradient w.r.t. NDC coordinates, i.e. already times 0.5*W / 0.5*H as
 * upstream), dL_dconic[P,3] (true partials w.r.t. A,B,C), dL_dopacity[P], dL_dcolors[P,3], dL_ddepths[P].
 * Double accumulators are used per pixel loop order; sums are float adds in pixel-major order. */
void ref_render_backward(int W, int H, const float *bg, const int *ranges, const uint32_t *point_list, const float *xy,
                         const float *conic_opacity, const float *rgb, const float *depths, const float *final_T,
                         const int *n_contrib, const float *dL_dpix /*[3,H,W]*/, const float *dL_dpixdepth /*[H,W] or NULL*/,
                         float *dL_dmean2D, float *dL_dconic, float *dL_dopacity, float *dL_dcolors, float *dL_ddepths) {
    const int gx = (W + TILE - 1) / TILE;
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
    for (int py = 0; py < H; ++py)
        for (int px = 0; px < W; ++px) {
            const int pix = py * W + px;
            const int tile = (py / TILE) * gx + (px / TILE);
            const int r0 = ranges[2 * tile];
            const int last = n_contrib[pix];
            const float T_final = final_T[pix];
            float T = T_final;
            float dLp[3] = {dL_dpix[0 * H * W + pix], dL_dpix[1 * H * W + pix], dL_dpix[2 * H * W + pix]};
            const float dLd = dL_dpixdepth ? dL_dpixdepth[pix] : 0.0f;
            float accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0}, accum_rec_d = 0.f, last_depth = 0.f, last_alpha = 0.f;
            const float bg_dot = bg[0] * dLp[0] + bg[1] * dLp[1] + bg[2] * dLp[2];
            for (int r = r0 + last - 1; r >= r0; --r) {
                const uint32_t g = point_list[r];
                const float dx = xy[2 * g] - (float)px, dy = xy[2 * g + 1] - (float)py;
                const float *co = conic_opacity + 4 * g;
                const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0.0f) continue;
                const float G = expf(power);
                const float alpha = fminf(0.99f, co[3] * G);
                if (alpha < 1.0f / 255.0f) continue;
                T = T / (1.0f - alpha);
                const float dch = alpha * T;
                float dL_dalpha = 0.0f;
                for (int ch = 0; ch < 3; ++ch) {
                    const float c = rgb[3 * g + ch];
                    accum_rec[ch] = last_alpha * last_color[ch] + (1.0f - last_alpha) * accum_rec[ch];
                    last_color[ch] = c;
                    dL_dalpha += (c - accum_rec[ch]) * dLp[ch];
                    dL_dcolors[3 * g + ch] += dch * dLp[ch];
                }
                const float cd = depths[g];
                accum_rec_d = last_alpha * last_depth + (1.0f - last_alpha) * accum_rec_d;
                last_depth = cd;
                dL_dalpha += (cd - accum_rec_d) * dLd;
                dL_ddepths[g] += dch * dLd;
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final / (1.0f - alpha)) * bg_dot;
                /* alpha = min(0.99, o*G): upstream's backward ignores the clamp (straight-through); so do we. */
                const float dL_dG = co[3] * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * co[0] - gdy * co[1];
                const float dG_ddely = -gdy * co[2] - gdx * co[1];
                dL_dmean2D[2 * g + 0] += dL_dG * dG_ddelx * ddelx_dx;
                dL_dmean2D[2 * g + 1] += dL_dG * dG_ddely * ddely_dy;
                dL_dconic[3 * g + 0] += -0.5f * gdx * dx * dL_dG;
                dL_dconic[3 * g + 1] += -1.0f * gdx * dy * dL_dG;
                dL_dconic[3 * g + 2] += -0.5f * gdy * dy * dL_dG;
                dL_dopacity[g] += G * dL_dalpha;
            }
        }
}

/* ---- B.5 backward: preprocess ------------------------------------------------------------------ */
/* Inputs: the forward inputs, radii (to skip culled), clamped, and the per-Gaussian grads from render backward.
 * Outputs (overwritten): dL_dmeans3D[P,3], dL_dcov3D[P,6], dL_dsh[P,M,3] (bands 0-3; zero elsewhere),
 * dL_dcolors_precomp handled by caller (== dL_dcolors), dL_dtau[6] = (rho[3], theta[3]) summed over P for the
 * left perturbation T_cw' = Exp(tau) T_cw, with viewmatrix, projmatrix = P_raw*viewmatrix and campos all derived
 * from T_cw'.  projmatrix_raw[16] (column-major) is needed only for dL_dtau. */
static void sh_backward(int deg, int M, const float *sh, const uint8_t *clamped3, float dxo, float dyo, float dzo,
                        const float *dL_drgb, float *dL_dsh, float *dL_ddir_orig /* out[3] */) {
    float len = sqrtf(dxo * dxo + dyo * dyo + dzo * dzo);
    float x = dxo / len, y = dyo / len, z = dzo / len;
    float g[3];
    for (int c = 0; c < 3; ++c) g[c] = clamped3[c] ? 0.0f : dL_drgb[c];
    float dRdx[3] = {0, 0, 0}, dRdy[3] = {0, 0, 0}, dRdz[3] = {0, 0, 0};
    for (int c = 0; c < 3; ++c) {
        dL_dsh[0 * 3 + c] = SH_C0 * g[c];
        if (deg > 0) {
            dL_dsh[1 * 3 + c] = -SH_C1 * y * g[c];
            dL_dsh[2 * 3 + c] = SH_C1 * z * g[c];
            dL_dsh[3 * 3 + c] = -SH_C1 * x * g[c];
            dRdx[c] = -SH_C1 * sh[3 * 3 + c];
            dRdy[c] = -SH_C1 * sh[1 * 3 + c];
            dRdz[c] = SH_C1 * sh[2 * 3 + c];
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                dL_dsh[4 * 3 + c] = SH_C2[0] * xy * g[c];
                dL_dsh[5 * 3 + c] = SH_C2[1] * yz * g[c];
                dL_dsh[6 * 3 + c] = SH_C2[2] * (2.0f * zz - xx - yy) * g[c];
                dL_dsh[7 * 3 + c] = SH_C2[3] * xz * g[c];
                dL_dsh[8 * 3 + c] = SH_C2[4] * (xx - yy) * g[c];
                dRdx[c] += SH_C2[0] * y * sh[4 * 3 + c] + SH_C2[2] * 2.0f * -x * sh[6 * 3 + c] + SH_C2[3] * z * sh[7 * 3 + c] +
                           SH_C2[4] * 2.0f * x * sh[8 * 3 + c];
                dRdy[c] += SH_C2[0] * x * sh[4 * 3 + c] + SH_C2[1] * z * sh[5 * 3 + c] + SH_C2[2] * 2.0f * -y * sh[6 * 3 + c] +
                           SH_C2[4] * 2.0f * -y * sh[8 * 3 + c];
                dRdz[c] += SH_C2[1] * y * sh[5 * 3 + c] + SH_C2[2] * 2.0f * 2.0f * z * sh[6 * 3 + c] + SH_C2[3] * x * sh[7 * 3 + c];
                if (deg > 2) {
                    dL_dsh[9 * 3 + c] = SH_C3[0] * y * (3.0f * xx - yy) * g[c];
                    dL_dsh[10 * 3 + c] = SH_C3[1] * xy * z * g[c];
                    dL_dsh[11 * 3 + c] = SH_C3[2] * y * (4.0f * zz - xx - yy) * g[c];
                    dL_dsh[12 * 3 + c] = SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * g[c];
                    dL_dsh[13 * 3 + c] = SH_C3[4] * x * (4.0f * zz - xx - yy) * g[c];
                    dL_dsh[14 * 3 + c] = SH_C3[5] * z * (xx - yy) * g[c];
                    dL_dsh[15 * 3 + c] = SH_C3[6] * x * (xx - 3.0f * yy) * g[c];
                    dRdx[c] += SH_C3[0] * sh[9 * 3 + c] * 3.0f * 2.0f * xy + SH_C3[1] * sh[10 * 3 + c] * yz +
                               SH_C3[2] * sh[11 * 3 + c] * -2.0f * xy + SH_C3[3] * sh[12 * 3 + c] * -3.0f * 2.0f * xz +
                               SH_C3[4] * sh[13 * 3 + c] * (-3.0f * xx + 4.0f * zz - yy) + SH_C3[5] * sh[14 * 3 + c] * 2.0f * xz +
                               SH_C3[6] * sh[15 * 3 + c] * 3.0f * (xx - yy);
                    dRdy[c] += SH_C3[0] * sh[9 * 3 + c] * 3.0f * (xx - yy) + SH_C3[1] * sh[10 * 3 + c] * xz +
                               SH_C3[2] * sh[11 * 3 + c] * (-3.0f * yy + 4.0f * zz - xx) + SH_C3[3] * sh[12 * 3 + c] * -3.0f * 2.0f * yz +
                               SH_C3[4] * sh[13 * 3 + c] * -2.0f * xy + SH_C3[5] * sh[14 * 3 + c] * -2.0f * yz +
                               SH_C3[6] * sh[15 * 3 + c] * -3.0f * 2.0f * xy;
                    dRdz[c] += SH_C3[1] * sh[10 * 3 + c] * xy + SH_C3[2] * sh[11 * 3 + c] * 4.0f * 2.0f * yz +
                               SH_C3[3] * sh[12 * 3 + c] * 3.0f * (2.0f * zz - xx - yy) + SH_C3[4] * sh[13 * 3 + c] * 4.0f * 2.0f * xz +
                               SH_C3[5] * sh[14 * 3 + c] * (xx - yy);
                }
            }
        }
    }
    (void)M;
    /* dL/d(normalised dir) */
    float ddx = dRdx[0] * g[0] + dRdx[1] * g[1] + dRdx[2] * g[2];
    float ddy = dRdy[0] * g[0] + dRdy[1] * g[1] + dRdy[2] * g[2];
    float ddz = dRdz[0] * g[0] + dRdz[1] * g[1] + dRdz[2] * g[2];
    /* through normalisation: d(v/|v|)/dv = (I - n n^T)/|v| */
    float dot = x * ddx + y * ddy + z * ddz;
    dL_ddir_orig[0] = (ddx - x * dot) / len;
    dL_ddir_orig[1] = (ddy - y * dot) / len;
    dL_ddir_orig[2] = (ddz - z * dot) / len;
}

void ref_preprocess_backward(const RefRasterIn *in, const float *projmatrix_raw, const int *radii, const uint8_t *clamped,
                             const float *dL_dmean2D, const float *dL_dconic, const float *dL_dcolors,
                             const float *dL_ddepths, float *dL_dmeans3D, float *dL_dcov3D, float *dL_dsh,
                             float *dL_dtau /* [6] or NULL */) {
    const int P = in->P, W = in->W, H = in->H;
    const float *vm = in->viewmatrix, *pm = in->projmatrix;
    const float fx = (float)W / (2.0f * in->tanfovx), fy = (float)H / (2.0f * in->tanfovy);
    double tau[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < P; ++i) {
        for (int k = 0; k < 3; ++k) dL_dmeans3D[3 * i + k] = 0.f;
        for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = 0.f;
        if (dL_dsh) for (int k = 0; k < in->M * 3; ++k) dL_dsh[(size_t)i * in->M * 3 + k] = 0.f;
        if (radii[i] <= 0) continue;
        const float px = in->means3D[3 * i], py = in->means3D[3 * i + 1], pz = in->means3D[3 * i + 2];
        const float vx = vm[0] * px + vm[4] * py + vm[8] * pz + vm[12];
        const float vy = vm[1] * px + vm[5] * py + vm[9] * pz + vm[13];
        const float vz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
        float dL_dpc[3] = {0, 0, 0};   /* gradient w.r.t. the camera-space point p_C = (vx,vy,vz) */
        float dL_dR[3][3] = {{0}};     /* gradient w.r.t. Rv entries */
        float dL_dp[3] = {0, 0, 0};    /* direct gradient w.r.t. world mean not via p_C (SH direction) */
        /* ---- cov2D path ---- */
        {
            const float limx = 1.3f * in->tanfovx, limy = 1.3f * in->tanfovy;
            const float txtz = vx / vz, tytz = vy / vz;
            const float tx = fminf(limx, fmaxf(-limx, txtz)) * vz, ty = fminf(limy, fmaxf(-limy, tytz)) * vz, tz = vz;
            const float xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
            const float ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
            const float J00 = fx / tz, J02 = -(fx * tx) / (tz * tz), J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
            float M0[3], M1[3];
            for (int c = 0; c < 3; ++c) {
                M0[c] = J00 * vm[4 * c + 0] + J02 * vm[4 * c + 2];
                M1[c] = J11 * vm[4 * c + 1] + J12 * vm[4 * c + 2];
            }
            const float *cv = in->cov3D + 6 * i;
            const float S[3][3] = {{cv[0], cv[1], cv[2]}, {cv[1], cv[3], cv[4]}, {cv[2], cv[4], cv[5]}};
            float t0[3], t1[3];
            for (int k = 0; k < 3; ++k) {
                t0[k] = S[k][0] * M0[0] + S[k][1] * M0[1] + S[k][2] * M0[2];
                t1[k] = S[k][0] * M1[0] + S[k][1] * M1[1] + S[k][2] * M1[2];
            }
            const float a = M0[0] * t0[0] + M0[1] * t0[1] + M0[2] * t0[2] + 0.3f;
            const float b = M0[0] * t1[0] + M0[1] * t1[1] + M0[2] * t1[2];
            const float c = M1[0] * t1[0] + M1[1] * t1[1] + M1[2] * t1[2] + 0.3f;
            const float det = a * c - b * b;
            const float d2inv = 1.0f / (det * det + 0.0000001f);
            const float gA = dL_dconic[3 * i], gB = dL_dconic[3 * i + 1], gC = dL_dconic[3 * i + 2];
            float ga = 0.f, gb = 0.f, gc = 0.f;
            if (d2inv != 0.0f) {
                ga = d2inv * (-c * c * gA + b * c * gB - b * b * gC);
                gc = d2inv * (-b * b * gA + a * b * gB - a * a * gC);
                gb = d2inv * (2.0f * b * c * gA - (det + 2.0f * b * b) * gB + 2.0f * a * b * gC);
            }
            float *gS = dL_dcov3D + 6 * i;
            gS[0] = M0[0] * M0[0] * ga + M0[0] * M1[0] * gb + M1[0] * M1[0] * gc;
            gS[3] = M0[1] * M0[1] * ga + M0[1] * M1[1] * gb + M1[1] * M1[1] * gc;
            gS[5] = M0[2] * M0[2] * ga + M0[2] * M1[2] * gb + M1[2] * M1[2] * gc;
            gS[1] = 2.f * M0[0] * M0[1] * ga + (M0[0] * M1[1] + M0[1] * M1[0]) * gb + 2.f * M1[0] * M1[1] * gc;
            gS[2] = 2.f * M0[0] * M0[2] * ga + (M0[0] * M1[2] + M0[2] * M1[0]) * gb + 2.f * M1[0] * M1[2] * gc;
            gS[4] = 2.f * M0[1] * M0[2] * ga + (M0[1] * M1[2] + M0[2] * M1[1]) * gb + 2.f * M1[1] * M1[2] * gc;
            /* dL/dM rows */
            float gM0[3], gM1[3];
            for (int k = 0; k < 3; ++k) {
                gM0[k] = 2.f * ga * t0[k] + gb * t1[k];
                gM1[k] = 2.f * gc * t1[k] + gb * t0[k];
            }
            /* M0[c] = J00*R[0][c] + J02*R[2][c];  M1[c] = J11*R[1][c] + J12*R[2][c] */
            float gJ00 = 0, gJ02 = 0, gJ11 = 0, gJ12 = 0;
            for (int cc = 0; cc < 3; ++cc) {
                gJ00 += gM0[cc] * vm[4 * cc + 0]; gJ02 += gM0[cc] * vm[4 * cc + 2];
                gJ11 += gM1[cc] * vm[4 * cc + 1]; gJ12 += gM1[cc] * vm[4 * cc + 2];
                dL_dR[0][cc] += J00 * gM0[cc];
                dL_dR[1][cc] += J11 * gM1[cc];
                dL_dR[2][cc] += J02 * gM0[cc] + J12 * gM1[cc];
            }
            const float tz2 = 1.0f / (tz * tz), tz3 = tz2 / tz;
            const float gtx = xmul * (-fx * tz2) * gJ02;
            const float gty = ymul * (-fy * tz2) * gJ12;
            const float gtz = -fx * tz2 * gJ00 - fy * tz2 * gJ11 + (2.f * fx * tx) * tz3 * gJ02 + (2.f * fy * ty) * tz3 * gJ12;
            dL_dpc[0] += gtx; dL_dpc[1] += gty; dL_dpc[2] += gtz;
        }
        /* ---- projected mean path: ndc = (P_raw * p_C).xy / (w + 1e-7) ---- */
        {
            const float hx = pm[0] * px + pm[4] * py + pm[8] * pz + pm[12];
            const float hy = pm[1] * px + pm[5] * py + pm[9] * pz + pm[13];
            const float hw = pm[3] * px + pm[7] * py + pm[11] * pz + pm[15];
            const float m_w = 1.0f / (hw + 0.0000001f);
            const float mul1 = hx * m_w * m_w, mul2 = hy * m_w * m_w;
            const float g2x = dL_dmean2D[2 * i], g2y = dL_dmean2D[2 * i + 1];
            /* world-space expression (as upstream) -- equals R^T * (camera-space gradient) */
            float gw[3];
            gw[0] = (pm[0] * m_w - pm[3] * mul1) * g2x + (pm[1] * m_w - pm[3] * mul2) * g2y;
            gw[1] = (pm[4] * m_w - pm[7] * mul1) * g2x + (pm[5] * m_w - pm[7] * mul2) * g2y;
            gw[2] = (pm[8] * m_w - pm[11] * mul1) * g2x + (pm[9] * m_w - pm[11] * mul2) * g2y;
            /* convert to a camera-space gradient: g_pc = R * g_world (R orthonormal rotation of T_cw) */
            for (int r = 0; r < 3; ++r)
                dL_dpc[r] += vm[0 + r] * gw[0] + vm[4 + r] * gw[1] + vm[8 + r] * gw[2];
            (void)projmatrix_raw;
        }
        /* ---- depth path ---- */
        dL_dpc[2] += dL_ddepths[i];
        /* ---- colour path ---- */
        if (in->shs) {
            float gdir[3];
            sh_backward(in->D, in->M, in->shs + (size_t)i * in->M * 3, clamped + 3 * i, px - in->campos[0],
                        py - in->campos[1], pz - in->campos[2], dL_dcolors + 3 * i, dL_dsh + (size_t)i * in->M * 3, gdir);
            for (int k = 0; k < 3; ++k) dL_dp[k] += gdir[k];
            /* campos = -R^T t  =>  d campos / d rho = -R^T ; dir = p - campos => dL/drho = R * gdir */
            for (int r = 0; r < 3; ++r)
                tau[r] += (double)(vm[0 + r] * gdir[0] + vm[4 + r] * gdir[1] + vm[8 + r] * gdir[2]);
        }
        /* ---- assemble: p_C = R p + t ---- */
        for (int k = 0; k < 3; ++k) /* dL/dp = R^T dL/dp_C : R^T[k][r] = R[r][k] = vm[4*k + r] */
            dL_dmeans3D[3 * i + k] = dL_dp[k] + vm[4 * k + 0] * dL_dpc[0] + vm[4 * k + 1] * dL_dpc[1] + vm[4 * k + 2] * dL_dpc[2];
        /* dL/dR from p_C = R p : dL/dR[r][c] += dL_dpc[r] * p[c]  -- folded into tau below via -[p_C]x */
        /* tau: rho -> dp_C/drho = I ; theta -> dp_C/dtheta = -[p_C]x  => dL/dtheta = p_C x dL_dpc */
        tau[0] += dL_dpc[0]; tau[1] += dL_dpc[1]; tau[2] += dL_dpc[2];
        tau[3] += (double)(vy * dL_dpc[2] - vz * dL_dpc[1]);
        tau[4] += (double)(vz * dL_dpc[0] - vx * dL_dpc[2]);
        tau[5] += (double)(vx * dL_dpc[1] - vy * dL_dpc[0]);
        /* rotation entries used by the covariance: R' = (I + [theta]x) R => dR = [theta]x R ;
         * dL/dtheta_k = sum_ij dL_dR[i][j] * ([e_k]x R)[i][j] = sum_j (R[:,j] x dL_dR[:,j])_k */
        for (int j = 0; j < 3; ++j) {
            const float r0 = vm[4 * j + 0], r1 = vm[4 * j + 1], r2 = vm[4 * j + 2];
            const float g0 = dL_dR[0][j], g1 = dL_dR[1][j], g2 = dL_dR[2][j];
            tau[3] += (double)(r1 * g2 - r2 * g1);
            tau[4] += (double)(r2 * g0 - r0 * g2);
            tau[5] += (double)(r0 * g1 - r1 * g0);
        }
    }
    if (dL_dtau) for (int k = 0; k < 6; ++k) dL_dtau[k] = (float)tau[k];
}
