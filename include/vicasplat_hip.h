/*
 * vicasplat_hip.h -- C ABI of libvicasplat_hip.so (MI355X / gfx950 only).
 *
 * This is the drop-in boundary for VicaSplat's feed-forward hot path.  Every entry point takes
 * plain device pointers + sizes + a hipStream_t (passed as void*), never a torch type.  The Python packages
 * vicasplat_amd.diff_gaussian_rasterization / vicasplat_amd.curope / vicasplat_amd.model.* are thin ctypes
 * wrappers over these symbols and re-export the reference's own call surface.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in _host;
 *   - every function is asynchronous on `stream` except where noted (vs_raster_forward performs exactly one
 *     stream synchronisation to learn the number of (Gaussian,tile) instances, like the reference extension);
 *   - return value: >= 0 on success, < 0 on error; vs_last_error() returns a thread-local message
 *     (mirrors TORCH_CHECK in /root/reference/src/model/encoder/backbone/croco/curope/curope.cpp:54-59);
 *   - the library never frees caller memory and keeps no global mutable state besides the thread-local error
 *     string; variable-size scratch comes from the caller's allocator callback (PyTorch caching allocator).
 *   - matrices follow the reference's storage: viewmatrix/projmatrix are 16 floats with element [4*c+r] =
 *     M[r][c] (what cuda_splatting.py:191-194 hands to diff_gaussian_rasterization).
 */
#ifndef VICASPLAT_HIP_H
#define VICASPLAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *vs_stream_t; /* hipStream_t */

const char *vs_last_error(void);
/* ABI version, bumped on any signature change. */
int vs_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Rasterizer (replaces diff_gaussian_rasterization._C.rasterize_gaussians / _backward, imported at
 * /root/reference/src/model/decoder/cuda_splatting.py:5-8 and called at :207-235; per-view Python loop
 * at :199-238 becomes ONE batched call over all cameras of all scenes).
 * ------------------------------------------------------------------------------------------------ */
typedef struct VsRasterIn {
    int32_t num_cameras; /* C: rendered views in this call */
    int32_t num_scenes;  /* S: distinct Gaussian sets; Gaussians are shared by all cameras of a scene */
    int32_t P;           /* Gaussians per scene */
    int32_t sh_degree;   /* active degree as given by the caller (VicaSplat passes 4; bands > 3 are ignored) */
    int32_t sh_coeffs;   /* M: coefficients per channel in memory (25) */
    int32_t width, height;
    int32_t flags;       /* VS_RASTER_* */
    const float *means3D;        /* [S,P,3] */
    const float *cov3D;          /* [S,P,6]  xx xy xz yy yz zz (cov3D_precomp) */
    const float *shs;            /* [S,P,M,3] or NULL */
    const float *colors_precomp; /* [S,P,3]  or NULL (exactly one of shs / colors_precomp) */
    const float *opacities;      /* [S,P] */
    const int32_t *cam_scene;    /* [C] scene index of every camera, or NULL (camera c -> scene c % S) */
    const float *viewmatrix;     /* [C,16] */
    const float *projmatrix;     /* [C,16] */
    const float *campos;         /* [C,3] */
    const float *tanfov;         /* [C,2] (x,y) -- on the DEVICE: no .item() host sync (cuda_splatting.py:210-211) */
    const float *background;     /* [C,3] */
    int64_t capacity;            /* 0: exact mode -- the instance count R is copied back to the host to size the key / list buffers (ONE
                                    stream synchronisation per call, as upstream's rasterize_gaussians does per view).
                                    > 0: no host synchronisation; the buffers are sized for `capacity` (Gaussian, tile) instances.  If the
                                    call produces more, nothing is rendered (background only), buffers[VS_BUF_MISC] int64[2] is set to 1 and
                                    int64[0] holds the R that was needed: the caller checks the flag whenever it next synchronises and
                                    repeats the call with a larger capacity. */
} VsRasterIn;

enum {
    VS_RASTER_COUNT_TOUCHED = 1, /* fill n_touched (MonoGS bookkeeping; VicaSplat discards it) */
    VS_RASTER_SAVE_FOR_BACKWARD = 2, /* (ABI 8) the forward also stores VS_BUF_CHECKPOINT; vs_raster_backward then needs saved->color (and
                                        saved->depth when dL_ddepth is given) alive as well.  Without it the backward walks every list in one piece */
    VS_RASTER_SH_RGB_MAJOR = 4,  /* shs laid out [S,P,3,M] (the encoder's native `harmonics` layout,
                                    gaussian_adapter.py:183) instead of [S,P,M,3]: saves the transpose copy of
                                    cuda_splatting.py:182 */
    VS_RASTER_COV_3X3 = 8,       /* cov3D laid out [S,P,3,3] (full symmetric matrix) instead of [S,P,6]: saves the
                                    triu gather of cuda_splatting.py:224,232 */
};

/* Scratch / saved-state buffers are requested through this callback, tagged so that the caller can keep the
 * ones the backward needs.  Must return 256-byte aligned device memory valid until the caller releases it. */
enum {
    VS_BUF_GEOM = 0,      /* [C,P,12] f32: x y ext_x ext_y | conic.x conic.y conic.z opacity | r g b depth (ext = {alpha>=1/255} half extents) */
    VS_BUF_RECT = 1,      /* [C,P,4] u16 tile rectangle (min.x min.y max.x max.y) */
    VS_BUF_CLAMPED = 2,   /* [C,P] u8 bit c set => colour channel c clamped at 0 */
    VS_BUF_TILE_RANGES = 3, /* [C,tiles,2] i32, followed by [C*tiles] i32: the (camera, tile) ids in the launch order of the per-tile kernels (largest lists first) */
    VS_BUF_TILE_CURSOR = 4, /* [C,tiles] i32 scratch */
    VS_BUF_KEYS = 5,      /* [R] u64 (depth_bits<<32 | gaussian) */
    VS_BUF_POINT_LIST = 6, /* [R] u32 sorted Gaussian ids, per (camera,tile) segment */
    VS_BUF_SORT_SCRATCH = 7, /* [R] u64 bucketized / ping-pong keys of the large-tile sort + its segment table */
    VS_BUF_FINAL_T = 8,   /* [C,H,W] f32 */
    VS_BUF_N_CONTRIB = 9, /* [C,H,W] i32 */
    VS_BUF_MISC = 10,     /* int64[4] control block: [0] R, [1] largest tile population, [2] capacity overflow flag */
    VS_BUF_DEPTH = 11,    /* [C,P] f32 view-space depth of the visible pairs (sort-key source; undefined where rect is all-zero) */
    VS_BUF_CHECKPOINT = 12, /* (ABI 8; only under VS_RASTER_SAVE_FOR_BACKWARD) [slots,5,256] f32 blending state (T | Cr | Cg | Cb | D) of a tile's 256
                               pixels in front of every 512th entry of its list, then [slots] {i32 tile, i32 segment}; slots = (R >> 9) + C * tiles.
                               With it vs_raster_backward replays the 512-entry segments of a list independently (front to back) */
    VS_BUF_COUNT = 13
};
typedef void *(*VsAllocFn)(void *ctx, int32_t tag, size_t bytes);

typedef struct VsRasterOut {
    float *color;       /* [C,3,H,W] */
    float *depth;       /* [C,H,W]   alpha-weighted depth (un-normalised) */
    float *opacity;     /* [C,H,W]   1 - T */
    int32_t *radii;     /* [C,P] */
    int32_t *n_touched; /* [C,P] (zero-filled unless VS_RASTER_COUNT_TOUCHED) */
    int64_t num_rendered; /* R, filled on return */
    void *buffers[VS_BUF_COUNT]; /* filled on return with what the allocator handed out */
} VsRasterOut;

/* returns R (>= 0; `capacity` in the capacity mode, where R itself stays on the device: buffers[VS_BUF_MISC] int64[0]) or < 0 */
int64_t vs_raster_forward(const VsRasterIn *in, VsRasterOut *out, VsAllocFn alloc, void *alloc_ctx, vs_stream_t stream);

typedef struct VsRasterGrads {
    /* incoming */
    const float *dL_dcolor; /* [C,3,H,W] */
    const float *dL_ddepth; /* [C,H,W] or NULL */
    /* outgoing; each is accumulated over the cameras of a scene */
    float *dL_dmeans3D;   /* [S,P,3] */
    float *dL_dcov3D;     /* [S,P,6], or [S,P,3,3] under VS_RASTER_COV_3X3 (off-diagonal partials split in halves over the two symmetric entries) */
    float *dL_dshs;       /* [S,P,M,3] or NULL */
    float *dL_dcolors_precomp; /* [S,P,3] or NULL */
    float *dL_dopacities; /* [S,P] */
    float *dL_dmeans2D;   /* [C,P,2] (NDC units, as upstream's screen-space gradient holder) or NULL */
    float *dL_dtau;       /* [C,6] (rho, theta) for T_cw' = Exp(tau) T_cw, or NULL */
} VsRasterGrads;

/* `saved` is the VsRasterOut of the matching forward (buffers[] must still be alive). */
int vs_raster_backward(const VsRasterIn *in, const VsRasterOut *saved, const VsRasterGrads *g, VsAllocFn alloc,
                       void *alloc_ctx, vs_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * 2-D RoPE (replaces curope.rope_2d, /root/reference/src/model/encoder/backbone/croco/curope/curope.cpp:49-69,
 * kernels.cu:17-108).  In place on tokens viewed as [B,N,H,D]: requires stride(3)=1, stride(2)=D;
 * stride(0)=sB, stride(1)=sN in elements.  pos: int64 [B,N,2] contiguous (y,x).  dtype: 0=f32 1=f16 2=bf16.
 * fwd = +F0 forward, -F0 backward.
 * ------------------------------------------------------------------------------------------------ */
int vs_rope2d(void *tokens, const int64_t *pos, int32_t B, int32_t N, int32_t H, int32_t D, int64_t sB, int64_t sN,
              float base, float fwd, int32_t dtype, vs_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * ViT block operators (replace nn.LayerNorm / nn.Linear / softmax attention / F.scaled_dot_product_attention on the
 * encoder path: croco/blocks.py:73-130, backbone_vica.py:76-126,152-191,268-335).  dtype: 1 = f16, 2 = bf16 operands;
 * accumulation, LayerNorm, softmax and the residual stream are f32.
 * dtype 3 = f32 operands -- the reference-precision path (the reference stores fp32 and multiplies in TF32,
 * backbone_vica.py:9; gfx950 has no TF32, so the parity path is exact f32 on v_mfma_f32_16x16x4_f32 at 1/16 of the
 * 16-bit matrix rate): accepted by vs_gemm_bias_act / vs_gemm_resid / vs_gemm_qkv_rope (K % 32 == 0, lda/ldw % 4 == 0),
 * vs_attention(_lse), vs_conv3x3_nhwc (Cin % 16 == 0) and vs_upsample2x_nhwc.  A, W, activations and EVERY output that is
 * "16-bit" for dtype 1/2 (epilogues 0, 1, the RoPE epilogue, attention / convolution / upsample outputs, residuals) are
 * float arrays then; strides stay in elements.
 * dtype 4 = SPLIT operands (round 3) -- f32-class results at a third of the 16-bit matrix rate, the tolerance-meeting precision of
 * the headline benchmark: activations and outputs are float arrays exactly as for dtype 3; a weight is packed once by
 * vs_split_pack_weight into f16 (hi, lo) pairs, hi = rne16(w 2^e), lo = rne16(w 2^e - hi); the kernels convert the activation
 * fragments the same way in registers and form every product as three v_mfma_f32_16x16x32_f16 (lo_w hi_a + hi_w lo_a + hi_w hi_a,
 * f32 accumulate; the dropped lo lo term is 2^-22 relative).  Entry points: vs_gemm_split, vs_conv3x3_split_nhwc, and dtype 4 of
 * vs_attention(_lse) (q | k | v | out f32; Q, K, V and P are split in the kernel).  Replaces the reference's TF32 nn.Linear /
 * nn.Conv2d / attention matmuls (backbone_vica.py:9) with arithmetic that is at least as precise as theirs in every product.
 * ------------------------------------------------------------------------------------------------ */

/* W [N, K] f32 (row stride ldw floats; K % 32 == 0) -> out [N, K] 4-byte units (row stride ldo units): per block of 32 k, 32 hi halves
 * then 32 lo halves of w * 2^scale_exp (inside a block, 8-half chunk g holds k = {4g..4g+3, 16+4g..16+4g+3}: the order in which a lane
 * group reads an f32 activation row).  The caller picks scale_exp so that max|w| 2^scale_exp stays below 2^15 (f16 range) and passes
 * acc_scale = 2^-scale_exp to the product entry points. */
int vs_split_pack_weight(const float *w, int64_t ldw, void *out, int64_t ldo, int32_t N, int32_t K, int32_t scale_exp, vs_stream_t stream);

/* out = epilogue(acc_scale * (A Wp^T) + bias), A [M, K] f32, Wp from vs_split_pack_weight, out f32.  epilogue 0 / 3 store, 1 exact-erf
 * GELU, 2 out = (resid ? resid : out) + (1 + gate) * (...), 4 packed q|k|v with RoPE (pos, kind, C, base2d, theta1d as vs_gemm_qkv_rope).
 * Row maps / gate / strides (in floats) as vs_gemm_bias_act.  Epilogue 5 (late round 5; vs_gemm_split only): out = (A Wp^T) * GELU'(z), z = resid
 * [M, ldo] f32 in the layout of out (no bias / gate / row map / packed output) -- the gradient of the MLP's pre-activation straight from the dX
 * GEMM of its second linear (torch autograd of fc2(act(fc1(x))), croco/blocks.py:60-72 under model_wrapper.py:184-321). */
int vs_gemm_split(const float *A, const void *Wp, float acc_scale, const float *bias, float *out, const float *gate, const float *resid,
                  int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldw, int32_t ldo, int32_t epilogue, int32_t grp_in, int32_t grp_out,
                  int32_t grp_off, int32_t gate_rows, int32_t gate_ld, int32_t a_grp_in, int32_t a_grp_out, int32_t a_grp_off,
                  const int32_t *pos, const uint8_t *kind, int32_t C, float base2d, float theta1d, vs_stream_t stream);

/* vs_conv3x3_split_nhwc with a second f32 residual: out = act(conv(in) + bias + residual + residual2) -- the FeatureFusionBlock's
 * x + ResidualConvUnit(skip) (heads/dpt_block.py:196-208) folded into the unit's last convolution (no element-wise add pass). */
int vs_conv3x3_split_res2_nhwc(const float *in, const void *wp, float acc_scale, const float *bias, const float *residual, const float *residual2,
                               float *out, int32_t Nimg, int32_t Hin, int32_t Win, int32_t Cin, int32_t Cout, int32_t stride, int32_t relu_in,
                               int32_t relu_out, vs_stream_t stream);
/* vs_gemm_split with the A operand ALREADY in the packed (hi, lo) form (Ap = vs_split_pack_weight(A, scale_exp 0) or a producer that
 * writes that form; lda in 4-byte units): the main loops skip the in-kernel conversion.  M > 64 routes only (tile kernels). */
int vs_gemm_split_packed(const void *Ap, const void *Wp, float acc_scale, const float *bias, float *out, const float *gate, const float *resid,
                         int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldw, int32_t ldo, int32_t epilogue, int32_t grp_in, int32_t grp_out,
                         int32_t grp_off, int32_t gate_rows, int32_t gate_ld, int32_t a_grp_in, int32_t a_grp_out, int32_t a_grp_off,
                         const int32_t *pos, const uint8_t *kind, int32_t C, float base2d, float theta1d, vs_stream_t stream);
/* vs_conv7x7_rgb_nhwc on split operands: in_padded f32 [Nimg, Hp, Wp, 3] (+ one spare padded row + 64 spare floats), wp = packed
 * [Cout, 8 * 32] (kernel row dy at columns dy * 32 + dx * 3 + c), out f32 [Nimg, H, W, Cout], Cout % 256 == 0 */
int vs_conv7x7_rgb_split_nhwc(const float *in_padded, const void *wp, float acc_scale, const float *bias, float *out, int32_t Nimg, int32_t H,
                              int32_t W, int32_t Hp, int32_t Wp, int32_t Cout, vs_stream_t stream);
/* The split-class stem with the Gaussian-parameter head's "bilinear x2 of the trunk + ReLU(stem)" (dpt_gs_head.py:142-150) fused into its
 * epilogue (ABI 5): out_packed [Nimg, H, W, Cout] in the packed (hi, lo) form = bilinear_x2(up_src [Nimg, H/2, W/2, Cout] f32, align_corners)
 * + relu(conv7x7(in_padded) + bias): what vs_upsample2x_nhwc(up_src, add = stem output, relu_add + 16) produced, without the f32 stem map. */
int vs_conv7x7_rgb_split_up_nhwc(const float *in_padded, const void *wp, float acc_scale, const float *bias, const float *up_src, void *out_packed,
                                 int32_t Nimg, int32_t H, int32_t W, int32_t Hp, int32_t Wp, int32_t Cout, vs_stream_t stream);

/* conv3x3(Cin -> 128) -> relu_out -> conv1x1(128 -> C2 <= 4) in one kernel on split operands (dot-product form of
 * vs_conv3x3_head1x1_nhwc): in f32 NHWC, wp packed [128, 9 * Cin], w2 f32 [C2, 128], bias2 f32 [4], out2 f32 [N*H*W, ld2] */
int vs_conv3x3_head_dot_split_nhwc(const float *in, const void *wp, float acc_scale, const float *bias, const float *w2, const float *bias2,
                                   float *out2, int32_t Nimg, int32_t H, int32_t W, int32_t Cin, int32_t C2, int32_t ld2, int32_t relu_in,
                                   int32_t relu_out, vs_stream_t stream);

/* conv3x3(Cin -> 256) -> relu_out -> conv1x1(256 -> C2pad <= 96) in one kernel on split operands (MFMA form of vs_conv3x3_head1x1_nhwc):
 * w2p = packed [C2pad, 256], bias2 f32 [C2pad], out2 f32 [N*H*W, ld2 >= C2pad] */
int vs_conv3x3_head1x1_split_nhwc(const float *in, const void *wp, float acc_scale, const float *bias, const void *w2p, float acc_scale2,
                                  const float *bias2, float *out2, int32_t Nimg, int32_t H, int32_t W, int32_t Cin, int32_t C2, int32_t C2pad,
                                  int32_t ld2, int32_t relu_out, vs_stream_t stream);

/* vs_conv3x3_nhwc on split operands: in / residual / out f32 NHWC, wp = packed [Cout, 9 * Cin] (tap-major, channel-minor; Cin % 32 == 0) */
int vs_conv3x3_split_nhwc(const float *in, const void *wp, float acc_scale, const float *bias, const float *residual, float *out, int32_t Nimg,
                          int32_t Hin, int32_t Win, int32_t Cin, int32_t Cout, int32_t stride, int32_t relu_in, int32_t relu_out,
                          vs_stream_t stream);

/* y = LN(x; w, b, eps) [* (1 + scale[row / mod_rows]) + shift[row / mod_rows]].  x f32 [M,C] (row stride ldx);
 * out_dtype 0 = f32, 1 = f16, 2 = bf16 (row stride ldo); output row = (row / grp_in) * grp_out + grp_off + row % grp_in
 * (grp_in <= 0: identity).  scale/shift may be NULL. */
int vs_layernorm_mod(const float *x, int64_t ldx, const float *w, const float *b, const float *scale, const float *shift,
                     int32_t mod_rows, int32_t mod_ld, void *out, int64_t ldo, int32_t out_dtype, int32_t M, int32_t C,
                     float eps, int32_t grp_in, int32_t grp_out, int32_t grp_off, vs_stream_t stream);

/* Tiny f32 linear layers of the camera-token path (intrinsic embedding 9 -> 1024, backbone_vica.py:393,535-536; pose / fov heads
 * ReLU -> Linear, vicasplat.py:118-138): out[m,n] = sum_k act(x[m,k]) w[n,k] + bias[n], relu_in 0 | 1; all f32, strides in elements. */
int vs_linear_f32(const float *x, int64_t ldx, const float *w, int64_t ldw, const float *bias, float *out, int64_t ldo, int32_t M, int32_t N,
                  int32_t K, int32_t relu_in, vs_stream_t stream);
/* out[i] = silu(x[i]), x f32 -> out_dtype 0 f32 / 1 f16 / 2 bf16 (the SiLU in front of the AdaLN projections, backbone_vica.py:210-212); n % 4 == 0 */
int vs_silu_cast(const float *x, void *out, int64_t n, int32_t out_dtype, vs_stream_t stream);

/* out = epilogue(A[M,K] * W[N,K]^T + bias).  epilogue: 0 store 16-bit, 1 exact-erf GELU then store 16-bit,
 * 2 f32 residual update out += (1 + gate[row / gate_rows]) * (.), 3 store f32.  K % 64 == 0; lda/ldw % 8 == 0.
 * Output row mapping (grp_*) as in vs_layernorm_mod; a_grp_* is the same mapping applied to the rows of A that are
 * READ (lets a GEMM consume only the image-token rows, or only the camera-token rows, of an interleaved buffer). */
int vs_gemm_bias_act(const void *A, const void *W, const float *bias, void *out, const float *gate, int32_t M, int32_t N,
                     int32_t K, int32_t lda, int32_t ldw, int32_t ldo, int32_t epilogue, int32_t dtype, int32_t grp_in,
                     int32_t grp_out, int32_t grp_off, int32_t gate_rows, int32_t gate_ld, int32_t a_grp_in,
                     int32_t a_grp_out, int32_t a_grp_off, vs_stream_t stream);

/* Packed qkv projection with the rotary embedding fused into the epilogue: out16[row, :] = A W^T + bias, then q (columns
 * [0,C)) and k ([C,2C)) of every 64-wide head are rotated exactly as vs_rope_qk would (pos [rows,2] int32 and kind [rows]
 * uint8 are indexed by OUTPUT row; kind NULL = all 0).  Replaces nn.Linear + RoPE2D.forward (croco/blocks.py:94-104,
 * pos_embed.py:106-157) / the temporal rope of backbone_vica.py:95-118 without a second pass over the qkv buffer. */
int vs_gemm_qkv_rope(const void *A, const void *W, const float *bias, void *out, int32_t M, int32_t N, int32_t K, int32_t lda,
                     int32_t ldw, int32_t ldo, int32_t dtype, int32_t grp_in, int32_t grp_out, int32_t grp_off, int32_t a_grp_in,
                     int32_t a_grp_out, int32_t a_grp_off, const int32_t *pos, const uint8_t *kind, int32_t C, float base2d,
                     float theta1d, vs_stream_t stream);

/* In-place RoPE on the q (column 0) and k (column k_col) blocks of a packed projection buffer [rows, ld], H heads of
 * 64.  pos int32 [rows,2] (y,x) or (t,-); kind uint8 [rows] (0 = 2-D, 1 = temporal 1-D interleaved, 2 = none) or NULL. */
int vs_rope_qk(void *buf, int64_t ld, int32_t rows, int32_t H, int32_t k_col, const int32_t *pos, const uint8_t *kind,
               float base2d, float theta1d, int32_t dtype, vs_stream_t stream);
/* Same with a direction: dir = +1 forward, -1 the inverse rotation (= the backward pass of the embedding on dq, dk). */
int vs_rope_qk_dir(void *buf, int64_t ld, int32_t rows, int32_t H, int32_t k_col, const int32_t *pos, const uint8_t *kind,
                   float base2d, float theta1d, float dir, int32_t dtype, vs_stream_t stream);

/* Fused attention, head dim 64.  Batch item b, head h: queries rows b*q_batch_rows + [0,Lq) of q (row stride ldq,
 * head h at column h*64); keys/values rows b*k_batch_rows + [0,Lk) of k / v -- or, when kv_seg != NULL, the two row
 * segments kv_seg[b] = {base0, len0, base1, len1} concatenated.  q_kvlen (optional, [nbatch*Lq]) limits query i to
 * the first q_kvlen keys.  out: [rows, ldo], head h at column h*64. */
int vs_attention(const void *q, const void *k, const void *v, void *out, int32_t nbatch, int32_t H, int32_t Lq, int32_t Lk,
                 int64_t q_batch_rows, int64_t k_batch_rows, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                 const int32_t *kv_seg, const int32_t *q_kvlen, float scale, int32_t dtype, vs_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused DPT post-process + Gaussian adapter (replaces heads/postprocess.py:46-56, the raw_gaussians cat of
 * vicasplat.py:256 and MyGaussianAdapter.forward, common/gaussian_adapter.py:168-212).  One Gaussian per pixel.
 * pts: 3 channels, gs: 8 + 3*d_sh channels (opacity | scale 3 | quaternion xyzw | SH rgb-major), both addressed
 * as base[pixel * *_pix + channel * *_ch] (elements), in_dtype 0 f32 / 1 f16 / 2 bf16.  scale_act 0 bounded,
 * 1 exp, 2 softplus.  opacity_exponent = 2^x of map_pdf_to_opacity (<= 0: predict_opacity, no mapping).
 * Outputs (f32): means [n,3], cov [n,3,3], harmonics [n,3,d_sh], opacities [n], scales [n,3], rotations [n,4],
 * raw [n, 11 + 3*d_sh] (may be NULL).
 * ------------------------------------------------------------------------------------------------ */
int vs_gaussian_adapter(const void *pts, int64_t pts_pix, int64_t pts_ch, const void *gs, int64_t gs_pix, int64_t gs_ch,
                        int32_t in_dtype, int64_t npix, int32_t d_sh, const float *sh_mask, int32_t scale_act,
                        float scale_min, float scale_max, float opacity_exponent, float *means, float *cov, float *harmonics,
                        float *opacities, float *scales, float *rotations, float *raw, vs_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * DPT-head convolutions on NHWC 16-bit activations (replace nn.Conv2d(k=3,s=1,p=1), ResidualConvUnit_custom and
 * F.interpolate(scale_factor=2, bilinear, align_corners=True) of heads/dpt_block.py:79-218,316-343).
 *   vs_conv3x3_nhwc : out = [relu](conv3x3([relu](in)) + bias [+ residual]); implicit GEMM on MFMA, no im2col buffer.
 *                     in [N,H,W,Cin], w [Cout,3,3,Cin] (tap-major, channel-minor), out/residual [N,H,W,Cout]; Cin % 64 == 0.
 *                     relu_out == 2: `residual` is not added but used as a mask, out = residual > 0 ? conv : 0 -- the data gradient
 *                     of a conv whose input went through a ReLU (the conv runs on dY with flipped weights, residual = that input).
 *   vs_upsample2x_nhwc : out [N,2H,2W,C] = bilinear(in [N,H,W,C]) [+ add | + relu(add)]; C % 8 == 0.
 * ------------------------------------------------------------------------------------------------ */
int vs_conv3x3_nhwc(const void *in, const void *w, const float *bias, const void *residual, void *out, int32_t Nimg, int32_t Hin,
                    int32_t Win, int32_t Cin, int32_t Cout, int32_t stride /* 1 or 2; out = (in-1)/stride+1 */, int32_t relu_in,
                    int32_t relu_out, int32_t dtype, vs_stream_t stream);
/* out2[pixel, 0..C2) = W2 * act(conv3x3(in) + bias) + bias2 in ONE kernel -- the last two layers of both DPT heads (conv3 -> ReLU ->
 * conv1, heads/dpt_block.py:316-343): the 3x3 result never leaves the workgroup.  Form A (Gaussian-parameter head): Cout = 256,
 * w2 [C2pad, 256] 16-bit (rows >= C2 zero), bias2 [C2pad] f32, C2pad a multiple of 16 <= 96, ld2 >= C2pad: second GEMM on the MFMA
 * through LDS.  Form B (pts3d head): Cout = 128, C2 <= 4, w2 [C2, 128], bias2 [4], ld2 >= 4: VALU dot products.  stride 1, pad 1,
 * N*H*W % 256 == 0, relu_out 0 | 1 applies to the 3x3 result, dtype 1 f16 / 2 bf16. */
int vs_conv3x3_head1x1_nhwc(const void *in, const void *w, const float *bias, const void *w2, const float *bias2, void *out2, int32_t Nimg,
                            int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t C2, int32_t C2pad, int32_t ld2, int32_t relu_in,
                            int32_t relu_out, int32_t dtype, vs_stream_t stream);
int vs_upsample2x_nhwc(const void *in, const void *add, void *out, int32_t Nimg, int32_t H, int32_t W, int32_t C,
                       int32_t relu_add, int32_t dtype, vs_stream_t stream);
/* Backward of vs_upsample2x_nhwc (without add): din [N,H,W,C] = bilinear-x2^T applied to dout [N,2H,2W,C]; H, W = INPUT size. */
int vs_upsample2x_backward_nhwc(const void *dout, void *din, int32_t Nimg, int32_t H, int32_t W, int32_t C, int32_t dtype,
                                vs_stream_t stream);

/* 7x7, stride 1, pad 3 convolution of an RGB image (gs head input_merger, heads/dpt_gs_head.py:112-118; replaces the
 * nn.Conv2d(3, C, 7, 1, 3) call) as a window GEMM on MFMA without an im2col buffer.
 *   in_padded [Nimg, Hp, Wp, 3] 16-bit NHWC, zero border: 3 rows/cols before the image, >= 3 after (Hp >= H+6, Wp >= W+6),
 *             plus >= Wp*3 + 64 readable halfs after the last element (the 256-tile route's zero-weight eighth kernel row);
 *   w         [Cout, 8, 32]: w[co, dy, dx*3 + c] = weight[co, c, dy, dx], entries 21..31 of every row and the whole row dy = 7 zero;
 *   out       [Nimg, H, W, Cout] 16-bit (+ bias[Cout] f32 or NULL). */
int vs_conv7x7_rgb_nhwc(const void *in_padded, const void *w, const float *bias, void *out, int32_t Nimg, int32_t H, int32_t W,
                        int32_t Hp, int32_t Wp, int32_t Cout, int32_t dtype, vs_stream_t stream);

/* vs_attention that also saves, per (row, head), the log2-domain logsumexp of the scaled scores (lse [rows, H] f32) --
 * the only forward state the flash-style backward needs besides out. */
int vs_attention_lse(const void *q, const void *k, const void *v, void *out, int32_t nbatch, int32_t H, int32_t Lq, int32_t Lk,
                     int64_t q_batch_rows, int64_t k_batch_rows, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                     const int32_t *kv_seg, const int32_t *q_kvlen, float scale, int32_t dtype, float *lse, vs_stream_t stream);

/* Backward of vs_attention (same addressing, mask and key segments).  o = forward output, dout = its gradient (row stride
 * lddo), lse from vs_attention_lse; delta [rows, H] f32 is scratch.  dq: 16-bit [rows, lddq] (written); dk, dv: f32 with row
 * strides lddk / lddv, indexed by KEY row, ADDED to with f32 atomics (zero them first: with key segments one K/V row
 * receives gradient from several batch items).  max_keys = largest len0 + len1 when kv_seg is used. */
int vs_attention_backward(const void *q, const void *k, const void *v, const void *o, const void *dout, const float *lse, float *delta,
                          void *dq, float *dk, float *dv, int32_t nbatch, int32_t H, int32_t Lq, int32_t Lk, int64_t q_batch_rows,
                          int64_t k_batch_rows, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo, int32_t lddo, int32_t lddq,
                          int32_t lddk, int32_t lddv, const int32_t *kv_seg, const int32_t *q_kvlen, int32_t max_keys, float scale,
                          int32_t dtype, vs_stream_t stream);

/* The same without key segments (every K/V row belongs to one batch item): dk, dv are 16-bit [key rows, lddk / lddv] and are WRITTEN
 * (plain stores: no atomics, no zero fill, no f32 -> 16-bit cast pass afterwards), e.g. the k | v blocks of a packed [rows, 3*H*64]
 * gradient buffer.  Reference: torch autograd of F.scaled_dot_product_attention in croco/blocks.py:106-110. */
int vs_attention_backward16(const void *q, const void *k, const void *v, const void *o, const void *dout, const float *lse, float *delta,
                            void *dq, void *dk, void *dv, int32_t nbatch, int32_t H, int32_t Lq, int32_t Lk, int64_t q_batch_rows,
                            int64_t k_batch_rows, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo, int32_t lddo, int32_t lddq,
                            int32_t lddk, int32_t lddv, const int32_t *q_kvlen, float scale, int32_t dtype, vs_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Encoder backward building blocks (training_step, model_wrapper.py:184-321: the reference differentiates the encoder
 * with torch autograd).  They are assembled into torch.autograd.Functions in vicasplat_amd/autograd.py and drive
 * vicasplat_amd.callers.training_step.
 *   vs_transpose16        out[c, r] = in[r, c] for r < R (zero for R <= r < Rpad); 16-bit elements.  With it the NT GEMM
 *                         (vs_gemm_bias_act) computes dX = dY W (A = dY, W-operand = W^T) and dW = dY^T X (A = dY^T,
 *                         W-operand = X^T, epilogue 3) of nn.Linear; Rpad pads the reduction dimension to a multiple of 64.
 *   vs_colsum             out[n] = sum_m x[m, n] (bias gradient); dtype 0 f32, 1 f16, 2 bf16; out is overwritten.
 *   vs_gelu_backward      dz = dy * d/dz gelu_erf(z) on n 16-bit elements (n % 8 == 0).
 *   vs_layernorm_backward backward of vs_layernorm_mod: out = (xhat w + b)(1 + scale[g]) + shift[g], g = row / mod_rows.
 *                         dout is read at the forward's OUTPUT row (grp_* mapping) in dtype do_dtype (0 f32, 1 f16, 2 bf16);
 *                         dx f32 [M,C] is written (accumulate_dx = 0) or added to (1); dw, db [C] and dscale, dshift
 *                         [G, mod_ld] are ADDED to (f32 atomics): zero them before the first call of a step.
 * ------------------------------------------------------------------------------------------------ */
/* dx = x > 0 ? dx : 0, in place, n 16-bit elements (f16 or bf16: same sign/zero encoding); backward of a ReLU on x. */
int vs_relu_mask16(void *dx, const void *x, int64_t n, vs_stream_t stream);
/* out = x > 0 ? dy : 0 (out may be dy): the same without a copy when the incoming gradient must stay intact. */
int vs_relu_mask16_to(const void *dy, const void *x, void *out, int64_t n, vs_stream_t stream);
/* out32[M,N] += A[M,K] W[N,K]^T, K cut into ksplit slices (separate workgroups, f32 atomics): long thin reductions such as
 * weight gradients.  K % (32 * ksplit) == 0; A, W may start at any 2-byte aligned address (shifted views).  Runs on 256x256 tiles
 * (the phase-interleaved main loop of the forward GEMMs) when M % 256 == 0, N % 256 == 0 and K % (128 * ksplit) == 0, on 128x128
 * tiles otherwise. */
int vs_gemm_splitk_accumulate(const void *A, const void *W, float *out, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldw,
                              int32_t ldo, int32_t ksplit, int32_t dtype, vs_stream_t stream);
/* ntaps (<= 9) split-K GEMMs sharing A in one launch: out32[t] [M,N] (t-th block of tap_out_stride floats) += A (W + shifts[t])^T;
 * shifts is a HOST array of element offsets.  The 3x3 convolution weight gradient (A = dY^T, W = zero-bordered X^T, shift = tap). */
int vs_gemm_taps_accumulate(const void *A, const void *W, float *out, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldw,
                            int32_t ldo, int64_t tap_out_stride, const int32_t *shifts, int32_t ntaps, int32_t ksplit, int32_t dtype,
                            vs_stream_t stream);
/* General form of the two above: ntaps = 0 (plain) or 1..9 (taps), plus an optional workspace of >= slices * max(ntaps, 1) * M * N
 * floats (slices = ksplit, or 2 when ksplit == 1 on the 128x128 tiling; 16-byte aligned).  With it the K slices store partial tiles
 * and a second kernel sums them into out -- f32 atomics sustain ~0.3 TB/s on MI355X, plain stores ten times that -- without it
 * they meet through atomics as above.  out is added to (accumulate != 0) or, workspace mode only, overwritten (accumulate == 0).
 * a_slice_stride / w_slice_stride (elements): 0 = K slice s is columns [s K/ksplit, (s+1) K/ksplit) of A / W; > 0 = slice-blocked
 * operands [slice][channel][K/ksplit (+ halo)] as vs_transpose16_ex writes them: slice s is columns [0, K/ksplit) of the matrix at
 * A + s * a_slice_stride.  Blocking keeps the rows a workgroup walks KBs apart instead of the whole reduction length (MBs). */
int vs_gemm_wgrad(const void *A, const void *W, float *out, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldw, int32_t ldo,
                  int64_t a_slice_stride, int64_t w_slice_stride, int64_t tap_out_stride, const int32_t *shifts, int32_t ntaps,
                  int32_t ksplit, int32_t dtype, void *workspace, int64_t workspace_bytes, int32_t accumulate, vs_stream_t stream);
/* Weight gradient from REDUCTION-MAJOR operands (no transposed copies): out32[M,N] (+)= sum_{k < Kred} A[k,m] W[k,n], A [Kred,M] and
 * W [Kred,N] row-major 16-bit, i.e. dW = dY^T X straight from dY [tokens, out features] and X [tokens, in features].  The kernel
 * gathers its MFMA operands with the LDS transpose read ds_read_b64_tr_b16.  lda, ldw multiples of 8 (padded rows allowed); A, W
 * 16-byte aligned; M, N arbitrary (256 x 256 tiles, zero page beyond the row storage); ksplit slices of the reduction, zero-filled past Kred; workspace / accumulate as in vs_gemm_wgrad. */
int vs_gemm_wgrad_tn(const void *A, const void *W, float *out, int32_t M, int32_t N, int32_t Kred, int32_t lda, int32_t ldw, int32_t ldo,
                     int32_t ksplit, int32_t dtype, void *workspace, int64_t workspace_bytes, int32_t accumulate, vs_stream_t stream);
/* Weight gradient of nn.Conv2d(k=3, s=1, p=1) from the NHWC tensors as they are (reduction-major, LDS transpose reads, zero page
 * outside the image -- no zero-bordered transposed copies): out32[tap][ci][co] (+)= sum over pixels of act(x)[pixel + tap][ci] *
 * dy[pixel][co]; x [Nimg,H,W,Cin], dy [Nimg,H,W,Cout] contiguous 16-bit, out [9, Cin, Cout], tap = ky*3 + kx; relu_in = the
 * ResidualConvUnit's activation-before-conv on x.  Cin, Cout multiples of 8 (256 x 256 tiles: efficient for multiples of 256).
 * workspace >= ksplit * 9 * Cin * Cout floats or null (atomics); accumulate as in vs_gemm_wgrad. */
int vs_conv3x3_wgrad_tn(const void *x, const void *dy, float *out, int32_t Nimg, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                        int32_t relu_in, int32_t ksplit, int32_t dtype, void *workspace, int64_t workspace_bytes, int32_t accumulate,
                        vs_stream_t stream);
/* Epilogue 2 of vs_gemm_bias_act with the residual read from a second buffer: out32 = resid + (1 + gate) * (A W^T + bias); resid and out
 * share the layout (ldo).  For callers that must keep the old residual stream (training) without cloning it. */
int vs_gemm_resid(const void *A, const void *W, const float *bias, const float *resid, float *out, const float *gate, int32_t M, int32_t N,
                  int32_t K, int32_t lda, int32_t ldw, int32_t ldo, int32_t dtype, int32_t gate_rows, int32_t gate_ld, vs_stream_t stream);
int vs_transpose16(const void *in, int64_t ld_in, void *out, int64_t ld_out, int32_t R, int32_t C, int32_t Rpad, vs_stream_t stream);
/* vs_transpose16 with extras: colsum (nullable, f32 [C], overwritten) = column sums of the input, i.e. the bias gradient rides on
 * the transpose of dY that the weight-gradient GEMM needs anyway (dtype 1 f16 / 2 bf16); border_h, border_w > 0: the R input rows
 * are the pixels of zero-bordered (border_h + 2) x (border_w + 2) maps whose interior is read from the unpadded NHWC tensor `in`
 * (the operands of the conv weight gradient without a padded copy); relu != 0 writes negative inputs as zero.
 * nslices > 1: slice-blocked output [slice][C][SL + 2 halo], SL = Rpad / nslices: slice z = transposed rows [z SL - halo, (z+1) SL + halo)
 * (zeros outside [0, R)) at out + z * slice_stride, row stride ld_out -- the operand layout of vs_gemm_wgrad; the halo columns are
 * what the shifted tap views of a 3x3 weight gradient read beyond their own slice (colsum needs halo == 0). */
int vs_transpose16_ex(const void *in, int64_t ld_in, void *out, int64_t ld_out, int32_t R, int32_t C, int32_t Rpad, float *colsum,
                      int32_t dtype, int32_t border_h, int32_t border_w, int32_t relu, int32_t nslices, int32_t halo,
                      int64_t slice_stride, vs_stream_t stream);
int vs_colsum(const void *x, int64_t ld, float *out, int32_t M, int32_t N, int32_t dtype, vs_stream_t stream);
/* out = gelu_erf(z) on n 16-bit elements (n % 8 == 0): the training forward's activation pass (z is kept for vs_gelu_backward). */
int vs_gelu16(const void *z, void *out, int64_t n, int32_t dtype, vs_stream_t stream);
int vs_gelu_backward(const void *dy, const void *z, void *dz, int64_t n, int32_t dtype, vs_stream_t stream);
/* Gated residual update of the decoder image stream: out[m,c] = x[m,c] + (1 + gate[m / gate_rows, c]) * y[yrow(m), c] with
 * yrow(m) = (m / grp_in) * grp_out + grp_off + m % grp_in (grp_in <= 0: identity); x, out f32 [M,C] (may alias), y 16-bit with row
 * stride ldy, gate f32 [G,C] or null (plain residual).  backbone_vica.py:274-278,302,327,331 in one pass. */
int vs_gated_resid(const float *x, const void *y, int64_t ldy, const float *gate, int32_t gate_rows, float *out, int32_t M, int32_t C,
                   int32_t grp_in, int32_t grp_out, int32_t grp_off, int32_t dtype, vs_stream_t stream);
/* Its backward on the branch side: dy[yrow(m), c] = dout[m,c] * (1 + gate) (16-bit, row stride lddy; rows outside the map are not
 * written) and dgate[g, c] += sum over the group's rows of dout * y (f32 atomics: zero dgate first); dx = dout needs no kernel. */
int vs_gated_resid_backward(const float *dout, const void *y, int64_t ldy, const float *gate, int32_t gate_rows, void *dy, int64_t lddy,
                            float *dgate, int32_t M, int32_t C, int32_t grp_in, int32_t grp_out, int32_t grp_off, int32_t dtype,
                            vs_stream_t stream);
/* vs_layernorm_backward with the residual-path gradient read from its own buffer (dx = dx_add + LayerNorm gradient; dx_add null, another
 * buffer, or dx itself) and an optional 16-bit copy of dx (dx16 with row stride ld16, dx16_dtype 1 f16 / 2 bf16) for the GEMMs that
 * consume the gradient next: no clone of the incoming gradient and no separate cast pass. */
int vs_layernorm_backward_ex(const void *dout, int64_t ld_do, int32_t do_dtype, const float *x, int64_t ldx, const float *w, const float *b,
                             const float *scale, int32_t mod_rows, int32_t mod_ld, float *dx, int64_t ld_dx, const float *dx_add,
                             int64_t ld_add, void *dx16, int64_t ld16, int32_t dx16_dtype, float *dw, float *db, float *dscale,
                             float *dshift, int32_t M, int32_t C, float eps, int32_t grp_in, int32_t grp_out, int32_t grp_off,
                             vs_stream_t stream);
int vs_layernorm_backward(const void *dout, int64_t ld_do, int32_t do_dtype, const float *x, int64_t ldx, const float *w, const float *b,
                          const float *scale, int32_t mod_rows, int32_t mod_ld, float *dx, int64_t ld_dx, int32_t accumulate_dx,
                          float *dw, float *db, float *dscale, float *dshift, int32_t M, int32_t C, float eps, int32_t grp_in,
                          int32_t grp_out, int32_t grp_off, vs_stream_t stream);

/* Backward of vs_gaussian_adapter for dense NHWC 16-bit head outputs (training): gradients of means [npix,3], covariances [npix,3,3],
 * harmonics [npix,3,d_sh], opacities [npix] and (nullable) of the raw output [npix, 11 + 3 d_sh] -> d_pts [npix, d_pts_ld], d_gs
 * [npix, d_gs_ld] in the inputs' dtype (0 f32 -- the split / f32 classes --, 1 f16, 2 bf16); the row strides may exceed the channel counts (padding columns are zeroed).  Backward of MyGaussianAdapter.forward + the 'exp' depth post-process
 * (common/gaussian_adapter.py:168-212, heads/postprocess.py:46-56) in one pass. */
int vs_gaussian_adapter_backward(const void *pts, int32_t pts_pix, const void *gs, int32_t in_dtype, int64_t npix, int32_t d_sh,
                                 const float *sh_mask, int32_t scale_act, float scale_min, float scale_max, float opacity_exponent,
                                 const float *d_means, const float *d_cov, const float *d_harmonics, const float *d_opacities,
                                 const float *d_raw, void *d_pts, int32_t d_pts_ld, void *d_gs, int32_t d_gs_ld, vs_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Split-class BACKWARD (round 3; the reference trains in fp32 / TF32: model_wrapper.py:184-321, config/experiment/re10k_8view.yaml:75-80).
 * Every activation and gradient is an f32 tensor; every MFMA product is three f16 MFMAs on (hi, lo) pairs (dtype 4 above).
 *   dX = dY W        vs_gemm_split on the packed transposed weight
 *   dW = dY^T X      vs_gemm_wgrad(dtype 4): A = vs_transpose_f32(dY) [N, rows], W = vs_transpose_pack_split(X) [K, rows] (packed)
 *   conv 3x3         dgrad = vs_conv3x3_split_nhwc on the flipped weights; wgrad = nine tap GEMMs, W = vs_transpose_pack_split with the
 *                    tap (and the ResidualConvUnit's ReLU) applied on the way
 *   attention        vs_attention_backward_split on the (hi, lo) images vs_split16 writes
 * Gradients are kept in the f16 range by the caller's power-of-two loss scale (exact in f32); lo degrades gracefully into f16
 * subnormals (absolute floor 2^-25 per element).
 * ------------------------------------------------------------------------------------------------ */
/* out[c][r] = act(in[src(r)][c]) for r < R, 0 for R <= r < Rpad (Rpad % 64 == 0, ld_out >= Rpad, ld_out % 4 == 0); relu != 0: act = max(., 0).
 * conv_H, conv_W > 0: r indexes the pixels of whole H x W images and src(r) = r + tap_dy * W + tap_dx when that pixel is inside the
 * image (zero otherwise) -- the shifted operand of one tap of a 3x3 convolution's weight gradient; 0, 0: src(r) = r.  tap_dy = 2: r
 * indexes the pixels of zero-BORDERED (H + 2) x (W + 2) maps (R = their total), the interior read from the unpadded tensor `in` -- the
 * operands of the tap-fused weight gradient vs_gemm_wgrad(dtype 4, ntaps = 9), whose tap shifts move the f32 A operand.
 * colsum (nullable, f32 [C], overwritten): the column sums of `in` over its R rows -- the bias gradient rides on the transpose of dY that
 * the weight gradient needs anyway (as vs_transpose16_ex; plain and bordered forms only, relu == 0; partial sums meet through f32 atomics). */
int vs_transpose_f32(const float *in, int64_t ld_in, float *out, int64_t ld_out, int32_t R, int32_t C, int32_t Rpad, int32_t relu, int32_t conv_H,
                     int32_t conv_W, int32_t tap_dy, int32_t tap_dx, float *colsum, vs_stream_t stream);
/* The same, written as the packed split operand of vs_gemm_split / vs_gemm_wgrad(dtype 4): out [C, ld_out] 4-byte units, rows scaled by
 * 2^scale_exp (0 for activations), layout of vs_split_pack_weight. */
int vs_transpose_pack_split(const float *in, int64_t ld_in, void *out, int64_t ld_out, int32_t R, int32_t C, int32_t Rpad, int32_t relu,
                            int32_t conv_H, int32_t conv_W, int32_t tap_dy, int32_t tap_dx, int32_t scale_exp, float *colsum, vs_stream_t stream);
/* Split-class weight gradient with ONE operand read as it is in memory (reduction-major, no transposed copy): out32[M, N] (+)= sum_{k < Kred}
 * A[k, m] Wp[n, k], A [Kred, M] f32 (row stride lda floats), Wp = vs_transpose_pack_split of the other operand, [N, Kpad] (row stride ldw 4-byte
 * units, zero beyond Kred).  M, N multiples of 256, Kpad a multiple of 64 * ksplit; workspace (>= ksplit * M * N floats) / accumulate as in
 * vs_gemm_wgrad.  transpose_out != 0: out is [N, M] (row stride ldo), the transpose of the product (workspace required).  nn.Linear: A = X
 * [tokens, K], Wp = pack(dY^T) (whose pass also yields the bias gradient, colsum), transpose_out = 1 -> dW [N, K]: the vs_transpose_f32 /
 * vs_transpose_pack_split pass over X is gone (model_wrapper.py:184-321 differentiates these layers with torch autograd). */
int vs_gemm_wgrad_split_atn(const float *A, const void *Wp, float *out, int32_t M, int32_t N, int32_t Kred, int32_t Kpad, int32_t lda, int32_t ldw,
                            int32_t ldo, int32_t ksplit, int32_t transpose_out, void *workspace, int64_t workspace_bytes, int32_t accumulate,
                            vs_stream_t stream);
/* The 3x3 convolution's weight gradient on the same main loop: out32[tap][ci][co] (+)= sum over pixels p of act(x)[p + tap offset][ci] dy[p][co] (zero
 * outside the image), x [Nimg,H,W,Cin] f32 NHWC as it is (no zero-bordered transposed copy), dyTp = vs_transpose_pack_split(dy [pixels, Cout])
 * [Cout, Ppad]; Cin, Cout multiples of 256, Ppad % (64 * ksplit) == 0; workspace >= ksplit * 9 * Cin * Cout floats or null (atomics). */
int vs_conv3x3_wgrad_split_atn(const float *x, const void *dyTp, float *out, int32_t Nimg, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t Ppad,
                               int32_t ldw, int32_t relu_in, int32_t ksplit, void *workspace, int64_t workspace_bytes, int32_t accumulate,
                               vs_stream_t stream);
/* hi = rne16(x), lo = rne16(x - hi) as two 16-bit images [rows, ld_out] of the f32 tensor in [rows, ld_in] (C columns, C % 4 == 0). */
int vs_split16(const float *in, int64_t ld_in, void *hi, void *lo, int64_t ld_out, int64_t rows, int32_t C, vs_stream_t stream);
/* Backward of vs_attention(_lse) with dtype 4 (same addressing, mask and key segments as vs_attention_backward).  *_hi / *_lo: the
 * vs_split16 images of q, k, v, dout (row strides ldq / ldk / ldv / lddo in 16-bit elements, shared by hi and lo); o, dout: the f32
 * tensors (row strides ldo32 / lddo32, floats) for delta; lse from vs_attention_lse; delta [rows, H] scratch.  dq f32 [rows, lddq]
 * written; dk, dv f32 by key row: written when kv_seg is null, ADDED to with f32 atomics otherwise (zero them first). */
int vs_attention_backward_split(const void *q_hi, const void *q_lo, const void *k_hi, const void *k_lo, const void *v_hi, const void *v_lo,
                                const void *do_hi, const void *do_lo, const float *o, const float *dout, const float *lse, float *delta, float *dq,
                                float *dk, float *dv, int32_t nbatch, int32_t H, int32_t Lq, int32_t Lk, int64_t q_batch_rows, int64_t k_batch_rows,
                                int32_t ldq, int32_t ldk, int32_t ldv, int32_t lddo, int32_t ldo32, int32_t lddo32, int32_t lddq, int32_t lddk,
                                int32_t lddv, const int32_t *kv_seg, const int32_t *q_kvlen, int32_t max_keys, float scale, vs_stream_t stream);
/* f32 forms of the element-wise operators of the training step (n % 4 == 0, 16-byte aligned): exact-erf GELU and its backward, the ReLU
 * mask out = x > 0 ? dy : 0, the gated residual update / its backward with an f32 branch (vs_gated_resid / _backward), and the transpose of
 * the bilinear x2 interpolation (vs_upsample2x_backward_nhwc). */
int vs_gelu_f32(const float *z, float *out, int64_t n, vs_stream_t stream);
int vs_gelu_backward_f32(const float *dy, const float *z, float *dz, int64_t n, vs_stream_t stream);
int vs_relu_mask_f32(const float *dy, const float *x, float *out, int64_t n, vs_stream_t stream);
int vs_gated_resid_f32(const float *x, const float *y, int64_t ldy, const float *gate, int32_t gate_rows, float *out, int32_t M, int32_t C,
                       int32_t grp_in, int32_t grp_out, int32_t grp_off, vs_stream_t stream);
int vs_gated_resid_backward_f32(const float *dout, const float *y, int64_t ldy, const float *gate, int32_t gate_rows, float *dy, int64_t lddy,
                                float *dgate, int32_t M, int32_t C, int32_t grp_in, int32_t grp_out, int32_t grp_off, vs_stream_t stream);
int vs_upsample2x_backward_f32_nhwc(const float *dout, float *din, int32_t Nimg, int32_t H, int32_t W, int32_t C, vs_stream_t stream);
/* im2col rows of the Gaussian-parameter head's 7x7 / pad 3 RGB stem (dpt_gs_head.py:112-118) for the training path, where the stem runs as a GEMM
 * over them: out[p, k] = frames[n, c, y + ky - 3, x + kx - 3] (zero outside the image) for k = c * 49 + ky * 7 + kx < 147 (= weight.flatten(1) order),
 * zeros for 147 <= k < ld.  frames [N, 3, H, W] f32 contiguous; out [N*H*W, ld] in out_dtype (0 f32, 1 f16, 2 bf16), ld % 8 == 0, 16-byte aligned.
 * F.unfold(..., 7, padding=3).transpose(1, 2) + F.pad in one pass. */
int vs_im2col7x7_rgb(const float *frames, void *out, int32_t N, int32_t H, int32_t W, int32_t ld, int32_t out_dtype, vs_stream_t stream);
/* Backward of the LAST 1x1 convolution of a DPT head fused with the ReLU backward of its input (split class, one pass over the full-resolution
 * tensors): t = relu(conv3x3(...)) [P, Cin], y = t W^T + b [P, Cout] (heads/dpt_block.py:316-343 head.2 -> head.4; dpt_gs_head.py:120-157) ->
 *   dt[p, c] = (relu ? t[p, c] > 0 : 1) * sum_n dy[p, n] W[n, c]      dW[n, c] = sum_p dy[p, n] t[p, c]      db[n] = sum_p dy[p, n]
 * dy [P, Cout] f32 with row stride ldy (Cout <= ldy <= 128; padding columns are not read as data), t [P, Cin], dt [P, Cin] f32 contiguous, all 16-byte aligned, W [Cout, Cin] f32 (the module's parameter, packed in the kernel
 * with scale 2^w_scale_exp).  Cin = 128 | 256, Cout <= 96, P % 32 == 0.  nwg persistent workgroups; dW / db leave as per-workgroup partial
 * sums dw_part [nwg, R, Cin], db_part [nwg, R] (R = 96 if Cout > 16 else 16; rows >= Cout are padding) which the caller adds up --
 * deterministic.  Replaces autograd's nn.Conv2d(k=1) backward + the ReLU backward of the reference's training step (model_wrapper.py:207-230). */
int vs_head1x1_backward_split(const float *dy, int64_t ldy, const float *t, const float *w, int32_t w_scale_exp, float *dt, float *dw_part,
                              float *db_part, int64_t P, int32_t Cin, int32_t Cout, int32_t relu, int32_t nwg, vs_stream_t stream);
/* The Gaussian-parameter head's 7x7 RGB stem fused with its upsample-add (dpt_gs_head.py:112-118,142-150), split class, streaming form of
 * vs_conv7x7_rgb_split_up_nhwc: out = packed (hi, lo) rows [N*H*W, Cout] of bilinear_x2(trunk [N, H/2, W/2, Cout] f32) + relu(conv7x7(image) + bias).
 * img_padded: zero-bordered NHWC f32 frames [N, Hp, Wp, 3] (3 pixels before, >= 3 after); w: the module's [Cout, 3, 7, 7] f32 parameter, packed in
 * the kernel with scale 2^w_scale_exp.  Cout == 256, H even, W % 32 == 0; nwg persistent workgroups (one per CU). */
int vs_stem7x7_up_split_stream(const float *img_padded, const float *w, int32_t w_scale_exp, const float *bias, const float *trunk, void *out,
                               int32_t N, int32_t H, int32_t W, int32_t Hp, int32_t Wp, int32_t Cout, int32_t nwg, vs_stream_t stream);
/* Weight (and bias) gradient of a 3x3 convolution (stride 1, pad 1), split class, as ONE streaming pass over x [N,H,W,Cin] and dy [N,H,W,Cout]
 * (f32 NHWC as they are: no transposed / bordered copies) for the narrow layers of the DPT heads (heads/dpt_block.py:316-343; Cin, Cout multiples of
 * 64, W a multiple of 32): dw_part [workers, 9, Cin, Cout] (tap = ky * 3 + kx), db_part [workers, Cout] = per-worker partial sums the caller adds
 * up (deterministic); relu_in = the convolution read relu(x).  workers: a multiple of 8; the launch has workers * (Cin / 64) * (Cout / 64)
 * persistent workgroups -- choose it so that they are all resident (one per CU). */
int vs_conv3x3_wgrad_split_stream(const float *x, const float *dy, float *dw_part, float *db_part, int32_t N, int32_t H, int32_t W, int32_t Cin,
                                  int32_t Cout, int32_t relu_in, int32_t workers, vs_stream_t stream);
/* The same in the 16-bit operand classes: dy [P, Cout] (row stride ldy elements), t, dt [P, Cin] f16 (dtype 1) or bf16 (2), W f32
 * (converted in the kernel), dw_part / db_part f32 as above. */
int vs_head1x1_backward16(const void *dy, int64_t ldy, const void *t, const float *w, void *dt, float *dw_part, float *db_part, int64_t P,
                          int32_t Cin, int32_t Cout, int32_t relu, int32_t nwg, int32_t dtype, vs_stream_t stream);

/* Measurement aid of bench.py (`roofline.sustained_mfma_tflops`): back-to-back v_mfma_f32_16x16x32_f16 on register operands filled from
 * `operands` (>= 1 MiB of f16 data), no memory traffic in the loop, 512 workgroups x 4 waves x iters x 16 MFMAs; scratch >= 131072 floats.
 * Asynchronous on `stream`; *flop_out_host receives the FLOP count of the launch.  Its rate is what the chip sustains on the matrix pipe
 * alone (power-limited: ~0.67 of the 2.5 PFLOP/s dense-f16 headline on MI355X). */
int vs_probe_mfma_rate(const void *operands, float *scratch, int32_t iters, double *flop_out_host, vs_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Range guard of the split operand class (ABI 5, round 4).  The reference stores fp32 and multiplies in TF32 (backbone_vica.py:9):
 * fp32 RANGE.  The split class multiplies f16 (hi, lo) pairs of the f32 activations, unscaled: |x| >= 65520 turns hi into +-inf.
 * vs_range_check scans one activation operand and ORs into flags[slot]: 1 = a finite value with |x| >= limit (pass 65520), 2 = a
 * non-finite f32 value, 4 (kind 1) = a hi half that is already +-inf / NaN.  kind 0: f32 [rows, cols], row stride ld floats, cols % 4 == 0;
 * kind 1: the packed (hi, lo) layout of vs_split_pack_weight / the "+16" producers, [rows, cols] 4-byte units, cols % 32 == 0.  One
 * atomic per wavefront that found something; flags: int32 device array owned by the caller (zeroed by the caller).  A debug aid: the
 * product kernels never call it (vicasplat_amd.ops.range_guard drives it around every split-class GEMM / convolution / attention call).
 * ------------------------------------------------------------------------------------------------ */
int vs_range_check(const void *x, int64_t rows, int32_t cols, int64_t ld, int32_t kind, float limit, int32_t *flags, int32_t slot,
                   vs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VICASPLAT_HIP_H */
