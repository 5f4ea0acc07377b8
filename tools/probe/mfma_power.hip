// Sustained (power-limited) MFMA rate of the two f16 shapes on register operands with random data: which shape does more work per joule?
// hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ void __launch_bounds__(256) k(const half8 *in, float *out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[(t * 8 + i) & 65535]; b[i] = in[(t * 8 + 4 + i) & 65535]; }
    float s = 0.f;
    if constexpr (SHAPE == 16) {
        f4 c[16];
        for (int i = 0; i < 16; ++i) c[i] = f4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) c[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], c[i * 4 + j], 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    } else {
        f16v c[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)      // 8 MFMAs of 32x32x16 = the flops of 16 MFMAs of 16x16x32
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) c[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i + 2 * r], b[j + 2 * r], c[i * 2 + j], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += c[i][e];
    }
    out[t] = s;
}

int main() {
    std::vector<_Float16> h(65536 * 8);
    srand(1);
    for (auto &v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
    half8 *din; float *dout;
    hipMalloc(&din, h.size() * 2); hipMalloc(&dout, 256 * 4 * 256 * 4);
    hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int shape : {16, 32, 16, 32}) {
        const int iters = 200000, blocks = 256 * 2;   // 2 workgroups of 4 waves per CU = 2 waves per SIMD
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, din, dout, iters);
            else hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(256), 0, 0, din, dout, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)blocks * 4 * iters * 16 * (2.0 * 16 * 16 * 32);
            printf("shape %dx%d: %.1f ms  %.0f TFLOP/s\n", shape, shape, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
