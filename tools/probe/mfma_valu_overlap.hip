// Do MFMA and VALU work overlap on a gfx950 SIMD?  2 workgroups x 4 waves per CU = 2 waves per SIMD.  Modes:
//   0  every wave: MFMA only (16 independent accumulators)          1  every wave: VALU only (v_exp_f32 + fma chains, 4 independent chains)
//   2  every wave: MFMA block then VALU block, alternating (what a softmax-between-matmuls wave does; no intra-wave interleave)
//   3  every wave: MFMAs and VALU ops interleaved 1 : 3 in program order (independent of each other)
//   4  one 8-wave workgroup per CU: waves 0-3 MFMA only, waves 4-7 VALU only (wave i and i + 4 share a SIMD: the perfect ping-pong)
//   7 / 8  as 4 / 6 with 48 plain FMAs instead of the mix; 9 / 10 with 16 v_exp_f32 only
//   11 .. 14  ONE wave per SIMD alone: MFMA only / the mix / FMAs / v_exp (what each half of the ping-pong costs by itself)
//   15  MFMA wave + integer-VALU wave; 16 integer VALU in all 8 waves; 17 integer VALU, one wave per SIMD alone
//   1xx  mode xx with s_setprio 3 in the VALU waves (4-7)
//   5 / 6  the same launch shape with all 8 waves MFMA only / VALU only (references for mode 4)
// hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define MFMA16(c) _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) c[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], c[i * 4 + j], 0, 0, 0);
// 48 VALU: 4 chains x (8 fma + 4 exp)
#define VALU48(x) _Pragma("unroll") for (int r = 0; r < 4; ++r) _Pragma("unroll") for (int q = 0; q < 4; ++q) { x[q] = x[q] * 0.999f + 0.001f; x[q] = x[q] * 1.0001f - 0.0001f; x[q] = __builtin_amdgcn_exp2f(x[q] * 0.01f); }

#define FMA48(x) _Pragma("unroll") for (int r = 0; r < 4; ++r) _Pragma("unroll") for (int q = 0; q < 4; ++q) { x[q] = x[q] * 0.999f + 0.001f; x[q] = x[q] * 1.0001f - 0.0001f; x[q] = x[q] * 0.9999f + 0.0002f; }
#define INT48(y) _Pragma("unroll") for (int r = 0; r < 4; ++r) _Pragma("unroll") for (int q = 0; q < 4; ++q) { y[q] = y[q] * 3 + 7; y[q] ^= y[q] >> 3; y[q] += 11; }
#define EXP16(x) _Pragma("unroll") for (int r = 0; r < 4; ++r) _Pragma("unroll") for (int q = 0; q < 4; ++q) { x[q] = __builtin_amdgcn_exp2f(x[q]); }
template <int MODE, int PRIO = 0>
__global__ void __launch_bounds__(512, 1) k(const half8 *in, float *out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[(t * 8 + i) & 65535]; b[i] = in[(t * 8 + 4 + i) & 65535]; }
    f4 c[16];
    for (int i = 0; i < 16; ++i) c[i] = f4{0, 0, 0, 0};
    float x[4] = {(float)a[0][0], (float)a[0][1], (float)a[0][2], (float)a[0][3]};
    int y[4] = {t, t + 1, t + 2, t + 3};
    const bool do_i = (MODE == 15 && threadIdx.x >= 256) || MODE == 16 || (MODE == 17 && threadIdx.x >= 256);
    if (PRIO && threadIdx.x >= 256) __builtin_amdgcn_s_setprio(3);
    const long long t0 = __builtin_readcyclecounter();
    const bool do_m = MODE == 0 || MODE == 2 || MODE == 3 || ((MODE == 4 || MODE == 7 || MODE == 9 || MODE == 11 || MODE == 15) && threadIdx.x < 256) || MODE == 5;
    const bool do_v = MODE == 1 || MODE == 2 || MODE == 3 || ((MODE == 4 || MODE == 12) && threadIdx.x >= 256) || MODE == 6;
    const bool do_f = ((MODE == 7 || MODE == 13) && threadIdx.x >= 256) || MODE == 8, do_e = ((MODE == 9 || MODE == 14) && threadIdx.x >= 256) || MODE == 10;
    if (MODE == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    c[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], c[i * 4 + j], 0, 0, 0);
                    const int q = j;
                    x[q] = x[q] * 0.999f + 0.001f; x[q] = x[q] * 1.0001f - 0.0001f; x[q] = __builtin_amdgcn_exp2f(x[q] * 0.01f);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            if (do_m) { MFMA16(c) }
            __builtin_amdgcn_sched_barrier(0);
            if (do_v) { VALU48(x) }
            if (do_f) { FMA48(x) }
            if (do_e) { EXP16(x) }
            if (do_i) { INT48(y) }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (t == 0) reinterpret_cast<long long *>(out + 256 * 4 * 256)[0] = t1 - t0;
    float s = x[0] + x[1] + x[2] + x[3] + (float)(y[0] + y[1] + y[2] + y[3]);
    for (int i = 0; i < 16; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[t] = s;
}

int main() {
    std::vector<_Float16> h(65536 * 8);
    srand(1);
    for (auto &v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
    half8 *din; float *dout;
    hipMalloc(&din, h.size() * 2); hipMalloc(&dout, 256 * 4 * 256 * 4 + 64);
    hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 50000, blocks = 256 * 2;
    for (int mode : {7, 107, 4, 104, 15, 115})
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            switch (mode) {
                case 107: hipLaunchKernelGGL((k<7, 1>), dim3(256), dim3(512), 0, 0, din, dout, iters); break;
                case 104: hipLaunchKernelGGL((k<4, 1>), dim3(256), dim3(512), 0, 0, din, dout, iters); break;
                case 115: hipLaunchKernelGGL((k<15, 1>), dim3(256), dim3(512), 0, 0, din, dout, iters); break;
                case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, din, dout, iters); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, din, dout, iters); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, din, dout, iters); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, din, dout, iters); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, din, dout, iters); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(256), dim3(512), 0, 0, din, dout, iters); break;
                case 6: hipLaunchKernelGGL(k<6>, dim3(256), dim3(512), 0, 0, din, dout, iters); break;
                case 7: hipLaunchKernelGGL(k<7>, dim3(256), dim3(512), 0, 0, din, dout, iters); break;
                case 8: hipLaunchKernelGGL(k<8>, dim3(256), dim3(512), 0, 0, din, dout, iters); break;
                case 9: hipLaunchKernelGGL(k<9>, dim3(256), dim3(512), 0, 0, din, dout, iters); break;
                case 10: hipLaunchKernelGGL(k<10>, dim3(256), dim3(512), 0, 0, din, dout, iters); break;
                case 11: hipLaunchKernelGGL(k<11>, dim3(256), dim3(512), 0, 0, din, dout, iters); break;
                case 12: hipLaunchKernelGGL(k<12>, dim3(256), dim3(512), 0, 0, din, dout, iters); break;
                case 13: hipLaunchKernelGGL(k<13>, dim3(256), dim3(512), 0, 0, din, dout, iters); break;
                case 14: hipLaunchKernelGGL(k<14>, dim3(256), dim3(512), 0, 0, din, dout, iters); break;
                case 15: hipLaunchKernelGGL(k<15>, dim3(256), dim3(512), 0, 0, din, dout, iters); break;
                case 16: hipLaunchKernelGGL(k<16>, dim3(256), dim3(512), 0, 0, din, dout, iters); break;
                default: hipLaunchKernelGGL(k<17>, dim3(256), dim3(512), 0, 0, din, dout, iters); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long cyc = 0; hipMemcpy(&cyc, (char *)dout + 256 * 4 * 256 * 4, 8, hipMemcpyDeviceToHost);
            if (rep) printf("[%.2f GHz] ", cyc / (ms * 1e6));
            if (rep) printf("mode %d: %.2f ms  (per trip and wave: %.0f ns; 16 MFMAs = 256 matrix cycles, 48 VALU incl. 16 v_exp)\n", mode, ms, ms * 1e6 / iters);
        }
    return 0;
}
