// FETCH_SIZE calibration on gfx950 for the two read patterns of the rasterizer ("calibrate FETCH_SIZE on the 16-B
// gather pattern before quoting traffic").  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/probe/fetch_calib.hip -o /tmp/fetch_calib
//   cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/fc -- /tmp/fetch_calib
// Kernels (known bytes):
//   stream_kernel : every lane reads consecutive 16-byte words of a 1 GiB buffer                      -> 1 GiB read
//   gather_kernel : lane i reads the three 16-byte words of the 48-byte record perm[i] of a 768 MiB
//                   table (random permutation: no reuse, table > the 256 MiB Infinity Cache)         -> N * 48 B requested,
//                   N * (lines touched) * 64 B moved: a 48-B record at a 48-B stride covers 1.5 64-byte lines on average
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void stream_kernel(const float4 *__restrict__ x, float *__restrict__ out, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0.f;
    for (; i < n; i += (long long)gridDim.x * blockDim.x) { const float4 v = x[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) out[0] = s;
}

__global__ void gather_kernel(const float4 *__restrict__ rec, const unsigned *__restrict__ perm, float *__restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t g = perm[i];
    const float4 a = rec[g * 3], b = rec[g * 3 + 1], c = rec[g * 3 + 2];
    const float s = a.x + b.y + c.z;
    if (s == 12345.678f) out[0] = s;
}

int main() {
    const long long nstream = (1LL << 30) / 16;
    const long long nrec = 16LL << 20;  // 16 M records x 48 B = 768 MiB
    float4 *x, *rec; unsigned *perm; float *out;
    hipMalloc(&x, nstream * 16); hipMalloc(&rec, nrec * 48); hipMalloc(&perm, nrec * 4); hipMalloc(&out, 64);
    hipMemset(x, 0, nstream * 16); hipMemset(rec, 0, nrec * 48);
    std::vector<unsigned> h(nrec);
    for (long long i = 0; i < nrec; ++i) h[i] = (unsigned)i;
    unsigned long long st = 88172645463325252ULL;
    for (long long i = nrec - 1; i > 0; --i) {  // Fisher-Yates with xorshift64
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        const long long j = (long long)(st % (unsigned long long)(i + 1));
        const unsigned t = h[i]; h[i] = h[j]; h[j] = t;
    }
    hipMemcpy(perm, h.data(), nrec * 4, hipMemcpyHostToDevice);
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(stream_kernel, dim3(256 * 8), dim3(256), 0, 0, x, out, nstream);
        hipLaunchKernelGGL(gather_kernel, dim3((unsigned)(nrec / 256)), dim3(256), 0, 0, rec, perm, out, nrec);
    }
    hipDeviceSynchronize();
    printf("stream_kernel: %lld bytes requested per launch; gather_kernel: %lld bytes requested (+ %lld of indices), %lld records\n",
           nstream * 16, nrec * 48, nrec * 4, nrec);
    return 0;
}
