// Probe: does global_load_lds_dwordx4 accept 8-byte / 4-byte / 2-byte aligned global addresses on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned short *src, unsigned short *dst, int off_halfs) {
    __shared__ __attribute__((aligned(1024))) unsigned short sm[64 * 8];
    typedef void __attribute__((address_space(3))) *lptr_t;
    const unsigned lds = (unsigned)(size_t)(lptr_t)sm;
    const unsigned short *gp = src + off_halfs + threadIdx.x * 8;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_waitcnt vmcnt(0)" ::"s"(lds), "v"(gp) : "memory");
    __syncthreads();
    for (int i = 0; i < 8; ++i) dst[threadIdx.x * 8 + i] = sm[threadIdx.x * 8 + i];
}
int main() {
    std::vector<unsigned short> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (unsigned short)i;
    unsigned short *d, *o;
    hipMalloc(&d, 8192); hipMalloc(&o, 1024);
    hipMemcpy(d, h.data(), 8192, hipMemcpyHostToDevice);
    for (int off : {0, 8, 4, 2, 1, 3}) {
        hipMemset(o, 0xff, 1024);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, off);
        hipError_t e = hipDeviceSynchronize();
        std::vector<unsigned short> r(512);
        hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 512; ++i) bad += r[i] != (unsigned short)(i + off);
        printf("offset %d halfs (%d bytes): err=%d mismatches=%d first=%d,%d\n", off, off * 2, (int)e, bad, r[0], r[1]);
    }
    return 0;
}
