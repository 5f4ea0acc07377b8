// Probe of ds_read_b64_tr_b16 on gfx950: LDS holds u16 element e at byte 2e; lane l supplies byte address A(l) and receives 4
// halfs.  Prints, for two address patterns, which LDS elements every lane got -- the mapping a TN (reduction-major) GEMM operand
// read needs.  Build: hipcc --offload-arch=gfx950 -o tools/probe/tr_read tools/probe/tr_read.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned short *out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    // mode 0: lane l -> 8 consecutive bytes at 8*l (a [4 rows][16 cols] block per 16 lanes when rows are 32 B = 16 halfs)
    // mode 1: rows of 256 halfs (512 B): lane t of a 16-lane group -> row (t>>2) of the group's 4 rows, cols (t&3)*4; group g -> rows 4g..4g+3
    unsigned addr;
    if (mode == 0) addr = 8u * l;
    else { const int g = l >> 4, t = l & 15; addr = (unsigned)(((4 * g + (t >> 2)) * 256 + (t & 3) * 4) * 2); }
    typedef void __attribute__((address_space(3))) *lptr_t;
    const unsigned base = (unsigned)(size_t)(lptr_t)lds;
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base + addr) : "memory");
    out[l * 4 + 0] = (unsigned short)(v & 0xffff); out[l * 4 + 1] = (unsigned short)((v >> 16) & 0xffff);
    out[l * 4 + 2] = (unsigned short)((v >> 32) & 0xffff); out[l * 4 + 3] = (unsigned short)(v >> 48);
}
int main() {
    unsigned short *d, h[256];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
        printf("mode %d (lane: elements received)\n", mode);
        for (int l = 0; l < 64; ++l) {
            if (mode == 0) printf("%2d: %4d %4d %4d %4d%s", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3], (l & 3) == 3 ? "\n" : "   ");
            else printf("%2d: (r%2d,c%2d) (r%2d,c%2d) (r%2d,c%2d) (r%2d,c%2d)%s", l, h[4*l]/256, h[4*l]%256, h[4*l+1]/256, h[4*l+1]%256, h[4*l+2]/256, h[4*l+2]%256, h[4*l+3]/256, h[4*l+3]%256, (l & 1) ? "\n" : "   ");
        }
    }
    return 0;
}
