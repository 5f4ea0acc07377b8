// Probe of v_permlane16_swap_b32 (gfx950): which lanes exchange what.  hipcc --offload-arch=gfx950 tools/probe/permlane_swap.hip -o /tmp/pl && /tmp/pl
#include <hip/hip_runtime.h>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned *o) {
    unsigned a = threadIdx.x, b = threadIdx.x + 100;
    u2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x * 2] = r.x; o[threadIdx.x * 2 + 1] = r.y;
}
int main() {
    unsigned *d; hipMalloc(&d, 64 * 2 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) if (l % 16 == 0 || l % 16 == 5) printf("lane %2d: V0=%3u V1=%3u\n", l, h[2 * l], h[2 * l + 1]);
    return 0;
}
