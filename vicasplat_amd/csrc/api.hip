// Error channel + ABI version of libvicasplat_hip.so.
#include "common.h"

namespace vs {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace vs

extern "C" const char *vs_last_error(void) { return vs::g_err; }
extern "C" int vs_abi_version(void) { return 3; }   // 2: VsRasterIn.capacity, VS_BUF_DEPTH (round 2); 3: split operands (dtype 4, vs_gemm_split, ...)
