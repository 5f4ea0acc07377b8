// Shared host-side helpers for libvicasplat_hip.so (gfx950 only; no CUDA / multi-backend paths).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../../include/vicasplat_hip.h"

namespace vs {

void set_error(const char *fmt, ...);

#define VS_CHECK(cond, ...)            \
    do {                               \
        if (!(cond)) {                 \
            vs::set_error(__VA_ARGS__); \
            return -1;                 \
        }                              \
    } while (0)

#define VS_HIP(expr)                                                                           \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            vs::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return -2;                                                                         \
        }                                                                                      \
    } while (0)

// VS_DETERMINISTIC=1 (read per call: tests flip it): no result may depend on the arrival order of f32 atomics -- the GEMM routes that
// split K over workgroups (residual epilogue) run unsplit or through the two-pass reduction instead (tests/test_encoder_gpu.py).
static inline bool deterministic() {
    const char *e = getenv("VS_DETERMINISTIC");
    return e && e[0] != '0' && e[0] != 0;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr int kTile = 16;           // 16x16 pixel tiles (upstream BLOCK_X/BLOCK_Y)
constexpr int kGeomFloats = 12;     // VS_BUF_GEOM record
constexpr int kCkShift = 9;         // VS_BUF_CHECKPOINT: the render kernel stores the blending state of a tile every 2^kCkShift list entries
constexpr int kCkSeg = 1 << kCkShift;
constexpr int kCkFloats = 5 * 256;  // T | Cr | Cg | Cb | D of the 256 pixels, in 2x2-block order

}  // namespace vs
