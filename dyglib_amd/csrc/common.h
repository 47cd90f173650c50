// Shared helpers for libdygnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/dygnn.h"

namespace dygnn {

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(dygnn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kWave = 64;   // CDNA4 wavefront

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace dygnn

#define DYGNN_REQUIRE(cond, ...)                                  \
    do {                                                          \
        if (!(cond)) {                                            \
            dygnn::set_error(__VA_ARGS__);                        \
            return DYGNN_E_INVALID;                               \
        }                                                         \
    } while (0)

#define DYGNN_HIP(expr)                                                                        \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            dygnn::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return DYGNN_E_HIP;                                                                \
        }                                                                                      \
    } while (0)

#define DYGNN_LAUNCH_CHECK() DYGNN_HIP(hipGetLastError())
