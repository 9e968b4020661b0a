// Shared helpers for the gfx950 kernels of libkgan_hip.so (see include/kgan_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kgan_hip.h"

void kg_set_error(const char* fmt, ...);

#define KG_REQUIRE(cond, ...)           \
    do {                                \
        if (!(cond)) {                  \
            kg_set_error(__VA_ARGS__);  \
            return -1;                  \
        }                               \
    } while (0)

static inline int kg_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        kg_set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

static inline int kg_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

typedef float kg_f32x16 __attribute__((ext_vector_type(16)));

// activation and its derivative expressed on the activation OUTPUT
__device__ __forceinline__ float kg_act(float v, int act, float slope) {
    if (act == KG_ACT_LRELU) return v > 0.f ? v : v * slope;
    if (act == KG_ACT_TANH) return tanhf(v);
    return v;
}
__device__ __forceinline__ float kg_dact_from_out(float o, int act, float slope) {
    if (act == KG_ACT_LRELU) return o > 0.f ? 1.f : slope;
    if (act == KG_ACT_TANH) return 1.f - o * o;
    return 1.f;
}
