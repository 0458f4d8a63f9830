// Shared device helpers for the gfx950 kernels (wave = 64 lanes, fp32 rows of H features).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ALIGNN_WAVE 64
#define ALIGNN_EPS_GATE 1e-6f

#define ALIGNN_CHECK_LAUNCH()                      \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

static inline int alignn_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float fast_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ float4 f4_ld(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void f4_st(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) {
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 f4_sub(float4 a, float4 b) {
    return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
}
__device__ __forceinline__ float4 f4_mul(float4 a, float4 b) {
    return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
}
__device__ __forceinline__ float4 f4_fma(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 f4_sigmoid(float4 a) {
    return make_float4(fast_sigmoid(a.x), fast_sigmoid(a.y), fast_sigmoid(a.z), fast_sigmoid(a.w));
}

// silu(z) = z*s, silu'(z) = s*(1 + z*(1-s)) with s = sigmoid(z)
__device__ __forceinline__ float silu_f(float z) { return z * fast_sigmoid(z); }
__device__ __forceinline__ float dsilu_f(float z) {
    float s = fast_sigmoid(z);
    return s * (1.0f + z * (1.0f - s));
}
