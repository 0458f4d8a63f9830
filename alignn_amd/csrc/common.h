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

// sigmoid with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division (v_div_scale x2, v_rcp, four fma,
// v_div_fmas, v_div_fixup: ten instructions per element, 40 % of the line-graph backward kernel's VALU work); the exponential
// is __expf (2 ulp) already.  Every kernel - forward, recomputation in backward, the oracle comparison - goes through this
// one function.
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// flat element index i -> (row, quad) of a [rows, Q quads] matrix.  Q is a run-time value, so `i / Q` on a 64-bit index
// is a ~100-instruction emulated division PER ELEMENT in kernels that otherwise do a handful of FMAs; feature counts are
// powers of two almost everywhere (256, 64, 1024): shift and mask then, 32-bit division when the index fits, the general
// case last.  (Wave-uniform branches.)
struct RowQuad {
    int Q, shift;  // shift = log2(Q) when Q is a power of two, else -1
    __device__ __forceinline__ explicit RowQuad(int q) : Q(q), shift((q & (q - 1)) == 0 ? __ffs(q) - 1 : -1) {}
    __device__ __forceinline__ void split(int64_t i, int64_t total, int64_t& r, int& q) const {
        if (shift >= 0) {
            r = i >> shift;
            q = (int)(i & (Q - 1));
        } else if (total < ((int64_t)1 << 31)) {
            const unsigned ri = (unsigned)i / (unsigned)Q;
            r = ri;
            q = (int)((unsigned)i - ri * (unsigned)Q);
        } else {
            r = i / Q;
            q = (int)(i - r * Q);
        }
    }
};

__device__ __forceinline__ float4 f4_ld(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void f4_st(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// streaming (read-once / write-once) variants: nontemporal hint, measured +9 % on a 2-read 1-write pass over 2 GB
typedef float alignn_v4f __attribute__((ext_vector_type(4)));
template <bool STREAM>
__device__ __forceinline__ float4 f4_lds(const float* p) {
    if (!STREAM) return *reinterpret_cast<const float4*>(p);
    alignn_v4f v = __builtin_nontemporal_load(reinterpret_cast<const alignn_v4f*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
template <bool STREAM>
__device__ __forceinline__ void f4_sts(float* p, float4 r) {
    if (!STREAM) {
        *reinterpret_cast<float4*>(p) = r;
        return;
    }
    alignn_v4f v = {r.x, r.y, r.z, r.w};
    __builtin_nontemporal_store(v, reinterpret_cast<alignn_v4f*>(p));
}
__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) {
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 f4_sub(float4 a, float4 b) {
    return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
}
__device__ __forceinline__ float4 f4_scale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
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

// Running max|x| of a tensor, tracked by the kernel that produces it (consumed by the f16x3 GEMMs to pick their
// power-of-two operand scale).  Wave max by DPP-free shuffles, then ONE atomicMax per workgroup on the bit pattern:
// non-negative floats order like unsigned ints and max is order independent, so the result is deterministic.
// Every thread of the workgroup must call it (it contains a barrier); `amax` may be null.
__device__ __forceinline__ void block_amax_commit(float m, float* amax) {
    if (amax == nullptr) return;  // uniform
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    __shared__ float amax_sh[16];
    const int wave = threadIdx.x >> 6, nw = (blockDim.x * blockDim.y + 63) >> 6;
    if ((threadIdx.x & 63) == 0) amax_sh[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = amax_sh[0];
        for (int w = 1; w < nw; ++w) r = fmaxf(r, amax_sh[w]);
        // most workgroups find the running maximum already above their own: look before paying for the atomic (a
        // stale read only costs a redundant atomicMax)
        if (r > 0.0f && r > *reinterpret_cast<volatile float*>(amax))
            atomicMax(reinterpret_cast<unsigned*>(amax), __float_as_uint(r));
    }
    __syncthreads();  // the scratch array may be reused by a second commit
}
__device__ __forceinline__ float f4_absmax(float4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }
