// What streaming pattern reaches the HBM rate on MI355X?  y = a + silu(b) over T x 256 floats (2 reads + 1 write,
// the shape of bn_silu_fwd), with different grid sizes / loads in flight / cache policies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_ld(const float4* p) { v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void nt_st(float4 r, float4* p) { v4f v = {r.x, r.y, r.z, r.w}; __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p)); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d line %d\n", (int)e_, __LINE__); return 1; } } while (0)
__device__ __forceinline__ float4 op(float4 a, float4 b) {
    return make_float4(a.x + b.x / (1.f + __expf(-b.x)), a.y + b.y / (1.f + __expf(-b.y)), a.z + b.z / (1.f + __expf(-b.z)), a.w + b.w / (1.f + __expf(-b.w)));
}
template <int U, int NT>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ y, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        float4 va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            va[u] = NT ? nt_ld(a + i + u * stride) : a[i + u * stride];
            vb[u] = NT ? nt_ld(b + i + u * stride) : b[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float4 r = op(va[u], vb[u]);
            if (NT) nt_st(r, y + i + u * stride); else y[i + u * stride] = r;
        }
    }
    for (; i < n; i += stride) y[i] = op(a[i], b[i]);
}
// block-contiguous variant: each block owns a contiguous chunk (better DRAM page locality per CU?)
template <int U>
__global__ __launch_bounds__(256) void kc(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ y, int64_t n) {
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t beg = per * blockIdx.x, end = beg + per < n ? beg + per : n;
    int64_t i = beg + threadIdx.x;
    for (; i + (U - 1) * 256 < end; i += U * 256) {
        float4 va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { va[u] = a[i + u * 256]; vb[u] = b[i + u * 256]; }
#pragma unroll
        for (int u = 0; u < U; ++u) y[i + u * 256] = op(va[u], vb[u]);
    }
    for (; i < end; i += 256) y[i] = op(a[i], b[i]);
}
template <typename F>
static float timeit(F f) {
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    for (int i = 0; i < 2; ++i) f();
    (void)hipEventRecord(s); for (int i = 0; i < 10; ++i) f(); (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e); return ms / 10;
}
int main() {
    const int64_t n = (int64_t)676200 * 64;  // float4 elements
    float4 *a, *b, *y;
    CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&y, n * 16));
    CK(hipMemset(a, 0, n * 16)); CK(hipMemset(b, 0, n * 16));
    const double gb = 3.0 * n * 16 / 1e9;
#define RUN(NAME, KERN, GRID) { float ms = timeit([&] { hipLaunchKernelGGL(KERN, dim3(GRID), dim3(256), 0, 0, a, b, y, n); }); printf("%-34s grid %7d: %7.1f us  %6.0f GB/s\n", NAME, (int)(GRID), ms * 1e3, gb / (ms * 1e-3)); }
    for (int grid : {1024, 2048, 4096, 8192, 16384, 65536}) {
        RUN("strided U=1", (k<1, 0>), grid);
        RUN("strided U=2", (k<2, 0>), grid);
        RUN("strided U=4", (k<4, 0>), grid);
        RUN("strided U=4 nontemporal", (k<4, 1>), grid);
        RUN("block-contiguous U=4", (kc<4>), grid);
    }
    RUN("one float4 per thread", (k<1, 0>), (int)((n + 255) / 256));
    RUN("one float4 per thread nt", (k<1, 1>), (int)((n + 255) / 256));
    return 0;
}
