// Micro-benchmark: sustained rate of v_mfma_f32_32x32x16_bf16 / 16x16x32 with W waves per SIMD, no memory traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
template <int NACC>
__global__ __launch_bounds__(512) void k32(float* out, int iters, int rnd) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    unsigned seed = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = 0; i < 8; ++i) {
        seed = seed * 1664525u + 1013904223u; a[i] = rnd ? (__bf16)(((int)(seed >> 8) & 0xffff) * (1.0f / 65536.f) - 0.5f) : (__bf16)(float)(threadIdx.x + i);
        seed = seed * 1664525u + 1013904223u; b[i] = rnd ? (__bf16)(((int)(seed >> 8) & 0xffff) * (1.0f / 65536.f) - 0.5f) : (__bf16)(float)(i);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        if (rnd) { a = __builtin_shufflevector(a, a, 1, 2, 3, 4, 5, 6, 7, 0); b = __builtin_shufflevector(b, b, 7, 0, 1, 2, 3, 4, 5, 6); }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(512) void k16(float* out, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
static void run(const char* name, F launch, double flops) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    launch(); hipDeviceSynchronize();
    hipEventRecord(s); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); ms /= 5;
    printf("%-40s %8.1f us  %8.1f TF\n", name, ms * 1e3, flops / (ms * 1e-3) / 1e12);
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 8 * 512 * 4);
    for (int rnd : {0, 1})
        for (int iters : {4000, 40000, 400000}) {
            const int blocks = 256, threads = 512;
            double waves = (double)blocks * threads / 64;
            char nm[128];
            snprintf(nm, 128, "32x32x16 bf16 acc8 rnd=%d iters=%d", rnd, iters);
            run(nm, [&] { hipLaunchKernelGGL(k32<8>, dim3(blocks), dim3(threads), 0, 0, out, iters, rnd); }, waves * iters * 8 * 32768.0);
        }
    return 0;
}
