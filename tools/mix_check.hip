// v_fma_mixlo/hi_f16 against the three-instruction route for the low fp16 slice: bit-for-bit over random fp32 pairs and scales.
// build: hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/mix_check.hip -o tools/_mix_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned lo_pair(float x0, float x1, float s, unsigned hpair) {
    unsigned l = 0;
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x0), "v"(s), "v"(hpair));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(s), "v"(hpair));
    return l;
}
__global__ void k(const float* x, float s, unsigned* out_ref, unsigned* out_new, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x0 = x[2 * i], x1 = x[2 * i + 1];
    f16x2 h; h[0] = (_Float16)(x0 * s); h[1] = (_Float16)(x1 * s);
    f16x2 l; l[0] = (_Float16)(x0 * s - (float)h[0]); l[1] = (_Float16)(x1 * s - (float)h[1]);
    out_ref[i] = __builtin_bit_cast(unsigned, l);
    out_new[i] = lo_pair(x0, x1, s, __builtin_bit_cast(unsigned, h));
}
int main() {
    const int n = 1 << 22;
    float* hx = (float*)malloc(2 * n * 4);
    srand(1);
    for (int i = 0; i < 2 * n; ++i) { float u = (rand() / (float)RAND_MAX - 0.5f); int e = rand() % 40 - 30; hx[i] = ldexpf(u, e); }
    float *dx; unsigned *a, *b; hipMalloc(&dx, 2 * n * 4); hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
    hipMemcpy(dx, hx, 2 * n * 4, hipMemcpyHostToDevice);
    for (float s : {1.0f, 1024.0f, 16384.0f, 0.125f}) {
        hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, s, a, b, n);
        unsigned *ha = (unsigned*)malloc(n * 4), *hb = (unsigned*)malloc(n * 4);
        hipMemcpy(ha, a, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, b, n * 4, hipMemcpyDeviceToHost);
        int bad = 0; for (int i = 0; i < n; ++i) bad += ha[i] != hb[i];
        printf("scale %g: %d of %d pairs differ\n", s, bad, n);
    }
    return 0;
}
