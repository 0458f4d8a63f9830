// Which SIMD does wave w of a 512-thread workgroup land on?  (HW_REG_HW_ID: wave_id[3:0] simd_id[5:4] cu_id[11:8] ...)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned* out) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main() {
    unsigned* d; (void)hipMalloc(&d, 4 * 8 * 4);
    hipLaunchKernelGGL(k, dim3(4), dim3(512), 0, 0, d);
    unsigned h[32]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 4; ++b) { printf("block %d:", b); for (int w = 0; w < 8; ++w) printf("  w%d simd=%u wave=%u cu=%u", w, (h[b*8+w] >> 4) & 3, h[b*8+w] & 15, (h[b*8+w] >> 8) & 15); printf("\n"); }
    return 0;
}
