// Does the ACCESS PATTERN of the split-product NT kernel (not its arithmetic) cost HBM efficiency?
// Traffic model of one T x 256 x 256 projection: every workgroup reads a 128 x 256 fp32 tile of A and writes a
// 128 x 256 fp32 tile of C, no MFMA, 256 threads, 64 KiB of LDS requested so that two workgroups share a CU as in
// the real kernel.  Read patterns: K (the kernel's: 16 k-steps, each touching 64 B of all 128 rows, ring depth D)
// and L (linear: each step reads 8 whole rows).  Write patterns: P (the epilogue's 256 B row segments, 4 rows per
// instruction) and L (linear).  Floor: 1.385 GB at the copy rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d line %d\n", (int)e_, __LINE__); return 1; } } while (0)
__device__ __forceinline__ v4f ld(const float* p, bool nt) { return nt ? __builtin_nontemporal_load((const v4f*)p) : *(const v4f*)p; }
__device__ __forceinline__ void st(v4f v, float* p, bool nt) { if (nt) __builtin_nontemporal_store(v, (v4f*)p); else *(v4f*)p = v; }

template <int RK, int WP, int D, int NTL>
__global__ __launch_bounds__(256) void tile(const float* __restrict__ A, float* __restrict__ C, int rows) {
    extern __shared__ float lds[];
    const int t = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 128;
    if (r0 + 128 > rows) return;  // (tail tile skipped: 5282 of 5283 tiles timed)
    const float* a = A + r0 * 256;
    float* c = C + r0 * 256;
    v4f acc = {0, 0, 0, 0};
    // ---- read phase: 16 steps x 2 float4 per thread, D steps in flight
    v4f buf[D][2];
    auto issue = [&](int s, int slot) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float* p = RK ? a + (int64_t)((t >> 2) + 64 * j) * 256 + s * 16 + (t & 3) * 4  // 64 B of every row
                                : a + (int64_t)s * 2048 + j * 1024 + t * 4;                      // 8 whole rows
            buf[slot][j] = ld(p, NTL);
        }
    };
#pragma unroll
    for (int s = 0; s < D - 1; ++s) issue(s, s);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        if (s + D - 1 < 16) issue(s + D - 1, (s + D - 1) % D);
        // consume step s (forces the wait the MFMA loop would have)
        asm volatile("" ::"v"(buf[s % D][0]), "v"(buf[s % D][1]));
        acc += buf[s % D][0] + buf[s % D][1];
        if (D < 16) __builtin_amdgcn_sched_barrier(0);
    }
    lds[t] = acc.x;  // keep the LDS allocation alive
    // ---- write phase: 128 x 256 floats = 32 float4 per thread
    const int w = t >> 6, l = t & 63;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
        float* p;
        if (WP) {  // wave w owns rows 64*(w>>1).., cols 128*(w&1)..; per instruction 4 rows x 256 B
            const int half = i >> 4, ii = i & 15;                 // two 64-column halves of the wave's 128 columns
            const int row = 64 * (w >> 1) + ii * 4 + (l >> 4);
            const int col = 128 * (w & 1) + 64 * half + (l & 15) * 4;
            p = c + (int64_t)row * 256 + col;
        } else {
            p = c + (int64_t)i * 1024 + t * 4;
        }
        st(acc + (float)i, p, NTL);
    }
}
template <typename F>
static float timeit(F f) {
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    for (int i = 0; i < 2; ++i) f();
    (void)hipEventRecord(s); for (int i = 0; i < 10; ++i) f(); (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e); return ms / 10;
}
int main() {
    const int rows = 676200;
    float *A, *C;
    CK(hipMalloc(&A, (size_t)rows * 1024)); CK(hipMalloc(&C, (size_t)rows * 1024));
    CK(hipMemset(A, 0, (size_t)rows * 1024));
    const double gb = 2.0 * rows * 1024 / 1e9;
    const int grid = (rows + 127) / 128;
#define RUN(NAME, RK, WP, D, NTL, LDSB) { auto kf = tile<RK, WP, D, NTL>; \
    CK(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, 65536)); \
    float ms = timeit([&] { hipLaunchKernelGGL(kf, dim3(grid), dim3(256), LDSB, 0, A, C, rows); }); \
    CK(hipGetLastError()); printf("%-58s %7.1f us  %6.0f GB/s\n", NAME, ms * 1e3, gb / (ms * 1e-3)); }
    RUN("read K-step 64B/row (depth 2)  write 256B patches   2 WG/CU", 1, 1, 2, 1, 65536);
    RUN("read K-step 64B/row (depth 3)  write 256B patches   2 WG/CU", 1, 1, 3, 1, 65536);
    RUN("read K-step 64B/row (depth 4)  write 256B patches   2 WG/CU", 1, 1, 4, 1, 65536);
    RUN("read K-step 64B/row (depth 16) write 256B patches   2 WG/CU", 1, 1, 16, 1, 65536);
    RUN("read linear 8 rows/step (d 2)  write 256B patches   2 WG/CU", 0, 1, 2, 1, 65536);
    RUN("read linear 8 rows/step (d 4)  write 256B patches   2 WG/CU", 0, 1, 4, 1, 65536);
    RUN("read linear (depth 16)         write linear         2 WG/CU", 0, 0, 16, 1, 65536);
    RUN("read K-step 64B/row (depth 2)  write linear         2 WG/CU", 1, 0, 2, 1, 65536);
    RUN("read K-step 64B/row (depth 2)  write 256B patches   4 WG/CU", 1, 1, 2, 1, 32768);
    RUN("read K-step 64B/row (depth 4)  write 256B patches   4 WG/CU", 1, 1, 4, 1, 32768);
    RUN("read K-step 64B/row (depth 2)  write patches, cached ld/st ", 1, 1, 2, 0, 65536);
    RUN("read linear (depth 16)         write linear, 8 WG/CU       ", 0, 0, 16, 1, 16384);
    return 0;
}
