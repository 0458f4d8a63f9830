// Second stage of the packed-fp32 reproducer (tools/pk_f32_repro.hip reproduced the fault stand-alone: REGISTER-ONLY packed
// arithmetic returns wrong values in lanes 48-63 while the library's one-tile f16x3 projection runs on the other stream).
// This tool narrows it down on both sides:
//   victims   (all register-only, inline asm, every result computed twice and compared with unpacked instructions computed twice):
//       P1 v_pk_add_f32      P2 v_pk_mul_f32      P3 v_pk_fma_f32      P4 v_pk_fma_f32 op_sel:[0,1,0] neg_lo/neg_hi
//       S  the same arithmetic in v_add_f32 / v_mul_f32 / v_fma_f32 only (control)
//       K  canary: 24 registers hold a pattern through a long loop of v_mov-free waiting (s_sleep), then are compared
//   neighbours on the second stream:
//       0 none   1 library projection with addend (T rows)   2 library projection without addend   3 library projection, 8192 rows
//       4 DMA only: global_load_lds_dwordx4 in a loop, 512 threads, 64 KiB LDS     5 DMA dword (x1) only
//       6 MFMA + ds_read_b128 + 256 registers, no DMA      7 = 4 + 6 in one kernel
//       8 library gather projection? (not here)            9 library fused dgrad+wgrad (gemm_dw) if exported
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pk_f32_repro2.hip -o tools/_pk_repro2 -ldl
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                      \
        }                                                                                 \
    } while (0)

struct Report {
    unsigned long long wrong;        // results that differ from the unpacked reference
    unsigned long long packed_flip;  // packed computed twice: the two differ (transient)
    unsigned long long scalar_flip;  // unpacked computed twice: the two differ (transient in the control)
    unsigned quarter[4];             // per wave quarter
    unsigned comp[2];                // .x / .y of the packed result
    unsigned n_ex;
    float ex[8][12];                 // examples: x.x x.y y.x y.y a.x a.y b.x b.y got.x got.y want.x want.y
    unsigned ex_lane[8];
};

template <int KIND>
__device__ __forceinline__ v2f packed_op(v2f x, v2f y, v2f a, v2f b) {
    v2f r;
    if (KIND == 1) asm volatile("v_pk_add_f32 %0, %1, %2" : "=&v"(r) : "v"(x), "v"(y));
    if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=&v"(r) : "v"(x), "v"(y));
    if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=&v"(r) : "v"(a), "v"(b), "v"(x));
    if (KIND == 4)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=&v"(r) : "v"(a), "v"(b), "v"(x));
    if (KIND == 6) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=&v"(r) : "v"(a), "v"(b), "v"(x));
    if (KIND == 7) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=&v"(r) : "v"(a), "v"(b), "v"(x));
    if (KIND == 8) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(r) : "v"(x), "v"(y));
    if (KIND == 9) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(r) : "v"(x), "v"(y));
    if (KIND == 10) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=&v"(r) : "v"(a), "v"(b), "v"(x));
    if (KIND == 11) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=&v"(r) : "v"(a), "v"(b), "v"(x));
    if (KIND == 12) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=&v"(r) : "v"(a), "v"(b), "v"(x));
    if (KIND == 13) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=&v"(r) : "v"(a), "v"(b), "v"(x));
    if (KIND == 14 || KIND == 15 || KIND == 16) {  // fp16 payloads: y.x's bits hold two halves made from (y.x, y.y)
        unsigned hp, out;
        asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(hp) : "v"(y.x), "v"(y.y));
        if (KIND == 14)  // the low-slice instruction of split8s: f32 x - f16 high half of a packed register
            asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %0, %3, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]"
                         : "=&v"(out) : "v"(x.x), "v"(hp), "v"(x.y));
        if (KIND == 15) asm volatile("v_pk_mul_f16 %0, %1, %2 op_sel:[0,1]" : "=&v"(out) : "v"(hp), "v"(hp));
        if (KIND == 16) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=&v"(out) : "v"(x.x), "v"(hp), "v"(x.y));
        r.x = __uint_as_float(out), r.y = 0.0f;
    }
    if (KIND == 5) {  // control: unpacked instructions standing in for the "packed" side too
        float r0, r1;
        asm volatile("v_fma_f32 %0, %2, %4, %6\n\tv_fma_f32 %1, %3, %5, %7" : "=&v"(r0), "=&v"(r1) : "v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y), "v"(x.x), "v"(x.y));
        r.x = r0, r.y = r1;
    }
    return r;
}
template <int KIND>
__device__ __forceinline__ v2f scalar_op(v2f x, v2f y, v2f a, v2f b) {
    float r0, r1;
    if (KIND == 1) asm volatile("v_add_f32 %0, %2, %4\n\tv_add_f32 %1, %3, %5" : "=&v"(r0), "=&v"(r1) : "v"(x.x), "v"(x.y), "v"(y.x), "v"(y.y));
    if (KIND == 2) asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %5" : "=&v"(r0), "=&v"(r1) : "v"(x.x), "v"(x.y), "v"(y.x), "v"(y.y));
    if (KIND == 3 || KIND == 5)
        asm volatile("v_fma_f32 %0, %2, %4, %6\n\tv_fma_f32 %1, %3, %5, %7" : "=&v"(r0), "=&v"(r1) : "v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y), "v"(x.x), "v"(x.y));
    if (KIND == 4)
        asm volatile("v_fma_f32 %0, -%2, %4, %5\n\tv_fma_f32 %1, -%3, %4, %6" : "=&v"(r0), "=&v"(r1) : "v"(a.x), "v"(a.y), "v"(b.y), "v"(x.x), "v"(x.y));
    auto fma2 = [&](float a0, float b0, float c0, float a1, float b1, float c1) {
        asm volatile("v_fma_f32 %0, %2, %4, %6\n\tv_fma_f32 %1, %3, %5, %7" : "=&v"(r0), "=&v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1));
    };
    if (KIND == 6) fma2(a.x, b.y, x.x, a.y, b.y, x.y);
    if (KIND == 7) fma2(a.x, b.x, x.x, a.y, b.x, x.y);
    if (KIND == 8) asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %4" : "=&v"(r0), "=&v"(r1) : "v"(x.x), "v"(x.y), "v"(y.y));
    if (KIND == 9) asm volatile("v_add_f32 %0, %2, %4\n\tv_add_f32 %1, %3, %4" : "=&v"(r0), "=&v"(r1) : "v"(x.x), "v"(x.y), "v"(y.y));
    if (KIND == 10) fma2(-a.x, b.x, x.x, -a.y, b.y, x.y);
    if (KIND == 11) fma2(a.y, b.x, x.x, a.y, b.y, x.y);
    if (KIND == 12) fma2(a.x, b.x, x.y, a.y, b.y, x.y);
    if (KIND == 13) fma2(a.x, b.y, x.x, a.y, b.x, x.y);
    if (KIND == 14 || KIND == 15 || KIND == 16) {  // the same values through unpacked / un-swizzled instructions
        unsigned hp, out;
        asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(hp) : "v"(y.x), "v"(y.y));
        unsigned hi_only = hp >> 16, lo_only = hp & 0xffffu;
        asm volatile("" : "+v"(hi_only), "+v"(lo_only));
        if (KIND == 14) {
            unsigned o0, o1;
            asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=&v"(o0) : "v"(x.x), "v"(hi_only), "0"(0u));
            asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=&v"(o1) : "v"(x.y), "v"(lo_only), "0"(0u));
            out = (o0 & 0xffffu) | (o1 << 16);
        }
        if (KIND == 15) {
            const unsigned swz = hi_only | (hi_only << 16);  // (src1: high half for BOTH results? no: op_sel:[0,1] = low result from src1.hi)
            unsigned b = hi_only | (hp & 0xffff0000u);       // low lane takes src1.hi, high lane takes src1.hi (op_sel_hi default 1)
            (void)swz;
            asm volatile("v_pk_mul_f16 %0, %1, %2" : "=&v"(out) : "v"(hp), "v"(b));
        }
        if (KIND == 16) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=&v"(out) : "v"(x.x), "v"(hi_only), "v"(x.y));
        r0 = __uint_as_float(out), r1 = 0.0f;
    }
    v2f r = {r0, r1};
    return r;
}

template <int KIND>
__global__ void victim_kernel(const float* __restrict__ in, int n, int rounds, Report* __restrict__ rep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    for (int rd = 0; rd < rounds; ++rd) {
        const int j = (i + rd * 977) % n;
        const v2f x = {in[8 * j], in[8 * j + 1]}, y = {in[8 * j + 2], in[8 * j + 3]}, a = {in[8 * j + 4], in[8 * j + 5]},
                  b = {in[8 * j + 6], in[8 * j + 7]};
        const v2f r = packed_op<KIND>(x, y, a, b), r2 = packed_op<KIND>(x, y, a, b);
        const v2f e = scalar_op<KIND>(x, y, a, b), e2 = scalar_op<KIND>(x, y, a, b);
        const bool w0 = __float_as_uint(r.x) != __float_as_uint(e.x), w1 = __float_as_uint(r.y) != __float_as_uint(e.y);
        if (__float_as_uint(r.x) != __float_as_uint(r2.x) || __float_as_uint(r.y) != __float_as_uint(r2.y)) atomicAdd(&rep->packed_flip, 1ull);
        if (__float_as_uint(e.x) != __float_as_uint(e2.x) || __float_as_uint(e.y) != __float_as_uint(e2.y)) atomicAdd(&rep->scalar_flip, 1ull);
        if (w0 || w1) {
            atomicAdd(&rep->wrong, 1ull);
            atomicAdd(&rep->quarter[lane >> 4], 1u);
            if (w0) atomicAdd(&rep->comp[0], 1u);
            if (w1) atomicAdd(&rep->comp[1], 1u);
            const unsigned k = atomicAdd(&rep->n_ex, 1u);
            if (k < 8) {
                float* o = rep->ex[k];
                o[0] = x.x, o[1] = x.y, o[2] = y.x, o[3] = y.y, o[4] = a.x, o[5] = a.y, o[6] = b.x, o[7] = b.y;
                o[8] = r.x, o[9] = r.y, o[10] = e.x, o[11] = e.y;
                rep->ex_lane[k] = lane;
            }
        }
    }
}

// K: registers that nothing touches
__global__ void canary_kernel(int rounds, Report* __restrict__ rep) {
    const int lane = threadIdx.x & 63;
    unsigned v[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) {
        v[k] = 0x9e3779b9u * (unsigned)(threadIdx.x + 1) + 0x01000193u * (unsigned)k;
        asm volatile("" : "+v"(v[k]));
    }
    for (int rd = 0; rd < rounds; ++rd) {
        asm volatile("s_sleep 8" ::: "memory");
#pragma unroll
        for (int k = 0; k < 24; ++k) asm volatile("" : "+v"(v[k]));
    }
    unsigned bad = 0;
#pragma unroll
    for (int k = 0; k < 24; ++k) bad += v[k] != 0x9e3779b9u * (unsigned)(threadIdx.x + 1) + 0x01000193u * (unsigned)k;
    if (bad) {
        atomicAdd(&rep->wrong, (unsigned long long)bad);
        atomicAdd(&rep->quarter[lane >> 4], bad);
    }
}

// ---- own neighbours
template <int BYTES>
__global__ __launch_bounds__(512) void dma_neighbour_kernel(const unsigned char* __restrict__ src, size_t bytes, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned lds0 = (unsigned)(size_t)lds + wave * 8192;
    size_t off = ((size_t)blockIdx.x * 8 + wave) * 8192;
    for (int it = 0; it < iters; ++it) {
        const unsigned long long so_ = off % (bytes - 8192);
        const unsigned char* sb = src + (((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(so_ >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)so_));
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const unsigned dst = lds0 + p * 1024;
            const unsigned lo = (unsigned)(p * 1024 + lane * BYTES);
            if (BYTES == 16)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(lo), "s"(sb) : "memory");
            else
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(dst), "v"(lo), "s"(sb) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        off += (size_t)gridDim.x * 8 * 8192;
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = reinterpret_cast<float*>(lds)[threadIdx.x];
}

// MFMA + LDS reads in a 512-thread workgroup at 2 waves per SIMD (the projection's shape, without its DMA); DMA != 0 adds the DMA
template <bool DMA, bool MFMA = true, bool LDSR = true>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void mfma_neighbour_kernel(const unsigned char* __restrict__ src, size_t bytes,
                                                                                                     float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<float*>(lds)[i] = (float)(i & 127) * 0.01f;
    __syncthreads();
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const unsigned lds0 = (unsigned)(size_t)lds + wave * 8192;
    size_t off = ((size_t)blockIdx.x * 8 + wave) * 8192;
    for (int it = 0; it < iters; ++it) {
        if (DMA) {
            const unsigned long long so_ = off % (bytes - 8192);
            const unsigned char* sb = src + (((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(so_ >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)so_));
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const unsigned dst = lds0 + ((it & 3) * 2 + p) * 1024;
                const unsigned lo = (unsigned)(p * 1024 + lane * 16);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(lo), "s"(sb) : "memory");
            }
            off += (size_t)gridDim.x * 8 * 2048;
        }
        f16x8 a[2], b[4];
        if (LDSR || it == 0) {
#pragma unroll
            for (int k = 0; k < 2; ++k) a[k] = *reinterpret_cast<const f16x8*>(lds + ((it * 2 + k) & 31) * 1024 + lane * 16);
#pragma unroll
            for (int k = 0; k < 4; ++k) b[k] = *reinterpret_cast<const f16x8*>(lds + 32768 + ((it * 4 + k) & 31) * 1024 + lane * 16);
        }
        if (MFMA) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 1], b[i >> 1], acc[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i][0] += (float)a[i & 1][0] + (float)b[i >> 1][1];
        }
        if (DMA) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.0f;
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void small_mfma_kernel(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(float)(threadIdx.x + i), b[i] = (_Float16)(float)(i);
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    float s = 0.0f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    const char* libpath = argc > 2 ? argv[2] : "alignn_amd/libalignn_hip.so";
    hipStream_t sa, sb;
    CK(hipStreamCreate(&sa));
    CK(hipStreamCreate(&sb));
    float* spin_out;
    CK(hipMalloc(&spin_out, 4096 * 512 * sizeof(float)));
    unsigned char* spin_src;
    const size_t spin_bytes = (size_t)1 << 30;
    CK(hipMalloc(&spin_src, spin_bytes));
    CK(hipMemset(spin_src, 0, spin_bytes));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(dma_neighbour_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(dma_neighbour_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_neighbour_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_neighbour_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_neighbour_kernel<false, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_neighbour_kernel<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));

    typedef int (*gemm_fn)(const float*, int64_t, const float*, const void*, const float*, const float*, const float*, int64_t, float*, int64_t,
                           int64_t, int, int, void*);
    typedef int (*split_fn)(const float*, int64_t, int, int, int, const float*, void*, void*);
    typedef size_t (*bytes_fn)(int, int);
    typedef int (*absmax_fn)(const float*, int64_t, int64_t, int, float*, void*);
    gemm_fn lib_gemm = nullptr;
    const int64_t TR = 1012986;
    float *pj_a = nullptr, *pj_w = nullptr, *pj_add = nullptr, *pj_c = nullptr, *pj_amax = nullptr;
    void* pj_img = nullptr;
    if (void* h = dlopen(libpath, RTLD_NOW | RTLD_LOCAL)) {
        lib_gemm = (gemm_fn)dlsym(h, "alignn_gemm_nt_f16x3");
        split_fn lib_split = (split_fn)dlsym(h, "alignn_split_f16x2");
        bytes_fn lib_bytes = (bytes_fn)dlsym(h, "alignn_split_f16x2_bytes");
        absmax_fn lib_absmax = (absmax_fn)dlsym(h, "alignn_absmax");
        if (lib_gemm && lib_split && lib_bytes && lib_absmax) {
            CK(hipMalloc(&pj_a, TR * 256 * 4));
            CK(hipMalloc(&pj_add, TR * 256 * 4));
            CK(hipMalloc(&pj_c, TR * 256 * 4));
            CK(hipMalloc(&pj_w, 256 * 256 * 4));
            CK(hipMalloc(&pj_amax, 256));
            CK(hipMalloc(&pj_img, lib_bytes(256, 256)));
            std::vector<float> hw(256 * 256), ha(1 << 20);
            for (size_t i = 0; i < hw.size(); ++i) hw[i] = ((int)(i * 2654435761u >> 12) % 2001 - 1000) * 1.0e-4f;
            for (size_t i = 0; i < ha.size(); ++i) ha[i] = ((int)(i * 40503u >> 7) % 2001 - 1000) * 1.0e-3f;
            CK(hipMemcpy(pj_w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
            for (int64_t o = 0; o < TR * 256; o += (int64_t)ha.size()) {
                const size_t nb = (size_t)((TR * 256 - o) < (int64_t)ha.size() ? (TR * 256 - o) : (int64_t)ha.size());
                CK(hipMemcpy(pj_a + o, ha.data(), nb * 4, hipMemcpyHostToDevice));
                CK(hipMemcpy(pj_add + o, ha.data(), nb * 4, hipMemcpyHostToDevice));
            }
            lib_absmax(pj_a, 256, TR, 256, pj_amax, nullptr);
            lib_absmax(pj_w, 256, 256, 256, pj_amax + 1, nullptr);
            lib_split(pj_w, 256, 256, 256, 1, pj_amax + 1, pj_img, nullptr);
            CK(hipDeviceSynchronize());
        } else
            lib_gemm = nullptr;
    }
    if (!lib_gemm) printf("# (%s not found: no library neighbours)\n", libpath);

    const char* nb_name[] = {"alone",
                             "library projection + addend, T rows",
                             "library projection, no addend, T rows",
                             "library projection + addend, 8192 rows x 100",
                             "DMA only (global_load_lds_dwordx4)",
                             "DMA only (global_load_lds_dword)",
                             "MFMA + ds_read_b128, 512 threads, 2 waves/SIMD",
                             "MFMA + ds_read_b128 + DMA x4, one kernel",
                             "MFMA only, 512 threads, 2 waves/SIMD",
                             "ds_read_b128 only, 512 threads",
                             "MFMA only, 256 threads, 4 accumulators"};
    auto spin = [&](int nb) {
        if (nb == 1 && lib_gemm)
            for (int k = 0; k < 3; ++k) lib_gemm(pj_a, 256, pj_amax, pj_img, pj_amax + 1, nullptr, pj_add, 256, pj_c, 256, TR, 256, 256, (void*)sb);
        if (nb == 2 && lib_gemm)
            for (int k = 0; k < 3; ++k) lib_gemm(pj_a, 256, pj_amax, pj_img, pj_amax + 1, nullptr, nullptr, 0, pj_c, 256, TR, 256, 256, (void*)sb);
        if (nb == 3 && lib_gemm)
            for (int k = 0; k < 100; ++k) lib_gemm(pj_a, 256, pj_amax, pj_img, pj_amax + 1, nullptr, pj_add, 256, pj_c, 256, 8192, 256, 256, (void*)sb);
        if (nb == 4) hipLaunchKernelGGL(dma_neighbour_kernel<16>, dim3(512), dim3(512), 65536, sb, spin_src, spin_bytes, spin_out, 600);
        if (nb == 5) hipLaunchKernelGGL(dma_neighbour_kernel<4>, dim3(512), dim3(512), 65536, sb, spin_src, spin_bytes, spin_out, 1200);
        if (nb == 6) hipLaunchKernelGGL(mfma_neighbour_kernel<false>, dim3(512), dim3(512), 65536, sb, spin_src, spin_bytes, spin_out, 3000);
        if (nb == 8) hipLaunchKernelGGL((mfma_neighbour_kernel<false, true, false>), dim3(512), dim3(512), 65536, sb, spin_src, spin_bytes, spin_out, 3000);
        if (nb == 9) hipLaunchKernelGGL((mfma_neighbour_kernel<false, false, true>), dim3(512), dim3(512), 65536, sb, spin_src, spin_bytes, spin_out, 3000);
        if (nb == 10) hipLaunchKernelGGL(small_mfma_kernel, dim3(1024), dim3(256), 0, sb, spin_out, 4000);
        if (nb == 7) hipLaunchKernelGGL(mfma_neighbour_kernel<true>, dim3(512), dim3(512), 65536, sb, spin_src, spin_bytes, spin_out, 3000);
    };

    const int n = 1 << 16;
    std::vector<float> h(8 * n);
    unsigned s = 12345u;
    for (auto& v : h) {
        s = s * 1664525u + 1013904223u;
        v = ((int)(s >> 8) % 20001 - 10000) * 1.0e-3f;
    }
    float* d_in;
    CK(hipMalloc(&d_in, h.size() * 4));
    CK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    Report* d_rep;
    CK(hipMalloc(&d_rep, sizeof(Report)));
    hipEvent_t e0, e1, n0, n1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventCreate(&n0));
    CK(hipEventCreate(&n1));

    auto run = [&](const char* name, int nb, auto launch) {
        CK(hipMemset(d_rep, 0, sizeof(Report)));
        CK(hipDeviceSynchronize());
        float v_ms = 0, n_ms = 0;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(n0, sb));
            spin(nb);
            CK(hipEventRecord(n1, sb));
            CK(hipEventRecord(e0, sa));
            launch();
            CK(hipEventRecord(e1, sa));
            CK(hipDeviceSynchronize());
            float a, b;
            CK(hipEventElapsedTime(&a, e0, e1));
            CK(hipEventElapsedTime(&b, n0, n1));
            v_ms += a, n_ms += b;
        }
        Report rp;
        CK(hipMemcpy(&rp, d_rep, sizeof(rp), hipMemcpyDeviceToHost));
        printf("%-44s | %-46s | wrong %8llu  packed twice differ %8llu  unpacked twice differ %6llu | quarters %u %u %u %u | .x %u .y %u | victim %.2f ms, neighbour %.2f ms per round\n",
               name, nb_name[nb], rp.wrong, rp.packed_flip, rp.scalar_flip, rp.quarter[0], rp.quarter[1], rp.quarter[2], rp.quarter[3], rp.comp[0],
               rp.comp[1], v_ms / reps, n_ms / reps);
        for (unsigned k = 0; k < (rp.n_ex < 3 ? rp.n_ex : 3); ++k) {
            const float* o = rp.ex[k];
            printf("      lane %2u: x (%.9g, %.9g) y (%.9g, %.9g) a (%.9g, %.9g) b (%.9g, %.9g): packed (%.9g, %.9g) unpacked (%.9g, %.9g)\n", rp.ex_lane[k],
                   o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8], o[9], o[10], o[11]);
        }
        fflush(stdout);
    };
    printf("# tools/pk_f32_repro2.hip: %d rounds per line; victim = 2048 x 256 threads x 400 results per round\n", reps);
#define VICTIM(K_) [&] { hipLaunchKernelGGL(victim_kernel<K_>, dim3(2048), dim3(256), 0, sa, (const float*)d_in, n, 400, d_rep); }
    const bool brief = argc > 3;
    for (int nb = 0; nb < 11; ++nb) {
        if (!lib_gemm && nb >= 1 && nb <= 3) continue;
        if (nb == 0 || nb == 1 || nb == 6 || nb >= 8) {
            run("P5  v_pk_fma_f32 op_sel:[0,1,0]", nb, VICTIM(6));
            run("P6  v_pk_fma_f32 op_sel_hi:[1,0,1]", nb, VICTIM(7));
            run("P7  v_pk_mul_f32 op_sel:[0,1]", nb, VICTIM(8));
            run("P8  v_pk_add_f32 op_sel:[0,1]", nb, VICTIM(9));
            run("P9  v_pk_fma_f32 neg_lo:[1,0,0] neg_hi:[1,0,0]", nb, VICTIM(10));
            run("P10 v_pk_fma_f32 op_sel:[1,0,0]", nb, VICTIM(11));
            run("P11 v_pk_fma_f32 op_sel:[0,0,1]", nb, VICTIM(12));
            run("P12 v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1]", nb, VICTIM(13));
            run("M1  v_fma_mixlo/hi_f16 op_sel:[0,0,1] (split8s)", nb, VICTIM(14));
            run("M2  v_pk_mul_f16 op_sel:[0,1]", nb, VICTIM(15));
            run("M3  v_fma_mix_f32 op_sel:[0,1,0] op_sel_hi:[0,1,0]", nb, VICTIM(16));
        }
        if (brief && nb < 8) continue;
        run("P1 v_pk_add_f32", nb, [&] { hipLaunchKernelGGL(victim_kernel<1>, dim3(2048), dim3(256), 0, sa, (const float*)d_in, n, 400, d_rep); });
        run("P2 v_pk_mul_f32", nb, [&] { hipLaunchKernelGGL(victim_kernel<2>, dim3(2048), dim3(256), 0, sa, (const float*)d_in, n, 400, d_rep); });
        run("P3 v_pk_fma_f32", nb, [&] { hipLaunchKernelGGL(victim_kernel<3>, dim3(2048), dim3(256), 0, sa, (const float*)d_in, n, 400, d_rep); });
        run("P4 v_pk_fma_f32 op_sel:[0,1,0] neg:[1,0,0]", nb, [&] { hipLaunchKernelGGL(victim_kernel<4>, dim3(2048), dim3(256), 0, sa, (const float*)d_in, n, 400, d_rep); });
        run("S  v_fma_f32 only (control)", nb, [&] { hipLaunchKernelGGL(victim_kernel<5>, dim3(2048), dim3(256), 0, sa, (const float*)d_in, n, 400, d_rep); });
        run("K  canary registers", nb, [&] { hipLaunchKernelGGL(canary_kernel, dim3(2048), dim3(256), 0, sa, 2000, d_rep); });
    }
    return 0;
}
