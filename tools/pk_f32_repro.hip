// Stand-alone reproducer for the packed-fp32 fault of DESIGN.md section 4.6 (round 5: ln_silu_bwd_kernel<1> returned one float4
// component of lanes 48-63 wrong when an MFMA kernel of another stream shared the compute unit; worked around by building
// norm.hip / dual.hip / convln.hip with -fno-slp-vectorize).  Three experiments, each run ALONE and BESIDE an MFMA spinner on a
// second stream (the stand-in for a projection of another lane, or an RCCL kernel):
//   A. the suspected instruction pair in isolation: v_pk_add_f32 -> [s_nop N] -> dependent v_pk_fma_f32 ... op_sel:[0,1,0]
//      (hand-written, checked per lane against the same arithmetic in unpacked instructions);
//   B. ... the same pair fed by transcendentals (v_exp_f32 / v_rcp_f32 one instruction ahead: lanes 48-63 are the LAST quarter
//      pass of a quarter-rate instruction);
//   C. the REAL kernel: csrc/norm.hip is compiled into this tool (whatever flags the tool is built with), alignn_ln_silu_bwd run
//      on fixed inputs, every output compared bit for bit with the first run.
// Build both ways and run:   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pk_f32_repro.hip -o tools/_pk_repro_slp
//                            hipcc ... -fno-slp-vectorize tools/pk_f32_repro.hip -o tools/_pk_repro_noslp
#include "../alignn_amd/csrc/norm.hip"

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                \
        }                                                                           \
    } while (0)

// ---- the MFMA spinner: `waves` waves per workgroup, dependent-free MFMAs for `iters` rounds
__global__ void spinner_kernel(float* out, int iters) {
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

// ---- other neighbours: LDS traffic (ds_read / ds_write in a loop, 32 KiB per workgroup) and streaming global loads
__global__ void lds_spinner_kernel(float* out, int iters) {
    __shared__ float4 buf[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) buf[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    float4 a = make_float4(0, 0, 0, 0);
    int j = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        const float4 v = buf[j & 2047];
        a.x += v.x, a.y += v.y, a.z += v.z, a.w += v.w;
        buf[(j + 777) & 2047] = a;
        j = j * 5 + 1;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a.x + a.y + a.z + a.w;
}
__global__ void mem_spinner_kernel(const float4* __restrict__ src, size_t n, float* out, int iters) {
    float4 a = make_float4(0, 0, 0, 0);
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        const float4 v = src[i % n];
        a.x += v.x, a.y += v.y, a.z += v.z, a.w += v.w;
        i += (size_t)gridDim.x * blockDim.x;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a.x + a.y + a.z + a.w;
}

// quarter-rate instructions in a loop (what lane T's LayerNorm / SiLU kernels are made of)
__global__ void trans_spinner_kernel(float* out, int iters) {
    float a = 0.001f * (float)(threadIdx.x + 1), b = 1.0f + a, c = 2.0f + a, d = 3.0f + a;
    for (int it = 0; it < iters; ++it) {
        a = __builtin_amdgcn_rcpf(1.0f + __expf(-a));
        b = __builtin_amdgcn_rcpf(1.0f + __expf(-b));
        c = __builtin_amdgcn_rcpf(1.0f + __expf(-c));
        d = __builtin_amdgcn_rcpf(1.0f + __expf(-d));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
__global__ void copy_spinner_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n, int passes) {
    for (int p = 0; p < passes; ++p)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// ---- D: a packed-fp32 result read as the DATA operand of an LDS instruction by the very next instruction - the butterfly of
//      wave_sum() as the SLP vectoriser compiles it:  v_pk_add_f32 v[a:a+1], ... ; ds_bpermute_b32 x, addr, v[a] ; ... v[a+1]
template <int NOP>
__global__ void pk_lds_kernel(const float* __restrict__ in, int n, int rounds, unsigned long long* __restrict__ bad,
                              unsigned* __restrict__ bad_lane) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const unsigned addr = (unsigned)((lane ^ 32) << 2);
    unsigned long long wrong = 0;
    for (int rd = 0; rd < rounds; ++rd) {
        const int j = (i + rd * 977) % n;
        v2f x = {in[8 * j], in[8 * j + 1]}, y = {in[8 * j + 2], in[8 * j + 3]};
        float p0, p1;  // (the packed sum lives in v[200:201]: the LDS instructions name its halves)
        if (NOP == 0)
            asm volatile("v_pk_add_f32 v[200:201], %2, %3\n\tds_bpermute_b32 %0, %4, v200\n\tds_bpermute_b32 %1, %4, v201\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(p0), "=&v"(p1)
                         : "v"(x), "v"(y), "v"(addr)
                         : "v200", "v201");
        else
            asm volatile("v_pk_add_f32 v[200:201], %2, %3\n\ts_nop %5\n\tds_bpermute_b32 %0, %4, v200\n\tds_bpermute_b32 %1, %4, v201\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(p0), "=&v"(p1)
                         : "v"(x), "v"(y), "v"(addr), "n"(NOP - 1)
                         : "v200", "v201");
        // what the partner lane (lane ^ 32) must have sent: the halves of ITS packed sum
        const float expect = __shfl_xor(x.x + y.x, 32, 64), expect1 = __shfl_xor(x.y + y.y, 32, 64);
        if (__float_as_uint(p0) != __float_as_uint(expect) || __float_as_uint(p1) != __float_as_uint(expect1)) {
            ++wrong;
            atomicOr(bad_lane + (lane >> 4), 1u);
        }
    }
    if (wrong) atomicAdd(bad, wrong);
}

// ---- F: the data registers of a 128-bit store overwritten by packed arithmetic a few instructions later - what the SLP build of
//      ln_silu_bwd_kernel<1, true> does (GS1 store v[42:45]; s_waitcnt vmcnt(1); s_nop 0; v_pk_mul_f32 v[42:43]; v_pk_mul_f32 v[44:45];
//      GS0 store).  Every wave stores known values, overwrites the registers, stores the products elsewhere, ... with all waves of
//      the chip storing at once (the memory pipeline's queue is full, as beside lane T's kernels).  Checked afterwards: does the
//      FIRST buffer hold the values the registers had when its store was issued?
template <int NOP, bool PACKED>
__global__ void store_war_kernel(float* __restrict__ out1, float* __restrict__ out2, int rounds) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (int rd = 0; rd < rounds; ++rd) {
        const size_t i = t + (size_t)rd * stride;
        float4 v = make_float4((float)(i & 0xffff) + 1.0f, (float)(i & 0xffff) + 2.0f, (float)(i & 0xffff) + 3.0f, (float)(i & 0xffff) + 4.0f);
        float4* p1 = reinterpret_cast<float4*>(out1) + i;
        float4* p2 = reinterpret_cast<float4*>(out2) + i;
        const float m = -0.5f;
        // (v[200:203] = the value; store; [s_nop]; two packed (or four plain) multiplies in place; second store)
        if (PACKED)
            asm volatile(
                "v_mov_b32 v200, %2\n\tv_mov_b32 v201, %3\n\tv_mov_b32 v202, %4\n\tv_mov_b32 v203, %5\n\tv_mov_b32 v204, %6\n\tv_mov_b32 v205, %6\n\t"
                "s_nop 4\n\t"
                "global_store_dwordx4 %0, v[200:203], off\n\t"
                "s_nop %7\n\t"
                "v_pk_mul_f32 v[200:201], v[204:205], v[200:201]\n\t"
                "v_pk_mul_f32 v[202:203], v[204:205], v[202:203]\n\t"
                "s_nop 4\n\t"
                "global_store_dwordx4 %1, v[200:203], off\n\t"
                "s_waitcnt vmcnt(0)"
                :
                : "v"(p1), "v"(p2), "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "v"(m), "n"(NOP)
                : "memory", "v200", "v201", "v202", "v203", "v204", "v205");
        else
            asm volatile(
                "v_mov_b32 v200, %2\n\tv_mov_b32 v201, %3\n\tv_mov_b32 v202, %4\n\tv_mov_b32 v203, %5\n\tv_mov_b32 v204, %6\n\t"
                "s_nop 4\n\t"
                "global_store_dwordx4 %0, v[200:203], off\n\t"
                "s_nop %7\n\t"
                "v_mul_f32 v200, v204, v200\n\tv_mul_f32 v201, v204, v201\n\tv_mul_f32 v202, v204, v202\n\tv_mul_f32 v203, v204, v203\n\t"
                "s_nop 4\n\t"
                "global_store_dwordx4 %1, v[200:203], off\n\t"
                "s_waitcnt vmcnt(0)"
                :
                : "v"(p1), "v"(p2), "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "v"(m), "n"(NOP)
                : "memory", "v200", "v201", "v202", "v203", "v204");
    }
}
__global__ void store_war_check(const float* __restrict__ out1, size_t n4, unsigned long long* __restrict__ bad, unsigned* __restrict__ lane_q,
                                unsigned* __restrict__ comp) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(out1)[i];
        const float b = (float)(i & 0xffff);
        const float e[4] = {b + 1.0f, b + 2.0f, b + 3.0f, b + 4.0f}, g[4] = {v.x, v.y, v.z, v.w};
        for (int c = 0; c < 4; ++c)
            if (g[c] != e[c]) {
                atomicAdd(bad, 1ull);
                atomicAdd(lane_q + ((i & 63) >> 4), 1u);
                atomicAdd(comp + c, 1u);
            }
    }
}

// ---- G: the WHOLE tail of ln_silu_bwd_kernel<1>'s row loop as hipcc -O3 emits it with the SLP vectoriser on (csrc/norm.hip, the
//      listing of profiles/r05_ln_concurrency.txt section 6), verbatim - registers and all - behind the last butterfly step of
//      wave_sum(); every lane checks its four results against the same data flow in plain float arithmetic
#pragma clang fp contract(off)
__device__ __forceinline__ void tail_reference(const float* in, float b42, float b43, float* o) {
    const float v16 = in[0], v17 = in[1], v30 = in[2], v32 = in[4], v33 = in[5], v34 = in[6], v35 = in[7], v37 = in[9], v38 = in[10],
                v39 = in[11], v40 = in[12], v41 = in[13], v50 = in[14], v51 = in[15];
    const float t40 = v40 + b42, t41 = v41 + b43;
    const float m42 = v16 * t40, m43 = v17 * t41;
    const float f40 = __builtin_fmaf(-v16, t40, v50);
    const float a33 = v33 * m43, a41 = v37 - m42, a32 = v32 * m43;
    const float p36 = v38 - m42, p37 = v39 - m42;
    const float q32 = f40 - a32, q33 = a41 - a33;
    const float r34 = __builtin_fmaf(-v34, m43, p36), r35 = __builtin_fmaf(-v35, m43, p37);
    o[0] = v30 * q32, o[1] = v30 * q33, o[2] = v30 * r34, o[3] = v30 * r35;
}
__global__ void tail_kernel(const float* __restrict__ in, int n, int rounds, unsigned long long* __restrict__ bad,
                            unsigned* __restrict__ bad_lane) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const unsigned addr = (unsigned)((lane ^ 1) << 2);
    unsigned long long wrong = 0;
    for (int rd = 0; rd < rounds; ++rd) {
        const int j = (i + rd * 977) % n;
        float x[16];
        for (int k = 0; k < 16; ++k) x[k] = in[16 * j + k];
        float r0, r1, r2, r3;
        asm volatile(
            "v_mov_b32 v16, %4\n\tv_mov_b32 v17, %5\n\tv_mov_b32 v30, %6\n\tv_mov_b32 v31, %7\n\tv_mov_b32 v32, %8\n\tv_mov_b32 v33, %9\n\t"
            "v_mov_b32 v34, %10\n\tv_mov_b32 v35, %11\n\tv_mov_b32 v36, %12\n\tv_mov_b32 v37, %13\n\tv_mov_b32 v38, %14\n\tv_mov_b32 v39, %15\n\t"
            "v_mov_b32 v40, %16\n\tv_mov_b32 v41, %17\n\tv_mov_b32 v50, %18\n\tv_mov_b32 v51, %19\n\t"
            "s_nop 4\n\t"
            "ds_bpermute_b32 v42, %20, v40\n\t"
            "ds_bpermute_b32 v43, %20, v41\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_pk_add_f32 v[40:41], v[40:41], v[42:43]\n\t"
            "s_nop 0\n\t"
            "v_pk_mul_f32 v[42:43], v[16:17], v[40:41]\n\t"
            "v_pk_fma_f32 v[40:41], v[16:17], v[40:41], v[50:51] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
            "v_mul_f32_e32 v33, v33, v43\n\t"
            "v_sub_f32_e32 v41, v37, v42\n\t"
            "v_mul_f32_e32 v32, v32, v43\n\t"
            "v_pk_add_f32 v[36:37], v[38:39], v[42:43] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 v[32:33], v[40:41], v[32:33] neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_fma_f32 v[34:35], v[34:35], v[42:43], v[36:37] op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
            "v_pk_mul_f32 v[32:33], v[30:31], v[32:33] op_sel_hi:[0,1]\n\t"
            "v_pk_mul_f32 v[34:35], v[30:31], v[34:35] op_sel_hi:[0,1]\n\t"
            "s_nop 4\n\t"
            "v_mov_b32 %0, v32\n\tv_mov_b32 %1, v33\n\tv_mov_b32 %2, v34\n\tv_mov_b32 %3, v35"
            : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
            : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(x[9]), "v"(x[10]),
              "v"(x[11]), "v"(x[12]), "v"(x[13]), "v"(x[14]), "v"(x[15]), "v"(addr)
            : "v16", "v17", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v50", "v51");
        const float b42 = __shfl_xor(x[12], 1, 64), b43 = __shfl_xor(x[13], 1, 64);
        float e[4];
        tail_reference(x, b42, b43, e);
        const float g[4] = {r0, r1, r2, r3};
        for (int c = 0; c < 4; ++c)
            if (__float_as_uint(g[c]) != __float_as_uint(e[c])) {
                ++wrong;
                atomicOr(bad_lane + (lane >> 4), 1u << c);
            }
    }
    if (wrong) atomicAdd(bad, wrong);
}

// ---- A / B: the instruction pair.  r = (-a) * (b.hi, b.hi) + (x + y) per half, as the kernel's
//      v_pk_add_f32 t, x, y ; v_pk_fma_f32 r, a, b, t op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]
template <int NOP, bool TRANS>
__global__ void pair_kernel(const float* __restrict__ in, int n, int rounds, unsigned long long* __restrict__ bad,
                            unsigned* __restrict__ bad_lane) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long wrong = 0;
    for (int rd = 0; rd < rounds; ++rd) {
        const int j = (i + rd * 977) % n;
        v2f x = {in[8 * j], in[8 * j + 1]}, y = {in[8 * j + 2], in[8 * j + 3]}, a = {in[8 * j + 4], in[8 * j + 5]},
            b = {in[8 * j + 6], in[8 * j + 7]};
        if (TRANS) {  // the operands come out of the sigmoid's instructions (the kernel's dsilu_f): rcp(1 + exp(-x))
            x.x = __builtin_amdgcn_rcpf(1.0f + __expf(-x.x));
            x.y = __builtin_amdgcn_rcpf(1.0f + __expf(-x.y));
            b.y = __builtin_amdgcn_rcpf(1.0f + __expf(-b.y));
        }
        v2f r, t;
        if (NOP == 0)
            asm volatile("v_pk_add_f32 %1, %2, %3\n\tv_pk_fma_f32 %0, %4, %5, %1 op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]"
                         : "=&v"(r), "=&v"(t)
                         : "v"(x), "v"(y), "v"(a), "v"(b));
        else
            asm volatile("v_pk_add_f32 %1, %2, %3\n\ts_nop %6\n\tv_pk_fma_f32 %0, %4, %5, %1 op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]"
                         : "=&v"(r), "=&v"(t)
                         : "v"(x), "v"(y), "v"(a), "v"(b), "n"(NOP - 1));
        // the same arithmetic, one component at a time
        float t0, t1, e0, e1;
        asm volatile("v_add_f32 %0, %4, %6\n\tv_add_f32 %1, %5, %7\n\ts_nop 4\n\tv_fma_f32 %2, -%8, %10, %0\n\tv_fma_f32 %3, -%9, %10, %1"
                     : "=&v"(t0), "=&v"(t1), "=&v"(e0), "=&v"(e1)
                     : "v"(x.x), "v"(x.y), "v"(y.x), "v"(y.y), "v"(a.x), "v"(a.y), "v"(b.y));
        if (__float_as_uint(r.x) != __float_as_uint(e0) || __float_as_uint(r.y) != __float_as_uint(e1)) {
            ++wrong;
            atomicOr(bad_lane + ((threadIdx.x & 63) >> 4), 1u);
        }
    }
    if (wrong) atomicAdd(bad, wrong);
}

__global__ void diff_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n, unsigned long long* __restrict__ cnt,
                            unsigned* __restrict__ lane_q, unsigned* __restrict__ comp) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (__float_as_uint(a[i]) != __float_as_uint(b[i])) {
            atomicAdd(cnt, 1ull);
            atomicAdd(lane_q + (((i & 255) >> 2) >> 4), 1u);  // quarter of the wave (16 lanes each) the element's lane is in
            atomicAdd(comp + (i & 3), 1u);                    // component of the lane's float4
        }
}

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 131072, F = 256, reps = argc > 2 ? atoi(argv[2]) : 20;
    hipStream_t sa, sb, sc, sd;
    CK(hipStreamCreate(&sa));
    CK(hipStreamCreate(&sb));
    CK(hipStreamCreate(&sc));
    CK(hipStreamCreate(&sd));
    float* spin_out;
    CK(hipMalloc(&spin_out, 4096 * 256 * sizeof(float)));
    float4* spin_src;
    const size_t spin_n = (size_t)64 << 20;  // 1 GiB of float4
    CK(hipMalloc(&spin_src, spin_n * sizeof(float4)));
    CK(hipMemset(spin_src, 0, spin_n * sizeof(float4)));
    int neighbour = 1;  // 1 MFMA, 2 LDS traffic, 3 streaming loads, 4 all three, 5 transcendentals, 6 the library's T-row projection
    const char* nb_name[] = {"alone", "beside MFMA kernel", "beside LDS kernel", "beside load kernel", "beside all three", "beside exp/rcp kernel",
                             "beside gemm_nt_x6_kernel<addend> (T rows)"};
    // neighbour 6: the kernel that runs beside the failing launches inside the model (tools/rocpd_overlap.py): the one-tile f16x3
    // projection with a residual addend over T rows, out of the shipped library
    typedef int (*gemm_fn)(const float*, int64_t, const float*, const void*, const float*, const float*, const float*, int64_t, float*, int64_t,
                           int64_t, int, int, void*);
    typedef int (*split_fn)(const float*, int64_t, int, int, int, const float*, void*, void*);
    typedef size_t (*bytes_fn)(int, int);
    typedef int (*absmax_fn)(const float*, int64_t, int64_t, int, float*, void*);
    gemm_fn lib_gemm = nullptr;
    const int64_t TR = 1012986;
    float *pj_a = nullptr, *pj_w = nullptr, *pj_add = nullptr, *pj_c = nullptr, *pj_amax = nullptr;
    void* pj_img = nullptr;
    if (void* h = dlopen(argc > 3 ? argv[3] : "alignn_amd/libalignn_hip.so", RTLD_NOW | RTLD_LOCAL)) {
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
    if (!lib_gemm) printf("# (alignn_amd/libalignn_hip.so not found: no neighbour 6)\n");
    auto spin = [&](int iters) {
        if (neighbour == 1 || neighbour == 4) hipLaunchKernelGGL(spinner_kernel, dim3(1024), dim3(256), 0, sb, spin_out, iters);
        if (neighbour == 2 || neighbour == 4) hipLaunchKernelGGL(lds_spinner_kernel, dim3(1024), dim3(256), 0, neighbour == 4 ? sc : sb, spin_out, iters * 2);
        if (neighbour == 6 && lib_gemm)
            for (int k = 0; k < 3; ++k)
                lib_gemm(pj_a, 256, pj_amax, pj_img, pj_amax + 1, nullptr, pj_add, 256, pj_c, 256, TR, 256, 256, (void*)sb);
        if (neighbour == 5) hipLaunchKernelGGL(trans_spinner_kernel, dim3(2048), dim3(256), 0, sb, spin_out, iters);
        if (neighbour == 3 || neighbour == 4) hipLaunchKernelGGL(mem_spinner_kernel, dim3(2048), dim3(256), 0, neighbour == 4 ? sd : sb, (const float4*)spin_src, spin_n, spin_out, iters / 8);
    };
    unsigned long long* d_bad;
    unsigned* d_q;
    CK(hipMalloc(&d_bad, 8));
    CK(hipMalloc(&d_q, 64));
#ifdef __FAST_MATH__
    printf("fast-math build\n");
#endif
    printf("# tools/pk_f32_repro.hip (%s)\n",
#ifdef PK_NOSLP
           "built with -fno-slp-vectorize"
#else
           "default build: SLP vectoriser on"
#endif
    );
    // ---- F
    {
        const int blocks = 4096, threads = 256, rounds = 64;
        const size_t n4 = (size_t)blocks * threads * rounds;
        float *o1, *o2;
        CK(hipMalloc(&o1, n4 * 16));
        CK(hipMalloc(&o2, n4 * 16));
        float4* cp_dst;
        CK(hipMalloc(&cp_dst, spin_n * sizeof(float4) / 2));
        bool beside_copy = false;
        auto runF = [&](const char* name, auto kern) {
            unsigned long long total = 0;
            unsigned q[4] = {0, 0, 0, 0}, cmp[4] = {0, 0, 0, 0};
            for (int r = 0; r < 5; ++r) {
                CK(hipMemset(o1, 0, n4 * 16));
                CK(hipMemset(d_bad, 0, 8));
                CK(hipMemset(d_q, 0, 64));
                CK(hipDeviceSynchronize());
                if (beside_copy)  // (a streaming copy of 512 MiB x 6 on another stream: ~2 ms of loads and stores on every CU)
                    hipLaunchKernelGGL(copy_spinner_kernel, dim3(2048), dim3(256), 0, sb, (const float4*)spin_src, cp_dst, spin_n / 2, 6);
                hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, sa, o1, o2, rounds);
                hipLaunchKernelGGL(store_war_check, dim3(1024), dim3(256), 0, sa, (const float*)o1, n4, d_bad, d_q, d_q + 4);
                CK(hipDeviceSynchronize());
                unsigned long long bad;
                unsigned qq[8];
                CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(qq, d_q, 32, hipMemcpyDeviceToHost));
                total += bad;
                for (int k = 0; k < 4; ++k) q[k] += qq[k], cmp[k] += qq[4 + k];
            }
            printf("%-78s %-20s stored values that are NOT the registers' contents at issue: %llu of %.3g   (wave quarters: %u %u %u %u; components: %u %u %u %u)\n",
                   name, beside_copy ? "beside a copy kernel" : "alone", total, 5.0 * 4 * n4, q[0], q[1], q[2], q[3], cmp[0], cmp[1], cmp[2], cmp[3]);
        };
        for (int bc = 0; bc < 2; ++bc) {
            beside_copy = bc != 0;
        runF("F  store x4; v_pk_mul_f32 x2 over its data, s_nop 0 between (1 wait state)", store_war_kernel<0, true>);
        runF("F  ... s_nop 1 (2 wait states: what hipcc's hazard recogniser leaves)", store_war_kernel<1, true>);
        runF("F  ... s_nop 2", store_war_kernel<2, true>);
        runF("F  ... s_nop 3", store_war_kernel<3, true>);
        runF("F  ... s_nop 7", store_war_kernel<7, true>);
        runF("F  store x4; v_mul_f32 x4 over its data, s_nop 0 between", store_war_kernel<0, false>);
        runF("F  ... s_nop 1", store_war_kernel<1, false>);
        runF("F  ... s_nop 3", store_war_kernel<3, false>);
        }
        CK(hipFree(cp_dst));
        CK(hipFree(o1));
        CK(hipFree(o2));
    }
    // ---- A, B
    {
        const int n = 1 << 16;
        std::vector<float> h(16 * n);
        unsigned s = 12345u;
        for (auto& v : h) {
            s = s * 1664525u + 1013904223u;
            v = ((int)(s >> 8) % 20001 - 10000) * 1.0e-3f;
        }
        float* d_in;
        CK(hipMalloc(&d_in, h.size() * 4));
        CK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        auto run = [&](const char* name, auto kern, bool with_spinner) {
            CK(hipMemset(d_bad, 0, 8));
            CK(hipMemset(d_q, 0, 64));
            CK(hipDeviceSynchronize());
            for (int r = 0; r < reps; ++r) {
                if (with_spinner) spin(4000);
                hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, sa, (const float*)d_in, n, 400, d_bad, d_q);
            }
            CK(hipDeviceSynchronize());
            unsigned long long bad;
            unsigned q[4];
            CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(q, d_q, 16, hipMemcpyDeviceToHost));
            printf("%-58s %-18s wrong results: %llu of %.3g   (wave quarters hit: %u %u %u %u)\n", name,
                   with_spinner ? nb_name[neighbour] : "alone", bad, (double)reps * 2048 * 256 * 400, q[0], q[1], q[2], q[3]);
        };
        for (int sp = 0; sp < 7; ++sp) {
            neighbour = sp;
            run("G  the kernel's whole packed tail, verbatim (bit mask of wrong components per quarter)", tail_kernel, sp);
            run("D  v_pk_add_f32 -> ds_bpermute_b32 of its result, back to back", pk_lds_kernel<0>, sp);
            run("D  ... s_nop 0 between", pk_lds_kernel<1>, sp);
            run("D  ... s_nop 3 between", pk_lds_kernel<4>, sp);
            run("A  v_pk_add_f32 -> v_pk_fma_f32 op_sel, back to back", pair_kernel<0, false>, sp);
            run("A  ... s_nop 0 between", pair_kernel<1, false>, sp);
            run("A  ... s_nop 3 between", pair_kernel<4, false>, sp);
            run("B  operands from v_exp_f32 / v_rcp_f32, back to back", pair_kernel<0, true>, sp);
            run("B  ... s_nop 3 between the packed pair", pair_kernel<4, true>, sp);
        }
    }
    // ---- C: the real kernel
    {
        const size_t n = (size_t)rows * F;
        std::vector<float> gy(n), x(n), gam(F), bet(F), st(2 * (size_t)rows);
        unsigned s = 777u;
        auto rnd = [&]() {
            s = s * 1664525u + 1013904223u;
            return ((int)(s >> 8) % 20001 - 10000) * 1.0e-4f;
        };
        for (auto& v : gy) v = rnd();
        for (auto& v : x) v = rnd() * 3.0f;
        for (auto& v : gam) v = 1.0f + rnd();
        for (auto& v : bet) v = rnd();
        for (size_t r = 0; r < (size_t)rows; ++r) st[2 * r] = rnd() * 0.1f, st[2 * r + 1] = 1.0f + rnd();
        float *d_gy, *d_x, *d_g, *d_b, *d_st, *d_o0, *d_o1, *d_part, *d_amax;
        CK(hipMalloc(&d_gy, n * 4));
        CK(hipMalloc(&d_x, n * 4));
        CK(hipMalloc(&d_g, F * 4));
        CK(hipMalloc(&d_b, F * 4));
        CK(hipMalloc(&d_st, 2 * (size_t)rows * 4));
        CK(hipMalloc(&d_o0, n * 4));
        CK(hipMalloc(&d_o1, n * 4));
        CK(hipMalloc(&d_part, (size_t)alignn_ln_slabs(rows) * 2 * F * 4));
        CK(hipMalloc(&d_amax, 256));
        CK(hipMemcpy(d_gy, gy.data(), n * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_x, x.data(), n * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_g, gam.data(), F * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_b, bet.data(), F * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_st, st.data(), 2 * (size_t)rows * 4, hipMemcpyHostToDevice));
        CK(hipMemset(d_amax, 0, 256));
        auto ln = [&](float* out) {
            int rc = alignn_ln_silu_bwd(d_gy, F, d_x, F, d_g, d_b, d_st, out, F, d_part, rows, F, d_amax, (alignn_stream_t)sa);
            if (rc) {
                printf("alignn_ln_silu_bwd rc %d\n", rc);
                exit(2);
            }
        };
        ln(d_o0);
        CK(hipDeviceSynchronize());
        for (int sp = 0; sp < 7; ++sp) {
            neighbour = sp;
            unsigned long long total = 0, runs_bad = 0;
            unsigned q[4] = {0, 0, 0, 0}, cmp[4] = {0, 0, 0, 0};
            for (int r = 0; r < reps; ++r) {
                CK(hipMemset(d_bad, 0, 8));
                CK(hipMemset(d_q, 0, 64));
                CK(hipDeviceSynchronize());
                if (sp) spin(20000);
                ln(d_o1);
                CK(hipStreamSynchronize(sa));
                hipLaunchKernelGGL(diff_kernel, dim3(1024), dim3(256), 0, sa, (const float*)d_o0, (const float*)d_o1, n, d_bad, d_q, d_q + 4);
                CK(hipDeviceSynchronize());
                unsigned long long bad;
                unsigned qq[8];
                CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(qq, d_q, 32, hipMemcpyDeviceToHost));
                total += bad;
                runs_bad += bad != 0;
                for (int k = 0; k < 4; ++k) q[k] += qq[k], cmp[k] += qq[4 + k];
            }
            printf("C  alignn_ln_silu_bwd, %d x %d, %d runs vs the first run     %-18s differing elements: %llu in %llu runs   "
                   "(wave quarters: %u %u %u %u; float4 components: %u %u %u %u)\n",
                   rows, F, reps, nb_name[sp], total, runs_bad, q[0], q[1], q[2], q[3], cmp[0], cmp[1], cmp[2],
                   cmp[3]);
        }
    }
    // ---- E: the instance that failed inside the model (tools/ff_alloc_diff2.py with a build of norm.hip without the flag: the first
    //      differing buffer is the Ux block of a bond-graph convolution's GP and its GS1 / GS0): alignn_ln_silu_bwd_node - the NODE
    //      variant with its IEEE divisions - over a few thousand atom rows, launched many times beside long-running neighbours
    {
        const int rows_e = 5760;
        const size_t n = (size_t)rows_e * F;
        std::vector<float> gy(n), x(n), gam(F), bet(F), st(2 * (size_t)rows_e), s0v(n), hv(n);
        unsigned s = 4242u;
        auto rnd = [&]() {
            s = s * 1664525u + 1013904223u;
            return ((int)(s >> 8) % 20001 - 10000) * 1.0e-4f;
        };
        for (auto& v : gy) v = rnd();
        for (auto& v : x) v = rnd() * 3.0f;
        for (auto& v : gam) v = 1.0f + rnd();
        for (auto& v : bet) v = rnd();
        for (auto& v : s0v) v = 2.0f + rnd();
        for (auto& v : hv) v = rnd();
        for (size_t r = 0; r < (size_t)rows_e; ++r) st[2 * r] = rnd() * 0.1f, st[2 * r + 1] = 1.0f + rnd();
        float *d_gy, *d_x, *d_g, *d_b, *d_st, *d_s0, *d_h, *d_out[2], *d_part, *d_amax;
        CK(hipMalloc(&d_gy, n * 4));
        CK(hipMalloc(&d_x, n * 4));
        CK(hipMalloc(&d_g, F * 4));
        CK(hipMalloc(&d_b, F * 4));
        CK(hipMalloc(&d_st, 2 * (size_t)rows_e * 4));
        CK(hipMalloc(&d_s0, n * 4));
        CK(hipMalloc(&d_h, n * 4));
        for (int k = 0; k < 2; ++k) CK(hipMalloc(&d_out[k], 6 * n * 4));  // GP [rows, 4F] (Ux block written) | GS1 | GS0
        CK(hipMalloc(&d_part, (size_t)alignn_ln_slabs(rows_e) * 2 * F * 4));
        CK(hipMalloc(&d_amax, 256));
        CK(hipMemcpy(d_gy, gy.data(), n * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_x, x.data(), n * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_g, gam.data(), F * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_b, bet.data(), F * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_st, st.data(), 2 * (size_t)rows_e * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_s0, s0v.data(), n * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_h, hv.data(), n * 4, hipMemcpyHostToDevice));
        CK(hipMemset(d_amax, 0, 256));
        for (int k = 0; k < 2; ++k) CK(hipMemset(d_out[k], 0, 6 * n * 4));
        auto lnn = [&](float* out) {
            int rc = alignn_ln_silu_bwd_node(d_gy, F, d_x, F, d_g, d_b, d_st, out + 3 * F, 4 * F, d_part, rows_e, F, d_amax, d_s0, d_h,
                                             out + 4 * n, out + 5 * n, (alignn_stream_t)sa);
            if (rc) {
                printf("alignn_ln_silu_bwd_node rc %d\n", rc);
                exit(2);
            }
        };
        lnn(d_out[0]);
        CK(hipDeviceSynchronize());
        for (int sp = 0; sp < 7; ++sp) {
            neighbour = sp;
            unsigned long long total = 0, runs_bad = 0;
            unsigned q[4] = {0, 0, 0, 0}, cmp[4] = {0, 0, 0, 0};
            const int launches = 50 * reps;
            for (int r = 0; r < launches; ++r) {
                if (r % 50 == 0) {
                    CK(hipDeviceSynchronize());
                    if (sp) spin(40000);
                }
                CK(hipMemsetAsync(d_bad, 0, 8, sa));
                CK(hipMemsetAsync(d_q, 0, 64, sa));
                lnn(d_out[1]);
                hipLaunchKernelGGL(diff_kernel, dim3(256), dim3(256), 0, sa, (const float*)d_out[0], (const float*)d_out[1], 6 * n, d_bad,
                                   d_q, d_q + 4);
                unsigned long long bad;
                unsigned qq[8];
                CK(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, sa));
                CK(hipMemcpyAsync(qq, d_q, 32, hipMemcpyDeviceToHost, sa));
                CK(hipStreamSynchronize(sa));
                total += bad;
                runs_bad += bad != 0;
                for (int k = 0; k < 4; ++k) q[k] += qq[k], cmp[k] += qq[4 + k];
            }
            printf("E  alignn_ln_silu_bwd_node, %d x %d, %d launches vs the first     %-18s differing elements: %llu in %llu launches   "
                   "(wave quarters: %u %u %u %u; float4 components: %u %u %u %u)\n",
                   rows_e, F, launches, nb_name[sp], total, runs_bad, q[0], q[1], q[2], q[3], cmp[0], cmp[1], cmp[2], cmp[3]);
        }
    }
    return 0;
}
