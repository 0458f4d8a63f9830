// Probe of gfx950's LDS transpose read (ds_read_b64_tr_b16) for csrc/gemm_dw.hip:
//  (1) semantics: which 16-bit LDS element lands in (lane, j) for arbitrary per-lane addresses;
//  (2) the MFMA-operand addressing the fused input-gradient + weight-gradient kernel uses (a row-major [rows][256] fp16 plane
//      read as the A operand of G^T: lane = feature, registers = 8 consecutive rows), checked element by element;
//  (3) cycles per read for that pattern under three row swizzles (bank conflicts).
// hipcc --offload-arch=gfx950 -O3 tools/tr_probe.hip -o tools/_tr_probe && tools/_tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u2 tr_read(unsigned addr) {
    u2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

// (1) lds[i] = i (u16); lane l reads at byte address addr[l]; out[l][0..3] = the four 16-bit results
__global__ void probe_kernel(const unsigned* __restrict__ addr, unsigned short* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const u2 v = tr_read((unsigned)(size_t)lds + addr[threadIdx.x]);
    out[threadIdx.x * 4 + 0] = (unsigned short)(v.x & 0xffff);
    out[threadIdx.x * 4 + 1] = (unsigned short)(v.x >> 16);
    out[threadIdx.x * 4 + 2] = (unsigned short)(v.y & 0xffff);
    out[threadIdx.x * 4 + 3] = (unsigned short)(v.y >> 16);
}

// swizzle of the 16-byte slot index inside a 512-byte row half (32 slots): XOR of the low four bits with g(row)
__host__ __device__ inline int swz(int mode, int row) {
    if (mode == 0) return 0;
    if (mode == 1) return row & 15;
    return ((row & 3) << 2) | ((row >> 2) & 3);
}
// byte offset of (row, feature f) in a plane image [row][1024 B: hi 512 | lo 512]
__host__ __device__ inline unsigned plane_off(int mode, int row, int f) {
    const int slot = (f >> 3) ^ swz(mode, row);
    return (unsigned)(row * 1024 + slot * 16 + (f & 7) * 2);
}

// (2)+(3): 64 rows x 256 features; element (r, f) = r * 256 + f.  A wave reads the A operand of G^T for m-block mb (32 features)
// and k-block kb (16 rows): lane l: feature 32 mb + l % 32, rows 16 kb + 8 (l / 32) + 0..7.
__global__ void operand_kernel(int mode, int iters, unsigned short* __restrict__ out, long long* __restrict__ cycles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 64 * 256; i += blockDim.x) {
        const int r = i >> 8, f = i & 255;
        *reinterpret_cast<unsigned short*>(smem + plane_off(mode, r, f)) = (unsigned short)i;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, i16 = lane & 15, grp = lane >> 4;
    unsigned acc = 0;
    long long t0 = 0, t1 = 0;
    for (int rep = 0; rep < 2; ++rep) {  // rep 1 is timed
        if (rep) t0 = __builtin_readcyclecounter();
        for (int it = 0; it < (rep ? iters : 1); ++it)
            for (int kb = 0; kb < 4; ++kb)
                for (int mb = 0; mb < 8; ++mb) {
                    const int fbase = 32 * mb + 16 * (grp & 1), kbase = 16 * kb + 8 * (grp >> 1);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int row = kbase + 4 * h + (i16 >> 2), f = fbase + 4 * (i16 & 3);
                        const u2 v = tr_read((unsigned)(size_t)smem + plane_off(mode, row, f));
                        if (rep == 0 && threadIdx.x < 64) {
                            unsigned short* o = out + (((kb * 8 + mb) * 64 + lane) * 8 + 4 * h);
                            o[0] = (unsigned short)(v.x & 0xffff), o[1] = (unsigned short)(v.x >> 16);
                            o[2] = (unsigned short)(v.y & 0xffff), o[3] = (unsigned short)(v.y >> 16);
                        }
                        acc += v.x ^ v.y;
                    }
                }
        if (rep) t1 = __builtin_readcyclecounter();
    }
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc == 0x12345u) out[0] = 1;
}

// (3b) the plain reads of the same image (input-gradient A operand: lane = row, 8 consecutive features): ds_read_b128
__global__ void plain_kernel(int mode, int iters, long long* __restrict__ cycles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 64 * 256; i += blockDim.x) {
        const int r = i >> 8, f = i & 255;
        *reinterpret_cast<unsigned short*>(smem + plane_off(mode, r, f)) = (unsigned short)i;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, il = lane & 31, half = lane >> 5;
    unsigned acc = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it)
        for (int kt = 0; kt < 16; ++kt)
            for (int a = 0; a < 2; ++a) {
                uint4 v;
                const unsigned ad = (unsigned)(size_t)smem + plane_off(mode, 32 * a + il, 16 * kt + 8 * half);
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ad) : "memory");
                acc += v.x ^ v.w;
            }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc == 0x12345u) cycles[1] = 1;
}

int main() {
    unsigned* d_addr;
    unsigned short* d_out;
    long long* d_cyc;
    (void)hipMalloc(&d_addr, 64 * 4);
    (void)hipMalloc(&d_out, 4 * 8 * 64 * 8 * 2 + 1024);
    (void)hipMalloc(&d_cyc, 64 * 8);
    // ---- (1) semantics
    struct Pat { const char* name; unsigned a[64]; } pats[3];
    pats[0].name = "addr = 8 l (contiguous)";
    for (int l = 0; l < 64; ++l) pats[0].a[l] = 8 * l;
    pats[1].name = "addr = 512 (l%16 / 4) + 8 (l%4) + 2048 (l/16)   (4 rows of 512 B per 16-lane group)";
    for (int l = 0; l < 64; ++l) pats[1].a[l] = 512 * ((l & 15) >> 2) + 8 * (l & 3) + 2048 * (l >> 4);
    pats[2].name = "addr = 1000 (uniform)";
    for (int l = 0; l < 64; ++l) pats[2].a[l] = 1000;
    for (auto& p : pats) {
        (void)hipMemcpy(d_addr, p.a, sizeof(p.a), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        unsigned short h[256];
        (void)hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
        printf("# %s\n", p.name);
        // model: inside a 16-lane group, lane i element j = the 16-bit word (i % 4) of lane (4 j + i / 4)'s 8 bytes
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                const int g = l & ~15, i = l & 15, srcl = g + 4 * j + (i >> 2);
                const unsigned expect = p.a[srcl] / 2 + (i & 3);
                if (h[l * 4 + j] != expect) ++bad;
            }
        printf("model 'lane i elem j <- lane 4j + i/4, word i%%4': %s (%d mismatches)\n", bad ? "WRONG" : "holds", bad);
        for (int l = 0; l < 64; l += (bad ? 1 : 21)) printf("  lane %2d: %5u %5u %5u %5u\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    // ---- (2) operand addressing, (3) cycles
    for (int mode = 0; mode < 3; ++mode) {
        const int iters = 200;
        hipLaunchKernelGGL(operand_kernel, dim3(1), dim3(64), 65536, 0, mode, iters, d_out, d_cyc);
        std::vector<unsigned short> h(4 * 8 * 64 * 8);
        long long cyc1 = 0, cyc8 = 0, cycp1 = 0, cycp8 = 0;
        (void)hipMemcpy(h.data(), d_out, h.size() * 2, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&cyc1, d_cyc, 8, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int kb = 0; kb < 4; ++kb)
            for (int mb = 0; mb < 8; ++mb)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 8; ++e) {
                        const int f = 32 * mb + (l & 31), r = 16 * kb + 8 * (l >> 5) + e;
                        if (h[((kb * 8 + mb) * 64 + l) * 8 + e] != (unsigned short)(r * 256 + f)) ++bad;
                    }
        hipLaunchKernelGGL(operand_kernel, dim3(1), dim3(512), 65536, 0, mode, iters, d_out, d_cyc);
        (void)hipMemcpy(&cyc8, d_cyc, 8, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(plain_kernel, dim3(1), dim3(64), 65536, 0, mode, iters, d_cyc);
        (void)hipMemcpy(&cycp1, d_cyc, 8, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(plain_kernel, dim3(1), dim3(512), 65536, 0, mode, iters, d_cyc);
        (void)hipMemcpy(&cycp8, d_cyc, 8, hipMemcpyDeviceToHost);
        printf("swizzle mode %d: G^T operand via 2 tr reads: %s (%d wrong of %zu); cycles per dependent tr read: %.1f (1 wave) %.1f (8 waves); "
               "per dependent ds_read_b128 of the row-major operand: %.1f (1 wave) %.1f (8 waves)\n",
               mode, bad ? "WRONG" : "exact", bad, h.size(), (double)cyc1 / (iters * 64.0), (double)cyc8 / (iters * 64.0),
               (double)cycp1 / (iters * 32.0), (double)cycp8 / (iters * 32.0));
    }
    return 0;
}
