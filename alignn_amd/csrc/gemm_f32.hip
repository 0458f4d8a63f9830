// fp32 GEMMs on the CDNA4 matrix cores: v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate, exact
// fp32 == an fmaf chain; 157 TF peak on MI355X - there is no TF32/xf32 on gfx950).
//
// One kernel template covers the three products a Linear layer needs:
//   NT  C[M,N] = A[M,K] W[N,K]^T (+bias +addend)   forward      (alignn/models/alignn.py:98-110,175,341)
//   NN  C[M,K] = G[M,N] W[N,K]   (+addend)         input grad
//   TN  dW[N,K] = G[M,N]^T A[M,K]                  weight grad  (split over M, fixed-order slab sum)
// by describing each operand as either "reduction-contiguous" (RC: element (i, r) at S[i*ld + r])
// or "index-contiguous" (IC: element (i, r) at S[r*ld + i]).  Tiles are staged global -> registers
// -> LDS (double buffered, one barrier per 32-deep step) as whole 16-byte vectors; RC tiles are
// stored [i][36] (144-byte rows: conflict-free ds_read_b128 of 4 consecutive r), IC tiles [r][BMN]
// (conflict-free ds_read_b32 across lanes).  Because a sum over r may be taken in any order as
// long as both operands agree, MFMA step s of a 32-deep block uses r = 8*(s/4) + 4*(lane/32) + s%4,
// which lets an RC operand fetch four steps' worth of values with one ds_read_b128.
#include "common.h"
#include "../../include/alignn_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int RC_LD = BK + 4;  // padded row of an RC tile (floats)

template <int BMN, bool RC>
struct TileShape {
    static constexpr int kFloats = RC ? BMN * RC_LD : BK * BMN;
};

// global -> registers -> LDS staging of one operand tile (BMN indices x BK reduction steps)
template <int BMN, bool RC, int NT>
struct Stager {
    static constexpr int NV = (BMN * BK / 4) / NT;
    static_assert((BMN * BK / 4) % NT == 0, "tile must divide over the block");
    float4 v[NV];

    __device__ __forceinline__ void load(const float* __restrict__ S, int64_t ld, int64_t i0, int64_t i_ext,
                                         int64_t r0, int64_t r_ext, int t) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int id = t + j * NT;
            int64_t gi, gr;
            if (RC) {
                gi = i0 + id / (BK / 4);
                gr = r0 + (id % (BK / 4)) * 4;
            } else {
                gr = r0 + id / (BMN / 4);
                gi = i0 + (id % (BMN / 4)) * 4;
            }
            if (gi < i_ext && gr < r_ext)
                v[j] = RC ? f4_ld(S + gi * ld + gr) : f4_ld(S + gr * ld + gi);
            else
                v[j] = f4_zero();
        }
    }
    __device__ __forceinline__ void store(float* T, int t) const {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int id = t + j * NT;
            if (RC)
                f4_st(T + (id / (BK / 4)) * RC_LD + (id % (BK / 4)) * 4, v[j]);
            else
                f4_st(T + (id / (BMN / 4)) * BMN + (id % (BMN / 4)) * 4, v[j]);
        }
    }
};

// fetch the fragment values of 4 consecutive MFMA steps (group j) for index row `irow`
template <int BMN, bool RC>
__device__ __forceinline__ void fetch4(const float* T, int irow, int j, int half, float (&out)[4]) {
    if (RC) {
        float4 q = f4_ld(T + irow * RC_LD + 8 * j + 4 * half);
        out[0] = q.x;
        out[1] = q.y;
        out[2] = q.z;
        out[3] = q.w;
    } else {
        const float* p = T + (8 * j + 4 * half) * BMN + irow;
        out[0] = p[0];
        out[1] = p[BMN];
        out[2] = p[2 * BMN];
        out[3] = p[3 * BMN];
    }
}

struct GemmArgs {
    const float* A;
    int64_t lda;
    const float* B;
    int64_t ldb;
    const float* bias;    // [No] or null
    const float* addend;  // [Mo, No] or null
    int64_t ldadd;
    float* C;
    int64_t ldc;
    int64_t Mo;      // output rows
    int No;          // output cols
    int64_t R;       // reduction length
    int64_t r_chunk; // reduction rows per blockIdx.z (split-R); C advances by c_split per z
    int64_t c_split;
    int xcd_group;   // != 0: 1-D launch of tiles*splits workgroups, remapped so that all output tiles of one
                     // reduction slab run back-to-back on ONE XCD (they re-read the same G/A rows: L2 hits)
    int tiles_m, tiles_n, splits;
};

template <int BM, int BN, int WM, int WN, bool A_RC, bool B_RC, bool HAS_ADD>
__global__ __launch_bounds__(WM * WN * 64) void gemm_mfma_kernel(GemmArgs g) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int RM = TM / 32, RN = TN / 32;
    constexpr int A_FL = TileShape<BM, A_RC>::kFloats, B_FL = TileShape<BN, B_RC>::kFloats;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int STAGE = A_FL + B_FL;  // stage s: A tile at smem + s*STAGE, B tile right behind it

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int il = lane & 31, half = lane >> 5;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (g.xcd_group) {
        // hardware deals workgroup L to XCD L % 8.  Give XCD c the slabs z = c, c+8, ...; within an XCD the
        // tiles of a slab are consecutive in dispatch order.
        const int tiles = g.tiles_m * g.tiles_n;
        const int L = blockIdx.x;
        const int xcd = L & 7, j = L >> 3;          // j-th workgroup of this XCD
        bz = xcd + 8 * (j / tiles);
        const int tile = j % tiles;
        bx = tile % g.tiles_m;
        by = tile / g.tiles_m;
        if (bz >= g.splits) return;                  // padding workgroups (uniform per block)
    }
    const int64_t m0 = (int64_t)bx * BM;
    const int64_t n0 = (int64_t)by * BN;
    const int64_t rbeg = (int64_t)bz * g.r_chunk;
    int64_t rend = rbeg + g.r_chunk;
    if (rend > g.R) rend = g.R;
    float* C = g.C + (int64_t)bz * g.c_split;

    f32x16 acc[RM][RN];
#pragma unroll
    for (int a = 0; a < RM; ++a)
#pragma unroll
        for (int b = 0; b < RN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    const bool vec = ((g.No & 3) == 0) && ((g.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
                     (g.addend == nullptr || (((g.ldadd & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.addend) & 15) == 0)));
    // vector epilogue: the bias is fetched HERE, ahead of the whole k-loop, so that no load sits between the
    // epilogue's stores (hipcc would put a vmcnt(0) - a wait for the previous STORE - in front of every store)
    const int64_t ecol = n0 + wn * TN + (lane & 15) * 4;
    float4 ebias = f4_zero();
    if (vec && g.bias && ecol < g.No) ebias = f4_ld(g.bias + ecol);

    Stager<BM, A_RC, NT> sa;
    Stager<BN, B_RC, NT> sb;
    const int nk = rend > rbeg ? (int)((rend - rbeg + BK - 1) / BK) : 0;
    if (nk > 0) {
        sa.load(g.A, g.lda, m0, g.Mo, rbeg, rend, t);
        sb.load(g.B, g.ldb, n0, g.No, rbeg, rend, t);
        sa.store(smem, t);
        sb.store(smem + A_FL, t);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            sa.load(g.A, g.lda, m0, g.Mo, rbeg + (int64_t)(kt + 1) * BK, rend, t);
            sb.load(g.B, g.ldb, n0, g.No, rbeg + (int64_t)(kt + 1) * BK, rend, t);
        }
        const float* At = smem + cur * STAGE;
        const float* Bt = At + A_FL;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float af[RM][4], bf[RN][4];
#pragma unroll
            for (int a = 0; a < RM; ++a) fetch4<BM, A_RC>(At, wm * TM + a * 32 + il, j, half, af[a]);
#pragma unroll
            for (int b = 0; b < RN; ++b) fetch4<BN, B_RC>(Bt, wn * TN + b * 32 + il, j, half, bf[b]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int a = 0; a < RM; ++a)
#pragma unroll
                    for (int b = 0; b < RN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][s], bf[b][s], acc[a][b], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            sa.store(smem + (cur ^ 1) * STAGE, t);
            sb.store(smem + (cur ^ 1) * STAGE + A_FL, t);
        }
        __syncthreads();
    }

    // ---- epilogue.  acc reg r of a 32x32 tile sits at row (r&3) + 8*(r>>2) + 4*half, col lane&31.
    // Fast path: each wave transposes its 32-row slabs through a private LDS patch ([32][TN+4] floats,
    // carved out of the now idle staging buffers) and then moves whole 256-byte row segments as float4:
    // bias / addend loads and the C stores are 16 B per lane and row-contiguous.
    if (vec) {
        constexpr int PLD = TN + 4;
        static_assert(WM * WN * 32 * PLD <= 2 * STAGE, "epilogue patch must fit in the staging LDS");
        float* patch = smem + wave * (32 * PLD);
        const int prow = lane >> 4, pc4 = (lane & 15) * 4;  // TN == 64: 16 lanes cover one row segment
        static_assert(TN == 64, "epilogue assumes 64-column wave tiles");
        const int64_t col = ecol;
        const int64_t colc = col < g.No ? col : 0;
        const float4 bv = ebias;
#pragma unroll
        for (int a = 0; a < RM; ++a) {
            __syncthreads();  // staging buffers (or the previous slab) no longer read
#pragma unroll
            for (int b = 0; b < RN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    patch[((r & 3) + 8 * (r >> 2) + 4 * half) * PLD + b * 32 + il] = acc[a][b][r];
            __syncthreads();
            float4 ov[8], av[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) ov[i] = f4_ld(patch + (i * 4 + prow) * PLD + pc4);
            const int64_t row0 = m0 + wm * TM + a * 32 + prow;
            if (HAS_ADD) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    int64_t row = row0 + i * 4;
                    if (row >= g.Mo) row = g.Mo - 1;
                    av[i] = f4_ld(g.addend + row * g.ldadd + colc);
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t row = row0 + i * 4;
                float4 v = f4_add(ov[i], bv);
                if (HAS_ADD) v = f4_add(v, av[i]);
                if (row < g.Mo && col < g.No) f4_st(C + row * g.ldc + col, v);
            }
        }
        return;
    }
#pragma unroll
    for (int a = 0; a < RM; ++a) {
#pragma unroll
        for (int b = 0; b < RN; ++b) {
            const int64_t col = n0 + wn * TN + b * 32 + il;
            if (col >= g.No) continue;
            const float bv = g.bias ? g.bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wm * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < g.Mo) {
                    float v = acc[a][b][r] + bv;
                    if (HAS_ADD) v += g.addend[row * g.ldadd + col];
                    C[row * g.ldc + col] = v;
                }
            }
        }
    }
}

// Plain VALU fallback for shapes the MFMA path does not take (tiny or not multiples of 4):
// element (i, r) of A at A[i*sai + r*sar], of B at B[j*sbj + r*sbr].
__global__ void gemm_naive_kernel(const float* __restrict__ A, int64_t sai, int64_t sar, const float* __restrict__ B,
                                  int64_t sbj, int64_t sbr, const float* __restrict__ bias,
                                  const float* __restrict__ addend, int64_t ldadd, float* __restrict__ C, int64_t ldc,
                                  int64_t Mo, int No, int64_t R) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Mo * No) return;
    int64_t i = idx / No;
    int j = (int)(idx - i * No);
    float acc = 0.0f;
    for (int64_t r = 0; r < R; ++r) acc = fmaf(A[i * sai + r * sar], B[j * sbj + r * sbr], acc);
    if (bias) acc += bias[j];
    if (addend) acc += addend[i * ldadd + j];
    C[i * ldc + j] = acc;
}

// The same for few outputs with a long contiguous reduction (the one-wide readout fc: [B, 256] x [256] -> [B]: 30 us on the
// kernel above, one thread walking 256 dependent loads per output): one wavefront per output, lane l takes r = l, l + 64, ...,
// fixed-order butterfly over the lanes.
__global__ __launch_bounds__(256) void gemm_naive_wave_kernel(const float* __restrict__ A, int64_t sai, const float* __restrict__ B,
                                                              int64_t sbj, const float* __restrict__ bias,
                                                              const float* __restrict__ addend, int64_t ldadd, float* __restrict__ C,
                                                              int64_t ldc, int64_t Mo, int No, int64_t R) {
    const int lane = threadIdx.x & 63;
    const int64_t idx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= Mo * No) return;  // (wave-uniform)
    const int64_t i = idx / No;
    const int j = (int)(idx - i * No);
    float acc = 0.0f;
    for (int64_t r = lane; r < R; r += 64) acc = fmaf(A[i * sai + r], B[j * sbj + r], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) {
        if (bias) acc += bias[j];
        if (addend) acc += addend[i * ldadd + j];
        C[i * ldc + j] = acc;
    }
}

// out[idx] = sum_k ws[k][idx], 32 outputs x 32 slab-lanes per block, fp64 partials, fixed order
__global__ __launch_bounds__(1024) void slab_reduce_kernel(const float* __restrict__ ws, int splits, int64_t count,
                                                           int cols, float* __restrict__ out, int64_t ldo) {
    __shared__ double sh[32][32];
    const int64_t idx = (int64_t)blockIdx.x * 32 + threadIdx.x;
    double s = 0.0;
    if (idx < count)
        for (int k = threadIdx.y; k < splits; k += 32) s += (double)ws[(int64_t)k * count + idx];
    sh[threadIdx.y][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y == 0 && idx < count) {
        double t = 0.0;
#pragma unroll 8
        for (int k = 0; k < 32; ++k) t += sh[k][threadIdx.x];
        out[(idx / cols) * ldo + (idx % cols)] = (float)t;
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// same sum, four consecutive outputs per thread: 64 output lanes x 16 slab lanes per workgroup, every slab row read
// as 1 KiB contiguous per wave, four loads in flight per lane; fp64 partials combined through LDS in lane order
__global__ __launch_bounds__(1024) void slab_reduce4_kernel(const float* __restrict__ ws, int splits, int64_t count,
                                                            int cols, float* __restrict__ out, int64_t ldo) {
    __shared__ double sh[16][64][4];
    const int64_t idx = ((int64_t)blockIdx.x * 64 + threadIdx.x) * 4;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (idx < count) {
        int k = threadIdx.y;
        for (; k + 48 < splits; k += 64) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = f4_ld(ws + (int64_t)(k + 16 * u) * count + idx);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s0 += (double)v[u].x;
                s1 += (double)v[u].y;
                s2 += (double)v[u].z;
                s3 += (double)v[u].w;
            }
        }
        for (; k < splits; k += 16) {
            float4 v = f4_ld(ws + (int64_t)k * count + idx);
            s0 += (double)v.x;
            s1 += (double)v.y;
            s2 += (double)v.z;
            s3 += (double)v.w;
        }
    }
    double* mine = sh[threadIdx.y][threadIdx.x];
    mine[0] = s0, mine[1] = s1, mine[2] = s2, mine[3] = s3;
    __syncthreads();
    if (threadIdx.y == 0 && idx < count) {
        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll 4
        for (int k = 0; k < 16; ++k) {
            const double* p = sh[k][threadIdx.x];
            t0 += p[0], t1 += p[1], t2 += p[2], t3 += p[3];
        }
        f4_st(out + (idx / cols) * ldo + (idx % cols), make_float4((float)t0, (float)t1, (float)t2, (float)t3));
    }
}

// out[N,K] (leading dimension ldo) = sum over `splits` slabs of ws[k][N*K]
inline void launch_slab_reduce(const float* ws, int splits, int64_t count, int cols, float* out, int64_t ldo,
                               hipStream_t st) {
    if ((count & 3) == 0 && (cols & 3) == 0 && (ldo & 3) == 0 && aligned16(out) && aligned16(ws))
        hipLaunchKernelGGL(slab_reduce4_kernel, dim3(alignn_ceil_div(count, 256)), dim3(64, 16), 0, st, ws, splits, count,
                           cols, out, ldo);
    else
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(alignn_ceil_div(count, 32)), dim3(32, 32), 0, st, ws, splits, count,
                           cols, out, ldo);
}

template <int BM, int BN, int WM, int WN, bool A_RC, bool B_RC, bool HAS_ADD>
int launch_k(const GemmArgs& g, int splits, hipStream_t stream);

template <int BM, int BN, int WM, int WN, bool A_RC, bool B_RC>
int launch(const GemmArgs& g, int splits, hipStream_t stream) {
    return g.addend ? launch_k<BM, BN, WM, WN, A_RC, B_RC, true>(g, splits, stream)
                    : launch_k<BM, BN, WM, WN, A_RC, B_RC, false>(g, splits, stream);
}

template <int BM, int BN, int WM, int WN, bool A_RC, bool B_RC, bool HAS_ADD>
int launch_k(const GemmArgs& g, int splits, hipStream_t stream) {
    constexpr int NT = WM * WN * 64;
    constexpr size_t lds = 2 * (TileShape<BM, A_RC>::kFloats + TileShape<BN, B_RC>::kFloats) * sizeof(float);
    auto kern = gemm_mfma_kernel<BM, BN, WM, WN, A_RC, B_RC, HAS_ADD>;
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid(alignn_ceil_div(g.Mo, BM), alignn_ceil_div(g.No, BN), splits);
    if (g.xcd_group) {
        GemmArgs h = g;
        h.tiles_m = grid.x;
        h.tiles_n = grid.y;
        h.splits = splits;
        const int per_xcd = alignn_ceil_div(splits, 8) * h.tiles_m * h.tiles_n;
        hipLaunchKernelGGL(kern, dim3(per_xcd * 8), dim3(NT), lds, stream, h);
        ALIGNN_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, stream, g);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int naive(const float* A, int64_t sai, int64_t sar, const float* B, int64_t sbj, int64_t sbr, const float* bias,
          const float* addend, int64_t ldadd, float* C, int64_t ldc, int64_t Mo, int No, int64_t R,
          hipStream_t stream) {
    if (Mo * No == 0) return 0;
    if (sar == 1 && sbr == 1 && R >= 128 && Mo * No <= 4096) {  // few outputs, long contiguous reduction: a wave per output
        hipLaunchKernelGGL(gemm_naive_wave_kernel, dim3(alignn_ceil_div(Mo * No, 4)), dim3(256), 0, stream, A, sai, B, sbj, bias,
                           addend, ldadd, C, ldc, Mo, No, R);
        ALIGNN_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(gemm_naive_kernel, dim3(alignn_ceil_div(Mo * No, 256)), dim3(256), 0, stream, A, sai, sar, B,
                       sbj, sbr, bias, addend, ldadd, C, ldc, Mo, No, R);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}


// Weight-gradient GEMM: the output is only (N/128)*(K/BN) tiles, so the reduction over the M rows is split
// into slabs until there are ~768 workgroups (3 per CU); each slab is a multiple of BK rows.
inline int64_t tn_chunk(int64_t M, int N, int K) {
    const int bn = K > 64 ? 128 : 64;
    const int bm = (N <= 64 && K <= 64) ? 64 : 128;
    const int64_t tiles = (int64_t)alignn_ceil_div(N, bm) * alignn_ceil_div(K, bn);
    int64_t want = (768 + tiles - 1) / tiles;
    if (want < 1) want = 1;
    int64_t chunk = (M + want - 1) / want;
    chunk = ((chunk + BK - 1) / BK) * BK;
    if (chunk < 4 * BK) chunk = 4 * BK;
    return chunk;
}
inline int tn_splits(int64_t M, int N, int K) {
    const int64_t c = tn_chunk(M, N, K);
    int64_t s = (M + c - 1) / c;
    return (int)(s < 1 ? 1 : s);
}

}  // namespace

extern "C" {

int alignn_gemm_nt(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, const float* addend,
                   int64_t ldadd, float* C, int64_t ldc, int64_t M, int N, int K, alignn_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0) return (int)hipErrorInvalidValue;
    if (M == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const bool vec_ok = (K % 4 == 0) && (lda % 4 == 0) && (ldw % 4 == 0) && aligned16(A) && aligned16(W);
    if (!vec_ok || N < 16)
        return naive(A, lda, 1, W, ldw, 1, bias, addend, ldadd, C, ldc, M, N, K, st);
    GemmArgs g{A, lda, W, ldw, bias, addend, ldadd, C, ldc, M, N, K, K, 0, 0, 0, 0, 0};
    // tile choice: the big 128x256 tile (A read once, 1 workgroup/CU) needs >= ~4 waves of workgroups to
    // amortise its tail; mid-size problems take 64x128 tiles (2-3 workgroups/CU, finer granularity)
    const int64_t big_blocks = (int64_t)alignn_ceil_div(M, 128) * alignn_ceil_div(N, 256);
    if (N > 128 && big_blocks >= 1024) return launch<128, 256, 2, 4, true, true>(g, 1, st);
    if (N > 64) return launch<64, 128, 2, 2, true, true>(g, 1, st);
    return launch<128, 64, 4, 1, true, true>(g, 1, st);
}

int alignn_gemm_nn(const float* G, int64_t ldg, const float* W, int64_t ldw, const float* addend, int64_t ldadd,
                   float* C, int64_t ldc, int64_t M, int N, int K, alignn_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0) return (int)hipErrorInvalidValue;
    if (M == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    // C[M,K] = sum_n G[m,n] W[n,k]: A=G reduction-contiguous, B=W index-contiguous (index k)
    const bool vec_ok = (N % 4 == 0) && (K % 4 == 0) && (ldg % 4 == 0) && (ldw % 4 == 0) && aligned16(G) && aligned16(W);
    if (!vec_ok || K < 16)
        return naive(G, ldg, 1, W, 1, ldw, nullptr, addend, ldadd, C, ldc, M, K, N, st);
    GemmArgs g{G, ldg, W, ldw, nullptr, addend, ldadd, C, ldc, M, K, N, N, 0, 0, 0, 0, 0};
    if (K > 128) return launch<128, 256, 2, 4, true, false>(g, 1, st);
    if (K > 64) return launch<128, 128, 2, 2, true, false>(g, 1, st);
    return launch<128, 64, 4, 1, true, false>(g, 1, st);
}

// ---- split reduction for C[M,K] = G[M,N] W[N,K] (+ addend) with FEW output tiles and a LONG reduction: the input gradient
// of the fused node projection of a bond-graph convolution is [3 840 x 1 024] x [1 024 x 256] - 30 tiles of 128 x 256
// on 256 CUs, each walking 1 024 reduction steps (52 us); eight slabs of 128 make it 240 workgroups (~12 us) plus one
// pass that adds the slabs in a fixed order (and the addend once).
static int nn_splits(int64_t M, int N, int K) {
    const int64_t tiles = (int64_t)alignn_ceil_div(M, 128) * alignn_ceil_div(K, 256);
    if (K <= 128 || tiles >= 128 || N < 512 || (N % 128) != 0) return 1;
    int s = N / 128;
    while (s > 8) s >>= 1;
    return (N % s == 0 && (N / s) % BK == 0) ? s : 1;
}

namespace {
__global__ __launch_bounds__(256) void splitk_sum_kernel(const float* __restrict__ ws, int splits, int64_t slab, int cols,
                                                         const float* __restrict__ addend, int64_t ldadd,
                                                         float* __restrict__ out, int64_t ldo, int64_t rows) {
    const int Q = cols >> 2;
    const RowQuad rq(Q);
    const int64_t total = rows * Q;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t r;
        int q;
        rq.split(i, total, r, q);
        float4 s = addend ? f4_ld(addend + r * ldadd + q * 4) : f4_zero();
        for (int z = 0; z < splits; ++z) s = f4_add(s, f4_ld(ws + (int64_t)z * slab + r * cols + q * 4));
        f4_st(out + r * ldo + q * 4, s);
    }
}
}  // namespace

size_t alignn_gemm_nn_split_workspace(int64_t M, int N, int K) {
    const int s = nn_splits(M, N, K);
    return s > 1 ? (size_t)s * (size_t)M * (size_t)K * sizeof(float) : 0;
}

int alignn_gemm_nn_split(const float* G, int64_t ldg, const float* W, int64_t ldw, const float* addend, int64_t ldadd,
                         float* C, int64_t ldc, int64_t M, int N, int K, void* workspace, size_t workspace_bytes,
                         alignn_stream_t stream) {
    const int splits = nn_splits(M, N, K);
    const bool vec_ok = (N % 4 == 0) && (K % 4 == 0) && (ldg % 4 == 0) && (ldw % 4 == 0) && (ldc % 4 == 0) && aligned16(G) &&
                        aligned16(W) && aligned16(C) && (addend == nullptr || ((ldadd % 4 == 0) && aligned16(addend)));
    if (splits <= 1 || !vec_ok || workspace == nullptr || workspace_bytes < alignn_gemm_nn_split_workspace(M, N, K) ||
        !aligned16(workspace))
        return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    GemmArgs g{G, ldg, W, ldw, nullptr, nullptr, 0, (float*)workspace, K, M, K, N, N / splits, M * (int64_t)K, 0, 0, 0, 0};
    int rc = launch<128, 256, 2, 4, true, false>(g, splits, st);
    if (rc) return rc;
    const int64_t quads = M * (int64_t)(K >> 2);
    int grid = (int)((quads + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(splitk_sum_kernel, dim3(grid), dim3(256), 0, st, (const float*)workspace, splits, M * (int64_t)K, K,
                       addend, ldadd, C, ldc, M);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

// bf16x6 variant (csrc/gemm_x6.hip) for the large, aligned weight gradients
int alignn_gemm_tn_x6_supported(int64_t M, int N, int K);
size_t alignn_gemm_tn_x6_workspace(int64_t M, int N, int K);
int alignn_gemm_tn_x6_splits(int64_t M, int N, int K);
static bool tn_use_x6(const float* G, int64_t ldg, const float* A, int64_t lda, int64_t M, int N, int K) {
    return alignn_gemm_tn_x6_supported(M, N, K) && (ldg % 4 == 0) && (lda % 4 == 0) && aligned16(G) && aligned16(A);
}

size_t alignn_gemm_tn_workspace(int64_t M, int N, int K) {
    const size_t f32 = (size_t)tn_splits(M, N, K) * (size_t)N * (size_t)K * sizeof(float);
    if (alignn_gemm_tn_x6_supported(M, N, K)) {
        const size_t x6 = alignn_gemm_tn_x6_workspace(M, N, K);
        return x6 > f32 ? x6 : f32;
    }
    return f32;
}

int alignn_gemm_tn(const float* G, int64_t ldg, const float* g_amax, const float* A, int64_t lda, const float* a_amax,
                   float* dW, int64_t lddw, int64_t M, int N, int K, void* workspace, size_t workspace_bytes,
                   alignn_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    // dW[n,k] = sum_m G[m,n] A[m,k]: both operands index-contiguous, reduction over rows m
    const bool vec_ok = (N % 4 == 0) && (K % 4 == 0) && (ldg % 4 == 0) && (lda % 4 == 0) && aligned16(G) && aligned16(A);
    if (!vec_ok || M == 0) return naive(G, 1, ldg, A, 1, lda, nullptr, nullptr, 0, dW, lddw, N, K, M, st);
    if (workspace_bytes < alignn_gemm_tn_workspace(M, N, K) || workspace == nullptr) return (int)hipErrorInvalidValue;
    float* ws = (float*)workspace;
    if (tn_use_x6(G, ldg, A, lda, M, N, K)) {
        const bool f16 = g_amax != nullptr && a_amax != nullptr;  // both maxima known: three fp16-slice products
        int rc6 = alignn_gemm_tn_x6_partials(G, ldg, f16 ? g_amax : nullptr, A, lda, f16 ? a_amax : nullptr, M, N, K,
                                             workspace, workspace_bytes, stream);
        if (rc6) return rc6;
        const int64_t count6 = (int64_t)N * K;
        launch_slab_reduce(ws, alignn_gemm_tn_x6_splits(M, N, K), count6, K, dW, lddw, st);
        ALIGNN_CHECK_LAUNCH();
        return 0;
    }
    const int splits = tn_splits(M, N, K);
    GemmArgs g{G, ldg, A, lda, nullptr, nullptr, 0, ws, K, N, K, M, tn_chunk(M, N, K), (int64_t)N * K, 1, 0, 0, 0};
    int rc;
    if (K > 64)
        rc = launch<128, 128, 2, 2, false, false>(g, splits, st);
    else if (N > 64)
        rc = launch<128, 64, 4, 1, false, false>(g, splits, st);
    else  // the first embedding layers (64 x 40): a 128-row tile would spend half its fp32 MFMAs on padding
        rc = launch<64, 64, 2, 1, false, false>(g, splits, st);
    if (rc) return rc;
    const int64_t count = (int64_t)N * K;
    launch_slab_reduce(ws, splits, count, K, dW, lddw, st);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
