// RBF expansion fused with the first MLPLayer of an embedding:  y = silu(BatchNorm(rbf(d) W^T + b))
//
// Reference: the first two modules of ALIGNN's edge / angle embeddings, nn.Sequential(RBFExpansion(bins),
// MLPLayer(bins, F), ...) (alignn/models/alignn.py:201-222; RBFExpansion.forward alignn/models/utils.py:40-44,
// MLPLayer alignn/models/alignn.py:170-184) and torch.autograd's backward of them.
//
// The RBF matrix [rows, bins] is a function of ONE scalar per row, and the layer behind it is narrow (bins = 40 / 80 ->
// F = 64): materialising it (107 MB at T = 676 200 rows) and pushing it through a GEMM makes a K = 40 product that the
// matrix-core kernels run at a quarter of the HBM roofline (236 us for 281 MB), plus a statistics pass, a normalise pass
// and - backward - a weight-gradient GEMM that reads it again.  Here every pass recomputes what it needs from the scalar:
// a lane owns a ROW, evaluates its `bins` exponentials once and accumulates the F pre-activations with F x bins FMAs whose
// weights are wave-uniform (scalar loads of W^T), so neither the RBF matrix nor the pre-activation [rows, F] ever goes to
// memory (SURVEY.md section 8(d) counts this layer that way: "RBF expansion recomputed in-kernel").  BatchNorm's global
// barrier makes it two passes forward (statistics; normalise + activate + write y) and three backward (BatchNorm-backward
// sums; the pre-activation gradient + bias-gradient slabs; the weight gradient).  Column statistics leave as pivot slabs
// (csrc/norm.hip), all reductions are fixed-order.
//
// Work per pass at T rows: 2 F bins = 5 120 flop per row = 3.5 GFLOP on the vector ALU (~45 us) against 173 MB of
// [rows, F] written or read (~35 us at 5 TB/s): the passes are about balanced - an MFMA formulation would make them purely
// HBM-bound, at several times the code.
#include "common.h"
#include "../../include/alignn_hip.h"

namespace {

constexpr int kWaves = 2;  // (LDS: one [64][F + 1] tile per wave; the weight gradient keeps [bins][F] per wave)
constexpr int kThreads = kWaves * ALIGNN_WAVE;
constexpr int kMaxBlocks = 1024;  // = statistic / reduction slabs

inline int blocks_for(int64_t rows) {
    int64_t b = (rows + kThreads - 1) / kThreads;
    if (b < 1) b = 1;
    if (b > kMaxBlocks) b = kMaxBlocks;
    return (int)b;
}

// pre[j] = b[j] + sum_k Wt[k][j] exp(-gamma (x - c_k)^2) for this lane's row (Wt = W^T, [bins][F]: wave-uniform loads)
template <int F>
__device__ __forceinline__ void preactivation(float x, const float* __restrict__ centers, float gamma,
                                              const float* __restrict__ Wt, const float* __restrict__ bias, int bins,
                                              float (&acc)[F]) {
#pragma unroll
    for (int j = 0; j < F; ++j) acc[j] = bias[j];
    for (int k = 0; k < bins; ++k) {
        const float t = x - centers[k];
        const float r = __expf(-gamma * t * t);
        const float* w = Wt + k * F;
#pragma unroll
        for (int j = 0; j < F; ++j) acc[j] = fmaf(w[j], r, acc[j]);
    }
}

// (n, p, S, SS) <- union with (nb, pb, Sb, SSb), about a's pivot (scalar form of norm.hip's pivot_merge)
__device__ __forceinline__ void pivot_merge1(float& na, float& pa, float& Sa, float& SSa, float nb, float pb, float Sb,
                                             float SSb) {
    if (nb == 0.0f) return;
    if (na == 0.0f) {
        na = nb, pa = pb, Sa = Sb, SSa = SSb;
        return;
    }
    const float d = pb - pa;
    SSa += SSb + d * (2.0f * Sb + nb * d);
    Sa += Sb + nb * d;
    na += nb;
}

// A wave's tile = 64 consecutive rows of a [rows][F] matrix = ONE contiguous run of 64 F floats: moved as F / 4 float4 per
// lane (whole 1 KiB segments per instruction, all loads in flight at once) to / from the LDS tile [64][F + 1].
template <int F>
__device__ __forceinline__ void tile_load(const float* __restrict__ G, int64_t r0, int valid, float* tl, int lane) {
    constexpr int LD = F + 1;
    float4 v[F / 4];
#pragma unroll
    for (int q = 0; q < F / 4; ++q) {
        const int e = (q * 64 + lane) * 4;  // flat element of the tile
        v[q] = (e / F) < valid ? f4_ld(G + r0 * F + e) : f4_zero();
    }
#pragma unroll
    for (int q = 0; q < F / 4; ++q) {
        const int e = (q * 64 + lane) * 4;
        float* o = tl + (e / F) * LD + (e % F);
        o[0] = v[q].x, o[1] = v[q].y, o[2] = v[q].z, o[3] = v[q].w;
    }
}
template <int F>
__device__ __forceinline__ float tile_store(float* __restrict__ Y, int64_t r0, int valid, const float* tl, int lane) {
    constexpr int LD = F + 1;
    float am = 0.0f;
#pragma unroll
    for (int q = 0; q < F / 4; ++q) {
        const int e = (q * 64 + lane) * 4;
        const float* o = tl + (e / F) * LD + (e % F);
        const float4 v = make_float4(o[0], o[1], o[2], o[3]);
        if ((e / F) < valid) {
            f4_st(Y + r0 * F + e, v);
            am = fmaxf(am, f4_absmax(v));
        }
    }
    return am;
}
// column sum over the tile's 64 rows for column `lane` (rows past `valid` hold zeros)
template <int F>
__device__ __forceinline__ float tile_colsum(const float* tl, int lane) {
    constexpr int LD = F + 1;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int i = 0; i < 64; i += 4) {
        s0 += tl[i * LD + lane];
        s1 += tl[(i + 1) * LD + lane];
        s2 += tl[(i + 2) * LD + lane];
        s3 += tl[(i + 3) * LD + lane];
    }
    return (s0 + s1) + (s2 + s3);
}

enum Mode { STATS = 0, APPLY = 1, BWD_REDUCE = 2, BWD_APPLY = 3 };

// One wave = 64 consecutive rows per step; LDS tile [64][F + 1] per wave turns "lane = row" into "lane = column" for the
// column reductions and into whole-row segments for coalesced global accesses.
template <int F, int MODE>
__global__ __launch_bounds__(kThreads) void rbf_mlp_kernel(
    const float* __restrict__ d, const float* __restrict__ centers, float gamma, const float* __restrict__ Wt,
    const float* __restrict__ bias, int bins, int64_t rows,
    const float* __restrict__ stat,  // [4][F] mean, rstd, scale, beta           (APPLY, BWD_*)
    const float* __restrict__ GY,    // [rows][F] gradient of y                  (BWD_*)
    const float* __restrict__ red,   // [2][F] sum gz, sum gz*xhat               (BWD_APPLY, training)
    int eval_mode, float inv_n,
    float* __restrict__ Y,        // APPLY: y [rows][F];  BWD_APPLY: g_pre [rows][F]
    float* __restrict__ partial,  // STATS: pivot slabs [grid][3][F] | counts[grid];  BWD_REDUCE: [grid][2][F];  BWD_APPLY: [grid][F]
    float* __restrict__ amax) {
    constexpr int LD = F + 1;
    __shared__ float tile[kWaves][ALIGNN_WAVE * LD];
    __shared__ float mrg[kWaves][4][F];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* tl = tile[wave];
    const int64_t n_batches = (rows + 63) / 64;
    // column state of lane j < F (wave-private)
    float cn = 0.0f, cp = 0.0f, cS = 0.0f, cSS = 0.0f;  // STATS: pivot sums;  BWD_REDUCE: cS = sum gz, cSS = sum gz*xhat;  BWD_APPLY: cS = sum g_pre
    float am = 0.0f;
    float mean_j = 0.f, rstd_j = 0.f;
    (void)mean_j, (void)rstd_j;
    for (int64_t b = (int64_t)blockIdx.x * kWaves + wave; b < n_batches; b += (int64_t)gridDim.x * kWaves) {
        const int64_t r0 = b * 64;
        const int valid = rows - r0 < 64 ? (int)(rows - r0) : 64;
        const bool row_ok = lane < valid;
        float acc[F];
        preactivation<F>(row_ok ? d[r0 + lane] : 0.0f, centers, gamma, Wt, bias, bins, acc);
        if (MODE == STATS) {
#pragma unroll
            for (int j = 0; j < F; ++j) tl[lane * LD + j] = acc[j];
            if (lane < F) {
                if (cn == 0.0f) cp = tl[lane];  // (row r0 is always valid)
#pragma unroll 16
                for (int i = 0; i < 64; ++i) {
                    const float dv = i < valid ? tl[i * LD + lane] - cp : 0.0f;
                    cS += dv;
                    cSS = fmaf(dv, dv, cSS);
                }
                cn += (float)valid;
            }
        } else if (MODE == APPLY) {
#pragma unroll
            for (int j = 0; j < F; ++j) {
                const float z = fmaf(acc[j] - stat[j], stat[2 * F + j], stat[3 * F + j]);
                tl[lane * LD + j] = silu_f(z);
            }
            am = fmaxf(am, tile_store<F>(Y, r0, valid, tl, lane));
        } else {
            // gy tile in (coalesced float4, all in flight), then lane = row again
            tile_load<F>(GY, r0, valid, tl, lane);
            float gz[F], xh[F];
#pragma unroll
            for (int j = 0; j < F; ++j) {
                const float xc = acc[j] - stat[j];
                const float z = fmaf(xc, stat[2 * F + j], stat[3 * F + j]);
                gz[j] = row_ok ? tl[lane * LD + j] * dsilu_f(z) : 0.0f;
                xh[j] = xc * stat[F + j];
            }
            if (MODE == BWD_REDUCE) {
#pragma unroll
                for (int j = 0; j < F; ++j) tl[lane * LD + j] = gz[j];
                if (lane < F) cS += tile_colsum<F>(tl, lane);
#pragma unroll
                for (int j = 0; j < F; ++j) tl[lane * LD + j] = gz[j] * xh[j];
                if (lane < F) cSS += tile_colsum<F>(tl, lane);
            } else {  // BWD_APPLY
#pragma unroll
                for (int j = 0; j < F; ++j) {
                    float o;
                    if (eval_mode)
                        o = gz[j] * stat[2 * F + j];
                    else
                        o = stat[2 * F + j] * (gz[j] - inv_n * (red[j] + xh[j] * red[F + j]));
                    tl[lane * LD + j] = row_ok ? o : 0.0f;
                }
                if (lane < F) cS += tile_colsum<F>(tl, lane);
                am = fmaxf(am, tile_store<F>(Y, r0, valid, tl, lane));
            }
        }
    }
    // ---- block-level merge of the four waves' column states, in wave order
    if (MODE == STATS || MODE == BWD_REDUCE || MODE == BWD_APPLY) {
        if (lane < F) {
            mrg[wave][0][lane] = cn;
            mrg[wave][1][lane] = cp;
            mrg[wave][2][lane] = cS;
            mrg[wave][3][lane] = cSS;
        }
        __syncthreads();
        if (wave == 0 && lane < F) {
            if (MODE == STATS) {
                float n = mrg[0][0][lane], p = mrg[0][1][lane], S = mrg[0][2][lane], SS = mrg[0][3][lane];
#pragma unroll
                for (int w = 1; w < kWaves; ++w) pivot_merge1(n, p, S, SS, mrg[w][0][lane], mrg[w][1][lane], mrg[w][2][lane], mrg[w][3][lane]);
                float* slab = partial + (size_t)blockIdx.x * 3 * F;
                slab[lane] = p;
                slab[F + lane] = S;
                slab[2 * F + lane] = SS;
                if (lane == 0) partial[(size_t)gridDim.x * 3 * F + blockIdx.x] = n;
            } else if (MODE == BWD_REDUCE) {
                float s0 = mrg[0][2][lane], s1 = mrg[0][3][lane];
#pragma unroll
                for (int w = 1; w < kWaves; ++w) s0 += mrg[w][2][lane], s1 += mrg[w][3][lane];
                partial[(size_t)blockIdx.x * 2 * F + lane] = s0;
                partial[(size_t)blockIdx.x * 2 * F + F + lane] = s1;
            } else {
                float s0 = mrg[0][2][lane];
#pragma unroll
                for (int w = 1; w < kWaves; ++w) s0 += mrg[w][2][lane];
                partial[(size_t)blockIdx.x * F + lane] = s0;
            }
        }
    }
    if (MODE == APPLY || MODE == BWD_APPLY) block_amax_commit(am, amax);
}

// Weight gradient dW[j][k] = sum_t G[t][j] rbf_k(d_t).  Per 64-row tile: G goes to LDS by float4 (coalesced, all in
// flight), every lane (= row) writes its `bins` rbf values to LDS as R[k][row]; then lane j walks the 64 rows:
// acc[k] += G[i][j] R[k][i] with R read as wave-uniform (broadcast) float4 over four rows.  Slabs [grid][F][bins].
template <int F, int BINS_MAX>
__global__ __launch_bounds__(kThreads) void rbf_wgrad_kernel(const float* __restrict__ d, const float* __restrict__ centers,
                                                             float gamma, const float* __restrict__ G, int bins, int64_t rows,
                                                             float* __restrict__ partial) {
    constexpr int LD = F + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tl = smem + (threadIdx.x >> 6) * (64 * LD + BINS_MAX * 64);  // per wave: G tile [64][F+1] | R [BINS_MAX][64]
    float* R = tl + 64 * LD;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float acc[BINS_MAX];
#pragma unroll
    for (int k = 0; k < BINS_MAX; ++k) acc[k] = 0.0f;
    const int64_t n_batches = (rows + 63) / 64;
    for (int64_t b = (int64_t)blockIdx.x * kWaves + wave; b < n_batches; b += (int64_t)gridDim.x * kWaves) {
        const int64_t r0 = b * 64;
        const int valid = rows - r0 < 64 ? (int)(rows - r0) : 64;
        tile_load<F>(G, r0, valid, tl, lane);  // (rows past the end: zeros, so their rbf values do not matter)
        const float x = lane < valid ? d[r0 + lane] : 0.0f;
        for (int k = 0; k < bins; ++k) {
            const float t = x - centers[k];
            R[k * 64 + lane] = __expf(-gamma * t * t);
        }
        if (lane < F) {
#pragma unroll 2
            for (int i = 0; i < 64; i += 4) {
                const float g0 = tl[i * LD + lane], g1 = tl[(i + 1) * LD + lane], g2 = tl[(i + 2) * LD + lane],
                            g3 = tl[(i + 3) * LD + lane];
#pragma unroll
                for (int k = 0; k < BINS_MAX; ++k) {
                    if (k < bins) {
                        const float4 r = *reinterpret_cast<const float4*>(R + k * 64 + i);  // same address in every lane
                        acc[k] = fmaf(g3, r.w, fmaf(g2, r.z, fmaf(g1, r.y, fmaf(g0, r.x, acc[k]))));
                    }
                }
            }
        }
    }
    // the waves' partial dW through LDS (reusing the tile space), summed in wave order
    __syncthreads();
    float* mrg = smem;  // [kWaves][BINS_MAX][F] <= the tiles' space (checked on the host side)
    if (lane < F) {
#pragma unroll
        for (int k = 0; k < BINS_MAX; ++k) mrg[(wave * BINS_MAX + k) * F + lane] = acc[k];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < F * bins; e += kThreads) {
        const int j = e / bins, k = e - j * bins;
        float sum = mrg[k * F + j];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) sum += mrg[(w * BINS_MAX + k) * F + j];
        partial[(size_t)blockIdx.x * F * bins + e] = sum;
    }
}

template <int F, int BINS_MAX>
constexpr size_t wgrad_lds() { return (size_t)kWaves * (64 * (F + 1) + BINS_MAX * 64) * sizeof(float); }

template <int F, int BINS_MAX>
int launch_wgrad(const float* d, const float* centers, float gamma, const float* G, int bins, int64_t rows, float* partial,
                 int grid, hipStream_t stream) {
    constexpr size_t lds = wgrad_lds<F, BINS_MAX>();
    static_assert(lds <= 160 * 1024, "LDS of the weight-gradient kernel");
    static_assert((size_t)kWaves * BINS_MAX * F * sizeof(float) <= lds, "the merge buffer reuses the tiles");
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)rbf_wgrad_kernel<F, BINS_MAX>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((rbf_wgrad_kernel<F, BINS_MAX>), dim3(grid), dim3(kThreads), lds, stream, d, centers, gamma, G, bins, rows,
                       partial);
    return 0;
}

inline bool shape_ok(int F, int bins) { return (F == 16 || F == 32 || F == 48 || F == 64) && bins > 0 && bins <= 128; }

#define RBF_DISPATCH_F(F_, CALL)           \
    switch (F_) {                          \
        case 16: { constexpr int FF = 16; CALL; } break; \
        case 32: { constexpr int FF = 32; CALL; } break; \
        case 48: { constexpr int FF = 48; CALL; } break; \
        default: { constexpr int FF = 64; CALL; } break; \
    }

}  // namespace

extern "C" {

int alignn_rbf_mlp_supported(int F, int bins) { return shape_ok(F, bins) ? 1 : 0; }
int alignn_rbf_mlp_slabs(int64_t rows) { return blocks_for(rows); }

int alignn_rbf_mlp_stats(const float* d, const float* centers, float gamma, const float* Wt, const float* bias, int64_t rows,
                         int bins, int F, float* partial, alignn_stream_t stream) {
    if (!shape_ok(F, bins) || rows < 0 || partial == nullptr) return (int)hipErrorInvalidValue;
    const int grid = blocks_for(rows);
    RBF_DISPATCH_F(F, hipLaunchKernelGGL((rbf_mlp_kernel<FF, STATS>), dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, d, centers,
                                         gamma, Wt, bias, bins, rows, nullptr, nullptr, nullptr, 0, 0.0f, nullptr, partial, nullptr));
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_rbf_mlp_fwd(const float* d, const float* centers, float gamma, const float* Wt, const float* bias, int64_t rows,
                       int bins, int F, const float* stat, float* Y, float* amax, alignn_stream_t stream) {
    if (!shape_ok(F, bins) || rows < 0 || stat == nullptr || Y == nullptr) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    const int grid = blocks_for(rows);
    RBF_DISPATCH_F(F, hipLaunchKernelGGL((rbf_mlp_kernel<FF, APPLY>), dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, d, centers,
                                         gamma, Wt, bias, bins, rows, stat, nullptr, nullptr, 0, 0.0f, Y, nullptr, amax));
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_rbf_mlp_bwd_reduce(const float* d, const float* centers, float gamma, const float* Wt, const float* bias,
                              int64_t rows, int bins, int F, const float* stat, const float* GY, float* partial,
                              alignn_stream_t stream) {
    if (!shape_ok(F, bins) || rows < 0 || stat == nullptr || GY == nullptr || partial == nullptr) return (int)hipErrorInvalidValue;
    const int grid = blocks_for(rows);
    RBF_DISPATCH_F(F, hipLaunchKernelGGL((rbf_mlp_kernel<FF, BWD_REDUCE>), dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, d,
                                         centers, gamma, Wt, bias, bins, rows, stat, GY, nullptr, 0, 0.0f, nullptr, partial, nullptr));
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_rbf_mlp_bwd_apply(const float* d, const float* centers, float gamma, const float* Wt, const float* bias,
                             int64_t rows, int bins, int F, const float* stat, const float* GY, const float* red,
                             int eval_mode, float* GPRE, float* gb_partial, float* amax, alignn_stream_t stream) {
    if (!shape_ok(F, bins) || rows < 0 || stat == nullptr || GY == nullptr || GPRE == nullptr || gb_partial == nullptr ||
        (!eval_mode && red == nullptr))
        return (int)hipErrorInvalidValue;
    const int grid = blocks_for(rows);
    const float inv_n = rows > 0 ? 1.0f / (float)rows : 0.0f;
    RBF_DISPATCH_F(F, hipLaunchKernelGGL((rbf_mlp_kernel<FF, BWD_APPLY>), dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, d,
                                         centers, gamma, Wt, bias, bins, rows, stat, GY, red, eval_mode, inv_n, GPRE, gb_partial, amax));
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_rbf_mlp_wgrad(const float* d, const float* centers, float gamma, const float* GPRE, int64_t rows, int bins, int F,
                         float* partial, alignn_stream_t stream) {
    if (!shape_ok(F, bins) || rows < 0 || GPRE == nullptr || partial == nullptr) return (int)hipErrorInvalidValue;
    const int grid = blocks_for(rows);
    int rc = 0;
    if (bins <= 40) {
        RBF_DISPATCH_F(F, rc = (launch_wgrad<FF, 40>(d, centers, gamma, GPRE, bins, rows, partial, grid, (hipStream_t)stream)));
    } else if (bins <= 80) {
        RBF_DISPATCH_F(F, rc = (launch_wgrad<FF, 80>(d, centers, gamma, GPRE, bins, rows, partial, grid, (hipStream_t)stream)));
    } else {
        RBF_DISPATCH_F(F, rc = (launch_wgrad<FF, 128>(d, centers, gamma, GPRE, bins, rows, partial, grid, (hipStream_t)stream)));
    }
    if (rc != 0) return rc;
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
