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
                for (int i = 0; i < valid; ++i) {
                    const float v = tl[i * LD + lane];
                    if (cn == 0.0f) cp = v;
                    const float dv = v - cp;
                    cS += dv;
                    cSS = fmaf(dv, dv, cSS);
                    cn += 1.0f;
                }
            }
        } else if (MODE == APPLY) {
#pragma unroll
            for (int j = 0; j < F; ++j) {
                const float z = fmaf(acc[j] - stat[j], stat[2 * F + j], stat[3 * F + j]);
                tl[lane * LD + j] = silu_f(z);
            }
            // rows of the tile as 256-byte-or-so segments: lane l writes column l of row i
            for (int i = 0; i < valid; ++i) {
                if (lane < F) {
                    const float v = tl[i * LD + lane];
                    Y[(r0 + i) * F + lane] = v;
                    am = fmaxf(am, fabsf(v));
                }
            }
        } else {
            // gy tile in (coalesced), then lane = row again
            for (int i = 0; i < valid; ++i)
                if (lane < F) tl[i * LD + lane] = GY[(r0 + i) * F + lane];
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
                if (lane < F)
                    for (int i = 0; i < valid; ++i) cS += tl[i * LD + lane];
#pragma unroll
                for (int j = 0; j < F; ++j) tl[lane * LD + j] = gz[j] * xh[j];
                if (lane < F)
                    for (int i = 0; i < valid; ++i) cSS += tl[i * LD + lane];
            } else {  // BWD_APPLY
#pragma unroll
                for (int j = 0; j < F; ++j) {
                    float o;
                    if (eval_mode)
                        o = gz[j] * stat[2 * F + j];
                    else
                        o = stat[2 * F + j] * (gz[j] - inv_n * (red[j] + xh[j] * red[F + j]));
                    tl[lane * LD + j] = o;
                }
                for (int i = 0; i < valid; ++i) {
                    if (lane < F) {
                        const float v = tl[i * LD + lane];
                        Y[(r0 + i) * F + lane] = v;
                        cS += v;
                        am = fmaxf(am, fabsf(v));
                    }
                }
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

// Weight gradient dW[j][k] = sum_t G[t][j] rbf_k(d_t): lane j owns row j of dW (bins accumulators); the rbf values of a
// row are computed by lanes k < bins (one exponential each) and broadcast lane by lane.  Slabs [grid][F][bins].
template <int F, int BINS_MAX>
__global__ __launch_bounds__(kThreads) void rbf_wgrad_kernel(const float* __restrict__ d, const float* __restrict__ centers,
                                                             float gamma, const float* __restrict__ G, int bins, int64_t rows,
                                                             float* __restrict__ partial) {
    __shared__ float mrg[kWaves][BINS_MAX][F];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float acc[BINS_MAX];
#pragma unroll
    for (int k = 0; k < BINS_MAX; ++k) acc[k] = 0.0f;
    const float ck0 = lane < bins ? centers[lane] : 0.0f;
    const float ck1 = (lane + 64) < bins ? centers[lane + 64] : 0.0f;  // bins <= 128: a second value per lane
    const int64_t n_batches = (rows + 63) / 64;
    for (int64_t b = (int64_t)blockIdx.x * kWaves + wave; b < n_batches; b += (int64_t)gridDim.x * kWaves) {
        const int64_t r0 = b * 64;
        const int valid = rows - r0 < 64 ? (int)(rows - r0) : 64;
        const float dl = lane < valid ? d[r0 + lane] : 0.0f;  // this wave's 64 distances, one per lane
        for (int i = 0; i < valid; ++i) {
            const float x = __shfl(dl, i, 64);
            const float t0 = x - ck0, t1 = x - ck1;
            const float r_lo = __expf(-gamma * t0 * t0), r_hi = __expf(-gamma * t1 * t1);
            const float g = lane < F ? G[(r0 + i) * F + lane] : 0.0f;
#pragma unroll
            for (int k = 0; k < BINS_MAX; ++k) {
                if (k < bins) {
                    const float rk = k < 64 ? __shfl(r_lo, k, 64) : __shfl(r_hi, k - 64, 64);
                    acc[k] = fmaf(g, rk, acc[k]);
                }
            }
        }
    }
    if (lane < F) {
#pragma unroll
        for (int k = 0; k < BINS_MAX; ++k) mrg[wave][k][lane] = acc[k];
    }
    __syncthreads();
    // slab [F][bins]: fixed-order sum of the four waves
    for (int e = threadIdx.x; e < F * bins; e += kThreads) {
        const int j = e / bins, k = e - j * bins;
        float s = mrg[0][k][j];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) s += mrg[w][k][j];
        partial[(size_t)blockIdx.x * F * bins + e] = s;
    }
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
    if (bins <= 40) {
        RBF_DISPATCH_F(F, hipLaunchKernelGGL((rbf_wgrad_kernel<FF, 40>), dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, d, centers,
                                             gamma, GPRE, bins, rows, partial));
    } else if (bins <= 80) {
        RBF_DISPATCH_F(F, hipLaunchKernelGGL((rbf_wgrad_kernel<FF, 80>), dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, d, centers,
                                             gamma, GPRE, bins, rows, partial));
    } else {
        RBF_DISPATCH_F(F, hipLaunchKernelGGL((rbf_wgrad_kernel<FF, 128>), dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, d,
                                             centers, gamma, GPRE, bins, rows, partial));
    }
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
