// Column statistics, BatchNorm1d(+SiLU, +residual) forward/backward for [rows, F] fp32 matrices.
// HBM-bound streaming kernels: float4 per lane, fixed-order two-level reductions (no atomics).
//
// Reference semantics: torch.nn.BatchNorm1d in training mode followed by F.silu and the residual
// add, alignn/models/alignn.py:122-127 (conv) and :175-179 (MLPLayer).
#include "common.h"
#include "../../include/alignn_hip.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxSlabs = 1024;

// one slab (= one workgroup) per 32 rows, at most kMaxSlabs: small matrices (3 840 atom rows) still spread over
// ~120 workgroups instead of crawling through 256 rows each on 15 CUs
__host__ __device__ inline int slabs_for(int64_t rows) {
    int64_t s = (rows + 31) / 32;
    if (s < 1) s = 1;
    if (s > kMaxSlabs) s = kMaxSlabs;
    return (int)s;
}

// ---------------------------------------------------------------------------------------------
// Generic column reduction: every thread owns one feature quad q = t % Q and walks rows
// rl, rl+RP, ...  (RP = 256 / Q rows in flight per block); the RP row-lanes are then summed
// through LDS in a fixed order.  Produces two float4 sums per quad (slab layout [2][F]).
// ---------------------------------------------------------------------------------------------
template <class Fn>
__global__ __launch_bounds__(kThreads) void col_reduce_kernel(Fn fn, int64_t rows, int F, int slabs,
                                                              float* __restrict__ partial) {
    const int Q = F >> 2;
    const int RP = kThreads / Q;
    const int t = threadIdx.x;
    const int q = t % Q;
    const int rl = t / Q;
    // slab b owns the row groups b, b + slabs, b + 2 slabs, ... (RP rows each): all workgroups walk the matrix
    // together as one moving window (5.4 TB/s at 1024 workgroups in tools/stream_bench.hip) instead of each crawling
    // through its own contiguous chunk (4.4 TB/s).  The row -> slab map is static, so sums stay bit-reproducible.
    const int64_t stride = (int64_t)slabs * RP;
    float4 a0 = f4_zero(), a1 = f4_zero();
    if (rl < RP) {
        // four independent accumulator pairs: four row loads in flight per thread (narrow matrices - F = 64 at T rows -
        // have few bytes per row group and were latency-bound with two: 115 us for 173 MB)
        float4 b0 = f4_zero(), b1 = f4_zero(), c0 = f4_zero(), c1 = f4_zero(), d0 = f4_zero(), d1 = f4_zero();
        int64_t r = (int64_t)blockIdx.x * RP + rl;
        for (; r + 3 * stride < rows; r += 4 * stride) {
            fn(r, q, a0, a1);
            fn(r + stride, q, b0, b1);
            fn(r + 2 * stride, q, c0, c1);
            fn(r + 3 * stride, q, d0, d1);
        }
        for (; r < rows; r += stride) fn(r, q, a0, a1);
        a0 = f4_add(f4_add(a0, b0), f4_add(c0, d0));
        a1 = f4_add(f4_add(a1, b1), f4_add(c1, d1));
    }
    __shared__ float4 sh[2][kThreads];
    sh[0][t] = a0;
    sh[1][t] = a1;
    __syncthreads();
    if (rl == 0) {
        for (int k = 1; k < RP; ++k) {
            a0 = f4_add(a0, sh[0][k * Q + q]);
            a1 = f4_add(a1, sh[1][k * Q + q]);
        }
        float* out = partial + (size_t)blockIdx.x * 2 * F;
        f4_st(out + q * 4, a0);
        f4_st(out + F + q * 4, a1);
    }
}

template <bool STREAM>
struct StatsFn {
    const float* X;
    int64_t ld;
    __device__ __forceinline__ void operator()(int64_t r, int q, float4& a0, float4& a1) const {
        float4 v = f4_lds<STREAM>(X + r * ld + q * 4);
        a0 = f4_add(a0, v);
        a1 = f4_fma(v, v, a1);
    }
};

template <bool STREAM>
struct BwdReduceFn {
    const float* GY;
    int64_t ldgy;
    const float* X;
    int64_t ldx;
    const float* stat;  // [4][F]: mean, rstd, scale=gamma*rstd, beta
    int F;
    __device__ __forceinline__ void operator()(int64_t r, int q, float4& a0, float4& a1) const {
        float4 gy = f4_lds<STREAM>(GY + r * ldgy + q * 4);
        float4 x = f4_lds<STREAM>(X + r * ldx + q * 4);
        float4 mean = f4_ld(stat + q * 4), rstd = f4_ld(stat + F + q * 4);
        float4 sc = f4_ld(stat + 2 * F + q * 4), be = f4_ld(stat + 3 * F + q * 4);
        float4 xc = f4_sub(x, mean);
        float4 z = f4_fma(xc, sc, be);
        float4 gz = make_float4(gy.x * dsilu_f(z.x), gy.y * dsilu_f(z.y), gy.z * dsilu_f(z.z), gy.w * dsilu_f(z.w));
        float4 xh = f4_mul(xc, rstd);
        a0 = f4_add(a0, gz);
        a1 = f4_fma(gz, xh, a1);
    }
};

// ---------------------------------------------------------------------------------------------
// Slab reductions: blocks of 16 columns x 64 slab-lanes.  Lane y sums slabs y, y+64, ... in double,
// then the 64 lanes are combined through LDS in a fixed order (bit-reproducible, ~slabs/64 steps).
// ---------------------------------------------------------------------------------------------
// 4 columns x 64 slab-lanes = 256 threads: F/4 workgroups.  (Round 2 ran 16 x 64 = 1 024 threads per workgroup: inside a
// replayed step such a block needs sixteen free wave slots on ONE CU at once, and beside a T-row kernel's resident
// workgroups it waited for them - rocprofv3 showed 5 us slab sums taking 290 us, profiles/r03_default_timeline.txt.)
constexpr int kRedCols = 4, kRedLanes = 64;

__device__ __forceinline__ double lane_tree_sum(double v, double (*sh)[kRedCols]) {
    sh[threadIdx.y][threadIdx.x] = v;
    __syncthreads();
    double out = 0.0;
    if (threadIdx.y == 0) {
#pragma unroll 8
        for (int k = 0; k < kRedLanes; ++k) out += sh[k][threadIdx.x];
    }
    __syncthreads();
    return out;
}

__global__ __launch_bounds__(kRedCols* kRedLanes) void bn_finalize_kernel(
    const float* __restrict__ partial, int slabs, int64_t rows, int F, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float momentum, float* __restrict__ running_mean,
    float* __restrict__ running_var, float* __restrict__ stat) {
    __shared__ double sh[kRedLanes][kRedCols];
    const int f = blockIdx.x * kRedCols + threadIdx.x;
    const bool ok = f < F;
    double s = 0.0, ss = 0.0;
    if (ok) {
        // (eight slabs' loads in flight before the first add: the slabs were just written by other XCDs, every load is a
        // last-level-cache round trip and the compiler otherwise chains them behind the float64 adds; same order of adds)
        int k = threadIdx.y;
        for (; k + 7 * kRedLanes < slabs; k += 8 * kRedLanes) {
            float a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[u] = partial[(size_t)(k + u * kRedLanes) * 2 * F + f];
                b[u] = partial[(size_t)(k + u * kRedLanes) * 2 * F + F + f];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)a[u], ss += (double)b[u];
        }
        for (; k < slabs; k += kRedLanes) {
            s += (double)partial[(size_t)k * 2 * F + f];
            ss += (double)partial[(size_t)k * 2 * F + F + f];
        }
    }
    s = lane_tree_sum(s, sh);
    ss = lane_tree_sum(ss, sh);
    if (!ok || threadIdx.y != 0) return;
    float mean, var;
    if (slabs > 0) {
        double n = (double)rows;
        double m = s / n;
        double v = ss / n - m * m;
        if (v < 0.0) v = 0.0;
        mean = (float)m;
        var = (float)v;
        if (running_mean != nullptr) {
            double unbiased = rows > 1 ? v * n / (n - 1.0) : v;
            running_mean[f] = (1.0f - momentum) * running_mean[f] + momentum * mean;
            running_var[f] = (1.0f - momentum) * running_var[f] + momentum * (float)unbiased;
        }
    } else {
        mean = running_mean[f];
        var = running_var[f];
    }
    float rstd = 1.0f / sqrtf(var + eps);
    float g = gamma ? gamma[f] : 1.0f, b = beta ? beta[f] : 0.0f;
    stat[f] = mean;
    stat[F + f] = rstd;
    stat[2 * F + f] = g * rstd;
    stat[3 * F + f] = b;
}

__global__ __launch_bounds__(kRedCols* kRedLanes) void slab_sum_kernel(const float* __restrict__ partial, int slabs,
                                                                        int width, int stride,
                                                                        float* __restrict__ out) {
    __shared__ double sh[kRedLanes][kRedCols];
    const int f = blockIdx.x * kRedCols + threadIdx.x;
    double s = 0.0;
    if (f < width) {
        int k = threadIdx.y;
        for (; k + 7 * kRedLanes < slabs; k += 8 * kRedLanes) {  // (batched loads, same order of adds: see bn_finalize_kernel)
            float a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = partial[(size_t)(k + u * kRedLanes) * stride + f];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)a[u];
        }
        for (; k < slabs; k += kRedLanes) s += (double)partial[(size_t)k * stride + f];
    }
    s = lane_tree_sum(s, sh);
    if (f < width && threadIdx.y == 0) out[f] = (float)s;
}

// Fold many slabs into kFold: out[g][f] = sum over slabs k = g, g + kFold, ... of partial[k][f] (double accumulation, fixed
// order).  The projection epilogues emit one slab per ROW TILE (5 283 at T rows); the column-block finalisers above walk
// slabs/64 steps per lane with 64-byte accesses - 60 us for that many -, this pre-pass reads whole 256-byte row
// segments with kFold x width/64 workgroups (5 us) and leaves them 64 slabs.
constexpr int kFold = 64;
__global__ __launch_bounds__(256) void slab_fold_kernel(const float* __restrict__ partial, int slabs, int width,
                                                        float* __restrict__ out) {
    __shared__ double sh[4][64];
    const int f = blockIdx.x * 64 + threadIdx.x, g = blockIdx.y, lane = threadIdx.y;
    double s = 0.0;
    if (f < width)
        for (int k = g + lane * kFold; k < slabs; k += 4 * kFold) s += (double)partial[(size_t)k * width + f];
    sh[lane][threadIdx.x] = s;
    __syncthreads();
    if (lane == 0 && f < width) out[(size_t)g * width + f] = (float)(((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + sh[2][threadIdx.x]) + sh[3][threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------
// Well-conditioned BatchNorm statistics ("pivot slabs").  sum x / sum x^2 slabs lose (mean / std)^2 digits when the
// variance is taken as E[x^2] - mean^2.  Here every thread sums d = x - p and d^2 about a pivot p = the first value it
// sees; partial results are merged by RE-CENTRING onto one of the two pivots (p_b - p_a is exact for nearby floats, and all
// sums stay of the size of the spread, not of the mean - a float32 mean could not even hold the sub-ulp part); a slab is
// (pivot, S, SS) with its row count, and the slabs are re-centred onto slab 0's pivot and summed in float64.  Fixed order
// everywhere.  Layout: partial[slabs][3][F] (pivot, S = sum (x - pivot), SS = sum (x - pivot)^2) | counts[slabs] (float).
// ---------------------------------------------------------------------------------------------
// (n, p, S, SS) of set a  <-  union with set b, expressed about a's pivot; empty sets pass through
__device__ __forceinline__ void pivot_merge(float& na, float4& pa, float4& Sa, float4& SSa, float nb, float4 pb, float4 Sb,
                                            float4 SSb) {
    if (nb == 0.0f) return;
    if (na == 0.0f) {
        na = nb, pa = pb, Sa = Sb, SSa = SSb;
        return;
    }
    const float4 d = f4_sub(pb, pa);
    // sum (x - pa)^2 over b = SSb + 2 d Sb + nb d^2 ;  sum (x - pa) over b = Sb + nb d
    SSa = make_float4(SSa.x + SSb.x + d.x * (2.0f * Sb.x + nb * d.x), SSa.y + SSb.y + d.y * (2.0f * Sb.y + nb * d.y),
                      SSa.z + SSb.z + d.z * (2.0f * Sb.z + nb * d.z), SSa.w + SSb.w + d.w * (2.0f * Sb.w + nb * d.w));
    Sa = make_float4(Sa.x + Sb.x + nb * d.x, Sa.y + Sb.y + nb * d.y, Sa.z + Sb.z + nb * d.z, Sa.w + Sb.w + nb * d.w);
    na += nb;
}

template <bool STREAM>
__global__ __launch_bounds__(kThreads) void col_stats_welford_kernel(const float* __restrict__ X, int64_t ldx, int64_t rows,
                                                                     int F, int slabs, float* __restrict__ partial) {
    const int Q = F >> 2;
    const int RP = kThreads / Q;
    const int t = threadIdx.x;
    const int q = t % Q;
    const int rl = t / Q;
    const int64_t stride = (int64_t)slabs * RP;  // same row -> slab map as col_reduce_kernel
    float n = 0.0f;
    float4 p = f4_zero(), S = f4_zero(), SS = f4_zero();
    if (rl < RP) {
        int64_t r = (int64_t)blockIdx.x * RP + rl;
        if (r < rows) p = f4_lds<STREAM>(X + r * ldx + q * 4);
        for (; r < rows; r += stride) {
            const float4 d = f4_sub(f4_lds<STREAM>(X + r * ldx + q * 4), p);
            S = f4_add(S, d);
            SS = f4_fma(d, d, SS);
            n += 1.0f;
        }
    }
    __shared__ float4 sh[3][kThreads];
    __shared__ float shn[kThreads];
    sh[0][t] = p;
    sh[1][t] = S;
    sh[2][t] = SS;
    shn[t] = n;
    __syncthreads();
    if (rl == 0) {
        for (int k = 1; k < RP; ++k) pivot_merge(n, p, S, SS, shn[k * Q + q], sh[0][k * Q + q], sh[1][k * Q + q], sh[2][k * Q + q]);
        float* out = partial + (size_t)blockIdx.x * 3 * F;
        f4_st(out + q * 4, p);
        f4_st(out + F + q * 4, S);
        f4_st(out + 2 * F + q * 4, SS);
        if (q == 0) partial[(size_t)slabs * 3 * F + blockIdx.x] = n;  // every column of the slab saw the same rows
    }
}

// finalise pivot slabs in float64: re-centre every slab onto slab 0's pivot P, then mean = P + S / n, var = SS / n - (S / n)^2
// (S / n is of the size of the spread: the subtraction costs nothing).  kWfCols columns x kWfLanes slab-lanes per
// workgroup, two-stage fixed-order tree (64 -> 8 -> 1).
constexpr int kWfCols = 4, kWfLanes = 64;  // (256 threads: see kRedCols)
__global__ __launch_bounds__(kWfCols* kWfLanes) void bn_finalize_welford_kernel(
    const float* __restrict__ partial, const float* __restrict__ counts, int slabs, int64_t rows, int F,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
    float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ stat) {
    __shared__ double sh[2][kWfLanes][kWfCols];
    __shared__ double sh2[2][8][kWfCols];
    const int f = blockIdx.x * kWfCols + threadIdx.x;
    const bool ok = f < F;
    double s = 0.0, ss = 0.0, P = 0.0;
    if (ok) {
        // the pivot of the first non-empty slab (the same one for every slab-lane: fixed)
        int k0 = 0;
        while (k0 < slabs - 1 && counts[k0] == 0.0f) ++k0;
        P = (double)partial[(size_t)k0 * 3 * F + f];
        auto merge = [&](float nkf, float pk, float Skf, float SSk) {
            const double nk = (double)nkf;
            if (nk > 0.0) {
                const double d = (double)pk - P, Sk = (double)Skf;
                s += Sk + nk * d;
                ss += (double)SSk + d * (2.0 * Sk + nk * d);
            }
        };
        int k = threadIdx.y;
        for (; k + 3 * kWfLanes < slabs; k += 4 * kWfLanes) {  // (four slabs' loads in flight; same order of merges)
            float nk[4], pk[4], Sk[4], SSk[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* sl = partial + (size_t)(k + u * kWfLanes) * 3 * F;
                nk[u] = counts[k + u * kWfLanes], pk[u] = sl[f], Sk[u] = sl[F + f], SSk[u] = sl[2 * F + f];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) merge(nk[u], pk[u], Sk[u], SSk[u]);
        }
        for (; k < slabs; k += kWfLanes) {
            const float* sl = partial + (size_t)k * 3 * F;
            merge(counts[k], sl[f], sl[F + f], sl[2 * F + f]);
        }
    }
    sh[0][threadIdx.y][threadIdx.x] = s;
    sh[1][threadIdx.y][threadIdx.x] = ss;
    __syncthreads();
    if (threadIdx.y < 8) {
        double a = 0.0, b2 = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a += sh[0][threadIdx.y * 8 + k][threadIdx.x];
            b2 += sh[1][threadIdx.y * 8 + k][threadIdx.x];
        }
        sh2[0][threadIdx.y][threadIdx.x] = a;
        sh2[1][threadIdx.y][threadIdx.x] = b2;
    }
    __syncthreads();
    if (!ok || threadIdx.y != 0) return;
    s = ss = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        s += sh2[0][k][threadIdx.x];
        ss += sh2[1][k][threadIdx.x];
    }
    const double n = (double)rows;
    const double ms = s / n;
    double v = ss / n - ms * ms;
    if (v < 0.0) v = 0.0;
    const float mean = (float)(P + ms), var = (float)v;
    if (running_mean != nullptr) {
        const double unbiased = rows > 1 ? v * n / (n - 1.0) : v;
        running_mean[f] = (1.0f - momentum) * running_mean[f] + momentum * mean;
        running_var[f] = (1.0f - momentum) * running_var[f] + momentum * (float)unbiased;
    }
    const float rstd = 1.0f / sqrtf(var + eps);
    const float g = gamma ? gamma[f] : 1.0f, b = beta ? beta[f] : 0.0f;
    stat[f] = mean;
    stat[F + f] = rstd;
    stat[2 * F + f] = g * rstd;
    stat[3 * F + f] = b;
}

// Y = R + silu((X-mean)*scale + beta)
template <bool HAS_RES, bool STREAM>
__global__ __launch_bounds__(kThreads) void bn_silu_fwd_kernel(const float* __restrict__ X, int64_t ldx,
                                                               const float* __restrict__ R, int64_t ldr,
                                                               const float* __restrict__ stat,
                                                               float* __restrict__ Y, int64_t ldy,
                                                               int64_t rows, int F, float* __restrict__ amax) {
    const int Q = F >> 2;
    const RowQuad rq(Q);
    const int64_t total = rows * Q;
    float am = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        int64_t r;
        int q;
        rq.split(i, total, r, q);
        float4 x = f4_lds<STREAM>(X + r * ldx + q * 4);
        float4 mean = f4_ld(stat + q * 4);
        float4 sc = f4_ld(stat + 2 * F + q * 4), be = f4_ld(stat + 3 * F + q * 4);
        float4 z = f4_fma(f4_sub(x, mean), sc, be);
        float4 o = make_float4(silu_f(z.x), silu_f(z.y), silu_f(z.z), silu_f(z.w));
        if (HAS_RES) o = f4_add(o, f4_lds<STREAM>(R + r * ldr + q * 4));
        f4_sts<STREAM>(Y + r * ldy + q * 4, o);
        am = fmaxf(am, f4_absmax(o));
    }
    block_amax_commit(am, amax);
}

// column-walking form of bn_silu_fwd_kernel for tall matrices: a thread keeps its quad's three constant rows in registers
// and walks the rows of its slab (the element-wise form reloads them for every element)
template <bool HAS_RES, bool STREAM>
__global__ __launch_bounds__(kThreads) void bn_silu_fwd_cols_kernel(const float* __restrict__ X, int64_t ldx,
                                                                    const float* __restrict__ R, int64_t ldr,
                                                                    const float* __restrict__ stat, float* __restrict__ Y,
                                                                    int64_t ldy, int64_t rows, int F, int slabs,
                                                                    float* __restrict__ amax) {
    const int Q = F >> 2;
    const int RP = kThreads / Q;
    const int q = threadIdx.x % Q, rl = threadIdx.x / Q;
    const int64_t stride = (int64_t)slabs * RP;
    float am = 0.0f;
    if (rl < RP) {
        const float4 mean = f4_ld(stat + q * 4), sc = f4_ld(stat + 2 * F + q * 4), be = f4_ld(stat + 3 * F + q * 4);
        for (int64_t r = (int64_t)blockIdx.x * RP + rl; r < rows; r += stride) {
            const float4 x = f4_lds<STREAM>(X + r * ldx + q * 4);
            const float4 z = f4_fma(f4_sub(x, mean), sc, be);
            float4 o = make_float4(silu_f(z.x), silu_f(z.y), silu_f(z.z), silu_f(z.w));
            if (HAS_RES) o = f4_add(o, f4_lds<STREAM>(R + r * ldr + q * 4));
            f4_sts<STREAM>(Y + r * ldy + q * 4, o);
            am = fmaxf(am, f4_absmax(o));
        }
    }
    block_amax_commit(am, amax);
}

// NODE: GX is the gradient of a convolution's x_pre = Ux + S1 / (S0 + eps) - the quotient's adjoints
// gS1 = GX / (S0 + eps), gS0 = -gS1 * h (h = S1 / (S0 + eps), saved by the forward) go out in the same pass
// (alignn_egc_node_bwd's arithmetic on the values this pass has in registers: same bits, one launch and one read of GX less)
template <bool STREAM, bool NODE = false>
__global__ __launch_bounds__(kThreads) void bn_silu_bwd_apply_kernel(
    const float* __restrict__ GY, int64_t ldgy, const float* __restrict__ X, int64_t ldx,
    const float* __restrict__ stat, const float* __restrict__ gamma, const float* __restrict__ red, int eval_mode,
    float* __restrict__ GX, int64_t ldgx, int64_t rows, int F, float* __restrict__ amax,
    const float* __restrict__ S0, const float* __restrict__ HH, float* __restrict__ GS1, float* __restrict__ GS0) {
    const int Q = F >> 2;
    const RowQuad rq(Q);
    const int64_t total = rows * Q;
    const float inv_n = 1.0f / (float)rows;
    float am = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        int64_t r;
        int q;
        rq.split(i, total, r, q);
        float4 gy = f4_lds<STREAM>(GY + r * ldgy + q * 4);
        float4 x = f4_lds<STREAM>(X + r * ldx + q * 4);
        float4 mean = f4_ld(stat + q * 4), rstd = f4_ld(stat + F + q * 4);
        float4 sc = f4_ld(stat + 2 * F + q * 4), be = f4_ld(stat + 3 * F + q * 4);
        float4 xc = f4_sub(x, mean);
        float4 z = f4_fma(xc, sc, be);
        float4 gz = make_float4(gy.x * dsilu_f(z.x), gy.y * dsilu_f(z.y), gy.z * dsilu_f(z.z), gy.w * dsilu_f(z.w));
        float4 o;
        if (eval_mode) {
            o = f4_mul(gz, sc);
        } else {
            float4 xh = f4_mul(xc, rstd);
            float4 c0 = f4_ld(red + q * 4), c1 = f4_ld(red + F + q * 4);
            // gamma*rstd*(gz - c0/n - xh*c1/n) ; scale = gamma*rstd
            o.x = sc.x * (gz.x - inv_n * (c0.x + xh.x * c1.x));
            o.y = sc.y * (gz.y - inv_n * (c0.y + xh.y * c1.y));
            o.z = sc.z * (gz.z - inv_n * (c0.z + xh.z * c1.z));
            o.w = sc.w * (gz.w - inv_n * (c0.w + xh.w * c1.w));
        }
        f4_sts<STREAM>(GX + r * ldgx + q * 4, o);
        am = fmaxf(am, f4_absmax(o));
        if constexpr (NODE) {
            const float4 s0 = f4_ld(S0 + r * F + q * 4), h = f4_ld(HH + r * F + q * 4);
            float4 g1;
            g1.x = o.x / (s0.x + ALIGNN_EPS_GATE);
            g1.y = o.y / (s0.y + ALIGNN_EPS_GATE);
            g1.z = o.z / (s0.z + ALIGNN_EPS_GATE);
            g1.w = o.w / (s0.w + ALIGNN_EPS_GATE);
            f4_st(GS1 + r * F + q * 4, g1);
            f4_st(GS0 + r * F + q * 4, make_float4(-g1.x * h.x, -g1.y * h.y, -g1.z * h.z, -g1.w * h.w));
        }
    }
    block_amax_commit(am, amax);
}

// The same pass in the column-walking form of col_reduce_kernel: a thread owns ONE feature quad (its six constant rows are
// loaded once, not per element) and walks the rows of its slab, so it can also sum what it writes - the column sums of GX
// are the bias gradient of the Linear in front of the norm (MLPLayer: alignn/models/alignn.py:170-184), which otherwise
// costs another full pass over GX (alignn_col_sum: 0.11 ms at T x 256).  partial: [gridDim.x][F].
template <bool STREAM>
__global__ __launch_bounds__(kThreads) void bn_silu_bwd_apply_sum_kernel(
    const float* __restrict__ GY, int64_t ldgy, const float* __restrict__ X, int64_t ldx,
    const float* __restrict__ stat, const float* __restrict__ red, int eval_mode, float* __restrict__ GX, int64_t ldgx,
    int64_t rows, int F, int slabs, float* __restrict__ amax, float* __restrict__ partial) {
    const int Q = F >> 2;
    const int RP = kThreads / Q;
    const int t = threadIdx.x;
    const int q = t % Q;
    const int rl = t / Q;
    const float inv_n = 1.0f / (float)rows;
    const int64_t stride = (int64_t)slabs * RP;
    float4 acc = f4_zero();
    float am = 0.0f;
    if (rl < RP) {
        const float4 mean = f4_ld(stat + q * 4), rstd = f4_ld(stat + F + q * 4);
        const float4 sc = f4_ld(stat + 2 * F + q * 4), be = f4_ld(stat + 3 * F + q * 4);
        float4 c0 = f4_zero(), c1 = f4_zero();
        if (!eval_mode) c0 = f4_ld(red + q * 4), c1 = f4_ld(red + F + q * 4);
        for (int64_t r = (int64_t)blockIdx.x * RP + rl; r < rows; r += stride) {
            const float4 gy = f4_lds<STREAM>(GY + r * ldgy + q * 4);
            const float4 x = f4_lds<STREAM>(X + r * ldx + q * 4);
            const float4 xc = f4_sub(x, mean);
            const float4 z = f4_fma(xc, sc, be);
            const float4 gz = make_float4(gy.x * dsilu_f(z.x), gy.y * dsilu_f(z.y), gy.z * dsilu_f(z.z), gy.w * dsilu_f(z.w));
            float4 o;
            if (eval_mode) {
                o = f4_mul(gz, sc);
            } else {
                const float4 xh = f4_mul(xc, rstd);
                o.x = sc.x * (gz.x - inv_n * (c0.x + xh.x * c1.x));
                o.y = sc.y * (gz.y - inv_n * (c0.y + xh.y * c1.y));
                o.z = sc.z * (gz.z - inv_n * (c0.z + xh.z * c1.z));
                o.w = sc.w * (gz.w - inv_n * (c0.w + xh.w * c1.w));
            }
            f4_sts<STREAM>(GX + r * ldgx + q * 4, o);
            acc = f4_add(acc, o);
            am = fmaxf(am, f4_absmax(o));
        }
    }
    __shared__ float4 sh[kThreads];
    sh[t] = acc;
    __syncthreads();
    if (rl == 0) {
        for (int k = 1; k < RP; ++k) acc = f4_add(acc, sh[k * Q + q]);
        f4_st(partial + (size_t)blockIdx.x * F + q * 4, acc);
    }
    block_amax_commit(am, amax);
}

// ---------------------------------------------------------------------------------------------
// LayerNorm(+SiLU, +residual) - the ALIGNNAtomWise flavour (alignn/models/alignn_atomwise.py:151,155 and
// alignn/models/utils.py:277-292): statistics per ROW over the F features, eps 1e-5, affine gamma/beta.
// One wavefront per row; lane l owns features [4l, 4l+4) of each 256-feature chunk (NC chunks, F <= 1024);
// two-pass variance in registers; row sums by __shfl_xor butterflies.  No global barrier is needed, so the
// forward is one pass and the backward is one pass (+ fixed-order slabs for dgamma/dbeta).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float f4_hsum(float4 a) { return (a.x + a.y) + (a.z + a.w); }

template <int NC, bool HAS_RES>
__global__ __launch_bounds__(kThreads) void ln_silu_fwd_kernel(const float* __restrict__ X, int64_t ldx,
                                                               const float* __restrict__ R, int64_t ldr,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps,
                                                               float* __restrict__ Y, int64_t ldy,
                                                               float* __restrict__ stats, int64_t rows, int F,
                                                               float* __restrict__ amax) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    const int64_t stride = (int64_t)gridDim.x * (kThreads / 64);
    const float inv_f = 1.0f / (float)F;
    float am = 0.0f;
    for (int64_t r = wave0; r < rows; r += stride) {
        float4 x[NC];
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int f = c * 256 + 4 * lane;
            x[c] = f < F ? f4_ld(X + r * ldx + f) : f4_zero();
            s += f4_hsum(x[c]);
        }
        const float mean = wave_sum(s) * inv_f;
        float v = 0.0f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int f = c * 256 + 4 * lane;
            if (f < F) {
                float4 d = make_float4(x[c].x - mean, x[c].y - mean, x[c].z - mean, x[c].w - mean);
                v += f4_hsum(f4_mul(d, d));
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(v) * inv_f + eps);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int f = c * 256 + 4 * lane;
            if (f < F) {
                float4 g = f4_ld(gamma + f), b = f4_ld(beta + f);
                float4 z;
                z.x = (x[c].x - mean) * rstd * g.x + b.x;
                z.y = (x[c].y - mean) * rstd * g.y + b.y;
                z.z = (x[c].z - mean) * rstd * g.z + b.z;
                z.w = (x[c].w - mean) * rstd * g.w + b.w;
                float4 o = make_float4(silu_f(z.x), silu_f(z.y), silu_f(z.z), silu_f(z.w));
                if (HAS_RES) o = f4_add(o, f4_ld(R + r * ldr + f));
                f4_st(Y + r * ldy + f, o);
                am = fmaxf(am, f4_absmax(o));
            }
        }
        if (stats && lane == 0) {
            stats[2 * r] = mean;
            stats[2 * r + 1] = rstd;
        }
    }
    block_amax_commit(am, amax);
}

// NOTE: this file is compiled with -fno-slp-vectorize (alignn_amd/build.py): with packed-fp32 instructions hipcc 7.2's code for
// the kernel below was not bit-reproducible beside kernels of another stream.
// NODE: the rows are the node pre-activations xpre = Ux + h of an edge-gated convolution, h = S1 / (S0 + eps): the adjoints of the
// two segment sums leave with the gradient (alignn_egc_node_bwd's arithmetic: GS1 = g / (S0 + eps), GS0 = -GS1 * h) - one launch
// instead of two on the bond-row chain of every LayerNorm-flavoured convolution's reverse.
template <int NC, bool NODE>
__global__ __launch_bounds__(kThreads) void ln_silu_bwd_kernel(const float* __restrict__ GY, int64_t ldgy,
                                                               const float* __restrict__ X, int64_t ldx,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               const float* __restrict__ stats,
                                                               float* __restrict__ GX, int64_t ldgx,
                                                               float* __restrict__ partial, int64_t rows, int F,
                                                               float* __restrict__ amax, const float* __restrict__ S0,
                                                               const float* __restrict__ HH, float* __restrict__ GS1,
                                                               float* __restrict__ GS0) {
    __shared__ float4 sh[2][kThreads / 64][64];
    float am = 0.0f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wave0 = (int64_t)blockIdx.x * (kThreads / 64) + wave;
    const int64_t stride = (int64_t)gridDim.x * (kThreads / 64);
    const float inv_f = 1.0f / (float)F;
    float4 dg[NC], db[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) dg[c] = db[c] = f4_zero();
    for (int64_t r = wave0; r < rows; r += stride) {
        const float mean = stats[2 * r], rstd = stats[2 * r + 1];
        float4 xh[NC], gh[NC];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int f = c * 256 + 4 * lane;
            xh[c] = gh[c] = f4_zero();
            if (f < F) {
                float4 x = f4_ld(X + r * ldx + f), gy = f4_ld(GY + r * ldgy + f);
                float4 g = f4_ld(gamma + f), b = f4_ld(beta + f);
                xh[c] = make_float4((x.x - mean) * rstd, (x.y - mean) * rstd, (x.z - mean) * rstd, (x.w - mean) * rstd);
                float4 z = f4_fma(xh[c], g, b);
                float4 gz = make_float4(gy.x * dsilu_f(z.x), gy.y * dsilu_f(z.y), gy.z * dsilu_f(z.z), gy.w * dsilu_f(z.w));
                db[c] = f4_add(db[c], gz);
                dg[c] = f4_fma(gz, xh[c], dg[c]);
                gh[c] = f4_mul(gz, g);
                s1 += f4_hsum(gh[c]);
                s2 += f4_hsum(f4_mul(gh[c], xh[c]));
            }
        }
        const float c1 = wave_sum(s1) * inv_f, c2 = wave_sum(s2) * inv_f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int f = c * 256 + 4 * lane;
            if (f < F) {
                float4 o;
                o.x = rstd * (gh[c].x - c1 - xh[c].x * c2);
                o.y = rstd * (gh[c].y - c1 - xh[c].y * c2);
                o.z = rstd * (gh[c].z - c1 - xh[c].z * c2);
                o.w = rstd * (gh[c].w - c1 - xh[c].w * c2);
                f4_st(GX + r * ldgx + f, o);
                am = fmaxf(am, f4_absmax(o));
                if (NODE) {
                    const float4 s0 = f4_ld(S0 + r * F + f), h = f4_ld(HH + r * F + f);
                    float4 g1;
                    g1.x = o.x / (s0.x + ALIGNN_EPS_GATE);
                    g1.y = o.y / (s0.y + ALIGNN_EPS_GATE);
                    g1.z = o.z / (s0.z + ALIGNN_EPS_GATE);
                    g1.w = o.w / (s0.w + ALIGNN_EPS_GATE);
                    f4_st(GS1 + r * F + f, g1);
                    f4_st(GS0 + r * F + f, make_float4(-g1.x * h.x, -g1.y * h.y, -g1.z * h.z, -g1.w * h.w));
                }
            }
        }
    }
    block_amax_commit(am, amax);
    // slab [blockIdx.x][2][F]: row 0 = dbeta partial (sum gz), row 1 = dgamma partial (sum gz*xhat)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        sh[0][wave][lane] = db[c];
        sh[1][wave][lane] = dg[c];
        __syncthreads();
        const int f = c * 256 + 4 * lane;
        if (wave == 0 && f < F) {
            float4 a = sh[0][0][lane], b = sh[1][0][lane];
#pragma unroll
            for (int w = 1; w < kThreads / 64; ++w) {
                a = f4_add(a, sh[0][w][lane]);
                b = f4_add(b, sh[1][w][lane]);
            }
            f4_st(partial + (size_t)blockIdx.x * 2 * F + f, a);
            f4_st(partial + (size_t)blockIdx.x * 2 * F + F + f, b);
        }
        __syncthreads();
    }
}

inline int ln_blocks(int64_t rows) {
    int64_t b = (rows + (kThreads / 64) - 1) / (kThreads / 64);
    if (b < 1) b = 1;
    if (b > kMaxSlabs) b = kMaxSlabs;
    return (int)b;
}

inline bool feat_ok(int F) { return F >= 4 && (F & 3) == 0 && F <= 1024; }
// One float4 per thread: on MI355X a 2-read 1-write pass over 2 GB runs at 6.1 TB/s that way (6.7 with nontemporal
// accesses) against 4.6-4.8 TB/s for a 2048-workgroup grid-stride loop (tools/stream_bench.hip) - workgroups are
// dispatched in address order, so the accesses in flight form one moving window instead of drifting apart.
inline int stream_grid(int64_t total, bool big = true) {
    int64_t g = (total + kThreads - 1) / kThreads;
    if (g > (1 << 22)) g = 1 << 22;
    // cache-resident (E-, N-row) tensors gain nothing from the window pattern but pay the amax commit of every
    // workgroup (a launch-wide burst of atomics): 1024 grid-stride workgroups: 30 us instead of 54 us at E rows
    if (!big && g > 1024) g = 1024;
    if (g < 1) g = 1;
    return (int)g;
}
// read-once / write-once hint only for tensors that cannot stay in the 256 MiB last-level cache anyway
inline bool streaming(int64_t rows, int F) { return rows * (int64_t)F * 4 >= (int64_t)128 << 20; }

}  // namespace

extern "C" {

int alignn_col_stats_slabs(int64_t rows) { return slabs_for(rows); }

int alignn_col_stats(const float* X, int64_t ldx, int64_t rows, int F, float* partial, alignn_stream_t stream) {
    if (!feat_ok(F) || rows < 0) return (int)hipErrorInvalidValue;
    int slabs = slabs_for(rows);
    if (streaming(rows, F)) {
        StatsFn<true> fn{X, ldx};
        hipLaunchKernelGGL(col_reduce_kernel<StatsFn<true>>, dim3(slabs), dim3(kThreads), 0, (hipStream_t)stream, fn,
                           rows, F, slabs, partial);
    } else {
        StatsFn<false> fn{X, ldx};
        hipLaunchKernelGGL(col_reduce_kernel<StatsFn<false>>, dim3(slabs), dim3(kThreads), 0, (hipStream_t)stream, fn,
                           rows, F, slabs, partial);
    }
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_col_stats_welford(const float* X, int64_t ldx, int64_t rows, int F, float* partial, alignn_stream_t stream) {
    if (!feat_ok(F) || rows < 0) return (int)hipErrorInvalidValue;
    int slabs = slabs_for(rows);
    if (streaming(rows, F))
        hipLaunchKernelGGL(col_stats_welford_kernel<true>, dim3(slabs), dim3(kThreads), 0, (hipStream_t)stream, X, ldx, rows, F,
                           slabs, partial);
    else
        hipLaunchKernelGGL(col_stats_welford_kernel<false>, dim3(slabs), dim3(kThreads), 0, (hipStream_t)stream, X, ldx, rows,
                           F, slabs, partial);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_bn_finalize_welford(const float* partial, int slabs, int64_t rows, int F, const float* gamma, const float* beta,
                               float eps, float momentum, float* running_mean, float* running_var, float* stat,
                               alignn_stream_t stream) {
    if (F <= 0 || slabs <= 0 || partial == nullptr) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(bn_finalize_welford_kernel, dim3(alignn_ceil_div(F, kWfCols)), dim3(kWfCols, kWfLanes), 0,
                       (hipStream_t)stream, partial, partial + (size_t)slabs * 3 * F, slabs, rows, F, gamma, beta, eps, momentum,
                       running_mean, running_var, stat);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_col_sum(const float* X, int64_t ldx, int64_t rows, int F, float* out, float* workspace,
                   alignn_stream_t stream) {
    if (F < 4 || (F & 3) || rows < 0) return (int)hipErrorInvalidValue;
    int slabs = slabs_for(rows);
    // wide matrices (the [n,4H] projection gradient) go in column panels of <= 1024
    for (int c = 0; c < F; c += 1024) {
        const int w = F - c < 1024 ? F - c : 1024;
        if (streaming(rows, w)) {  // (a T-row gradient read once: bias gradient of the angle embedding)
            StatsFn<true> fn{X + c, ldx};
            hipLaunchKernelGGL(col_reduce_kernel<StatsFn<true>>, dim3(slabs), dim3(kThreads), 0, (hipStream_t)stream, fn,
                               rows, w, slabs, workspace);
        } else {
            StatsFn<false> fn{X + c, ldx};
            hipLaunchKernelGGL(col_reduce_kernel<StatsFn<false>>, dim3(slabs), dim3(kThreads), 0, (hipStream_t)stream, fn,
                               rows, w, slabs, workspace);
        }
        hipLaunchKernelGGL(slab_sum_kernel, dim3(alignn_ceil_div(w, kRedCols)), dim3(kRedCols, kRedLanes), 0,
                           (hipStream_t)stream, workspace, slabs, w, 2 * w, out + c);
    }
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_bn_finalize(const float* partial, int slabs, int64_t rows, int F, const float* gamma, const float* beta,
                       float eps, float momentum, float* running_mean, float* running_var, float* stat,
                       alignn_stream_t stream) {
    if (F <= 0 || (slabs == 0 && running_mean == nullptr)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(alignn_ceil_div(F, kRedCols)), dim3(kRedCols, kRedLanes), 0,
                       (hipStream_t)stream, partial, slabs, rows, F, gamma, beta, eps, momentum, running_mean,
                       running_var, stat);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_bn_silu_fwd(const float* X, int64_t ldx, const float* R, int64_t ldr, const float* stat, float* Y,
                       int64_t ldy, int64_t rows, int F, float* amax, alignn_stream_t stream) {
    if (!feat_ok(F)) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    if (F <= 4 * kThreads && rows >= 16384) {  // tall: column-walking form
        const int slabs = slabs_for(rows);
#define ALIGNN_BNFC(RES_, ST_)                                                                                          \
    hipLaunchKernelGGL((bn_silu_fwd_cols_kernel<RES_, ST_>), dim3(slabs), dim3(kThreads), 0, (hipStream_t)stream, X, ldx, R, \
                       ldr, stat, Y, ldy, rows, F, slabs, amax)
        if (streaming(rows, F)) {
            if (R) ALIGNN_BNFC(true, true); else ALIGNN_BNFC(false, true);
        } else {
            if (R) ALIGNN_BNFC(true, false); else ALIGNN_BNFC(false, false);
        }
#undef ALIGNN_BNFC
        ALIGNN_CHECK_LAUNCH();
        return 0;
    }
    int grid = stream_grid(rows * (F >> 2), streaming(rows, F));
#define ALIGNN_BNF(RES_, ST_)                                                                                      \
    hipLaunchKernelGGL((bn_silu_fwd_kernel<RES_, ST_>), dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, X, ldx, R, \
                       ldr, stat, Y, ldy, rows, F, amax)
    if (streaming(rows, F)) {
        if (R) ALIGNN_BNF(true, true); else ALIGNN_BNF(false, true);
    } else {
        if (R) ALIGNN_BNF(true, false); else ALIGNN_BNF(false, false);
    }
#undef ALIGNN_BNF
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_bn_silu_bwd_reduce(const float* GY, int64_t ldgy, const float* X, int64_t ldx, const float* stat,
                              int64_t rows, int F, float* partial, alignn_stream_t stream) {
    if (!feat_ok(F)) return (int)hipErrorInvalidValue;
    int slabs = slabs_for(rows);
    if (streaming(rows, F)) {  // T-sized: the re-read by the apply / conv-backward pass cannot come from cache anyway
        BwdReduceFn<true> fn{GY, ldgy, X, ldx, stat, F};
        hipLaunchKernelGGL(col_reduce_kernel<BwdReduceFn<true>>, dim3(slabs), dim3(kThreads), 0, (hipStream_t)stream, fn,
                           rows, F, slabs, partial);
    } else {
        BwdReduceFn<false> fn{GY, ldgy, X, ldx, stat, F};
        hipLaunchKernelGGL(col_reduce_kernel<BwdReduceFn<false>>, dim3(slabs), dim3(kThreads), 0, (hipStream_t)stream, fn,
                           rows, F, slabs, partial);
    }
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_bn_bwd_finalize(const float* partial, int slabs, int F, float* red, alignn_stream_t stream) {
    if (F <= 0 || slabs <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(slab_sum_kernel, dim3(alignn_ceil_div(2 * F, kRedCols)), dim3(kRedCols, kRedLanes), 0,
                       (hipStream_t)stream, partial, slabs, 2 * F, 2 * F, red);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_ln_slabs(int64_t rows) { return ln_blocks(rows); }

int alignn_ln_silu_fwd(const float* X, int64_t ldx, const float* R, int64_t ldr, const float* gamma, const float* beta,
                       float eps, float* Y, int64_t ldy, float* stats, int64_t rows, int F, float* amax,
                       alignn_stream_t stream) {
    if (!feat_ok(F)) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    const int nc = (F + 255) / 256;
    dim3 grid(ln_blocks(rows)), block(kThreads);
    hipStream_t st = (hipStream_t)stream;
#define ALIGNN_LN_FWD(NC_)                                                                                         \
    if (R)                                                                                                         \
        hipLaunchKernelGGL((ln_silu_fwd_kernel<NC_, true>), grid, block, 0, st, X, ldx, R, ldr, gamma, beta, eps, Y, \
                           ldy, stats, rows, F, amax);                                                                   \
    else                                                                                                           \
        hipLaunchKernelGGL((ln_silu_fwd_kernel<NC_, false>), grid, block, 0, st, X, ldx, R, ldr, gamma, beta, eps, Y, \
                           ldy, stats, rows, F, amax);
    switch (nc) {
        case 1: ALIGNN_LN_FWD(1) break;
        case 2: ALIGNN_LN_FWD(2) break;
        case 3: ALIGNN_LN_FWD(3) break;
        default: ALIGNN_LN_FWD(4) break;
    }
#undef ALIGNN_LN_FWD
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_ln_silu_bwd(const float* GY, int64_t ldgy, const float* X, int64_t ldx, const float* gamma,
                       const float* beta, const float* stats, float* GX, int64_t ldgx, float* partial, int64_t rows,
                       int F, float* amax, alignn_stream_t stream) {
    if (!feat_ok(F) || rows < 0) return (int)hipErrorInvalidValue;
    // rows == 0 is launched on purpose (ln_blocks(0) == 1): alignn_ln_slabs(0) == 1 and alignn_ln_bwd_finalize reads that slab,
    // which this one workgroup writes as zeros
    const int nc = (F + 255) / 256;
    dim3 grid(ln_blocks(rows)), block(kThreads);
    hipStream_t st = (hipStream_t)stream;
#define ALIGNN_LNB(NC_)                                                                                                        \
    hipLaunchKernelGGL((ln_silu_bwd_kernel<NC_, false>), grid, block, 0, st, GY, ldgy, X, ldx, gamma, beta, stats, GX, ldgx, partial, \
                       rows, F, amax, nullptr, nullptr, nullptr, nullptr)
    switch (nc) {
        case 1: ALIGNN_LNB(1); break;
        case 2: ALIGNN_LNB(2); break;
        case 3: ALIGNN_LNB(3); break;
        default: ALIGNN_LNB(4); break;
    }
#undef ALIGNN_LNB
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_ln_silu_bwd_node(const float* GY, int64_t ldgy, const float* X, int64_t ldx, const float* gamma, const float* beta,
                            const float* stats, float* GX, int64_t ldgx, float* partial, int64_t rows, int F, float* amax,
                            const float* S0, const float* HH, float* GS1, float* GS0, alignn_stream_t stream) {
    if (!feat_ok(F) || rows < 0 || !S0 || !HH || !GS1 || !GS0) return (int)hipErrorInvalidValue;
    const int nc = (F + 255) / 256;  // (rows == 0: one workgroup writes the zero slab, as alignn_ln_silu_bwd)
    dim3 grid(ln_blocks(rows)), block(kThreads);
    hipStream_t st = (hipStream_t)stream;
#define ALIGNN_LNB(NC_)                                                                                                       \
    hipLaunchKernelGGL((ln_silu_bwd_kernel<NC_, true>), grid, block, 0, st, GY, ldgy, X, ldx, gamma, beta, stats, GX, ldgx, partial, \
                       rows, F, amax, S0, HH, GS1, GS0)
    switch (nc) {
        case 1: ALIGNN_LNB(1); break;
        case 2: ALIGNN_LNB(2); break;
        case 3: ALIGNN_LNB(3); break;
        default: ALIGNN_LNB(4); break;
    }
#undef ALIGNN_LNB
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_slab_fold_slabs(void) { return kFold; }

int alignn_slab_fold(const float* partial, int slabs, int width, float* out, alignn_stream_t stream) {
    if (width <= 0 || slabs <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(slab_fold_kernel, dim3(alignn_ceil_div(width, 64), kFold), dim3(64, 4), 0, (hipStream_t)stream,
                       partial, slabs, width, out);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_slab_sum(const float* partial, int slabs, int width, float* out, alignn_stream_t stream) {
    if (width <= 0 || slabs <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(slab_sum_kernel, dim3(alignn_ceil_div(width, kRedCols)), dim3(kRedCols, kRedLanes), 0,
                       (hipStream_t)stream, partial, slabs, width, width, out);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_bn_silu_bwd_apply(const float* GY, int64_t ldgy, const float* X, int64_t ldx, const float* stat,
                             const float* gamma, const float* red, int eval_mode, float* GX, int64_t ldgx,
                             int64_t rows, int F, float* amax, alignn_stream_t stream) {
    if (!feat_ok(F)) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    int grid = stream_grid(rows * (F >> 2), streaming(rows, F));
    if (streaming(rows, F))
        hipLaunchKernelGGL((bn_silu_bwd_apply_kernel<true, false>), dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, GY, ldgy,
                           X, ldx, stat, gamma, red, eval_mode, GX, ldgx, rows, F, amax, nullptr, nullptr, nullptr, nullptr);
    else
        hipLaunchKernelGGL((bn_silu_bwd_apply_kernel<false, false>), dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, GY,
                           ldgy, X, ldx, stat, gamma, red, eval_mode, GX, ldgx, rows, F, amax, nullptr, nullptr, nullptr, nullptr);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_bn_silu_bwd_apply_node(const float* GY, int64_t ldgy, const float* X, int64_t ldx, const float* stat,
                                  const float* gamma, const float* red, int eval_mode, float* GX, int64_t ldgx, int64_t rows,
                                  int F, float* amax, const float* S0, const float* HH, float* GS1, float* GS0,
                                  alignn_stream_t stream) {
    if (!feat_ok(F) || S0 == nullptr || HH == nullptr || GS1 == nullptr || GS0 == nullptr) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    int grid = stream_grid(rows * (F >> 2), streaming(rows, F));
    if (streaming(rows, F))
        hipLaunchKernelGGL((bn_silu_bwd_apply_kernel<true, true>), dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, GY, ldgy,
                           X, ldx, stat, gamma, red, eval_mode, GX, ldgx, rows, F, amax, S0, HH, GS1, GS0);
    else
        hipLaunchKernelGGL((bn_silu_bwd_apply_kernel<false, true>), dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, GY,
                           ldgy, X, ldx, stat, gamma, red, eval_mode, GX, ldgx, rows, F, amax, S0, HH, GS1, GS0);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

/* alignn_bn_silu_bwd_apply that also leaves the column sums of GX (the bias gradient of the Linear in front of the norm)
 * as [alignn_col_stats_slabs(rows)][F] slabs for alignn_slab_sum; F <= 1024. */
int alignn_bn_silu_bwd_apply_sum(const float* GY, int64_t ldgy, const float* X, int64_t ldx, const float* stat,
                                 const float* red, int eval_mode, float* GX, int64_t ldgx, int64_t rows, int F,
                                 float* amax, float* partial, alignn_stream_t stream) {
    if (!feat_ok(F) || F > 4 * kThreads || partial == nullptr || (!eval_mode && red == nullptr)) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    const int slabs = slabs_for(rows);
    if (streaming(rows, F))
        hipLaunchKernelGGL(bn_silu_bwd_apply_sum_kernel<true>, dim3(slabs), dim3(kThreads), 0, (hipStream_t)stream, GY, ldgy, X,
                           ldx, stat, red, eval_mode, GX, ldgx, rows, F, slabs, amax, partial);
    else
        hipLaunchKernelGGL(bn_silu_bwd_apply_sum_kernel<false>, dim3(slabs), dim3(kThreads), 0, (hipStream_t)stream, GY, ldgy,
                           X, ldx, stat, red, eval_mode, GX, ldgx, rows, F, slabs, amax, partial);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
