// The bond-angle embedding of ALIGNN on T rows without materialising anything T x 256 but its output:
//   z = MLPLayer(64 -> 256)( MLPLayer(bins -> 64)( RBFExpansion(h) ) ),   MLPLayer = Linear + BatchNorm1d (batch statistics) + SiLU
// (alignn/models/alignn.py:215-222 angle_embedding, :170-184 MLPLayer; alignn/models/utils.py:9-47 RBFExpansion).
//
// Every row of the embedding is a function of ONE scalar, the cosine h[t]: the [T, bins] expansion, the [T, 64] and [T, 256]
// pre-activations and the [T, 64] activation (1.15 GB per 64 crystals) exist in the chain of separate kernels only to be
// read again by the next one - 3.2 GB of traffic forward and 4.5 GB backward for 0.69 GB of result.  Here every pass
// RECOMPUTES what it needs from h (2.7 MB) on the matrix cores and only the real operands cross HBM:
//   forward   pass 1  statistics of layer 1's pre-activation               reads h
//             pass 2  statistics of layer 2's pre-activation,
//                     mean and covariance of layer 1's activation a1       reads h
//             pass 3  z = SiLU(BatchNorm(.))                               reads h, writes z [T, 256]
//   backward  pass 4  BatchNorm-backward sums of layer 2 AND
//                     G = sum gz (a1 - mean), which becomes dW2            reads h, g_z [T, 256]
//             pass 5  da1 = dx2 W2                                         reads h, g_z, writes da1 [T, 64]
//             pass 6  BatchNorm-backward sums of layer 1                   reads h, da1
//             pass 7  dW1 = dx1^T rbf (and db1)                            reads h, da1
// BatchNorm's global statistics are what forces the passes apart (each is a grid-wide dependency).
//
// Arithmetic: products as f16x3 split products (hi * hi + hi * lo + lo * hi, fp32 accumulation: v_mfma_f32_32x32x16_f16) of
// operands scaled by powers of two from tracked / bounded maxima, like the projections of csrc/gemm_x6.hip; everything
// else fp32, column sums in float64.  Register tiles change orientation (rows <-> features across lanes) by a product
// with an identity fragment - exact for fp16 payloads - instead of an LDS round trip.
//
// Work decomposition.  Layer 1 of 32 rows is one wave's MFMA output with (lane = row, registers = features) - which IS the
// operand layout of the layer-2 product, so a wave keeps its rows' activations in registers.  Passes 2, 3, 5: one wave
// per block of 32 rows, W2's fragments in LDS, no barrier after the set-up (angle_rb_kernel).  Pass 4 accumulates dW2 and
// therefore splits the FEATURES over the waves of a workgroup (32 each), which share the rows' activations through LDS
// (angle_sums_dw2_kernel).  Passes 1, 6, 7 work on 64 features only (angle_l1_stats_kernel, angle_l1_bwd_kernel).
// Measured at T = 676 200 on MI355X (tools/angle_time.py, profiles/r04_angle_*): forward 0.36 ms, backward 0.71 ms, against
// 0.70 ms + 1.38 ms for the chain of kernels it replaces.
#include "../../include/alignn_hip.h"
#include <type_traits>

#include "common.h"

#ifndef ANGLE_GRID_DEFAULT
#define ANGLE_GRID_DEFAULT 224  // (headline step eager, one box, two rounds: 256: 14.45 / 14.59, 224: 14.31 / 14.55, 192: 14.38 / 14.54, 160: 14.56 / 14.56)
#endif

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kE = 64;         // embedding_features: layer-1 outputs
constexpr int kH = 256;        // hidden_features: layer-2 outputs
constexpr int kBinsMax = 48;   // three k-steps of 16
constexpr int kThreads = 256;
constexpr int kTile = 128;
constexpr int kGrid = 512;     // two workgroups per compute unit
// scal[] (device scalars that live from the forward to the backward of one step)
constexpr int kAmaxW1 = 0, kAmaxW2 = 1, kBoundA1 = 2, kBoundDx2 = 3, kBoundDx1 = 4, kAmaxGz2 = 5, kAmaxXh2 = 6, kAmaxGz1 = 7,
              kAmaxXh1 = 8, kDall = 9, kShift = 16,  // kShift .. kShift + 63: layer-1 shift of the statistics pass
              kScalHead = 128,             // what the forward zeroes (scalars that kernels raise atomically)
              kMeanA1 = 128,               // [64] mean of a1 (layer-1 activations), written by the forward
              kCovA1 = 192,                // [64][64] sum over rows of (a1 - mean)(a1 - mean)^T, written by the forward
              kScalFloats = 192 + 64 * 64; // 4288
constexpr float kRbfScale = 16384.0f;     // rbf values lie in (0, 1]

__device__ __forceinline__ f32x16 mfma(const f16x8& a, const f16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.0f;
    return z;
}
// power-of-two scale that puts max|x| just below 2^15 (csrc/gemm_x6.hip f16_scale)
__device__ __forceinline__ float f16_scale(float amax) {
    const int e = (int)((__float_as_uint(amax) >> 23) & 255u);
    if (e == 0 || e == 255) return 1.0f;
    int se = 268 - e;
    se = se > 254 ? 254 : se;
    return __uint_as_float((unsigned)se << 23);
}
// three products of a split pair
__device__ __forceinline__ f32x16 mfma3(const f16x8& ah, const f16x8& al, const f16x8& bh, const f16x8& bl, f32x16 c) {
    c = mfma(ah, bh, c);
    c = mfma(ah, bl, c);
    c = mfma(al, bh, c);
    return c;
}

// 8 floats already scaled -> hi = RN_f16(x), lo = RN_f16(x - hi): v_cvt_pk_f16_f32 for the high slice, one v_fma_mix per
// element for the low one (csrc/gemm_x6.hip slice8_f16_lo: same bits as the two-conversion form hipcc emits, half the work).
// The compiler does not look into inline assembly when it places the wait states gfx950 wants between dependent
// instructions of different pipes (a transcendental's result into a VALU read; a VALU result into an MFMA operand): the
// block carries its own - s_nop in front for whatever produced its inputs, s_nop behind for whatever consumes its outputs.
// (Without them: results that change from run to run, 1e-3 off.)
__device__ __forceinline__ void split8s(const float (&xs)[8], f16x8& h, f16x8& l) {
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (_Float16)xs[j];
    const uint4 hp = __builtin_bit_cast(uint4, h);
    unsigned l0, l1, l2, l3;
    asm volatile(
        "s_nop 1\n\t"
        "v_fma_mixlo_f16 %0, %4, 1.0, -%12 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %1, %6, 1.0, -%13 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %2, %8, 1.0, -%14 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %3, %10, 1.0, -%15 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %5, 1.0, -%12 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %7, 1.0, -%13 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %2, %9, 1.0, -%14 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %3, %11, 1.0, -%15 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "s_nop 3"
        : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
        : "v"(xs[0]), "v"(xs[1]), "v"(xs[2]), "v"(xs[3]), "v"(xs[4]), "v"(xs[5]), "v"(xs[6]), "v"(xs[7]), "v"(hp.x), "v"(hp.y),
          "v"(hp.z), "v"(hp.w));
    l = __builtin_bit_cast(f16x8, make_uint4(l0, l1, l2, l3));
}
constexpr float kLog2e = 1.4426950408889634f;
// x * sigmoid(x * k) with k folded by the caller: silu of a value that carries a scale (k = 1 / scale); k = 1: silu_f
__device__ __forceinline__ float silu_scaled(float x, float neg_k_log2e) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * neg_k_log2e));
}
__device__ __forceinline__ float dsilu_fast(float z) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -kLog2e));
    return s * fmaf(z, 1.0f - s, 1.0f);
}

struct P {  // kernel parameters (by value)
    const float* h;
    int64_t rows;
    const float* centers;
    float gamma;
    int bins;
    const float *W1, *b1, *gamma1, *beta1, *W2, *b2;
    const float *stat1, *stat2;  // [4, 64], [4, 256]: mean | rstd | gamma * rstd | beta
    float* scal;
    float *z, *z_amax;
    const float* gz;
    const float *red1, *red2;    // [2, F]: sum gz | sum gz * xhat
    float* da1;
    void* partial;
    float* partial_b;
    double* partial_d;
};

// ---------------------------------------------------------------------------------------------------------------------
// Layer 1 on a block of 32 rows (one wave): x1^T = W1 rbf^T as D[m = feature][n = row] - lane = row (il), register r of
// block cb = feature 32 cb + 8 (r >> 2) + 4 hh + (r & 3).  W1's fragments (A operand: lane m = feature 32 cb + il,
// k = bin 16 s + 8 hh + i) and the per-feature constants of what follows live in LDS (registers are what the T x 256
// passes are short of).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kL1Consts = 7;
struct L1Shared {
    uint4 w[2][3][2][64];       // [cb][s][hi | lo][lane]
    float c[kL1Consts][kE];     // per pass, see l1_setup
    float cen[kBinsMax];        // the centres, padded far away so that the padded bins expand to 0
};
enum L1Mode { kL1Stats, kL1Act, kL1Bwd };

// feature of register r of block cb for a lane of half hh
__device__ __forceinline__ int l1_col(int cb, int r, int hh) { return 32 * cb + 8 * (r >> 2) + 4 * hh + (r & 3); }

// all threads of the workgroup; ends with a barrier.  kL1Act: the activations come out scaled by sa (operand scale of a1).
__device__ __forceinline__ void l1_setup(const P& p, L1Shared& sh, L1Mode mode, float sa) {
    const int lane = threadIdx.x & 63, il = lane & 31, hh = lane >> 5, w = threadIdx.x >> 6;
    const float sw = f16_scale(p.scal[kAmaxW1]);
    const float inv = 1.0f / (kRbfScale * sw);
    for (int combo = w; combo < 6; combo += kThreads / 64) {
        const int cb = combo / 3, s = combo % 3;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = 16 * s + 8 * hh + i;
            v[i] = k < p.bins ? p.W1[(32 * cb + il) * p.bins + k] * sw : 0.0f;
        }
        f16x8 hi, lo;
        split8s(v, hi, lo);
        sh.w[cb][s][0][lane] = __builtin_bit_cast(uint4, hi);
        sh.w[cb][s][1][lane] = __builtin_bit_cast(uint4, lo);
    }
    if (threadIdx.x >= kThreads - kBinsMax && threadIdx.x < kThreads) {  // (workgroups of 256 or 512 threads)
        const int k = threadIdx.x - (kThreads - kBinsMax);
        sh.cen[k] = k < p.bins ? p.centers[k] : 1.0e18f;
    }
    if (threadIdx.x < kE) {
        const int f = threadIdx.x;
        const float b1 = p.b1[f];
        if (mode == kL1Stats) {  // d = x1 - shift = acc * c0 + c1
            sh.c[0][f] = inv;
            sh.c[1][f] = b1 - p.scal[kShift + f];
        } else {
            const float mean = p.stat1[f], rstd = p.stat1[kE + f], sc = p.stat1[2 * kE + f], be = p.stat1[3 * kE + f];
            if (mode == kL1Act) {  // sa z1 = acc * c0 + c1
                sh.c[0][f] = sc * inv * sa;
                sh.c[1][f] = fmaf(b1 - mean, sc, be) * sa;
            } else {  // xhat = acc * c0 + c1;  z1 = xhat * c2 + c3;  dx1 = c4 gz1 + xhat c5 + c6  (c5, c6 carry -1/n sums)
                const float inv_n = 1.0f / (float)p.rows;
                sh.c[0][f] = rstd * inv;
                sh.c[1][f] = (b1 - mean) * rstd;
                sh.c[2][f] = p.gamma1[f];
                sh.c[3][f] = be;
                sh.c[4][f] = sc * sa;
                sh.c[5][f] = p.red1 != nullptr ? -sc * sa * p.red1[kE + f] * inv_n : 0.0f;
                sh.c[6][f] = p.red1 != nullptr ? -sc * sa * p.red1[f] * inv_n : 0.0f;
            }
        }
    }
    __syncthreads();
}

// 2^14 exp(-gamma (h - c)^2) of this lane's row as fragments (lane = row, k = bin 16 s + 8 hh + i); ONES: slot `ones` holds
// 2^14 instead (a column of ones: the product with it sums the other operand's columns)
template <bool ONES>
__device__ __forceinline__ void l1_rbf(const L1Shared& sh, float g2, float hv, int hh, int ones, f16x8 (&r_hi)[3], f16x8 (&r_lo)[3]) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const float4 c0 = f4_ld(&sh.cen[16 * s + 8 * hh]), c1 = f4_ld(&sh.cen[16 * s + 8 * hh + 4]);
        const float cen[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float t = hv - cen[i];
            // (the multiply is also what keeps a transcendental's result from feeding split8s' inline assembly directly:
            // gfx950 wants a wait state there and the compiler cannot see into the asm to insert it)
            v[i] = __builtin_amdgcn_exp2f(g2 * t * t) * kRbfScale;
            if (ONES && 16 * s + 8 * hh + i == ones) v[i] = kRbfScale;
        }
        split8s(v, r_hi[s], r_lo[s]);
    }
}
__device__ __forceinline__ void l1_product(const L1Shared& sh, int lane, const f16x8 (&r_hi)[3], const f16x8 (&r_lo)[3],
                                           f32x16 (&acc)[2]) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        acc[cb] = zero16();
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const f16x8 wh = __builtin_bit_cast(f16x8, sh.w[cb][s][0][lane]);
            const f16x8 wl = __builtin_bit_cast(f16x8, sh.w[cb][s][1][lane]);
            acc[cb] = mfma3(wh, wl, r_hi[s], r_lo[s], acc[cb]);
        }
    }
}
// constants k of the four features registers [4 q, 4 q + 4) of block cb hold
__device__ __forceinline__ void l1_const4(const L1Shared& sh, int k, int cb, int q, int hh, float (&o)[4]) {
    const float4 v = f4_ld(&sh.c[k][32 * cb + 8 * q + 4 * hh]);
    o[0] = v.x, o[1] = v.y, o[2] = v.z, o[3] = v.w;
}

// the cosine of row `row` (rows past the end read the last one; what they produce is masked)
__device__ __forceinline__ float load_h(const P& p, int64_t row) { return p.h[row < p.rows ? row : p.rows - 1]; }

// cross-lane sum over the 32 rows of a half (lanes il = 0..31 keep hh)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ void atomic_max_pos(float* p, float v) {
    if (v > 0.0f && v > *reinterpret_cast<volatile float*>(p)) atomicMax(reinterpret_cast<unsigned*>(p), __float_as_uint(v));
}

// ---------------------------------------------------------------------------------------------------------------------
// prep: max|W1|, max|W2| and the shift of the layer-1 statistics (x1 of row 0 - any value near the column means will do)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void angle_prep_kernel(P p) {
    // blocks 0 .. gridDim.x - 2: a slice of W2 each; the last block: W1 and the shift.  scal[] was zeroed before.
    __shared__ float sh[kThreads];
    const int t = threadIdx.x, nb = gridDim.x - 1;
    const bool last = (int)blockIdx.x == nb;
    float m = 0.0f;
    if (last) {
        for (int i = t; i < kE * p.bins; i += kThreads) m = fmaxf(m, fabsf(p.W1[i]));
    } else {
        const int per = (kH * kE + nb - 1) / nb, beg = blockIdx.x * per, end = beg + per < kH * kE ? beg + per : kH * kE;
        for (int i = beg + t; i < end; i += kThreads) m = fmaxf(m, fabsf(p.W2[i]));
    }
    sh[t] = m;
    __syncthreads();
    for (int o = kThreads / 2; o > 0; o >>= 1) {
        if (t < o) sh[t] = fmaxf(sh[t], sh[t + o]);
        __syncthreads();
    }
    if (t == 0) atomic_max_pos(p.scal + (last ? kAmaxW1 : kAmaxW2), sh[0]);
    if (last && t < kE) {
        const float hv = p.rows > 0 ? p.h[0] : 0.0f;
        float acc = p.b1[t];
        for (int k = 0; k < p.bins; ++k) {
            const float d = hv - p.centers[k];
            acc = fmaf(__expf(-p.gamma * d * d), p.W1[t * p.bins + k], acc);
        }
        p.scal[kShift + t] = acc;
    }
}

// scal[] = 0 (a kernel, not hipMemsetAsync: a memset node captured into a hipGraph raced with the launch behind it in round 3,
// profiles/README.md "hipMemsetAsync inside a hipGraph")
__global__ void angle_zero_kernel(float* __restrict__ p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0f;
}

// evaluation mode: the bound of |a1| from the weights alone - rbf lies in (0, 1], so |x1 - b1| <= sum_k |W1[f][k]|
__global__ __launch_bounds__(kE) void angle_infer_bound_kernel(P p) {
    const int f = threadIdx.x;
    float l1 = 0.0f;
    for (int k = 0; k < p.bins; ++k) l1 += fabsf(p.W1[f * p.bins + k]);
    const float bound = fabsf(p.stat1[2 * kE + f]) * (l1 + fabsf(p.b1[f] - p.stat1[f])) + fabsf(p.stat1[3 * kE + f]);
    atomic_max_pos(p.scal + kBoundA1, bound);
}

// ---------------------------------------------------------------------------------------------------------------------
// pass 1: shifted column sums of x1 = W1 rbf(h) + b1.  One wave per block of 32 rows; partial[wave][2][64] =
// sum (x - c) | sum (x - c)^2; scal[kDall] = max |x - c| over everything.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads, 2) void angle_l1_stats_kernel(P p) {
    __shared__ L1Shared sh;
    l1_setup(p, sh, kL1Stats, 1.0f);
    const int lane = threadIdx.x & 63, il = lane & 31, hh = lane >> 5;
    const int wave = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6), nwaves = gridDim.x * (kThreads / 64);
    const float g2 = -p.gamma * kLog2e;
    float s1[2][16], s2[2][16], dall = 0.0f;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s1[cb][r] = s2[cb][r] = 0.0f;
    const int64_t nblk = (p.rows + 31) / 32;
    float hv_next = load_h(p, (int64_t)wave * 32 + il);
    for (int64_t blk = wave; blk < nblk; blk += nwaves) {
        const int64_t row = blk * 32 + il;
        const float ok = row < p.rows ? 1.0f : 0.0f;
        const float hv = hv_next;
        hv_next = load_h(p, (blk + nwaves) * 32 + il);
        f16x8 r_hi[3], r_lo[3];
        l1_rbf<false>(sh, g2, hv, hh, 0, r_hi, r_lo);
        f32x16 acc[2];
        l1_product(sh, lane, r_hi, r_lo, acc);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float u[4], v[4];
                l1_const4(sh, 0, cb, q, hh, u);
                l1_const4(sh, 1, cb, q, hh, v);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * q + e;
                    const float d = ok * fmaf(acc[cb][r], u[e], v[e]);
                    s1[cb][r] += d;
                    s2[cb][r] = fmaf(d, d, s2[cb][r]);
                    dall = fmaxf(dall, fabsf(d));
                }
            }
    }
    float* out = static_cast<float*>(p.partial) + (size_t)wave * 2 * kE;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float a = half_sum(s1[cb][r]), b = half_sum(s2[cb][r]);
            if (il == 0) {
                const int col = l1_col(cb, r, hh);
                out[col] = a;
                out[kE + col] = b;
            }
        }
    block_amax_commit(dall, p.scal + kDall);
}

// statistics of a layer from slabs.  LAYER2 = false: float slabs [slabs][2][F] = shifted sum | shifted sum of squares
// (shift in scal[kShift..], max |x - c| in scal[kDall]) (layer 1: also writes the bound of |a1|); true: double slabs [slabs][2][F] of the RAW
// accumulator sums: x2 = acc * unit + bias with unit = 1 / (a1 scale * W2 scale) - the variance does not see the bias.
// 4 columns x 64 slab lanes per workgroup (csrc/norm.hip bn_finalize_kernel).
template <bool LAYER2>
__global__ __launch_bounds__(256) void angle_stat_finalize_kernel(const void* __restrict__ partial, int slabs, int64_t rows, int F,
                                                                  const float* __restrict__ bias, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float eps, float momentum,
                                                                  float* __restrict__ rm, float* __restrict__ rv,
                                                                  float* __restrict__ stat, float* __restrict__ scal) {
    __shared__ double shs[64][4], shq[64][4];
    __shared__ float shm[64][4];
    const int c = threadIdx.x & 3, y = threadIdx.x >> 2;
    const int f = blockIdx.x * 4 + c;
    double s = 0.0, q = 0.0;
    float dmax = 0.0f;
    for (int k = y; k < slabs; k += 64) {
        if constexpr (LAYER2) {
            const double* pp = static_cast<const double*>(partial) + (size_t)k * 2 * F;
            s += pp[f];
            q += pp[F + f];
        } else {
            const float* pp = static_cast<const float*>(partial) + (size_t)k * 2 * F;
            s += (double)pp[f];
            q += (double)pp[F + f];
        }
    }
    shs[y][c] = s;
    shq[y][c] = q;
    shm[y][c] = dmax;
    __syncthreads();
    if (y != 0) return;
    s = 0.0, q = 0.0, dmax = 0.0f;
    for (int k = 0; k < 64; ++k) s += shs[k][c], q += shq[k][c], dmax = fmaxf(dmax, shm[k][c]);
    const double n = (double)rows;
    double unit = 1.0, shift;
    if constexpr (LAYER2) {
        unit = 1.0 / ((double)f16_scale(scal[kBoundA1]) * (double)f16_scale(scal[kAmaxW2]));
        shift = (double)bias[f];
    } else
        shift = (double)scal[kShift + f];
    const double dm = s / n * unit;
    double v = (q / n) * unit * unit - dm * dm;
    if (v < 0.0) v = 0.0;
    const float mean = (float)(shift + dm), var = (float)v;
    if (rm != nullptr) {
        const double unbiased = rows > 1 ? v * n / (n - 1.0) : v;
        rm[f] = (1.0f - momentum) * rm[f] + momentum * mean;
        rv[f] = (1.0f - momentum) * rv[f] + momentum * (float)unbiased;
    }
    const float rstd = 1.0f / sqrtf(var + eps);
    const float g = gamma[f], b = beta[f];
    stat[f] = mean;
    stat[F + f] = rstd;
    stat[2 * F + f] = g * rstd;
    stat[3 * F + f] = b;
    if constexpr (!LAYER2) {  // |a1| = |silu(z)| <= |z| <= |gamma rstd| (max|x - c| + |c - mean|) + |beta|
        const float bound = fabsf(g * rstd) * (scal[kDall] + fabsf((float)dm)) + fabsf(b);
        atomic_max_pos(scal + kBoundA1, bound);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Phase A of the T x 256 passes: the layer-1 activations a1 of 32 rows as fp16 slices in operand layout.
// K-step s2 = 2 cb + s of the layer-2 products takes registers [8 s, 8 s + 8) of block cb: slot (hh, i) of that step is
// feature 16 s2 + 8 (i >> 2) + 4 hh + (i & 3) - W2's fragments are loaded with the same permutation.
// LDS: a_frag[rb][s2][hi | lo][lane] (16 B each).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int afrag_idx(int rb, int s2, int hl, int lane) { return ((rb * 4 + s2) * 2 + hl) * 64 + lane; }

// KEEP: the fragments of each block of 32 features are also handed to per_block - of a1 MINUS `centre` (64 floats in LDS,
// scaled like a1) when that is given: the caller's product with them is then free of a1's mean
template <bool KEEP, typename PerBlock>
__device__ __forceinline__ void phase_a(const L1Shared& sh, float g2, float neg_k, float hv, int lane, int rb, uint4* a_frag,
                                        PerBlock per_block, const float* centre = nullptr) {
    const int hh = lane >> 5;
    f32x16 acc[2];
    {
        f16x8 r_hi[3], r_lo[3];
        l1_rbf<false>(sh, g2, hv, hh, 0, r_hi, r_lo);
        l1_product(sh, lane, r_hi, r_lo, acc);
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        f16x8 keep_hi[2], keep_lo[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float a[8];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float u[4], v[4];
                l1_const4(sh, 0, cb, 2 * s + q, hh, u);
                l1_const4(sh, 1, cb, 2 * s + q, hh, v);
#pragma unroll
                for (int e = 0; e < 4; ++e) a[4 * q + e] = silu_scaled(fmaf(acc[cb][8 * s + 4 * q + e], u[e], v[e]), neg_k);
            }
            f16x8 hi, lo;
            split8s(a, hi, lo);
            a_frag[afrag_idx(rb, 2 * cb + s, 0, lane)] = __builtin_bit_cast(uint4, hi);
            a_frag[afrag_idx(rb, 2 * cb + s, 1, lane)] = __builtin_bit_cast(uint4, lo);
            if constexpr (KEEP) {
                if (centre != nullptr) {  // (uniform)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const float4 c4 = f4_ld(centre + 32 * cb + 8 * (2 * s + q) + 4 * hh);
                        a[4 * q] -= c4.x, a[4 * q + 1] -= c4.y, a[4 * q + 2] -= c4.z, a[4 * q + 3] -= c4.w;
                    }
                    split8s(a, keep_hi[s], keep_lo[s]);
                } else
                    keep_hi[s] = hi, keep_lo[s] = lo;
            }
        }
        if constexpr (KEEP) per_block(cb, keep_hi, keep_lo);  // steps 2 cb, 2 cb + 1 = features 32 cb .. 32 cb + 31
    }
}

__device__ __forceinline__ void wave_double_pair_store(double a, double b, double* out_a, double* out_b, int hh) {
    // the two halves of a wave hold different rows of the same feature
    a += __shfl_xor(a, 32, 64);
    b += __shfl_xor(b, 32, 64);
    if (hh == 0) *out_a = a, *out_b = b;
}

// row of register r of row block rb for a lane of half hh, relative to the tile
__device__ __forceinline__ int n_row(int rb, int r, int hh) { return 32 * rb + 8 * (r >> 2) + 4 * hh + (r & 3); }

// BatchNorm-backward sums from double slabs [slabs][2][F] -> red [2, F]; and the bound of |dx| = |gamma rstd (gz - (c0 + xhat c1) / n)|
__global__ __launch_bounds__(256) void angle_red_finalize_kernel(const double* __restrict__ partial, int slabs, int64_t rows, int F,
                                                                 const float* __restrict__ stat, float* __restrict__ red,
                                                                 float* __restrict__ scal, int amax_gz, int amax_xh, int bound) {
    __shared__ double shs[64][4], shq[64][4];
    const int c = threadIdx.x & 3, y = threadIdx.x >> 2;
    const int f = blockIdx.x * 4 + c;
    double s = 0.0, q = 0.0;
    for (int k = y; k < slabs; k += 64) {
        s += partial[(size_t)k * 2 * F + f];
        q += partial[(size_t)k * 2 * F + F + f];
    }
    shs[y][c] = s;
    shq[y][c] = q;
    __syncthreads();
    if (y != 0) return;
    s = 0.0, q = 0.0;
    for (int k = 0; k < 64; ++k) s += shs[k][c], q += shq[k][c];
    red[f] = (float)s;
    red[F + f] = (float)q;
    const float inv_n = 1.0f / (float)rows;
    const float b = fabsf(stat[2 * F + f]) * (scal[amax_gz] + inv_n * (fabsf((float)s) + scal[amax_xh] * fabsf((float)q)));
    atomic_max_pos(scal + bound, b);
}

// moments of a1 from their reduced sums [64 + 64 x 64] (scaled by sa, sa^2) -> scal[kMeanA1] = mean,
// scal[kCovA1] = sum_rows (a1 - mean)(a1 - mean)^T
__global__ __launch_bounds__(256) void angle_moments_finalize_kernel(const float* __restrict__ sums, int64_t rows,
                                                                     float* __restrict__ scal) {
    const int i = blockIdx.x * 256 + threadIdx.x;  // 0 .. 64 * 64 - 1: (k, j)
    if (i >= kE * kE) return;
    const int k = i / kE, j = i % kE;
    const double sa = (double)f16_scale(scal[kBoundA1]), n = (double)rows;
    const double mk = (double)sums[k] / (n * sa), mj = (double)sums[j] / (n * sa);
    scal[kCovA1 + i] = (float)((double)sums[kE + i] / (sa * sa) - n * mk * mj);
    if (k == 0) scal[kMeanA1 + j] = (float)mj;
}

// identity fragments for changing the orientation of a register tile: B[k][n] = 1 where slot k of step s names feature n
//   PERM: slot (hh, i) of step s = 16 s + 8 (i >> 2) + 4 hh + (i & 3)   (tiles that came out of an MFMA)
//   else: slot (hh, i) of step s = 16 s + 8 hh + i                      (tiles built in natural order)
template <bool PERM>
__device__ __forceinline__ f16x8 identity_frag(int s, int il, int hh) {
    f16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = PERM ? 16 * s + 8 * (i >> 2) + 4 * hh + (i & 3) : 16 * s + 8 * hh + i;
        v[i] = k == il ? (_Float16)1.0f : (_Float16)0.0f;
    }
    return v;
}
// registers [8 s, 8 s + 8) of an accumulator that holds fp16 payloads exactly -> fragment
__device__ __forceinline__ f16x8 pack8(const f32x16& d, int s) {
    f16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (_Float16)d[8 * s + i];
    return v;
}

// dx2 scaled by sd, per-feature constants folded:  sd dx2 = E gz + (acc F + G)  with  E = sd gamma rstd,
// F = -E c1 rstd / (n scale),  G = -E (c0 + (b2 - mean) rstd c1) / n
struct Dx2Const {
    float A, B, E, F, G;
    __device__ __forceinline__ void load(const P& p, int f, float inv2, float sd) {
        const float inv_n = 1.0f / (float)p.rows;
        const float d = p.b2[f] - p.stat2[f], rstd = p.stat2[kH + f], sc = p.stat2[2 * kH + f];
        const float c0 = p.red2[f] * inv_n, c1 = p.red2[kH + f] * inv_n;
        A = inv2 * sc;
        B = fmaf(d, sc, p.stat2[3 * kH + f]);
        E = sd * sc;
        F = -E * c1 * rstd * inv2;
        G = -E * fmaf(d * rstd, c1, c0);
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// pass 5: dW2[f][j] = sum_rows dx2[row][f] a1[row][j],  dx2 = gamma rstd (gz - (c0 + xhat c1) / n): the recomputed tile
// D[row][feature] (lane = feature, registers = rows) IS the A operand [m = feature][k = rows] of that product; its B operand,
// a1 as (lane = j, registers = rows), is a1's fragment turned by an identity product in phase A and filed in LDS.
// partial[workgroup][256][64], partial_b[workgroup][256] (column sums of dx2 = db2) floats.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int bfrag_idx(int rb, int jb, int s, int hl, int lane) { return (((rb * 2 + jb) * 2 + s) * 2 + hl) * 64 + lane; }

// This pass keeps 64 x 64 of dW2 per 64 features in accumulators: with four waves of 64 features each that, W2's fragments
// and one block of g_z leave no registers (hipcc spilled 90-230 of them: 0.3-0.7 ms).  So: EIGHT waves per workgroup, a
// tile of 256 rows (wave w: phase A of row block w), 32 features per wave (16 + 16 fragment registers of W2, 32 of dW2),
// one workgroup per compute unit (142 KiB of LDS).
//
// ONE pass for the BatchNorm-backward sums AND dW2.  dx2 = gamma rstd (gz - c0 / n - xhat c1 / n) needs the sums c0 = sum gz,
// c1 = sum gz xhat first - but its product with a1 does not:
//   dW2[f][j] = gamma rstd [ sum_r gz[r][f] (a1[r][j] - mean_j)  -  (c1[f] / n) rstd[f] sum_k W2[f][k] Cov[k][j] ],
// with mean and Cov = sum_r (a1 - mean)(a1 - mean)^T of a1 known from the forward (scal[kMeanA1], scal[kCovA1]): the c0 term
// drops out against the centred a1, and xhat[r][f] = rstd[f] sum_k W2[f][k] (a1[r][k] - mean_k).  So this kernel reads g_z
// once, accumulates c0, c1 and G[f][j] = sum_r gz (a1 - mean), and angle_dw2_fixup_kernel finishes dW2 once c1 is known.  (db2,
// the column sums of dx2, is zero by construction and is written as such.)
// gz has no known bound here (its maximum comes out of THIS pass): every 32 x 32 tile is sliced with its own power-of-two
// scale, multiplied into a fresh accumulator and added to dW2's with the inverse scale - the fp32 addition the MFMA itself
// would have made.
constexpr int kDwThreads = 512, kDwTile = 256, kDwGrid = 256;
#ifndef DW2_PIPE
#define DW2_PIPE 0  // (measured: 314 us with the recompute issued one block ahead - 256 registers, 27 spilled - against 295 us without)
#endif

__global__ __launch_bounds__(kDwThreads, 1) void angle_sums_dw2_kernel(P p) {
    __shared__ L1Shared sh;
    __shared__ uint4 a_frag[8 * 4 * 2 * 64];      // 64 KiB
    __shared__ uint4 b_frag[8 * 2 * 2 * 2 * 64];  // 64 KiB
    __shared__ __attribute__((aligned(16))) float centre[kE];  // sa * mean of a1
    const float sa = f16_scale(p.scal[kBoundA1]), sw = f16_scale(p.scal[kAmaxW2]);
    if (threadIdx.x < kE) centre[threadIdx.x] = sa * p.scal[kMeanA1 + threadIdx.x];
    l1_setup(p, sh, kL1Act, sa);  // (ends with a barrier)
    const int lane = threadIdx.x & 63, il = lane & 31, hh = lane >> 5, w = threadIdx.x >> 6;
    const float inv2 = 1.0f / (sa * sw), neg_k = -kLog2e / sa;
    const float g2 = -p.gamma * kLog2e;
    const int f = 32 * w + il;  // this lane's feature
    // W2 fragments of the wave's 32 features: lane n = feature, slot (hh, i) of step s2 = input 16 s2 + 8 (i >> 2) + 4 hh + (i & 3)
    f16x8 w_hi[4], w_lo[4];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
        const float* wp = p.W2 + (size_t)f * kE + 16 * s2 + 4 * hh;
        const float4 a = f4_ld(wp), b = f4_ld(wp + 8);
        const float v[8] = {a.x * sw, a.y * sw, a.z * sw, a.w * sw, b.x * sw, b.y * sw, b.z * sw, b.w * sw};
        split8s(v, w_hi[s2], w_lo[s2]);
    }
    // zl = acc A + B,  xhat = acc C + D
    const float d_ = p.b2[f] - p.stat2[f], rstd_ = p.stat2[kH + f], sc_ = p.stat2[2 * kH + f];
    const float cA = inv2 * sc_, cB = fmaf(d_, sc_, p.stat2[3 * kH + f]), cC = inv2 * rstd_, cD = d_ * rstd_;
    double c0 = 0.0, c1 = 0.0;
    float am0 = 0.0f, am1 = 0.0f;
    f32x16 dw[2] = {zero16(), zero16()};
    const int64_t ntiles = (p.rows + kDwTile - 1) / kDwTile;
    // (a tile that lies wholly inside the tensor: one 64-bit base per row block and group of eight rows, the rest immediate
    // offsets - the clamped form spent 3 compares / selects and a 64-bit multiply-add per load: 130 of the loop's 530 instructions)
    auto load_g = [&](int64_t row0, int rb, float (&g)[16]) {
        if (row0 + kDwTile <= p.rows) {  // uniform
            const float* base = p.gz + (row0 + 32 * rb + 4 * hh) * kH + f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float* bq = base + 8 * q * kH;
                asm volatile("" : "+v"(bq));  // (keep it ONE pointer per group: offsets 0 / 1 / 2 / 3 KiB are immediates)
#pragma unroll
                for (int e = 0; e < 4; ++e) g[4 * q + e] = __builtin_nontemporal_load(bq + e * kH);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int64_t row = row0 + n_row(rb, r, hh);
                row = row < p.rows ? row : p.rows - 1;
                g[r] = __builtin_nontemporal_load(p.gz + row * kH + f);
            }
        }
    };
    // one row block: recompute the wave's 32 features, gz and the sums, its share of G
    // (g: g_z of this row block on entry; refilled with the NEXT block's - of row0n / rbn - as soon as it has been consumed, so
    // one set of registers is both the operand and the prefetch: a second set spilled 85-91 registers)
    // the wave's 32 features of row block rb, recomputed: issued one block AHEAD of its use (DW2_PIPE), so that the matrix
    // pipe works on block rb + 1 while the vector pipe runs the element chain of block rb - with two waves per SIMD there is
    // nobody else to fill either pipe
    auto recompute = [&](int rb) {
        f32x16 acc = zero16();
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            const f16x8 ah = __builtin_bit_cast(f16x8, a_frag[afrag_idx(rb, s2, 0, lane)]);
            const f16x8 al = __builtin_bit_cast(f16x8, a_frag[afrag_idx(rb, s2, 1, lane)]);
            acc = mfma3(ah, al, w_hi[s2], w_lo[s2], acc);
        }
        return acc;
    };
    auto block = [&](auto full_c, int64_t row0, int rb, float (&g)[16], int64_t row0n, int rbn, f32x16& acc) {
        constexpr bool FULL = decltype(full_c)::value;
#if !DW2_PIPE
        acc = recompute(rb);
#endif
        float gzv[16], s0 = 0.0f, s1 = 0.0f, mx = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float a = acc[r];
            float gz = g[r] * dsilu_fast(fmaf(a, cA, cB)), xh = fmaf(a, cC, cD);
            if constexpr (!FULL) {  // (multiplies, not branches)
                const float m = row0 + n_row(rb, r, hh) < p.rows ? 1.0f : 0.0f;
                gz *= m, xh *= m;
            }
            gzv[r] = gz;
            s0 += gz;
            s1 = fmaf(gz, xh, s1);
            mx = fmaxf(mx, fabsf(gz));
            am1 = fmaxf(am1, fabsf(xh));
        }
        load_g(row0n, rbn, g);
#if DW2_PIPE
        if (rb < 7) acc = recompute(rb + 1);  // (acc's values are all in gzv / the sums by now)
#endif
        c0 += (double)s0;
        c1 += (double)s1;
        am0 = fmaxf(am0, mx);
        // the tile's own scale (wave-uniform: the 32 x 32 tile is the wave's)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const float sl = f16_scale(mx), inv_sl = 1.0f / sl;
        f16x8 d_hi[2], d_lo[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float dx[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) dx[i] = gzv[8 * s + i] * sl;
            split8s(dx, d_hi[s], d_lo[s]);
        }
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            f32x16 t = zero16();
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f16x8 bh = __builtin_bit_cast(f16x8, b_frag[bfrag_idx(rb, jb, s, 0, lane)]);
                const f16x8 bl = __builtin_bit_cast(f16x8, b_frag[bfrag_idx(rb, jb, s, 1, lane)]);
                t = mfma3(d_hi[s], d_lo[s], bh, bl, t);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) dw[jb][r] = fmaf(t[r], inv_sl, dw[jb][r]);
        }
    };
    float hv = load_h(p, (int64_t)blockIdx.x * kDwTile + 32 * w + il);
    float g0[16];
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * kDwTile;
        const bool full = row0 + kDwTile <= p.rows;  // uniform
        if (tile == (int64_t)blockIdx.x) load_g(row0, 0, g0);  // (later tiles: requested by the previous tile's last block)
        // a1 - mean of this wave's rows also as (lane = j, registers = rows), block of 32 features by block
        phase_a<true>(sh, g2, neg_k, hv, lane, w, a_frag, [&](int jb, const f16x8 (&k_hi)[2], const f16x8 (&k_lo)[2]) {
            const f16x8 id0 = identity_frag<true>(0, il, hh), id1 = identity_frag<true>(1, il, hh);  // (rebuilt: registers)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) {  // high slices, then low slices: one 16-register tile at a time
                f32x16 t = mfma(hl ? k_lo[0] : k_hi[0], id0, zero16());
                t = mfma(hl ? k_lo[1] : k_hi[1], id1, t);
#pragma unroll
                for (int s = 0; s < 2; ++s) b_frag[bfrag_idx(w, jb, s, hl, lane)] = __builtin_bit_cast(uint4, pack8(t, s));
                __builtin_amdgcn_sched_barrier(0);
            }
        }, centre);
        hv = load_h(p, row0 + (int64_t)gridDim.x * kDwTile + 32 * w + il);
        __syncthreads();
        const int64_t row0_next = row0 + (int64_t)gridDim.x * kDwTile;
        auto row_blocks = [&](auto full_c) {
            f32x16 acc;
#if DW2_PIPE
            acc = recompute(0);
#endif
#pragma unroll 1
            for (int rb = 0; rb < 8; ++rb) block(full_c, row0, rb, g0, rb < 7 ? row0 : row0_next, rb < 7 ? rb + 1 : 0, acc);
        };
        if (full)
            row_blocks(std::true_type{});
        else
            row_blocks(std::false_type{});
        __syncthreads();
    }
    // dw[jb]: D[m = feature][n = j]: lane = j, register r = feature 32 w + 8 (r >> 2) + 4 hh + (r & 3); G = dw / sa
    float* out = static_cast<float*>(p.partial) + (size_t)blockIdx.x * (kH * kE);
    const float inv_sa = 1.0f / sa;
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(32 * w + 8 * (r >> 2) + 4 * hh + (r & 3)) * kE + 32 * jb + il] = dw[jb][r] * inv_sa;
    double* so = p.partial_d + (size_t)blockIdx.x * 2 * kH;
    wave_double_pair_store(c0, c1, so + f, so + kH + f, hh);
    block_amax_commit(am0, p.scal + kAmaxGz2);
    block_amax_commit(am1, p.scal + kAmaxXh2);
}

// dW2 = gamma rstd (G - (c1 / n) rstd W2 Cov) in place on the reduced G;  db2 = 0
__global__ __launch_bounds__(kE) void angle_dw2_fixup_kernel(P p, float* __restrict__ gW, float* __restrict__ gb) {
    const int f = blockIdx.x, j = threadIdx.x;
    const float* cov = p.scal + kCovA1;
    float x = 0.0f;
    for (int k = 0; k < kE; ++k) x = fmaf(p.W2[f * kE + k], cov[k * kE + j], x);
    const float rstd = p.stat2[kH + f], sc = p.stat2[2 * kH + f], c1n = p.red2[kH + f] / (float)p.rows;
    gW[f * kE + j] = sc * (gW[f * kE + j] - c1n * rstd * x);
    if (j == 0) gb[f] = 0.0f;
}

// ---------------------------------------------------------------------------------------------------------------------
// The T x 256 passes, one wave per block of 32 rows: eight waves per workgroup, one workgroup per compute unit, W2's
// fragments in LDS (64 KiB per orientation) - so a wave keeps its rows' layer-1 activations in REGISTERS (they are its own
// MFMA output), shares nothing with the other waves and meets them at no barrier after the set-up.  Per feature block fb of
// 32: 12 products recompute x2, then
//   MODE 0  column sums of the raw accumulators and their squares    -> partial[wave][2][256] double
//   MODE 1  z = silu((x2 - mean) gamma rstd + beta), max|z|           -> z
//   MODE 2  sums of gz = g_z silu'(.) and gz xhat, maxima             -> partial[wave][2][256] double
//   MODE 3  da1 += dx2 W2 (x2 recomputed with the operands swapped: lane = row, registers = features, which IS the A
//           operand of that product; a wave owns all 256 features of its rows, so da1 needs no reduction across waves)
// Per-feature constants folded:  zl = acc A + B,  xhat = acc C + D,  sd dx2 = E gz + acc F + G (Dx2Const).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kRbGrid = 256, kRbThreadsMax = 512;
// waves per workgroup (= per compute unit) by pass
// (measured before the statistics pass also took a1's moments: that pass 96-99 us with twelve waves, 101-107 with eight; the z
// pass 143 us with eight and its feature loop unrolled, 151-158 with eight / twelve / sixteen waves and the loop rolled to fit
// them - occupancy is not their limit)
constexpr int rb_threads(int mode) { return mode >= 0 ? 512 : 0; }
__device__ __forceinline__ int w2a_idx(int fb, int s2, int hl, int lane) { return ((fb * 4 + s2) * 2 + hl) * 64 + lane; }
__device__ __forceinline__ int w2b_idx(int fb, int s, int jb, int hl, int lane) { return ((((fb * 2 + s) * 2 + jb) * 2) + hl) * 64 + lane; }

template <int MODE>
__global__ __launch_bounds__(rb_threads(MODE), 1) void angle_rb_kernel(P p) {
    constexpr int kRbThreads = rb_threads(MODE);
    __shared__ L1Shared sh;
    __shared__ uint4 w2a[8 * 4 * 2 * 64];                         // 64 KiB: lane = feature 32 fb + il, slots = permuted inputs
    __shared__ uint4 w2b[MODE == 3 ? 8 * 2 * 2 * 2 * 64 : 1];     // 64 KiB: lane = j 32 jb + il, slots = permuted features
    __shared__ float cst[MODE == 0 ? 1 : 5][kH];
    // MODE 0 also takes the first two moments of a1 (sum and sum of outer products, 64 + 64 x 64 per workgroup): what the
    // backward needs to get dW2 without knowing BatchNorm's backward sums first (angle_sums_dw2_kernel)
    // (its cross-wave reduction at the end reuses w2a's 64 KiB: 4 x 50 x 64 floats)
    __shared__ double col_sums[MODE == 0 ? kRbThreads / 64 : 1][MODE == 0 ? 2 * kH : 1];  // MODE 0: per wave, 4 KiB each
    float (*mom_red)[50][64] = reinterpret_cast<float (*)[50][64]>(w2a);
    const float sa = f16_scale(p.scal[kBoundA1]), sw = f16_scale(p.scal[kAmaxW2]);
    const float sd = MODE == 3 ? f16_scale(p.scal[kBoundDx2]) : 1.0f;
    const float inv2 = 1.0f / (sa * sw), neg_k = -kLog2e / sa, g2 = -p.gamma * kLog2e;
    const int lane = threadIdx.x & 63, il = lane & 31, hh = lane >> 5, w = threadIdx.x >> 6;
    if (w < 8) {  // wave w builds feature block w of both orientations
        const int fb = w;
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            const float* wp = p.W2 + (size_t)(32 * fb + il) * kE + 16 * s2 + 4 * hh;
            const float4 a = f4_ld(wp), b = f4_ld(wp + 8);
            const float v[8] = {a.x * sw, a.y * sw, a.z * sw, a.w * sw, b.x * sw, b.y * sw, b.z * sw, b.w * sw};
            f16x8 hi, lo;
            split8s(v, hi, lo);
            w2a[w2a_idx(fb, s2, 0, lane)] = __builtin_bit_cast(uint4, hi);
            w2a[w2a_idx(fb, s2, 1, lane)] = __builtin_bit_cast(uint4, lo);
        }
        if constexpr (MODE == 3) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int f = 32 * fb + 16 * s + 8 * (i >> 2) + 4 * hh + (i & 3);
                        v[i] = p.W2[(size_t)f * kE + 32 * jb + il] * sw;
                    }
                    f16x8 hi, lo;
                    split8s(v, hi, lo);
                    w2b[w2b_idx(fb, s, jb, 0, lane)] = __builtin_bit_cast(uint4, hi);
                    w2b[w2b_idx(fb, s, jb, 1, lane)] = __builtin_bit_cast(uint4, lo);
                }
        }
    }
    if constexpr (MODE != 0) {
        for (int f = threadIdx.x; f < kH; f += kRbThreads) {
            const float d = p.b2[f] - p.stat2[f], sc = p.stat2[2 * kH + f];
            cst[0][f] = inv2 * sc;                           // A
            cst[1][f] = fmaf(d, sc, p.stat2[3 * kH + f]);    // B
            if constexpr (MODE == 3) {
                Dx2Const k;
                k.load(p, f, inv2, sd);
                cst[2][f] = k.E, cst[3][f] = k.F, cst[4][f] = k.G;
            }
        }
    }
    l1_setup(p, sh, kL1Act, sa);  // (ends with the barrier that also publishes w2a / w2b / cst)
    const int wave = blockIdx.x * (kRbThreads / 64) + w, nwaves = gridDim.x * (kRbThreads / 64);
    float am0 = 0.0f;
    if constexpr (MODE == 0)
        for (int i = lane; i < 2 * kH; i += 64) col_sums[w][i] = 0.0;  // (a wave's own block: no barrier needed)
    f32x16 m2[2][2];       // MODE 0: sum over rows of (sa a1)[j] (sa a1)[j'] - D[m = j][n = j']: lane = j' 32 jb2 + il, registers = j of block jb
    float m1[2] = {0.0f, 0.0f};  // MODE 0: sum over rows of (sa a1)[j], j = 32 jb + il
    f16x8 idp0, idp1;
    if constexpr (MODE == 0) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) m2[a][b] = zero16();
        idp0 = identity_frag<true>(0, il, hh), idp1 = identity_frag<true>(1, il, hh);
    }
    const float invd = MODE == 3 ? 1.0f / (sd * sw) : 1.0f;
    const int64_t nblk = (p.rows + 31) / 32;
    float hv_next = load_h(p, (int64_t)wave * 32 + il);
    for (int64_t blk = wave; blk < nblk; blk += nwaves) {
        const int64_t row0 = blk * 32;
        const bool full = row0 + 32 <= p.rows;  // uniform
        const float hv = hv_next;
        hv_next = load_h(p, (blk + nwaves) * 32 + il);
        // ---- layer 1 -> a1 fragments (registers): step s2 = 2 cb + s, slots = permuted features (see phase_a)
        f16x8 a_hi[4], a_lo[4];
        {
            f32x16 acc1[2];
            {
                f16x8 r_hi[3], r_lo[3];
                l1_rbf<false>(sh, g2, hv, hh, 0, r_hi, r_lo);
                l1_product(sh, lane, r_hi, r_lo, acc1);
            }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float a[8];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        float u[4], v[4];
                        l1_const4(sh, 0, cb, 2 * s + q, hh, u);
                        l1_const4(sh, 1, cb, 2 * s + q, hh, v);
#pragma unroll
                        for (int e = 0; e < 4; ++e) a[4 * q + e] = silu_scaled(fmaf(acc1[cb][8 * s + 4 * q + e], u[e], v[e]), neg_k);
                    }
                    if constexpr (MODE == 0) {  // rows past the end must not reach the moments (their x2 is masked anyway)
                        const float okf = row0 + il < p.rows ? 1.0f : 0.0f;
#pragma unroll
                        for (int i = 0; i < 8; ++i) a[i] *= okf;
                    }
                    split8s(a, a_hi[2 * cb + s], a_lo[2 * cb + s]);
                }
        }
        if constexpr (MODE == 0) {
            // a1 turned to (lane = j, registers = rows) by identity products, then  m2 += a1^T a1,  m1 += column sums
            f16x8 t_hi[2][2], t_lo[2][2];
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                f32x16 th = mfma(a_hi[2 * jb], idp0, zero16());
                th = mfma(a_hi[2 * jb + 1], idp1, th);
                f32x16 tl = mfma(a_lo[2 * jb], idp0, zero16());
                tl = mfma(a_lo[2 * jb + 1], idp1, tl);
                float cs = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) cs += th[r] + tl[r];
                m1[jb] += cs;
#pragma unroll
                for (int s = 0; s < 2; ++s) t_hi[jb][s] = pack8(th, s), t_lo[jb][s] = pack8(tl, s);
            }
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int jb2 = jb; jb2 < 2; ++jb2)  // (symmetric: block (1, 0) is the transpose of (0, 1))
#pragma unroll
                    for (int s = 0; s < 2; ++s) m2[jb][jb2] = mfma3(t_hi[jb][s], t_lo[jb][s], t_hi[jb2][s], t_lo[jb2][s], m2[jb][jb2]);
            __builtin_amdgcn_sched_barrier(0);  // (keep the feature-block loop's operands out of this block's registers)
        }
        // ---- the eight feature blocks
        auto run = [&](auto full_c) {
            constexpr bool FULL = decltype(full_c)::value;
            f32x16 da[2] = {zero16(), zero16()};
            const float okf = FULL || row0 + il < p.rows ? 1.0f : 0.0f;  // MODE 3: this lane's row
            int64_t grow3 = row0 + il;
            grow3 = grow3 < p.rows ? grow3 : p.rows - 1;
            // g_z of feature block fb (MODE 3): 16 features of row row0 + il
            auto load_g = [&](int fb, float (&g)[16]) {
                if constexpr (MODE == 3) {
                    const float* gp = p.gz + grow3 * kH + 32 * fb + 4 * hh;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = f4_ld(gp + 8 * q);  // (cached: the four quads and two halves of a row share lines)
                        g[4 * q] = v.x, g[4 * q + 1] = v.y, g[4 * q + 2] = v.z, g[4 * q + 3] = v.w;
                    }
                }
            };
            auto block = [&](int fb, const float (&g)[16]) {
                f32x16 acc = zero16();
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    const f16x8 wh = __builtin_bit_cast(f16x8, w2a[w2a_idx(fb, s2, 0, lane)]);
                    const f16x8 wl = __builtin_bit_cast(f16x8, w2a[w2a_idx(fb, s2, 1, lane)]);
                    if constexpr (MODE == 3)
                        acc = mfma3(wh, wl, a_hi[s2], a_lo[s2], acc);  // D[m = feature][n = row]
                    else
                        acc = mfma3(a_hi[s2], a_lo[s2], wh, wl, acc);  // D[m = row][n = feature]
                }
                if constexpr (MODE == 3) {
                    f16x8 d_hi[2], d_lo[2];
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        float dx[8];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int f = 32 * fb + 8 * (2 * s + q) + 4 * hh;
                            const float4 A4 = f4_ld(&cst[0][f]), B4 = f4_ld(&cst[1][f]), E4 = f4_ld(&cst[2][f]);
                            const float4 F4 = f4_ld(&cst[3][f]), G4 = f4_ld(&cst[4][f]);
                            const float cA[4] = {A4.x, A4.y, A4.z, A4.w}, cB[4] = {B4.x, B4.y, B4.z, B4.w};
                            const float cE[4] = {E4.x, E4.y, E4.z, E4.w}, cF[4] = {F4.x, F4.y, F4.z, F4.w}, cG[4] = {G4.x, G4.y, G4.z, G4.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int r = 8 * s + 4 * q + e;
                                const float a = acc[r];
                                const float gz = g[r] * dsilu_fast(fmaf(a, cA[e], cB[e]));
                                float o = fmaf(cE[e], gz, fmaf(a, cF[e], cG[e]));
                                if constexpr (!FULL) o *= okf;  // (a multiply, not a branch)
                                dx[4 * q + e] = o;
                            }
                        }
                        split8s(dx, d_hi[s], d_lo[s]);
                    }
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int jb = 0; jb < 2; ++jb) {
                            const f16x8 bh = __builtin_bit_cast(f16x8, w2b[w2b_idx(fb, s, jb, 0, lane)]);
                            const f16x8 bl = __builtin_bit_cast(f16x8, w2b[w2b_idx(fb, s, jb, 1, lane)]);
                            da[jb] = mfma3(d_hi[s], d_lo[s], bh, bl, da[jb]);
                        }
                } else {
                    const int f = 32 * fb + il;
                    float cA = 0.0f, cB = 0.0f;
                    if constexpr (MODE != 0) cA = cst[0][f], cB = cst[1][f];
                    float s = 0.0f, q = 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t row = row0 + n_row(0, r, hh);
                        const float m = FULL || row < p.rows ? 1.0f : 0.0f;
                        if constexpr (MODE == 0) {
                            const float a = FULL ? acc[r] : acc[r] * m;
                            s += a;
                            q = fmaf(a, a, q);
                        } else {
                            const float zl = fmaf(acc[r], cA, cB);
                            const float zz = silu_scaled(zl, -kLog2e);
                            if (FULL || row < p.rows) __builtin_nontemporal_store(zz, p.z + row * kH + f);
                            am0 = fmaxf(am0, FULL ? fabsf(zz) : fabsf(zz) * m);
                        }
                    }
                    if constexpr (MODE == 0) {  // float64 column sums of this wave in its LDS block (the two halves hold
                        s += __shfl_xor(s, 32, 64);  // different rows of feature f)
                        q += __shfl_xor(q, 32, 64);
                        if (hh == 0) col_sums[w][f] += (double)s, col_sums[w][kH + f] += (double)q;
                    }
                }
            };
            float g0[16], g1[16];
            if constexpr (MODE == 0) {  // (nothing indexed by fb lives in registers: rolled, the moments need the registers)
#pragma unroll 1
                for (int fb = 0; fb < 8; ++fb) block(fb, g0);
                return;
            }
            load_g(0, g0);
#pragma unroll
            for (int fb = 0; fb < 8; fb += 2) {  // g_z of the next feature block in flight under the current one
                load_g(fb + 1, g1);
                block(fb, g0);
                if (fb < 6) load_g(fb + 2, g0);
                block(fb + 1, g1);
            }
            if constexpr (MODE == 3) {  // da[jb][r]: row = row0 + 8 (r >> 2) + 4 hh + (r & 3), j = 32 jb + il
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t orow = row0 + n_row(0, r, hh);
                        if (FULL || orow < p.rows) p.da1[orow * kE + 32 * jb + il] = da[jb][r] * invd;
                    }
            }
        };
        if (full)
            run(std::true_type{});
        else
            run(std::false_type{});
    }
    if constexpr (MODE == 0) {
        double* out = static_cast<double*>(p.partial) + (size_t)wave * 2 * kH;
        for (int i = lane; i < 2 * kH; i += 64) out[i] = col_sums[w][i];
        // moments: the eight waves' 48 + 2 registers added through LDS in a fixed order (4..7 onto 0..3, 2..3 onto 0..1, 1 onto 0)
        m1[0] += __shfl_xor(m1[0], 32, 64);  // (the two halves of a wave hold different rows of the same j)
        m1[1] += __shfl_xor(m1[1], 32, 64);
        auto put = [&](int slot) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = a; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mom_red[slot][(a + b) * 16 + r][lane] = m2[a][b][r];
            mom_red[slot][48][lane] = m1[0];
            mom_red[slot][49][lane] = m1[1];
        };
        auto take = [&](int slot) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = a; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) m2[a][b][r] += mom_red[slot][(a + b) * 16 + r][lane];
            m1[0] += mom_red[slot][48][lane];
            m1[1] += mom_red[slot][49][lane];
        };
        static_assert(rb_threads(0) == 512, "the reduction below is written for eight waves");
        for (int span = 4; span >= 1; span >>= 1) {
            __syncthreads();
            if (w >= span && w < 2 * span) put(w - span);
            __syncthreads();
            if (w < span) take(w);
        }
        if (w == 0) {  // slab of this workgroup: [64] sums | [64][64] outer products (row = j, column = j')
            float* mo = p.partial_b + (size_t)blockIdx.x * (kE + kE * kE);
            if (hh == 0) {
                mo[il] = m1[0];
                mo[32 + il] = m1[1];
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = a; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int j = 32 * a + 8 * (r >> 2) + 4 * hh + (r & 3), j2 = 32 * b + il;
                        mo[kE + j * kE + j2] = m2[a][b][r];
                        if (a != b) mo[kE + j2 * kE + j] = m2[a][b][r];  // the transposed block
                    }
        }
    }
    if constexpr (MODE == 1) block_amax_commit(am0, p.z_amax);
}

// ---------------------------------------------------------------------------------------------------------------------
// passes 7, 8 (one wave per block of 32 rows, nothing shared): gz1 = da1 silu'(z1), then
//   MODE 0  sums of gz1 and gz1 xhat1, maxima                        -> partial[wave][2][64] double
//   MODE 1  dx1 = gamma rstd (gz1 - (c0 + xhat c1) / n);  dW1[c][k] = sum_rows dx1[row][c] rbf[row][k]: both operands turned to
//           (registers = rows) by identity products; the expansion carries a column of ones in slot `bins`, so column `bins`
//           of the product is db1                                      -> partial[wave][64][bins + 1] floats
// ---------------------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kThreads, 2) void angle_l1_bwd_kernel(P p) {
    __shared__ L1Shared sh;
    const float sx = MODE == 1 ? f16_scale(p.scal[kBoundDx1]) : 1.0f;
    l1_setup(p, sh, kL1Bwd, sx);
    const int lane = threadIdx.x & 63, il = lane & 31, hh = lane >> 5;
    const int wave = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6), nwaves = gridDim.x * (kThreads / 64);
    const float g2 = -p.gamma * kLog2e;
    float s0[2][16], s1[2][16];
    float am0 = 0.0f, am1 = 0.0f;
    f32x16 dw[2][2];
    f16x8 ip0, ip1, in0, in1;
    if constexpr (MODE == 0) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s0[cb][r] = s1[cb][r] = 0.0f;
    } else {
        ip0 = identity_frag<true>(0, il, hh), ip1 = identity_frag<true>(1, il, hh);
        in0 = identity_frag<false>(0, il, hh), in1 = identity_frag<false>(1, il, hh);
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) dw[jb][nb] = zero16();
    }
    const int64_t nblk = (p.rows + 31) / 32;
    auto load_d = [&](int64_t blk, float4 (&g)[2][4]) {
        int64_t row = blk * 32 + il;
        row = row < p.rows ? row : p.rows - 1;
        const float* drow = p.da1 + row * kE + 4 * hh;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) g[cb][q] = f4_ld(drow + 32 * cb + 8 * q);
    };
    float hv_next = load_h(p, (int64_t)wave * 32 + il);
    float4 g4[2][4], g4n[2][4];
    if constexpr (MODE == 0) load_d(wave, g4n);
    for (int64_t blk = wave; blk < nblk; blk += nwaves) {
        const int64_t row = blk * 32 + il;
        const float okf = row < p.rows ? 1.0f : 0.0f;
        const float hv = hv_next;
        hv_next = load_h(p, (blk + nwaves) * 32 + il);
        if constexpr (MODE == 0) {  // (the weight-gradient pass has no registers left for a second set)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) g4[cb][q] = g4n[cb][q];
            load_d(blk + nwaves, g4n);
        } else
            load_d(blk, g4);
        f16x8 r_hi[3], r_lo[3];
        l1_rbf<MODE == 1>(sh, g2, hv, hh, p.bins, r_hi, r_lo);
        f32x16 acc[2];
        l1_product(sh, lane, r_hi, r_lo, acc);
        f16x8 b_hi[2][2], b_lo[2][2];
        if constexpr (MODE == 1) {
            // rbf (lane = row, slots = bins in natural order) -> (lane = bin 32 nb + il, registers = rows)
            f32x16 th = mfma(r_hi[0], in0, zero16());
            th = mfma(r_hi[1], in1, th);
            f32x16 tl = mfma(r_lo[0], in0, zero16());
            tl = mfma(r_lo[1], in1, tl);
#pragma unroll
            for (int s = 0; s < 2; ++s) b_hi[0][s] = pack8(th, s), b_lo[0][s] = pack8(tl, s);
            th = mfma(r_hi[2], in0, zero16());
            tl = mfma(r_lo[2], in0, zero16());
#pragma unroll
            for (int s = 0; s < 2; ++s) b_hi[1][s] = pack8(th, s), b_lo[1][s] = pack8(tl, s);
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            f16x8 x_hi[2], x_lo[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float dx[8];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int qq = 2 * s + q;
                    const float4 gq = g4[cb][qq];
                    const float gg[4] = {gq.x, gq.y, gq.z, gq.w};
                    float ux[4], vx[4], ga[4], be[4];
                    l1_const4(sh, 0, cb, qq, hh, ux);
                    l1_const4(sh, 1, cb, qq, hh, vx);
                    l1_const4(sh, 2, cb, qq, hh, ga);
                    l1_const4(sh, 3, cb, qq, hh, be);
                    float c4[4], c5[4], c6[4];
                    if constexpr (MODE == 1) {
                        l1_const4(sh, 4, cb, qq, hh, c4);
                        l1_const4(sh, 5, cb, qq, hh, c5);
                        l1_const4(sh, 6, cb, qq, hh, c6);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * qq + e;
                        const float xh = fmaf(acc[cb][r], ux[e], vx[e]);
                        const float gz = gg[e] * dsilu_fast(fmaf(xh, ga[e], be[e])) * okf;
                        if constexpr (MODE == 0) {
                            s0[cb][r] += gz;
                            s1[cb][r] = fmaf(gz, xh, s1[cb][r]);
                            am0 = fmaxf(am0, fabsf(gz));
                            am1 = fmaxf(am1, fabsf(xh) * okf);
                        } else
                            dx[4 * q + e] = fmaf(c4[e], gz, fmaf(xh, c5[e], c6[e])) * okf;
                    }
                }
                if constexpr (MODE == 1) split8s(dx, x_hi[s], x_lo[s]);
            }
            if constexpr (MODE == 1) {
                // dx1 (lane = row, slots = features of block cb) -> (lane = feature 32 cb + il, registers = rows)
                f32x16 th = mfma(x_hi[0], ip0, zero16());
                th = mfma(x_hi[1], ip1, th);
                f32x16 tl = mfma(x_lo[0], ip0, zero16());
                tl = mfma(x_lo[1], ip1, tl);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const f16x8 ah = pack8(th, s), al = pack8(tl, s);
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) dw[cb][nb] = mfma3(ah, al, b_hi[nb][s], b_lo[nb][s], dw[cb][nb]);
                }
            }
        }
    }
    if constexpr (MODE == 0) {
        double* out = static_cast<double*>(p.partial) + (size_t)wave * 2 * kE;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float a = half_sum(s0[cb][r]), b = half_sum(s1[cb][r]);
                if (il == 0) {
                    const int col = l1_col(cb, r, hh);
                    out[col] = (double)a;
                    out[kE + col] = (double)b;
                }
            }
        block_amax_commit(am0, p.scal + kAmaxGz1);
        block_amax_commit(am1, p.scal + kAmaxXh1);
    } else {
        // dw[jb][nb]: D[m = feature][n = bin]: lane = bin 32 nb + il, register r = feature 32 jb + 8 (r >> 2) + 4 hh + (r & 3)
        const int ld = p.bins + 1;
        float* out = static_cast<float*>(p.partial) + (size_t)wave * kE * ld;
        const float inv = 1.0f / (sx * kRbfScale);
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const int bin = 32 * nb + il;
                if (bin < ld) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) out[(32 * jb + 8 * (r >> 2) + 4 * hh + (r & 3)) * ld + bin] = dw[jb][nb][r] * inv;
                }
            }
    }
}

// dW1 | db1 from the reduced [64][bins + 1] product
__global__ void angle_unpack_dw1_kernel(const float* __restrict__ src, int bins, float* __restrict__ gW, float* __restrict__ gb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= kE * (bins + 1)) return;
    const int c = i / (bins + 1), k = i - c * (bins + 1);
    if (k < bins) gW[c * bins + k] = src[i];
    else gb[c] = src[i];
}

inline int grid_for(int64_t rows, int rows_per_block) {
    const int64_t need = (rows + rows_per_block - 1) / rows_per_block;
    return (int)(need < kGrid ? (need > 0 ? need : 1) : kGrid);
}
inline size_t al256(size_t b) { return (b + 255) / 256 * 256; }
// The T x 256 passes take whole compute units (512 threads, 64-142 KiB of LDS, up to 250 registers per lane): what the caller's
// other streams launch beside them waits for a unit they do not use.  ALIGNN_AMD_ANGLE_GRID (default kAngleGridDefault, a
// multiple of 8) caps their grids; every pass walks its tiles with a grid stride and hands the launched count to its reductions.
inline int angle_grid_cap() {
    static const int cap = [] {
        const char* e = getenv("ALIGNN_AMD_ANGLE_GRID");
        int v = e ? atoi(e) : ANGLE_GRID_DEFAULT;
        v = v < 8 ? 8 : (v > kRbGrid ? kRbGrid : v);
        return v & ~7;
    }();
    return cap;
}
inline int rb_grid(int64_t rows, int mode) {
    const int g = grid_for(rows, 32 * (rb_threads(mode) / 64));
    const int cap = angle_grid_cap();
    return g < cap ? g : cap;
}

P make_params(const alignn_angle_args& a) {
    P p{};
    p.h = a.h;
    p.rows = a.rows;
    p.centers = a.centers;
    p.gamma = a.gamma;
    p.bins = a.bins;
    p.W1 = a.l1.W, p.b1 = a.l1.b, p.gamma1 = a.l1.gamma, p.beta1 = a.l1.beta;
    p.W2 = a.l2.W, p.b2 = a.l2.b;
    p.stat1 = a.stat1, p.stat2 = a.stat2;
    p.scal = a.scal;
    p.z = a.z, p.z_amax = a.z_amax;
    p.gz = a.gz;
    p.red1 = a.l1.red, p.red2 = a.l2.red;
    return p;
}

bool shape_ok(const alignn_angle_args& a) {
    return a.rows > 0 && a.bins > 0 && a.bins < kBinsMax && a.l1.in == a.bins && a.l1.out == kE && a.l2.in == kE && a.l2.out == kH;
}

}  // namespace

extern "C" {

size_t alignn_angle_args_sizeof(void) { return sizeof(alignn_angle_args); }
int alignn_angle_embed_scal_floats(void) { return kScalFloats; }

int alignn_angle_embed_supported(int bins, int embed, int hidden) { return bins > 0 && bins < kBinsMax && embed == kE && hidden == kH; }

size_t alignn_angle_embed_workspace(int64_t rows, int bins, int backward) {
    (void)rows;
    const size_t waves = (size_t)kGrid * (kThreads / 64);
    const size_t sums = al256((size_t)kRbGrid * (kRbThreadsMax / 64) * 2 * kH * sizeof(double));  // one slab per wave
    if (!backward) {
        const size_t a = al256(waves * 2 * kE * sizeof(float));
        return (a > sums ? a : sums) + al256((size_t)(kRbGrid + 1) * (kE + kE * kE) * sizeof(float));  // + the moment slabs of a1
    }
    size_t total = al256((size_t)rows * kE * sizeof(float));                       // da1
    total += sums;                                                                  // sums (layer 2, then layer 1: waves * 2 * 64)
    total += al256((size_t)kGrid * kH * kE * sizeof(float));                        // dW2 slabs
    total += al256((size_t)kGrid * kH * sizeof(float));                             // db2 slabs
    total += al256((size_t)alignn_slab_fold_slabs() * kH * kE * sizeof(float));     // folded dW2 slabs
    total += al256(waves * kE * (size_t)(bins + 1) * sizeof(float));                // dW1 | db1 slabs
    total += 2 * al256((size_t)alignn_slab_fold_slabs() * kE * (size_t)(bins + 1) * sizeof(float));
    return total;
}

// forward: z [rows, 256] (+ max|z|), stat1 / stat2 / scal kept for the backward, running statistics updated
int alignn_angle_embed_fwd(const alignn_angle_args* a, alignn_stream_t stream) {
    if (a == nullptr || !shape_ok(*a) || !a->h || !a->centers || !a->stat1 || !a->stat2 || !a->scal || !a->z || !a->workspace)
        return (int)hipErrorInvalidValue;
    if (a->workspace_bytes < alignn_angle_embed_workspace(a->rows, a->bins, 0)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    P p = make_params(*a);
    p.partial = a->workspace;
    p.partial_b = reinterpret_cast<float*>(static_cast<char*>(a->workspace) + a->workspace_bytes -
                                           al256((size_t)(kRbGrid + 1) * (kE + kE * kE) * sizeof(float)));  // moment slabs (at the end)
    if (a->workspace_bytes != alignn_angle_embed_workspace(a->rows, a->bins, 0)) {  // (any larger block: keep the layout of the exact one)
        p.partial_b = reinterpret_cast<float*>(static_cast<char*>(a->workspace) + alignn_angle_embed_workspace(a->rows, a->bins, 0) -
                                               al256((size_t)(kRbGrid + 1) * (kE + kE * kE) * sizeof(float)));
    }
    hipLaunchKernelGGL(angle_zero_kernel, dim3(1), dim3(kScalHead), 0, st, a->scal, kScalHead);
    hipLaunchKernelGGL(angle_prep_kernel, dim3(17), dim3(kThreads), 0, st, p);
    const int g1 = grid_for(a->rows, 32 * (kThreads / 64)), g2 = grid_for(a->rows, kTile);
    hipLaunchKernelGGL(angle_l1_stats_kernel, dim3(g1), dim3(kThreads), 0, st, p);
    hipLaunchKernelGGL(angle_stat_finalize_kernel<false>, dim3(kE / 4), dim3(256), 0, st, (const void*)p.partial, g1 * (kThreads / 64),
                       a->rows, kE, a->l1.b, a->l1.gamma, a->l1.beta, a->eps, a->momentum, a->l1.rm, a->l1.rv, a->stat1, a->scal);
    const int grb = rb_grid(a->rows, 0);
    const int rb_slabs = grb * (rb_threads(0) / 64);
    (void)g2;
    hipLaunchKernelGGL(angle_rb_kernel<0>, dim3(grb), dim3(rb_threads(0)), 0, st, p);
    {   // the workgroups' moment slabs -> one [64 + 64 x 64] block behind them -> mean and covariance
        float* msum = p.partial_b + (size_t)kRbGrid * (kE + kE * kE);
        const int rc = alignn_slab_sum(p.partial_b, grb, kE + kE * kE, msum, stream);
        if (rc != 0) return rc;
        hipLaunchKernelGGL(angle_moments_finalize_kernel, dim3(kE * kE / 256), dim3(256), 0, st, (const float*)msum, a->rows, a->scal);
    }
    hipLaunchKernelGGL(angle_stat_finalize_kernel<true>, dim3(kH / 4), dim3(256), 0, st, (const void*)p.partial, rb_slabs, a->rows, kH,
                       a->l2.b, a->l2.gamma, a->l2.beta, a->eps, a->momentum, a->l2.rm, a->l2.rv, a->stat2, a->scal);
    hipLaunchKernelGGL(angle_rb_kernel<1>, dim3(rb_grid(a->rows, 1)), dim3(rb_threads(1)), 0, st, p);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

// evaluation mode (BatchNorm = the affine map of its running statistics, nothing kept): one pass, z only
int alignn_angle_embed_infer(const alignn_angle_args* a, alignn_stream_t stream) {
    if (a == nullptr || !shape_ok(*a) || !a->h || !a->centers || !a->stat1 || !a->stat2 || !a->scal || !a->z || !a->l1.rm || !a->l1.rv ||
        !a->l2.rm || !a->l2.rv)
        return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    P p = make_params(*a);
    hipLaunchKernelGGL(angle_zero_kernel, dim3(1), dim3(kScalHead), 0, st, a->scal, kScalHead);
    hipLaunchKernelGGL(angle_prep_kernel, dim3(17), dim3(kThreads), 0, st, p);
    ALIGNN_CHECK_LAUNCH();
    int rc;
    if ((rc = alignn_bn_finalize(nullptr, 0, a->rows, kE, a->l1.gamma, a->l1.beta, a->eps, a->momentum, a->l1.rm, a->l1.rv, a->stat1, stream)) != 0)
        return rc;
    if ((rc = alignn_bn_finalize(nullptr, 0, a->rows, kH, a->l2.gamma, a->l2.beta, a->eps, a->momentum, a->l2.rm, a->l2.rv, a->stat2, stream)) != 0)
        return rc;
    hipLaunchKernelGGL(angle_infer_bound_kernel, dim3(1), dim3(kE), 0, st, p);
    hipLaunchKernelGGL(angle_rb_kernel<1>, dim3(rb_grid(a->rows, 1)), dim3(rb_threads(1)), 0, st, p);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

// backward: gradients of both layers' parameters (gW, gb, red = dbeta | dgamma) from g_z [rows, 256]; `side` (optional):
// the stream the slab reductions of the weight gradients go to (the caller orders it before and after)
int alignn_angle_embed_bwd(const alignn_angle_args* a, alignn_stream_t stream) {
    if (a == nullptr || !shape_ok(*a) || !a->h || !a->centers || !a->stat1 || !a->stat2 || !a->scal || !a->gz || !a->workspace ||
        !a->l1.gW || !a->l1.gb || !a->l1.red || !a->l2.gW || !a->l2.gb || !a->l2.red)
        return (int)hipErrorInvalidValue;
    if (a->workspace_bytes < alignn_angle_embed_workspace(a->rows, a->bins, 1)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const int waves_per = kThreads / 64;
    const size_t waves = (size_t)kGrid * waves_per;
    char* ws = static_cast<char*>(a->workspace);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void* q = ws + off;
        off += al256(bytes);
        return q;
    };
    float* da1 = static_cast<float*>(take((size_t)a->rows * kE * sizeof(float)));
    double* sums = static_cast<double*>(take((size_t)kRbGrid * (kRbThreadsMax / 64) * 2 * kH * sizeof(double)));
    float* dw2 = static_cast<float*>(take((size_t)kGrid * kH * kE * sizeof(float)));
    (void)take((size_t)kGrid * kH * sizeof(float));  // (db2 slabs of the earlier form: the layout stays)
    float* dw2f = static_cast<float*>(take((size_t)alignn_slab_fold_slabs() * kH * kE * sizeof(float)));
    const int ld1 = a->bins + 1;
    float* dw1 = static_cast<float*>(take(waves * kE * (size_t)ld1 * sizeof(float)));
    float* dw1f = static_cast<float*>(take((size_t)alignn_slab_fold_slabs() * kE * (size_t)ld1 * sizeof(float)));
    float* dw1s = static_cast<float*>(take((size_t)alignn_slab_fold_slabs() * kE * (size_t)ld1 * sizeof(float)));
    P p = make_params(*a);
    p.da1 = da1;
    const int g1 = grid_for(a->rows, 32 * waves_per);
    // layer 2: BatchNorm-backward sums and G = sum gz (a1 - mean) in one pass over g_z, then dW2 from G once c1 is known
    p.partial = dw2;
    p.partial_d = sums;
    const int gdw = grid_for(a->rows, kDwTile) < angle_grid_cap() ? grid_for(a->rows, kDwTile) : angle_grid_cap();
    hipLaunchKernelGGL(angle_sums_dw2_kernel, dim3(gdw), dim3(kDwThreads), 0, st, p);
    hipLaunchKernelGGL(angle_red_finalize_kernel, dim3(kH / 4), dim3(256), 0, st, (const double*)sums, gdw, a->rows, kH, a->stat2,
                       a->l2.red, a->scal, kAmaxGz2, kAmaxXh2, kBoundDx2);
    ALIGNN_CHECK_LAUNCH();
    int rc;
    if (gdw > alignn_slab_fold_slabs()) {
        if ((rc = alignn_slab_fold(dw2, gdw, kH * kE, dw2f, stream)) != 0) return rc;
        if ((rc = alignn_slab_sum(dw2f, alignn_slab_fold_slabs(), kH * kE, a->l2.gW, stream)) != 0) return rc;
    } else if ((rc = alignn_slab_sum(dw2, gdw, kH * kE, a->l2.gW, stream)) != 0)
        return rc;
    hipLaunchKernelGGL(angle_dw2_fixup_kernel, dim3(kH), dim3(kE), 0, st, p, a->l2.gW, a->l2.gb);
    hipLaunchKernelGGL(angle_rb_kernel<3>, dim3(rb_grid(a->rows, 3)), dim3(rb_threads(3)), 0, st, p);
    ALIGNN_CHECK_LAUNCH();
    // layer 1
    p.partial = sums;
    hipLaunchKernelGGL(angle_l1_bwd_kernel<0>, dim3(g1), dim3(kThreads), 0, st, p);
    hipLaunchKernelGGL(angle_red_finalize_kernel, dim3(kE / 4), dim3(256), 0, st, (const double*)sums, g1 * waves_per, a->rows, kE,
                       a->stat1, a->l1.red, a->scal, kAmaxGz1, kAmaxXh1, kBoundDx1);
    p.partial = dw1;
    hipLaunchKernelGGL(angle_l1_bwd_kernel<1>, dim3(g1), dim3(kThreads), 0, st, p);
    ALIGNN_CHECK_LAUNCH();
    const int slabs1 = g1 * waves_per, w1 = kE * ld1;
    if (slabs1 > alignn_slab_fold_slabs()) {
        if ((rc = alignn_slab_fold(dw1, slabs1, w1, dw1f, stream)) != 0) return rc;
        if ((rc = alignn_slab_sum(dw1f, alignn_slab_fold_slabs(), w1, dw1s, stream)) != 0) return rc;
    } else if ((rc = alignn_slab_sum(dw1, slabs1, w1, dw1s, stream)) != 0)
        return rc;
    hipLaunchKernelGGL(angle_unpack_dw1_kernel, dim3((w1 + 255) / 256), dim3(256), 0, st, (const float*)dw1s, (int)a->bins, a->l1.gW,
                       a->l1.gb);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
