// Line-graph convolutions of the LayerNorm flavour (ALIGNNAtomWise: alignn/models/alignn_atomwise.py:151-208) with the edge
// LayerNorm INSIDE the gate passes.  LayerNorm needs nothing from other rows, and in the gate passes of conv.hip / dual.hip a
// wavefront already holds a whole 256-feature row of m (lane l: features [4l, 4l + 4)), so
//
//   forward   y' = y + silu(LN(m))                    is written by the pass that takes the two segment sums of sigma(m),
//   reverse   the LayerNorm / SiLU adjoint of g_y'    is formed in the pass that adds the gate's adjoint to it,
//
// for values and for duals (value + directional derivative: the force-training pass, dual.hip).  Per T-row tensor and
// convolution that removes: forward 1 read of m; reverse 1 write + 1 read of the branch gradient and 1 read of m; the same
// twice over on duals.  Arithmetic and summation orders are those of the separate kernels (norm.hip ln_silu_*, conv.hip
// egc_gate_fwd / egc_bwd_lg_dense, dual.hip ln_silu_dual_* / egc_gate_dual_tan / egc_dual_bwd_lg_dense); the LayerNorm
// parameter gradients leave as one [2][H] slab per workgroup, summed in slab order by alignn_bn_bwd_finalize.
//
// H <= 256 (one feature panel per wave: the row statistics are wave reductions) - alignn_egc_ln_fused_supported.
//
// NOTE: this file is compiled with -fno-slp-vectorize like norm.hip / dual.hip (alignn_amd/build.py).
#include <cstdlib>

#include "common.h"
#include "../../include/alignn_hip.h"

namespace {

constexpr int kW = 4, kT = kW * ALIGNN_WAVE, kMaxBlocks = 1024, kK = 4;

// Sum over the 64 lanes of a wavefront, the result in every lane.  Within a row of 16 lanes by DPP adds (xor 1, xor 2, mirror
// of 8, mirror of 16: every lane of the row ends with the row's sum), the four rows by v_readlane - 4 DPP adds + 4 readlanes + 3
// adds instead of the 6 ds_bpermute round trips of the __shfl_xor butterfly (the LayerNorm passes below take 2-7 such sums per
// row).  Fixed order: ((r0 + r1) + (r2 + r3)) over the row sums.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum1(float v) {
    v += dpp_mov<0xB1>(v);   // quad_perm [1, 0, 3, 2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2, 3, 0, 1]
    v += dpp_mov<0x141>(v);  // row_half_mirror
    v += dpp_mov<0x140>(v);  // row_mirror
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}
template <int K>
__device__ __forceinline__ void wave_sum_k(float (&v)[K]) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = wave_sum1(v[k]);
}
__device__ __forceinline__ float hsum4(float4 a) { return (a.x + a.y) + (a.z + a.w); }
__device__ __forceinline__ float sig_f(float x) { return fast_sigmoid(x); }
__device__ __forceinline__ void dsilu2(float z, float& d1, float& d2) {  // silu'(z), silu''(z) (dual.hip)
    const float s = sig_f(z), sp = s * (1.0f - s);
    d1 = s + z * sp;
    d2 = sp * (2.0f + z * (1.0f - 2.0f * s));
}
inline int seg_blocks(int64_t n_seg) {
    int64_t b = (n_seg + kW - 1) / kW;
    if (b < 1) b = 1;
    if (b > kMaxBlocks) b = kMaxBlocks;
    return (int)b;
}
inline bool big_stream(int64_t rows, int H) { return rows * (int64_t)H * 4 >= (int64_t)128 << 20; }

// the workgroup's LayerNorm parameter-gradient slab: [0] = dbeta, [1] = dgamma, the four waves in wave order
__device__ __forceinline__ void ln_slab_store(float4 db, float4 dg, float4 (*sh)[kW][ALIGNN_WAVE], float* slab, int H, int f,
                                              bool active, int wave, int lane) {
    __syncthreads();
    sh[0][wave][lane] = db;
    sh[1][wave][lane] = dg;
    __syncthreads();
    if (wave == 0 && active) {
        float4 a = sh[0][0][lane], b = sh[1][0][lane];
#pragma unroll
        for (int w = 1; w < kW; ++w) {
            a = f4_add(a, sh[0][w][lane]);
            b = f4_add(b, sh[1][w][lane]);
        }
        f4_st(slab + f, a);
        f4_st(slab + H + f, b);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward: egc_gate_fwd_kernel<STREAM, false, PRE = true> (M holds m = A[u] + Bd[v] + C already) + ln_silu_fwd_kernel on m
// ---------------------------------------------------------------------------------------------------------------------
template <bool STREAM>
__global__ __launch_bounds__(kT) void egc_gate_fwd_ln_kernel(
    const float* __restrict__ P, const float* __restrict__ M, const int32_t* __restrict__ seg_ptr,
    const int32_t* __restrict__ seg_node, const int32_t* __restrict__ src, int n_seg, int H, float* __restrict__ XPRE,
    float* __restrict__ S0, float* __restrict__ HH, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    const float* __restrict__ Y, float* __restrict__ YOUT, float* __restrict__ e_stat, float* __restrict__ y_amax) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t ldp = 4 * (int64_t)H;
    const int first = blockIdx.x * kW + wave, stride = gridDim.x * kW;
    const int f = 4 * lane;
    const bool active = f < H;
    const float inv_f = 1.0f / (float)H;
    const float4 ga = active ? f4_ld(gamma + f) : f4_zero(), be = active ? f4_ld(beta + f) : f4_zero();
    float y_am = 0.0f;
    for (int s = first; s < n_seg; s += stride) {
        const int beg = seg_ptr[s], end = seg_ptr[s + 1];
        const int i = seg_node ? seg_node[s] : s;
        const float4 ux = active ? f4_ld(P + (int64_t)i * ldp + 3 * H + f) : f4_zero();
        float4 s1 = f4_zero(), s0 = f4_zero();
        for (int e = beg; e < end; e += kK) {
            float4 m[kK], bh[kK], yv[kK];
            float sm[kK], sv[kK];
#pragma unroll
            for (int k = 0; k < kK; ++k) {
                m[k] = bh[k] = yv[k] = f4_zero();
                if (e + k < end && active) {
                    bh[k] = f4_ld(P + (int64_t)src[e + k] * ldp + 2 * H + f);
                    m[k] = f4_lds<STREAM>(M + (int64_t)(e + k) * H + f);
                    if (Y) yv[k] = f4_lds<STREAM>(Y + (int64_t)(e + k) * H + f);
                }
                sm[k] = hsum4(m[k]);
            }
            wave_sum_k(sm);
            float4 d[kK];
#pragma unroll
            for (int k = 0; k < kK; ++k) {
                const float mean = sm[k] * inv_f;
                d[k] = active ? make_float4(m[k].x - mean, m[k].y - mean, m[k].z - mean, m[k].w - mean) : f4_zero();
                sv[k] = hsum4(f4_mul(d[k], d[k]));
            }
            wave_sum_k(sv);
#pragma unroll
            for (int k = 0; k < kK; ++k) {
                if (e + k < end) {  // (wave-uniform)
                    const float mean = sm[k] * inv_f, rstd = 1.0f / sqrtf(sv[k] * inv_f + eps);
                    if (active) {
                        float4 z;
                        z.x = d[k].x * rstd * ga.x + be.x;
                        z.y = d[k].y * rstd * ga.y + be.y;
                        z.z = d[k].z * rstd * ga.z + be.z;
                        z.w = d[k].w * rstd * ga.w + be.w;
                        float4 o = make_float4(silu_f(z.x), silu_f(z.y), silu_f(z.z), silu_f(z.w));
                        if (Y) o = f4_add(o, yv[k]);
                        f4_sts<STREAM>(YOUT + (int64_t)(e + k) * H + f, o);
                        y_am = fmaxf(y_am, f4_absmax(o));
                        const float4 sg = f4_sigmoid(m[k]);
                        s1 = f4_fma(sg, bh[k], s1);
                        s0 = f4_add(s0, sg);
                    }
                    if (lane == 0) {
                        e_stat[2 * (int64_t)(e + k)] = mean;
                        e_stat[2 * (int64_t)(e + k) + 1] = rstd;
                    }
                }
            }
        }
        if (active) {
            float4 h;
            h.x = s1.x / (s0.x + ALIGNN_EPS_GATE);
            h.y = s1.y / (s0.y + ALIGNN_EPS_GATE);
            h.z = s1.z / (s0.z + ALIGNN_EPS_GATE);
            h.w = s1.w / (s0.w + ALIGNN_EPS_GATE);
            f4_st(XPRE + (int64_t)i * H + f, f4_add(ux, h));
            if (S0) f4_st(S0 + (int64_t)i * H + f, s0);
            if (HH) f4_st(HH + (int64_t)i * H + f, h);
        }
    }
    block_amax_commit(y_am, y_amax);
}

// ---------------------------------------------------------------------------------------------------------------------
// reverse: ln_silu_bwd_kernel on (GY, M) + egc_bwd_lg_dense_kernel<2, STREAM> (dense, source-sorted line-graph blocks: the
// index arithmetic is explained there)
// ---------------------------------------------------------------------------------------------------------------------
template <bool STREAM, int KS, int KB>
__global__ __launch_bounds__(kT) void egc_bwd_lg_dense_ln_kernel(
    const float* __restrict__ GY, const float* __restrict__ M, const float* __restrict__ P, const float* __restrict__ GS1,
    const float* __restrict__ GS0, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ e_stat, const int32_t* __restrict__ grp_seg_ptr, const int32_t* __restrict__ grp_src_ptr,
    const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ seg_node, int H, float* __restrict__ GM,
    float* __restrict__ GP, float* __restrict__ gb_partial, float* __restrict__ ln_partial, float* __restrict__ gm_amax,
    float* __restrict__ gp_amax) {
    __shared__ float4 sh[2][kW][ALIGNN_WAVE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t ldp = 4 * (int64_t)H;
    const int j = blockIdx.x;
    const int p_beg = grp_src_ptr[j], n_src = grp_src_ptr[j + 1] - p_beg;
    const int s_beg = grp_seg_ptr[j], s_end = grp_seg_ptr[j + 1];
    const int f = 4 * lane;
    const bool active = f < H;
    const float inv_f = 1.0f / (float)H;
    const float4 lg = active ? f4_ld(gamma + f) : f4_zero(), lb = active ? f4_ld(beta + f) : f4_zero();
    float gm_am = 0.0f, gp_am = 0.0f;
    float4 gb = f4_zero(), db = f4_zero(), dg = f4_zero();
    for (int qb = 0; qb < n_src || qb == 0; qb += kW * KS) {
        float4 bh[KS], ga[KS], gbh[KS];
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int q = qb + wave + kW * k;
            bh[k] = (active && q < n_src) ? f4_ld(P + (int64_t)(p_beg + q) * ldp + 2 * H + f) : f4_zero();
            ga[k] = gbh[k] = f4_zero();
        }
        for (int s = s_beg; s < s_end; ++s) {
            const int i = seg_node ? seg_node[s] : s;
            const int e0 = seg_ptr[s];
            int self_q = i - p_beg;
            if (self_q < 0 || self_q >= n_src || seg_ptr[s + 1] - e0 == n_src) self_q = -1;
            const float4 g1 = active ? f4_ld(GS1 + (int64_t)i * H + f) : f4_zero();
            const float4 g0 = active ? f4_ld(GS0 + (int64_t)i * H + f) : f4_zero();
            float4 gbd = f4_zero();
            // (rows in sub-batches of KB: the LayerNorm part keeps three rows' worth of registers per row alive across its
            // wave reductions - all four at once cost the pass a wave per SIMD)
#pragma unroll
            for (int kb = 0; kb < KS; kb += KB) {
                float4 m[KB], xh[KB], gh[KB];
                float s1[KB], s2[KB], rs[KB];
                int row[KB];
#pragma unroll
                for (int kk = 0; kk < KB; ++kk) {
                    const int q = qb + wave + kW * (kb + kk);
                    row[kk] = (q < n_src && q != self_q) ? e0 + q - ((self_q >= 0 && q > self_q) ? 1 : 0) : -1;
                    m[kk] = xh[kk] = gh[kk] = f4_zero();
                    s1[kk] = s2[kk] = rs[kk] = 0.0f;
                    if (row[kk] >= 0) {  // (wave-uniform)
                        const float mean = e_stat[2 * (int64_t)row[kk]];
                        rs[kk] = e_stat[2 * (int64_t)row[kk] + 1];
                        if (active) {
                            m[kk] = f4_lds<STREAM>(M + (int64_t)row[kk] * H + f);
                            const float4 gy = f4_lds<STREAM>(GY + (int64_t)row[kk] * H + f);
                            xh[kk] = make_float4((m[kk].x - mean) * rs[kk], (m[kk].y - mean) * rs[kk], (m[kk].z - mean) * rs[kk],
                                                 (m[kk].w - mean) * rs[kk]);
                            const float4 z = f4_fma(xh[kk], lg, lb);
                            const float4 gz = make_float4(gy.x * dsilu_f(z.x), gy.y * dsilu_f(z.y), gy.z * dsilu_f(z.z), gy.w * dsilu_f(z.w));
                            db = f4_add(db, gz);
                            dg = f4_fma(gz, xh[kk], dg);
                            gh[kk] = f4_mul(gz, lg);
                            s1[kk] = hsum4(gh[kk]);
                            s2[kk] = hsum4(f4_mul(gh[kk], xh[kk]));
                        }
                    }
                }
                wave_sum_k(s1);
                wave_sum_k(s2);
#pragma unroll
                for (int kk = 0; kk < KB; ++kk) {
                    const int k = kb + kk;
                    if (row[kk] >= 0 && active) {
                        const float c1 = s1[kk] * inv_f, c2 = s2[kk] * inv_f;
                        const float4 sg = f4_sigmoid(m[kk]);
                        const float4 gsig = f4_fma(g1, bh[k], g0);
                        float4 gm;
                        gm.x = gsig.x * sg.x * (1.0f - sg.x);
                        gm.y = gsig.y * sg.y * (1.0f - sg.y);
                        gm.z = gsig.z * sg.z * (1.0f - sg.z);
                        gm.w = gsig.w * sg.w * (1.0f - sg.w);
                        float4 gl;
                        gl.x = rs[kk] * (gh[kk].x - c1 - xh[kk].x * c2);
                        gl.y = rs[kk] * (gh[kk].y - c1 - xh[kk].y * c2);
                        gl.z = rs[kk] * (gh[kk].z - c1 - xh[kk].z * c2);
                        gl.w = rs[kk] * (gh[kk].w - c1 - xh[kk].w * c2);
                        gm = f4_add(gm, gl);
                        f4_sts<STREAM>(GM + (int64_t)row[kk] * H + f, gm);
                        gm_am = fmaxf(gm_am, f4_absmax(gm));
                        ga[k] = f4_add(ga[k], gm);
                        gbh[k] = f4_fma(sg, g1, gbh[k]);
                        gbd = f4_add(gbd, gm);
                    }
                }
            }
            float4(*buf)[ALIGNN_WAVE] = sh[(s - s_beg) & 1];
            buf[wave][lane] = gbd;
            __syncthreads();
            if (wave == 0 && active) {
                float4 a = buf[0][lane];
#pragma unroll
                for (int w = 1; w < kW; ++w) a = f4_add(a, buf[w][lane]);
                gb = f4_add(gb, a);
                float* out = GP + (int64_t)i * ldp + H + f;
                if (qb > 0) a = f4_add(f4_ld(out), a);  // written by this very thread in the previous pass
                f4_st(out, a);
                gp_am = fmaxf(gp_am, f4_absmax(a));
            }
        }
        if (active) {
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const int q = qb + wave + kW * k;
                if (q < n_src) {
                    f4_st(GP + (int64_t)(p_beg + q) * ldp + f, ga[k]);
                    f4_st(GP + (int64_t)(p_beg + q) * ldp + 2 * H + f, gbh[k]);
                    gp_am = fmaxf(gp_am, fmaxf(f4_absmax(ga[k]), f4_absmax(gbh[k])));
                }
            }
        }
        __syncthreads();  // sh is reused by the next pass
    }
    if (active && gb_partial && wave == 0) f4_st(gb_partial + (size_t)blockIdx.x * H + f, gb);
    ln_slab_store(db, dg, sh, ln_partial + (size_t)blockIdx.x * 2 * H, H, f, active, wave, lane);
    block_amax_commit(gm_am, gm_amax);
    block_amax_commit(gp_am, gp_amax);
}

// ---------------------------------------------------------------------------------------------------------------------
// dual forward, tangents only: egc_gate_dual_tan_kernel + ln_silu_dual_fwd_kernel (Y == NULL) on (m, mt)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kT) void egc_gate_dual_tan_ln_kernel(
    const float* __restrict__ P, const float* __restrict__ Pt, const float* __restrict__ M, float* __restrict__ Mt,
    const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ seg_node, const int32_t* __restrict__ src, int n_seg, int H,
    float* __restrict__ XPREt, const float* __restrict__ S0, const float* __restrict__ HH, float* __restrict__ S0t,
    float* __restrict__ HHt, const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ e_stat,
    const float* __restrict__ Rt, float* __restrict__ Yt, float* __restrict__ amax2) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t ldp = 4 * (int64_t)H;
    const int first = blockIdx.x * kW + wave, stride = gridDim.x * kW;
    const int f = 4 * lane;
    const bool active = f < H;
    const float inv_f = 1.0f / (float)H;
    const float4 ga = active ? f4_ld(gamma + f) : f4_zero(), be = active ? f4_ld(beta + f) : f4_zero();
    float amt = 0.0f;
    for (int s = first; s < n_seg; s += stride) {
        const int beg = seg_ptr[s], end = seg_ptr[s + 1];
        const int i = seg_node ? seg_node[s] : s;
        const float4 bdt = active ? f4_ld(Pt + (int64_t)i * ldp + H + f) : f4_zero();
        float4 s1t = f4_zero(), s0t = f4_zero();
        for (int e = beg; e < end; e += kK) {
            float4 xh[kK], mt[kK];
            float st[kK], sxt[kK], rs[kK];
#pragma unroll
            for (int k = 0; k < kK; ++k) {
                xh[k] = mt[k] = f4_zero();
                st[k] = sxt[k] = rs[k] = 0.0f;
                if (e + k < end) {  // (wave-uniform)
                    const float mean = e_stat[2 * (int64_t)(e + k)];
                    rs[k] = e_stat[2 * (int64_t)(e + k) + 1];
                    if (active) {
                        const int64_t u = src[e + k];
                        const float4 m = f4_ld(M + (int64_t)(e + k) * H + f);
                        mt[k] = f4_add(f4_add(f4_ld(Pt + u * ldp + f), bdt), f4_ld(Mt + (int64_t)(e + k) * H + f));
                        const float4 bh = f4_ld(P + u * ldp + 2 * H + f), bht = f4_ld(Pt + u * ldp + 2 * H + f);
                        f4_st(Mt + (int64_t)(e + k) * H + f, mt[k]);
                        const float4 sg = f4_sigmoid(m);
                        const float4 sgt = make_float4(sg.x * (1.0f - sg.x) * mt[k].x, sg.y * (1.0f - sg.y) * mt[k].y,
                                                       sg.z * (1.0f - sg.z) * mt[k].z, sg.w * (1.0f - sg.w) * mt[k].w);
                        s1t = f4_fma(sgt, bh, f4_fma(sg, bht, s1t));
                        s0t = f4_add(s0t, sgt);
                        xh[k] = make_float4((m.x - mean) * rs[k], (m.y - mean) * rs[k], (m.z - mean) * rs[k], (m.w - mean) * rs[k]);
                        st[k] = hsum4(mt[k]);
                        sxt[k] = hsum4(f4_mul(xh[k], mt[k]));
                    }
                }
            }
            wave_sum_k(st);
            wave_sum_k(sxt);
#pragma unroll
            for (int k = 0; k < kK; ++k) {
                if (e + k < end && active) {
                    const float m1 = st[k] * inv_f, m2 = sxt[k] * inv_f;
                    float4 ot;
#define ALIGNN_LN_T(q)                                                  \
    {                                                                   \
        const float th = rs[k] * (mt[k].q - m1 - xh[k].q * m2);         \
        const float z = fmaf(xh[k].q, ga.q, be.q), zt = ga.q * th;      \
        const float sg = sig_f(z);                                      \
        ot.q = (sg + z * sg * (1.0f - sg)) * zt;                        \
    }
                    ALIGNN_LN_T(x) ALIGNN_LN_T(y) ALIGNN_LN_T(z) ALIGNN_LN_T(w)
#undef ALIGNN_LN_T
                    if (Rt) ot = f4_add(ot, f4_ld(Rt + (int64_t)(e + k) * H + f));
                    f4_st(Yt + (int64_t)(e + k) * H + f, ot);
                    amt = fmaxf(amt, f4_absmax(ot));
                }
            }
        }
        if (active) {
            const float4 s0 = f4_ld(S0 + (int64_t)i * H + f), h = f4_ld(HH + (int64_t)i * H + f);
            float4 ht;
            ht.x = (s1t.x - h.x * s0t.x) / (s0.x + ALIGNN_EPS_GATE);
            ht.y = (s1t.y - h.y * s0t.y) / (s0.y + ALIGNN_EPS_GATE);
            ht.z = (s1t.z - h.z * s0t.z) / (s0.z + ALIGNN_EPS_GATE);
            ht.w = (s1t.w - h.w * s0t.w) / (s0.w + ALIGNN_EPS_GATE);
            f4_st(XPREt + (int64_t)i * H + f, f4_add(f4_ld(Pt + (int64_t)i * ldp + 3 * H + f), ht));
            f4_st(S0t + (int64_t)i * H + f, s0t);
            f4_st(HHt + (int64_t)i * H + f, ht);
        }
    }
    if (amax2 != nullptr) {
        block_amax_commit(0.0f, amax2);
        block_amax_commit(amt, amax2 + 1);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// dual reverse: ln_silu_dual_bwd_kernel on (GY, GYt, M, Mt) + egc_dual_bwd_lg_dense_kernel<true, STREAM>
// ---------------------------------------------------------------------------------------------------------------------
template <bool STREAM, int KS, int KB, bool EARLY>
__global__ __launch_bounds__(kT) void egc_dual_bwd_lg_dense_ln_kernel(
    const float* __restrict__ GY, const float* __restrict__ GYt, const float* __restrict__ M, const float* __restrict__ Mt,
    const float* __restrict__ P, const float* __restrict__ Pt, const float* __restrict__ Q1, const float* __restrict__ Q0,
    const float* __restrict__ Q1t, const float* __restrict__ Q0t, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ e_stat, const int32_t* __restrict__ grp_seg_ptr, const int32_t* __restrict__ grp_src_ptr,
    const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ seg_node, int H, float* __restrict__ GM,
    float* __restrict__ GMt, float* __restrict__ GP, float* __restrict__ GPt, float* __restrict__ gb_partial,
    float* __restrict__ ln_partial, float* __restrict__ gm_amax2, float* __restrict__ gp_amax2) {
    __shared__ float4 sh[2][2][kW][ALIGNN_WAVE];  // [buffer][value | tangent][wave][lane]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t ldp = 4 * (int64_t)H;
    const int j = blockIdx.x;
    const int p_beg = grp_src_ptr[j], n_src = grp_src_ptr[j + 1] - p_beg;
    const int s_beg = grp_seg_ptr[j], s_end = grp_seg_ptr[j + 1];
    const int f = 4 * lane;
    const bool active = f < H;
    const float inv_f = 1.0f / (float)H;
    const float4 lg = active ? f4_ld(gamma + f) : f4_zero(), lb = active ? f4_ld(beta + f) : f4_zero();
    float am = 0.0f, amt = 0.0f, pam = 0.0f, pamt = 0.0f;
    float4 gb = f4_zero(), db = f4_zero(), dg = f4_zero();
    for (int qb = 0; qb < n_src || qb == 0; qb += kW * KS) {
        float4 bh[KS], bht[KS], ga[KS], gat[KS], gbh[KS], gbht[KS];
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int q = qb + wave + kW * k;
            const bool have = active && q < n_src;
            bh[k] = have ? f4_ld(P + (int64_t)(p_beg + q) * ldp + 2 * H + f) : f4_zero();
            bht[k] = have ? f4_ld(Pt + (int64_t)(p_beg + q) * ldp + 2 * H + f) : f4_zero();
            ga[k] = gat[k] = gbh[k] = gbht[k] = f4_zero();
        }
        for (int s = s_beg; s < s_end; ++s) {
            const int i = seg_node ? seg_node[s] : s;
            const int e0 = seg_ptr[s];
            int self_q = i - p_beg;
            if (self_q < 0 || self_q >= n_src || seg_ptr[s + 1] - e0 == n_src) self_q = -1;
            const float4 q1 = active ? f4_ld(Q1 + (int64_t)i * H + f) : f4_zero(), q0 = active ? f4_ld(Q0 + (int64_t)i * H + f) : f4_zero();
            const float4 q1t = active ? f4_ld(Q1t + (int64_t)i * H + f) : f4_zero(), q0t = active ? f4_ld(Q0t + (int64_t)i * H + f) : f4_zero();
            float4 gbd = f4_zero(), gbdt = f4_zero();
            // per row: m, x-hat, t = mt, the LayerNorm adjoints a (of t-hat) and b (of x-hat); rows in sub-batches of KB
#pragma unroll
            for (int kb = 0; kb < KS; kb += KB) {
                float4 m[KB], xh[KB], t[KB], a[KB], b[KB], gye[EARLY ? KB : 1], gyte[EARLY ? KB : 1];
                float r0[KB], r1[KB], r2[KB], rs[KB], m2[KB];
                int row[KB];
#pragma unroll
                for (int kk = 0; kk < KB; ++kk) {
                    const int q = qb + wave + kW * (kb + kk);
                    row[kk] = (q < n_src && q != self_q) ? e0 + q - ((self_q >= 0 && q > self_q) ? 1 : 0) : -1;
                    m[kk] = xh[kk] = t[kk] = a[kk] = b[kk] = f4_zero();
                    r0[kk] = r1[kk] = rs[kk] = 0.0f;
                    if (row[kk] >= 0) {  // (wave-uniform)
                        const float mean = e_stat[2 * (int64_t)row[kk]];
                        rs[kk] = e_stat[2 * (int64_t)row[kk] + 1];
                        if (active) {
                            m[kk] = f4_lds<STREAM>(M + (int64_t)row[kk] * H + f);
                            t[kk] = f4_lds<STREAM>(Mt + (int64_t)row[kk] * H + f);
                            if (EARLY) {  // (requested with m and mt: one exposed latency per sub-batch instead of two)
                                gye[kk] = f4_lds<STREAM>(GY + (int64_t)row[kk] * H + f);
                                gyte[kk] = f4_lds<STREAM>(GYt + (int64_t)row[kk] * H + f);
                            }
                            xh[kk] = make_float4((m[kk].x - mean) * rs[kk], (m[kk].y - mean) * rs[kk], (m[kk].z - mean) * rs[kk],
                                                 (m[kk].w - mean) * rs[kk]);
                            r0[kk] = hsum4(t[kk]);
                            r1[kk] = hsum4(f4_mul(xh[kk], t[kk]));
                        }
                    }
                }
                wave_sum_k(r0);
                wave_sum_k(r1);
#pragma unroll
                for (int kk = 0; kk < KB; ++kk) {
                    const float m1 = r0[kk] * inv_f;
                    m2[kk] = r1[kk] * inv_f;
                    r0[kk] = r1[kk] = r2[kk] = 0.0f;  // -> sum a, sum a x-hat, sum a t-hat
                    if (row[kk] >= 0 && active) {
                        const float4 gy = EARLY ? gye[kk] : f4_lds<STREAM>(GY + (int64_t)row[kk] * H + f);
                        const float4 gyt = EARLY ? gyte[kk] : f4_lds<STREAM>(GYt + (int64_t)row[kk] * H + f);
#define ALIGNN_LN_DB(q)                                                                     \
    {                                                                                       \
        const float th = rs[kk] * (t[kk].q - m1 - xh[kk].q * m2[kk]);                       \
        const float z = fmaf(xh[kk].q, lg.q, lb.q), zt = lg.q * th;                         \
        float d1, d2;                                                                       \
        dsilu2(z, d1, d2);                                                                  \
        const float gzt = gyt.q * d1;                                                       \
        const float gz = gy.q * d1 + gyt.q * d2 * zt;                                       \
        db.q += gz;                                                                         \
        dg.q += gz * xh[kk].q + gzt * th;                                                   \
        a[kk].q = lg.q * gzt;                                                               \
        b[kk].q = lg.q * gz;                                                                \
        r0[kk] += a[kk].q;                                                                  \
        r1[kk] += a[kk].q * xh[kk].q;                                                       \
        r2[kk] += a[kk].q * th;                                                             \
    }
                        ALIGNN_LN_DB(x) ALIGNN_LN_DB(y) ALIGNN_LN_DB(z) ALIGNN_LN_DB(w)
#undef ALIGNN_LN_DB
                    }
                }
                wave_sum_k(r0);
                wave_sum_k(r1);
                wave_sum_k(r2);
                float A1[KB], A2[KB], A3[KB];
#pragma unroll
                for (int kk = 0; kk < KB; ++kk) {
                    A1[kk] = r0[kk] * inv_f, A2[kk] = r1[kk] * inv_f, A3[kk] = r2[kk] * inv_f;
                    r0[kk] = r1[kk] = 0.0f;  // -> sum b, sum b x-hat
                    if (row[kk] >= 0 && active) {
                        b[kk].x -= rs[kk] * (a[kk].x * m2[kk] + t[kk].x * A2[kk]);
                        b[kk].y -= rs[kk] * (a[kk].y * m2[kk] + t[kk].y * A2[kk]);
                        b[kk].z -= rs[kk] * (a[kk].z * m2[kk] + t[kk].z * A2[kk]);
                        b[kk].w -= rs[kk] * (a[kk].w * m2[kk] + t[kk].w * A2[kk]);
                        r0[kk] = hsum4(b[kk]);
                        r1[kk] = hsum4(f4_mul(b[kk], xh[kk]));
                    }
                }
                wave_sum_k(r0);
                wave_sum_k(r1);
#pragma unroll
                for (int kk = 0; kk < KB; ++kk) {
                    const int k = kb + kk;
                    if (row[kk] >= 0 && active) {
                        const float B1 = r0[kk] * inv_f, B2 = r1[kk] * inv_f;
                        float4 gm, gmt;
#define ALIGNN_GDL(c)                                                                                     \
    {                                                                                                     \
        gmt.c = rs[kk] * (a[kk].c - A1[kk] - xh[kk].c * A2[kk]);                                           \
        gm.c = rs[kk] * (b[kk].c - B1 - xh[kk].c * B2) - rs[kk] * xh[kk].c * A3[kk];                       \
        const float sg = sig_f(m[kk].c), sp = sg * (1.0f - sg);                                           \
        const float gs = q1.c * bh[k].c + q0.c + q1t.c * bht[k].c; /* adjoint of sigma */                  \
        const float gst = q1t.c * bh[k].c + q0t.c;                 /* adjoint of sigma-dot */              \
        gm.c += gs * sp + gst * sp * (1.0f - 2.0f * sg) * t[kk].c;                                         \
        gmt.c += gst * sp;                                                                                \
        const float sgt = sp * t[kk].c;                                                                   \
        gbh[k].c += sg * q1.c + sgt * q1t.c;                                                              \
        gbht[k].c += sg * q1t.c;                                                                          \
    }
                        ALIGNN_GDL(x) ALIGNN_GDL(y) ALIGNN_GDL(z) ALIGNN_GDL(w)
#undef ALIGNN_GDL
                        f4_sts<STREAM>(GM + (int64_t)row[kk] * H + f, gm);
                        f4_sts<STREAM>(GMt + (int64_t)row[kk] * H + f, gmt);
                        am = fmaxf(am, f4_absmax(gm));
                        amt = fmaxf(amt, f4_absmax(gmt));
                        ga[k] = f4_add(ga[k], gm);
                        gat[k] = f4_add(gat[k], gmt);
                        gbd = f4_add(gbd, gm);
                        gbdt = f4_add(gbdt, gmt);
                    }
                }
            }
            float4(*buf)[kW][ALIGNN_WAVE] = sh[(s - s_beg) & 1];
            buf[0][wave][lane] = gbd;
            buf[1][wave][lane] = gbdt;
            __syncthreads();
            if (wave == 0 && active) {
                float4 sa = buf[0][0][lane], sat = buf[1][0][lane];
#pragma unroll
                for (int w = 1; w < kW; ++w) {
                    sa = f4_add(sa, buf[0][w][lane]);
                    sat = f4_add(sat, buf[1][w][lane]);
                }
                gb = f4_add(gb, sa);
                float* out = GP + (int64_t)i * ldp + H + f;
                float* outt = GPt + (int64_t)i * ldp + H + f;
                if (qb > 0) {  // (written by this very thread in the previous pass)
                    sa = f4_add(f4_ld(out), sa);
                    sat = f4_add(f4_ld(outt), sat);
                }
                f4_st(out, sa);
                f4_st(outt, sat);
                pam = fmaxf(pam, f4_absmax(sa));
                pamt = fmaxf(pamt, f4_absmax(sat));
            }
        }
        if (active) {
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const int q = qb + wave + kW * k;
                if (q < n_src) {
                    f4_st(GP + (int64_t)(p_beg + q) * ldp + f, ga[k]);
                    f4_st(GPt + (int64_t)(p_beg + q) * ldp + f, gat[k]);
                    f4_st(GP + (int64_t)(p_beg + q) * ldp + 2 * H + f, gbh[k]);
                    f4_st(GPt + (int64_t)(p_beg + q) * ldp + 2 * H + f, gbht[k]);
                    pam = fmaxf(pam, fmaxf(f4_absmax(ga[k]), f4_absmax(gbh[k])));
                    pamt = fmaxf(pamt, fmaxf(f4_absmax(gat[k]), f4_absmax(gbht[k])));
                }
            }
        }
        __syncthreads();  // sh is reused by the next pass
    }
    if (active && gb_partial && wave == 0) f4_st(gb_partial + (size_t)blockIdx.x * H + f, gb);
    ln_slab_store(db, dg, sh[0], ln_partial + (size_t)blockIdx.x * 2 * H, H, f, active, wave, lane);
    if (gm_amax2 != nullptr) {
        block_amax_commit(am, gm_amax2);
        block_amax_commit(amt, gm_amax2 + 1);
    }
    if (gp_amax2 != nullptr) {
        block_amax_commit(pam, gp_amax2);
        block_amax_commit(pamt, gp_amax2 + 1);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// the same for graphs WITHOUT dense blocks (the bond graph g; line graphs with the dense reverse switched off): the
// destination-ordered halves egc_bwd_dst_kernel<2> / egc_dual_bwd_dst_kernel with the LayerNorm adjoint of the edge output formed
// in the pass (one wavefront per destination segment, rows one at a time; the source-ordered halves egc_bwd_src / egc_dual_bwd_src
// follow unchanged).  These are bond-row passes - cache-resident, launch- and latency-bound on the chain the T-row lane waits
// for: what they save is a launch and a round trip of the branch gradient per convolution and reverse.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kT) void egc_bwd_dst_ln_kernel(
    const float* __restrict__ GY, const float* __restrict__ M, const float* __restrict__ P, const float* __restrict__ GS1,
    const float* __restrict__ GS0, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ e_stat, const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ seg_node,
    const int32_t* __restrict__ src, int n_seg, int H, float* __restrict__ GM, float* __restrict__ GP,
    float* __restrict__ gb_partial, float* __restrict__ ln_partial, float* __restrict__ gm_amax, float* __restrict__ gp_amax) {
    __shared__ float4 sh[2][kW][ALIGNN_WAVE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t ldp = 4 * (int64_t)H;
    const int first = blockIdx.x * kW + wave, stride = gridDim.x * kW;
    const int f = 4 * lane;
    const bool active = f < H;
    const float inv_f = 1.0f / (float)H;
    const float4 lg = active ? f4_ld(gamma + f) : f4_zero(), lb = active ? f4_ld(beta + f) : f4_zero();
    float gm_am = 0.0f, gp_am = 0.0f;
    float4 gb = f4_zero(), db = f4_zero(), dg = f4_zero();
    for (int s = first; s < n_seg; s += stride) {
        const int beg = seg_ptr[s], end = seg_ptr[s + 1];
        const int i = seg_node ? seg_node[s] : s;
        const float4 gs1 = active ? f4_ld(GS1 + (int64_t)i * H + f) : f4_zero();
        const float4 gs0 = active ? f4_ld(GS0 + (int64_t)i * H + f) : f4_zero();
        float4 gbd = f4_zero();
        for (int e = beg; e < end; ++e) {
            const float mean = e_stat[2 * (int64_t)e], rstd = e_stat[2 * (int64_t)e + 1];
            float4 m = f4_zero(), bh = f4_zero(), xh = f4_zero(), gh = f4_zero();
            float s1 = 0.0f, s2 = 0.0f;
            if (active) {
                m = f4_ld(M + (int64_t)e * H + f);
                bh = f4_ld(P + (int64_t)src[e] * ldp + 2 * H + f);
                const float4 gy = f4_ld(GY + (int64_t)e * H + f);
                xh = make_float4((m.x - mean) * rstd, (m.y - mean) * rstd, (m.z - mean) * rstd, (m.w - mean) * rstd);
                const float4 z = f4_fma(xh, lg, lb);
                const float4 gz = make_float4(gy.x * dsilu_f(z.x), gy.y * dsilu_f(z.y), gy.z * dsilu_f(z.z), gy.w * dsilu_f(z.w));
                db = f4_add(db, gz);
                dg = f4_fma(gz, xh, dg);
                gh = f4_mul(gz, lg);
                s1 = hsum4(gh);
                s2 = hsum4(f4_mul(gh, xh));
            }
            const float c1 = wave_sum1(s1) * inv_f, c2 = wave_sum1(s2) * inv_f;
            if (active) {
                const float4 sg = f4_sigmoid(m);
                const float4 gsig = f4_fma(gs1, bh, gs0);
                float4 gm;
                gm.x = gsig.x * sg.x * (1.0f - sg.x) + rstd * (gh.x - c1 - xh.x * c2);
                gm.y = gsig.y * sg.y * (1.0f - sg.y) + rstd * (gh.y - c1 - xh.y * c2);
                gm.z = gsig.z * sg.z * (1.0f - sg.z) + rstd * (gh.z - c1 - xh.z * c2);
                gm.w = gsig.w * sg.w * (1.0f - sg.w) + rstd * (gh.w - c1 - xh.w * c2);
                f4_st(GM + (int64_t)e * H + f, gm);
                gm_am = fmaxf(gm_am, f4_absmax(gm));
                gbd = f4_add(gbd, gm);
            }
        }
        if (active) {
            f4_st(GP + (int64_t)i * ldp + H + f, gbd);
            gp_am = fmaxf(gp_am, f4_absmax(gbd));
            gb = f4_add(gb, gbd);
        }
    }
    sh[0][wave][lane] = gb;
    __syncthreads();
    if (wave == 0 && active && gb_partial) {
        float4 a = sh[0][0][lane];
#pragma unroll
        for (int w = 1; w < kW; ++w) a = f4_add(a, sh[0][w][lane]);
        f4_st(gb_partial + (size_t)blockIdx.x * H + f, a);
    }
    ln_slab_store(db, dg, sh, ln_partial + (size_t)blockIdx.x * 2 * H, H, f, active, wave, lane);
    block_amax_commit(gm_am, gm_amax);
    block_amax_commit(gp_am, gp_amax);
}

__global__ __launch_bounds__(kT) void egc_dual_bwd_dst_ln_kernel(
    const float* __restrict__ GY, const float* __restrict__ GYt, const float* __restrict__ M, const float* __restrict__ Mt,
    const float* __restrict__ P, const float* __restrict__ Pt, const float* __restrict__ Q1, const float* __restrict__ Q0,
    const float* __restrict__ Q1t, const float* __restrict__ Q0t, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ e_stat, const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ seg_node,
    const int32_t* __restrict__ src, int n_seg, int H, float* __restrict__ GM, float* __restrict__ GMt, float* __restrict__ GP,
    float* __restrict__ GPt, float* __restrict__ gb_partial, float* __restrict__ ln_partial, float* __restrict__ gm_amax2,
    float* __restrict__ gp_amax2) {
    __shared__ float4 sh[2][kW][ALIGNN_WAVE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t ldp = 4 * (int64_t)H;
    const int first = blockIdx.x * kW + wave, stride = gridDim.x * kW;
    const int f = 4 * lane;
    const bool active = f < H;
    const float inv_f = 1.0f / (float)H;
    const float4 lg = active ? f4_ld(gamma + f) : f4_zero(), lb = active ? f4_ld(beta + f) : f4_zero();
    float am = 0.0f, amt = 0.0f, pam = 0.0f, pamt = 0.0f;
    float4 gb = f4_zero(), db = f4_zero(), dg = f4_zero();
    for (int s = first; s < n_seg; s += stride) {
        const int beg = seg_ptr[s], end = seg_ptr[s + 1];
        const int i = seg_node ? seg_node[s] : s;
        const float4 q1 = active ? f4_ld(Q1 + (int64_t)i * H + f) : f4_zero(), q0 = active ? f4_ld(Q0 + (int64_t)i * H + f) : f4_zero();
        const float4 q1t = active ? f4_ld(Q1t + (int64_t)i * H + f) : f4_zero(), q0t = active ? f4_ld(Q0t + (int64_t)i * H + f) : f4_zero();
        float4 gbd = f4_zero(), gbdt = f4_zero();
        for (int e = beg; e < end; ++e) {
            const float mean = e_stat[2 * (int64_t)e], rs = e_stat[2 * (int64_t)e + 1];
            float4 m = f4_zero(), t = f4_zero(), xh = f4_zero(), a = f4_zero(), b = f4_zero(), bh = f4_zero(), bht = f4_zero();
            float4 gy = f4_zero(), gyt = f4_zero();
            float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
            if (active) {
                const int64_t u = src[e];
                m = f4_ld(M + (int64_t)e * H + f);
                t = f4_ld(Mt + (int64_t)e * H + f);
                gy = f4_ld(GY + (int64_t)e * H + f);
                gyt = f4_ld(GYt + (int64_t)e * H + f);
                bh = f4_ld(P + u * ldp + 2 * H + f);
                bht = f4_ld(Pt + u * ldp + 2 * H + f);
                xh = make_float4((m.x - mean) * rs, (m.y - mean) * rs, (m.z - mean) * rs, (m.w - mean) * rs);
                r0 = hsum4(t);
                r1 = hsum4(f4_mul(xh, t));
            }
            const float m1 = wave_sum1(r0) * inv_f, m2 = wave_sum1(r1) * inv_f;
            r0 = r1 = r2 = 0.0f;
            if (active) {
#define ALIGNN_LN_DB(q)                                              \
    {                                                                \
        const float th = rs * (t.q - m1 - xh.q * m2);                \
        const float z = fmaf(xh.q, lg.q, lb.q), zt = lg.q * th;      \
        float d1, d2;                                                \
        dsilu2(z, d1, d2);                                           \
        const float gzt = gyt.q * d1;                                \
        const float gz = gy.q * d1 + gyt.q * d2 * zt;                \
        db.q += gz;                                                  \
        dg.q += gz * xh.q + gzt * th;                                \
        a.q = lg.q * gzt;                                            \
        b.q = lg.q * gz;                                             \
        r0 += a.q;                                                   \
        r1 += a.q * xh.q;                                            \
        r2 += a.q * th;                                              \
    }
                ALIGNN_LN_DB(x) ALIGNN_LN_DB(y) ALIGNN_LN_DB(z) ALIGNN_LN_DB(w)
#undef ALIGNN_LN_DB
            }
            const float A1 = wave_sum1(r0) * inv_f, A2 = wave_sum1(r1) * inv_f, A3 = wave_sum1(r2) * inv_f;
            r0 = r1 = 0.0f;
            if (active) {
                b.x -= rs * (a.x * m2 + t.x * A2);
                b.y -= rs * (a.y * m2 + t.y * A2);
                b.z -= rs * (a.z * m2 + t.z * A2);
                b.w -= rs * (a.w * m2 + t.w * A2);
                r0 = hsum4(b);
                r1 = hsum4(f4_mul(b, xh));
            }
            const float B1 = wave_sum1(r0) * inv_f, B2 = wave_sum1(r1) * inv_f;
            if (active) {
                float4 gm, gmt;
#define ALIGNN_GD(c)                                                            \
    {                                                                           \
        gmt.c = rs * (a.c - A1 - xh.c * A2);                                    \
        gm.c = rs * (b.c - B1 - xh.c * B2) - rs * xh.c * A3;                    \
        const float sg = sig_f(m.c), sp = sg * (1.0f - sg);                     \
        const float gs = q1.c * bh.c + q0.c + q1t.c * bht.c; /* adj. sigma */   \
        const float gst = q1t.c * bh.c + q0t.c;              /* adj. sigma-dot */ \
        gm.c += gs * sp + gst * sp * (1.0f - 2.0f * sg) * t.c;                  \
        gmt.c += gst * sp;                                                      \
    }
                ALIGNN_GD(x) ALIGNN_GD(y) ALIGNN_GD(z) ALIGNN_GD(w)
#undef ALIGNN_GD
                f4_st(GM + (int64_t)e * H + f, gm);
                f4_st(GMt + (int64_t)e * H + f, gmt);
                am = fmaxf(am, f4_absmax(gm));
                amt = fmaxf(amt, f4_absmax(gmt));
                gbd = f4_add(gbd, gm);
                gbdt = f4_add(gbdt, gmt);
            }
        }
        if (active) {
            f4_st(GP + (int64_t)i * ldp + H + f, gbd);
            f4_st(GPt + (int64_t)i * ldp + H + f, gbdt);
            pam = fmaxf(pam, f4_absmax(gbd));
            pamt = fmaxf(pamt, f4_absmax(gbdt));
            gb = f4_add(gb, gbd);
        }
    }
    sh[0][wave][lane] = gb;
    __syncthreads();
    if (wave == 0 && active && gb_partial) {
        float4 a = sh[0][0][lane];
#pragma unroll
        for (int w = 1; w < kW; ++w) a = f4_add(a, sh[0][w][lane]);
        f4_st(gb_partial + (size_t)blockIdx.x * H + f, a);
    }
    ln_slab_store(db, dg, sh, ln_partial + (size_t)blockIdx.x * 2 * H, H, f, active, wave, lane);
    if (gm_amax2 != nullptr) {
        block_amax_commit(am, gm_amax2);
        block_amax_commit(amt, gm_amax2 + 1);
    }
    if (gp_amax2 != nullptr) {
        block_amax_commit(pam, gp_amax2);
        block_amax_commit(pamt, gp_amax2 + 1);
    }
}

// sources per wave and pass / rows per sub-batch of the two reverse kernels (ALIGNN_AMD_LN_REV = "<value><dual>", A/B runs)
inline int reverse_variant(int which) {  // (read per call: A/B runs inside one process)
    const char* e = std::getenv("ALIGNN_AMD_LN_REV");
    if (e == nullptr || e[0] < '0' || e[0] > '9') return 0;
    if (which == 0) return e[0] - '0';
    return (e[1] >= '0' && e[1] <= '9') ? e[1] - '0' : 0;
}
inline bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

/* 1 when the fused passes below are the ones to take for [m_rows, H] edge tensors: H % 4 == 0, H <= 256 (one feature panel per
   wavefront), and tensors that cannot stay in the last-level cache (>= 64 MiB each) - there a pass less is time less; on
   cache-resident line graphs (a 200-atom MD cell: 35 k rows) the separate LayerNorm kernels win by their finer grids (tools/
   ln_rev_time.py: 70 vs 93 us at 1 x 200 atoms, 134 vs 130 at 8 x 60, 727 vs 426 at 16 x 200).  ALIGNN_AMD_LN_FUSED=0: never,
   =2: whenever H allows (tests).  Read per call. */
static bool ln_h_ok(int H) { return H >= 4 && (H & 3) == 0 && H <= 4 * ALIGNN_WAVE; }
int alignn_egc_ln_fused_supported(int H, int64_t m_rows) {
    const char* e = std::getenv("ALIGNN_AMD_LN_FUSED");
    if (e != nullptr && e[0] == '0') return 0;
    if (!ln_h_ok(H)) return 0;
    if (e != nullptr && e[0] == '2') return 1;
    return m_rows * (int64_t)H * 4 >= ((int64_t)64 << 20) ? 1 : 0;
}

int alignn_egc_gate_fwd_pre_ln(const float* P, const float* M, const int32_t* seg_ptr, const int32_t* seg_node, const int32_t* src,
                               int64_t n_seg, int64_t m_rows, int H, float* XPRE, float* S0, float* HH, const float* gamma,
                               const float* beta, float eps, const float* Y, float* YOUT, float* e_stat, float* y_amax,
                               alignn_stream_t stream) {
    if (!ln_h_ok(H) || n_seg < 0 || n_seg > INT32_MAX || m_rows < 0 || !YOUT || !e_stat || !gamma || !beta)
        return (int)hipErrorInvalidValue;
    if (n_seg == 0) return 0;
    if (big_stream(m_rows, H))
        hipLaunchKernelGGL(egc_gate_fwd_ln_kernel<true>, dim3(seg_blocks(n_seg)), dim3(kT), 0, (hipStream_t)stream, P, M, seg_ptr,
                           seg_node, src, (int)n_seg, H, XPRE, S0, HH, gamma, beta, eps, Y, YOUT, e_stat, y_amax);
    else
        hipLaunchKernelGGL(egc_gate_fwd_ln_kernel<false>, dim3(seg_blocks(n_seg)), dim3(kT), 0, (hipStream_t)stream, P, M, seg_ptr,
                           seg_node, src, (int)n_seg, H, XPRE, S0, HH, gamma, beta, eps, Y, YOUT, e_stat, y_amax);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_egc_bwd_lg_dense_ln(const float* GY, const float* M, const float* P, const float* GS1, const float* GS0,
                               const float* gamma, const float* beta, const float* e_stat, int64_t m_rows,
                               const int32_t* grp_seg_ptr, const int32_t* grp_src_ptr, int64_t n_groups, int max_group_src,
                               const int32_t* seg_ptr, const int32_t* seg_node, int H, float* GM, float* GP, float* gb_partial,
                               float* ln_partial, float* gm_amax, float* gp_amax, alignn_stream_t stream) {
    if (!ln_h_ok(H) || n_groups <= 0 || n_groups > INT32_MAX || max_group_src <= 0 || !GY || !e_stat ||
        !ln_partial)
        return (int)hipErrorInvalidValue;
    const dim3 grid((int)n_groups), block(kT);
    const bool big = big_stream(m_rows, H);
#define ALIGNN_LNV(ST_, KS_, KB_)                                                                                              \
    hipLaunchKernelGGL((egc_bwd_lg_dense_ln_kernel<ST_, KS_, KB_>), grid, block, 0, (hipStream_t)stream, GY, M, P, GS1, GS0, gamma, \
                       beta, e_stat, grp_seg_ptr, grp_src_ptr, seg_ptr, seg_node, H, GM, GP, gb_partial, ln_partial, gm_amax,    \
                       gp_amax)
    // (tools/ln_rev_time.py, line graph of 16 x 200 atoms: one row at a time 426 us, rows in pairs 478; separate kernels 715-745)
    switch (reverse_variant(0)) {
        case 1: if (big) ALIGNN_LNV(true, 4, 2); else ALIGNN_LNV(false, 4, 2); break;
        case 2: if (big) ALIGNN_LNV(true, 2, 2); else ALIGNN_LNV(false, 2, 2); break;
        default: if (big) ALIGNN_LNV(true, 4, 1); else ALIGNN_LNV(false, 4, 1); break;
    }
#undef ALIGNN_LNV
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_egc_gate_dual_tan_ln(const float* P, const float* Pt, const float* M, float* Mt, const int32_t* seg_ptr,
                                const int32_t* seg_node, const int32_t* src, int64_t n, int64_t m, int H, float* xpre_t,
                                const float* s0, const float* hh, float* s0t, float* hht, const float* gamma, const float* beta,
                                const float* e_stat, const float* Rt, float* Yt, float* amax2, alignn_stream_t stream) {
    if (!ln_h_ok(H) || !a16(P) || !a16(M) || !e_stat || !Yt) return (int)hipErrorInvalidValue;
    (void)m;
    if (n == 0) return 0;
    hipLaunchKernelGGL(egc_gate_dual_tan_ln_kernel, dim3(seg_blocks(n)), dim3(kT), 0, (hipStream_t)stream, P, Pt, M, Mt, seg_ptr,
                       seg_node, src, (int)n, H, xpre_t, s0, hh, s0t, hht, gamma, beta, e_stat, Rt, Yt, amax2);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_egc_dual_bwd_lg_dense_ln(const float* GY, const float* GYt, const float* M, const float* Mt, const float* P,
                                    const float* Pt, const float* q1, const float* q0, const float* q1t, const float* q0t,
                                    const float* gamma, const float* beta, const float* e_stat, int64_t m_rows,
                                    const int32_t* grp_seg_ptr, const int32_t* grp_src_ptr, int64_t n_groups,
                                    const int32_t* seg_ptr, const int32_t* seg_node, int H, float* GM, float* GMt, float* GP,
                                    float* GPt, float* gb_partial, float* ln_partial, float* gm_amax2, float* gp_amax2,
                                    alignn_stream_t stream) {
    if (!ln_h_ok(H) || !GY || !GYt || !e_stat || !ln_partial || n_groups < 0 || n_groups > INT32_MAX)
        return (int)hipErrorInvalidValue;
    if (n_groups == 0) return 0;
    const dim3 grid((unsigned)n_groups), block(kT);
    const bool big = big_stream(m_rows, H);
#define ALIGNN_LND(ST_, KS_, KB_, EA_)                                                                                          \
    hipLaunchKernelGGL((egc_dual_bwd_lg_dense_ln_kernel<ST_, KS_, KB_, EA_>), grid, block, 0, (hipStream_t)stream, GY, GYt, M, Mt, P, \
                       Pt, q1, q0, q1t, q0t, gamma, beta, e_stat, grp_seg_ptr, grp_src_ptr, seg_ptr, seg_node, H, GM, GMt, GP, GPt, \
                       gb_partial, ln_partial, gm_amax2, gp_amax2)
    // measured on the line graph of 16 x 200 atoms (tools/ln_rev_time.py; separate kernels 1 390-1 450 us): 4 sources per wave
    // and pass, one row at a time, all four loads of a row requested together: 850-900 us; rows in pairs 1 020-1 050; 2 sources
    // per wave (two passes over the segments, three waves per SIMD) 875-940
    switch (reverse_variant(1)) {
        case 1: if (big) ALIGNN_LND(true, 4, 2, false); else ALIGNN_LND(false, 4, 2, false); break;
        case 2: if (big) ALIGNN_LND(true, 2, 1, false); else ALIGNN_LND(false, 2, 1, false); break;
        default: if (big) ALIGNN_LND(true, 4, 1, true); else ALIGNN_LND(false, 4, 1, true); break;
    }
#undef ALIGNN_LND
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

/* Destination-ordered reverse passes with the LayerNorm inside, for graphs without dense blocks (the bond graph): what
   alignn_ln_silu_bwd + alignn_egc_bwd_dst(e_stat = NULL) resp. alignn_ln_silu_dual_bwd + alignn_egc_dual_bwd_dst compute; the
   source-ordered halves (alignn_egc_bwd_src / alignn_egc_dual_bwd_src) follow as before.  Slabs: alignn_egc_ln_dst_slabs(n_seg)
   of [H] (bias gradient) and [2][H] (LayerNorm parameter gradients). */
int alignn_egc_ln_dst_slabs(int64_t n_seg) { return seg_blocks(n_seg); }
int alignn_egc_ln_dst_supported(int H) {  // (any size: these are bond-row passes; ALIGNN_AMD_LN_FUSED=0 switches them off too)
    const char* e = std::getenv("ALIGNN_AMD_LN_FUSED");
    return (e != nullptr && e[0] == '0') ? 0 : (ln_h_ok(H) ? 1 : 0);
}

int alignn_egc_bwd_dst_ln(const float* GY, const float* M, const float* P, const float* GS1, const float* GS0, const float* gamma,
                          const float* beta, const float* e_stat, const int32_t* seg_ptr, const int32_t* seg_node,
                          const int32_t* src, int64_t n_seg, int H, float* GM, float* GP, float* gb_partial, float* ln_partial,
                          float* gm_amax, float* gp_amax, alignn_stream_t stream) {
    if (!ln_h_ok(H) || n_seg < 0 || n_seg > INT32_MAX || !GY || !e_stat || !ln_partial) return (int)hipErrorInvalidValue;
    if (n_seg == 0) return 0;
    hipLaunchKernelGGL(egc_bwd_dst_ln_kernel, dim3(seg_blocks(n_seg)), dim3(kT), 0, (hipStream_t)stream, GY, M, P, GS1, GS0, gamma,
                       beta, e_stat, seg_ptr, seg_node, src, (int)n_seg, H, GM, GP, gb_partial, ln_partial, gm_amax, gp_amax);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_egc_dual_bwd_dst_ln(const float* GY, const float* GYt, const float* M, const float* Mt, const float* P, const float* Pt,
                               const float* q1, const float* q0, const float* q1t, const float* q0t, const float* gamma,
                               const float* beta, const float* e_stat, const int32_t* seg_ptr, const int32_t* seg_node,
                               const int32_t* src, int64_t n_seg, int H, float* GM, float* GMt, float* GP, float* GPt,
                               float* gb_partial, float* ln_partial, float* gm_amax2, float* gp_amax2, alignn_stream_t stream) {
    if (!ln_h_ok(H) || n_seg < 0 || n_seg > INT32_MAX || !GY || !GYt || !e_stat || !ln_partial) return (int)hipErrorInvalidValue;
    if (n_seg == 0) return 0;
    hipLaunchKernelGGL(egc_dual_bwd_dst_ln_kernel, dim3(seg_blocks(n_seg)), dim3(kT), 0, (hipStream_t)stream, GY, GYt, M, Mt, P, Pt, q1,
                       q0, q1t, q0t, gamma, beta, e_stat, seg_ptr, seg_node, src, (int)n_seg, H, GM, GMt, GP, GPt, gb_partial,
                       ln_partial, gm_amax2, gp_amax2);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
