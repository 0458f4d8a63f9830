// Dual-number (value + directional derivative) kernels of the LayerNorm-flavoured convolution stack, forward and
// reverse: what training THROUGH the forces needs (ALIGNNAtomWise with calculate_gradient=True:
// alignn/models/alignn_atomwise.py:512-565 takes pair forces with autograd.grad(create_graph=True), alignn/train.py:387
// differentiates the force / stress loss through them).
//
// With w_e = dL/d(pair force of bond e) the force + stress part of the loss is  sum_e w_e . f_e(theta),
// f_e = -dE_tot/dr_e, i.e. MINUS the directional derivative D_w E_tot of the total energy along the bond-vector
// displacement w.  Its parameter gradient is therefore the ordinary reverse-mode gradient of a forward pass that
// carries, next to every activation p, its tangent pdot = D_w p ("dual numbers").  These kernels are the
// non-linear pieces of that dual pass; every matrix product in it is an ordinary projection (the tangent of a Linear
// layer is the same Linear layer without bias), so the MFMA kernels of gemm_x6.hip / gemm_f32.hip are reused as is.
//
//   ln_silu_dual_fwd / _bwd     y = r + silu(LN(x)),  ydot = rdot + d[silu o LN](x) . xdot   and its reverse
//   egc_gate_dual_fwd           the gate pass (u_add_v, sigmoid, the two segment sums, h = S1/(S0+eps)) on duals
//   egc_node_dual_bwd           reverse of the node-level quotient
//   egc_dual_bwd_dst / _src     reverse of the gate pass: destination-ordered and source-ordered halves
//
// Layout as in conv.hip / norm.hip: one wavefront per row (LayerNorm) or per segment (gate), lane l owns features
// [4l, 4l+4) of each 256-feature chunk; no atomics; fixed summation orders.  "t" suffixes are tangents.
#include "common.h"
#include "../../include/alignn_hip.h"

namespace {

constexpr int kW = 4, kT = kW * ALIGNN_WAVE, kMaxBlocks = 1024;

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float hsum4(float4 a) { return (a.x + a.y) + (a.z + a.w); }
__device__ __forceinline__ float sig_f(float x) { return fast_sigmoid(x); }
// silu'(z), silu''(z)
__device__ __forceinline__ void dsilu2(float z, float& d1, float& d2) {
    const float s = sig_f(z), sp = s * (1.0f - s);
    d1 = s + z * sp;
    d2 = sp * (2.0f + z * (1.0f - 2.0f * s));
}
inline int row_blocks(int64_t rows) {
    int64_t b = (rows + kW - 1) / kW;
    if (b < 1) b = 1;
    if (b > kMaxBlocks) b = kMaxBlocks;
    return (int)b;
}

// two amax slots (value, tangent) committed by every thread of the workgroup
__device__ __forceinline__ void amax2_commit(float a, float b, float* amax2) {
    if (amax2 == nullptr) return;
    block_amax_commit(a, amax2);
    block_amax_commit(b, amax2 + 1);
}

// ---------------------------------------------------------------------------------------------
// LayerNorm + SiLU (+ residual) on duals.  x-hat = (x - mean) rstd;  t-hat = rstd (t - mean(t) - x-hat mean(x-hat t))
// ---------------------------------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(kT) void ln_silu_dual_fwd_kernel(
    const float* __restrict__ X, const float* __restrict__ Xt, int64_t ldx, const float* __restrict__ R,
    const float* __restrict__ Rt, int64_t ldr, const float* __restrict__ gamma, const float* __restrict__ beta,
    float eps, float* __restrict__ Y, float* __restrict__ Yt, int64_t ldy, float* __restrict__ stats, int64_t rows,
    int F, float* __restrict__ amax2) {
    const int lane = threadIdx.x & 63;
    const int64_t w0 = (int64_t)blockIdx.x * kW + (threadIdx.x >> 6), stride = (int64_t)gridDim.x * kW;
    const float inv_f = 1.0f / (float)F;
    float am = 0.0f, amt = 0.0f;
    for (int64_t r = w0; r < rows; r += stride) {
        float4 x[NC], t[NC];
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int f = c * 256 + 4 * lane;
            x[c] = f < F ? f4_ld(X + r * ldx + f) : f4_zero();
            t[c] = f < F ? f4_ld(Xt + r * ldx + f) : f4_zero();
            s += hsum4(x[c]);
        }
        const float mean = wsum(s) * inv_f;
        float v = 0.0f, st = 0.0f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int f = c * 256 + 4 * lane;
            if (f < F) {
                x[c] = make_float4(x[c].x - mean, x[c].y - mean, x[c].z - mean, x[c].w - mean);
                v += hsum4(f4_mul(x[c], x[c]));
                st += hsum4(t[c]);
            }
        }
        const float rstd = 1.0f / sqrtf(wsum(v) * inv_f + eps);
        const float m1 = wsum(st) * inv_f;
        float sxt = 0.0f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int f = c * 256 + 4 * lane;
            if (f < F) {
                x[c] = f4_scale(x[c], rstd);  // x-hat
                sxt += hsum4(f4_mul(x[c], t[c]));
            }
        }
        const float m2 = wsum(sxt) * inv_f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int f = c * 256 + 4 * lane;
            if (f < F) {
                const float4 g = f4_ld(gamma + f), b = f4_ld(beta + f);
                float4 o, ot;
#define ALIGNN_LN_DUAL(q)                                            \
    {                                                                \
        const float th = rstd * (t[c].q - m1 - x[c].q * m2);         \
        const float z = fmaf(x[c].q, g.q, b.q), zt = g.q * th;       \
        const float sg = sig_f(z);                                   \
        o.q = z * sg;                                                \
        ot.q = (sg + z * sg * (1.0f - sg)) * zt;                     \
    }
                ALIGNN_LN_DUAL(x) ALIGNN_LN_DUAL(y) ALIGNN_LN_DUAL(z) ALIGNN_LN_DUAL(w)
#undef ALIGNN_LN_DUAL
                if (R) {
                    if (Y) o = f4_add(o, f4_ld(R + r * ldr + f));
                    ot = f4_add(ot, f4_ld(Rt + r * ldr + f));
                }
                if (Y) {  // (NULL: the caller holds the value output already - tangent-only forward, see ff2.py)
                    f4_st(Y + r * ldy + f, o);
                    am = fmaxf(am, f4_absmax(o));
                }
                f4_st(Yt + r * ldy + f, ot);
                amt = fmaxf(amt, f4_absmax(ot));
            }
        }
        if (stats && lane == 0) {
            stats[2 * r] = mean;
            stats[2 * r + 1] = rstd;
        }
    }
    amax2_commit(am, amt, amax2);
}

// reverse: (gy, gyt) -> (gx, gxt) and slab partials of dbeta (row 0) / dgamma (row 1)
// NODE: the rows are the node pre-activations of an edge-gated convolution - the reverse of the node-level quotient
// (egc_node_dual_bwd_kernel's arithmetic) leaves with the gradient: (g, gt) -> Q1, Q0, Q1t, Q0t, one launch instead of two.
template <int NC, bool NODE>
__global__ __launch_bounds__(kT) void ln_silu_dual_bwd_kernel(
    const float* __restrict__ GY, const float* __restrict__ GYt, int64_t ldg, const float* __restrict__ X,
    const float* __restrict__ Xt, int64_t ldx, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ stats, float* __restrict__ GX, float* __restrict__ GXt, int64_t ldo,
    float* __restrict__ partial, int64_t rows, int F, float* __restrict__ amax2, const float* __restrict__ S0,
    const float* __restrict__ HH, const float* __restrict__ S0t, const float* __restrict__ HHt, float* __restrict__ Q1,
    float* __restrict__ Q0, float* __restrict__ Q1t, float* __restrict__ Q0t) {
    __shared__ float4 sh[2][kW][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t w0 = (int64_t)blockIdx.x * kW + wave, stride = (int64_t)gridDim.x * kW;
    const float inv_f = 1.0f / (float)F;
    float am = 0.0f, amt = 0.0f;
    float4 dg[NC], db[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) dg[c] = db[c] = f4_zero();
    for (int64_t r = w0; r < rows; r += stride) {
        const float mean = stats[2 * r], rstd = stats[2 * r + 1];
        float4 xh[NC], t[NC], a[NC], b[NC];
        float st = 0.0f, sxt = 0.0f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int f = c * 256 + 4 * lane;
            xh[c] = t[c] = a[c] = b[c] = f4_zero();
            if (f < F) {
                const float4 x = f4_ld(X + r * ldx + f);
                t[c] = f4_ld(Xt + r * ldx + f);
                xh[c] = make_float4((x.x - mean) * rstd, (x.y - mean) * rstd, (x.z - mean) * rstd, (x.w - mean) * rstd);
                st += hsum4(t[c]);
                sxt += hsum4(f4_mul(xh[c], t[c]));
            }
        }
        const float m1 = wsum(st) * inv_f, m2 = wsum(sxt) * inv_f;
        float sa = 0.0f, sax = 0.0f, sat = 0.0f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int f = c * 256 + 4 * lane;
            if (f < F) {
                const float4 g = f4_ld(gamma + f), be = f4_ld(beta + f);
                const float4 gy = f4_ld(GY + r * ldg + f), gyt = f4_ld(GYt + r * ldg + f);
#define ALIGNN_LN_DUAL_B(q)                                                          \
    {                                                                                \
        const float th = rstd * (t[c].q - m1 - xh[c].q * m2);                        \
        const float z = fmaf(xh[c].q, g.q, be.q), zt = g.q * th;                     \
        float d1, d2;                                                                \
        dsilu2(z, d1, d2);                                                           \
        const float gzt = gyt.q * d1;                /* adjoint of zdot */           \
        const float gz = gy.q * d1 + gyt.q * d2 * zt; /* adjoint of z    */          \
        db[c].q += gz;                                                               \
        dg[c].q += gz * xh[c].q + gzt * th;                                          \
        a[c].q = g.q * gzt;                          /* adjoint of t-hat */          \
        b[c].q = g.q * gz;                           /* direct adjoint of x-hat */   \
        sa += a[c].q;                                                                \
        sax += a[c].q * xh[c].q;                                                     \
        sat += a[c].q * th;                                                          \
    }
                ALIGNN_LN_DUAL_B(x) ALIGNN_LN_DUAL_B(y) ALIGNN_LN_DUAL_B(z) ALIGNN_LN_DUAL_B(w)
#undef ALIGNN_LN_DUAL_B
            }
        }
        const float A1 = wsum(sa) * inv_f, A2 = wsum(sax) * inv_f, A3 = wsum(sat) * inv_f;
        // total adjoint of x-hat: direct part - rstd (a m2 + t A2)
        float sb = 0.0f, sbx = 0.0f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int f = c * 256 + 4 * lane;
            if (f < F) {
                b[c].x -= rstd * (a[c].x * m2 + t[c].x * A2);
                b[c].y -= rstd * (a[c].y * m2 + t[c].y * A2);
                b[c].z -= rstd * (a[c].z * m2 + t[c].z * A2);
                b[c].w -= rstd * (a[c].w * m2 + t[c].w * A2);
                sb += hsum4(b[c]);
                sbx += hsum4(f4_mul(b[c], xh[c]));
            }
        }
        const float B1 = wsum(sb) * inv_f, B2 = wsum(sbx) * inv_f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int f = c * 256 + 4 * lane;
            if (f < F) {
                float4 gx, gxt;
#define ALIGNN_LN_DUAL_O(q)                                                              \
    gxt.q = rstd * (a[c].q - A1 - xh[c].q * A2);                                          \
    gx.q = rstd * (b[c].q - B1 - xh[c].q * B2) - rstd * xh[c].q * A3;
                ALIGNN_LN_DUAL_O(x) ALIGNN_LN_DUAL_O(y) ALIGNN_LN_DUAL_O(z) ALIGNN_LN_DUAL_O(w)
#undef ALIGNN_LN_DUAL_O
                f4_st(GX + r * ldo + f, gx);
                f4_st(GXt + r * ldo + f, gxt);
                am = fmaxf(am, f4_absmax(gx));
                amt = fmaxf(amt, f4_absmax(gxt));
                if (NODE) {
                    const float4 s0 = f4_ld(S0 + r * F + f), h = f4_ld(HH + r * F + f);
                    const float4 s0t = f4_ld(S0t + r * F + f), ht = f4_ld(HHt + r * F + f);
                    float4 q1, q0, q1t, q0t;
#define ALIGNN_QN(c)                                    \
    {                                                   \
        const float d = s0.c + ALIGNN_EPS_GATE;         \
        q1t.c = gxt.c / d;                              \
        q0t.c = -q1t.c * h.c;                           \
        const float gh = gx.c - q1t.c * s0t.c;          \
        q1.c = gh / d;                                  \
        q0.c = -q1t.c * ht.c - q1.c * h.c;              \
    }
                    ALIGNN_QN(x) ALIGNN_QN(y) ALIGNN_QN(z) ALIGNN_QN(w)
#undef ALIGNN_QN
                    f4_st(Q1 + r * F + f, q1);
                    f4_st(Q0 + r * F + f, q0);
                    f4_st(Q1t + r * F + f, q1t);
                    f4_st(Q0t + r * F + f, q0t);
                }
            }
        }
    }
    amax2_commit(am, amt, amax2);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        sh[0][wave][lane] = db[c];
        sh[1][wave][lane] = dg[c];
        __syncthreads();
        const int f = c * 256 + 4 * lane;
        if (wave == 0 && f < F) {
            float4 s0 = sh[0][0][lane], s1 = sh[1][0][lane];
#pragma unroll
            for (int w = 1; w < kW; ++w) {
                s0 = f4_add(s0, sh[0][w][lane]);
                s1 = f4_add(s1, sh[1][w][lane]);
            }
            f4_st(partial + (size_t)blockIdx.x * 2 * F + f, s0);
            f4_st(partial + (size_t)blockIdx.x * 2 * F + F + f, s1);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// gate pass on duals.  P = [A | Bd | Bh | Ux] rows of 4H; M holds C on entry, m = A[u] + Bd[v] + C on exit (same for
// the tangents).  Node outputs: xpre = Ux + h, xpre_t, and S0, h, S0_t, h_t for the reverse pass.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kT) void egc_gate_dual_fwd_kernel(
    const float* __restrict__ P, const float* __restrict__ Pt, float* __restrict__ M, float* __restrict__ Mt,
    const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ seg_node, const int32_t* __restrict__ src,
    int n_seg, int H, float* __restrict__ XPRE, float* __restrict__ XPREt, float* __restrict__ S0,
    float* __restrict__ HH, float* __restrict__ S0t, float* __restrict__ HHt) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t ldp = 4 * (int64_t)H;
    const int first = blockIdx.x * kW + wave, stride = gridDim.x * kW;
    for (int c0 = 0; c0 < H; c0 += 4 * ALIGNN_WAVE) {
        const int f = c0 + 4 * lane;
        if (f >= H) continue;
        for (int s = first; s < n_seg; s += stride) {
            const int beg = seg_ptr[s], end = seg_ptr[s + 1];
            const int i = seg_node ? seg_node[s] : s;
            const float4 bd = f4_ld(P + (int64_t)i * ldp + H + f), bdt = f4_ld(Pt + (int64_t)i * ldp + H + f);
            float4 s1 = f4_zero(), s0 = f4_zero(), s1t = f4_zero(), s0t = f4_zero();
            for (int e = beg; e < end; ++e) {
                const int64_t u = src[e];
                const float4 m = f4_add(f4_add(f4_ld(P + u * ldp + f), bd), f4_ld(M + (int64_t)e * H + f));
                const float4 mt = f4_add(f4_add(f4_ld(Pt + u * ldp + f), bdt), f4_ld(Mt + (int64_t)e * H + f));
                const float4 bh = f4_ld(P + u * ldp + 2 * H + f), bht = f4_ld(Pt + u * ldp + 2 * H + f);
                f4_st(M + (int64_t)e * H + f, m);
                f4_st(Mt + (int64_t)e * H + f, mt);
                const float4 sg = f4_sigmoid(m);
                const float4 sgt = make_float4(sg.x * (1.0f - sg.x) * mt.x, sg.y * (1.0f - sg.y) * mt.y,
                                               sg.z * (1.0f - sg.z) * mt.z, sg.w * (1.0f - sg.w) * mt.w);
                s1 = f4_fma(sg, bh, s1);
                s0 = f4_add(s0, sg);
                s1t = f4_fma(sgt, bh, f4_fma(sg, bht, s1t));
                s0t = f4_add(s0t, sgt);
            }
            float4 h, ht;
#define ALIGNN_Q(q)                                  \
    {                                                \
        const float d = s0.q + ALIGNN_EPS_GATE;      \
        h.q = s1.q / d;                              \
        ht.q = (s1t.q - h.q * s0t.q) / d;            \
    }
            ALIGNN_Q(x) ALIGNN_Q(y) ALIGNN_Q(z) ALIGNN_Q(w)
#undef ALIGNN_Q
            f4_st(XPRE + (int64_t)i * H + f, f4_add(f4_ld(P + (int64_t)i * ldp + 3 * H + f), h));
            f4_st(XPREt + (int64_t)i * H + f, f4_add(f4_ld(Pt + (int64_t)i * ldp + 3 * H + f), ht));
            f4_st(S0 + (int64_t)i * H + f, s0);
            f4_st(HH + (int64_t)i * H + f, h);
            f4_st(S0t + (int64_t)i * H + f, s0t);
            f4_st(HHt + (int64_t)i * H + f, ht);
        }
    }
}

// The TANGENT half of the gate pass alone, for a forward whose values are already known (the force evaluation that preceded
// the dual pass computed them: M holds m, S0 / HH the node sums): Mt holds Ct on entry and mt on exit; writes xpre_t, S0_t,
// h_t.  One read of m and one read + write of mt per row instead of two of each.  Same formulas as egc_gate_dual_fwd_kernel.
__global__ __launch_bounds__(kT) void egc_gate_dual_tan_kernel(
    const float* __restrict__ P, const float* __restrict__ Pt, const float* __restrict__ M, float* __restrict__ Mt,
    const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ seg_node, const int32_t* __restrict__ src,
    int n_seg, int H, float* __restrict__ XPREt, const float* __restrict__ S0, const float* __restrict__ HH,
    float* __restrict__ S0t, float* __restrict__ HHt) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t ldp = 4 * (int64_t)H;
    const int first = blockIdx.x * kW + wave, stride = gridDim.x * kW;
    for (int c0 = 0; c0 < H; c0 += 4 * ALIGNN_WAVE) {
        const int f = c0 + 4 * lane;
        if (f >= H) continue;
        for (int s = first; s < n_seg; s += stride) {
            const int beg = seg_ptr[s], end = seg_ptr[s + 1];
            const int i = seg_node ? seg_node[s] : s;
            const float4 bdt = f4_ld(Pt + (int64_t)i * ldp + H + f);
            float4 s1t = f4_zero(), s0t = f4_zero();
            for (int e = beg; e < end; ++e) {
                const int64_t u = src[e];
                const float4 m = f4_ld(M + (int64_t)e * H + f);
                const float4 mt = f4_add(f4_add(f4_ld(Pt + u * ldp + f), bdt), f4_ld(Mt + (int64_t)e * H + f));
                const float4 bh = f4_ld(P + u * ldp + 2 * H + f), bht = f4_ld(Pt + u * ldp + 2 * H + f);
                f4_st(Mt + (int64_t)e * H + f, mt);
                const float4 sg = f4_sigmoid(m);
                const float4 sgt = make_float4(sg.x * (1.0f - sg.x) * mt.x, sg.y * (1.0f - sg.y) * mt.y,
                                               sg.z * (1.0f - sg.z) * mt.z, sg.w * (1.0f - sg.w) * mt.w);
                s1t = f4_fma(sgt, bh, f4_fma(sg, bht, s1t));
                s0t = f4_add(s0t, sgt);
            }
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
}

// reverse of h = S1/D, ht = (S1t - h S0t)/D, D = S0 + eps:  (g_h, g_ht) -> adjoints Q1, Q0, Q1t, Q0t of S1, S0, S1t, S0t
__global__ __launch_bounds__(256) void egc_node_dual_bwd_kernel(
    const float* __restrict__ G, const float* __restrict__ Gt, int64_t ldg, const float* __restrict__ S0,
    const float* __restrict__ HH, const float* __restrict__ S0t, const float* __restrict__ HHt,
    float* __restrict__ Q1, float* __restrict__ Q0, float* __restrict__ Q1t, float* __restrict__ Q0t, int64_t n, int H) {
    const int Q = H >> 2;
    const RowQuad rq(Q);
    const int64_t total = n * Q;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t r;
        int q;
        rq.split(i, total, r, q);
        const float4 g = f4_ld(G + r * ldg + q * 4), gt = f4_ld(Gt + r * ldg + q * 4);
        const float4 s0 = f4_ld(S0 + r * H + q * 4), h = f4_ld(HH + r * H + q * 4);
        const float4 s0t = f4_ld(S0t + r * H + q * 4), ht = f4_ld(HHt + r * H + q * 4);
        float4 q1, q0, q1t, q0t;
#define ALIGNN_QB(c)                                   \
    {                                                  \
        const float d = s0.c + ALIGNN_EPS_GATE;        \
        q1t.c = gt.c / d;                              \
        q0t.c = -q1t.c * h.c;                          \
        const float gh = g.c - q1t.c * s0t.c;          \
        q1.c = gh / d;                                 \
        q0.c = -q1t.c * ht.c - q1.c * h.c;             \
    }
        ALIGNN_QB(x) ALIGNN_QB(y) ALIGNN_QB(z) ALIGNN_QB(w)
#undef ALIGNN_QB
        f4_st(Q1 + r * H + q * 4, q1);
        f4_st(Q0 + r * H + q * 4, q0);
        f4_st(Q1t + r * H + q * 4, q1t);
        f4_st(Q0t + r * H + q * 4, q0t);
    }
}

// reverse of the gate pass, destination order: GM / GMt (adjoints of m / mt = of C / Ct), the Bd blocks of GP / GPt
// (segment sums) and the column-sum slabs of GM (edge_gate bias gradient).  GL / GLt: adjoints of m / mt coming from the
// edge LayerNorm branch (both null when the edge output is dead).
__global__ __launch_bounds__(kT) void egc_dual_bwd_dst_kernel(
    const float* __restrict__ GL, const float* __restrict__ GLt, const float* __restrict__ M,
    const float* __restrict__ Mt, const float* __restrict__ P, const float* __restrict__ Pt,
    const float* __restrict__ Q1, const float* __restrict__ Q0, const float* __restrict__ Q1t,
    const float* __restrict__ Q0t, const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ seg_node,
    const int32_t* __restrict__ src, int n_seg, int H, float* __restrict__ GM, float* __restrict__ GMt,
    float* __restrict__ GP, float* __restrict__ GPt, float* __restrict__ gb_partial, float* __restrict__ gm_amax2,
    float* __restrict__ gp_amax2) {
    __shared__ float4 sh[kW][ALIGNN_WAVE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t ldp = 4 * (int64_t)H;
    const int first = blockIdx.x * kW + wave, stride = gridDim.x * kW;
    float am = 0.0f, amt = 0.0f, pam = 0.0f, pamt = 0.0f;
    for (int c0 = 0; c0 < H; c0 += 4 * ALIGNN_WAVE) {
        const int f = c0 + 4 * lane;
        const bool active = f < H;
        float4 gb = f4_zero();
        if (active) {
            for (int s = first; s < n_seg; s += stride) {
                const int beg = seg_ptr[s], end = seg_ptr[s + 1];
                const int i = seg_node ? seg_node[s] : s;
                const float4 q1 = f4_ld(Q1 + (int64_t)i * H + f), q0 = f4_ld(Q0 + (int64_t)i * H + f);
                const float4 q1t = f4_ld(Q1t + (int64_t)i * H + f), q0t = f4_ld(Q0t + (int64_t)i * H + f);
                float4 gbd = f4_zero(), gbdt = f4_zero();
                for (int e = beg; e < end; ++e) {
                    const int64_t u = src[e];
                    const float4 m = f4_ld(M + (int64_t)e * H + f), mt = f4_ld(Mt + (int64_t)e * H + f);
                    const float4 bh = f4_ld(P + u * ldp + 2 * H + f), bht = f4_ld(Pt + u * ldp + 2 * H + f);
                    float4 gm = GL ? f4_ld(GL + (int64_t)e * H + f) : f4_zero();
                    float4 gmt = GL ? f4_ld(GLt + (int64_t)e * H + f) : f4_zero();
#define ALIGNN_GD(c)                                                            \
    {                                                                           \
        const float sg = sig_f(m.c), sp = sg * (1.0f - sg);                     \
        const float gs = q1.c * bh.c + q0.c + q1t.c * bht.c; /* adj. sigma */   \
        const float gst = q1t.c * bh.c + q0t.c;              /* adj. sigma-dot */ \
        gm.c += gs * sp + gst * sp * (1.0f - 2.0f * sg) * mt.c;                 \
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
                f4_st(GP + (int64_t)i * ldp + H + f, gbd);
                f4_st(GPt + (int64_t)i * ldp + H + f, gbdt);
                pam = fmaxf(pam, f4_absmax(gbd));
                pamt = fmaxf(pamt, f4_absmax(gbdt));
                gb = f4_add(gb, gbd);
            }
        }
        if (gb_partial) {
            sh[wave][lane] = gb;
            __syncthreads();
            if (wave == 0 && active) {
                float4 a = sh[0][lane];
#pragma unroll
                for (int w = 1; w < kW; ++w) a = f4_add(a, sh[w][lane]);
                f4_st(gb_partial + (size_t)blockIdx.x * H + f, a);
            }
            __syncthreads();
        }
    }
    amax2_commit(am, amt, gm_amax2);
    amax2_commit(pam, pamt, gp_amax2);
}

// source order: the A and Bh blocks of GP / GPt as segment sums over the out-slots of every node
__global__ __launch_bounds__(kT) void egc_dual_bwd_src_kernel(
    const float* __restrict__ GM, const float* __restrict__ GMt, const float* __restrict__ M,
    const float* __restrict__ Mt, const float* __restrict__ Q1, const float* __restrict__ Q1t,
    const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ out_slot, const int32_t* __restrict__ dst,
    int n_nodes, int H, float* __restrict__ GP, float* __restrict__ GPt, float* __restrict__ gp_amax2) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t ldp = 4 * (int64_t)H;
    const int first = blockIdx.x * kW + wave, stride = gridDim.x * kW;
    float pam = 0.0f, pamt = 0.0f;
    for (int c0 = 0; c0 < H; c0 += 4 * ALIGNN_WAVE) {
        const int f = c0 + 4 * lane;
        if (f < H) {
            for (int j = first; j < n_nodes; j += stride) {
                const int beg = out_ptr[j], end = out_ptr[j + 1];
                float4 ga = f4_zero(), gat = f4_zero(), gbh = f4_zero(), gbht = f4_zero();
                for (int k = beg; k < end; ++k) {
                    const int64_t slot = out_slot[k];
                    const int64_t v = dst[slot];
                    const float4 m = f4_ld(M + slot * H + f), mt = f4_ld(Mt + slot * H + f);
                    const float4 q1 = f4_ld(Q1 + v * H + f), q1t = f4_ld(Q1t + v * H + f);
                    ga = f4_add(ga, f4_ld(GM + slot * H + f));
                    gat = f4_add(gat, f4_ld(GMt + slot * H + f));
#define ALIGNN_GS(c)                                              \
    {                                                             \
        const float sg = sig_f(m.c), sgt = sg * (1.0f - sg) * mt.c; \
        gbh.c += sg * q1.c + sgt * q1t.c;                         \
        gbht.c += sg * q1t.c;                                     \
    }
                    ALIGNN_GS(x) ALIGNN_GS(y) ALIGNN_GS(z) ALIGNN_GS(w)
#undef ALIGNN_GS
                }
                f4_st(GP + (int64_t)j * ldp + f, ga);
                f4_st(GPt + (int64_t)j * ldp + f, gat);
                f4_st(GP + (int64_t)j * ldp + 2 * H + f, gbh);
                f4_st(GPt + (int64_t)j * ldp + 2 * H + f, gbht);
                pam = fmaxf(pam, fmaxf(f4_absmax(ga), f4_absmax(gbh)));
                pamt = fmaxf(pamt, fmaxf(f4_absmax(gat), f4_absmax(gbht)));
            }
        }
    }
    amax2_commit(pam, pamt, gp_amax2);
}

// The two passes above as ONE for line graphs whose blocks are dense and source-sorted (CSRGraph.dense_max_src > 0; the
// structure and the index arithmetic of egc_bwd_lg_dense_kernel, csrc/conv.hip): the workgroup of centre atom j walks its
// (sources x segments) block segment by segment; wave w owns the sources q = w, w + 4, ... and keeps their Bh / Bh-dot rows
// and their four source-side sums (g_A, g_A-dot, g_Bh, g_Bh-dot) in registers for the whole block, a segment's four adjoint
// rows Q are loaded once per segment, g_Bd / g_Bd-dot of the segment are the fixed-order sums of the four waves' partial
// sums.  Every T-row of M, Mt, GL, GLt is read once and GM, GMt written once: 6 row passes instead of the 10 of
// egc_dual_bwd_dst + egc_dual_bwd_src (which re-reads M, Mt, GM, GMt by source).  Same values; sums in a different but
// fixed order.
constexpr int kDualDenseK = 4;  // sources per wave and pass: 16 per workgroup

template <bool HAS_GL, bool STREAM>
__global__ __launch_bounds__(kT) void egc_dual_bwd_lg_dense_kernel(
    const float* __restrict__ GL, const float* __restrict__ GLt, const float* __restrict__ M,
    const float* __restrict__ Mt, const float* __restrict__ P, const float* __restrict__ Pt,
    const float* __restrict__ Q1, const float* __restrict__ Q0, const float* __restrict__ Q1t,
    const float* __restrict__ Q0t, const int32_t* __restrict__ grp_seg_ptr, const int32_t* __restrict__ grp_src_ptr,
    const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ seg_node, int H, float* __restrict__ GM,
    float* __restrict__ GMt, float* __restrict__ GP, float* __restrict__ GPt, float* __restrict__ gb_partial,
    float* __restrict__ gm_amax2, float* __restrict__ gp_amax2) {
    constexpr int KMAX = kDualDenseK;
    __shared__ float4 sh[2][2][kW][ALIGNN_WAVE];  // [buffer][value | tangent][wave][lane]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t ldp = 4 * (int64_t)H;
    const int j = blockIdx.x;
    const int p_beg = grp_src_ptr[j], n_src = grp_src_ptr[j + 1] - p_beg;
    const int s_beg = grp_seg_ptr[j], s_end = grp_seg_ptr[j + 1];
    float am = 0.0f, amt = 0.0f, pam = 0.0f, pamt = 0.0f;
    for (int c0 = 0; c0 < H; c0 += 4 * ALIGNN_WAVE) {
        const int f = c0 + 4 * lane;
        const bool active = f < H;
        float4 gb = f4_zero();
        for (int qb = 0; qb < n_src || qb == 0; qb += kW * KMAX) {
            float4 bh[KMAX], bht[KMAX], ga[KMAX], gat[KMAX], gbh[KMAX], gbht[KMAX];
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const int q = qb + wave + kW * k;
                const bool have = active && q < n_src;
                bh[k] = have ? f4_ld(P + (int64_t)(p_beg + q) * ldp + 2 * H + f) : f4_zero();
                bht[k] = have ? f4_ld(Pt + (int64_t)(p_beg + q) * ldp + 2 * H + f) : f4_zero();
                ga[k] = gat[k] = gbh[k] = gbht[k] = f4_zero();
            }
            for (int s = s_beg; s < s_end; ++s) {
                const int i = seg_node ? seg_node[s] : s;
                const int e0 = seg_ptr[s];
                int self_q = i - p_beg;  // the segment's own bond among the sources (self image) is not listed
                if (self_q < 0 || self_q >= n_src || seg_ptr[s + 1] - e0 == n_src) self_q = -1;
                float4 gbd = f4_zero(), gbdt = f4_zero();
                if (active) {
                    const float4 q1 = f4_ld(Q1 + (int64_t)i * H + f), q0 = f4_ld(Q0 + (int64_t)i * H + f);
                    const float4 q1t = f4_ld(Q1t + (int64_t)i * H + f), q0t = f4_ld(Q0t + (int64_t)i * H + f);
                    float4 m[KMAX], mt[KMAX], gm[KMAX], gmt[KMAX];
                    int row[KMAX];
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) {
                        const int q = qb + wave + kW * k;
                        row[k] = (q < n_src && q != self_q) ? e0 + q - ((self_q >= 0 && q > self_q) ? 1 : 0) : -1;
                        if (row[k] >= 0) {
                            m[k] = f4_lds<STREAM>(M + (int64_t)row[k] * H + f);
                            mt[k] = f4_lds<STREAM>(Mt + (int64_t)row[k] * H + f);
                            gm[k] = HAS_GL ? f4_lds<STREAM>(GL + (int64_t)row[k] * H + f) : f4_zero();
                            gmt[k] = HAS_GL ? f4_lds<STREAM>(GLt + (int64_t)row[k] * H + f) : f4_zero();
                        }
                    }
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) {
                        if (row[k] >= 0) {
#define ALIGNN_GDD(c)                                                                                     \
    {                                                                                                     \
        const float sg = sig_f(m[k].c), sp = sg * (1.0f - sg);                                            \
        const float gs = q1.c * bh[k].c + q0.c + q1t.c * bht[k].c; /* adjoint of sigma */                  \
        const float gst = q1t.c * bh[k].c + q0t.c;                 /* adjoint of sigma-dot */              \
        gm[k].c += gs * sp + gst * sp * (1.0f - 2.0f * sg) * mt[k].c;                                      \
        gmt[k].c += gst * sp;                                                                             \
        const float sgt = sp * mt[k].c;                                                                   \
        gbh[k].c += sg * q1.c + sgt * q1t.c;                                                              \
        gbht[k].c += sg * q1t.c;                                                                          \
    }
                            ALIGNN_GDD(x) ALIGNN_GDD(y) ALIGNN_GDD(z) ALIGNN_GDD(w)
#undef ALIGNN_GDD
                            f4_sts<STREAM>(GM + (int64_t)row[k] * H + f, gm[k]);
                            f4_sts<STREAM>(GMt + (int64_t)row[k] * H + f, gmt[k]);
                            am = fmaxf(am, f4_absmax(gm[k]));
                            amt = fmaxf(amt, f4_absmax(gmt[k]));
                            ga[k] = f4_add(ga[k], gm[k]);
                            gat[k] = f4_add(gat[k], gmt[k]);
                            gbd = f4_add(gbd, gm[k]);
                            gbdt = f4_add(gbdt, gmt[k]);
                        }
                    }
                }
                // g_Bd[s], g_Bd-dot[s]: the four waves' partial sums in wave order (double-buffered, one barrier per segment)
                float4(*buf)[kW][ALIGNN_WAVE] = sh[(s - s_beg) & 1];
                buf[0][wave][lane] = gbd;
                buf[1][wave][lane] = gbdt;
                __syncthreads();
                if (wave == 0 && active) {
                    float4 a = buf[0][0][lane], at = buf[1][0][lane];
#pragma unroll
                    for (int w = 1; w < kW; ++w) {
                        a = f4_add(a, buf[0][w][lane]);
                        at = f4_add(at, buf[1][w][lane]);
                    }
                    gb = f4_add(gb, a);
                    float* out = GP + (int64_t)i * ldp + H + f;
                    float* outt = GPt + (int64_t)i * ldp + H + f;
                    if (qb > 0) {  // (written by this very thread in the previous pass)
                        a = f4_add(f4_ld(out), a);
                        at = f4_add(f4_ld(outt), at);
                    }
                    f4_st(out, a);
                    f4_st(outt, at);
                    pam = fmaxf(pam, f4_absmax(a));
                    pamt = fmaxf(pamt, f4_absmax(at));
                }
            }
            if (active) {
#pragma unroll
                for (int k = 0; k < KMAX; ++k) {
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
            __syncthreads();  // sh is reused by the next pass / feature panel
        }
        if (active && gb_partial && wave == 0) f4_st(gb_partial + (size_t)blockIdx.x * H + f, gb);
    }
    amax2_commit(am, amt, gm_amax2);
    amax2_commit(pam, pamt, gp_amax2);
}

inline bool feat_ok(int F) { return F >= 4 && (F & 3) == 0 && F <= 1024; }
inline bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

int alignn_dual_slabs(int64_t rows) { return row_blocks(rows); }

int alignn_ln_silu_dual_fwd(const float* X, const float* Xt, int64_t ldx, const float* R, const float* Rt, int64_t ldr,
                            const float* gamma, const float* beta, float eps, float* Y, float* Yt, int64_t ldy,
                            float* stats, int64_t rows, int F, float* amax2, alignn_stream_t stream) {
    if (!feat_ok(F) || (R == nullptr) != (Rt == nullptr)) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    dim3 grid(row_blocks(rows)), block(kT);
    hipStream_t st = (hipStream_t)stream;
#define ALIGNN_CASE(NC_)                                                                                              \
    hipLaunchKernelGGL((ln_silu_dual_fwd_kernel<NC_>), grid, block, 0, st, X, Xt, ldx, R, Rt, ldr, gamma, beta, eps, Y, \
                       Yt, ldy, stats, rows, F, amax2)
    switch ((F + 255) / 256) {
        case 1: ALIGNN_CASE(1); break;
        case 2: ALIGNN_CASE(2); break;
        case 3: ALIGNN_CASE(3); break;
        default: ALIGNN_CASE(4); break;
    }
#undef ALIGNN_CASE
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_ln_silu_dual_bwd(const float* GY, const float* GYt, int64_t ldg, const float* X, const float* Xt, int64_t ldx,
                            const float* gamma, const float* beta, const float* stats, float* GX, float* GXt,
                            int64_t ldo, float* partial, int64_t rows, int F, float* amax2, alignn_stream_t stream) {
    if (!feat_ok(F) || rows < 0) return (int)hipErrorInvalidValue;
    // rows == 0 still launches its one workgroup: alignn_dual_slabs(0) == 1 and the finalize pass reads that slab - it must hold
    // zeros, not whatever the workspace held (same rule as alignn_ln_silu_bwd / _node in norm.hip)
    dim3 grid(row_blocks(rows)), block(kT);
    hipStream_t st = (hipStream_t)stream;
#define ALIGNN_CASE(NC_)                                                                                                    \
    hipLaunchKernelGGL((ln_silu_dual_bwd_kernel<NC_, false>), grid, block, 0, st, GY, GYt, ldg, X, Xt, ldx, gamma, beta, stats, GX, \
                       GXt, ldo, partial, rows, F, amax2, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr)
    switch ((F + 255) / 256) {
        case 1: ALIGNN_CASE(1); break;
        case 2: ALIGNN_CASE(2); break;
        case 3: ALIGNN_CASE(3); break;
        default: ALIGNN_CASE(4); break;
    }
#undef ALIGNN_CASE
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_ln_silu_dual_bwd_node(const float* GY, const float* GYt, int64_t ldg, const float* X, const float* Xt, int64_t ldx,
                                 const float* gamma, const float* beta, const float* stats, float* GX, float* GXt, int64_t ldo,
                                 float* partial, int64_t rows, int F, float* amax2, const float* s0, const float* hh,
                                 const float* s0t, const float* hht, float* q1, float* q0, float* q1t, float* q0t,
                                 alignn_stream_t stream) {
    if (!feat_ok(F) || rows < 0 || !s0 || !hh || !s0t || !hht || !q1 || !q0 || !q1t || !q0t) return (int)hipErrorInvalidValue;
    // (rows == 0: one workgroup writes the zero slab, see alignn_ln_silu_dual_bwd)
    dim3 grid(row_blocks(rows)), block(kT);
    hipStream_t st = (hipStream_t)stream;
#define ALIGNN_CASE(NC_)                                                                                                   \
    hipLaunchKernelGGL((ln_silu_dual_bwd_kernel<NC_, true>), grid, block, 0, st, GY, GYt, ldg, X, Xt, ldx, gamma, beta, stats, GX, \
                       GXt, ldo, partial, rows, F, amax2, s0, hh, s0t, hht, q1, q0, q1t, q0t)
    switch ((F + 255) / 256) {
        case 1: ALIGNN_CASE(1); break;
        case 2: ALIGNN_CASE(2); break;
        case 3: ALIGNN_CASE(3); break;
        default: ALIGNN_CASE(4); break;
    }
#undef ALIGNN_CASE
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_egc_gate_dual_fwd(const float* P, const float* Pt, float* M, float* Mt, const int32_t* seg_ptr,
                             const int32_t* seg_node, const int32_t* src, int64_t n, int64_t m, int H, float* xpre,
                             float* xpre_t, float* s0, float* hh, float* s0t, float* hht, alignn_stream_t stream) {
    if (!feat_ok(H) || !a16(P) || !a16(M)) return (int)hipErrorInvalidValue;
    (void)m;
    if (n == 0) return 0;
    hipLaunchKernelGGL(egc_gate_dual_fwd_kernel, dim3(row_blocks(n)), dim3(kT), 0, (hipStream_t)stream, P, Pt, M, Mt,
                       seg_ptr, seg_node, src, (int)n, H, xpre, xpre_t, s0, hh, s0t, hht);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_egc_gate_dual_fwd_tangent(const float* P, const float* Pt, const float* M, float* Mt, const int32_t* seg_ptr,
                                     const int32_t* seg_node, const int32_t* src, int64_t n, int64_t m, int H, float* xpre_t,
                                     const float* s0, const float* hh, float* s0t, float* hht, alignn_stream_t stream) {
    if (!feat_ok(H)) return (int)hipErrorInvalidValue;
    (void)m;
    if (n == 0) return 0;
    hipLaunchKernelGGL(egc_gate_dual_tan_kernel, dim3(row_blocks(n)), dim3(kT), 0, (hipStream_t)stream, P, Pt, M, Mt, seg_ptr,
                       seg_node, src, (int)n, H, xpre_t, s0, hh, s0t, hht);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_egc_node_dual_bwd(const float* g, const float* gt, int64_t ldg, const float* s0, const float* hh,
                             const float* s0t, const float* hht, float* q1, float* q0, float* q1t, float* q0t, int64_t n,
                             int H, alignn_stream_t stream) {
    if (!feat_ok(H)) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    int64_t blocks = (n * (H / 4) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(egc_node_dual_bwd_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, g, gt, ldg, s0, hh,
                       s0t, hht, q1, q0, q1t, q0t, n, H);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_egc_dual_bwd_dst(const float* GL, const float* GLt, const float* M, const float* Mt, const float* P,
                            const float* Pt, const float* q1, const float* q0, const float* q1t, const float* q0t,
                            const int32_t* seg_ptr, const int32_t* seg_node, const int32_t* src, int64_t n, int H,
                            float* GM, float* GMt, float* GP, float* GPt, float* gb_partial, float* gm_amax2,
                            float* gp_amax2, alignn_stream_t stream) {
    if (!feat_ok(H) || (GL == nullptr) != (GLt == nullptr)) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    hipLaunchKernelGGL(egc_dual_bwd_dst_kernel, dim3(row_blocks(n)), dim3(kT), 0, (hipStream_t)stream, GL, GLt, M, Mt, P,
                       Pt, q1, q0, q1t, q0t, seg_ptr, seg_node, src, (int)n, H, GM, GMt, GP, GPt, gb_partial, gm_amax2,
                       gp_amax2);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_egc_dual_bwd_src(const float* GM, const float* GMt, const float* M, const float* Mt, const float* q1,
                            const float* q1t, const int32_t* out_ptr, const int32_t* out_slot, const int32_t* dst,
                            int64_t n, int H, float* GP, float* GPt, float* gp_amax2, alignn_stream_t stream) {
    if (!feat_ok(H)) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    hipLaunchKernelGGL(egc_dual_bwd_src_kernel, dim3(row_blocks(n)), dim3(kT), 0, (hipStream_t)stream, GM, GMt, M, Mt, q1,
                       q1t, out_ptr, out_slot, dst, (int)n, H, GP, GPt, gp_amax2);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_egc_dual_bwd_lg_dense(const float* GL, const float* GLt, const float* M, const float* Mt, const float* P,
                                 const float* Pt, const float* q1, const float* q0, const float* q1t, const float* q0t,
                                 int64_t m_rows, const int32_t* grp_seg_ptr, const int32_t* grp_src_ptr, int64_t n_groups,
                                 const int32_t* seg_ptr, const int32_t* seg_node, int H, float* GM, float* GMt, float* GP,
                                 float* GPt, float* gb_partial, float* gm_amax2, float* gp_amax2, alignn_stream_t stream) {
    if (!feat_ok(H) || (GL == nullptr) != (GLt == nullptr) || n_groups < 0 || n_groups > INT32_MAX) return (int)hipErrorInvalidValue;
    if (n_groups == 0) return 0;
    const bool big = m_rows * (int64_t)H * 4 >= ((int64_t)128 << 20);  // the T-row tensors cannot stay in the last-level cache
    const dim3 grid((unsigned)n_groups), block(kT);
    hipStream_t st = (hipStream_t)stream;
#define ALIGNN_LAUNCH_DD(GLF, ST)                                                                                         \
    hipLaunchKernelGGL((egc_dual_bwd_lg_dense_kernel<GLF, ST>), grid, block, 0, st, GL, GLt, M, Mt, P, Pt, q1, q0, q1t, q0t, \
                       grp_seg_ptr, grp_src_ptr, seg_ptr, seg_node, H, GM, GMt, GP, GPt, gb_partial, gm_amax2, gp_amax2)
    if (GL != nullptr) {
        if (big) ALIGNN_LAUNCH_DD(true, true);
        else ALIGNN_LAUNCH_DD(true, false);
    } else {
        if (big) ALIGNN_LAUNCH_DD(false, true);
        else ALIGNN_LAUNCH_DD(false, false);
    }
#undef ALIGNN_LAUNCH_DD
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
