// Edge-gated graph convolution core: gather -> gate -> segment-sum, forward and backward.
//
// One wavefront (64 lanes) per destination segment; lane l owns features [4l, 4l+4) of each
// 256-feature chunk, so a feature row of H=256 fp32 is exactly one coalesced 1 KiB access per
// wave-instruction.  Source-node rows (A[u], Bh[u]) are gathered as whole rows, the segment sum is
// kept in registers, and the per-feature BatchNorm statistics of the edge pre-activations are
// accumulated on the fly (wave partials -> fixed-order block sum -> slab).  No atomics anywhere:
// results are bit-reproducible.
//
// Reference: EdgeGatedGraphConv.forward, alignn/models/alignn.py:78-129 (DGL u_add_v / u_mul_e+sum /
// copy_e+sum at :100,:105-108) and DGL's autograd of the same calls for the backward passes.
#include "common.h"
#include "../../include/alignn_hip.h"

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kThreads = kWavesPerBlock * ALIGNN_WAVE;
constexpr int kMaxSlabs = 1024;  // workgroups of the segment kernels (= statistic slabs they emit)

__host__ __device__ inline int egc_blocks(int64_t n_seg) {
    int64_t b = (n_seg + kWavesPerBlock - 1) / kWavesPerBlock;
    if (b < 1) b = 1;
    if (b > kMaxSlabs) b = kMaxSlabs;
    return (int)b;
}

// BatchNorm statistics as "pivot slabs" (csrc/norm.hip: col_stats_welford_kernel): every lane sums d = v - p and d^2 about
// a pivot p = the first value it sees; the four waves are merged in wave order by re-centring onto wave 0's pivot (all
// sums stay of the size of the spread - no E[x^2] - mean^2 anywhere, and no mean rounded to float32 either); the slab
// [3][H] = (pivot, S, SS) goes out with its row count: partial[gridDim.x][3][H] | counts[gridDim.x].
struct ShiftAcc {
    float4 p, S, SS;
    __device__ __forceinline__ void reset() { p = S = SS = f4_zero(); }
    __device__ __forceinline__ void add(float4 v, bool first) {
        if (first) p = v;  // (wave-uniform condition)
        const float4 d = f4_sub(v, p);
        S = f4_add(S, d);
        SS = f4_fma(d, d, SS);
    }
};
__device__ __forceinline__ void pivot_merge4(float& na, float4& pa, float4& Sa, float4& SSa, float nb, float4 pb, float4 Sb,
                                             float4 SSb) {
    if (nb == 0.0f) return;
    if (na == 0.0f) {
        na = nb, pa = pb, Sa = Sb, SSa = SSb;
        return;
    }
    const float4 d = f4_sub(pb, pa);
    SSa = make_float4(SSa.x + SSb.x + d.x * (2.0f * Sb.x + nb * d.x), SSa.y + SSb.y + d.y * (2.0f * Sb.y + nb * d.y),
                      SSa.z + SSb.z + d.z * (2.0f * Sb.z + nb * d.z), SSa.w + SSb.w + d.w * (2.0f * Sb.w + nb * d.w));
    Sa = make_float4(Sa.x + Sb.x + nb * d.x, Sa.y + Sb.y + nb * d.y, Sa.z + Sb.z + nb * d.z, Sa.w + Sb.w + nb * d.w);
    na += nb;
}
// `cnt`: values this wave accumulated (wave-uniform; lane 0 is active in every panel).  Writes slab [3][H] at `slab` and, for
// the first feature panel, the block's row count at `count_out`.
__device__ __forceinline__ void block_moments_store(const ShiftAcc& a, float cnt, float* slab, float* count_out, int H, int f,
                                                    bool active, bool first_panel,
                                                    float4 (*sh)[kWavesPerBlock][ALIGNN_WAVE], float* shn, int wave, int lane) {
    sh[0][wave][lane] = a.p;
    sh[1][wave][lane] = a.S;
    sh[2][wave][lane] = a.SS;
    if (lane == 0) shn[wave] = cnt;
    __syncthreads();
    if (wave == 0) {
        float n = shn[0];
        float4 p = sh[0][0][lane], S = sh[1][0][lane], SS = sh[2][0][lane];
#pragma unroll
        for (int w = 1; w < kWavesPerBlock; ++w) pivot_merge4(n, p, S, SS, shn[w], sh[0][w][lane], sh[1][w][lane], sh[2][w][lane]);
        if (active) {
            f4_st(slab + f, p);
            f4_st(slab + H + f, S);
            f4_st(slab + 2 * H + f, SS);
        }
        if (first_panel && lane == 0) *count_out = n;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
// INFER: nothing is kept for a backward pass and the edge normalisation is a fixed affine map (BatchNorm in eval
// mode: e_stat = mean, rstd, scale, shift from the running statistics), so the edge output
// y' = y + silu((m - mean) * scale + shift) is written straight from the gate pass and m itself never goes to memory:
// 1 read of C (+ 1 of y) + 1 write instead of read C, write m, read m, read y, write y'.
// PRE: M already holds m = A[u] + Bd[v] + C (the edge projection added the two gathered rows in its epilogue,
// alignn_gemm_nt_f16x3_gather): the pass only reads it - no A / Bd loads, no store of m.
template <bool STREAM, bool INFER, bool PRE = false>
__global__ __launch_bounds__(kThreads) void egc_gate_fwd_kernel(
    const float* __restrict__ P, float* __restrict__ M, const int32_t* __restrict__ seg_ptr,
    const int32_t* __restrict__ seg_node, const int32_t* __restrict__ src, int n_seg, int H,
    float* __restrict__ XPRE, float* __restrict__ S0, float* __restrict__ HH, float* __restrict__ e_partial,
    float* __restrict__ n_partial, const float* __restrict__ e_stat, const float* __restrict__ Y,
    float* __restrict__ YOUT, float* __restrict__ y_amax) {
    __shared__ float4 sh[3][kWavesPerBlock][ALIGNN_WAVE];
    __shared__ float shn[kWavesPerBlock];
    float y_am = 0.0f;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t ldp = 4 * (int64_t)H;
    const int first = blockIdx.x * kWavesPerBlock + wave;
    const int stride = gridDim.x * kWavesPerBlock;

    for (int c0 = 0; c0 < H; c0 += 4 * ALIGNN_WAVE) {
        const int f = c0 + 4 * lane;
        const bool active = f < H;
        ShiftAcc e_acc, n_acc;
        e_acc.reset();
        n_acc.reset();
        float e_cnt = 0.0f, n_cnt = 0.0f;  // wave-uniform: edges / segments this wave has accumulated
        float4 e_mean = f4_zero(), e_sc = f4_zero(), e_sh = f4_zero();
        if (INFER && active && YOUT) {
            e_mean = f4_ld(e_stat + f);
            e_sc = f4_ld(e_stat + 2 * H + f);
            e_sh = f4_ld(e_stat + 3 * H + f);
        }
        if (active) {
            for (int s = first; s < n_seg; s += stride) {
                const int beg = seg_ptr[s], end = seg_ptr[s + 1];
                const int i = seg_node ? seg_node[s] : s;
                const float* Pi = P + (int64_t)i * ldp;
                const float4 bd = f4_ld(Pi + H + f);
                const float4 ux = f4_ld(Pi + 3 * H + f);
                float4 s1 = f4_zero(), s0 = f4_zero();
                int e = beg;
                for (; e + 4 <= end; e += 4) {
                    int u[4];
                    float4 a[4], bh[4], c[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) u[k] = src[e + k];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float* Pu = P + (int64_t)u[k] * ldp;
                        if (!PRE) a[k] = f4_ld(Pu + f);
                        bh[k] = f4_ld(Pu + 2 * H + f);
                        c[k] = f4_lds<STREAM>(M + (int64_t)(e + k) * H + f);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float4 m = PRE ? c[k] : f4_add(f4_add(a[k], bd), c[k]);
                        if (INFER) {
                            if (YOUT) {
                                const float4 zz = f4_fma(f4_sub(m, e_mean), e_sc, e_sh);
                                float4 o = make_float4(silu_f(zz.x), silu_f(zz.y), silu_f(zz.z), silu_f(zz.w));
                                if (Y) o = f4_add(o, f4_lds<STREAM>(Y + (int64_t)(e + k) * H + f));
                                f4_sts<STREAM>(YOUT + (int64_t)(e + k) * H + f, o);
                                y_am = fmaxf(y_am, f4_absmax(o));
                            }
                        } else if (!PRE) {
                            f4_sts<STREAM>(M + (int64_t)(e + k) * H + f, m);
                        }
                        float4 sg = f4_sigmoid(m);
                        s1 = f4_fma(sg, bh[k], s1);
                        s0 = f4_add(s0, sg);
                        e_acc.add(m, e_cnt == 0.0f);
                        e_cnt += 1.0f;
                    }
                }
                for (; e < end; ++e) {
                    const float* Pu = P + (int64_t)src[e] * ldp;
                    float4 m = f4_lds<STREAM>(M + (int64_t)e * H + f);
                    if (!PRE) m = f4_add(f4_add(f4_ld(Pu + f), bd), m);
                    if (INFER) {
                        if (YOUT) {
                            const float4 zz = f4_fma(f4_sub(m, e_mean), e_sc, e_sh);
                            float4 o = make_float4(silu_f(zz.x), silu_f(zz.y), silu_f(zz.z), silu_f(zz.w));
                            if (Y) o = f4_add(o, f4_lds<STREAM>(Y + (int64_t)e * H + f));
                            f4_sts<STREAM>(YOUT + (int64_t)e * H + f, o);
                            y_am = fmaxf(y_am, f4_absmax(o));
                        }
                    } else if (!PRE) {
                        f4_sts<STREAM>(M + (int64_t)e * H + f, m);
                    }
                    float4 sg = f4_sigmoid(m);
                    s1 = f4_fma(sg, f4_ld(Pu + 2 * H + f), s1);
                    s0 = f4_add(s0, sg);
                    e_acc.add(m, e_cnt == 0.0f);
                    e_cnt += 1.0f;
                }
                float4 h;
                h.x = s1.x / (s0.x + ALIGNN_EPS_GATE);
                h.y = s1.y / (s0.y + ALIGNN_EPS_GATE);
                h.z = s1.z / (s0.z + ALIGNN_EPS_GATE);
                h.w = s1.w / (s0.w + ALIGNN_EPS_GATE);
                float4 xp = f4_add(ux, h);
                f4_st(XPRE + (int64_t)i * H + f, xp);
                if (S0) f4_st(S0 + (int64_t)i * H + f, s0);
                if (HH) f4_st(HH + (int64_t)i * H + f, h);
                n_acc.add(xp, n_cnt == 0.0f);
                n_cnt += 1.0f;
            }
        }
        // (lane 0 is active in every panel and carries the wave's counts; an inactive lane of a partial last panel
        // accumulated nothing and stores nothing)
        if (e_partial)
            block_moments_store(e_acc, e_cnt, e_partial + (size_t)blockIdx.x * 3 * H, e_partial + (size_t)gridDim.x * 3 * H + blockIdx.x,
                                H, f, active, c0 == 0, sh, shn, wave, lane);
        if (n_partial)
            block_moments_store(n_acc, n_cnt, n_partial + (size_t)blockIdx.x * 3 * H, n_partial + (size_t)gridDim.x * 3 * H + blockIdx.x,
                                H, f, active, c0 == 0, sh, shn, wave, lane);
    }
    if (INFER) block_amax_commit(y_am, y_amax);
}

__global__ __launch_bounds__(256) void egc_node_bwd_kernel(const float* __restrict__ GXPRE, int64_t ldg,
                                                           const float* __restrict__ S0,
                                                           const float* __restrict__ HH, float* __restrict__ GS1,
                                                           float* __restrict__ GS0, int64_t n, int H) {
    const int Q = H >> 2;
    const RowQuad rq(Q);
    const int64_t total = n * Q;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t r;
        int q;
        rq.split(i, total, r, q);
        float4 g = f4_ld(GXPRE + r * ldg + q * 4);
        float4 s0 = f4_ld(S0 + r * H + q * 4);
        float4 h = f4_ld(HH + r * H + q * 4);
        float4 g1, g0;
        g1.x = g.x / (s0.x + ALIGNN_EPS_GATE);
        g1.y = g.y / (s0.y + ALIGNN_EPS_GATE);
        g1.z = g.z / (s0.z + ALIGNN_EPS_GATE);
        g1.w = g.w / (s0.w + ALIGNN_EPS_GATE);
        g0 = make_float4(-g1.x * h.x, -g1.y * h.y, -g1.z * h.z, -g1.w * h.w);
        f4_st(GS1 + r * H + q * 4, g1);
        f4_st(GS0 + r * H + q * 4, g0);
    }
}

// ---------------------------------------------------------------------------------------------
// backward, destination order
// ---------------------------------------------------------------------------------------------
struct EdgeNorm {
    float4 mean, rstd, sc, sh, c0, c1;
};

// MODE 0: edge output dead (no normalised-branch gradient); 1: BatchNorm/SiLU backward fused here from GY;
// 2: GY already holds the finished normalised-branch gradient (LayerNorm flavour: alignn_ln_silu_bwd).
template <int MODE>
__device__ __forceinline__ float4 edge_grad(float4 m, float4 gy, float4 bh, float4 gs1, float4 gs0,
                                            const EdgeNorm& nrm, float inv_n, int eval_mode) {
    constexpr bool HAS_GY = MODE == 1;
    float4 sg = f4_sigmoid(m);
    float4 gsig = f4_fma(gs1, bh, gs0);
    float4 gm;
    gm.x = gsig.x * sg.x * (1.0f - sg.x);
    gm.y = gsig.y * sg.y * (1.0f - sg.y);
    gm.z = gsig.z * sg.z * (1.0f - sg.z);
    gm.w = gsig.w * sg.w * (1.0f - sg.w);
    if (HAS_GY) {
        float4 mc = f4_sub(m, nrm.mean);
        float4 z = f4_fma(mc, nrm.sc, nrm.sh);  // sh holds beta
        float4 gz = make_float4(gy.x * dsilu_f(z.x), gy.y * dsilu_f(z.y), gy.z * dsilu_f(z.z), gy.w * dsilu_f(z.w));
        if (eval_mode) {
            gm = f4_fma(gz, nrm.sc, gm);
        } else {
            float4 xh = f4_mul(mc, nrm.rstd);
            gm.x += nrm.sc.x * (gz.x - inv_n * (nrm.c0.x + xh.x * nrm.c1.x));
            gm.y += nrm.sc.y * (gz.y - inv_n * (nrm.c0.y + xh.y * nrm.c1.y));
            gm.z += nrm.sc.z * (gz.z - inv_n * (nrm.c0.z + xh.z * nrm.c1.z));
            gm.w += nrm.sc.w * (gz.w - inv_n * (nrm.c0.w + xh.w * nrm.c1.w));
        }
    }
    if (MODE == 2) gm = f4_add(gm, gy);
    return gm;
}

template <int MODE>
__global__ __launch_bounds__(kThreads) void egc_bwd_dst_kernel(
    const float* __restrict__ GY, const float* __restrict__ M, const float* __restrict__ P,
    const float* __restrict__ GS1, const float* __restrict__ GS0, const float* __restrict__ e_stat,
    const float* __restrict__ e_red, int e_eval, float inv_n, const int32_t* __restrict__ seg_ptr,
    const int32_t* __restrict__ seg_node, const int32_t* __restrict__ src, int n_seg, int H,
    float* __restrict__ GM, float* __restrict__ GP, float* __restrict__ gb_partial, float* __restrict__ gm_amax,
    float* __restrict__ gp_amax) {
    __shared__ float4 sh[kWavesPerBlock][ALIGNN_WAVE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t ldp = 4 * (int64_t)H;
    const int first = blockIdx.x * kWavesPerBlock + wave;
    const int stride = gridDim.x * kWavesPerBlock;
    float gm_am = 0.0f, gp_am = 0.0f;
    for (int c0 = 0; c0 < H; c0 += 4 * ALIGNN_WAVE) {
        const int f = c0 + 4 * lane;
        const bool active = f < H;
        float4 gb = f4_zero();  // column sum of GM over this wave's segments (= edge_gate bias gradient)
        if (active) {
        constexpr bool HAS_GY = MODE != 0;
        EdgeNorm nrm;
        if (MODE == 1) {
            nrm.mean = f4_ld(e_stat + f);
            nrm.rstd = f4_ld(e_stat + H + f);
            nrm.sc = f4_ld(e_stat + 2 * H + f);
            nrm.sh = f4_ld(e_stat + 3 * H + f);
            if (!e_eval) {
                nrm.c0 = f4_ld(e_red + f);
                nrm.c1 = f4_ld(e_red + H + f);
            }
        }
        for (int s = first; s < n_seg; s += stride) {
            const int beg = seg_ptr[s], end = seg_ptr[s + 1];
            const int i = seg_node ? seg_node[s] : s;
            const float4 gs1 = f4_ld(GS1 + (int64_t)i * H + f);
            const float4 gs0 = f4_ld(GS0 + (int64_t)i * H + f);
            float4 gbd = f4_zero();
            int e = beg;
            for (; e + 4 <= end; e += 4) {
                float4 m[4], bh[4], gy[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int u = src[e + k];
                    m[k] = f4_ld(M + (int64_t)(e + k) * H + f);
                    bh[k] = f4_ld(P + (int64_t)u * ldp + 2 * H + f);
                    if (HAS_GY) gy[k] = f4_ld(GY + (int64_t)(e + k) * H + f);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float4 gm = edge_grad<MODE>(m[k], gy[k], bh[k], gs1, gs0, nrm, inv_n, e_eval);
                    f4_st(GM + (int64_t)(e + k) * H + f, gm);
                    gm_am = fmaxf(gm_am, f4_absmax(gm));
                    gbd = f4_add(gbd, gm);
                }
            }
            for (; e < end; ++e) {
                const int u = src[e];
                float4 m = f4_ld(M + (int64_t)e * H + f);
                float4 bh = f4_ld(P + (int64_t)u * ldp + 2 * H + f);
                float4 gy = f4_zero();
                if (HAS_GY) gy = f4_ld(GY + (int64_t)e * H + f);
                float4 gm = edge_grad<MODE>(m, gy, bh, gs1, gs0, nrm, inv_n, e_eval);
                f4_st(GM + (int64_t)e * H + f, gm);
                gm_am = fmaxf(gm_am, f4_absmax(gm));
                gbd = f4_add(gbd, gm);
            }
            f4_st(GP + (int64_t)i * ldp + H + f, gbd);
            gp_am = fmaxf(gp_am, f4_absmax(gbd));
            gb = f4_add(gb, gbd);
        }
        }
        if (gb_partial) {  // fixed-order block sum of the four waves -> slab [blockIdx.x][H]
            sh[wave][lane] = gb;
            __syncthreads();
            if (wave == 0 && active) {
                float4 a = sh[0][lane];
#pragma unroll
                for (int w = 1; w < kWavesPerBlock; ++w) a = f4_add(a, sh[w][lane]);
                f4_st(gb_partial + (size_t)blockIdx.x * H + f, a);
            }
            __syncthreads();
        }
    }
    block_amax_commit(gm_am, gm_amax);
    block_amax_commit(gp_am, gp_amax);
}

// ---------------------------------------------------------------------------------------------
// backward, source order (deterministic scatter-by-source as a segment sum over out-slots)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void egc_bwd_src_kernel(
    const float* __restrict__ GM, const float* __restrict__ M, const float* __restrict__ GS1,
    const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ out_slot, const int32_t* __restrict__ dst,
    int n_nodes, int H, float* __restrict__ GP, float* __restrict__ gp_amax) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t ldp = 4 * (int64_t)H;
    const int first = blockIdx.x * kWavesPerBlock + wave;
    const int stride = gridDim.x * kWavesPerBlock;
    float gp_am = 0.0f;
    for (int c0 = 0; c0 < H; c0 += 4 * ALIGNN_WAVE) {
        const int f = c0 + 4 * lane;
        if (f >= H) continue;
        for (int j = first; j < n_nodes; j += stride) {
            const int beg = out_ptr[j], end = out_ptr[j + 1];
            float4 ga = f4_zero(), gbh = f4_zero();
            int k = beg;
            for (; k + 4 <= end; k += 4) {
                float4 gm[4], m[4], g1[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int slot = out_slot[k + t];
                    const int v = dst[slot];
                    gm[t] = f4_ld(GM + (int64_t)slot * H + f);
                    m[t] = f4_ld(M + (int64_t)slot * H + f);
                    g1[t] = f4_ld(GS1 + (int64_t)v * H + f);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    ga = f4_add(ga, gm[t]);
                    gbh = f4_fma(f4_sigmoid(m[t]), g1[t], gbh);
                }
            }
            for (; k < end; ++k) {
                const int slot = out_slot[k];
                const int v = dst[slot];
                ga = f4_add(ga, f4_ld(GM + (int64_t)slot * H + f));
                gbh = f4_fma(f4_sigmoid(f4_ld(M + (int64_t)slot * H + f)), f4_ld(GS1 + (int64_t)v * H + f), gbh);
            }
            f4_st(GP + (int64_t)j * ldp + f, ga);
            f4_st(GP + (int64_t)j * ldp + 2 * H + f, gbh);
            gp_am = fmaxf(gp_am, fmaxf(f4_absmax(ga), f4_absmax(gbh)));
        }
    }
    block_amax_commit(gp_am, gp_amax);
}

// ---------------------------------------------------------------------------------------------
// backward on a LINE GRAPH, destination- and source-ordered passes fused.
//
// L(g) has e1 -> e2 iff dst(e1) == src(e2) == atom j, so its edges fall into one dense block per atom j:
// sources = in-edges of j (contiguous L(g) node ids [grp_src_ptr[j], grp_src_ptr[j+1]) in the canonical layout),
// segments = out-edges of j (contiguous segment ranks [grp_seg_ptr[j], grp_seg_ptr[j+1])).  One workgroup owns one
// atom: phase 1 walks the block BY SOURCE (one wave per source, as egc_bwd_src does) but computes the edge
// gradient itself - reading GY and M once, writing GM once and keeping g_A / g_Bh in registers; phase 2 sums GM
// per segment (g_Bd) straight out of L2, since the same workgroup has just written those rows.  HBM traffic:
// 2 reads + 1 write per edge row instead of 4 reads + 1 write for the two separate passes; same summation
// orders, so the results are bit-identical to them.
// ---------------------------------------------------------------------------------------------
template <int MODE, bool STREAM>
__global__ __launch_bounds__(kThreads) void egc_bwd_lg_fused_kernel(
    const float* __restrict__ GY, const float* __restrict__ M, const float* __restrict__ P,
    const float* __restrict__ GS1, const float* __restrict__ GS0, const float* __restrict__ e_stat,
    const float* __restrict__ e_red, int e_eval, float inv_n, const int32_t* __restrict__ grp_seg_ptr,
    const int32_t* __restrict__ grp_src_ptr, const int32_t* __restrict__ seg_ptr,
    const int32_t* __restrict__ seg_node, const int32_t* __restrict__ dst, const int32_t* __restrict__ out_ptr,
    const int32_t* __restrict__ out_slot, int H, float* __restrict__ GM, float* __restrict__ GP,
    float* __restrict__ gb_partial, float* __restrict__ gm_amax, float* __restrict__ gp_amax) {
    __shared__ float4 sh[kWavesPerBlock][ALIGNN_WAVE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t ldp = 4 * (int64_t)H;
    const int j = blockIdx.x;
    float gm_am = 0.0f, gp_am = 0.0f;
    const int p_beg = grp_src_ptr[j], p_end = grp_src_ptr[j + 1];
    const int s_beg = grp_seg_ptr[j], s_end = grp_seg_ptr[j + 1];
    constexpr bool HAS_GY = MODE != 0;
    for (int c0 = 0; c0 < H; c0 += 4 * ALIGNN_WAVE) {
        const int f = c0 + 4 * lane;
        const bool active = f < H;
        float4 gb = f4_zero();
        if (active) {
            EdgeNorm nrm;
            if (MODE == 1) {
                nrm.mean = f4_ld(e_stat + f);
                nrm.rstd = f4_ld(e_stat + H + f);
                nrm.sc = f4_ld(e_stat + 2 * H + f);
                nrm.sh = f4_ld(e_stat + 3 * H + f);
                if (!e_eval) {
                    nrm.c0 = f4_ld(e_red + f);
                    nrm.c1 = f4_ld(e_red + H + f);
                }
            }
            // ---- phase 1: by source
            for (int p = p_beg + wave; p < p_end; p += kWavesPerBlock) {
                const float4 bh = f4_ld(P + (int64_t)p * ldp + 2 * H + f);
                float4 ga = f4_zero(), gbh = f4_zero();
                const int beg = out_ptr[p], end = out_ptr[p + 1];
                int k = beg;
                for (; k + 4 <= end; k += 4) {
                    float4 m[4], gy[4], g1[4], g0[4];
                    int slot[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        slot[t] = out_slot[k + t];
                        const int v = dst[slot[t]];
                        m[t] = f4_lds<STREAM>(M + (int64_t)slot[t] * H + f);
                        if (HAS_GY) gy[t] = f4_lds<STREAM>(GY + (int64_t)slot[t] * H + f);
                        g1[t] = f4_ld(GS1 + (int64_t)v * H + f);
                        g0[t] = f4_ld(GS0 + (int64_t)v * H + f);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float4 gm = edge_grad<MODE>(m[t], gy[t], bh, g1[t], g0[t], nrm, inv_n, e_eval);
                        f4_st(GM + (int64_t)slot[t] * H + f, gm);
                        gm_am = fmaxf(gm_am, f4_absmax(gm));
                        ga = f4_add(ga, gm);
                        gbh = f4_fma(f4_sigmoid(m[t]), g1[t], gbh);
                    }
                }
                for (; k < end; ++k) {
                    const int slot = out_slot[k];
                    const int v = dst[slot];
                    float4 m = f4_lds<STREAM>(M + (int64_t)slot * H + f);
                    float4 gy = f4_zero();
                    if (HAS_GY) gy = f4_lds<STREAM>(GY + (int64_t)slot * H + f);
                    float4 g1 = f4_ld(GS1 + (int64_t)v * H + f), g0 = f4_ld(GS0 + (int64_t)v * H + f);
                    float4 gm = edge_grad<MODE>(m, gy, bh, g1, g0, nrm, inv_n, e_eval);
                    f4_st(GM + (int64_t)slot * H + f, gm);
                    gm_am = fmaxf(gm_am, f4_absmax(gm));
                    ga = f4_add(ga, gm);
                    gbh = f4_fma(f4_sigmoid(m), g1, gbh);
                }
                f4_st(GP + (int64_t)p * ldp + f, ga);
                f4_st(GP + (int64_t)p * ldp + 2 * H + f, gbh);
                gp_am = fmaxf(gp_am, fmaxf(f4_absmax(ga), f4_absmax(gbh)));
            }
        }
        __syncthreads();  // (waits for the GM stores of every wave: they are in L2 now)
        // ---- phase 2: by destination segment, GM re-read from L2
        if (active) {
            for (int s = s_beg + wave; s < s_end; s += kWavesPerBlock) {
                const int i = seg_node ? seg_node[s] : s;
                float4 gbd = f4_zero();
                for (int e = seg_ptr[s]; e < seg_ptr[s + 1]; ++e) gbd = f4_add(gbd, f4_ld(GM + (int64_t)e * H + f));
                f4_st(GP + (int64_t)i * ldp + H + f, gbd);
                gp_am = fmaxf(gp_am, f4_absmax(gbd));
                gb = f4_add(gb, gbd);
            }
        }
        if (gb_partial) {
            sh[wave][lane] = gb;
            __syncthreads();
            if (wave == 0 && active) {
                float4 a = sh[0][lane];
#pragma unroll
                for (int w = 1; w < kWavesPerBlock; ++w) a = f4_add(a, sh[w][lane]);
                f4_st(gb_partial + (size_t)blockIdx.x * H + f, a);
            }
        }
        __syncthreads();
    }
    block_amax_commit(gm_am, gm_amax);
    block_amax_commit(gp_am, gp_amax);
}

// ---------------------------------------------------------------------------------------------
// backward on a line graph whose blocks are DENSE and source-sorted (CSRGraph.dense_max_src > 0: true line graphs,
// incl. the ones built on the device): segment s of atom j lists all sources q = 0..P-1 of j in order, minus s
// itself for a self-image bond, so the row of (q, s) is  seg_ptr[s] + q - (q > self_q)  - no index lists at all.
// The workgroup of atom j walks its block SEGMENT by segment (rows strictly sequential in memory); wave w owns the
// sources q = w, w+4, ... and keeps their Bh rows and their g_A / g_Bh sums in registers for the whole block, the
// segment's gS1 / gS0 rows are loaded once per segment, and g_Bd[s] is the fixed-order sum of the four waves'
// partial sums (LDS).  Every T-row of GY and M is read once and GM written once: PMC FETCH of the by-source kernel
// above is 3.6 row passes (GM re-read from HBM - a block's 176 KiB x 8 workgroups per CU overflow the L2 - plus
// ~0.6 pass of re-fetched gS1/gS0 rows), here 2.
// ---------------------------------------------------------------------------------------------
constexpr int kDenseK = 4;  // sources per wave and pass: 16 per workgroup (kNN-12 graphs: 12-16 in-edges per atom)

template <int MODE, bool STREAM>
__global__ __launch_bounds__(kThreads) void egc_bwd_lg_dense_kernel(
    const float* __restrict__ GY, const float* __restrict__ M, const float* __restrict__ P,
    const float* __restrict__ GS1, const float* __restrict__ GS0, const float* __restrict__ e_stat,
    const float* __restrict__ e_red, int e_eval, float inv_n, const int32_t* __restrict__ grp_seg_ptr,
    const int32_t* __restrict__ grp_src_ptr, const int32_t* __restrict__ seg_ptr,
    const int32_t* __restrict__ seg_node, int H, float* __restrict__ GM, float* __restrict__ GP,
    float* __restrict__ gb_partial, float* __restrict__ gm_amax, float* __restrict__ gp_amax) {
    constexpr int KMAX = kDenseK;
    __shared__ float4 sh[2][kWavesPerBlock][ALIGNN_WAVE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t ldp = 4 * (int64_t)H;
    const int j = blockIdx.x;
    const int p_beg = grp_src_ptr[j], n_src = grp_src_ptr[j + 1] - p_beg;
    const int s_beg = grp_seg_ptr[j], s_end = grp_seg_ptr[j + 1];
    constexpr bool HAS_GY = MODE != 0;
    float gm_am = 0.0f, gp_am = 0.0f;
    for (int c0 = 0; c0 < H; c0 += 4 * ALIGNN_WAVE) {
        const int f = c0 + 4 * lane;
        const bool active = f < H;
        EdgeNorm nrm;
        if (active && MODE == 1) {
            nrm.mean = f4_ld(e_stat + f);
            nrm.rstd = f4_ld(e_stat + H + f);
            nrm.sc = f4_ld(e_stat + 2 * H + f);
            nrm.sh = f4_ld(e_stat + 3 * H + f);
            if (!e_eval) {
                nrm.c0 = f4_ld(e_red + f);
                nrm.c1 = f4_ld(e_red + H + f);
            }
        }
        float4 gb = f4_zero();
        // atoms with more than 16 sources take further passes over their segments, 16 sources at a time (each row
        // still belongs to exactly one pass; g_Bd accumulates across the passes in pass order)
        for (int qb = 0; qb < n_src || qb == 0; qb += kWavesPerBlock * KMAX) {
            float4 bh[KMAX], ga[KMAX], gbh[KMAX];
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const int q = qb + wave + kWavesPerBlock * k;
                bh[k] = (active && q < n_src) ? f4_ld(P + (int64_t)(p_beg + q) * ldp + 2 * H + f) : f4_zero();
                ga[k] = gbh[k] = f4_zero();
            }
            for (int s = s_beg; s < s_end; ++s) {
                const int i = seg_node ? seg_node[s] : s;
                const int e0 = seg_ptr[s];
                int self_q = i - p_beg;  // the segment's own bond among the sources (self-image) is not listed
                if (self_q < 0 || self_q >= n_src || seg_ptr[s + 1] - e0 == n_src) self_q = -1;
                float4 gbd = f4_zero();
                if (active) {
                    const float4 g1 = f4_ld(GS1 + (int64_t)i * H + f), g0 = f4_ld(GS0 + (int64_t)i * H + f);
                    float4 m[KMAX], gy[KMAX];
                    int row[KMAX];
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) {
                        const int q = qb + wave + kWavesPerBlock * k;
                        row[k] = (q < n_src && q != self_q) ? e0 + q - ((self_q >= 0 && q > self_q) ? 1 : 0) : -1;
                        if (row[k] >= 0) {
                            m[k] = f4_lds<STREAM>(M + (int64_t)row[k] * H + f);
                            if (HAS_GY) gy[k] = f4_lds<STREAM>(GY + (int64_t)row[k] * H + f);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) {
                        if (row[k] >= 0) {
                            const float4 gm = edge_grad<MODE>(m[k], gy[k], bh[k], g1, g0, nrm, inv_n, e_eval);
                            f4_sts<STREAM>(GM + (int64_t)row[k] * H + f, gm);
                            gm_am = fmaxf(gm_am, f4_absmax(gm));
                            ga[k] = f4_add(ga[k], gm);
                            gbh[k] = f4_fma(f4_sigmoid(m[k]), g1, gbh[k]);
                            gbd = f4_add(gbd, gm);
                        }
                    }
                }
                // g_Bd[s]: the four waves' partial sums in wave order (double-buffered, one barrier per segment)
                float4(*buf)[ALIGNN_WAVE] = sh[(s - s_beg) & 1];
                buf[wave][lane] = gbd;
                __syncthreads();
                if (wave == 0 && active) {
                    float4 a = buf[0][lane];
#pragma unroll
                    for (int w = 1; w < kWavesPerBlock; ++w) a = f4_add(a, buf[w][lane]);
                    gb = f4_add(gb, a);
                    float* out = GP + (int64_t)i * ldp + H + f;
                    if (qb > 0) a = f4_add(f4_ld(out), a);  // written by this very thread in the previous pass
                    f4_st(out, a);
                    gp_am = fmaxf(gp_am, f4_absmax(a));
                }
            }
            if (active) {
#pragma unroll
                for (int k = 0; k < KMAX; ++k) {
                    const int q = qb + wave + kWavesPerBlock * k;
                    if (q < n_src) {
                        f4_st(GP + (int64_t)(p_beg + q) * ldp + f, ga[k]);
                        f4_st(GP + (int64_t)(p_beg + q) * ldp + 2 * H + f, gbh[k]);
                        gp_am = fmaxf(gp_am, fmaxf(f4_absmax(ga[k]), f4_absmax(gbh[k])));
                    }
                }
            }
            __syncthreads();  // sh is reused by the next pass / feature panel
        }
        if (active && gb_partial && wave == 0) f4_st(gb_partial + (size_t)blockIdx.x * H + f, gb);
    }
    block_amax_commit(gm_am, gm_amax);
    block_amax_commit(gp_am, gp_amax);
}

inline bool big_stream(int64_t rows, int H) { return rows * (int64_t)H * 4 >= (int64_t)128 << 20; }
inline bool h_ok(int H) { return H >= 4 && (H & 3) == 0 && H <= 1024; }

}  // namespace

extern "C" {

int alignn_egc_slabs(int64_t n_seg) { return egc_blocks(n_seg); }

int alignn_egc_gate_fwd(const float* P, float* M, const int32_t* seg_ptr, const int32_t* seg_node,
                        const int32_t* src, int64_t n_seg, int64_t m_rows, int H, float* XPRE, float* S0, float* HH,
                        float* e_partial, float* n_partial, alignn_stream_t stream) {
    if (!h_ok(H) || n_seg < 0 || n_seg > INT32_MAX || m_rows < 0) return (int)hipErrorInvalidValue;
    if (big_stream(m_rows, H))  // M cannot stay in the last-level cache: read-once / write-once hints
        hipLaunchKernelGGL((egc_gate_fwd_kernel<true, false>), dim3(egc_blocks(n_seg)), dim3(kThreads), 0,
                           (hipStream_t)stream, P, M, seg_ptr, seg_node, src, (int)n_seg, H, XPRE, S0, HH, e_partial,
                           n_partial, nullptr, nullptr, nullptr, nullptr);
    else
        hipLaunchKernelGGL((egc_gate_fwd_kernel<false, false>), dim3(egc_blocks(n_seg)), dim3(kThreads), 0,
                           (hipStream_t)stream, P, M, seg_ptr, seg_node, src, (int)n_seg, H, XPRE, S0, HH, e_partial,
                           n_partial, nullptr, nullptr, nullptr, nullptr);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_egc_gate_fwd_pre(const float* P, float* M, const int32_t* seg_ptr, const int32_t* seg_node,
                            const int32_t* src, int64_t n_seg, int64_t m_rows, int H, float* XPRE, float* S0, float* HH,
                            float* e_partial, float* n_partial, alignn_stream_t stream) {
    if (!h_ok(H) || n_seg < 0 || n_seg > INT32_MAX || m_rows < 0) return (int)hipErrorInvalidValue;
    if (big_stream(m_rows, H))
        hipLaunchKernelGGL((egc_gate_fwd_kernel<true, false, true>), dim3(egc_blocks(n_seg)), dim3(kThreads), 0,
                           (hipStream_t)stream, P, M, seg_ptr, seg_node, src, (int)n_seg, H, XPRE, S0, HH, e_partial,
                           n_partial, nullptr, nullptr, nullptr, nullptr);
    else
        hipLaunchKernelGGL((egc_gate_fwd_kernel<false, false, true>), dim3(egc_blocks(n_seg)), dim3(kThreads), 0,
                           (hipStream_t)stream, P, M, seg_ptr, seg_node, src, (int)n_seg, H, XPRE, S0, HH, e_partial,
                           n_partial, nullptr, nullptr, nullptr, nullptr);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_egc_gate_fwd_pre_norm(const float* P, const float* M, const int32_t* seg_ptr, const int32_t* seg_node,
                                 const int32_t* src, int64_t n_seg, int64_t m_rows, int H, float* XPRE, float* S0,
                                 float* HH, float* n_partial, const float* e_stat, const float* Y, float* YOUT,
                                 float* y_amax, alignn_stream_t stream) {
    if (!h_ok(H) || n_seg < 0 || n_seg > INT32_MAX || m_rows < 0 || !e_stat || !YOUT) return (int)hipErrorInvalidValue;
    float* Mm = const_cast<float*>(M);  // the <INFER, PRE> instantiation only reads it
    if (big_stream(m_rows, H))
        hipLaunchKernelGGL((egc_gate_fwd_kernel<true, true, true>), dim3(egc_blocks(n_seg)), dim3(kThreads), 0,
                           (hipStream_t)stream, P, Mm, seg_ptr, seg_node, src, (int)n_seg, H, XPRE, S0, HH, nullptr,
                           n_partial, e_stat, Y, YOUT, y_amax);
    else
        hipLaunchKernelGGL((egc_gate_fwd_kernel<false, true, true>), dim3(egc_blocks(n_seg)), dim3(kThreads), 0,
                           (hipStream_t)stream, P, Mm, seg_ptr, seg_node, src, (int)n_seg, H, XPRE, S0, HH, nullptr,
                           n_partial, e_stat, Y, YOUT, y_amax);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_egc_gate_infer(const float* P, const float* C, const int32_t* seg_ptr, const int32_t* seg_node,
                          const int32_t* src, int64_t n_seg, int64_t m_rows, int H, float* XPRE, const float* e_stat,
                          const float* Y, float* YOUT, float* y_amax, alignn_stream_t stream) {
    if (!h_ok(H) || n_seg < 0 || n_seg > INT32_MAX || m_rows < 0 || (YOUT && !e_stat)) return (int)hipErrorInvalidValue;
    float* Cm = const_cast<float*>(C);  // the INFER instantiation only reads it
    if (big_stream(m_rows, H))
        hipLaunchKernelGGL((egc_gate_fwd_kernel<true, true>), dim3(egc_blocks(n_seg)), dim3(kThreads), 0,
                           (hipStream_t)stream, P, Cm, seg_ptr, seg_node, src, (int)n_seg, H, XPRE, nullptr, nullptr,
                           nullptr, nullptr, e_stat, Y, YOUT, y_amax);
    else
        hipLaunchKernelGGL((egc_gate_fwd_kernel<false, true>), dim3(egc_blocks(n_seg)), dim3(kThreads), 0,
                           (hipStream_t)stream, P, Cm, seg_ptr, seg_node, src, (int)n_seg, H, XPRE, nullptr, nullptr,
                           nullptr, nullptr, e_stat, Y, YOUT, y_amax);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_egc_node_bwd(const float* GXPRE, int64_t ldg, const float* S0, const float* HH, float* GS1, float* GS0,
                        int64_t n_nodes, int H, alignn_stream_t stream) {
    if (!h_ok(H)) return (int)hipErrorInvalidValue;
    if (n_nodes == 0) return 0;
    int64_t g = (n_nodes * (H >> 2) + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(egc_node_bwd_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, GXPRE, ldg, S0, HH, GS1,
                       GS0, n_nodes, H);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_egc_bwd_dst(const float* GY, const float* M, const float* P, const float* GS1, const float* GS0,
                       const float* e_stat, const float* e_gamma, const float* e_red, int e_eval, int64_t m_rows,
                       const int32_t* seg_ptr, const int32_t* seg_node, const int32_t* src, int64_t n_seg, int H,
                       float* GM, float* GP, float* gb_partial, float* gm_amax, float* gp_amax,
                       alignn_stream_t stream) {
    (void)e_gamma;
    if (!h_ok(H) || n_seg > INT32_MAX) return (int)hipErrorInvalidValue;
    const float inv_n = m_rows > 0 ? 1.0f / (float)m_rows : 0.0f;
    dim3 grid(egc_blocks(n_seg)), block(kThreads);
    if (GY && e_stat)
        hipLaunchKernelGGL(egc_bwd_dst_kernel<1>, grid, block, 0, (hipStream_t)stream, GY, M, P, GS1, GS0, e_stat,
                           e_red, e_eval, inv_n, seg_ptr, seg_node, src, (int)n_seg, H, GM, GP, gb_partial,
                           gm_amax, gp_amax);
    else if (GY)  // e_stat == NULL: GY is the finished normalised-branch gradient (LayerNorm flavour)
        hipLaunchKernelGGL(egc_bwd_dst_kernel<2>, grid, block, 0, (hipStream_t)stream, GY, M, P, GS1, GS0, e_stat,
                           e_red, e_eval, inv_n, seg_ptr, seg_node, src, (int)n_seg, H, GM, GP, gb_partial,
                           gm_amax, gp_amax);
    else
        hipLaunchKernelGGL(egc_bwd_dst_kernel<0>, grid, block, 0, (hipStream_t)stream, GY, M, P, GS1, GS0, e_stat,
                           e_red, e_eval, inv_n, seg_ptr, seg_node, src, (int)n_seg, H, GM, GP, gb_partial,
                           gm_amax, gp_amax);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_egc_bwd_lg_fused(const float* GY, const float* M, const float* P, const float* GS1, const float* GS0,
                            const float* e_stat, const float* e_red, int e_eval, int64_t m_rows,
                            const int32_t* grp_seg_ptr, const int32_t* grp_src_ptr, int64_t n_groups,
                            const int32_t* seg_ptr, const int32_t* seg_node, const int32_t* dst,
                            const int32_t* out_ptr, const int32_t* out_slot, int H, float* GM, float* GP,
                            float* gb_partial, float* gm_amax, float* gp_amax, alignn_stream_t stream) {
    if (!h_ok(H) || n_groups <= 0 || n_groups > INT32_MAX) return (int)hipErrorInvalidValue;
    const float inv_n = m_rows > 0 ? 1.0f / (float)m_rows : 0.0f;
    dim3 grid((int)n_groups), block(kThreads);
    hipStream_t st = (hipStream_t)stream;
#define ALIGNN_LGF(MODE_)                                                                                           \
    if (big_stream(m_rows, H))                                                                                      \
        hipLaunchKernelGGL((egc_bwd_lg_fused_kernel<MODE_, true>), grid, block, 0, st, GY, M, P, GS1, GS0, e_stat,     \
                           e_red, e_eval, inv_n, grp_seg_ptr, grp_src_ptr, seg_ptr, seg_node, dst, out_ptr, out_slot, \
                           H, GM, GP, gb_partial, gm_amax, gp_amax);                                                \
    else                                                                                                            \
        hipLaunchKernelGGL((egc_bwd_lg_fused_kernel<MODE_, false>), grid, block, 0, st, GY, M, P, GS1, GS0, e_stat,    \
                           e_red, e_eval, inv_n, grp_seg_ptr, grp_src_ptr, seg_ptr, seg_node, dst, out_ptr, out_slot, \
                           H, GM, GP, gb_partial, gm_amax, gp_amax)
    if (GY && e_stat) {
        ALIGNN_LGF(1);
    } else if (GY) {
        ALIGNN_LGF(2);
    } else {
        ALIGNN_LGF(0);
    }
#undef ALIGNN_LGF
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_egc_bwd_lg_dense_supported(int max_group_src) { return max_group_src > 0 ? 1 : 0; }

int alignn_egc_bwd_lg_dense(const float* GY, const float* M, const float* P, const float* GS1, const float* GS0,
                            const float* e_stat, const float* e_red, int e_eval, int64_t m_rows,
                            const int32_t* grp_seg_ptr, const int32_t* grp_src_ptr, int64_t n_groups,
                            int max_group_src, const int32_t* seg_ptr, const int32_t* seg_node, int H, float* GM,
                            float* GP, float* gb_partial, float* gm_amax, float* gp_amax, alignn_stream_t stream) {
    if (!h_ok(H) || n_groups <= 0 || n_groups > INT32_MAX || !alignn_egc_bwd_lg_dense_supported(max_group_src))
        return (int)hipErrorInvalidValue;
    const float inv_n = m_rows > 0 ? 1.0f / (float)m_rows : 0.0f;
    dim3 grid((int)n_groups), block(kThreads);
    hipStream_t st = (hipStream_t)stream;
#define ALIGNN_LGD2(MODE_, ST_)                                                                                     \
    hipLaunchKernelGGL((egc_bwd_lg_dense_kernel<MODE_, ST_>), grid, block, 0, st, GY, M, P, GS1, GS0, e_stat, e_red,   \
                       e_eval, inv_n, grp_seg_ptr, grp_src_ptr, seg_ptr, seg_node, H, GM, GP, gb_partial, gm_amax,  \
                       gp_amax)
#define ALIGNN_LGD(MODE_)                            \
    if (big_stream(m_rows, H)) {                     \
        ALIGNN_LGD2(MODE_, true);                    \
    } else {                                         \
        ALIGNN_LGD2(MODE_, false);                   \
    }
    if (GY && e_stat) {
        ALIGNN_LGD(1)
    } else if (GY) {
        ALIGNN_LGD(2)
    } else {
        ALIGNN_LGD(0)
    }
#undef ALIGNN_LGD
#undef ALIGNN_LGD2
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_egc_bwd_src(const float* GM, const float* M, const float* GS1, const int32_t* out_ptr,
                       const int32_t* out_slot, const int32_t* dst, int64_t n_nodes, int H, float* GP,
                       float* gp_amax, alignn_stream_t stream) {
    if (!h_ok(H) || n_nodes > INT32_MAX) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(egc_bwd_src_kernel, dim3(egc_blocks(n_nodes)), dim3(kThreads), 0, (hipStream_t)stream, GM, M,
                       GS1, out_ptr, out_slot, dst, (int)n_nodes, H, GP, gp_amax);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
