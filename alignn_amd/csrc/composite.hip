// Composite entry points: one C call = one EdgeGatedGraphConv forward / backward (BatchNorm flavour, training).
//
// The kernels of a convolution are launched by ~11 (forward) and ~20 (backward) separate C-ABI calls when the host code
// sequences them (alignn_amd/ops.py EdgeGatedConvFn); at ~10 us of interpreter + ctypes + allocator work per call that is
// 9 ms of host time per training step at the benchmark batch and MORE than the GPU time at small batches (8 crystals:
// 9.6 ms eagerly launched vs 4.9 ms replayed).  These entry points issue the SAME launches with the SAME arguments in the
// SAME order from C - results are bit-identical to the per-kernel path (tests flip ops.COMPOSITE) - so an eagerly
// launched step costs three C calls per convolution instead of ~30.  Reference: the body of EdgeGatedGraphConv.forward,
// alignn/models/alignn.py:78-129, and torch.autograd's backward of it.
//
// Like every entry point of this library they only enqueue on the given stream, never allocate and never synchronise;
// all buffers (outputs, saved tensors, one scratch block whose size alignn_egc_conv_*_scratch returns) are the caller's.
#include "common.h"
#include "../../include/alignn_hip.h"

namespace {

inline size_t al(size_t floats) { return (floats + 63) / 64 * 64; }  // 256-byte aligned carving of the scratch block

#define ALIGNN_TRY(call)            \
    do {                            \
        int rc__ = (call);          \
        if (rc__ != 0) return rc__; \
    } while (0)

constexpr int kFoldAbove = 1024;  // ops.FOLD_ABOVE: more slabs than this are pre-summed to alignn_slab_fold_slabs() first

// forward scratch: [e_part | e_fold | n_part | bd2 (edge_kind 2)]
struct FwdScratch {
    size_t e_part, e_fold, n_part, bd2, total;
    int e_slabs, n_slabs;
};
FwdScratch fwd_scratch(int64_t n, int64_t m, int H, int Kin, int edge_kind) {
    FwdScratch s{};
    s.n_slabs = alignn_egc_slabs(n);
    s.e_slabs = edge_kind >= 1 ? alignn_gemm_nt_x6_row_tiles(m, H, Kin) : s.n_slabs;
    size_t off = 0;
    s.e_part = off, off += al((size_t)(s.e_slabs + 1) * (3 * H + 1));  // (the gate pass writes pivot slabs [3][H] + counts)
    s.e_fold = off, off += al((size_t)alignn_slab_fold_slabs() * 2 * H);
    s.n_part = off, off += al((size_t)s.n_slabs * (3 * H + 1));
    s.bd2 = off, off += edge_kind == 2 ? al((size_t)n * H) : 0;
    s.total = off;
    return s;
}

// backward scratch: [n_red_part | e_red_part | dx_ws | dy_red_part | dy_red_fold]
struct BwdScratch {
    size_t n_red_part, e_red_part, dx_ws, dy_part, dy_fold, total;
    int n_slabs, e_slabs, dy_tiles;
    size_t dx_bytes;
};
BwdScratch bwd_scratch(int64_t n, int64_t m, int H, int Kin, int dx_kind, int dy_kind) {
    BwdScratch s{};
    s.n_slabs = alignn_col_stats_slabs(n);
    s.e_slabs = alignn_col_stats_slabs(m);
    size_t off = 0;
    s.n_red_part = off, off += al((size_t)s.n_slabs * 2 * H);
    s.e_red_part = off, off += al((size_t)s.e_slabs * 2 * H);
    s.dx_bytes = dx_kind == 2 ? alignn_gemm_nn_split_workspace(n, 4 * H, Kin) : 0;
    s.dx_ws = off, off += al(s.dx_bytes / 4);
    s.dy_tiles = dy_kind == 1 ? alignn_gemm_nt_x6_row_tiles(m, Kin, H) : 0;
    s.dy_part = off, off += al((size_t)(s.dy_tiles + 1) * 2 * Kin);
    s.dy_fold = off, off += al((size_t)alignn_slab_fold_slabs() * 2 * Kin);
    s.total = off;
    return s;
}

// BatchNorm statistics from slabs: fold first when there are very many (ops._bn_finalize)
int bn_finalize_folded(const float* partial, int slabs, float* fold, int64_t rows, int F, const float* gamma,
                       const float* beta, float eps, float momentum, float* rm, float* rv, float* stat,
                       alignn_stream_t st) {
    if (slabs > kFoldAbove) {
        ALIGNN_TRY(alignn_slab_fold(partial, slabs, 2 * F, fold, st));
        partial = fold;
        slabs = alignn_slab_fold_slabs();
    }
    return alignn_bn_finalize(partial, slabs, rows, F, gamma, beta, eps, momentum, rm, rv, stat, st);
}

// fork / join events of the composite entry points, one pair per device (alignn_fork_events_init)
constexpr int kMaxDevices = 32;
hipEvent_t g_fork_ev[kMaxDevices][2];
bool g_fork_ok[kMaxDevices];

inline bool fork_events(hipEvent_t* fork, hipEvent_t* join) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices || !g_fork_ok[dev]) return false;
    *fork = g_fork_ev[dev][0];
    *join = g_fork_ev[dev][1];
    return true;
}

}  // namespace

extern "C" {

int alignn_fork_events_init(void) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= kMaxDevices) return (int)hipErrorInvalidDevice;
    if (g_fork_ok[dev]) return 0;
    for (int i = 0; i < 2; ++i) {
        e = hipEventCreateWithFlags(&g_fork_ev[dev][i], hipEventDisableTiming);
        if (e != hipSuccess) return (int)e;
    }
    g_fork_ok[dev] = true;
    return 0;
}

size_t alignn_egc_args_sizeof(int which) {
    switch (which) {
        case 0: return sizeof(alignn_egc_fwd_args);
        case 1: return sizeof(alignn_egc_bwd_args);
        case 2: return sizeof(alignn_egc_wgrad_args);
        default: return 0;
    }
}

size_t alignn_egc_conv_fwd_scratch(int64_t n, int64_t m, int H, int Kin, int edge_kind) {
    return fwd_scratch(n, m, H, Kin, edge_kind).total * sizeof(float);
}

int alignn_egc_conv_fwd(const alignn_egc_fwd_args* a, alignn_stream_t st) {
    if (a == nullptr || a->scratch == nullptr || a->H <= 0 || (a->H & 3)) return (int)hipErrorInvalidValue;
    const int H = a->H, Kin = a->Kin;
    const int64_t n = a->n, m = a->m;
    const FwdScratch s = fwd_scratch(n, m, H, Kin, a->edge_kind);
    if (a->scratch_bytes < s.total * sizeof(float)) return (int)hipErrorInvalidValue;
    float* e_part = a->scratch + s.e_part;
    float* e_fold = a->scratch + s.e_fold;
    float* n_part = a->scratch + s.n_part;
    // ---- node projection P = x [W_sg; W_dg; W_du; W_su]^T + b = A | Bd | Bh | Ux
    if (a->node_kind == 1)
        ALIGNN_TRY(alignn_gemm_nt_f16x3(a->x, Kin, a->x_amax, a->wcat_img, a->wcat_amax, a->bcat, nullptr, 0, a->P, 4 * H, n,
                                        4 * H, Kin, st));
    else
        ALIGNN_TRY(alignn_gemm_nt(a->x, Kin, a->wcat, Kin, a->bcat, nullptr, 0, a->P, 4 * H, n, 4 * H, Kin, st));
    // ---- edge branch + gate pass
    if (a->edge_kind >= 1) {
        // u_add_v and the BatchNorm statistics in the projection's epilogue; the gate pass normalises itself
        if (a->edge_kind == 2) {  // line graphs: the destination term from a segment-ordered copy of Bd (same values)
            if (a->seg_rank == nullptr || a->seg_node == nullptr) return (int)hipErrorInvalidValue;
            float* bd2 = a->scratch + s.bd2;
            ALIGNN_TRY(alignn_gather_rows_ld(a->P + H, 4 * H, a->seg_node, bd2, H, n, H, st));
            ALIGNN_TRY(alignn_gemm_nt_f16x3_gather2(a->y, Kin, a->y_amax, a->weg_img, a->weg_amax, a->b_eg, a->M, H, m, H, Kin,
                                                    a->P, 4 * H, a->src, bd2, H, a->seg_rank, e_part, st));
        } else
            ALIGNN_TRY(alignn_gemm_nt_f16x3_gather(a->y, Kin, a->y_amax, a->weg_img, a->weg_amax, a->b_eg, a->M, H, m, H, Kin,
                                                   a->P, 4 * H, a->src, a->dst, e_part, st));
        ALIGNN_TRY(bn_finalize_folded(e_part, s.e_slabs, e_fold, m, H, a->e_gamma, a->e_beta, a->eps, a->momentum, a->e_rm,
                                      a->e_rv, a->e_stat, st));
        if (a->y_out != nullptr)
            ALIGNN_TRY(alignn_egc_gate_fwd_pre_norm(a->P, a->M, a->seg_ptr, a->seg_node, a->src, n, m, H, a->xpre, a->s0, a->hh,
                                                    n_part, a->e_stat, a->residual ? a->y : nullptr, a->y_out, a->y_out_amax, st));
        else
            ALIGNN_TRY(alignn_egc_gate_fwd_pre(a->P, a->M, a->seg_ptr, a->seg_node, a->src, n, m, H, a->xpre, a->s0, a->hh,
                                               nullptr, n_part, st));
    } else {
        ALIGNN_TRY(alignn_gemm_nt(a->y, Kin, a->w_eg, Kin, a->b_eg, nullptr, 0, a->M, H, m, H, Kin, st));
        ALIGNN_TRY(alignn_egc_gate_fwd(a->P, a->M, a->seg_ptr, a->seg_node, a->src, n, m, H, a->xpre, a->s0, a->hh, e_part,
                                       n_part, st));
        ALIGNN_TRY(alignn_bn_finalize_welford(e_part, s.e_slabs, m, H, a->e_gamma, a->e_beta, a->eps, a->momentum, a->e_rm,
                                              a->e_rv, a->e_stat, st));
        if (a->y_out != nullptr)
            ALIGNN_TRY(alignn_bn_silu_fwd(a->M, H, a->residual ? a->y : nullptr, a->residual ? Kin : 0, a->e_stat, a->y_out, H,
                                          m, H, a->y_out_amax, st));
    }
    // ---- node norm
    ALIGNN_TRY(alignn_bn_finalize_welford(n_part, s.n_slabs, n, H, a->n_gamma, a->n_beta, a->eps, a->momentum, a->n_rm,
                                          a->n_rv, a->n_stat, st));
    ALIGNN_TRY(alignn_bn_silu_fwd(a->xpre, H, a->residual ? a->x : nullptr, a->residual ? Kin : 0, a->n_stat, a->x_out, H, n, H,
                                  a->x_out_amax, st));
    return 0;
}

size_t alignn_egc_conv_bwd_scratch(int64_t n, int64_t m, int H, int Kin, int dx_kind, int dy_kind) {
    return bwd_scratch(n, m, H, Kin, dx_kind, dy_kind).total * sizeof(float);
}

int alignn_egc_conv_bwd(const alignn_egc_bwd_args* a, alignn_stream_t st) {
    if (a == nullptr || a->scratch == nullptr || a->H <= 0 || (a->H & 3)) return (int)hipErrorInvalidValue;
    const int H = a->H, Kin = a->Kin;
    const int64_t n = a->n, m = a->m;
    const BwdScratch s = bwd_scratch(n, m, H, Kin, a->dx_kind, a->dy_kind);
    if (a->scratch_bytes < s.total * sizeof(float)) return (int)hipErrorInvalidValue;
    float* n_red_part = a->scratch + s.n_red_part;
    float* e_red_part = a->scratch + s.e_red_part;
    float* g_xpre = a->GP + 3 * (size_t)H;  // the Ux block of the projection gradient
    // ---- node branch: BatchNorm / SiLU backward -> g_xpre, then the quotient's adjoints
    ALIGNN_TRY(alignn_bn_silu_bwd_reduce(a->gx_out, H, a->xpre, H, a->n_stat, n, H, n_red_part, st));
    ALIGNN_TRY(alignn_bn_bwd_finalize(n_red_part, s.n_slabs, H, a->n_red, st));
    ALIGNN_TRY(alignn_bn_silu_bwd_apply_node(a->gx_out, H, a->xpre, H, a->n_stat, a->n_gamma, a->n_red, 0, g_xpre, 4 * H, n, H,
                                             a->gp_amax, a->s0, a->hh, a->gs1, a->gs0, st));
    // ---- edge branch: the BatchNorm-backward sums of the edge output (unless the consumer's projection left them)
    const float* e_red = a->e_red_in;
    if (a->gy_out != nullptr && e_red == nullptr) {
        ALIGNN_TRY(alignn_bn_silu_bwd_reduce(a->gy_out, H, a->M, H, a->e_stat, m, H, e_red_part, st));
        ALIGNN_TRY(alignn_bn_bwd_finalize(e_red_part, s.e_slabs, H, a->e_red, st));
        e_red = a->e_red;
    }
    // ---- gate backward: GM, the A | Bd | Bh blocks of GP, column-sum slabs of GM
    if (a->gate_mode == 2) {
        ALIGNN_TRY(alignn_egc_bwd_lg_dense(a->gy_out, a->M, a->P, a->gs1, a->gs0, a->e_stat, e_red, 0, m, a->grp_seg_ptr,
                                           a->grp_src_ptr, a->n_groups, a->dense_max_src, a->seg_ptr, a->seg_node, H, a->GM,
                                           a->GP, a->gb_part, a->gm_amax, a->gp_amax, st));
    } else if (a->gate_mode == 1) {
        ALIGNN_TRY(alignn_egc_bwd_lg_fused(a->gy_out, a->M, a->P, a->gs1, a->gs0, a->e_stat, e_red, 0, m, a->grp_seg_ptr,
                                           a->grp_src_ptr, a->n_groups, a->seg_ptr, a->seg_node, a->dst, a->out_ptr, a->out_slot,
                                           H, a->GM, a->GP, a->gb_part, a->gm_amax, a->gp_amax, st));
    } else {
        ALIGNN_TRY(alignn_egc_bwd_dst(a->gy_out, a->M, a->P, a->gs1, a->gs0, a->e_stat, a->e_gamma, e_red, 0, m, a->seg_ptr,
                                      a->seg_node, a->src, n, H, a->GM, a->GP, a->gb_part, a->gm_amax, a->gp_amax, st));
        ALIGNN_TRY(alignn_egc_bwd_src(a->GM, a->M, a->gs1, a->out_ptr, a->out_slot, a->dst, n, H, a->GP, a->gp_amax, st));
    }
    // ---- input gradients (critical path): g_x = GP wcat (+ gx_out), g_y = GM w_eg (+ gy_out) - independent of each other:
    // with a second stream the (shorter) node product runs beside the edge product
    hipEvent_t ev_fork, ev_join;
    const bool forked = a->aux_stream != nullptr && fork_events(&ev_fork, &ev_join);
    alignn_stream_t st_main = st;
    if (forked) {
        ALIGNN_TRY((int)hipEventRecord(ev_fork, (hipStream_t)st_main));
        ALIGNN_TRY((int)hipStreamWaitEvent((hipStream_t)a->aux_stream, ev_fork, 0));
        st = a->aux_stream;
    }
    const float* addx = a->residual ? a->gx_out : nullptr;
    switch (a->dx_kind) {
        case 1:
            ALIGNN_TRY(alignn_gemm_nt_f16x3(a->GP, 4 * H, a->gp_amax, a->wcat_t_img, a->wcat_amax, nullptr, addx, addx ? H : 0, a->g_x, Kin, n,
                                            Kin, 4 * H, st));
            break;
        case 2:
            ALIGNN_TRY(alignn_gemm_nn_split(a->GP, 4 * H, a->wcat, Kin, addx, addx ? H : 0, a->g_x, Kin, n, 4 * H, Kin,
                                            a->scratch + s.dx_ws, s.dx_bytes, st));
            break;
        case 4:
            ALIGNN_TRY(alignn_gemm_nt(a->GP, 4 * H, a->wcat_t, 4 * H, nullptr, addx, addx ? H : 0, a->g_x, Kin, n, Kin, 4 * H, st));
            break;
        default:
            ALIGNN_TRY(alignn_gemm_nn(a->GP, 4 * H, a->wcat, Kin, addx, addx ? H : 0, a->g_x, Kin, n, 4 * H, Kin, st));
    }
    if (forked) {
        ALIGNN_TRY((int)hipEventRecord(ev_join, (hipStream_t)a->aux_stream));
        st = st_main;
    }
    const float* addy = (a->residual && a->gy_out != nullptr) ? a->gy_out : nullptr;
    switch (a->dy_kind) {
        case 1: {
            float* part = a->scratch + s.dy_part;
            ALIGNN_TRY(alignn_gemm_nt_f16x3_bnred(a->GM, H, a->gm_amax, a->weg_t_img, a->weg_amax, nullptr, addy, addy ? H : 0,
                                                  a->g_y, Kin, m, Kin, H, a->src_xn, a->src_ldxn, a->src_nstat, part, st));
            int tiles = s.dy_tiles;
            if (tiles > kFoldAbove) {
                ALIGNN_TRY(alignn_slab_fold(part, tiles, 2 * Kin, a->scratch + s.dy_fold, st));
                part = a->scratch + s.dy_fold;
                tiles = alignn_slab_fold_slabs();
            }
            ALIGNN_TRY(alignn_bn_bwd_finalize(part, tiles, Kin, a->src_red, st));
            break;
        }
        case 5:
            ALIGNN_TRY(alignn_gemm_nt_f16x3(a->GM, H, a->gm_amax, a->weg_t_img, a->weg_amax, nullptr, addy, addy ? H : 0, a->g_y,
                                            Kin, m, Kin, H, st));
            break;
        case 2: {
            const size_t bytes = alignn_gemm_nn_split_workspace(m, H, Kin);
            // (never chosen by the host code for edge rows today: the split kernel is for few-tile, long-reduction shapes)
            (void)bytes;
            return (int)hipErrorInvalidValue;
        }
        case 4:
            ALIGNN_TRY(alignn_gemm_nt(a->GM, H, a->weg_t, H, nullptr, addy, addy ? H : 0, a->g_y, Kin, m, Kin, H, st));
            break;
        default:
            ALIGNN_TRY(alignn_gemm_nn(a->GM, H, a->w_eg, Kin, addy, addy ? H : 0, a->g_y, Kin, m, H, Kin, st));
    }
    if (forked) ALIGNN_TRY((int)hipStreamWaitEvent((hipStream_t)st_main, ev_join, 0));
    return 0;
}

size_t alignn_egc_conv_wgrad_scratch(int64_t n, int64_t m, int H, int Kin) {
    return (al(alignn_gemm_tn_workspace(m, H, Kin) / 4 + 1) + al(alignn_gemm_tn_workspace(n, 4 * H, Kin) / 4 + 1) +
            al((size_t)alignn_col_stats_slabs(n) * 2 * 4 * H)) * sizeof(float);
}

int alignn_egc_conv_wgrad(const alignn_egc_wgrad_args* a, alignn_stream_t st) {
    if (a == nullptr || a->scratch == nullptr) return (int)hipErrorInvalidValue;
    const int H = a->H, Kin = a->Kin;
    const int64_t n = a->n, m = a->m;
    const size_t b1 = alignn_gemm_tn_workspace(m, H, Kin), b2 = alignn_gemm_tn_workspace(n, 4 * H, Kin);
    float* ws1 = a->scratch;
    float* ws2 = ws1 + al(b1 / 4 + 1);
    float* ws3 = ws2 + al(b2 / 4 + 1);
    if (a->scratch_bytes < alignn_egc_conv_wgrad_scratch(n, m, H, Kin)) return (int)hipErrorInvalidValue;
    ALIGNN_TRY(alignn_slab_sum(a->gb_part, a->gb_slabs, H, a->g_beg, st));
    ALIGNN_TRY(alignn_gemm_tn(a->GM, H, a->gm_amax, a->y, Kin, a->y_amax, a->g_weg, Kin, m, H, Kin, ws1, b1, st));
    ALIGNN_TRY(alignn_gemm_tn(a->GP, 4 * H, a->gp_amax, a->x, Kin, a->x_amax, a->g_wcat, Kin, n, 4 * H, Kin, ws2, b2, st));
    ALIGNN_TRY(alignn_col_sum(a->GP, 4 * H, n, 4 * H, a->g_bcat, ws3, st));
    return 0;
}

}  // extern "C"
