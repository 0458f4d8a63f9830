// Whole-model entry points: ONE C call = the forward (or the backward) of alignn.models.alignn.ALIGNN in training mode.
//
// The reference's training loop feeds a NEW (g, lg) batch every iteration (alignn/train.py:258-270), so nothing can be
// replayed from a captured hipGraph there: the ~350 kernels of a step have to be enqueued again, and when Python sequences
// them (alignn_amd/ops.py: ~95 autograd nodes, ~100 torch glue operations, one allocation per tensor) the host needs
// 9-23 ms per step against 15.8 ms of GPU work.  Here the same launches - same kernels, same arguments, therefore the same
// bits as the per-operator path - are issued from C over ONE workspace block whose layout is a pure function of the model
// dimensions and (N, E, T): no allocation, no interpreter, ~3 us per launch.  What ALIGNN.forward does
// (alignn/models/alignn.py:282-349): RBF + MLPLayer embeddings (:201-222, models/utils.py:11-44), alignn_layers x
// ALIGNNConv (:132-167: EdgeGatedGraphConv :48-129 on g, then on L(g)), gcn_layers x EdgeGatedGraphConv on g,
// AvgPooling + fc (:325,341); and torch.autograd's backward of all of it.
//
// Streams: the caller's stream plus up to three helpers (all optional; NULL = stay on the caller's stream):
//   lane_T  - the kernels over T = |edges of L(g)| rows (edge projection, gate pass, their backward, the angle embedding);
//   side    - weight / bias gradients (nothing on the critical path reads them);
//   aux     - the node input gradient of a bond-graph convolution beside the edge one.
// Dependencies are HIP events from a per-device pool (alignn_model_init); every helper stream is joined back into the
// caller's stream before a call returns, so the caller sees ordinary stream semantics and the calls are capture-safe.
//
// Workspace = [forward tape | backward temporaries]; the forward's tape (pre-activations, projections, statistics ...) is
// found again by the backward by re-running the same deterministic plan without launching.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/alignn_hip.h"
#include "common.h"

namespace {

constexpr int kFoldAbove = 1024;  // ops.FOLD_ABOVE
constexpr int kEvents = 512;
constexpr int kMaxDevices = 32;
constexpr int kAmaxSlots = 1024;

struct EventPool {
    hipEvent_t ev[kEvents];
    int next = 0;
    bool ok = false;
};
EventPool g_pool[kMaxDevices];

__global__ void fill_kernel(float* __restrict__ p, int64_t n, float v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void bump_kernel(int64_t* const* __restrict__ ptrs, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ptrs[i][0] += 1;
}
// out[c][r] = in[r][c]  (weights of at most a few hundred rows: one 32 x 32 tile per workgroup through LDS)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int64_t ld, int rows, int cols,
                                                        float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < rows && c0 + tx < cols) tile[j][tx] = in[(int64_t)(r0 + j) * ld + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < cols && r0 + tx < rows) out[(int64_t)(c0 + j) * rows + r0 + tx] = tile[tx][j];
}

struct Act {  // an activation [rows, F] and what its producer knows about it
    float* p = nullptr;
    float* amax = nullptr;  // device scalar max|p|, tracked by the producing kernel (NULL: not tracked)
    float* xn = nullptr;    // p = r + silu(BatchNorm(xn)): the pre-activation,
    float* stat = nullptr;  // its [4, F] statistics,
    float* red = nullptr;   // and where the producer's backward expects (sum gz, sum gz * xhat) [2, F]
    bool on_T = false;      // last written on lane T
};
struct Grad {
    float* p = nullptr;
    bool on_T = false;
    bool pre_red = false;  // the projection that wrote p also left the BatchNorm-backward sums in the producer's `red`
};

struct MlpTape {
    const alignn_mlp_params* p = nullptr;
    Act x, y;
    float *pre = nullptr, *stat = nullptr;
    int64_t rows = 0;
    bool lane = false;
};
struct ConvTape {
    const alignn_conv_params* p = nullptr;
    const alignn_graph_csr* g = nullptr;
    Act x, y, x_out, y_out;
    float *P = nullptr, *M = nullptr, *xpre = nullptr, *s0 = nullptr, *hh = nullptr, *n_stat = nullptr, *e_stat = nullptr;
    bool lane = false, need_y = true;
};
struct Tape {
    float *bl = nullptr, *rbf_e = nullptr, *rbf_a = nullptr, *pool = nullptr;
    float *hcos_buf = nullptr, *pred = nullptr, *seed = nullptr, *g_r = nullptr;  // force-field calls
    const float* hcos = nullptr;  // the bond-angle cosines the angle embedding read (batch->h, or recomputed from r)
    bool angle_fused = false, angle_lane = false;  // the angle embedding went through csrc/angle.hip: only these survive
    float *a_stat1 = nullptr, *a_stat2 = nullptr, *a_scal = nullptr;
    MlpTape atom, e1, e2, a1, a2;
    std::vector<ConvTape> convs;
};

std::vector<std::pair<size_t, size_t>> g_dbg_allocs;  // (offset, bytes) of every persistent allocation of the last launching call
bool dbg_allocs_on() {
    static int on = -1;
    if (on < 0) on = getenv("ALIGNN_AMD_DEBUG_ALLOCS") ? 1 : 0;
    return on == 1;
}

struct Ctx {
    const alignn_model_desc* d;
    const alignn_model_batch* b;
    char* base;
    size_t off = 0, cap = 0;
    bool launch = false;
    bool unsupported = false;
    int unsupported_line = 0;
    int rc = 0;
    hipStream_t main = nullptr, T = nullptr, side = nullptr, aux = nullptr;
    float* amax_arena = nullptr;
    int amax_next = 0;
    EventPool* pool = nullptr;

    float* alloc(size_t floats) {
        const size_t bytes = (floats * sizeof(float) + 255) / 256 * 256;
        float* p = reinterpret_cast<float*>(base + off);
        if (launch && dbg_allocs_on()) g_dbg_allocs.emplace_back(off, bytes);
        off += bytes;
        if (launch && off > cap && rc == 0) rc = (int)hipErrorInvalidValue;
        return p;
    }
    // stream-local scratch: transient buffers (split-reduction slabs, transposed weights) that only kernels of ONE stream
    // touch, one after the other - the region is rewound (tmp_reset) before the next group of launches on that stream
    // reuses it; stream order is the only synchronisation it needs (a captured graph keeps those edges)
    size_t sbase[4] = {0, 0, 0, 0}, scur[4] = {0, 0, 0, 0}, speak[4] = {0, 0, 0, 0};
    int which(hipStream_t st) const { return st == main ? 0 : st == T ? 1 : st == side ? 2 : 3; }
    void tmp_reset(hipStream_t st) { scur[which(st)] = 0; }
    float* tmp(hipStream_t st, size_t floats) {
        const int i = which(st);
        const size_t bytes = (floats * sizeof(float) + 255) / 256 * 256;
        float* p = reinterpret_cast<float*>(base + sbase[i] + scur[i]);
        scur[i] += bytes;
        if (scur[i] > speak[i]) speak[i] = scur[i];
        if (launch && sbase[i] + scur[i] > cap && rc == 0) rc = (int)hipErrorInvalidValue;
        return p;
    }
    float* new_amax() {
        if (amax_next >= kAmaxSlots) {
            if (rc == 0) rc = (int)hipErrorInvalidValue;
            return amax_arena;
        }
        return amax_arena + amax_next++;
    }
    float* new_amax2() {  // two consecutive scalars: (value, tangent) of a dual activation
        if (amax_next + 2 > kAmaxSlots) {
            if (rc == 0) rc = (int)hipErrorInvalidValue;
            return amax_arena;
        }
        amax_next += 2;
        return amax_arena + amax_next - 2;
    }
    bool track(int64_t rows) const { return rows >= d->amax_min_rows; }
    // force-field calls (alignn_ff_eval / alignn_ff_grad)
    const alignn_ff_desc* ff = nullptr;
    bool param_grads = true;  // false: the reverse pass of alignn_ff_eval (gradients w.r.t. the bond vectors only)
    bool geom = false;        // the first embedding layers hand back the gradient of their RBF input
    float *g_rbf_e = nullptr, *g_rbf_a = nullptr;
    bool g_rbf_a_on_T = false;
    ptrdiff_t toff = 0;  // floats from a weight-gradient destination to its tangent twin (alignn_ff_grad)
    // a second region of gradient destinations with tangent twins of its own: the optimizer's packed gradient buffer
    float* sink = nullptr;
    int64_t sink_floats = 0;
    ptrdiff_t sink_toff = 0;
    float* twin(float* dW) const { return dW + ((sink != nullptr && dW >= sink && dW < sink + sink_floats) ? sink_toff : toff); }
    // `waiter` continues only after everything enqueued on `src` so far
    int n_sync = 0, n_launch = 0;
    double t_sync = 0.0;
    bool timing = false;
    void sync(hipStream_t waiter, hipStream_t src) {
        if (!launch || rc != 0 || waiter == src) return;
        const auto t0 = timing ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
        hipEvent_t e = pool->ev[pool->next];
        pool->next = (pool->next + 1) % kEvents;
        rc = (int)hipEventRecord(e, src);
        if (rc == 0) rc = (int)hipStreamWaitEvent(waiter, e, 0);
        ++n_sync;
        if (timing) t_sync += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
};

#define UNSUP()                         \
    do {                                \
        c.unsupported = true;           \
        if (!c.unsupported_line) c.unsupported_line = __LINE__; \
    } while (0)

#define L(call)                                                                                        \
    do {                                                                                               \
        if (c.launch && c.rc == 0) {                                                                   \
            c.rc = (call);                                                                             \
            ++c.n_launch;                                                                              \
            if (c.rc != 0 && getenv("ALIGNN_AMD_DEBUG")) fprintf(stderr, "model.hip:%d rc %d: %s\n", __LINE__, c.rc, #call); \
        }                                                                                              \
    } while (0)

inline bool x6_shape_ok(const Ctx& c, int64_t M, int64_t lda, int N, int K) {
    const int64_t tiles = ((M + 63) / 64) * ((N + 255) / 256);
    return tiles >= c.d->x6_min_tiles && (lda % 4) == 0 && alignn_gemm_nt_x6_supported(M, N, K) != 0;
}

void fill(Ctx& c, float* p, int64_t n, float v, hipStream_t st) {
    if (!c.launch || c.rc != 0 || n <= 0) return;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n, v);
    c.rc = (int)hipGetLastError();
}

// ops.project (forward products): split-product kernel for the wide, deep shapes, exact-fp32 MFMA otherwise
void project(Ctx& c, const float* a, int64_t lda, const float* a_amax, const float* w, int64_t ldw, const void* img,
             const float* w_amax, const float* bias, float* out, int64_t ldo, int64_t M, int N, int K, hipStream_t st) {
    if (x6_shape_ok(c, M, lda, N, K)) {
        if (a_amax == nullptr || img == nullptr) {  // (the per-operator path would take bf16x6 / slice for itself)
            if (getenv("ALIGNN_AMD_DEBUG")) fprintf(stderr, "project M=%lld N=%d K=%d amax=%p img=%p\n", (long long)M, N, K, (void*)a_amax, img);
            UNSUP();
            return;
        }
        L(alignn_gemm_nt_f16x3(a, lda, a_amax, img, w_amax, bias, nullptr, 0, out, ldo, M, N, K, st));
    } else
        L(alignn_gemm_nt(a, lda, w, ldw, bias, nullptr, 0, out, ldo, M, N, K, st));
}

void bn_finalize_folded(Ctx& c, float* partial, int slabs, int64_t rows, int F, const float* gamma, const float* beta,
                        float* rm, float* rv, float* stat, hipStream_t st) {
    if (slabs > kFoldAbove) {
        float* fold = c.alloc((size_t)alignn_slab_fold_slabs() * 2 * F);
        L(alignn_slab_fold(partial, slabs, 2 * F, fold, st));
        partial = fold;
        slabs = alignn_slab_fold_slabs();
    }
    L(alignn_bn_finalize(partial, slabs, rows, F, gamma, beta, c.d->eps, c.d->momentum, rm, rv, stat, st));
}

void bn_bwd_finalize_folded(Ctx& c, float* partial, int slabs, int F, float* red, hipStream_t st) {
    if (slabs > kFoldAbove) {
        float* fold = c.alloc((size_t)alignn_slab_fold_slabs() * 2 * F);
        L(alignn_slab_fold(partial, slabs, 2 * F, fold, st));
        partial = fold;
        slabs = alignn_slab_fold_slabs();
    }
    L(alignn_bn_bwd_finalize(partial, slabs, F, red, st));
}

// ---------------------------------------------------------------------------------------------------------------------
// forward pieces
// ---------------------------------------------------------------------------------------------------------------------

// MLPLayer = Linear + BatchNorm1d (batch statistics) + SiLU, alignn/models/alignn.py:170-184  (ops.MLPLayerFn._fwd)
Act mlp_fwd_ln(Ctx& c, MlpTape& t, const alignn_mlp_params& p, const Act& x, int64_t rows);
Act mlp_fwd(Ctx& c, MlpTape& t, const alignn_mlp_params& p, const Act& x, int64_t rows) {
    if (c.d->norm == 1) return mlp_fwd_ln(c, t, p, x, rows);
    const int F = p.out, K = p.in;
    t.p = &p;
    t.x = x;
    t.rows = rows;
    t.lane = c.T != c.main && rows >= c.d->lane_min_rows;
    hipStream_t st = t.lane ? c.T : c.main;
    if (t.lane != x.on_T) c.sync(st, x.on_T ? c.T : c.main);
    t.pre = c.alloc((size_t)rows * F);
    t.stat = c.alloc((size_t)4 * F);
    const bool fused_stats = x.amax != nullptr && x6_shape_ok(c, rows, K, F, K);
    if (fused_stats) {  // the projection's epilogue leaves the column sums BatchNorm needs
        if (p.img == nullptr) {
            UNSUP();
            return Act{};
        }
        const int tiles = alignn_gemm_nt_x6_row_tiles(rows, F, K);
        float* partial = c.alloc((size_t)(tiles + 1) * 2 * F);
        L(alignn_gemm_nt_f16x3_stats(x.p, K, x.amax, p.img, p.w_amax, p.b, t.pre, F, rows, F, K, partial, st));
        bn_finalize_folded(c, partial, tiles, rows, F, p.gamma, p.beta, p.rm, p.rv, t.stat, st);
    } else {
        project(c, x.p, K, x.amax, p.W, K, p.img, p.w_amax, p.b, t.pre, F, rows, F, K, st);
        const int slabs = alignn_col_stats_slabs(rows);
        float* partial = c.alloc((size_t)slabs * (3 * F + 1));
        L(alignn_col_stats_welford(t.pre, F, rows, F, partial, st));
        L(alignn_bn_finalize_welford(partial, slabs, rows, F, p.gamma, p.beta, c.d->eps, c.d->momentum, p.rm, p.rv, t.stat, st));
    }
    Act y;
    y.p = c.alloc((size_t)rows * F);
    y.amax = c.track(rows) ? c.new_amax() : nullptr;
    L(alignn_bn_silu_fwd(t.pre, F, nullptr, 0, t.stat, y.p, F, rows, F, y.amax, st));
    y.xn = t.pre;
    y.stat = t.stat;
    y.red = p.red;
    y.on_T = t.lane;
    t.y = y;
    return y;
}

// EdgeGatedGraphConv.forward, alignn/models/alignn.py:78-129  (ops.EdgeGatedConvFn.forward, BatchNorm / training)
void conv_fwd_ln(Ctx& c, ConvTape& t, const alignn_conv_params& p, const alignn_graph_csr& g, const Act& x, const Act& y,
                 bool need_y);
void conv_fwd(Ctx& c, ConvTape& t, const alignn_conv_params& p, const alignn_graph_csr& g, const Act& x, const Act& y,
              bool need_y) {
    if (c.d->norm == 1) return conv_fwd_ln(c, t, p, g, x, y, need_y);
    const int H = c.d->H, Kin = H;
    const int64_t n = g.n, m = g.m;
    t.p = &p;
    t.g = &g;
    t.x = x;
    t.y = y;
    t.need_y = need_y;
    t.lane = c.T != c.main && m >= c.d->lane_min_rows;
    hipStream_t main = c.main, T = t.lane ? c.T : c.main;
    if (x.on_T) c.sync(main, c.T);
    if (!t.lane && y.on_T) c.sync(main, c.T);
    // ---- node side, part 1: P = x [W_sg; W_dg; W_du; W_su]^T + b = A | Bd | Bh | Ux
    t.P = c.alloc((size_t)n * 4 * H);
    project(c, x.p, Kin, x.amax, p.wcat, Kin, p.wcat_img, p.wcat_amax, p.bcat, t.P, 4 * H, n, 4 * H, Kin, main);
    t.xpre = c.alloc((size_t)n * H);
    t.s0 = c.alloc((size_t)n * H);
    t.hh = c.alloc((size_t)n * H);
    const int n_slabs = alignn_egc_slabs(n);
    float* n_part = c.alloc((size_t)n_slabs * (3 * H + 1));
    t.M = c.alloc((size_t)m * H);
    t.n_stat = c.alloc((size_t)8 * H);
    t.e_stat = t.n_stat + 4 * H;
    // u_add_v (and the BatchNorm statistics) in the edge projection's epilogue when it runs on the split-product kernel
    const bool pre_added = y.amax != nullptr && x6_shape_ok(c, m, Kin, H, Kin);
    if (!pre_added && x6_shape_ok(c, m, Kin, H, Kin)) UNSUP();  // (bf16x6 scheme: per-operator path)
    if (t.lane) c.sync(T, main);  // lane T reads P (and y, if the caller's stream produced it)
    Act yo;
    if (need_y) {
        yo.p = c.alloc((size_t)m * H);
        yo.amax = c.track(m) ? c.new_amax() : nullptr;
    }
    if (pre_added) {
        if (p.weg_img == nullptr) {
            UNSUP();
            return;
        }
        const int tiles = alignn_gemm_nt_x6_row_tiles(m, H, Kin);
        float* e_part = c.alloc((size_t)(tiles + 1) * 2 * H);
        if (c.d->bd_segment_table && g.seg_node != nullptr && g.seg_rank != nullptr) {
            // line graphs: the destination term from a segment-ordered copy of Bd (consecutive rows, same values)
            float* bd2 = c.alloc((size_t)n * H);
            L(alignn_gather_rows_ld(t.P + H, 4 * H, g.seg_node, bd2, H, n, H, T));
            L(alignn_gemm_nt_f16x3_gather2(y.p, Kin, y.amax, p.weg_img, p.weg_amax, p.b_eg, t.M, H, m, H, Kin, t.P, 4 * H, g.src,
                                           bd2, H, g.seg_rank, e_part, T));
        } else
            L(alignn_gemm_nt_f16x3_gather(y.p, Kin, y.amax, p.weg_img, p.weg_amax, p.b_eg, t.M, H, m, H, Kin, t.P, 4 * H, g.src,
                                          g.dst, e_part, T));
        bn_finalize_folded(c, e_part, tiles, m, H, p.e_gamma, p.e_beta, p.e_rm, p.e_rv, t.e_stat, T);
        if (need_y)
            L(alignn_egc_gate_fwd_pre_norm(t.P, t.M, g.seg_ptr, g.seg_node, g.src, n, m, H, t.xpre, t.s0, t.hh, n_part, t.e_stat,
                                           y.p, yo.p, yo.amax, T));
        else
            L(alignn_egc_gate_fwd_pre(t.P, t.M, g.seg_ptr, g.seg_node, g.src, n, m, H, t.xpre, t.s0, t.hh, nullptr, n_part, T));
        if (t.lane) c.sync(main, T);
    } else {
        L(alignn_gemm_nt(y.p, Kin, p.w_eg, Kin, p.b_eg, nullptr, 0, t.M, H, m, H, Kin, T));
        float* e_part = c.alloc((size_t)n_slabs * (3 * H + 1));
        L(alignn_egc_gate_fwd(t.P, t.M, g.seg_ptr, g.seg_node, g.src, n, m, H, t.xpre, t.s0, t.hh, e_part, n_part, T));
        if (t.lane) c.sync(main, T);
        L(alignn_bn_finalize_welford(e_part, n_slabs, m, H, p.e_gamma, p.e_beta, c.d->eps, c.d->momentum, p.e_rm, p.e_rv,
                                     t.e_stat, T));
        if (need_y) L(alignn_bn_silu_fwd(t.M, H, y.p, Kin, t.e_stat, yo.p, H, m, H, yo.amax, T));
    }
    // ---- node side, part 2: node norm
    L(alignn_bn_finalize_welford(n_part, n_slabs, n, H, p.n_gamma, p.n_beta, c.d->eps, c.d->momentum, p.n_rm, p.n_rv, t.n_stat,
                                 main));
    Act xo;
    xo.p = c.alloc((size_t)n * H);
    xo.amax = c.track(n) ? c.new_amax() : nullptr;
    L(alignn_bn_silu_fwd(t.xpre, H, x.p, Kin, t.n_stat, xo.p, H, n, H, xo.amax, main));
    if (need_y) {
        yo.xn = t.M;
        yo.stat = t.e_stat;
        yo.red = p.e_red;
        yo.on_T = t.lane;
    }
    t.x_out = xo;
    t.y_out = yo;
}

// ---- LayerNorm flavour (ALIGNNAtomWise: alignn/models/alignn_atomwise.py:127-208 EdgeGatedGraphConv, alignn/models/utils.py:277-292
// MLPLayer): per-row statistics, no grid-wide dependency, stat = [rows, 2] (mean, rstd).  ops.MLPLayerFn / ops.EdgeGatedConvFn
// with norm == "layer", launch for launch.
Act mlp_fwd_ln(Ctx& c, MlpTape& t, const alignn_mlp_params& p, const Act& x, int64_t rows) {
    const int F = p.out, K = p.in;
    t.p = &p;
    t.x = x;
    t.rows = rows;
    t.lane = c.T != c.main && rows >= c.d->lane_min_rows;
    hipStream_t st = t.lane ? c.T : c.main;
    if (t.lane != x.on_T) c.sync(st, x.on_T ? c.T : c.main);
    t.pre = c.alloc((size_t)rows * F);
    project(c, x.p, K, x.amax, p.W, K, p.img, p.w_amax, p.b, t.pre, F, rows, F, K, st);
    t.stat = c.alloc((size_t)rows * 2);
    Act y;
    y.p = c.alloc((size_t)rows * F);
    y.amax = c.track(rows) ? c.new_amax() : nullptr;
    L(alignn_ln_silu_fwd(t.pre, F, nullptr, 0, p.gamma, p.beta, c.d->eps, y.p, F, t.stat, rows, F, y.amax, st));
    y.on_T = t.lane;
    t.y = y;
    return y;
}

void conv_fwd_ln(Ctx& c, ConvTape& t, const alignn_conv_params& p, const alignn_graph_csr& g, const Act& x, const Act& y,
                 bool need_y) {
    const int H = c.d->H, Kin = H;
    const int64_t n = g.n, m = g.m;
    t.p = &p;
    t.g = &g;
    t.x = x;
    t.y = y;
    t.need_y = need_y;
    t.lane = c.T != c.main && m >= c.d->lane_min_rows;
    hipStream_t main = c.main, T = t.lane ? c.T : c.main;
    if (x.on_T) c.sync(main, c.T);
    if (!t.lane && y.on_T) c.sync(main, c.T);
    t.P = c.alloc((size_t)n * 4 * H);
    project(c, x.p, Kin, x.amax, p.wcat, Kin, p.wcat_img, p.wcat_amax, p.bcat, t.P, 4 * H, n, 4 * H, Kin, main);
    t.xpre = c.alloc((size_t)n * H);
    t.s0 = c.alloc((size_t)n * H);
    t.hh = c.alloc((size_t)n * H);
    t.M = c.alloc((size_t)m * H);
    const bool x6 = x6_shape_ok(c, m, Kin, H, Kin);
    const bool pre_added = y.amax != nullptr && x6;  // u_add_v in the edge projection's epilogue (no statistics to take)
    if (!pre_added && x6) UNSUP();                   // (bf16x6 scheme: per-operator path)
    if (t.lane) c.sync(T, main);
    if (pre_added) {
        if (p.weg_img == nullptr) {
            UNSUP();
            return;
        }
        if (c.d->bd_segment_table && g.seg_node != nullptr && g.seg_rank != nullptr) {
            float* bd2 = c.alloc((size_t)n * H);
            L(alignn_gather_rows_ld(t.P + H, 4 * H, g.seg_node, bd2, H, n, H, T));
            L(alignn_gemm_nt_f16x3_gather2(y.p, Kin, y.amax, p.weg_img, p.weg_amax, p.b_eg, t.M, H, m, H, Kin, t.P, 4 * H, g.src,
                                           bd2, H, g.seg_rank, nullptr, T));
        } else
            L(alignn_gemm_nt_f16x3_gather(y.p, Kin, y.amax, p.weg_img, p.weg_amax, p.b_eg, t.M, H, m, H, Kin, t.P, 4 * H, g.src,
                                          g.dst, nullptr, T));
        if (!(need_y && alignn_egc_ln_fused_supported(H, m)))
            L(alignn_egc_gate_fwd_pre(t.P, t.M, g.seg_ptr, g.seg_node, g.src, n, m, H, t.xpre, t.s0, t.hh, nullptr, nullptr, T));
    } else {
        L(alignn_gemm_nt(y.p, Kin, p.w_eg, Kin, p.b_eg, nullptr, 0, t.M, H, m, H, Kin, T));
        L(alignn_egc_gate_fwd(t.P, t.M, g.seg_ptr, g.seg_node, g.src, n, m, H, t.xpre, t.s0, t.hh, nullptr, nullptr, T));
    }
    Act yo;
    if (need_y) {
        yo.p = c.alloc((size_t)m * H);
        yo.amax = c.track(m) ? c.new_amax() : nullptr;
        t.e_stat = c.alloc((size_t)m * 2);
        yo.on_T = t.lane;
    }
    if (need_y && pre_added && alignn_egc_ln_fused_supported(H, m)) {
        // the edge LayerNorm inside the gate pass (csrc/convln.hip): one read of m less
        L(alignn_egc_gate_fwd_pre_ln(t.P, t.M, g.seg_ptr, g.seg_node, g.src, n, m, H, t.xpre, t.s0, t.hh, p.e_gamma, p.e_beta, c.d->eps,
                                     y.p, yo.p, t.e_stat, yo.amax, T));
        if (t.lane) c.sync(main, T);
    } else {
        if (t.lane) c.sync(main, T);
        if (need_y) L(alignn_ln_silu_fwd(t.M, H, y.p, Kin, p.e_gamma, p.e_beta, c.d->eps, yo.p, H, t.e_stat, m, H, yo.amax, T));
    }
    t.n_stat = c.alloc((size_t)n * 2);
    Act xo;
    xo.p = c.alloc((size_t)n * H);
    xo.amax = c.track(n) ? c.new_amax() : nullptr;
    L(alignn_ln_silu_fwd(t.xpre, H, x.p, Kin, p.n_gamma, p.n_beta, c.d->eps, xo.p, H, t.n_stat, n, H, xo.amax, main));
    t.x_out = xo;
    t.y_out = yo;
}

int conv_count(const alignn_model_desc& d) { return 2 * d.alignn_layers + d.gcn_layers; }

// the arguments both directions of the fused angle embedding share; the three small buffers live on the forward tape
alignn_angle_args angle_args(Ctx& c, const Tape& tp) {
    const alignn_model_desc& d = *c.d;
    alignn_angle_args a{};
    a.h = tp.hcos;
    a.rows = c.b->lg.m;
    a.centers = d.angle_centers;
    a.gamma = d.angle_gamma;
    a.bins = d.angle_bins;
    a.l1 = d.angle1;
    a.l2 = d.angle2;
    a.eps = d.eps;
    a.momentum = d.momentum;
    a.stat1 = tp.a_stat1;
    a.stat2 = tp.a_stat2;
    a.scal = tp.a_scal;
    return a;
}

// the whole forward; with c.launch == false only the workspace plan (the tape's pointers) is produced
void run_forward(Ctx& c, Tape& tp, float* out) {
    const alignn_model_desc& d = *c.d;
    const alignn_model_batch& b = *c.b;
    const int64_t N = b.g.n, E = b.g.m, Tn = b.lg.m;
    const int H = d.H;
    c.amax_arena = c.alloc(kAmaxSlots);
    c.amax_next = 0;
    fill(c, c.amax_arena, kAmaxSlots, 0.0f, c.main);
    // The fused bond-angle embedding (csrc/angle.hip) reads the two angle layers' float32 weights itself - no slice images -, so
    // lane T may start it while the caller's stream still prepares the images (45 us at the head of lane T's chain otherwise).
    const bool angle_first = c.ff == nullptr && d.norm == 0 && d.angle_fused != 0 && Tn > 0 && c.T != c.main && Tn >= d.lane_min_rows &&
                             alignn_angle_embed_supported(d.angle_bins, d.angle1.out, d.angle2.out) != 0;
    if (angle_first) c.sync(c.T, c.main);  // (the zeroed arena)
    // The slice images of the weights (max|W| + two images each: 3 launches, ~55 us) are first read by the second bond-embedding
    // layer: with a helper stream they are made BESIDE the atom embedding and the first bond-embedding layer (exact-fp32
    // products, no images) instead of in front of them - the caller's stream is the critical chain at the head of a step (the
    // first bond-graph convolution hangs on it; lane T's angle embedding finishes earlier).
    const bool prep_aside = angle_first && c.aux != c.main && d.n_weights > 0 && c.ff == nullptr &&
                            !x6_shape_ok(c, N, d.atom.in, d.atom.out, d.atom.in) &&     // (the layers in between take the
                            !x6_shape_ok(c, E, d.edge1.in, d.edge1.out, d.edge1.in);    //  exact-fp32 product: no image read)
    hipStream_t prep_st = prep_aside ? c.aux : c.main;
    if (prep_aside) c.sync(c.aux, c.main);
    if (d.n_weights > 0) L(alignn_prepare_weights(d.weight_descs, d.n_weights, d.weight_amax, prep_st));
    if (d.n_bump > 0 && c.launch && c.rc == 0) {
        hipLaunchKernelGGL(bump_kernel, dim3((d.n_bump + 63) / 64), dim3(64), 0, c.main, (int64_t* const*)d.bump_ptrs, d.n_bump);
        c.rc = (int)hipGetLastError();
    }
    // ---- bond-angle cosines: the loader's lg.edata["h"], or - force field with lg_on_fly - recomputed from the bond vectors
    // (compute_bond_cosines inside the forward, alignn_atomwise.py:424-431: the cosines then carry a gradient w.r.t. r)
    tp.hcos = b.h;
    if (c.ff != nullptr && c.ff->lg_on_fly) {
        tp.hcos_buf = c.alloc((size_t)Tn);
        L(alignn_bond_cosine_fwd(b.r, b.lg.src, b.lg.dst, tp.hcos_buf, Tn, c.main));
        tp.hcos = tp.hcos_buf;
    }
    if (!angle_first) c.sync(c.T, c.main);  // parameters, weight images, the zeroed arena
    // ---- angle embedding (T rows: lane T), alignn.py:215-222
    Act z;
    tp.angle_fused = d.norm == 0 && d.angle_fused != 0 && Tn > 0 &&
                     alignn_angle_embed_supported(d.angle_bins, d.angle1.out, d.angle2.out) != 0;
    if (tp.angle_fused) {  // recomputing passes: nothing T x bins / T x 64 / T x 256 but z itself is written (csrc/angle.hip)
        tp.angle_lane = c.T != c.main && Tn >= d.lane_min_rows;
        hipStream_t st = tp.angle_lane ? c.T : c.main;
        tp.a_stat1 = c.alloc((size_t)4 * d.angle1.out);
        tp.a_stat2 = c.alloc((size_t)4 * d.angle2.out);
        tp.a_scal = c.alloc((size_t)alignn_angle_embed_scal_floats());
        alignn_angle_args a = angle_args(c, tp);
        z.p = c.alloc((size_t)Tn * H);
        z.amax = c.track(Tn) ? c.new_amax() : nullptr;
        z.on_T = tp.angle_lane;
        a.z = z.p;
        a.z_amax = z.amax;
        a.workspace_bytes = alignn_angle_embed_workspace(Tn, d.angle_bins, 0);
        a.workspace = c.alloc(a.workspace_bytes / sizeof(float));
        L(alignn_angle_embed_fwd(&a, st));
    } else {
        tp.rbf_a = c.alloc((size_t)Tn * d.angle_bins);
        L(alignn_rbf_fwd(tp.hcos, d.angle_centers, d.angle_gamma, tp.rbf_a, Tn, d.angle_bins, c.main));
        Act za;
        za.p = tp.rbf_a;
        z = mlp_fwd(c, tp.a2, d.angle2, mlp_fwd(c, tp.a1, d.angle1, za, Tn), Tn);
    }
    // ---- atom embedding, alignn.py:197-199
    Act xa;
    xa.p = const_cast<float*>(b.atom_features);
    Act x = mlp_fwd(c, tp.atom, d.atom, xa, N);
    // ---- edge embedding, alignn.py:201-214,313
    tp.bl = c.alloc((size_t)E);
    L(alignn_norm3_fwd(b.r, tp.bl, E, c.main));
    tp.rbf_e = c.alloc((size_t)E * d.edge_bins);
    L(alignn_rbf_fwd(tp.bl, d.edge_centers, d.edge_gamma, tp.rbf_e, E, d.edge_bins, c.main));
    Act ye;
    ye.p = tp.rbf_e;
    Act y1 = mlp_fwd(c, tp.e1, d.edge1, ye, E);
    if (prep_aside) c.sync(c.main, c.aux);  // the weight images: everything later is ordered behind the caller's stream from here
    Act y = mlp_fwd(c, tp.e2, d.edge2, y1, E);
    if (c.unsupported) return;
    // ---- ALIGNN layers (alignn.py:317-319), then GCN layers (:322-323); dead last-layer outputs are not materialised
    tp.convs.assign(conv_count(d), ConvTape{});
    int k = 0;
    for (int i = 0; i < d.alignn_layers; ++i) {
        ConvTape& tg = tp.convs[k];
        conv_fwd(c, tg, d.convs[k], b.g, x, y, true);
        ++k;
        ConvTape& tl = tp.convs[k];
        conv_fwd(c, tl, d.convs[k], b.lg, tg.y_out, z, i + 1 < d.alignn_layers);
        ++k;
        x = tg.x_out;
        y = tl.x_out;
        z = tl.y_out;
        if (c.unsupported) return;
    }
    for (int i = 0; i < d.gcn_layers; ++i) {
        ConvTape& tg = tp.convs[k];
        conv_fwd(c, tg, d.convs[k], b.g, x, y, i + 1 < d.gcn_layers);
        ++k;
        x = tg.x_out;
        y = tg.y_out;
        if (c.unsupported) return;
    }
    // ---- readout: AvgPooling + fc, alignn.py:325,341
    if (x.on_T) c.sync(c.main, c.T);
    tp.pool = c.alloc((size_t)b.B * H);
    L(alignn_segment_mean_fwd(x.p, b.graph_ptr, tp.pool, b.B, H, c.main));
    L(alignn_gemm_nt(tp.pool, H, d.fc_W, H, d.fc_b, nullptr, 0, out, d.out_features, b.B, d.out_features, H, c.main));
    c.sync(c.main, c.T);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward pieces
// ---------------------------------------------------------------------------------------------------------------------

void transpose(Ctx& c, const float* w, int64_t ld, int rows, int cols, float* out, hipStream_t st) {
    if (!c.launch || c.rc != 0) return;
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, st, w, ld, rows, cols, out);
    c.rc = (int)hipGetLastError();
}

// ops._dgrad / ops._dgrad_bnred: out[M, Kout] = g[M, Nred] w[Nred, Kout] (+ addend); with `src` (the norm the tensor this
// gradient belongs to came out of) and the split-product kernel, the BatchNorm-backward sums are taken in the epilogue
void dgrad(Ctx& c, const float* g, int64_t ldg, const float* g_amax, const float* w, int Nred, int Kout, const void* img_t,
           const float* w_amax, const float* addend, int64_t ldadd, float* out, int64_t M, hipStream_t st, const Act* src,
           bool* pre_red) {
    if (pre_red) *pre_red = false;
    const bool x6 = x6_shape_ok(c, M, ldg, Kout, Nred);
    if (x6) {
        if (g_amax == nullptr || img_t == nullptr) {
            UNSUP();
            return;
        }
        if (src != nullptr && src->xn != nullptr) {  // (never on `red`: a gradient pointer is not set yet when the plan is made)
            const int tiles = alignn_gemm_nt_x6_row_tiles(M, Kout, Nred);
            float* part = c.alloc((size_t)(tiles + 1) * 2 * Kout);
            L(alignn_gemm_nt_f16x3_bnred(g, ldg, g_amax, img_t, w_amax, nullptr, addend, ldadd, out, Kout, M, Kout, Nred, src->xn,
                                         Kout, src->stat, part, st));
            bn_bwd_finalize_folded(c, part, tiles, Kout, src->red, st);
            if (pre_red) *pre_red = true;
            return;
        }
        L(alignn_gemm_nt_f16x3(g, ldg, g_amax, img_t, w_amax, nullptr, addend, ldadd, out, Kout, M, Kout, Nred, st));
        return;
    }
    const size_t sb = alignn_gemm_nn_split_workspace(M, Nred, Kout);
    if (sb) {
        if ((ldg % 4) == 0 && (Kout % 4) == 0 && (addend == nullptr || (ldadd % 4) == 0)) {
            c.tmp_reset(st);
            float* ws = c.tmp(st, sb / 4);
            L(alignn_gemm_nn_split(g, ldg, w, Kout, addend, ldadd, out, Kout, M, Nred, Kout, ws, sb, st));
        } else
            L(alignn_gemm_nn(g, ldg, w, Kout, addend, ldadd, out, Kout, M, Nred, Kout, st));
        return;
    }
    if ((Nred % 4) == 0 && Kout >= 16) {  // reduction-contiguous (NT) kernel on a transposed copy of the weight
        c.tmp_reset(st);
        float* wt = c.tmp(st, (size_t)Kout * Nred);
        transpose(c, w, Kout, Nred, Kout, wt, st);
        L(alignn_gemm_nt(g, ldg, wt, Nred, nullptr, addend, ldadd, out, Kout, M, Kout, Nred, st));
        return;
    }
    L(alignn_gemm_nn(g, ldg, w, Kout, addend, ldadd, out, Kout, M, Nred, Kout, st));
}

void gemm_tn(Ctx& c, const float* g, int64_t ldg, const float* g_amax, const float* a, int64_t lda, const float* a_amax,
             float* dW, int64_t M, int N, int K, hipStream_t st) {
    if (g_amax == nullptr || a_amax == nullptr) g_amax = a_amax = nullptr;
    const size_t nb = alignn_gemm_tn_workspace(M, N, K);
    float* ws = c.tmp(st, nb / 4 + 1);
    L(alignn_gemm_tn(g, ldg, g_amax, a, lda, a_amax, dW, K, M, N, K, ws, nb, st));
}

void col_sum(Ctx& c, const float* x, int64_t ldx, int64_t rows, int F, float* out, hipStream_t st) {
    float* ws = c.tmp(st, (size_t)alignn_col_stats_slabs(rows) * 2 * F);
    L(alignn_col_sum(x, ldx, rows, F, out, ws, st));
}

// where a weight-gradient product over `rows` rows goes: the side stream, or (few rows, eagerly launched) the caller's
hipStream_t side_for(Ctx& c, int64_t rows) { return (c.side != c.main && rows >= c.d->side_min_rows) ? c.side : c.main; }

// Where the gradient of a convolution's edge input goes.  With desc->reuse_tape: over the convolution's own gate pre-activation
// m (t.M) - dead by then: its last readers (the norm's reductions, the gate reverse kernels) precede the product that writes the
// gradient on the SAME stream, and that product reads the pre-activation of the convolution BEFORE this one (the BatchNorm-
// backward epilogue's xn), never its own.  One T-row buffer per line-graph convolution less (2.8 GB at the benchmark batch).
// Not for the force-field calls: their second-order pass reads the evaluation's tape.
float* edge_grad_buffer(Ctx& c, const ConvTape& t, int64_t m, int Kin) {
    if (c.d->reuse_tape && c.ff == nullptr && t.M != nullptr && Kin == c.d->H) return t.M;
    return c.alloc((size_t)m * Kin);
}

// Backward of the edge-gate projection m = y W_eg^T (+ ...) in ONE pass over g_m (csrc/gemm_dw.hip) where its shape allows:
// g_y = g_m W_eg (+ gy) with the BatchNorm-backward sums of `src`, and dW_eg = g_m^T y.  -> false: the caller takes the two
// launches (dgrad + gemm_tn).  ops.gemm_dgrad_wgrad under the same rule (ops.dgrad_wgrad_applies).
bool dgrad_wgrad(Ctx& c, const ConvTape& t, const float* GM, const float* gm_amax, const float* gy, float* g_y, int64_t m, int H,
                 int Kin, hipStream_t st, const Act* src, bool* pre_red) {
    const alignn_conv_params& p = *t.p;
    if (c.d->dw_fused <= 0 || m < c.d->dw_fused || !c.param_grads || gm_amax == nullptr || t.y.amax == nullptr || p.weg_img_t == nullptr || Kin != H ||
        !alignn_gemm_dgrad_wgrad_supported(m, H, Kin) || !x6_shape_ok(c, m, H, Kin, H))
        return false;
    const size_t nb = alignn_gemm_dgrad_wgrad_workspace(m);
    c.tmp_reset(st);
    float* ws = c.tmp(st, nb / 4);
    const int slabs = 2 * alignn_gemm_dgrad_wgrad_slabs(m);
    float* part = src != nullptr ? c.alloc((size_t)slabs * 2 * Kin) : nullptr;
    L(alignn_gemm_dgrad_wgrad_f16x3(GM, H, gm_amax, t.y.p, Kin, t.y.amax, p.weg_img_t, p.weg_amax, gy, H, g_y, Kin,
                                    src != nullptr ? src->xn : nullptr, Kin, src != nullptr ? src->stat : nullptr, part, p.g_weg, Kin, m,
                                    ws, nb, st));
    if (src != nullptr) bn_bwd_finalize_folded(c, part, slabs, Kin, src->red, st);
    if (pre_red) *pre_red = src != nullptr;
    return true;
}

// ops.MLPLayerFn.backward
Grad mlp_bwd_ln(Ctx& c, const MlpTape& t, const Grad& gy, bool need_gx);
Grad mlp_bwd(Ctx& c, const MlpTape& t, const Grad& gy, bool need_gx) {
    if (c.d->norm == 1) return mlp_bwd_ln(c, t, gy, need_gx);
    const alignn_mlp_params& p = *t.p;
    const int F = p.out, K = p.in;
    const int64_t rows = t.rows;
    hipStream_t st = t.lane ? c.T : c.main;
    if (t.lane != gy.on_T) c.sync(st, gy.on_T ? c.T : c.main);
    float* gpre = c.alloc((size_t)rows * F);
    float* g_amax = c.track(rows) ? c.new_amax() : nullptr;
    const int slabs = alignn_col_stats_slabs(rows);
    if (!gy.pre_red) {
        float* part = c.alloc((size_t)slabs * 2 * F);
        L(alignn_bn_silu_bwd_reduce(gy.p, F, t.pre, F, t.stat, rows, F, part, st));
        L(alignn_bn_bwd_finalize(part, slabs, F, p.red, st));
    }
    float* gb_part = nullptr;
    if (F <= 1024) {  // the pass that writes gpre also sums its columns (the Linear's bias gradient)
        gb_part = c.alloc((size_t)slabs * F);
        L(alignn_bn_silu_bwd_apply_sum(gy.p, F, t.pre, F, t.stat, p.red, 0, gpre, F, rows, F, g_amax, gb_part, st));
    } else
        L(alignn_bn_silu_bwd_apply(gy.p, F, t.pre, F, t.stat, p.gamma, p.red, 0, gpre, F, rows, F, g_amax, st));
    Grad gx;
    if (need_gx) {
        gx.p = c.alloc((size_t)rows * K);
        gx.on_T = t.lane;
        const Act* src = (t.x.xn != nullptr && g_amax != nullptr) ? &t.x : nullptr;
        dgrad(c, gpre, F, g_amax, p.W, F, K, p.img_t, p.w_amax, nullptr, 0, gx.p, rows, st, src, &gx.pre_red);
    }
    hipStream_t sd = side_for(c, rows);
    if (sd != st) {
        c.sync(sd, c.main);
        if (t.lane) c.sync(sd, c.T);
    }
    c.tmp_reset(sd);
    gemm_tn(c, gpre, F, g_amax, t.x.p, K, t.x.amax, p.gW, rows, F, K, sd);
    if (gb_part != nullptr)
        L(alignn_slab_sum(gb_part, slabs, F, p.gb, sd));
    else
        col_sum(c, gpre, F, rows, F, p.gb, sd);
    return gx;
}

// ops.EdgeGatedConvFn.backward: -> gradients w.r.t. the node and edge inputs
void conv_bwd_ln(Ctx& c, const ConvTape& t, const Grad& gx_out, const Grad* gy_out, Grad& g_x, Grad& g_y, bool need_gx);
void conv_bwd(Ctx& c, const ConvTape& t, const Grad& gx_out, const Grad* gy_out, Grad& g_x, Grad& g_y, bool need_gx = true) {
    if (c.d->norm == 1) return conv_bwd_ln(c, t, gx_out, gy_out, g_x, g_y, need_gx);
    const alignn_conv_params& p = *t.p;
    const alignn_graph_csr& g = *t.g;
    const int H = c.d->H, Kin = H;
    const int64_t n = g.n, m = g.m;
    hipStream_t main = c.main, T = t.lane ? c.T : c.main;
    if (gx_out.on_T) c.sync(main, c.T);
    if (!t.lane && gy_out != nullptr && gy_out->on_T) c.sync(main, c.T);
    float* GP = c.alloc((size_t)n * 4 * H);
    float* gp_amax = c.track(n) ? c.new_amax() : nullptr;
    float* gm_amax = c.track(m) ? c.new_amax() : nullptr;
    float* g_xpre = GP + 3 * (size_t)H;
    // ---- node branch: BatchNorm / SiLU backward -> g_xpre (the Ux block of GP), the quotient's adjoints
    const int n_slabs = alignn_col_stats_slabs(n);
    float* n_part = c.alloc((size_t)n_slabs * 2 * H);
    L(alignn_bn_silu_bwd_reduce(gx_out.p, H, t.xpre, H, t.n_stat, n, H, n_part, main));
    L(alignn_bn_bwd_finalize(n_part, n_slabs, H, p.n_red, main));
    float* gs1 = c.alloc((size_t)n * H);
    float* gs0 = c.alloc((size_t)n * H);
    L(alignn_bn_silu_bwd_apply_node(gx_out.p, H, t.xpre, H, t.n_stat, p.n_gamma, p.n_red, 0, g_xpre, 4 * H, n, H, gp_amax, t.s0,
                                    t.hh, gs1, gs0, main));
    // ---- edge branch (lane T for the line graph)
    const float* gy = gy_out != nullptr ? gy_out->p : nullptr;
    if (t.lane && gy_out != nullptr && !gy_out->on_T) c.sync(T, main);
    const float* e_red = nullptr;
    if (gy != nullptr) {
        if (!gy_out->pre_red) {
            const int e_slabs = alignn_col_stats_slabs(m);
            float* e_part = c.alloc((size_t)e_slabs * 2 * H);
            L(alignn_bn_silu_bwd_reduce(gy, H, t.M, H, t.e_stat, m, H, e_part, T));
            L(alignn_bn_bwd_finalize(e_part, e_slabs, H, p.e_red, T));
        }
        e_red = p.e_red;
    }
    if (t.lane) c.sync(T, main);  // gs1, gs0, the Ux block of GP and its maximum are ready
    float* GM = c.alloc((size_t)m * H);
    const bool lg_blocks = g.grp_seg_ptr != nullptr;
    const bool dense = lg_blocks && g.dense_max_src > 0 && alignn_egc_bwd_lg_dense_supported(g.dense_max_src);
    const int gslabs = lg_blocks ? (int)g.n_groups : alignn_egc_slabs(n);
    float* gb_part = c.alloc((size_t)gslabs * H);
    if (dense)
        L(alignn_egc_bwd_lg_dense(gy, t.M, t.P, gs1, gs0, t.e_stat, e_red, 0, m, g.grp_seg_ptr, g.grp_src_ptr, g.n_groups,
                                  g.dense_max_src, g.seg_ptr, g.seg_node, H, GM, GP, gb_part, gm_amax, gp_amax, T));
    else if (lg_blocks)
        L(alignn_egc_bwd_lg_fused(gy, t.M, t.P, gs1, gs0, t.e_stat, e_red, 0, m, g.grp_seg_ptr, g.grp_src_ptr, g.n_groups,
                                  g.seg_ptr, g.seg_node, g.dst, g.out_ptr, g.out_slot, H, GM, GP, gb_part, gm_amax, gp_amax, T));
    else {
        L(alignn_egc_bwd_dst(gy, t.M, t.P, gs1, gs0, t.e_stat, p.e_gamma, e_red, 0, m, g.seg_ptr, g.seg_node, g.src, n, H, GM, GP,
                             gb_part, gm_amax, gp_amax, T));
        L(alignn_egc_bwd_src(GM, t.M, gs1, g.out_ptr, g.out_slot, g.dst, n, H, GP, gp_amax, T));
    }
    if (t.lane) c.sync(main, T);  // GP complete
    // ---- input gradients (critical path): g_x = GP wcat (+ gx_out) beside g_y = GM w_eg (+ gy_out)
    g_x.p = c.alloc((size_t)n * Kin);
    g_y.p = edge_grad_buffer(c, t, m, Kin);
    hipStream_t sx = main;
    if (!t.lane && c.aux != main) {  // bond graph: the (shorter) node product on the aux stream beside the edge product
        c.sync(c.aux, main);
        sx = c.aux;
    }
    dgrad(c, GP, 4 * H, gp_amax, p.wcat, 4 * H, Kin, p.wcat_img_t, p.wcat_amax, gx_out.p, H, g_x.p, n, sx, nullptr, nullptr);
    const Act* src = (t.y.xn != nullptr && gm_amax != nullptr) ? &t.y : nullptr;
    const bool dw = dgrad_wgrad(c, t, GM, gm_amax, gy, g_y.p, m, H, Kin, T, src, &g_y.pre_red);
    if (!dw) dgrad(c, GM, H, gm_amax, p.w_eg, H, Kin, p.weg_img_t, p.weg_amax, gy, H, g_y.p, m, T, src, &g_y.pre_red);
    g_y.on_T = t.lane;
    g_x.on_T = false;
    if (sx != main) c.sync(main, sx);
    // ---- weight / bias gradients
    hipStream_t sd = side_for(c, m);
    if (sd != main) {
        c.sync(sd, main);
        if (t.lane) c.sync(sd, c.T);
    } else if (t.lane)
        c.sync(main, c.T);
    c.tmp_reset(sd);
    L(alignn_slab_sum(gb_part, gslabs, H, p.g_beg, sd));
    if (!dw) gemm_tn(c, GM, H, gm_amax, t.y.p, Kin, t.y.amax, p.g_weg, m, H, Kin, sd);
    gemm_tn(c, GP, 4 * H, gp_amax, t.x.p, Kin, t.x.amax, p.g_wcat, n, 4 * H, Kin, sd);
    col_sum(c, GP, 4 * H, n, 4 * H, p.g_bcat, sd);
}

// ops.MLPLayerFn.backward, norm == "layer"
Grad mlp_bwd_ln(Ctx& c, const MlpTape& t, const Grad& gy, bool need_gx) {
    const alignn_mlp_params& p = *t.p;
    const int F = p.out, K = p.in;
    const int64_t rows = t.rows;
    hipStream_t st = t.lane ? c.T : c.main;
    if (t.lane != gy.on_T) c.sync(st, gy.on_T ? c.T : c.main);
    float* gpre = c.alloc((size_t)rows * F);
    float* g_amax = c.track(rows) ? c.new_amax() : nullptr;
    const int slabs = alignn_ln_slabs(rows);
    float* part = c.alloc((size_t)slabs * 2 * F);
    L(alignn_ln_silu_bwd(gy.p, F, t.pre, F, p.gamma, p.beta, t.stat, gpre, F, part, rows, F, g_amax, st));
    if (c.param_grads) L(alignn_bn_bwd_finalize(part, slabs, F, p.red, st));
    Grad gx;
    if (need_gx) {
        gx.p = c.alloc((size_t)rows * K);
        gx.on_T = t.lane;
        dgrad(c, gpre, F, g_amax, p.W, F, K, p.img_t, p.w_amax, nullptr, 0, gx.p, rows, st, nullptr, nullptr);
    }
    if (!c.param_grads) return gx;
    hipStream_t sd = side_for(c, rows);
    if (sd != st) {
        c.sync(sd, c.main);
        if (t.lane) c.sync(sd, c.T);
    }
    c.tmp_reset(sd);
    gemm_tn(c, gpre, F, g_amax, t.x.p, K, t.x.amax, p.gW, rows, F, K, sd);
    col_sum(c, gpre, F, rows, F, p.gb, sd);
    return gx;
}

// ops.EdgeGatedConvFn.backward, norm == "layer".  need_gx == false: nobody reads the gradient of the node input (the first
// convolution of the force evaluation: the atom features are data)
void conv_bwd_ln(Ctx& c, const ConvTape& t, const Grad& gx_out, const Grad* gy_out, Grad& g_x, Grad& g_y, bool need_gx) {
    const alignn_conv_params& p = *t.p;
    const alignn_graph_csr& g = *t.g;
    const int H = c.d->H, Kin = H;
    const int64_t n = g.n, m = g.m;
    hipStream_t main = c.main, T = t.lane ? c.T : c.main;
    if (gx_out.on_T) c.sync(main, c.T);
    if (!t.lane && gy_out != nullptr && gy_out->on_T) c.sync(main, c.T);
    float* GP = c.alloc((size_t)n * 4 * H);
    float* gp_amax = c.track(n) ? c.new_amax() : nullptr;
    float* gm_amax = c.track(m) ? c.new_amax() : nullptr;
    float* g_xpre = GP + 3 * (size_t)H;
    // ---- node branch: LayerNorm / SiLU backward straight into the Ux block of GP, the quotient's adjoints
    const int n_slabs = alignn_ln_slabs(n);
    float* n_part = c.alloc((size_t)n_slabs * 2 * H);
    float* gs1 = c.alloc((size_t)n * H);
    float* gs0 = c.alloc((size_t)n * H);
    L(alignn_ln_silu_bwd_node(gx_out.p, H, t.xpre, H, p.n_gamma, p.n_beta, t.n_stat, g_xpre, 4 * H, n_part, n, H, gp_amax, t.s0, t.hh,
                              gs1, gs0, main));  // (+ the quotient's adjoints: alignn_egc_node_bwd in the same pass)
    if (c.param_grads) L(alignn_bn_bwd_finalize(n_part, n_slabs, H, p.n_red, main));
    // ---- edge branch (lane T for the line graph): the finished normalised-branch gradient, handed over as it is
    const float* gy = gy_out != nullptr ? gy_out->p : nullptr;
    if (t.lane && gy_out != nullptr && !gy_out->on_T) c.sync(T, main);
    const bool lg_blocks = g.grp_seg_ptr != nullptr;
    const bool dense = lg_blocks && g.dense_max_src > 0 && alignn_egc_bwd_lg_dense_supported(g.dense_max_src);
    const bool ln_inside = gy != nullptr && dense && alignn_egc_ln_fused_supported(H, m);  // (csrc/convln.hip)
    const bool ln_dst = gy != nullptr && !lg_blocks && alignn_egc_ln_dst_supported(H);       // (the bond graph: same file)
    float* g_branch = nullptr;
    if (gy != nullptr && !ln_inside && !ln_dst) {
        g_branch = c.alloc((size_t)m * H);
        const int e_slabs = alignn_ln_slabs(m);
        float* e_part = c.alloc((size_t)e_slabs * 2 * H);
        L(alignn_ln_silu_bwd(gy, H, t.M, H, p.e_gamma, p.e_beta, t.e_stat, g_branch, H, e_part, m, H, nullptr, T));
        if (c.param_grads) L(alignn_bn_bwd_finalize(e_part, e_slabs, H, p.e_red, T));
    }
    if (t.lane) c.sync(T, main);
    float* GM = c.alloc((size_t)m * H);
    const int gslabs = lg_blocks ? (int)g.n_groups : (ln_dst ? alignn_egc_ln_dst_slabs(n) : alignn_egc_slabs(n));
    float* gb_part = c.alloc((size_t)gslabs * H);
    if (ln_dst) {
        float* e_part = c.alloc((size_t)gslabs * 2 * H);
        L(alignn_egc_bwd_dst_ln(gy, t.M, t.P, gs1, gs0, p.e_gamma, p.e_beta, t.e_stat, g.seg_ptr, g.seg_node, g.src, n, H, GM, GP, gb_part,
                                e_part, gm_amax, gp_amax, T));
        if (c.param_grads) L(alignn_bn_bwd_finalize(e_part, gslabs, H, p.e_red, T));
        L(alignn_egc_bwd_src(GM, t.M, gs1, g.out_ptr, g.out_slot, g.dst, n, H, GP, gp_amax, T));
    } else if (ln_inside) {
        float* e_part = c.alloc((size_t)gslabs * 2 * H);
        L(alignn_egc_bwd_lg_dense_ln(gy, t.M, t.P, gs1, gs0, p.e_gamma, p.e_beta, t.e_stat, m, g.grp_seg_ptr, g.grp_src_ptr, g.n_groups,
                                     g.dense_max_src, g.seg_ptr, g.seg_node, H, GM, GP, gb_part, e_part, gm_amax, gp_amax, T));
        if (c.param_grads) L(alignn_bn_bwd_finalize(e_part, gslabs, H, p.e_red, T));
    } else if (dense)
        L(alignn_egc_bwd_lg_dense(g_branch, t.M, t.P, gs1, gs0, nullptr, nullptr, 0, m, g.grp_seg_ptr, g.grp_src_ptr, g.n_groups,
                                  g.dense_max_src, g.seg_ptr, g.seg_node, H, GM, GP, gb_part, gm_amax, gp_amax, T));
    else if (lg_blocks)
        L(alignn_egc_bwd_lg_fused(g_branch, t.M, t.P, gs1, gs0, nullptr, nullptr, 0, m, g.grp_seg_ptr, g.grp_src_ptr, g.n_groups,
                                  g.seg_ptr, g.seg_node, g.dst, g.out_ptr, g.out_slot, H, GM, GP, gb_part, gm_amax, gp_amax, T));
    else {
        L(alignn_egc_bwd_dst(g_branch, t.M, t.P, gs1, gs0, nullptr, p.e_gamma, nullptr, 0, m, g.seg_ptr, g.seg_node, g.src, n, H, GM,
                             GP, gb_part, gm_amax, gp_amax, T));
        L(alignn_egc_bwd_src(GM, t.M, gs1, g.out_ptr, g.out_slot, g.dst, n, H, GP, gp_amax, T));
    }
    if (t.lane) c.sync(main, T);
    // ---- input gradients
    g_y.p = edge_grad_buffer(c, t, m, Kin);
    hipStream_t sx = main;
    if (need_gx) {
        g_x.p = c.alloc((size_t)n * Kin);
        if (!t.lane && c.aux != main) {
            c.sync(c.aux, main);
            sx = c.aux;
        }
        dgrad(c, GP, 4 * H, gp_amax, p.wcat, 4 * H, Kin, p.wcat_img_t, p.wcat_amax, gx_out.p, H, g_x.p, n, sx, nullptr, nullptr);
    }
    const bool dw = dgrad_wgrad(c, t, GM, gm_amax, gy, g_y.p, m, H, Kin, T, nullptr, nullptr);
    if (!dw) dgrad(c, GM, H, gm_amax, p.w_eg, H, Kin, p.weg_img_t, p.weg_amax, gy, H, g_y.p, m, T, nullptr, nullptr);
    g_y.on_T = t.lane;
    g_x.on_T = false;
    if (sx != main) c.sync(main, sx);
    if (!c.param_grads) return;
    // ---- weight / bias gradients
    hipStream_t sd = side_for(c, m);
    if (sd != main) {
        c.sync(sd, main);
        if (t.lane) c.sync(sd, c.T);
    } else if (t.lane)
        c.sync(main, c.T);
    c.tmp_reset(sd);
    L(alignn_slab_sum(gb_part, gslabs, H, p.g_beg, sd));
    if (!dw) gemm_tn(c, GM, H, gm_amax, t.y.p, Kin, t.y.amax, p.g_weg, m, H, Kin, sd);
    gemm_tn(c, GP, 4 * H, gp_amax, t.x.p, Kin, t.x.amax, p.g_wcat, n, 4 * H, Kin, sd);
    col_sum(c, GP, 4 * H, n, 4 * H, p.g_bcat, sd);
}

void run_backward(Ctx& c, const Tape& tp, const float* g_out) {
    const alignn_model_desc& d = *c.d;
    const alignn_model_batch& b = *c.b;
    const int64_t N = b.g.n;
    const int H = d.H, OF = d.out_features, B = b.B;
    c.amax_arena = c.alloc(kAmaxSlots);
    c.amax_next = 0;
    fill(c, c.amax_arena, kAmaxSlots, 0.0f, c.main);
    c.sync(c.T, c.main);
    c.sync(c.side, c.main);
    // ---- fc (ops.LinearFn.backward) and the pooling
    float* g_pool = c.alloc((size_t)B * H);
    dgrad(c, g_out, OF, nullptr, d.fc_W, OF, H, nullptr, nullptr, nullptr, 0, g_pool, B, c.main, nullptr, nullptr);
    if (c.param_grads) {
        c.tmp_reset(c.main);
        gemm_tn(c, g_out, OF, nullptr, tp.pool, H, nullptr, d.g_fc_W, B, OF, H, c.main);
        if ((OF % 4) == 0)
            col_sum(c, g_out, OF, B, OF, d.g_fc_b, c.main);
        else {  // e.g. the 1-wide readout: gb[n] = (gy^T ones)[n]
            float* ones = c.alloc((size_t)B);
            fill(c, ones, B, 1.0f, c.main);
            gemm_tn(c, g_out, OF, nullptr, ones, 1, nullptr, d.g_fc_b, B, OF, 1, c.main);
        }
    }
    Grad gx;
    gx.p = c.alloc((size_t)N * H);
    L(alignn_segment_mean_bwd(g_pool, b.graph_ptr, gx.p, B, H, c.main));
    // ---- GCN layers, then ALIGNN layers, in reverse
    int k = conv_count(d) - 1;
    Grad gy, gz;
    bool have_gy = false, have_gz = false;
    for (int i = d.gcn_layers - 1; i >= 0; --i, --k) {
        Grad nx, ny;
        conv_bwd(c, tp.convs[k], gx, have_gy ? &gy : nullptr, nx, ny);
        gx = nx;
        gy = ny;
        have_gy = true;
        if (c.unsupported) return;
    }
    // the cosines carry a gradient w.r.t. the bond vectors only when the forward recomputed them from r
    const bool angle_geom = c.geom && c.ff != nullptr && c.ff->lg_on_fly;
    for (int i = d.alignn_layers - 1; i >= 0; --i) {
        // line-graph convolution: node output = the bond features (gradient gy), edge output = the triplet features (gz)
        Grad gm, nz;
        conv_bwd(c, tp.convs[k], gy, have_gz ? &gz : nullptr, gm, nz);
        --k;
        gz = nz;
        have_gz = true;
        if (i == 0) {  // the angle embedding's backward (lane T) can start now, under the last bond-graph backward
            if (tp.angle_fused) {
                hipStream_t st = tp.angle_lane ? c.T : c.main;
                if (tp.angle_lane != gz.on_T) c.sync(st, gz.on_T ? c.T : c.main);
                alignn_angle_args a = angle_args(c, tp);
                a.gz = gz.p;
                a.workspace_bytes = alignn_angle_embed_workspace(a.rows, d.angle_bins, 1);
                a.workspace = c.alloc(a.workspace_bytes / sizeof(float));
                L(alignn_angle_embed_bwd(&a, st));
            } else if (c.param_grads || angle_geom) {
                Grad ga = mlp_bwd(c, tp.a2, gz, true);
                Grad gr = mlp_bwd(c, tp.a1, ga, angle_geom);
                c.g_rbf_a = gr.p;
                c.g_rbf_a_on_T = gr.on_T;
            }
        }
        Grad nx, ny;
        // (the atom features are data: without parameter gradients nobody reads the node input gradient of the first convolution)
        conv_bwd(c, tp.convs[k], gx, &gm, nx, ny, i > 0 || c.param_grads);
        --k;
        gx = nx;
        gy = ny;
        if (c.unsupported) return;
    }
    // ---- embeddings
    if (c.param_grads) mlp_bwd(c, tp.atom, gx, false);
    Grad ge = mlp_bwd(c, tp.e2, gy, true);
    Grad gre = mlp_bwd(c, tp.e1, ge, c.geom);
    c.g_rbf_e = gre.p;
    c.sync(c.main, c.T);
    c.sync(c.main, c.side);
}

// ---------------------------------------------------------------------------------------------------------------------
// inference: ALIGNN.forward in eval mode without autograd (alignn/pretrained.py, model.eval() under no_grad) - BatchNorm is
// the affine map of its running statistics, the edge output comes straight out of the gate pass (alignn_egc_gate_infer), m
// is never written and nothing is kept: ops.MLPLayerFn._fwd (eval) / ops.edge_gated_conv_infer launch for launch.
// ---------------------------------------------------------------------------------------------------------------------
Act mlp_infer(Ctx& c, const alignn_mlp_params& p, const Act& x, int64_t rows) {
    const int F = p.out, K = p.in;
    const bool lane = c.T != c.main && rows >= c.d->lane_min_rows;
    hipStream_t st = lane ? c.T : c.main;
    if (lane != x.on_T) c.sync(st, x.on_T ? c.T : c.main);
    float* pre = c.alloc((size_t)rows * F);
    float* stat = c.alloc((size_t)4 * F);
    project(c, x.p, K, x.amax, p.W, K, p.img, p.w_amax, p.b, pre, F, rows, F, K, st);
    L(alignn_bn_finalize(nullptr, 0, rows, F, p.gamma, p.beta, c.d->eps, c.d->momentum, p.rm, p.rv, stat, st));
    Act y;
    y.p = c.alloc((size_t)rows * F);
    y.amax = c.track(rows) ? c.new_amax() : nullptr;
    L(alignn_bn_silu_fwd(pre, F, nullptr, 0, stat, y.p, F, rows, F, y.amax, st));
    y.on_T = lane;
    return y;
}

void conv_infer(Ctx& c, const alignn_conv_params& p, const alignn_graph_csr& g, const Act& x, const Act& y, bool need_y,
                Act& x_out, Act& y_out) {
    const int H = c.d->H, Kin = H;
    const int64_t n = g.n, m = g.m;
    const bool lane = c.T != c.main && m >= c.d->lane_min_rows;
    hipStream_t main = c.main, T = lane ? c.T : c.main;
    if (x.on_T) c.sync(main, c.T);
    if (!lane && y.on_T) c.sync(main, c.T);
    float* P = c.alloc((size_t)n * 4 * H);
    project(c, x.p, Kin, x.amax, p.wcat, Kin, p.wcat_img, p.wcat_amax, p.bcat, P, 4 * H, n, 4 * H, Kin, main);
    float* stats = c.alloc((size_t)8 * H);
    L(alignn_bn_finalize(nullptr, 0, n, H, p.n_gamma, p.n_beta, c.d->eps, c.d->momentum, p.n_rm, p.n_rv, stats, main));
    L(alignn_bn_finalize(nullptr, 0, m, H, p.e_gamma, p.e_beta, c.d->eps, c.d->momentum, p.e_rm, p.e_rv, stats + 4 * H, main));
    if (lane) c.sync(T, main);
    float* Cm = c.alloc((size_t)m * H);
    project(c, y.p, Kin, y.amax, p.w_eg, Kin, p.weg_img, p.weg_amax, p.b_eg, Cm, H, m, H, Kin, T);
    float* xpre = c.alloc((size_t)n * H);
    Act yo;
    if (need_y) {
        yo.p = c.alloc((size_t)m * H);
        yo.amax = c.track(m) ? c.new_amax() : nullptr;
        yo.on_T = lane;
    }
    L(alignn_egc_gate_infer(P, Cm, g.seg_ptr, g.seg_node, g.src, n, m, H, xpre, stats + 4 * H, y.p, yo.p, yo.amax, T));
    if (lane) c.sync(main, T);
    x_out = Act{};
    x_out.p = c.alloc((size_t)n * H);
    x_out.amax = c.track(n) ? c.new_amax() : nullptr;
    L(alignn_bn_silu_fwd(xpre, H, x.p, Kin, stats, x_out.p, H, n, H, x_out.amax, main));
    y_out = yo;
}

void run_infer(Ctx& c, float* out) {
    const alignn_model_desc& d = *c.d;
    const alignn_model_batch& b = *c.b;
    const int64_t N = b.g.n, E = b.g.m, Tn = b.lg.m;
    c.amax_arena = c.alloc(kAmaxSlots);
    c.amax_next = 0;
    fill(c, c.amax_arena, kAmaxSlots, 0.0f, c.main);
    if (d.n_weights > 0) L(alignn_prepare_weights(d.weight_descs, d.n_weights, d.weight_amax, c.main));
    c.sync(c.T, c.main);
    Act z;
    if (d.angle_fused != 0 && Tn > 0 && alignn_angle_embed_supported(d.angle_bins, d.angle1.out, d.angle2.out) != 0) {
        const bool lane = c.T != c.main && Tn >= d.lane_min_rows;
        Tape tp;
        tp.hcos = b.h;
        tp.a_stat1 = c.alloc((size_t)4 * d.angle1.out);
        tp.a_stat2 = c.alloc((size_t)4 * d.angle2.out);
        tp.a_scal = c.alloc((size_t)alignn_angle_embed_scal_floats());
        alignn_angle_args a = angle_args(c, tp);
        z.p = c.alloc((size_t)Tn * d.H);
        z.amax = c.track(Tn) ? c.new_amax() : nullptr;
        z.on_T = lane;
        a.z = z.p;
        a.z_amax = z.amax;
        L(alignn_angle_embed_infer(&a, lane ? c.T : c.main));
    } else {
        float* rbf_a = c.alloc((size_t)Tn * d.angle_bins);
        L(alignn_rbf_fwd(b.h, d.angle_centers, d.angle_gamma, rbf_a, Tn, d.angle_bins, c.main));
        Act za;
        za.p = rbf_a;
        z = mlp_infer(c, d.angle2, mlp_infer(c, d.angle1, za, Tn), Tn);
    }
    Act xa;
    xa.p = const_cast<float*>(b.atom_features);
    Act x = mlp_infer(c, d.atom, xa, N);
    float* bl = c.alloc((size_t)E);
    L(alignn_norm3_fwd(b.r, bl, E, c.main));
    float* rbf_e = c.alloc((size_t)E * d.edge_bins);
    L(alignn_rbf_fwd(bl, d.edge_centers, d.edge_gamma, rbf_e, E, d.edge_bins, c.main));
    Act ye;
    ye.p = rbf_e;
    Act y = mlp_infer(c, d.edge2, mlp_infer(c, d.edge1, ye, E), E);
    if (c.unsupported) return;
    int k = 0;
    for (int i = 0; i < d.alignn_layers; ++i) {
        Act xo, m_, yo, zo;
        conv_infer(c, d.convs[k++], b.g, x, y, true, xo, m_);
        conv_infer(c, d.convs[k++], b.lg, m_, z, i + 1 < d.alignn_layers, yo, zo);
        x = xo, y = yo, z = zo;
        if (c.unsupported) return;
    }
    for (int i = 0; i < d.gcn_layers; ++i) {
        Act xo, yo;
        conv_infer(c, d.convs[k++], b.g, x, y, i + 1 < d.gcn_layers, xo, yo);
        x = xo, y = yo;
        if (c.unsupported) return;
    }
    if (x.on_T) c.sync(c.main, c.T);
    float* pool = c.alloc((size_t)b.B * d.H);
    L(alignn_segment_mean_fwd(x.p, b.graph_ptr, pool, b.B, d.H, c.main));
    L(alignn_gemm_nt(pool, d.H, d.fc_W, d.H, d.fc_b, nullptr, 0, out, d.out_features, b.B, d.out_features, d.H, c.main));
    c.sync(c.main, c.T);
}

// ---------------------------------------------------------------------------------------------------------------------
// ALIGNNAtomWise with the force / stress head (alignn/models/alignn_atomwise.py:364-660, calculate_gradient=True)
// ---------------------------------------------------------------------------------------------------------------------

// alignn_ff_eval: energies, forces, stresses as values = the LayerNorm forward with the bond vectors as a leaf + its reverse
// w.r.t. them (ALIGNNAtomWise._forward_fused(b, True) under ops.no_param_grad: what MD runs and what ForcesFn.forward runs)
void run_ff_eval(Ctx& c, Tape& tp, float* out, float* forces, float* stress) {
    const alignn_model_desc& d = *c.d;
    const alignn_model_batch& b = *c.b;
    const alignn_ff_desc& f = *c.ff;
    const int64_t N = b.g.n, E = b.g.m, Tn = b.lg.m;
    const int B = b.B;
    c.param_grads = false;
    c.geom = true;
    tp.pred = c.alloc((size_t)B);
    tp.seed = c.alloc((size_t)B);
    run_forward(c, tp, tp.pred);
    if (c.unsupported) return;
    // total energy per crystal (+ the short-bond penalty) and d(sum en_out)/d pred, alignn_atomwise.py:494-510
    L(alignn_ff_energy(tp.pred, tp.bl, b.graph_ptr, B, E, f.energy_mult_natoms, f.use_penalty, f.penalty_factor, f.penalty_threshold,
                       out, tp.seed, c.main));
    run_backward(c, tp, tp.seed);
    if (c.unsupported) return;
    // ---- geometry: d/dr of the RBF inputs (bond length; bond cosines when they were recomputed from r)
    float* g_bl = c.alloc((size_t)E);
    L(alignn_rbf_bwd(tp.bl, d.edge_centers, d.edge_gamma, c.g_rbf_e, g_bl, E, d.edge_bins, c.main));
    if (f.use_penalty) L(alignn_ff_penalty_bwd(tp.bl, g_bl, E, B, f.penalty_factor, f.penalty_threshold, c.main));
    float* gr_bl = c.alloc((size_t)3 * E);
    L(alignn_norm3_bwd(b.r, g_bl, gr_bl, E, c.main));
    tp.g_r = gr_bl;
    if (f.lg_on_fly) {
        // (run_backward joined lane T into the caller's stream before it returned)
        float* g_h = c.alloc((size_t)Tn);
        L(alignn_rbf_bwd(tp.hcos, d.angle_centers, d.angle_gamma, c.g_rbf_a, g_h, Tn, d.angle_bins, c.main));
        float* ga = c.alloc((size_t)3 * Tn);
        float* gb = c.alloc((size_t)3 * Tn);
        L(alignn_bond_cosine_bwd(b.r, b.lg.src, b.lg.dst, g_h, ga, gb, Tn, c.main));
        // dh/dr[e]: the triplets where e is the first bond (by source of L(g)) + those where it is the second (by destination)
        float* gra = c.alloc((size_t)3 * E);
        float* grb = c.alloc((size_t)3 * E);
        L(alignn_segment_sum(ga, 3, b.lg.out_ptr, b.lg.out_slot, nullptr, gra, 3, E, 3, c.main));
        L(alignn_segment_sum(gb, 3, b.lg.seg_ptr, nullptr, b.lg.seg_node, grb, 3, E, 3, c.main));
        tp.g_r = c.alloc((size_t)3 * E);
        L(alignn_add3(gra, grb, gr_bl, tp.g_r, 3 * E, c.main));
    }
    // ---- pair forces -> forces per atom, virial stress per crystal (:530-638)
    const float scale = f.grad_multiplier * (f.force_mult_natoms ? (float)N : 1.0f);
    L(alignn_pair_force_reduce(tp.g_r, scale, b.g.seg_ptr, b.g.out_ptr, b.g.out_slot, f.add_reverse_forces, forces, N, c.main));
    if (stress != nullptr && f.has_stress)
        L(alignn_virial_stress(b.r, tp.g_r, scale, b.graph_ptr, b.g.seg_ptr, f.volume, f.stress_multiplier * -160.21766208f, stress, B,
                               c.main));
}

// ---- the second-order pass (alignn_amd/ff2.py dual_pass, REUSE_FORWARD: tangents only - the values are alignn_ff_eval's tape)
struct DAct {  // value p and tangent t of one activation + the max|.| scalars their producers tracked (or NULL)
    float *p = nullptr, *t = nullptr, *amax_p = nullptr, *amax_t = nullptr;
    bool on_T = false;  // the tangent was last written on lane T
};
struct DMlpTape {
    const alignn_mlp_params* p = nullptr;
    DAct x, pre;
    float* stats = nullptr;
    int64_t rows = 0;
};
struct DConvTape {
    const alignn_conv_params* p = nullptr;
    const alignn_graph_csr* g = nullptr;
    DAct x, y, P, M, xpre;
    float *s0 = nullptr, *hh = nullptr, *s0t = nullptr, *hht = nullptr, *n_stats = nullptr, *e_stats = nullptr;
    bool need_y = true, lane = false;
};

// ff2._ln_fwd with value_out: the tangent of y = res + silu(LayerNorm(x)); the value is `known`
DAct dual_ln_fwd(Ctx& c, const DAct& x, const DAct* res, const float* gamma, const float* beta, const Act& known, int64_t rows, int F,
                 float** stats_out, hipStream_t st) {
    DAct y;
    y.p = known.p;
    y.amax_p = known.amax;
    y.t = c.alloc((size_t)rows * F);
    float* stats = c.alloc((size_t)rows * 2);
    float* amax2 = c.track(rows) ? c.new_amax2() : nullptr;
    L(alignn_ln_silu_dual_fwd(x.p, x.t, F, res ? res->p : nullptr, res ? res->t : nullptr, res ? F : 0, gamma, beta, c.d->eps, nullptr,
                              y.t, F, stats, rows, F, amax2, st));
    y.amax_t = amax2 ? amax2 + 1 : nullptr;
    if (y.amax_p == nullptr && amax2 != nullptr) y.amax_p = amax2;  // (Dual.am(0) falls back to the pair's first scalar)
    *stats_out = stats;
    return y;
}

DAct dual_mlp_fwd(Ctx& c, DMlpTape& t, const alignn_mlp_params& p, const MlpTape& fwd, const DAct& x, hipStream_t st) {
    const int F = p.out, K = p.in;
    const int64_t rows = fwd.rows;
    t.p = &p;
    t.x = x;
    t.rows = rows;
    t.pre.p = fwd.pre;
    t.pre.t = c.alloc((size_t)rows * F);
    project(c, x.t, K, x.amax_t, p.W, K, p.img, p.w_amax, nullptr, t.pre.t, F, rows, F, K, st);  // the tangent of x W^T + b
    DAct y = dual_ln_fwd(c, t.pre, nullptr, p.gamma, p.beta, fwd.y, rows, F, &t.stats, st);
    y.on_T = st != c.main;
    return y;
}

void dual_conv_fwd(Ctx& c, DConvTape& t, const alignn_conv_params& p, const ConvTape& fwd, const DAct& x, const DAct& y, bool need_y,
                   DAct& x_out, DAct& y_out) {
    const alignn_graph_csr& g = *fwd.g;
    const int H = c.d->H, Kin = H;
    const int64_t n = g.n, m = g.m;
    t.p = &p;
    t.g = &g;
    t.x = x;
    t.y = y;
    t.need_y = need_y;
    // lane T for the line graph (as in conv_fwd_ln): the T-row tangent projection and the gate pass run there, beside the
    // bond-row kernels of the caller's stream - the next convolution's node side, the bond-graph convolution that follows
    t.lane = c.T != c.main && m >= c.d->lane_min_rows;
    hipStream_t main = c.main, T = t.lane ? c.T : c.main;
    if (x.on_T) c.sync(main, c.T);
    if (!t.lane && y.on_T) c.sync(main, c.T);
    t.P.p = fwd.P;
    t.P.t = c.alloc((size_t)n * 4 * H);
    project(c, x.t, Kin, x.amax_t, p.wcat, Kin, p.wcat_img, p.wcat_amax, nullptr, t.P.t, 4 * H, n, 4 * H, Kin, main);
    if (t.lane) c.sync(T, main);
    t.M.p = fwd.M;
    t.M.t = c.alloc((size_t)m * H);
    project(c, y.t, Kin, y.amax_t, p.w_eg, Kin, p.weg_img, p.weg_amax, nullptr, t.M.t, H, m, H, Kin, T);
    t.xpre.p = fwd.xpre;
    t.xpre.t = c.alloc((size_t)n * H);
    t.s0 = fwd.s0;
    t.hh = fwd.hh;
    t.s0t = c.alloc((size_t)n * H);
    t.hht = c.alloc((size_t)n * H);
    y_out = DAct{};
    if (need_y && fwd.e_stat != nullptr && alignn_egc_ln_fused_supported(H, m)) {
        // the tangent of the edge LayerNorm inside the gate pass (csrc/convln.hip); the row statistics are the evaluation's
        y_out.p = fwd.y_out.p;
        y_out.amax_p = fwd.y_out.amax;
        y_out.t = c.alloc((size_t)m * H);
        float* amax2 = c.track(m) ? c.new_amax2() : nullptr;
        L(alignn_egc_gate_dual_tan_ln(t.P.p, t.P.t, t.M.p, t.M.t, g.seg_ptr, g.seg_node, g.src, n, m, H, t.xpre.t, t.s0, t.hh, t.s0t,
                                      t.hht, p.e_gamma, p.e_beta, fwd.e_stat, y.t, y_out.t, amax2, T));
        y_out.amax_t = amax2 ? amax2 + 1 : nullptr;
        if (y_out.amax_p == nullptr && amax2 != nullptr) y_out.amax_p = amax2;
        y_out.on_T = t.lane;
        t.e_stats = fwd.e_stat;
        if (t.lane) c.sync(main, T);
        x_out = dual_ln_fwd(c, t.xpre, &x, p.n_gamma, p.n_beta, fwd.x_out, n, H, &t.n_stats, main);
        return;
    }
    L(alignn_egc_gate_dual_fwd_tangent(t.P.p, t.P.t, t.M.p, t.M.t, g.seg_ptr, g.seg_node, g.src, n, m, H, t.xpre.t, t.s0, t.hh, t.s0t,
                                       t.hht, T));
    if (t.lane) c.sync(main, T);
    x_out = dual_ln_fwd(c, t.xpre, &x, p.n_gamma, p.n_beta, fwd.x_out, n, H, &t.n_stats, main);
    if (need_y) {
        y_out = dual_ln_fwd(c, t.M, &y, p.e_gamma, p.e_beta, fwd.y_out, m, H, &t.e_stats, T);
        y_out.on_T = t.lane;
    }
}

// ff2._ln_bwd: (g, gt) -> (gx, gxt) into out_p / out_t (leading dimension ldo); dbeta | dgamma -> red
DAct dual_ln_bwd(Ctx& c, const DAct& g, const DAct& x, const float* gamma, const float* beta, const float* stats, float* out_p,
                 float* out_t, int64_t ldo, float* amax2, int64_t rows, int F, float* red, hipStream_t st) {
    const int slabs = alignn_dual_slabs(rows);
    float* partial = c.alloc((size_t)slabs * 2 * F);
    L(alignn_ln_silu_dual_bwd(g.p, g.t, F, x.p, x.t, F, gamma, beta, stats, out_p, out_t, ldo, partial, rows, F, amax2, st));
    L(alignn_bn_bwd_finalize(partial, slabs, F, red, st));
    DAct o;
    o.p = out_p;
    o.t = out_t;
    o.amax_p = amax2;
    o.amax_t = amax2 ? amax2 + 1 : nullptr;
    return o;
}

// W-bar = g^T x + gt^T xt: the value half into dW, the tangent half into its twin (added once at the end of the call)
void dual_wgrad(Ctx& c, const DAct& g, int64_t ldg, const DAct& x, int64_t ldx, float* dW, int64_t M, int N, int K, hipStream_t st) {
    gemm_tn(c, g.p, ldg, g.amax_p, x.p, ldx, x.amax_p, dW, M, N, K, st);
    gemm_tn(c, g.t, ldg, g.amax_t, x.t, ldx, x.amax_t, c.twin(dW), M, N, K, st);
}

DAct dual_dgrad(Ctx& c, const DAct& g, int64_t ldg, const float* w, int Nred, int Kout, const void* img_t, const float* w_amax,
                const DAct* addend, int64_t M, hipStream_t st) {
    DAct o;
    o.p = c.alloc((size_t)M * Kout);
    o.t = c.alloc((size_t)M * Kout);
    dgrad(c, g.p, ldg, g.amax_p, w, Nred, Kout, img_t, w_amax, addend ? addend->p : nullptr, Kout, o.p, M, st, nullptr, nullptr);
    dgrad(c, g.t, ldg, g.amax_t, w, Nred, Kout, img_t, w_amax, addend ? addend->t : nullptr, Kout, o.t, M, st, nullptr, nullptr);
    o.on_T = st != c.main;
    return o;
}

DAct dual_mlp_bwd(Ctx& c, const DMlpTape& t, const DAct& g, bool need_gx, hipStream_t st) {
    const alignn_mlp_params& p = *t.p;
    const int F = p.out, K = p.in;
    const int64_t rows = t.rows;
    if (st == c.main && g.on_T) c.sync(c.main, c.T);
    float* amax2 = c.track(rows) ? c.new_amax2() : nullptr;
    float* gp = c.alloc((size_t)rows * F);
    float* gt = c.alloc((size_t)rows * F);
    DAct gpre = dual_ln_bwd(c, g, t.pre, p.gamma, p.beta, t.stats, gp, gt, F, amax2, rows, F, p.red, st);
    DAct gx;
    if (need_gx) gx = dual_dgrad(c, gpre, F, p.W, F, K, p.img_t, p.w_amax, nullptr, rows, st);
    hipStream_t sd = side_for(c, rows);
    if (sd == c.main && st != c.main) sd = st;  // (no side stream for this size: stay where the operands are)
    if (sd != st) c.sync(sd, st);
    c.tmp_reset(sd);
    dual_wgrad(c, gpre, F, t.x, K, p.gW, rows, F, K, sd);
    col_sum(c, gpre.p, F, rows, F, p.gb, sd);
    return gx;
}

// ff2.conv_bwd: adjoints (gx, gy) of the outputs (gy.p == NULL: dead edge output) -> adjoints of the inputs
void dual_conv_bwd(Ctx& c, const DConvTape& t, const DAct& gx, const DAct& gy, DAct& g_x, DAct& g_y) {
    const alignn_conv_params& p = *t.p;
    const alignn_graph_csr& g = *t.g;
    const int H = c.d->H, Kin = H;
    const int64_t n = g.n, m = g.m;
    // lane T for the line graph (as in conv_bwd_ln): the gate reverse and the edge input gradients - every T-row kernel - run
    // there, the node branch, the node input gradient and whatever bond-row work follows on the caller's stream
    hipStream_t main = c.main, T = t.lane ? c.T : c.main;
    if (gx.on_T) c.sync(main, c.T);
    if (!t.lane && gy.p != nullptr && gy.on_T) c.sync(main, c.T);
    DAct GP, GM;
    GP.p = c.alloc((size_t)n * 4 * H);
    GP.t = c.alloc((size_t)n * 4 * H);
    GP.amax_p = c.track(n) ? c.new_amax2() : nullptr;
    GP.amax_t = GP.amax_p ? GP.amax_p + 1 : nullptr;
    // node branch: LayerNorm / SiLU reverse straight into the Ux blocks
    float* q1 = c.alloc((size_t)n * H);
    float* q0 = c.alloc((size_t)n * H);
    float* q1t = c.alloc((size_t)n * H);
    float* q0t = c.alloc((size_t)n * H);
    // (the slab sums of the LayerNorm parameter gradients feed nothing on the way - they are issued with the weight gradients, on
    // the side stream, instead of standing in the bond-row chain lane T waits for)
    float *n_partial = nullptr, *e_partial = nullptr;
    int n_slabs_fin = 0, e_slabs_fin = 0;
    {  // (ff2._ln_bwd + alignn_egc_node_dual_bwd in one pass)
        const int slabs_n = alignn_dual_slabs(n);
        float* partial = c.alloc((size_t)slabs_n * 2 * H);
        L(alignn_ln_silu_dual_bwd_node(gx.p, gx.t, H, t.xpre.p, t.xpre.t, H, p.n_gamma, p.n_beta, t.n_stats, GP.p + 3 * (size_t)H,
                                       GP.t + 3 * (size_t)H, 4 * H, partial, n, H, GP.amax_p, t.s0, t.hh, t.s0t, t.hht, q1, q0, q1t, q0t,
                                       main));
        n_partial = partial;
        n_slabs_fin = slabs_n;
    }
    if (t.lane) c.sync(T, main);  // (the four adjoint rows per node; gy if the caller's stream wrote it)
    const bool dense = c.ff->dense_lg_reverse && g.grp_seg_ptr != nullptr && g.dense_max_src > 0;
    const bool ln_inside = gy.p != nullptr && dense && alignn_egc_ln_fused_supported(H, m);  // (csrc/convln.hip)
    const bool ln_dst = gy.p != nullptr && !dense && alignn_egc_ln_dst_supported(H);           // (the bond graph: same file)
    DAct GL;
    if (gy.p != nullptr && !ln_inside && !ln_dst) {
        float* amax2 = c.track(m) ? c.new_amax2() : nullptr;
        float* lp = c.alloc((size_t)m * H);
        float* lt = c.alloc((size_t)m * H);
        GL = dual_ln_bwd(c, gy, t.M, p.e_gamma, p.e_beta, t.e_stats, lp, lt, H, amax2, m, H, p.e_red, T);
    }
    GM.p = c.alloc((size_t)m * H);
    GM.t = c.alloc((size_t)m * H);
    GM.amax_p = c.track(m) ? c.new_amax2() : nullptr;
    GM.amax_t = GM.amax_p ? GM.amax_p + 1 : nullptr;
    int slabs;
    float* gb_part;
    if (ln_inside) {
        slabs = (int)g.n_groups;
        gb_part = c.alloc((size_t)slabs * H);
        float* e_part = c.alloc((size_t)slabs * 2 * H);
        L(alignn_egc_dual_bwd_lg_dense_ln(gy.p, gy.t, t.M.p, t.M.t, t.P.p, t.P.t, q1, q0, q1t, q0t, p.e_gamma, p.e_beta, t.e_stats, m,
                                          g.grp_seg_ptr, g.grp_src_ptr, slabs, g.seg_ptr, g.seg_node, H, GM.p, GM.t, GP.p, GP.t, gb_part,
                                          e_part, GM.amax_p, GP.amax_p, T));
        e_partial = e_part;
        e_slabs_fin = slabs;
    } else if (dense) {  // line graph: destination- and source-ordered halves in one pass over the dense blocks
        slabs = (int)g.n_groups;
        gb_part = c.alloc((size_t)slabs * H);
        L(alignn_egc_dual_bwd_lg_dense(GL.p, GL.t, t.M.p, t.M.t, t.P.p, t.P.t, q1, q0, q1t, q0t, m, g.grp_seg_ptr, g.grp_src_ptr, slabs,
                                       g.seg_ptr, g.seg_node, H, GM.p, GM.t, GP.p, GP.t, gb_part, GM.amax_p, GP.amax_p, T));
    } else if (ln_dst) {
        slabs = alignn_egc_ln_dst_slabs(n);
        gb_part = c.alloc((size_t)slabs * H);
        float* e_part = c.alloc((size_t)slabs * 2 * H);
        L(alignn_egc_dual_bwd_dst_ln(gy.p, gy.t, t.M.p, t.M.t, t.P.p, t.P.t, q1, q0, q1t, q0t, p.e_gamma, p.e_beta, t.e_stats, g.seg_ptr,
                                     g.seg_node, g.src, n, H, GM.p, GM.t, GP.p, GP.t, gb_part, e_part, GM.amax_p, GP.amax_p, T));
        e_partial = e_part;
        e_slabs_fin = slabs;
        L(alignn_egc_dual_bwd_src(GM.p, GM.t, t.M.p, t.M.t, q1, q1t, g.out_ptr, g.out_slot, g.dst, n, H, GP.p, GP.t, GP.amax_p, T));
    } else {
        slabs = alignn_dual_slabs(n);
        gb_part = c.alloc((size_t)slabs * H);
        L(alignn_egc_dual_bwd_dst(GL.p, GL.t, t.M.p, t.M.t, t.P.p, t.P.t, q1, q0, q1t, q0t, g.seg_ptr, g.seg_node, g.src, n, H, GM.p, GM.t,
                                  GP.p, GP.t, gb_part, GM.amax_p, GP.amax_p, T));
        L(alignn_egc_dual_bwd_src(GM.p, GM.t, t.M.p, t.M.t, q1, q1t, g.out_ptr, g.out_slot, g.dst, n, H, GP.p, GP.t, GP.amax_p, T));
    }
    // weight gradients on the side stream: GP / GM are complete here
    hipStream_t sd = side_for(c, m);
    if (sd == main && t.lane) sd = T;
    if (sd != main) c.sync(sd, main);
    if (t.lane && sd != T) c.sync(sd, T);
    if (t.lane) c.sync(main, T);  // GP complete: the node input gradient reads it
    g_x = dual_dgrad(c, GP, 4 * H, p.wcat, 4 * H, Kin, p.wcat_img_t, p.wcat_amax, &gx, n, main);
    // value and tangent halves of the edge-gate projection's backward: input gradient + weight gradient in one pass over each
    // half's g_m where the shape allows (csrc/gemm_dw.hip; the tangent half's weight gradient goes to the twin buffer as before)
    bool dw = c.d->dw_fused > 0 && m >= c.d->dw_fused && GM.amax_p != nullptr && t.y.amax_p != nullptr && t.y.amax_t != nullptr &&
              p.weg_img_t != nullptr && Kin == H && alignn_gemm_dgrad_wgrad_supported(m, H, Kin) && x6_shape_ok(c, m, H, Kin, H);
    if (dw) {
        g_y.p = c.alloc((size_t)m * Kin);
        g_y.t = c.alloc((size_t)m * Kin);
        g_y.on_T = T != c.main;
        const size_t nb = alignn_gemm_dgrad_wgrad_workspace(m);
        c.tmp_reset(T);
        float* ws = c.tmp(T, nb / 4);
        L(alignn_gemm_dgrad_wgrad_f16x3(GM.p, H, GM.amax_p, t.y.p, Kin, t.y.amax_p, p.weg_img_t, p.weg_amax, gy.p, H, g_y.p, Kin,
                                        nullptr, 0, nullptr, nullptr, p.g_weg, Kin, m, ws, nb, T));
        L(alignn_gemm_dgrad_wgrad_f16x3(GM.t, H, GM.amax_t, t.y.t, Kin, t.y.amax_t, p.weg_img_t, p.weg_amax, gy.t, H, g_y.t, Kin,
                                        nullptr, 0, nullptr, nullptr, c.twin(p.g_weg), Kin, m, ws, nb, T));
    } else
        g_y = dual_dgrad(c, GM, H, p.w_eg, H, Kin, p.weg_img_t, p.weg_amax, gy.p != nullptr ? &gy : nullptr, m, T);
    c.tmp_reset(sd);
    if (n_partial != nullptr) L(alignn_bn_bwd_finalize(n_partial, n_slabs_fin, H, p.n_red, sd));
    if (e_partial != nullptr) L(alignn_bn_bwd_finalize(e_partial, e_slabs_fin, H, p.e_red, sd));
    dual_wgrad(c, GP, 4 * H, t.x, Kin, p.g_wcat, n, 4 * H, Kin, sd);
    col_sum(c, GP.p, 4 * H, n, 4 * H, p.g_bcat, sd);
    if (!dw) dual_wgrad(c, GM, H, t.y, Kin, p.g_weg, m, H, Kin, sd);
    L(alignn_slab_sum(gb_part, slabs, H, p.g_beg, sd));
}

void run_ff_dual(Ctx& c, const Tape& tp, const float* g_out, const float* g_forces, const float* g_stress) {
    const alignn_model_desc& d = *c.d;
    const alignn_model_batch& b = *c.b;
    const alignn_ff_desc& f = *c.ff;
    const int64_t N = b.g.n, E = b.g.m, Tn = b.lg.m;
    const int H = d.H, B = b.B;
    c.param_grads = true;
    c.amax_arena = c.alloc(kAmaxSlots);
    c.amax_next = 0;
    fill(c, c.amax_arena, kAmaxSlots, 0.0f, c.main);
    c.sync(c.side, c.main);
    const bool lanes = c.T != c.main && Tn >= c.d->lane_min_rows;  // the T-row kernels on lane T (dual_conv_fwd / _bwd)
    hipStream_t sT = lanes ? c.T : c.main;
    // ---- w = dL/d(pair forces), the tangent direction rt = w / 2^k, the tangents of the geometry features
    float* wmax = c.new_amax();
    float* w = c.alloc((size_t)3 * E);
    L(alignn_ff_pair_weights(g_forces, f.has_stress ? g_stress : nullptr, b.r, b.g.src, b.g.dst, b.graph_ptr, b.g.seg_ptr, f.volume,
                             f.stress_multiplier * -160.21766208f, f.add_reverse_forces, B, E, w, wmax, c.main));
    float* rt = c.alloc((size_t)3 * E);
    float* dt = c.alloc((size_t)E);
    L(alignn_ff_tangent_geometry(b.r, w, wmax, tp.bl, rt, dt, E, c.main));
    const float cc = f.grad_multiplier * (f.force_mult_natoms ? (float)N : 1.0f);
    // ---- dual forward (tangents only)
    std::vector<DMlpTape> mt(5);
    std::vector<DConvTape> ct(conv_count(d));
    DAct xa;
    xa.p = const_cast<float*>(b.atom_features);
    xa.t = c.alloc((size_t)N * d.atom_in);
    fill(c, xa.t, N * d.atom_in, 0.0f, c.main);
    DAct x = dual_mlp_fwd(c, mt[0], d.atom, tp.atom, xa, c.main);
    DAct ye;
    ye.p = tp.rbf_e;
    ye.t = c.alloc((size_t)E * d.edge_bins);
    L(alignn_rbf_tangent(tp.bl, dt, d.edge_centers, d.edge_gamma, ye.t, E, d.edge_bins, c.main));
    if (lanes) c.sync(sT, c.main);  // (rt: the bond-angle tangents below run beside the atom and bond embeddings)
    DAct y = dual_mlp_fwd(c, mt[2], d.edge2, tp.e2, dual_mlp_fwd(c, mt[1], d.edge1, tp.e1, ye, c.main), c.main);
    float* ht = c.alloc((size_t)Tn);
    if (f.lg_on_fly)
        L(alignn_bond_cosine_tangent(b.r, rt, b.lg.src, b.lg.dst, ht, Tn, sT));
    else
        fill(c, ht, Tn, 0.0f, sT);
    DAct za;
    za.p = tp.rbf_a;
    za.t = c.alloc((size_t)Tn * d.angle_bins);
    L(alignn_rbf_tangent(tp.hcos, ht, d.angle_centers, d.angle_gamma, za.t, Tn, d.angle_bins, sT));
    DAct z = dual_mlp_fwd(c, mt[4], d.angle2, tp.a2, dual_mlp_fwd(c, mt[3], d.angle1, tp.a1, za, sT), sT);
    if (c.unsupported) return;
    int k = 0;
    for (int i = 0; i < d.alignn_layers; ++i) {
        DAct xo, mo, yo, zo;
        dual_conv_fwd(c, ct[k], d.convs[k], tp.convs[k], x, y, true, xo, mo);
        ++k;
        dual_conv_fwd(c, ct[k], d.convs[k], tp.convs[k], mo, z, i + 1 < d.alignn_layers, yo, zo);
        ++k;
        x = xo, y = yo, z = zo;
        if (c.unsupported) return;
    }
    for (int i = 0; i < d.gcn_layers; ++i) {
        DAct xo, yo;
        dual_conv_fwd(c, ct[k], d.convs[k], tp.convs[k], x, y, i + 1 < d.gcn_layers, xo, yo);
        ++k;
        x = xo, y = yo;
        if (c.unsupported) return;
    }
    // ---- readout E_g = fc(mean_i x_i) (alignn_atomwise.py:464-466) and its reverse with the two seeds
    float* hpt = c.alloc((size_t)B * H);
    L(alignn_segment_mean_fwd(x.t, b.graph_ptr, hpt, B, H, c.main));
    L(alignn_ff_fc_grad(g_out, cc, f.energy_mult_natoms, wmax, b.graph_ptr, tp.pool, hpt, d.g_fc_W, d.g_fc_b, B, H, c.main));
    DAct gx, gy, gz;
    gx.p = c.alloc((size_t)N * H);
    gx.t = c.alloc((size_t)N * H);
    L(alignn_ff_readout_seed(g_out, cc, f.energy_mult_natoms, wmax, b.graph_ptr, d.fc_W, gx.p, gx.t, B, N, H, c.main));
    // ---- reverse over the tape
    k = conv_count(d) - 1;
    for (int i = d.gcn_layers - 1; i >= 0; --i, --k) {
        DAct nx, ny;
        dual_conv_bwd(c, ct[k], gx, gy, nx, ny);
        gx = nx, gy = ny;
        if (c.unsupported) return;
    }
    for (int i = d.alignn_layers - 1; i >= 0; --i) {
        DAct gm, nz, nx, ny;
        dual_conv_bwd(c, ct[k], gy, gz, gm, nz);  // edge_update: nodes = bonds, edges = triplets
        --k;
        gz = nz;
        dual_conv_bwd(c, ct[k], gx, gm, nx, ny);  // node_update: outputs (x, m)
        --k;
        gx = nx, gy = ny;
        if (c.unsupported) return;
    }
    if (lanes && !gz.on_T) c.sync(sT, c.main);
    DAct gz1 = dual_mlp_bwd(c, mt[4], gz, true, sT);  // (the bond-angle embedding stays on lane T, beside the two below)
    dual_mlp_bwd(c, mt[3], gz1, false, sT);
    DAct gy1 = dual_mlp_bwd(c, mt[2], gy, true, c.main);
    dual_mlp_bwd(c, mt[1], gy1, false, c.main);
    dual_mlp_bwd(c, mt[0], gx, false, c.main);
    c.sync(c.main, c.side);
    c.sync(c.main, c.T);
}

bool desc_ok(const alignn_model_desc* d, const alignn_model_batch* b) {
    if (d == nullptr || b == nullptr || d->convs == nullptr) return false;
    if (d->alignn_layers < 1 || d->gcn_layers < 1 || d->H <= 0 || (d->H & 3) || d->out_features <= 0) return false;
    if (b->g.n <= 0 || b->g.m <= 0 || b->lg.m <= 0 || b->B <= 0 || b->lg.n != b->g.m) return false;
    return true;
}

void set_streams(Ctx& c, alignn_stream_t st) {
    c.main = (hipStream_t)st;
    c.T = c.d->lane_T ? (hipStream_t)c.d->lane_T : c.main;
    c.side = c.d->side ? (hipStream_t)c.d->side : c.main;
    c.aux = c.d->aux ? (hipStream_t)c.d->aux : c.main;
    // LayerNorm flavour: helper streams like the BatchNorm flavour's.  (Round 5 found its steps not bit-reproducible on helper
    // streams - forces off by 1e-3, run to run - and traced it to the packed-fp32 code hipcc emitted for ln_silu_bwd_kernel: one
    // float4 component of lanes 48-63 wrong while an MFMA kernel of another stream shared the compute unit; csrc/norm.hip and
    // csrc/dual.hip are built without SLP vectorisation since - alignn_amd/build.py)
    // ALIGNN_AMD_LN_STREAMS=0 puts the LayerNorm flavour on ONE stream (bit 0: lane T + aux, bit 1: side).
    if (c.d->norm == 1) {
        static int ln_streams = -1;
        if (ln_streams < 0) ln_streams = getenv("ALIGNN_AMD_LN_STREAMS") ? atoi(getenv("ALIGNN_AMD_LN_STREAMS")) : 3;
        if (!(ln_streams & 1)) c.T = c.aux = c.main;
        if (!(ln_streams & 2)) c.side = c.main;
    }
}

bool take_pool(Ctx& c) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices || !g_pool[dev].ok) return false;
    c.pool = &g_pool[dev];
    return true;
}

// The workspace layout of one (model, batch, stream configuration): [forward tape | backward buffers | scratch per stream].
struct Plan {
    size_t fwd_persist = 0, all_persist = 0, peak_fwd[4] = {0, 0, 0, 0}, peak_all[4] = {0, 0, 0, 0};
    bool unsupported = false;
    int line = 0, rc = 0;
    size_t fwd_total() const { return fwd_persist + peak_fwd[0] + peak_fwd[1] + peak_fwd[2] + peak_fwd[3]; }
    size_t total() const { return all_persist + peak_all[0] + peak_all[1] + peak_all[2] + peak_all[3]; }
    void place_scratch(Ctx& c, bool full) const {
        size_t off = full ? all_persist : fwd_persist;
        for (int i = 0; i < 4; ++i) {
            c.sbase[i] = off;
            off += full ? peak_all[i] : peak_fwd[i];
        }
    }
};

Plan make_plan(const alignn_model_desc* d, const alignn_model_batch* b, alignn_stream_t stream) {
    Plan pl;
    Ctx c{d, b, reinterpret_cast<char*>(4096)};  // (any non-null base: pointers are compared with NULL, never used)
    set_streams(c, stream);
    Tape tp;
    run_forward(c, tp, nullptr);
    pl.fwd_persist = c.off;
    for (int i = 0; i < 4; ++i) pl.peak_fwd[i] = c.speak[i];
    if (!c.unsupported) run_backward(c, tp, nullptr);
    pl.all_persist = c.off;
    for (int i = 0; i < 4; ++i) pl.peak_all[i] = c.speak[i];
    pl.unsupported = c.unsupported;
    pl.line = c.unsupported_line;
    pl.rc = c.rc;
    return pl;
}

bool ff_ok(const alignn_model_desc* d, const alignn_model_batch* b, const alignn_ff_desc* f) {
    if (!desc_ok(d, b) || f == nullptr || d->norm != 1 || d->out_features != 1) return false;
    if (b->g.out_ptr == nullptr || b->g.out_slot == nullptr || b->lg.out_ptr == nullptr || b->lg.out_slot == nullptr) return false;
    if (f->has_stress && f->volume == nullptr) return false;
    if (!f->lg_on_fly && b->h == nullptr) return false;
    return true;
}

Plan make_ff_plan(const alignn_model_desc* d, const alignn_model_batch* b, const alignn_ff_desc* f, alignn_stream_t stream) {
    Plan pl;
    Ctx c{d, b, reinterpret_cast<char*>(4096)};
    c.ff = f;
    set_streams(c, stream);
    Tape tp;
    run_ff_eval(c, tp, nullptr, nullptr, nullptr);
    pl.fwd_persist = c.off;
    for (int i = 0; i < 4; ++i) pl.peak_fwd[i] = c.speak[i];
    if (!c.unsupported) run_ff_dual(c, tp, nullptr, nullptr, nullptr);
    pl.all_persist = c.off;
    for (int i = 0; i < 4; ++i) pl.peak_all[i] = c.speak[i];
    pl.unsupported = c.unsupported;
    pl.line = c.unsupported_line;
    pl.rc = c.rc;
    return pl;
}

}  // namespace

extern "C" {

size_t alignn_ff_desc_sizeof(void) { return sizeof(alignn_ff_desc); }

// debugging aid (ALIGNN_AMD_DEBUG_ALLOCS=1): the (offset, bytes) pairs of the persistent workspace allocations of the launching
// calls since the last query, in program order; returns the count (at most cap pairs are written) - tools/ff_alloc_diff.py
int alignn_debug_allocs(size_t* out, int cap) {
    const int n = (int)g_dbg_allocs.size();
    for (int i = 0; i < n && i < cap; ++i) {
        out[2 * i] = g_dbg_allocs[i].first;
        out[2 * i + 1] = g_dbg_allocs[i].second;
    }
    g_dbg_allocs.clear();
    return n;
}

int alignn_ff_plan(const alignn_model_desc* d, const alignn_model_batch* b, const alignn_ff_desc* f, size_t* eval_bytes,
                   size_t* total_bytes) {
    if (!ff_ok(d, b, f) || eval_bytes == nullptr || total_bytes == nullptr) return (int)hipErrorInvalidValue;
    Plan pl = make_ff_plan(d, b, f, nullptr);
    if (pl.unsupported) {
        if (getenv("ALIGNN_AMD_DEBUG")) fprintf(stderr, "alignn_ff_plan: kernel choice not carried (model.hip:%d)\n", pl.line);
        return (int)hipErrorNotSupported;
    }
    *eval_bytes = pl.fwd_total();
    *total_bytes = pl.total();
    return pl.rc;
}

int alignn_ff_eval(const alignn_model_desc* d, const alignn_model_batch* b, const alignn_ff_desc* f, void* workspace,
                   size_t workspace_bytes, float* out, float* forces, float* stress, alignn_stream_t stream) {
    if (!ff_ok(d, b, f) || workspace == nullptr || out == nullptr || forces == nullptr) return (int)hipErrorInvalidValue;
    const Plan pl = make_ff_plan(d, b, f, stream);
    if (pl.unsupported) return (int)hipErrorNotSupported;
    Ctx c{d, b, static_cast<char*>(workspace)};
    c.cap = workspace_bytes;
    c.launch = true;
    c.ff = f;
    set_streams(c, stream);
    if (workspace_bytes >= pl.total())
        pl.place_scratch(c, true);
    else if (workspace_bytes >= pl.fwd_total())
        pl.place_scratch(c, false);  // (values only: MD, validation)
    else
        return (int)hipErrorInvalidValue;
    if ((c.T != c.main || c.side != c.main || c.aux != c.main) && !take_pool(c)) return (int)hipErrorNotInitialized;
    Tape tp;
    run_ff_eval(c, tp, out, forces, stress);
    if (c.unsupported) return (int)hipErrorNotSupported;
    return c.rc;
}

int alignn_ff_grad(const alignn_model_desc* d, const alignn_model_batch* b, const alignn_ff_desc* f, void* workspace,
                   size_t workspace_bytes, const float* g_out, const float* g_forces, const float* g_stress, float* gflat,
                   float* gflat_t, int64_t grad_floats, float* gsink, float* gsink_t, int64_t sink_floats, alignn_stream_t stream) {
    if (!ff_ok(d, b, f) || workspace == nullptr || gflat == nullptr || gflat_t == nullptr || grad_floats <= 0 || (grad_floats & 3))
        return (int)hipErrorInvalidValue;
    if (gsink != nullptr && (gsink_t == nullptr || sink_floats <= 0 || (sink_floats & 3) || (reinterpret_cast<uintptr_t>(gsink) & 15) ||
                             (reinterpret_cast<uintptr_t>(gsink_t) & 15)))
        return (int)hipErrorInvalidValue;
    const Plan pl = make_ff_plan(d, b, f, stream);
    if (pl.unsupported) return (int)hipErrorNotSupported;
    if (workspace_bytes < pl.total()) return (int)hipErrorInvalidValue;
    Ctx c{d, b, static_cast<char*>(workspace)};
    c.cap = workspace_bytes;
    c.ff = f;
    set_streams(c, stream);
    pl.place_scratch(c, true);
    if ((c.T != c.main || c.side != c.main || c.aux != c.main) && !take_pool(c)) return (int)hipErrorNotInitialized;
    Tape tp;
    run_ff_eval(c, tp, nullptr, nullptr, nullptr);  // (plan only: where the evaluation left its tape)
    if (c.unsupported) return (int)hipErrorNotSupported;
    c.launch = true;
    c.toff = gflat_t - gflat;
    fill(c, gflat_t, grad_floats, 0.0f, c.main);  // (only the weight blocks get a tangent half)
    if (gsink != nullptr) {
        c.sink = gsink;
        c.sink_floats = sink_floats;
        c.sink_toff = gsink_t - gsink;
        fill(c, gsink_t, sink_floats, 0.0f, c.main);
    }
    run_ff_dual(c, tp, g_out, g_forces, g_stress);
    if (c.unsupported) return (int)hipErrorNotSupported;
    if (c.rc == 0) c.rc = alignn_add_inplace(gflat, gflat_t, grad_floats, c.main);
    if (c.rc == 0 && gsink != nullptr) c.rc = alignn_add_inplace(gsink, gsink_t, sink_floats, c.main);
    return c.rc;
}

int alignn_model_init(void) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= kMaxDevices) return (int)hipErrorInvalidDevice;
    if (g_pool[dev].ok) return 0;
    for (int i = 0; i < kEvents; ++i) {
        e = hipEventCreateWithFlags(&g_pool[dev].ev[i], hipEventDisableTiming);
        if (e != hipSuccess) return (int)e;
    }
    g_pool[dev].ok = true;
    return 0;
}

size_t alignn_model_sizeof(int which) {
    switch (which) {
        case 0: return sizeof(alignn_mlp_params);
        case 1: return sizeof(alignn_conv_params);
        case 2: return sizeof(alignn_graph_csr);
        case 3: return sizeof(alignn_model_batch);
        case 4: return sizeof(alignn_model_desc);
        default: return 0;
    }
}

int alignn_model_plan(const alignn_model_desc* d, const alignn_model_batch* b, size_t* fwd_bytes, size_t* total_bytes) {
    if (!desc_ok(d, b) || fwd_bytes == nullptr || total_bytes == nullptr) return (int)hipErrorInvalidValue;
    Plan pl = make_plan(d, b, nullptr);
    if (pl.unsupported) {
        if (getenv("ALIGNN_AMD_DEBUG")) fprintf(stderr, "alignn_model_plan: kernel choice not carried (model.hip:%d)\n", pl.line);
        return (int)hipErrorNotSupported;
    }
    *fwd_bytes = pl.fwd_total();
    *total_bytes = pl.total();
    return pl.rc;
}

int alignn_model_fwd(const alignn_model_desc* d, const alignn_model_batch* b, void* workspace, size_t workspace_bytes,
                     float* out, alignn_stream_t stream) {
    if (!desc_ok(d, b) || workspace == nullptr || out == nullptr) return (int)hipErrorInvalidValue;
    const bool timing = getenv("ALIGNN_AMD_DEBUG_TIME") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    const Plan pl = make_plan(d, b, stream);
    const auto t1 = std::chrono::steady_clock::now();
    if (pl.unsupported) return (int)hipErrorNotSupported;
    Ctx c{d, b, static_cast<char*>(workspace)};
    c.cap = workspace_bytes;
    c.launch = true;
    c.timing = timing;
    set_streams(c, stream);
    if (workspace_bytes >= pl.total())
        pl.place_scratch(c, true);
    else if (workspace_bytes >= pl.fwd_total())
        pl.place_scratch(c, false);  // (a forward nobody will differentiate: the tape and the forward's scratch only)
    else
        return (int)hipErrorInvalidValue;
    if ((c.T != c.main || c.side != c.main || c.aux != c.main) && !take_pool(c)) return (int)hipErrorNotInitialized;
    Tape tp;
    run_forward(c, tp, out);
    if (timing) {
        const double tot = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        fprintf(stderr, "alignn_model_fwd: %.3f ms host (plan %.3f ms, %d stream syncs %.3f ms, %d launch calls)\n", tot * 1e3,
                std::chrono::duration<double>(t1 - t0).count() * 1e3, c.n_sync, c.t_sync * 1e3, c.n_launch);
    }
    if (c.unsupported) return (int)hipErrorNotSupported;
    return c.rc;
}

int alignn_model_bwd(const alignn_model_desc* d, const alignn_model_batch* b, void* workspace, size_t workspace_bytes,
                     const float* g_out, alignn_stream_t stream) {
    if (!desc_ok(d, b) || workspace == nullptr || g_out == nullptr) return (int)hipErrorInvalidValue;
    const bool timing = getenv("ALIGNN_AMD_DEBUG_TIME") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    const Plan pl = make_plan(d, b, stream);
    if (pl.unsupported) return (int)hipErrorNotSupported;
    if (workspace_bytes < pl.total()) return (int)hipErrorInvalidValue;
    Ctx c{d, b, static_cast<char*>(workspace)};
    c.cap = workspace_bytes;
    c.timing = timing;
    set_streams(c, stream);
    pl.place_scratch(c, true);
    if ((c.T != c.main || c.side != c.main || c.aux != c.main) && !take_pool(c)) return (int)hipErrorNotInitialized;
    Tape tp;
    run_forward(c, tp, nullptr);  // (plan only: where the forward left its tape)
    if (c.unsupported) return (int)hipErrorNotSupported;
    const auto t1 = std::chrono::steady_clock::now();
    c.launch = true;
    run_backward(c, tp, g_out);
    if (timing) {
        const double tot = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        fprintf(stderr, "alignn_model_bwd: %.3f ms host (plans %.3f ms, %d stream syncs %.3f ms, %d launch calls)\n", tot * 1e3,
                std::chrono::duration<double>(t1 - t0).count() * 1e3, c.n_sync, c.t_sync * 1e3, c.n_launch);
    }
    if (c.unsupported) return (int)hipErrorNotSupported;
    return c.rc;
}

size_t alignn_model_infer_workspace(const alignn_model_desc* d, const alignn_model_batch* b) {
    if (!desc_ok(d, b)) return 0;
    Ctx c{d, b, reinterpret_cast<char*>(4096)};
    set_streams(c, nullptr);
    run_infer(c, nullptr);
    return (c.unsupported || c.rc != 0) ? 0 : c.off;
}

int alignn_model_infer(const alignn_model_desc* d, const alignn_model_batch* b, void* workspace, size_t workspace_bytes,
                       float* out, alignn_stream_t stream) {
    if (!desc_ok(d, b) || workspace == nullptr || out == nullptr) return (int)hipErrorInvalidValue;
    Ctx c{d, b, static_cast<char*>(workspace)};
    c.cap = workspace_bytes;
    c.launch = true;
    set_streams(c, stream);
    if (c.T != c.main && !take_pool(c)) return (int)hipErrorNotInitialized;
    run_infer(c, out);
    if (c.unsupported) return (int)hipErrorNotSupported;
    return c.rc;
}

}  // extern "C"
