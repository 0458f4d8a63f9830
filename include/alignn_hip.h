/*
 * alignn_hip.h - C ABI of libalignn_hip.so: the MI355X (gfx950) kernels behind the
 * ALIGNN message-passing hot path.
 *
 * Drop-in boundary.  The reference (usnistgov/alignn) is pure Python; the "FFI" it has for this
 * path is the set of torch / DGL calls its model files make.  Each entry point below names the
 * reference call site(s) it replaces (paths relative to the reference root).  A maintainer binds
 * them with ctypes; nothing torch-specific crosses this boundary.
 *
 * Conventions
 *   - all pointers are DEVICE pointers into buffers owned by the caller (PyTorch's caching
 *     allocator in our host code); the library never allocates, frees or synchronises;
 *   - every call only enqueues work on `stream` (graph-capture safe) and returns a hipError_t
 *     as int (0 == hipSuccess); argument errors return hipErrorInvalidValue (1);
 *   - matrices are row-major fp32; `ld*` are leading dimensions in ELEMENTS;
 *   - graph structure is "canonical CSR": the m edges of a graph are stored so that the in-edges
 *     of one destination node are contiguous ("segment").  Segment s covers edge rows
 *     [seg_ptr[s], seg_ptr[s+1]) and belongs to node seg_node[s] (seg_node == NULL: node s).
 *     Edge-feature row k IS CSR slot k.  `src[k]` is the source node of slot k.  The by-source
 *     view (needed by backward) is out_ptr[n+1] / out_slot[m] (slots grouped by source node) and
 *     `dst[k]`, the destination node of slot k.  Indices are int32;
 *   - reductions are atomics-free and run in a fixed order: results are bit-reproducible from
 *     run to run (the reference trains with torch.use_deterministic_algorithms(True),
 *     alignn/train.py:164-179).
 */
#ifndef ALIGNN_HIP_H
#define ALIGNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* alignn_stream_t; /* hipStream_t */

/* Library / build identification: returns a static string such as "alignn_hip 0.1 gfx950". */
const char* alignn_version(void);

/* ------------------------------------------------------------------------------------------
 * Dense projections (fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32; exact fp32).
 * Replace torch.nn.Linear forward/backward at alignn/models/alignn.py:98-99,101,104,110
 * (src_gate, dst_gate, edge_gate, dst_update, src_update), :175-177 (MLPLayer) and :341 (fc).
 * ------------------------------------------------------------------------------------------ */

/* C[M,N] = A[M,K] * W[N,K]^T (+ bias[N]) (+ addend[M,N])      -- nn.Linear forward */
int alignn_gemm_nt(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                   const float* addend, int64_t ldadd, float* C, int64_t ldc,
                   int64_t M, int N, int K, alignn_stream_t stream);

/* C[M,K] = G[M,N] * W[N,K] (+ addend[M,K])                    -- nn.Linear input gradient */
int alignn_gemm_nn(const float* G, int64_t ldg, const float* W, int64_t ldw,
                   const float* addend, int64_t ldadd, float* C, int64_t ldc,
                   int64_t M, int N, int K, alignn_stream_t stream);
/* The same product with the reduction split into slabs when there are few output tiles and a long reduction (the input
 * gradient of the fused node projection on atom rows: [n, 4H] x [4H, H]): `_workspace` returns the bytes of slab storage the
 * shape wants, 0 when alignn_gemm_nn is the right call.  Slabs are added in a fixed order (bit-reproducible). */
size_t alignn_gemm_nn_split_workspace(int64_t M, int N, int K);
int alignn_gemm_nn_split(const float* G, int64_t ldg, const float* W, int64_t ldw, const float* addend, int64_t ldadd,
                         float* C, int64_t ldc, int64_t M, int N, int K, void* workspace, size_t workspace_bytes,
                         alignn_stream_t stream);

/* dW[N,K] = G[M,N]^T * A[M,K]                                  -- nn.Linear weight gradient.
 * Split over M in `splits` deterministic slabs; `workspace` holds splits*N*K floats
 * (alignn_gemm_tn_workspace tells how many bytes for a given shape). */
size_t alignn_gemm_tn_workspace(int64_t M, int N, int K);
/* g_amax / a_amax (both or neither; may be NULL): device scalars holding max|G| and max|A| (upper bounds are
 * fine) - when both are given and the shape suits the split-product kernel, the three-product fp16 scheme runs
 * instead of the six-product bf16 one (see alignn_gemm_nt_f16x3). */
int alignn_gemm_tn(const float* G, int64_t ldg, const float* g_amax, const float* A, int64_t lda,
                   const float* a_amax, float* dW, int64_t lddw, int64_t M, int N, int K, void* workspace,
                   size_t workspace_bytes, alignn_stream_t stream);

/* fp32-accurate projection on the bf16 matrix cores ("bf16x6", csrc/gemm_x6.hip): every fp32 value is cut
 * into three truncated bf16 slices and the six slice products of weight >= 2^-16 are accumulated in fp32
 * (error <= an fp32 FMA chain; 2.7x the fp32-MFMA rate, which makes the HxH projections HBM-bound).
 *   alignn_split_bf16x3: slices W[N,K] (transpose != 0: the [N,K] matrix W^T of a stored [K,N] W) into the
 *                        kernel's DMA image (three bf16 planes, k-blocked, rows padded to 256, XOR-swizzled);
 *                        `out` holds alignn_split_bf16x3_bytes(N, K) bytes; K % 16 == 0.  Run once per
 *                        weight per step (384 KiB for 256x256).
 *   alignn_gemm_nt_x6:   C[M,N] = A[M,K] * W[N,K]^T (+bias) (+addend), W given pre-sliced;
 *                        needs K % 16 == 0, N >= 128, N % 4 == 0 (alignn_gemm_nt_x6_supported). */
size_t alignn_split_bf16x3_bytes(int N, int K);
int alignn_split_bf16x3(const float* W, int64_t ldw, int N, int K, int transpose, void* out,
                        alignn_stream_t stream);
int alignn_gemm_nt_x6_supported(int64_t M, int N, int K);
int alignn_gemm_nt_x6(const float* A, int64_t lda, const void* Wsplit, const float* bias,
                      const float* addend, int64_t ldadd, float* C, int64_t ldc,
                      int64_t M, int N, int K, alignn_stream_t stream);

/* Weight gradient on the split product (both operands sliced in registers): slab partials ws[z][N][K] for
 * z < alignn_gemm_tn_x6_splits(M,N,K); needs N % 256 == 0, K % 256 == 0, M >= 4096.  alignn_gemm_tn uses it
 * automatically for such shapes and sums the slabs. */
int alignn_gemm_tn_x6_supported(int64_t M, int N, int K);
size_t alignn_gemm_tn_x6_workspace(int64_t M, int N, int K);
int alignn_gemm_tn_x6_splits(int64_t M, int N, int K);
int alignn_gemm_tn_x6_partials(const float* G, int64_t ldg, const float* g_amax, const float* X, int64_t ldx,
                               const float* x_amax, int64_t M, int N, int K, void* workspace,
                               size_t workspace_bytes, alignn_stream_t stream);

/* Same projection with HALF the matrix-core work ("f16x3"), for callers that know max|A|: operands are scaled
 * by a power of two into the fp16 range and cut into two fp16 slices (11 + 11 mantissa bits); the three products
 * hh + hl + lh are accumulated in fp32 and the scales undone in the epilogue - fp32-grade error (dropped term
 * < 2^-22), sustained-MFMA time 153 us instead of 306 us for T x 256 x 256, i.e. the projection is HBM-bound.
 *   alignn_absmax:        amax[0] = max|X| (device scalar; the fused producers track it themselves - see the
 *                         `amax` argument of alignn_bn_silu_fwd, alignn_egc_bwd_dst, alignn_egc_bwd_lg_fused, ...)
 *   alignn_split_f16x2:   pre-slice W * 2^s (s from *w_amax) into the two-plane DMA image
 *   alignn_gemm_nt_f16x3: C = A W^T (+bias)(+addend); shapes as alignn_gemm_nt_x6_supported */
int alignn_absmax(const float* X, int64_t ldx, int64_t rows, int F, float* amax, alignn_stream_t stream);
size_t alignn_split_f16x2_bytes(int N, int K);
int alignn_split_f16x2(const float* W, int64_t ldw, int N, int K, int transpose, const float* w_amax, void* out,
                       alignn_stream_t stream);
/* The image of W and the image of W^T (N % 16 == 0 as well) in ONE launch: what alignn_split_f16x2(transpose=0) and
 * alignn_split_f16x2(transpose=1) produce, bit for bit.  The forward products (alignn/models/alignn.py:98-110 nn.Linear)
 * and torch.autograd's input gradients of the same Linear use the two images of one optimizer step. */
int alignn_split_f16x2_both(const float* W, int64_t ldw, int N, int K, const float* w_amax, void* out, void* out_t,
                            alignn_stream_t stream);
/* max|W| + both images of EVERY split-product weight of a model in three stream operations (once per optimizer step; the
 * per-weight calls above cost two launches each, ~28 weights): `descs` = n_weights device records {const float* W;
 * int64_t ldw; int32_t N, K; float* amax; void* out; void* out_t} (48 bytes), `amax_slots` = the n_weights floats the
 * records' `amax` fields point into (zeroed here first); N, K multiples of 16, ldw of 4. */
int alignn_prepare_weights(const void* descs, int n_weights, float* amax_slots, alignn_stream_t stream);
/* *amax = max(*amax, max|X|): alignn_absmax (below) without its reset, for a slot known to hold 0 */
int alignn_absmax_raise(const float* X, int64_t ldx, int64_t rows, int F, float* amax, alignn_stream_t stream);
int alignn_gemm_nt_f16x3(const float* A, int64_t lda, const float* a_amax, const void* Wsplit,
                         const float* w_amax, const float* bias, const float* addend, int64_t ldadd, float* C,
                         int64_t ldc, int64_t M, int N, int K, alignn_stream_t stream);

/* The same projection when its OUTPUT is the gradient g_y of y = r + silu(BatchNorm(Xn)) (the next thing autograd does
 * with g_y is BatchNorm's backward: alignn/models/alignn.py:126 bn_edges / :178 MLPLayer's BatchNorm1d under
 * torch.autograd): additionally writes, per row tile of the product, the column sums over the tile's rows of
 *   gz = g_y * silu'((Xn - mean) * scale + beta)   and   gz * (Xn - mean) * rstd
 * into red_partial [row_tiles + 1][2][N] (the last slab is scratch for the kernel, never summed; nstat = [4][N] as
 * written by alignn_bn_finalize; a "row tile" is the 64 / 128-row unit alignn_gemm_nt_x6_row_tiles counts for the shape
 * - the block tile of the one-tile kernels, the 64-row wave strip of the persistent one).  alignn_bn_bwd_finalize
 * over `alignn_gemm_nt_x6_row_tiles(M, N, K)` slabs then gives dbeta / dgamma - what alignn_bn_silu_bwd_reduce
 * computes with two more passes over g_y and Xn. */
/* The edge-gate projection of EdgeGatedGraphConv with DGL's u_add_v folded in (alignn/models/alignn.py:98-101:
 * m = src_gate(x)[u] + dst_gate(x)[v] + edge_gate(y)):  C[e] = A[e] W^T + bias + P[src[e]][0:N] + P[dst[e]][N:2N],
 * P = the fused node projection [n, ldp] = A | Bd | Bh | Ux.  alignn_egc_gate_fwd_pre is the gate pass that takes such
 * an M (already m: it neither gathers A / Bd nor stores m) - same values, one write and one gather of a T-row
 * tensor less than alignn_gemm_nt_f16x3 + alignn_egc_gate_fwd. */
int alignn_gemm_nt_f16x3_gather(const float* A, int64_t lda, const float* a_amax, const void* Wsplit,
                                const float* w_amax, const float* bias, float* C, int64_t ldc, int64_t M, int N, int K,
                                const float* P, int64_t ldp, const int32_t* src, const int32_t* dst,
                                float* stats_partial, alignn_stream_t stream);
/* The same with the second gathered row from its own table: C[e] = A[e] W^T + bias + P[src[e]][0:N] + Bd2[rank[e]][0:N].
 * For a line graph: Bd2 = the Bd block of P with its rows in SEGMENT order (alignn_gather_rows_ld by seg_node), rank[e] =
 * the segment of edge row e - consecutive rows then read consecutive table rows instead of one random row per segment. */
int alignn_gemm_nt_f16x3_gather2(const float* A, int64_t lda, const float* a_amax, const void* Wsplit, const float* w_amax,
                                 const float* bias, float* C, int64_t ldc, int64_t M, int N, int K, const float* P, int64_t ldp,
                                 const int32_t* src, const float* Bd2, int64_t ldbd2, const int32_t* rank,
                                 float* stats_partial, alignn_stream_t stream);
/* out[r][0:F] = in[perm[r]][0:F] for matrices with leading dimensions (F, ld_in, ld_out multiples of 4) */
int alignn_gather_rows_ld(const float* in, int64_t ld_in, const int32_t* perm, float* out, int64_t ld_out, int64_t rows, int F,
                          alignn_stream_t stream);
/* stats_partial (above: may be NULL) / alignn_gemm_nt_f16x3_stats: the projection also leaves, per row tile, the column
 * sums of its output and of its square - [alignn_gemm_nt_x6_row_tiles + 1][2][N] (last slab: scratch), the slab layout
 * alignn_bn_finalize takes -
 * i.e. the BatchNorm statistics torch's BatchNorm1d would compute with another pass over the tensor
 * (alignn/models/alignn.py:122-127 bn_edges, :175-179 MLPLayer).  alignn_egc_gate_fwd_pre_norm is the gate pass for an M
 * that already holds m AND whose statistics are known: it writes Y' = Y + silu((m - mean) scale + beta) itself (e_stat as
 * from alignn_bn_finalize; Y may be NULL), so the edge branch needs no separate normalise / activate pass. */
int alignn_gemm_nt_f16x3_stats(const float* A, int64_t lda, const float* a_amax, const void* Wsplit,
                               const float* w_amax, const float* bias, float* C, int64_t ldc, int64_t M, int N, int K,
                               float* stats_partial, alignn_stream_t stream);
int alignn_egc_gate_fwd_pre_norm(const float* P, const float* M, const int32_t* seg_ptr, const int32_t* seg_node,
                                 const int32_t* src, int64_t n_seg, int64_t m_rows, int H, float* XPRE, float* S0,
                                 float* HH, float* n_partial, const float* e_stat, const float* Y, float* YOUT,
                                 float* y_amax, alignn_stream_t stream);
int alignn_egc_gate_fwd_pre(const float* P, float* M, const int32_t* seg_ptr, const int32_t* seg_node,
                            const int32_t* src, int64_t n_seg, int64_t m_rows, int H, float* XPRE, float* S0, float* HH,
                            float* e_partial, float* n_partial, alignn_stream_t stream);
int alignn_gemm_nt_x6_row_tiles(int64_t M, int N, int K);
int alignn_gemm_nt_f16x3_bnred(const float* A, int64_t lda, const float* a_amax, const void* Wsplit,
                               const float* w_amax, const float* bias, const float* addend, int64_t ldadd,
                               float* C, int64_t ldc, int64_t M, int N, int K, const float* Xn, int64_t ldxn,
                               const float* nstat, float* red_partial, alignn_stream_t stream);

/* Input gradient AND weight gradient of a 256 -> 256 projection in one pass over the projection's output gradient G
 * (csrc/gemm_dw.hip; replaces the pair alignn_gemm_nt_f16x3[_bnred] + alignn_gemm_tn, which each streamed G from HBM):
 *     C[M,256]    = G[M,256] W[256,256] (+ addend)      autograd's grad_input of nn.Linear, alignn/models/alignn.py:101 (edge_gate)
 *     dW[256,256] = G[M,256]^T Y[M,256]                 ... and its grad_weight; Y = the Linear's input
 * f16x3 split products as alignn_gemm_nt_f16x3 (same bits for C): `Wt_split` = alignn_split_f16x2(W, transpose = 1) with its
 * max|W| scalar, g_amax / y_amax = device scalars >= max|G|, max|Y|.  Optional, as alignn_gemm_nt_f16x3_bnred: Xn / nstat /
 * red_partial - the BatchNorm-backward column sums of C against the pre-activation Xn, left as
 * alignn_gemm_dgrad_wgrad_slabs(M) * 2 slabs of [2][256] for alignn_bn_bwd_finalize.  `workspace`:
 * alignn_gemm_dgrad_wgrad_workspace(M) bytes (one 256 x 256 partial of dW per workgroup, summed in fp64 in slab order:
 * bit-reproducible).  Shapes: N = K = 256, M >= 4096 (alignn_gemm_dgrad_wgrad_supported). */
int alignn_gemm_dgrad_wgrad_supported(int64_t M, int N, int K);
int alignn_gemm_dgrad_wgrad_slabs(int64_t M);
size_t alignn_gemm_dgrad_wgrad_workspace(int64_t M);
int alignn_gemm_dgrad_wgrad_f16x3(const float* G, int64_t ldg, const float* g_amax, const float* Y, int64_t ldy,
                                  const float* y_amax, const void* Wt_split, const float* w_amax, const float* addend,
                                  int64_t ldadd, float* C, int64_t ldc, const float* Xn, int64_t ldxn, const float* nstat,
                                  float* red_partial, float* dW, int64_t lddw, int64_t M, void* workspace,
                                  size_t workspace_bytes, alignn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Column statistics / BatchNorm1d + SiLU (+ residual).
 * Replace nn.BatchNorm1d (training: batch statistics over ALL rows; eps 1e-5; momentum 0.1,
 * unbiased running_var) + F.silu at alignn/models/alignn.py:122-127 and :175-179.
 * ------------------------------------------------------------------------------------------ */

/* number of partial slabs alignn_col_stats writes for `rows` rows (each slab = 2*F floats) */
int alignn_col_stats_slabs(int64_t rows);
/* partial[s][0][f] = sum_r X[r,f], partial[s][1][f] = sum_r X[r,f]^2 over slab s's rows */
int alignn_col_stats(const float* X, int64_t ldx, int64_t rows, int F, float* partial,
                     alignn_stream_t stream);
/* The same statistics as PIVOT SLABS - well-conditioned when |mean| >> std, where sum x^2 - (sum x)^2 / n cancels and a
 * float32 mean cannot even hold the sub-ulp part: partial[s][0][f] = a pivot p (the first value slab s saw in column f),
 * partial[s][1][f] = sum (x - p), partial[s][2][f] = sum (x - p)^2 over the slab's rows, followed by counts[s] (float) at
 * partial + slabs*3*F; partial holds slabs*(3*F+1) floats.  Threads / waves / slabs are merged by re-centring onto one
 * pivot (float64 in alignn_bn_finalize_welford; same outputs as alignn_bn_finalize).  The gate passes
 * (alignn_egc_gate_fwd, _pre, _pre_norm) write their e_partial / n_partial in THIS layout (slabs =
 * alignn_egc_slabs(n_seg)); the projection epilogues (alignn_gemm_nt_f16x3_stats / _gather) keep plain sum /
 * sum-of-squares slabs for alignn_bn_finalize. */
int alignn_col_stats_welford(const float* X, int64_t ldx, int64_t rows, int F, float* partial, alignn_stream_t stream);
int alignn_bn_finalize_welford(const float* partial, int slabs, int64_t rows, int F, const float* gamma, const float* beta,
                               float eps, float momentum, float* running_mean, float* running_var, float* stat,
                               alignn_stream_t stream);
/* column sums only: out[f] = sum_r X[r,f] (bias gradients); workspace = slabs*2*F floats */
int alignn_col_sum(const float* X, int64_t ldx, int64_t rows, int F, float* out, float* workspace,
                   alignn_stream_t stream);

/* Finalise: from `slabs` partial slabs over `rows` rows produce, per feature,
 *   stat[0]=mean, stat[1]=rstd, stat[2]=scale=gamma*rstd, stat[3]=beta   (stat is [4][F])
 * and (if running_mean != NULL) update running_mean/var in place with `momentum`.
 * slabs == 0: evaluation mode - take mean/var from running_mean/running_var. */
int alignn_bn_finalize(const float* partial, int slabs, int64_t rows, int F, const float* gamma,
                       const float* beta, float eps, float momentum, float* running_mean,
                       float* running_var, float* stat, alignn_stream_t stream);

/* Y[r,f] = (R ? R[r,f] : 0) + silu((X[r,f]-mean[f])*scale[f] + beta[f])
 * amax (here and below; may be NULL): device scalar the kernel raises to max|output| with one atomicMax per
 * workgroup - the caller zeroes it first; several kernels writing parts of one tensor may share it.  It is what
 * alignn_gemm_nt_f16x3 / alignn_gemm_tn need to run the next projection at half the matrix-core work. */
int alignn_bn_silu_fwd(const float* X, int64_t ldx, const float* R, int64_t ldr, const float* stat,
                       float* Y, int64_t ldy, int64_t rows, int F, float* amax, alignn_stream_t stream);

/* backward, phase 1: partial[s][0][f] = sum_r gz, partial[s][1][f] = sum_r gz*xhat, where
 * z = (X-mean)*scale+beta, gz = GY * silu'(z), xhat = (X-mean)*rstd */
int alignn_bn_silu_bwd_reduce(const float* GY, int64_t ldgy, const float* X, int64_t ldx,
                              const float* stat, int64_t rows, int F, float* partial,
                              alignn_stream_t stream);
/* phase 1b: red[0][f]=sum gz (=dbeta), red[1][f]=sum gz*xhat (=dgamma) from the slabs */
int alignn_bn_bwd_finalize(const float* partial, int slabs, int F, float* red, alignn_stream_t stream);
/* LayerNorm flavour (ALIGNNAtomWise: alignn/models/alignn_atomwise.py:151,155; MLPLayer of
 * alignn/models/utils.py:277-292): per-ROW statistics over the F features, eps, affine gamma/beta.
 *   fwd: Y = (R ? R : 0) + silu(LayerNorm(X)); stats[r] = (mean, rstd) (may be NULL for inference)
 *   bwd: GX = LayerNorm/SiLU backward of GY; partial[alignn_ln_slabs(rows)][2][F] = slabs of
 *        (sum gz, sum gz*xhat) = (dbeta, dgamma), finished with alignn_bn_bwd_finalize. */
int alignn_ln_slabs(int64_t rows);
int alignn_ln_silu_fwd(const float* X, int64_t ldx, const float* R, int64_t ldr, const float* gamma,
                       const float* beta, float eps, float* Y, int64_t ldy, float* stats, int64_t rows, int F,
                       float* amax, alignn_stream_t stream);
int alignn_ln_silu_bwd(const float* GY, int64_t ldgy, const float* X, int64_t ldx, const float* gamma,
                       const float* beta, const float* stats, float* GX, int64_t ldgx, float* partial,
                       int64_t rows, int F, float* amax, alignn_stream_t stream);
/* alignn_ln_silu_bwd on the node pre-activations xpre = Ux + S1 / (S0 + eps) of an edge-gated convolution, the adjoints of the two
 * segment sums written in the same pass (alignn_egc_node_bwd's arithmetic on the gradient just formed: GS1 = g / (S0 + eps),
 * GS0 = -GS1 * HH) - one launch instead of two on the reverse chain of every LayerNorm-flavoured convolution. */
int alignn_ln_silu_bwd_node(const float* GY, int64_t ldgy, const float* X, int64_t ldx, const float* gamma, const float* beta,
                            const float* stats, float* GX, int64_t ldgx, float* partial, int64_t rows, int F, float* amax,
                            const float* S0, const float* HH, float* GS1, float* GS0, alignn_stream_t stream);
/* out[f] = sum_s partial[s][f] over `slabs` slabs of `width` floats (fp64 accumulation, fixed order) */
int alignn_slab_sum(const float* partial, int slabs, int width, float* out, alignn_stream_t stream);
/* Pre-pass for very many slabs (one per row tile of a T-row projection): out[g][f] = sum of partial[k][f] over
 * k = g, g + G, ... with G = alignn_slab_fold_slabs() (64); out is [G][width].  Same fixed order every time. */
int alignn_slab_fold_slabs(void);
int alignn_slab_fold(const float* partial, int slabs, int width, float* out, alignn_stream_t stream);
/* phase 2: GX = gamma*rstd*(gz - red0/rows - xhat*red1/rows)   (training-mode BatchNorm backward)
 * eval_mode != 0: GX = gz*scale (running statistics are constants) */
int alignn_bn_silu_bwd_apply(const float* GY, int64_t ldgy, const float* X, int64_t ldx,
                             const float* stat, const float* gamma, const float* red, int eval_mode,
                             float* GX, int64_t ldgx, int64_t rows, int F, float* amax,
                             alignn_stream_t stream);
/* ... for the NODE norm of a convolution (GX = gradient of x_pre = Ux + S1 / (S0 + 1e-6), alignn/models/alignn.py:110-111,
 * 122-123): the quotient's adjoints GS1 = GX / (S0 + 1e-6), GS0 = -GS1 * HH in the same pass - alignn_bn_silu_bwd_apply
 * followed by alignn_egc_node_bwd, same bits, one launch.  S0, HH, GS1, GS0: [rows, F] contiguous. */
int alignn_bn_silu_bwd_apply_node(const float* GY, int64_t ldgy, const float* X, int64_t ldx, const float* stat,
                                  const float* gamma, const float* red, int eval_mode, float* GX, int64_t ldgx,
                                  int64_t rows, int F, float* amax, const float* S0, const float* HH, float* GS1, float* GS0,
                                  alignn_stream_t stream);
/* ... and, in the same pass, the column sums of GX - the bias gradient of the nn.Linear in front of the BatchNorm
 * (MLPLayer, alignn/models/alignn.py:170-184) - as [alignn_col_stats_slabs(rows)][F] slabs for alignn_slab_sum. */
int alignn_bn_silu_bwd_apply_sum(const float* GY, int64_t ldgy, const float* X, int64_t ldx, const float* stat,
                                 const float* red, int eval_mode, float* GX, int64_t ldgx, int64_t rows, int F,
                                 float* amax, float* partial, alignn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Edge-gated graph convolution core (gather -> gate -> segment-sum), one wavefront per
 * destination segment.  Replaces, in EdgeGatedGraphConv.forward (alignn/models/alignn.py:78-129):
 *   g.apply_edges(fn.u_add_v("e_src","e_dst","e_nodes"))            :100   (DGL g-SDDMM)
 *   m = e_nodes + edge_gate(edge_feats); sigma = sigmoid(m)           :101-103
 *   g.update_all(fn.u_mul_e("Bh","sigma","m"), fn.sum(...))           :105-107 (DGL g-SpMM)
 *   g.update_all(fn.copy_e("sigma","m"), fn.sum("m","sum_sigma"))     :108
 *   h = sum_sigma_h / (sum_sigma + 1e-6); x = src_update(x) + h       :109-110
 * and DGL's autograd of those calls (backward).
 *
 * P is the fused node projection [n, 4H] = [A | Bd | Bh | Ux] = x * [W_sg;W_dg;W_du;W_su]^T + b.
 * ------------------------------------------------------------------------------------------ */

/* number of stat slabs the gate kernel writes for a graph with n segments (pivot-slab layout: slabs*(3*H+1) floats, see
 * alignn_col_stats_welford) */
int alignn_egc_slabs(int64_t n_seg);

/* In : P[n,4H]; M[m,H] holds C = y*W_eg^T + b on entry.
 * Out: M[m,H] = m_pre = A[src] + Bd[dst] + C (in place); XPRE[n,H] = Ux + S1/(S0+1e-6);
 *      S0[n,H], HH[n,H] = S1/(S0+1e-6) (saved for backward; may be NULL for inference);
 *      e_partial: Welford column-stat slabs of m_pre; n_partial: of XPRE (each may be NULL; alignn_bn_finalize_welford).
 * m_rows = number of edge rows of M (only decides whether M is streamed with read-once/write-once hints). */
int alignn_egc_gate_fwd(const float* P, float* M, const int32_t* seg_ptr, const int32_t* seg_node,
                        const int32_t* src, int64_t n_seg, int64_t m_rows, int H, float* XPRE,
                        float* S0, float* HH, float* e_partial, float* n_partial,
                        alignn_stream_t stream);

/* Inference form of the same pass (no backward, edge normalisation = the fixed affine map of BatchNorm in eval
 * mode, e_stat = [mean, rstd, scale, shift] from the running statistics): C is only read, m_pre is never written,
 * and the edge output YOUT = (Y ? Y : 0) + silu((m_pre - mean) * scale + shift) comes straight out of the gate pass
 * (YOUT == NULL: the edge output is dead; y_amax, optional: raised to max|YOUT|).  XPRE as above. */
int alignn_egc_gate_infer(const float* P, const float* C, const int32_t* seg_ptr, const int32_t* seg_node,
                          const int32_t* src, int64_t n_seg, int64_t m_rows, int H, float* XPRE,
                          const float* e_stat, const float* Y, float* YOUT, float* y_amax,
                          alignn_stream_t stream);

/* Node-side backward glue: from g_xpre (BatchNorm-backward output on nodes) produce
 * GS1 = g_xpre/(S0+eps), GS0 = -g_xpre*HH/(S0+eps).  GXPRE has leading dimension ldg (it usually
 * lives in the Ux block of the [n,4H] projection gradient). */
int alignn_egc_node_bwd(const float* GXPRE, int64_t ldg, const float* S0, const float* HH, float* GS1, float* GS0,
                        int64_t n_nodes, int H, alignn_stream_t stream);

/* Destination-ordered backward pass.  Per slot e (dst i, src u):
 *   g_mbn = BatchNorm/SiLU backward of the edge branch (skipped when GY == NULL: dead edge output;
 *           when e_stat == NULL, GY is taken to BE g_mbn already - the LayerNorm flavour computes it
 *           with alignn_ln_silu_bwd)
 *   sigma = sigmoid(M[e]);  g_sigma = GS1[i]*Bh[u] + GS0[i]
 *   GM[e] = g_mbn + g_sigma*sigma*(1-sigma);   GP[i, H:2H] (g_Bd) = sum_e GM[e]
 * GP is the [n,4H] gradient of the fused node projection; this pass fills its Bd block.
 * gb_partial (optional): [alignn_egc_slabs(n_seg)][H] column-sum slabs of GM (the edge_gate bias gradient,
 * finished with alignn_slab_sum) - saves a separate pass over GM.
 * gm_amax / gp_amax (optional): raised to max|GM| and max|the GP entries written here| (see alignn_bn_silu_fwd). */
int alignn_egc_bwd_dst(const float* GY, const float* M, const float* P, const float* GS1,
                       const float* GS0, const float* e_stat, const float* e_gamma,
                       const float* e_red, int e_eval, int64_t m_rows,
                       const int32_t* seg_ptr, const int32_t* seg_node, const int32_t* src,
                       int64_t n_seg, int H, float* GM, float* GP, float* gb_partial,
                       float* gm_amax, float* gp_amax, alignn_stream_t stream);

/* Line-graph backward with the destination- and source-ordered passes fused (one workgroup per centre atom j;
 * L(g)'s edges form one dense block per atom: sources = in-edges of j = L(g) nodes [grp_src_ptr[j],
 * grp_src_ptr[j+1]), segments = out-edges of j = segment ranks [grp_seg_ptr[j], grp_seg_ptr[j+1])).  Same inputs,
 * outputs and bit-identical results as alignn_egc_bwd_dst followed by alignn_egc_bwd_src, with 2 reads + 1 write
 * per edge row instead of 4 + 1.  gb_partial: [n_groups][H] column-sum slabs of GM. */
int alignn_egc_bwd_lg_fused(const float* GY, const float* M, const float* P, const float* GS1,
                            const float* GS0, const float* e_stat, const float* e_red, int e_eval,
                            int64_t m_rows, const int32_t* grp_seg_ptr, const int32_t* grp_src_ptr,
                            int64_t n_groups, const int32_t* seg_ptr, const int32_t* seg_node,
                            const int32_t* dst, const int32_t* out_ptr, const int32_t* out_slot, int H,
                            float* GM, float* GP, float* gb_partial, float* gm_amax, float* gp_amax,
                            alignn_stream_t stream);

/* The same backward for line graphs whose blocks are DENSE and source-sorted (every segment of atom j lists all
 * in-edges of j in ascending order, minus itself for a self-image bond - true for DGL's g.line_graph() after
 * alignn_amd.graph's canonicalisation and for graphs built by graph.line_graph_of; max_group_src = the largest
 * in-degree, > 0 - atoms with more than 16 in-edges take extra passes of 16 sources).  Rows are addressed by index arithmetic, the block is walked segment by segment, every T-row of
 * GY and M is read once and GM written once (the by-source kernel above re-reads GM: 3.6 row passes of fetch vs 2).
 * Same outputs; sums in a different but fixed order. */
int alignn_egc_bwd_lg_dense_supported(int max_group_src);
int alignn_egc_bwd_lg_dense(const float* GY, const float* M, const float* P, const float* GS1,
                            const float* GS0, const float* e_stat, const float* e_red, int e_eval,
                            int64_t m_rows, const int32_t* grp_seg_ptr, const int32_t* grp_src_ptr,
                            int64_t n_groups, int max_group_src, const int32_t* seg_ptr,
                            const int32_t* seg_node, int H, float* GM, float* GP, float* gb_partial,
                            float* gm_amax, float* gp_amax, alignn_stream_t stream);

/* Source-ordered backward pass (deterministic scatter-by-source):
 *   GP[j, 0:H]   (g_A)  = sum_{e: src e = j} GM[e]
 *   GP[j, 2H:3H] (g_Bh) = sum_{e: src e = j} sigmoid(M[e]) * GS1[dst e] */
int alignn_egc_bwd_src(const float* GM, const float* M, const float* GS1, const int32_t* out_ptr,
                       const int32_t* out_slot, const int32_t* dst, int64_t n_nodes, int H,
                       float* GP, float* gp_amax, alignn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Featurisation and readout.
 * ------------------------------------------------------------------------------------------ */

/* RBFExpansion.forward (alignn/models/utils.py:40-44): out[r,k] = exp(-gamma (d[r]-c[k])^2) */
int alignn_rbf_fwd(const float* d, const float* centers, float gamma, float* out, int64_t rows,
                   int bins, alignn_stream_t stream);
/* compute_bond_cosines on L(g) (alignn/graphs.py:847-864; re-run inside ALIGNNAtomWise.forward when
 * lg_on_fly, alignn_atomwise.py:424-431): h[k] = clamp(-r[e1[k]].r[e2[k]] / (|r[e1]| |r[e2]|), -1, 1) */
int alignn_bond_cosine_fwd(const float* r, const int32_t* e1, const int32_t* e2, float* h, int64_t T,
                           alignn_stream_t stream);
/* torch.norm(r, dim=1) (alignn/models/alignn.py:313): out[r] = ||v[r,0:3]|| */
int alignn_norm3_fwd(const float* v, float* out, int64_t rows, alignn_stream_t stream);
/* First derivatives of the three featurisation steps w.r.t. the geometry - what torch autograd does for the
 * reference when ALIGNNAtomWise takes forces = -dE/dr (alignn_atomwise.py:512-539):
 *   rbf:    gd[r] = sum_k G[r,k] * d/dd exp(-gamma (d[r]-c[k])^2)
 *   norm3:  gv[r,:] = g[r] * v[r,:]/|v[r]|
 *   cosine: per triplet k, ga[k,:] = gh[k] dh/dr[e1[k]], gb[k,:] = gh[k] dh/dr[e2[k]] (zero where the clamp is
 *           active); the caller segment-sums ga by source and gb by destination of L(g) (alignn_segment_sum). */
int alignn_rbf_bwd(const float* d, const float* centers, float gamma, const float* G, float* gd, int64_t rows,
                   int bins, alignn_stream_t stream);
int alignn_norm3_bwd(const float* v, const float* g, float* gv, int64_t rows, alignn_stream_t stream);
int alignn_bond_cosine_bwd(const float* r, const int32_t* e1, const int32_t* e2, const float* gh, float* ga,
                           float* gb, int64_t T, alignn_stream_t stream);
/* dgl.nn.AvgPooling (alignn/models/alignn.py:325): out[b] = mean_{i in graph b} x[i]; graph_ptr[B+1] */
int alignn_segment_mean_fwd(const float* X, const int32_t* graph_ptr, float* out, int B, int H,
                            alignn_stream_t stream);
/* its backward: GX[i] = G[b(i)] / count[b(i)] */
int alignn_segment_mean_bwd(const float* G, const int32_t* graph_ptr, float* GX, int B, int H,
                            alignn_stream_t stream);
/* Generic segment sum (any width F): out[node ? node[s] : s, :] = sum_{k in [ptr[s], ptr[s+1])}
 * vals[slot ? slot[k] : k, :].  With alignn_gather_rows it forms the adjoint pair (gather by index <-> sum by
 * group) from which the twice-differentiable force path is composed: DGL copy_e/sum on g and on dgl.reverse(g)
 * (alignn/models/alignn_atomwise.py:547-565) and every gather/scatter of the conv when create_graph=True. */
int alignn_segment_sum(const float* vals, int64_t ldv, const int32_t* ptr, const int32_t* slot,
                       const int32_t* node, float* out, int64_t ldo, int64_t n_seg, int F,
                       alignn_stream_t stream);
/* row permutation: out[k] = in[perm[k]] (canonical reordering of edge features) */
int alignn_gather_rows(const float* in, const int32_t* perm, float* out, int64_t rows, int F,
                       alignn_stream_t stream);

/* alignn_ln_silu_dual_bwd on the node pre-activations of an edge-gated convolution + alignn_egc_node_dual_bwd in the same pass */
int alignn_ln_silu_dual_bwd_node(const float* GY, const float* GYt, int64_t ldg, const float* X, const float* Xt, int64_t ldx,
                                 const float* gamma, const float* beta, const float* stats, float* GX, float* GXt, int64_t ldo,
                                 float* partial, int64_t rows, int F, float* amax2, const float* s0, const float* hh,
                                 const float* s0t, const float* hht, float* q1, float* q0, float* q1t, float* q0t,
                                 alignn_stream_t stream);
/* ------------------------------------------------------------------------------------------
 * Dual-number kernels (value + directional derivative along a bond-vector displacement) of the LayerNorm-flavoured
 * stack - training THROUGH the forces: the reference takes pair forces with autograd.grad(create_graph=True)
 * (alignn/models/alignn_atomwise.py:512-565) and differentiates the force / stress loss through them
 * (alignn/train.py:387).  sum_e w_e . f_e = -D_w E_tot, so the loss gradient is the reverse pass of a forward pass
 * that carries tangents; these are its non-linear pieces (csrc/dual.hip), "t" = tangent, amax2 = two device scalars
 * (value, tangent) raised like `amax` elsewhere.  Slab counts: alignn_dual_slabs(rows).
 *   ln_silu_dual_fwd:  Y = R + silu(LN(X)), Yt = Rt + d[silu o LN](X).Xt;  stats[rows][2] = mean, rstd
 *   ln_silu_dual_bwd:  (GY, GYt) -> (GX, GXt); partial[slabs][2][F] = dbeta | dgamma partial sums
 *   egc_gate_dual_fwd: DGL u_add_v + sigmoid + u_mul_e/sum + copy_e/sum + h = S1/(S0+eps) (alignn_atomwise.py:177-189)
 *                      on duals; M / Mt hold C / Ct on entry and m / mt on exit
 *   egc_node_dual_bwd, egc_dual_bwd_dst, egc_dual_bwd_src: its reverse (adjoints Q of the four segment sums; GM / GMt;
 *                      the A | Bd | Bh blocks of GP / GPt; gb_partial[slabs][H] = column sums of GM) */
int alignn_dual_slabs(int64_t rows);
int alignn_ln_silu_dual_fwd(const float* X, const float* Xt, int64_t ldx, const float* R, const float* Rt, int64_t ldr,
                            const float* gamma, const float* beta, float eps, float* Y, float* Yt, int64_t ldy,
                            float* stats, int64_t rows, int F, float* amax2, alignn_stream_t stream);
int alignn_ln_silu_dual_bwd(const float* GY, const float* GYt, int64_t ldg, const float* X, const float* Xt, int64_t ldx,
                            const float* gamma, const float* beta, const float* stats, float* GX, float* GXt,
                            int64_t ldo, float* partial, int64_t rows, int F, float* amax2, alignn_stream_t stream);
int alignn_egc_gate_dual_fwd(const float* P, const float* Pt, float* M, float* Mt, const int32_t* seg_ptr,
                             const int32_t* seg_node, const int32_t* src, int64_t n, int64_t m, int H, float* xpre,
                             float* xpre_t, float* s0, float* hh, float* s0t, float* hht, alignn_stream_t stream);
/* the tangent half of alignn_egc_gate_dual_fwd alone, for a caller that already holds the values (M = m, s0, hh from the force
 * evaluation): Mt holds Ct on entry and mt on exit; writes xpre_t, s0t, hht.  alignn_ln_silu_dual_fwd likewise takes Y == NULL
 * (value output not written, R not read). */
int alignn_egc_gate_dual_fwd_tangent(const float* P, const float* Pt, const float* M, float* Mt, const int32_t* seg_ptr,
                                     const int32_t* seg_node, const int32_t* src, int64_t n, int64_t m, int H, float* xpre_t,
                                     const float* s0, const float* hh, float* s0t, float* hht, alignn_stream_t stream);
int alignn_egc_node_dual_bwd(const float* g, const float* gt, int64_t ldg, const float* s0, const float* hh,
                             const float* s0t, const float* hht, float* q1, float* q0, float* q1t, float* q0t, int64_t n,
                             int H, alignn_stream_t stream);
int alignn_egc_dual_bwd_dst(const float* GL, const float* GLt, const float* M, const float* Mt, const float* P,
                            const float* Pt, const float* q1, const float* q0, const float* q1t, const float* q0t,
                            const int32_t* seg_ptr, const int32_t* seg_node, const int32_t* src, int64_t n, int H,
                            float* GM, float* GMt, float* GP, float* GPt, float* gb_partial, float* gm_amax2,
                            float* gp_amax2, alignn_stream_t stream);
int alignn_egc_dual_bwd_src(const float* GM, const float* GMt, const float* M, const float* Mt, const float* q1,
                            const float* q1t, const int32_t* out_ptr, const int32_t* out_slot, const int32_t* dst,
                            int64_t n, int H, float* GP, float* GPt, float* gp_amax2, alignn_stream_t stream);
/* egc_dual_bwd_dst + egc_dual_bwd_src as ONE pass for line graphs with dense, source-sorted blocks (see
 * alignn_egc_bwd_lg_dense: one workgroup per centre atom, rows by index arithmetic): M, Mt, GL, GLt read once, GM, GMt
 * written once - 6 T-row passes instead of 10; same outputs, sums in a different but fixed order; gb_partial
 * [n_groups][H]. */
int alignn_egc_dual_bwd_lg_dense(const float* GL, const float* GLt, const float* M, const float* Mt, const float* P,
                                 const float* Pt, const float* q1, const float* q0, const float* q1t, const float* q0t,
                                 int64_t m_rows, const int32_t* grp_seg_ptr, const int32_t* grp_src_ptr, int64_t n_groups,
                                 const int32_t* seg_ptr, const int32_t* seg_node, int H, float* GM, float* GMt, float* GP,
                                 float* GPt, float* gb_partial, float* gm_amax2, float* gp_amax2, alignn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Line-graph convolutions of the LayerNorm flavour with the edge LayerNorm inside the gate passes (csrc/convln.hip).
 * Reference: EdgeGatedGraphConv.forward with nn.LayerNorm, alignn/models/alignn_atomwise.py:151-208 (:199-206 the two
 * normalised branches), and its autograd / double backward for the reverse passes.  A wavefront of the gate passes holds a
 * whole H <= 256 row of m, so the row statistics are wave reductions and y' = y + silu(LN(m)) (forward) resp. the LayerNorm /
 * SiLU adjoint of g_y' (reverse) cost no pass of their own.  Same formulas and summation orders as the separate kernels
 * (alignn_ln_silu_fwd / _bwd, alignn_ln_silu_dual_fwd / _bwd + the gate passes); LayerNorm parameter gradients leave as
 * ln_partial [n_groups][2][H] (dbeta | dgamma per workgroup; alignn_bn_bwd_finalize sums the slabs in order).
 * e_stat [m][2] = (mean, rstd) per row, written by the forward, read by everything else.
 * alignn_egc_ln_fused_supported(H, m_rows): 1 when they are the ones to take (H % 4 == 0, H <= 256, edge tensors of >= 64 MiB:
 * on cache-resident line graphs the separate kernels' finer grids win; ALIGNN_AMD_LN_FUSED = 0 never, 2 whenever H allows). */
int alignn_egc_ln_fused_supported(int H, int64_t m_rows);
/* alignn_egc_gate_fwd_pre (M holds m = A[u] + Bd[v] + C) + alignn_ln_silu_fwd(M, residual Y or NULL) -> YOUT, e_stat, y_amax */
int alignn_egc_gate_fwd_pre_ln(const float* P, const float* M, const int32_t* seg_ptr, const int32_t* seg_node, const int32_t* src,
                               int64_t n_seg, int64_t m_rows, int H, float* XPRE, float* S0, float* HH, const float* gamma,
                               const float* beta, float eps, const float* Y, float* YOUT, float* e_stat, float* y_amax,
                               alignn_stream_t stream);
/* alignn_ln_silu_bwd(GY, M) + alignn_egc_bwd_lg_dense(MODE 2): GY is the adjoint of the edge OUTPUT (before the LayerNorm) */
int alignn_egc_bwd_lg_dense_ln(const float* GY, const float* M, const float* P, const float* GS1, const float* GS0,
                               const float* gamma, const float* beta, const float* e_stat, int64_t m_rows,
                               const int32_t* grp_seg_ptr, const int32_t* grp_src_ptr, int64_t n_groups, int max_group_src,
                               const int32_t* seg_ptr, const int32_t* seg_node, int H, float* GM, float* GP, float* gb_partial,
                               float* ln_partial, float* gm_amax, float* gp_amax, alignn_stream_t stream);
/* The same for graphs without dense blocks (the bond graph g): the destination-ordered reverse halves with the LayerNorm adjoint
 * of the edge output formed in the pass - alignn_ln_silu_bwd + alignn_egc_bwd_dst(e_stat = NULL), resp. alignn_ln_silu_dual_bwd +
 * alignn_egc_dual_bwd_dst, as one launch; the source-ordered halves follow unchanged.  gb_partial [alignn_egc_ln_dst_slabs(n_seg)][H],
 * ln_partial [...][2][H].  alignn_egc_ln_dst_supported(H): H % 4 == 0, H <= 256, ALIGNN_AMD_LN_FUSED != 0 (any row count). */
int alignn_egc_ln_dst_slabs(int64_t n_seg);
int alignn_egc_ln_dst_supported(int H);
int alignn_egc_bwd_dst_ln(const float* GY, const float* M, const float* P, const float* GS1, const float* GS0, const float* gamma,
                          const float* beta, const float* e_stat, const int32_t* seg_ptr, const int32_t* seg_node,
                          const int32_t* src, int64_t n_seg, int H, float* GM, float* GP, float* gb_partial, float* ln_partial,
                          float* gm_amax, float* gp_amax, alignn_stream_t stream);
int alignn_egc_dual_bwd_dst_ln(const float* GY, const float* GYt, const float* M, const float* Mt, const float* P, const float* Pt,
                               const float* q1, const float* q0, const float* q1t, const float* q0t, const float* gamma,
                               const float* beta, const float* e_stat, const int32_t* seg_ptr, const int32_t* seg_node,
                               const int32_t* src, int64_t n_seg, int H, float* GM, float* GMt, float* GP, float* GPt,
                               float* gb_partial, float* ln_partial, float* gm_amax2, float* gp_amax2, alignn_stream_t stream);
/* alignn_egc_gate_dual_fwd_tangent + the tangent of y' = y + silu(LN(m)) (alignn_ln_silu_dual_fwd with Y == NULL): Mt holds Ct
 * on entry and mt on exit, Yt = Rt + d[silu o LN](m) . mt; amax2[1] is raised to max|Yt| */
int alignn_egc_gate_dual_tan_ln(const float* P, const float* Pt, const float* M, float* Mt, const int32_t* seg_ptr,
                                const int32_t* seg_node, const int32_t* src, int64_t n, int64_t m, int H, float* xpre_t,
                                const float* s0, const float* hh, float* s0t, float* hht, const float* gamma, const float* beta,
                                const float* e_stat, const float* Rt, float* Yt, float* amax2, alignn_stream_t stream);
/* alignn_ln_silu_dual_bwd(GY, GYt, M, Mt) + alignn_egc_dual_bwd_lg_dense: GY / GYt are the adjoints of the edge output's value
 * and tangent */
int alignn_egc_dual_bwd_lg_dense_ln(const float* GY, const float* GYt, const float* M, const float* Mt, const float* P,
                                    const float* Pt, const float* q1, const float* q0, const float* q1t, const float* q0t,
                                    const float* gamma, const float* beta, const float* e_stat, int64_t m_rows,
                                    const int32_t* grp_seg_ptr, const int32_t* grp_src_ptr, int64_t n_groups,
                                    const int32_t* seg_ptr, const int32_t* seg_node, int H, float* GM, float* GMt, float* GP,
                                    float* GPt, float* gb_partial, float* ln_partial, float* gm_amax2, float* gp_amax2,
                                    alignn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Periodic k-nearest-neighbour bond lists on the device (csrc/knn.hip).
 * Replace alignn/graphs.py:155-264 (nearest_neighbor_edges on jarvis' get_all_neighbors, canonize_edge :128-153,
 * build_undirected_edgedata :230-264), which alignn/ff/calculators.py:280-291 re-runs at every MD step.
 * One wavefront per site; float64 distances evaluated in the reference's operation order (bit-exact tie decisions);
 * no atomics on the output, no host synchronisation.  Inputs shared by the four passes:
 *   lat[B][9] float64 (rows a, b, c), cart[N][3] float64 Cartesian positions (frac . lat, the caller's fixed-order
 *   product), graph_ptr[B+1] site offsets, site_graph[N] crystal of each site, cut[B][levels] the crystal's cutoff
 *   sequence (cutoff, then "longest lattice vector if below it, else twice", graphs.py:170-188), reach[B][levels][3]
 *   image-box half-widths ceil(cut / plane spacing).
 *   knn_levels: crystal_level[B] (zeroed by the caller) = first level at which EVERY site of the crystal has >= k
 *               candidates (== levels: not reachable - the caller raises)
 *   knn_kth:    kth[N] = distance of the k-th nearest candidate (ties share it)
 *   knn_count:  count[N] = canonical bonds (a, b >= a, image) owned by site a: kept by a OR by b
 *   knn_emit:   offset[N] = exclusive prefix sum of count; writes both directions of bond e at rows 2e, 2e+1 of
 *               u, v (int64), r[.][3] float32 = src -> dst displacement, image[.][3] int32 (optional; forward image for
 *               both directions, like the reference's `images`); a site's bonds leave sorted by (b, image). */
int alignn_knn_levels(const double* lat, const double* cart, const int32_t* graph_ptr, const int32_t* site_graph,
                      const double* cut, const int32_t* reach, int levels, int k, int64_t n_sites, int32_t* crystal_level,
                      alignn_stream_t stream);
int alignn_knn_kth(const double* lat, const double* cart, const int32_t* graph_ptr, const int32_t* site_graph,
                   const double* cut, const int32_t* reach, int levels, int k, int64_t n_sites, const int32_t* crystal_level,
                   double* kth, alignn_stream_t stream);
int alignn_knn_count(const double* lat, const double* cart, const int32_t* graph_ptr, const int32_t* site_graph,
                     const double* cut, const int32_t* reach, int levels, int64_t n_sites, const int32_t* crystal_level,
                     const double* kth, int64_t* count, alignn_stream_t stream);
int alignn_knn_emit(const double* lat, const double* cart, const int32_t* graph_ptr, const int32_t* site_graph,
                    const double* cut, const int32_t* reach, int levels, int64_t n_sites, const int32_t* crystal_level,
                    const double* kth, const int64_t* offset, int64_t* u, int64_t* v, float* r, int32_t* image,
                    alignn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Composite entry points (csrc/composite.hip): ONE call = one EdgeGatedGraphConv forward / backward, BatchNorm
 * flavour in training mode (alignn/models/alignn.py:78-129 and torch.autograd's backward of it).  They issue exactly
 * the launches the per-kernel entry points above would, in the same order with the same arguments (bit-identical
 * results), so that an eagerly launched training step costs ~3 host calls per convolution instead of ~30.
 * Every pointer is a device pointer owned by the caller; `scratch` is one block of alignn_egc_conv_*_scratch() bytes.
 * Shapes: x [n,Kin], y [m,Kin], P [n,4H], M [m,H], everything else [n,H] / [m,H] / [4,H] / [2,H] as named.
 *   node_kind  0: P by alignn_gemm_nt(wcat)            1: alignn_gemm_nt_f16x3(wcat_img, x_amax)
 *   edge_kind  0: alignn_gemm_nt(w_eg) + alignn_egc_gate_fwd + alignn_bn_silu_fwd
 *              1: alignn_gemm_nt_f16x3_gather(+statistics) + alignn_egc_gate_fwd_pre_norm
 *   y_out == NULL: dead edge output (statistics / running buffers still updated)
 *   gate_mode  0: alignn_egc_bwd_dst + _src   1: alignn_egc_bwd_lg_fused   2: alignn_egc_bwd_lg_dense
 *   dx_kind / dy_kind (input gradients g_x = GP wcat, g_y = GM w_eg):
 *              1: f16x3 on the W^T image (dy: + BatchNorm-backward sums of (src_xn, src_nstat) -> src_red)
 *              2: alignn_gemm_nn_split (dx only)   3: alignn_gemm_nn   4: alignn_gemm_nt on a transposed copy (wcat_t / weg_t)
 *              5: f16x3 on the W^T image, no sums (dy only)
 * ------------------------------------------------------------------------------------------ */
typedef struct alignn_egc_fwd_args {
    const int32_t *seg_ptr, *seg_node, *src, *dst;
    const int32_t* seg_rank; /* edge_kind 2: [m] segment of every edge row (the destination term is then read from a
                                segment-ordered copy of P's Bd block, alignn_gather_rows_ld + alignn_gemm_nt_f16x3_gather2) */
    int64_t n, m;
    int32_t H, Kin, node_kind, edge_kind, residual, pad_;
    float eps, momentum;
    const float *x, *y, *x_amax, *y_amax;
    const float *wcat, *bcat, *wcat_amax;
    const void* wcat_img;
    const float *w_eg, *b_eg, *weg_amax;
    const void* weg_img;
    const float *n_gamma, *n_beta, *e_gamma, *e_beta;
    float *n_rm, *n_rv, *e_rm, *e_rv;
    float *P, *M, *xpre, *s0, *hh, *n_stat, *e_stat, *x_out, *y_out, *x_out_amax, *y_out_amax;
    float* scratch;
    size_t scratch_bytes;
} alignn_egc_fwd_args;
size_t alignn_egc_conv_fwd_scratch(int64_t n, int64_t m, int H, int Kin, int edge_kind);
int alignn_egc_conv_fwd(const alignn_egc_fwd_args* args, alignn_stream_t stream);

typedef struct alignn_egc_bwd_args {
    const int32_t *seg_ptr, *seg_node, *src, *dst, *out_ptr, *out_slot, *grp_seg_ptr, *grp_src_ptr;
    int64_t n, m, n_groups;
    int32_t H, Kin, gate_mode, dense_max_src, dx_kind, dy_kind, residual, pad_;
    const float *gx_out, *gy_out;                                       /* incoming gradients (gy_out may be NULL) */
    const float *P, *M, *xpre, *s0, *hh, *n_stat, *e_stat, *n_gamma, *e_gamma; /* saved by the forward */
    const float* e_red_in;                                              /* pre-reduced sums of the edge norm, or NULL */
    const float *wcat, *wcat_t, *wcat_amax;
    const void* wcat_t_img;
    const float *w_eg, *weg_t, *weg_amax;
    const void* weg_t_img;
    const float *src_xn, *src_nstat;                                    /* dy_kind 1: the norm y came out of */
    int64_t src_ldxn;
    float *GP, *GM, *gs1, *gs0, *n_red, *e_red, *gb_part, *g_x, *g_y, *src_red, *gp_amax, *gm_amax;
    alignn_stream_t aux_stream; /* optional second stream: the node input gradient g_x = GP wcat runs there, beside
                                   g_y = GM w_eg on the caller's stream (fork after the gate backward, join before the call
                                   returns; needs alignn_fork_events_init() to have run on this device - otherwise, or with
                                   NULL, everything stays on the one stream).  Same kernels, same results. */
    float* scratch;
    size_t scratch_bytes;
} alignn_egc_bwd_args;
/* creates (once per device, for the CURRENT device) the two events the composite entry points fork / join with; call it
 * outside stream capture.  Returns 0, or the HIP error. */
int alignn_fork_events_init(void);
size_t alignn_egc_conv_bwd_scratch(int64_t n, int64_t m, int H, int Kin, int dx_kind, int dy_kind);
int alignn_egc_conv_bwd(const alignn_egc_bwd_args* args, alignn_stream_t stream);

/* the weight / bias gradients of the same convolution (the host code runs this one on its side stream):
 * g_beg = column sums of GM (from gb_part), g_weg = GM^T y, g_wcat = GP^T x, g_bcat = column sums of GP */
typedef struct alignn_egc_wgrad_args {
    int64_t n, m;
    int32_t H, Kin, gb_slabs, pad_;
    const float *GM, *GP, *x, *y, *gb_part, *gm_amax, *gp_amax, *x_amax, *y_amax;
    float *g_weg, *g_beg, *g_wcat, *g_bcat;
    float* scratch;
    size_t scratch_bytes;
} alignn_egc_wgrad_args;
size_t alignn_egc_conv_wgrad_scratch(int64_t n, int64_t m, int H, int Kin);
int alignn_egc_conv_wgrad(const alignn_egc_wgrad_args* args, alignn_stream_t stream);
/* sizeof the three argument structs (0: fwd, 1: bwd, 2: wgrad) - a binding checks its own packing against these */
size_t alignn_egc_args_sizeof(int which);

/* ------------------------------------------------------------------------------------------
 * Whole-model entry points (csrc/model.hip): ONE call = ALIGNN.forward in training mode
 * (alignn/models/alignn.py:282-349: RBF + MLPLayer embeddings :201-222, ALIGNNConv x alignn_layers :132-167,
 * EdgeGatedGraphConv x gcn_layers :48-129, AvgPooling + fc :325,341) or torch.autograd's backward of it - what the
 * reference's per-batch loop (alignn/train.py:258-270: a NEW (g, lg) every iteration, so nothing can be replayed) pays
 * per step.  The same kernels with the same arguments as the per-operator entry points above (bit-identical results),
 * issued from C over ONE caller-owned workspace whose layout is a pure function of the model dimensions and (N, E, T).
 *   alignn_model_init:  once per device, outside stream capture - the event pool the helper streams are ordered with
 *   alignn_model_plan:  bytes of workspace the forward / forward + backward of this batch need; hipErrorNotSupported
 *                       (801) when a kernel choice is not carried here (the caller then sequences the operators itself)
 *   alignn_model_fwd:   out[B, out_features] = fc(pool(...)); running statistics and num_batches_tracked updated
 *   alignn_model_bwd:   parameter gradients into the g_* / *red pointers of the parameter blocks; the workspace must be
 *                       the one the forward of the same (desc, batch) filled
 * Gradient layout: `red` / `n_red` / `e_red` = [2, F] = (dbeta | dgamma) of the norm; g_wcat [4H, H] = the four node
 * projections' weight gradients as row blocks (src_gate, dst_gate, dst_update, src_update), g_bcat [4H] likewise.
 * Streams (desc): lane_T / side / aux optional helper streams (NULL: the caller's); every one is joined back into
 * `stream` before a call returns.
 * ------------------------------------------------------------------------------------------ */
typedef struct alignn_mlp_params {
    const float *W, *b, *gamma, *beta;         /* Linear [out, in] / [out], BatchNorm1d affine */
    float *rm, *rv;                            /* running_mean / running_var (updated) */
    float *gW, *gb, *red;                      /* gradients: [out, in], [out], [2, out] = dbeta | dgamma */
    const void *img, *img_t;                   /* f16x2 slice images of W / W^T (alignn_prepare_weights) or NULL */
    const float* w_amax;
    int32_t in, out;
} alignn_mlp_params;

typedef struct alignn_conv_params {
    const float *wcat, *bcat, *w_eg, *b_eg;    /* fused node projection [4H, H] / [4H]; edge_gate [H, H] / [H] */
    const float *n_gamma, *n_beta, *e_gamma, *e_beta;
    float *n_rm, *n_rv, *e_rm, *e_rv;
    float *g_wcat, *g_bcat, *g_weg, *g_beg, *n_red, *e_red;
    const void *wcat_img, *wcat_img_t, *weg_img, *weg_img_t;
    const float *wcat_amax, *weg_amax;
} alignn_conv_params;

typedef struct alignn_graph_csr {              /* canonical CSR (see the conventions at the top of this file) */
    const int32_t *seg_ptr, *seg_node, *src, *dst, *out_ptr, *out_slot;
    const int32_t *grp_seg_ptr, *grp_src_ptr, *seg_rank; /* line graphs: dense blocks per centre atom; segment of every row */
    int64_t n, m, n_groups;
    int32_t dense_max_src, pad_;
} alignn_graph_csr;

typedef struct alignn_model_batch {
    alignn_graph_csr g, lg;
    const int32_t* graph_ptr;                  /* [B + 1] atom offsets per crystal */
    const float *atom_features, *r, *h;        /* [N, atom_in], [E, 3] bond vectors, [T] bond-angle cosines (canonical order) */
    int32_t B, pad_;
} alignn_model_batch;

typedef struct alignn_model_desc {
    int32_t alignn_layers, gcn_layers, H, out_features, atom_in, edge_bins, angle_bins, embed;
    float edge_gamma, angle_gamma, eps, momentum;
    const float *edge_centers, *angle_centers;
    alignn_mlp_params atom, edge1, edge2, angle1, angle2;
    const alignn_conv_params* convs;           /* HOST array: (node_update, edge_update) x alignn_layers, then gcn_layers */
    const float *fc_W, *fc_b;
    float *g_fc_W, *g_fc_b;
    const void* weight_descs;                  /* alignn_prepare_weights' device table (n_weights records) or NULL */
    float* weight_amax;
    const void* bump_ptrs;                     /* device array of n_bump int64_t* (num_batches_tracked), each += 1 per forward */
    int32_t n_weights, n_bump;
    int32_t x6_min_tiles, bd_segment_table;    /* kernel-choice constants of the per-operator path (256, 1) */
    int32_t angle_fused, norm;                 /* 1: the angle embedding through alignn_angle_embed_fwd / _bwd where its shapes allow;
                                                  norm: 0 BatchNorm1d (ALIGNN), 1 LayerNorm (ALIGNNAtomWise: rm / rv / bump_ptrs unused) */
    int32_t reuse_tape, dw_fused;              /* reuse_tape 1: a convolution's edge input gradient is written over its own (then dead)
                                                  gate pre-activation m - one T-row buffer per line-graph convolution less; the tape
                                                  does not survive the backward (no second backward of a retained graph).
                                                  dw_fused n > 0: the edge-gate projection's input gradient and weight gradient in one
                                                  pass over g_m (csrc/gemm_dw.hip) for convolutions of >= n edge rows where
                                                  alignn_gemm_dgrad_wgrad_supported; 0: always the two launches */
    int64_t amax_min_rows, lane_min_rows, side_min_rows; /* 4096; rows from which a kernel goes to lane_T / side */
    alignn_stream_t lane_T, side, aux;
} alignn_model_desc;

int alignn_model_init(void);
size_t alignn_model_sizeof(int which); /* 0 mlp_params, 1 conv_params, 2 graph_csr, 3 model_batch, 4 model_desc */
int alignn_model_plan(const alignn_model_desc* desc, const alignn_model_batch* batch, size_t* fwd_bytes, size_t* total_bytes);
int alignn_model_fwd(const alignn_model_desc* desc, const alignn_model_batch* batch, void* workspace, size_t workspace_bytes,
                     float* out, alignn_stream_t stream);
int alignn_model_bwd(const alignn_model_desc* desc, const alignn_model_batch* batch, void* workspace, size_t workspace_bytes,
                     const float* g_out, alignn_stream_t stream);
/* ALIGNN.forward in eval mode without autograd (alignn/pretrained.py; model.eval() under torch.no_grad()): BatchNorm = the
 * affine map of its running statistics, edge outputs straight from alignn_egc_gate_infer, nothing kept.  _workspace: bytes
 * for this (desc, batch), 0 when a kernel choice is not carried here.  Uses desc->lane_T only. */
size_t alignn_model_infer_workspace(const alignn_model_desc* desc, const alignn_model_batch* batch);
int alignn_model_infer(const alignn_model_desc* desc, const alignn_model_batch* batch, void* workspace, size_t workspace_bytes,
                       float* out, alignn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * ALIGNNAtomWise with the force / stress head as whole-model calls (csrc/model.hip + csrc/ff.hip):
 * alignn/models/alignn_atomwise.py:364-660 with calculate_gradient=True - what alignn/train.py:291-387 trains and what
 * alignn/ff/calculators.py:280-291 evaluates at every MD step.  desc->norm must be 1 (LayerNorm flavour).
 *   alignn_ff_eval: energies out[B] (:494-510), forces[N, 3] = reduce(grad_multiplier dE_tot/dr) (:530-565), stresses
 *                   [B, 3, 3] (:615-638; NULL: none) - the forward pass with the bond vectors as a leaf and its reverse
 *                   w.r.t. them (no parameter gradients); bond cosines recomputed from r when lg_on_fly (:424-431),
 *                   else batch->h.
 *   alignn_ff_grad: parameter gradients of a loss L(out, forces, stresses) given g_out[B], g_forces[N, 3], g_stress[B, 3, 3]
 *                   (each may be NULL = zero): the force / stress part is linear in the pair forces, sum_e w_e . f_e =
 *                   c D_w E_tot, so its parameter gradient is ONE reverse pass over a forward pass that carries a tangent
 *                   next to every activation (forward-over-reverse instead of autograd.grad(create_graph=True) + a second
 *                   backward; csrc/dual.hip).  The value halves are the tape alignn_ff_eval left in the workspace, which
 *                   must be the one that call filled for the same (desc, batch, ff).  Gradients go to the g_* / *red
 *                   pointers of desc, which must all lie inside gflat[0 .. grad_floats) ; gflat_t is scratch of the same
 *                   size (the tangent halves of the weight gradients, added on return).  Optionally a SECOND region of
 *                   destinations, gsink[0 .. sink_floats) with scratch gsink_t of the same size: the optimizer's packed
 *                   gradient buffer (alignn_amd/optim.py FlatAdamW) - desc pointers inside it get their tangent halves in
 *                   gsink_t and the sum in place, so that no gather of per-parameter gradients follows (NULL, NULL, 0: none).
 *   alignn_ff_plan: workspace bytes for eval alone / eval + grad; 801 when a kernel choice is not carried here.
 * ------------------------------------------------------------------------------------------ */
typedef struct alignn_ff_desc {
    int32_t lg_on_fly, add_reverse_forces, force_mult_natoms, energy_mult_natoms, has_stress, use_penalty;
    int32_t dense_lg_reverse, pad_;
    float grad_multiplier, stress_multiplier, penalty_factor, penalty_threshold;
    const float* volume;                       /* [B] cell volumes (stress) or NULL */
} alignn_ff_desc;
size_t alignn_ff_desc_sizeof(void);
int alignn_debug_allocs(size_t* out, int cap); /* debugging aid, see csrc/model.hip */
int alignn_ff_plan(const alignn_model_desc* desc, const alignn_model_batch* batch, const alignn_ff_desc* ff, size_t* eval_bytes,
                   size_t* total_bytes);
int alignn_ff_eval(const alignn_model_desc* desc, const alignn_model_batch* batch, const alignn_ff_desc* ff, void* workspace,
                   size_t workspace_bytes, float* out, float* forces, float* stress, alignn_stream_t stream);
int alignn_ff_grad(const alignn_model_desc* desc, const alignn_model_batch* batch, const alignn_ff_desc* ff, void* workspace,
                   size_t workspace_bytes, const float* g_out, const float* g_forces, const float* g_stress, float* gflat,
                   float* gflat_t, int64_t grad_floats, float* gsink, float* gsink_t, int64_t sink_floats, alignn_stream_t stream);

/* The small kernels of the force-field head (csrc/ff.hip), also used one by one by the per-operator path
 * (alignn_amd/alignn_atomwise.py, alignn_amd/ff2.py) - they replace ~100 torch element-wise / index / reduce launches and a
 * vendor GEMM (torch.einsum for a 3 x 3 mat-vec per bond) per training step:
 *   pair_force_reduce   forces[i] = scale (sum_{e into i} g_r[e] - [add_reverse] sum_{e out of i} g_r[e])  (DGL copy_e / sum on
 *                       g and dgl.reverse(g), alignn_atomwise.py:547-565); scale = grad_multiplier (* N: force_mult_natoms)
 *   virial_stress       stress[g] = (k scale / V_g) sum_{e in g} r_e (x) g_r[e], k = stress_multiplier * -160.21766208 (:615-638)
 *   ff_energy           out[g] per alignn_atomwise.py:494-510 (penalty summed over ALL bonds of the batch) and seed[g] =
 *                       d(sum en_out)/d pred[g];  ff_penalty_bwd: g_bl[e] += -B factor [bl[e] < thr]
 *   ff_pair_weights     w[e] = dL/d(pair force e) = gF[dst] - [add_reverse] gF[src] + (kS / V_g) gS_g^T r[e]; *wmax raised to
 *                       max|w| (zeroed by the caller)
 *   ff_tangent_geometry rt = w / 2^floor(log2 wmax) (1 if wmax == 0), dt = r . rt / d
 *   rbf_tangent         out_t[r][k] = exp(-gamma (d - c_k)^2) (-2 gamma (d - c_k)) dt[r]
 *   bond_cosine_tangent ht[k] = D_rt of compute_bond_cosines (alignn/graphs.py:847-864), 0 where its clamp is active
 *   ff_readout_seed     adjoints of x_i under E_g = fc(mean_i x_i): gx[i] = (ge[g] / n_g) fc_w, gxt[i] = (c_g 2^k / n_g) fc_w,
 *                       c_g = c (* n_g: energy_mult_natoms);  ff_fc_grad: gW = sum_g ge[g] hp[g] + c_g 2^k hpt[g], gb = sum ge
 *   add_inplace / add3  a += b;  out = (a + b) + c   (n % 4 == 0 / any n) */
int alignn_pair_force_reduce(const float* g_r, float scale, const int32_t* seg_ptr, const int32_t* out_ptr, const int32_t* out_slot,
                             int add_reverse, float* forces, int64_t n_nodes, alignn_stream_t stream);
int alignn_virial_stress(const float* r, const float* g_r, float scale, const int32_t* graph_ptr, const int32_t* seg_ptr,
                         const float* volume, float k, float* stress, int B, alignn_stream_t stream);
int alignn_ff_energy(const float* pred, const float* bl, const int32_t* graph_ptr, int B, int64_t E, int mult_natoms,
                     int use_penalty, float factor, float thr, float* out, float* seed, alignn_stream_t stream);
int alignn_ff_penalty_bwd(const float* bl, float* g_bl, int64_t E, int B, float factor, float thr, alignn_stream_t stream);
int alignn_ff_pair_weights(const float* gF, const float* gS, const float* r, const int32_t* src, const int32_t* dst,
                           const int32_t* graph_ptr, const int32_t* seg_ptr, const float* volume, float kS, int add_reverse, int B,
                           int64_t E, float* w, float* wmax, alignn_stream_t stream);
int alignn_ff_tangent_geometry(const float* r, const float* w, const float* wmax, const float* d, float* rt, float* dt, int64_t E,
                               alignn_stream_t stream);
int alignn_rbf_tangent(const float* d, const float* dt, const float* centers, float gamma, float* out_t, int64_t rows, int bins,
                       alignn_stream_t stream);
int alignn_bond_cosine_tangent(const float* r, const float* rt, const int32_t* e1, const int32_t* e2, float* ht, int64_t T,
                               alignn_stream_t stream);
int alignn_ff_readout_seed(const float* ge, float c, int mult_natoms, const float* wmax, const int32_t* graph_ptr, const float* fc_w,
                           float* gx, float* gxt, int B, int64_t N, int H, alignn_stream_t stream);
int alignn_ff_fc_grad(const float* ge, float c, int mult_natoms, const float* wmax, const int32_t* graph_ptr, const float* hp,
                      const float* hpt, float* gW, float* gb, int B, int H, alignn_stream_t stream);
int alignn_add_inplace(float* a, const float* b, int64_t n, alignn_stream_t stream);
int alignn_add3(const float* a, const float* b, const float* c, float* out, int64_t n, alignn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The bond-angle embedding on T rows (csrc/angle.hip):  z = MLPLayer(64 -> 256)(MLPLayer(bins -> 64)(RBFExpansion(h)))
 * (alignn/models/alignn.py:215-222 angle_embedding, :170-184 MLPLayer, alignn/models/utils.py:9-47 RBFExpansion) and
 * the gradients of its eight parameter tensors, WITHOUT the [T, bins] / [T, 64] / [T, 256] intermediates: every row is a
 * function of one cosine, so each pass recomputes what it needs from h on the matrix cores (f16x3 split products) and only z
 * (forward) / g_z and a [T, 64] gradient (backward) cross HBM.  Shapes carried: bins <= 48, 64 embedding features, 256
 * hidden features (alignn_angle_embed_supported).  BatchNorm in training mode: batch statistics, running statistics
 * updated (momentum), statistics kept in stat1 / stat2 for the backward; l1 / l2: the two layers' parameters, running
 * statistics and gradient destinations (img / img_t / w_amax unused).  scal: 128 floats the forward zeroes and both
 * directions use (operand scales, bounds).  Backward writes l?.gW, l?.gb, l?.red (= dbeta | dgamma); h gets no gradient. */
typedef struct alignn_angle_args {
    const float* h;                /* [rows] cosines */
    int64_t rows;
    const float* centers;          /* [bins] */
    float gamma;
    int32_t bins;
    alignn_mlp_params l1, l2;
    float eps, momentum;
    float *stat1, *stat2;          /* [4, 64], [4, 256]: mean | rstd | gamma rstd | beta */
    float* scal;                   /* [alignn_angle_embed_scal_floats()] */
    float *z, *z_amax;             /* forward: [rows, 256] and (optional) the tracked max|z| */
    const float* gz;               /* backward: [rows, 256] */
    void* workspace;
    size_t workspace_bytes;
} alignn_angle_args;
int alignn_angle_embed_supported(int bins, int embed, int hidden);
size_t alignn_angle_embed_workspace(int64_t rows, int bins, int backward);
size_t alignn_angle_args_sizeof(void);
int alignn_angle_embed_scal_floats(void);
int alignn_angle_embed_fwd(const alignn_angle_args* args, alignn_stream_t stream);
int alignn_angle_embed_bwd(const alignn_angle_args* args, alignn_stream_t stream);
/* evaluation mode: BatchNorm as the affine map of l?.rm / l?.rv (not updated); writes z (and z_amax), uses stat1 / stat2 /
 * scal as scratch, needs no workspace */
int alignn_angle_embed_infer(const alignn_angle_args* args, alignn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Batch staging in one call (csrc/stage.hip): canonical CSR of g (slots = bonds stably sorted by
 * destination atom), its by-source view, the canonical line graph L(g) with its by-source view and segment ranks, the
 * bond vectors in slot order and the bond-angle cosines - from the COO bond list (u -> v, caller's order, int32) a loader
 * ships.  Replaces dgl.batch + g.to(device) + lg.to(device) of alignn/train.py:264-270 (alignn/lmdb_dataset.py:87-108
 * builds and ships L(g)'s T-sized COO list and compute_bond_cosines' output, alignn/graphs.py:847-864, per batch).
 * T = rows of L(g) = sum over bonds e2 of (in-degree of src(e2)) - [e2 is a self image] is a function of the bond list:
 * the caller computes it (and the largest in-degree, for alignn_egc_bwd_lg_dense) on the host when it packs the batch,
 * so nothing is read back.  Outputs (all caller-owned): seg_ptr / out_ptr [N+1], src / dst / out_slot [E] (out_slot is
 * also L(g)'s seg_node), perm / inv [E] int64 (slot k holds caller edge perm[k]), r_canon [E,3]; lg_seg_ptr / lg_out_ptr
 * [E+1], lg_src / lg_dst / lg_out_slot / lg_seg_rank [T], lg_ident [T] int64 = 0..T-1 (optional), h [T] (optional; needs
 * r).  workspace: alignn_stage_batch_workspace(N, E) bytes.  ~15 launches, no atomics, bit-identical to alignn_amd.graph's
 * build_csr + line_graph_of (tests).
 * ------------------------------------------------------------------------------------------ */
size_t alignn_stage_batch_workspace(int64_t n_nodes, int64_t n_edges);
int alignn_stage_batch(const int32_t* u, const int32_t* v, const float* r, int64_t n_nodes, int64_t n_edges, int64_t n_triplets,
                       int32_t* seg_ptr, int32_t* src, int32_t* dst, int32_t* out_ptr, int32_t* out_slot, int64_t* perm,
                       int64_t* inv, float* r_canon, int32_t* lg_seg_ptr, int32_t* lg_src, int32_t* lg_dst,
                       int32_t* lg_out_ptr, int32_t* lg_out_slot, int32_t* lg_seg_rank, int64_t* lg_ident, float* h,
                       int32_t* out_rank /* optional [E]: rank of every slot in the by-source order */,
                       void* workspace, size_t workspace_bytes, alignn_stream_t stream);
/* The caller's OWN edge list of L(g) (the lg of the reference's (g, lg) pair: dgl's g.line_graph(shared=True), edges lg_u[k] ->
 * lg_v[k] in the caller's ids of g's edges, int64) mapped onto the canonical rows alignn_stage_batch laid out, by index
 * arithmetic (no T-sized sort): perm[t] = k (pre-fill with -1), inv_lg[k] = t, *bad (zeroed by the caller) += edges that are
 * not line-graph edges of g or repeat an earlier one.  n_triplets (= the T alignn_stage_batch was given) edges with bad == 0:
 * the caller's lg IS the line graph, perm a bijection, and its edge features go to canonical order as h[perm]. */
int alignn_map_line_graph_rows(const int64_t* lg_u, const int64_t* lg_v, const int64_t* inv, const int32_t* seg_ptr,
                               const int32_t* src, const int32_t* dst, const int32_t* out_rank, const int32_t* lg_seg_ptr,
                               int64_t n_edges, int64_t n_triplets, int64_t* perm, int64_t* inv_lg, int32_t* bad,
                               alignn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Radius bond graphs on the device (csrc/radius.hip): alignn/graphs.py:267-364 (radius_graph),
 * the neighbour strategy of the reference's force-field configs, re-run by alignn/ff/calculators.py:280-291 at every MD
 * step.  One wavefront per site, float32 distances in the reference's dtype evaluated as one fixed operation sequence,
 * bonds emitted in torch.where order (source, then periodic image, then destination) - the reference's edge ORDER, no
 * sort, no atomics, no host synchronisation.  Inputs: lat[B][9] float32 (rows a, b, c), cart[N][3] float32, graph_ptr
 * [B+1], site_graph[N], box[B][levels][6] = nmin[3], nmax[3] of the image box per cutoff level (graphs.py:296-309),
 * cut[levels] = cutoff, cutoff + 0.5, ... (:349-358), atol (isclose-zero exclusion, 1e-5).
 *   radius_levels: crystal_level[B] = first level at which the crystal's last site has a neighbour (== levels: none)
 *   radius_count:  count[N] = bonds leaving each site at its crystal's level
 *   radius_emit:   offset[N] = exclusive prefix sum of count; u, v int64, r[.][3] float32 = x_dst - x_src, image[.][3]
 *                  int32 (optional)
 * ------------------------------------------------------------------------------------------ */
int alignn_radius_levels(const float* lat, const float* cart, const int32_t* graph_ptr, const int32_t* box, const float* cut,
                         float atol, int levels, int n_crystals, int32_t* crystal_level, alignn_stream_t stream);
int alignn_radius_count(const float* lat, const float* cart, const int32_t* graph_ptr, const int32_t* site_graph,
                        const int32_t* box, const float* cut, float atol, int levels, int64_t n_sites,
                        const int32_t* crystal_level, int64_t* count, alignn_stream_t stream);
int alignn_radius_emit(const float* lat, const float* cart, const int32_t* graph_ptr, const int32_t* site_graph,
                       const int32_t* box, const float* cut, float atol, int levels, int64_t n_sites,
                       const int32_t* crystal_level, const int64_t* offset, int64_t* u, int64_t* v, float* r, int32_t* image,
                       alignn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ALIGNN_HIP_H */
