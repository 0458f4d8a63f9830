// fp32-accurate GEMMs on the 16-bit matrix cores: C[M,N] = A[M,K] W[N,K]^T (+bias +addend)  (NT, forward/dgrad)
// and dW[N,K] = G[M,N]^T X[M,K] (TN, wgrad).  gfx950 has no TF32/xf32; split products are how the H x H
// projections get off the 157 TF fp32-MFMA roof and become HBM-bound.  Two slicing schemes share every kernel:
//
// bf16x6 (no knowledge of the operand range needed).  Each fp32 operand is cut into three bf16 slices of 8
// mantissa bits by TRUNCATION
//     x = h + m + l (+ <2^-24 |x|),   h = x & 0xffff0000,  m = (x-h) & 0xffff0000,  l = upper16(x-h-m)
// (the subtractions are exact) and the product is the six slice products of weight >= 2^-16:
// hh + (hm + mh) + (hl + lh + mm), accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The dropped terms are
// < 2^-24 relative.
//
// f16x3 (needs max|A| and max|W|: device scalars that the PRODUCER kernels maintain, see block_amax_commit in
// common.h).  Each operand is scaled by a power of two that puts its largest element near 2^14 (f16_scale), then
// cut into two fp16 slices h = RN(x s), l = RN(x s - h) (22 mantissa bits; the scaling keeps l out of the fp16
// subnormals for everything within 2^-24 of the maximum) and the product is hh + hl + lh by
// v_mfma_f32_32x32x16_f16; the scales are undone in the epilogue.  Same fp32-grade error as bf16x6 (measured
// <= 2e-6 relative, tests/test_gpu_kernels.py::test_gemm_f16x3_*), half the matrix-pipe work: at T x 256 x 256
// the MFMA time (153 us at the measured 1.74 PF) drops below the HBM time (220 us), bf16x6's (306 us) does not.
//
// NT layout: 128 x 256 (RM=2) or 64 x 256 (RM=1, small M or K <= 64) block tile, 4 waves (2 x 2) of 64|32 x 128, K step 16 (one
// MFMA step), a 2-deep DMA ring with counted vmcnt waits (2 x 24-32 KiB -> two or three workgroups per CU, so one's
// epilogue/prologue overlaps the other's MFMAs).  Both operands reach LDS by DMA (global_load_lds_dwordx4: no
// staging VGPRs, no ds_write): the activation tile A stays fp32 and is sliced in registers right before use, once
// per wave row-strip; the weights arrive PRE-SLICED by alignn_split_bf16x3 / alignn_split_f16x2 in the exact
// per-(slice, k-block) image the DMA copies linearly (L2-resident: 256-384 KiB for 256 x 256).  LDS images are
// unpadded; bank conflicts are removed by an XOR swizzle of the 16-byte chunk index that is applied to the DMA
// *source* address (A) or baked into the pre-sliced layout (W) and again on the ds_read_b128 address.  Epilogue
// as in gemm_f32.hip: per-wave LDS transpose, float4 row-segment stores (write-once hint on T-sized outputs).
//
// TN layout (namespace tn): both operands are activations, so both are sliced in the kernel - cooperatively: every
// thread owns one column of a 16-row stage (16 coalesced nontemporal dword loads, two stages in flight in
// registers), slices it ONCE and writes the 16-bit planes k-contiguous to LDS; the waves then fetch MFMA operands
// with ds_read_b128.  Split-K over M into slabs; the slab partials are summed by gemm_f32.hip's slab_reduce4.
#include <cstdlib>

#include "common.h"
#include "../../include/alignn_hip.h"

// Ablation switches for tools/ablate_x6.py (never defined in the shipped build)
#ifndef X6_ABL_NOSLICE
#define X6_ABL_NOSLICE 0
#endif
#ifndef X6_ABL_NOBLOAD
#define X6_ABL_NOBLOAD 0
#endif
#ifndef X6_ABL_NOALOAD
#define X6_ABL_NOALOAD 0
#endif
#ifndef X6_ABL_ONEMFMA
#define X6_ABL_ONEMFMA 0
#endif
#ifndef X6_PERSIST
#define X6_PERSIST 0  // 1: long products on 512 persistent workgroups (bit-identical, measured no faster: see PERSIST below)
#endif
#ifndef TN_PIPE
#define TN_PIPE 0  // 1: software-pipelined stage loop of the f16x3 weight-gradient kernel (see gemm_tn_x6_kernel)
#endif
#ifndef X6_EPI_NOSYNC
#define X6_EPI_NOSYNC 0  // 1: no per-round __syncthreads() in the epilogue (see there)
#endif
#ifndef X6_KPIPE
#define X6_KPIPE 0  // 1: f16x3 k-loop with the stage hand-over in the middle of a step's MFMAs (see the k-loop)
#endif
#ifndef X6P_NT
#define X6P_NT 1  // 0: plain (L2 write-back) stores in the persistent kernel's epilogue (experiment)
#endif
#ifndef X6_TRACE
#define X6_TRACE 0  // 1: wave 0 of every workgroup stamps its phases with s_memtime (tools/x6_trace.py reads them)
#endif
#if X6_TRACE
__device__ unsigned long long x6_trace_buf[8192 * 8];
#define X6_STAMP(i)                                                                                   \
    do {                                                                                              \
        if (threadIdx.x == 0 && blockIdx.x < 8192 && blockIdx.y == 0)                                 \
            x6_trace_buf[blockIdx.x * 8 + (i)] = __builtin_readcyclecounter();                        \
    } while (0)
#else
#define X6_STAMP(i) \
    do {            \
    } while (0)
#endif
#ifndef X6_ABL_NOSTORE
#define X6_ABL_NOSTORE 0  // 1: epilogue without its global stores, 2: no epilogue at all
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int WM = 2, WN = 2, NT = WM * WN * 64;  // 4 waves
constexpr int BN = 256, BK = 16;                  // one 16-deep MFMA step per stage
constexpr int TN = BN / WN, RN = TN / 32;         // 128 columns = 4 MFMA tiles per wave
constexpr int B_PLANE = BN * BK * 2;              // 8 KiB per 16-bit slice plane (32-byte rows)
// block height: RM 32-row MFMA tiles per wave along M.  RM = 2 (128 x 256 tile, 2 workgroups per CU) halves the
// weight-slice traffic per activation row - best for the deep T- and E-row products; RM = 1 (64 x 256, 3 per CU)
// fills the chip with shallow (K <= 64: 6.5 TB/s at T x 256 x 64) or short (atom-row) products.
template <int RM_>
struct Geo {
    static constexpr int RM = RM_, BM = WM * 32 * RM_, TM = BM / WM;
    static constexpr int A_BYTES = BM * BK * 4;              // fp32 (64-byte rows), XOR-swizzled 16-byte chunks
    static constexpr int A_DMA = A_BYTES / 1024 / (NT / 64);  // 1 KiB DMA pieces per wave
};
constexpr int EPI_BYTES = (NT / 64) * 32 * (64 + 4) * 4; // per-wave [32][68] fp32 transpose patches
constexpr int EPI1_LDS = EPI_BYTES + (NT / 64) * 2 * 2 * 256 * 4;  // ... + the EPI == 1 column-sum slots of every lane (50 KiB)
// per scheme: slice planes of the weight image (3 bf16 / 2 fp16), stage and LDS footprint
template <bool F16, int RM_ = 2>
struct Sch {
    static constexpr int NPL = F16 ? 2 : 3;
    static constexpr int STAGE = Geo<RM_>::A_BYTES + NPL * B_PLANE;  // RM = 2: 32 KiB / 24 KiB
    static constexpr int B_DMA = NPL * B_PLANE / 1024 / (NT / 64);   // pieces per wave
#ifdef X6_NSTAGE
    static constexpr int NSTAGE = X6_NSTAGE;
#else
    static constexpr int NSTAGE = 2;  // DMA ring depth (3 measured no faster for f16x3: the kernel is HBM-bound)
#endif
    static constexpr int LDS = NSTAGE * STAGE > EPI_BYTES ? NSTAGE * STAGE : EPI_BYTES;  // two workgroups per CU
    // persistent walk: ring of two stages, the transpose patches overlay stage 1 and run past the ring's end
    static constexpr int LDS_PERSIST = 2 * STAGE > STAGE + EPI_BYTES ? 2 * STAGE : STAGE + EPI_BYTES;
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// s_barrier without the vmcnt(0)/lgkmcnt(0) drain that __syncthreads() implies: DMA stages stay in flight across it
__device__ __forceinline__ void block_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

struct X6Args {
    const float* A;
    int64_t lda;
    const unsigned char* Ws;  // [K/16][Npad/256][planes][256 rows][2 chunks, swizzled][8 x 16 bit]: 8 KiB per plane
    const float* a_amax;      // f16x3 only: device scalars max|A| (or an upper bound) and max|W|
    const float* w_amax;
    const float* bias;
    const float* addend;
    int64_t ldadd;
    float* C;
    int64_t ldc;
    int64_t M;
    int N;
    int Npad;
    int K;
    int stream_out;  // C cannot stay in the last-level cache: write-once hint on its stores
    // EPI == 1 (BatchNorm-backward reductions of the tensor this product's OUTPUT is the gradient of, see the kernel):
    const float* xn;     // [M, N] pre-activation of that BatchNorm (m_pre of the producing convolution / MLP layer)
    int64_t ldxn;
    const float* nstat;  // [4][N]: mean, rstd, gamma*rstd, beta
    float* red_partial;  // [row tiles][2][N]: per-tile column sums of gz and gz*xhat
    // EPI == 2 (edge-gate projection of EdgeGatedGraphConv: the output row e also gets A[src e] + Bd[dst e]):
    const float* gp;      // node projection P = [A | Bd | Bh | Ux], rows of ldgp floats; A at column 0, Bd at column N
    int64_t ldgp;
    const int32_t* gsrc;  // [M] source node of edge row e
    const int32_t* gdst;  // [M] destination node of edge row e
    int strip_slabs;      // one-tile kernels: column-sum slabs per 64-row wave strip (see the end of gemm_nt_x6_body)
    int xcd_map;          // persistent kernel: 1 = each XCD walks its own contiguous range of row tiles (see gemm_nt_f16p_body)
    // EPI == 2, optional: the SECOND gathered row comes from its own table instead of P's Bd block - row gdst[e] of gp2
    // (leading dimension ldgp2).  The host passes the Bd rows permuted into SEGMENT order and gdst = the segment rank of
    // row e, so consecutive output rows read consecutive table rows (alignn_gemm_nt_f16x3_gather2).
    const float* gp2;
    int64_t ldgp2;
};

__device__ __forceinline__ unsigned hi_pair(float x1, float x0) {
    // (upper16(x1) << 16) | upper16(x0): two bf16 (truncated) in one dword
    return __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
}
__device__ __forceinline__ float trunc16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

// slice 8 consecutive-k floats into the three bf16x8 MFMA operands
__device__ __forceinline__ void slice8(const float4& lo, const float4& hi4, bf16x8& h, bf16x8& m, bf16x8& l) {
#if X6_ABL_NOSLICE
    h = __builtin_bit_cast(bf16x8, lo);
    m = __builtin_bit_cast(bf16x8, hi4);
    l = h;
    return;
#endif
    float x[8] = {lo.x, lo.y, lo.z, lo.w, hi4.x, hi4.y, hi4.z, hi4.w};
    uint4 hp, mp, lp;
    unsigned* hpp = reinterpret_cast<unsigned*>(&hp);
    unsigned* mpp = reinterpret_cast<unsigned*>(&mp);
    unsigned* lpp = reinterpret_cast<unsigned*>(&lp);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float x0 = x[2 * p], x1 = x[2 * p + 1];
        hpp[p] = hi_pair(x1, x0);
        const float r0 = x0 - trunc16(x0), r1 = x1 - trunc16(x1);
        mpp[p] = hi_pair(r1, r0);
        const float q0 = r0 - trunc16(r0), q1 = r1 - trunc16(r1);
        lpp[p] = hi_pair(q1, q0);
    }
    h = __builtin_bit_cast(bf16x8, hp);
    m = __builtin_bit_cast(bf16x8, mp);
    l = __builtin_bit_cast(bf16x8, lp);
}

// power-of-two scale that puts a tensor with the given max|x| just below 2^15 (fp16 max is 65504); 1 for an
// all-zero / denormal / non-finite tensor (nothing to protect; inf and nan propagate through the fp16 slices)
__device__ __forceinline__ float f16_scale(float amax) {
    const int e = (int)((__float_as_uint(amax) >> 23) & 255u);  // amax < 2^(e-126)
    if (e == 0 || e == 255) return 1.0f;
    int se = 268 - e;                                           // 2^(141-e)
    se = se > 254 ? 254 : se;
    return __uint_as_float((unsigned)se << 23);
}

// slice 8 consecutive-k floats (scaled by s) into the two fp16x8 MFMA operands - in two halves, so that the kernel
// can start the ah products while the VALU still works on the low slice
__device__ __forceinline__ void slice8_f16_hi(const float4& lo, const float4& hi4, float s, float (&xs)[8], f16x8& h) {
#if X6_ABL_NOSLICE
    h = __builtin_bit_cast(f16x8, lo);
    xs[0] = hi4.x, xs[1] = hi4.y, xs[2] = hi4.z, xs[3] = hi4.w;
    return;
#endif
    const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        xs[j] = x[j] * s;
        h[j] = (_Float16)xs[j];  // round to nearest even
    }
}
__device__ __forceinline__ void slice8_f16_lo(const float (&xs)[8], const f16x8& h, f16x8& l) {
#if X6_ABL_NOSLICE
    l = __builtin_bit_cast(f16x8, make_float4(xs[0], xs[1], xs[2], xs[3]));
    return;
#endif
    // l = RN_f16(xs - h) (the subtraction is exact in fp32).  As v_fma_mixlo/hi_f16 (fp32 fma of xs * 1.0 - h with the fp16
    // operand read straight out of the packed high slice, result rounded into one half of the destination): one
    // instruction per element where hipcc emits v_cvt_f32_f16 + v_pk_add_f32 + v_cvt_pk_f16_f32 (two per element) - and
    // the packed fp32 operations are the expensive neighbours of MFMAs.  Same bits (tools/mix_check.hip: 16 M pairs).
    const uint4 hp = __builtin_bit_cast(uint4, h);
    const unsigned hw[4] = {hp.x, hp.y, hp.z, hp.w};
    unsigned lw[4];
#pragma unroll
    for (int p2 = 0; p2 < 4; ++p2) {
        asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(lw[p2]) : "v"(xs[2 * p2]), "v"(hw[p2]));
        asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lw[p2]) : "v"(xs[2 * p2 + 1]), "v"(hw[p2]));
    }
    l = __builtin_bit_cast(f16x8, make_uint4(lw[0], lw[1], lw[2], lw[3]));
}
__device__ __forceinline__ void slice8_f16(const float4& lo, const float4& hi4, float s, f16x8& h, f16x8& l) {
    float xs[8];
    slice8_f16_hi(lo, hi4, s, xs, h);
    slice8_f16_lo(xs, h, l);
}

// 64 lanes x 16 B -> 1 KiB of LDS at lds_wave_base (wave-uniform) + lane*16, from sbase (wave-uniform, scalar
// registers) + lane_off (+ OFF).  Written as the instruction itself: through __builtin_amdgcn_global_load_lds hipcc
// builds a 64-bit vector address per piece and k-step (v_lshl_add_u64) and, in the pipelined loop, answered a ds_read
// whose destination landed on such an address pair with s_waitcnt vmcnt(0) - a wait for the DMA issued just before.
// The compiler neither sees these loads (every vmcnt wait on them is explicit, see wait_vmcnt) nor uses M0 for
// anything else on gfx950.  (Default cache policy: nt on the read-once activation tile measured 10 % slower.)
template <int OFF>
__device__ __forceinline__ void dma16(const void* sbase, unsigned lane_off, unsigned char* lds_wave_base) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3"
                 :
                 : "s"((unsigned)(size_t)lds_wave_base), "v"(lane_off), "s"(sbase), "n"(OFF)
                 : "memory");
}

// PERSIST: the launch has fewer workgroups than row tiles; a workgroup walks tiles blockIdx.x, +gridDim.x, ... and
// DMA-prefetches the first k-stage of its NEXT tile before it issues the stores of the current one.  A one-tile
// workgroup cannot retire (and free its LDS and registers for the next tile) before every store is acknowledged -
// two thirds of the measured epilogue cost (profiles/r01_final_f16x3_ablation.txt); here the stores drain under the
// next tile's first k-step.  vmcnt retires in issue order and counts stores, so that step waits with
// vmcnt(<stores issued after the prefetch>) instead of vmcnt(0).  Needs an even number of k-steps (the last step
// reads stage 1, stage 0 is free for the prefetch, the transpose patches overlay stage 1 onwards).
// STATUS: compiled only with -DX6_PERSIST=1.  Results are bit-identical to the one-tile kernel (tools/ablate_x6.py,
// ABL_CHECK=1, T x 256 x 256), but it measured no faster (374 vs 378 us, interleaved rounds): the second k-step's
// vmcnt(0) still meets the stores one step later.  Kept as the base for a deeper walk (three stages: two prefetched
// ahead of the stores) - see HISTORY.md section 8.
// EPI == 1: the output C is a gradient g_y = dL/dy of a tensor y = r + silu(BatchNorm(xn)) (the edge output of the previous
// line-graph convolution, or an MLPLayer output without r).  BatchNorm's backward needs the column sums
// sum_rows gz and sum_rows gz*xhat (gz = g_y * silu'(z)) over ALL rows before anything else can happen - a separate
// 2-row-pass kernel (col_reduce<BwdReduceFn>) when done on its own.  Here the tile that has just been computed is still
// in registers: read the matching xn tile (one row pass instead of two), form gz and accumulate the two sums per
// column in a fixed order (lane -> the 4 row groups of a wave by shuffles -> the 2 wave rows through LDS); one [2][N]
// slab per row tile, summed afterwards by alignn_bn_bwd_finalize (fp64, fixed order): bit-reproducible.
// EPI is a set of flags: 1 = BNRED (above), 2 = GATHER (C[e] += P[src e].A + P[dst e].Bd, below), 4 = STATS (per-tile
// column sums of C and C^2 into red_partial: the BatchNorm statistics of the tensor this product writes, so that no
// separate pass has to read it back for them).
template <bool HAS_ADD, bool F16, int RM_, bool PERSIST = false, int EPI = 0>
__device__ __forceinline__ void gemm_nt_x6_body(const X6Args& g) {
    constexpr bool BNRED = (EPI & 1) != 0, GATHER = (EPI & 2) != 0, STATS = (EPI & 4) != 0;
    static_assert(!(BNRED && STATS), "one set of column sums per launch");
    static_assert(EPI == 0 || !PERSIST, "the reduction epilogue exists for the one-tile kernel only");
    static_assert(!GATHER || !HAS_ADD, "the gather variant has its own two addends");
    // EPI == 2: C[e] = A-row . W + b  +  P[src e].A + P[dst e].Bd  - the u_add_v of the convolution
    // (alignn/models/alignn.py:100-101) folded into the projection that produces the third addend, so that m never
    // makes the round trip "write C, read C, write m".  (a + bd) + c is evaluated as c + (a + bd): same bits.
    constexpr int NPL = Sch<F16, RM_>::NPL, STAGE_BYTES = Sch<F16, RM_>::STAGE, B_DMA = Sch<F16, RM_>::B_DMA;
    constexpr int RM = Geo<RM_>::RM, BM = Geo<RM_>::BM, TM = Geo<RM_>::TM, A_BYTES = Geo<RM_>::A_BYTES,
                  A_DMA = Geo<RM_>::A_DMA;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int il = lane & 31, half = lane >> 5;
    const int64_t n_mt = (g.M + BM - 1) / BM;  // row tiles (PERSIST walks them; otherwise gridDim.x == n_mt)
    int64_t tile = blockIdx.x;
    int64_t m0 = tile * BM;
    const int n0 = blockIdx.y * BN;

    f32x16 acc[RM][RN];

#if X6_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 8192 && blockIdx.y == 0) {
        unsigned id, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        x6_trace_buf[blockIdx.x * 8 + 7] = ((unsigned long long)xcc << 32) | id;
    }
    X6_STAMP(0);
#endif

    // ---- DMA addressing (LDS image is lane-linear; the XOR swizzle lives in the SOURCE address)
    // A: piece q = wave*A_DMA + i holds tile positions p = q*64 + lane -> row p/4, stored chunk p%4, which is
    //    global chunk (p%4) ^ ((row>>2)&3) of that row.
    // Global addresses are a wave-uniform base (scalar registers, advanced per k-step) plus a loop-invariant 32-bit lane
    // offset: no per-piece 64-bit vector arithmetic in the loop and no short-lived address registers (a ds_read whose
    // destination the allocator had put on a just-used address pair drew an s_waitcnt vmcnt(0) from hipcc in the
    // pipelined loop - i.e. a wait for the DMA issued two instructions earlier).
    unsigned a_lane[A_DMA];
    const float* a_base = g.A;
    auto set_rows = [&](int64_t row0) {
        a_base = g.A + row0 * g.lda;
#pragma unroll
        for (int i = 0; i < A_DMA; ++i) {
            const int p = (wave * A_DMA + i) * 64 + lane;
            const int row = p >> 2, c = (p & 3) ^ ((row >> 2) & 3);
            int64_t grow = row0 + row;
            if (grow >= g.M) grow = g.M - 1;  // clamp: rows past the end are computed but never stored
            a_lane[i] = (unsigned)(((grow - row0) * g.lda + c * 4) * 4);
        }
    };
    set_rows(m0);
    // B: the slice planes of one (k-block, n-tile) are ONE contiguous 24 (16) KiB run in global memory, in
    // exactly the LDS image order (pre-swizzled), so a stage is sequential 1 KiB pieces and consecutive
    // k-blocks walk the weight image linearly (no power-of-two plane strides in the L2).
    const int64_t kb_stride = (int64_t)(g.Npad / BN) * (NPL * B_PLANE);
    const unsigned char* b_base = g.Ws + (int64_t)blockIdx.y * (NPL * B_PLANE) + (wave * B_DMA) * 1024;
    const unsigned b_lane = lane * 16;
    auto issue = [&](int kt, unsigned char* stage) {
#if X6_ABL_NOALOAD
        if (kt == 0)
#endif
#pragma unroll
        for (int i = 0; i < A_DMA; ++i)
            dma16<0>(a_base + kt * BK, a_lane[i], stage + (wave * A_DMA + i) * 1024);
#if X6_ABL_NOBLOAD
        if (kt == 0)
#endif
#pragma unroll
        for (int i = 0; i < B_DMA; ++i)
            dma16<0>(b_base + kt * kb_stride + i * 1024, b_lane, stage + A_BYTES + (wave * B_DMA + i) * 1024);
    };

    // reader addresses (bytes inside a stage)
    int a_off0[RM], a_off1[RM];
#pragma unroll
    for (int a = 0; a < RM; ++a) {
        const int arow = wm * TM + a * 32 + il;
        const int a_f = (arow >> 2) & 3;
        a_off0[a] = arow * (BK * 4) + (((2 * half) ^ a_f) << 4);
        a_off1[a] = arow * (BK * 4) + (((2 * half + 1) ^ a_f) << 4);
    }
    int b_off[RN];
#pragma unroll
    for (int b = 0; b < RN; ++b) {
        const int n = wn * TN + b * 32 + il;
        b_off[b] = A_BYTES + n * (BK * 2) + ((half ^ ((n >> 3) & 1)) << 4);
    }

    // bias for this wave's 64-column groups: fetched now, consumed after the k-loop, so the epilogue has no load
    // of its own in front of its stores (hipcc would otherwise wait vmcnt(0) - i.e. for the previous STORE -
    // before every store of the epilogue)
    const int e_prow = lane >> 4, e_pc4 = (lane & 15) * 4;
    float4 bias_v[RN / 2];
#pragma unroll
    for (int hb = 0; hb < RN / 2; ++hb) {
        const int col = n0 + wn * TN + hb * 64 + e_pc4;
        bias_v[hb] = (g.bias && col < g.N) ? f4_ld(g.bias + col) : f4_zero();
    }

    // f16x3: power-of-two operand scales (device scalars written by the producers), undone in the epilogue
    float sa = 1.0f, inv_sa = 1.0f, inv_sw = 1.0f;
    if constexpr (F16) {
        sa = f16_scale(*g.a_amax);
        inv_sa = 1.0f / sa;
        inv_sw = 1.0f / f16_scale(*g.w_amax);
    }

    // DMA ring of NSTAGE stages: stage kt+NSTAGE-1 is issued at the top of step kt (into the slot every wave has
    // finished reading), so a stage has NSTAGE-1 k-steps to land; COUNTED vmcnt - only the stage about to be read
    // must have arrived, younger ones stay in flight across the barrier.
    constexpr int NSTAGE = Sch<F16, RM_>::NSTAGE, PIECES = A_DMA + B_DMA;
    static_assert(NSTAGE == 2 || NSTAGE == 3, "vmcnt cases below");
    static_assert(!PERSIST || NSTAGE == 2, "the persistent walk prefetches into stage 0 of a two-stage ring");
    constexpr int EPI_STORES = RM * (RN / 2) * 8;  // global stores per lane and tile, all issued after the prefetch
    static_assert(EPI_STORES < 64, "vmcnt is a 6-bit counter");
    const int nk = g.K / BK;
#pragma unroll
    for (int s0 = 0; s0 < NSTAGE - 1; ++s0)
        if (s0 < nk) issue(s0, smem + s0 * STAGE_BYTES);
    bool first = true;
    for (;;) {  // row tiles of this workgroup (one iteration unless PERSIST)
#pragma unroll
    for (int a = 0; a < RM; ++a)
#pragma unroll
        for (int b = 0; b < RN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    if constexpr (F16 && X6_KPIPE && !PERSIST) {
        // Software-pipelined form of the loop below.  A wave's step there is a serial chain - wait for the stage, barrier,
        // issue the next DMA, LDS reads, high slices, 24 MFMAs - of which only the MFMAs are hidden by the co-resident
        // workgroup.  Here the stage hand-over sits in the MIDDLE of a step's products: after the first two passes of step
        // kt (16 MFMAs in the pipe) the wave waits for stage kt+1, passes the barrier, issues the DMA of stage kt+NSTAGE
        // into the slot of stage kt (every wave's LDS reads of it have completed: block_barrier() drains lgkmcnt) and
        // reads the operands of step kt+1 - into the registers the finished passes have released - while the third pass
        // runs.  Same products in the same order per accumulator: bit-identical.
        f16x8 ah[RM], al[RM], bh[RN], bl[RN];
        float xs[RM][8];
        auto read_hi = [&](const unsigned char* stage) {
#pragma unroll
            for (int b = 0; b < RN; ++b) bh[b] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(stage + b_off[b]));
#pragma unroll
            for (int a = 0; a < RM; ++a)
                slice8_f16_hi(*reinterpret_cast<const float4*>(stage + a_off0[a]),
                              *reinterpret_cast<const float4*>(stage + a_off1[a]), sa, xs[a], ah[a]);
        };
        auto read_lo = [&](const unsigned char* stage) {
#pragma unroll
            for (int b = 0; b < RN; ++b)
                bl[b] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(stage + b_off[b] + B_PLANE));
        };
#define X6_PASS(AA, BB)                                                                              \
    _Pragma("unroll") for (int a = 0; a < RM; ++a) _Pragma("unroll") for (int b = 0; b < RN; ++b)    \
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AA[a], BB[b], acc[a][b], 0, 0, 0);
        // stage 0: landed -> barrier -> DMA of stage NSTAGE-1 -> operands of step 0
        if (NSTAGE == 3 && 1 < nk)
            wait_vmcnt<PIECES>();
        else
            wait_vmcnt<0>();
        block_barrier();
        if (NSTAGE - 1 < nk) issue(NSTAGE - 1, smem + ((NSTAGE - 1) % NSTAGE) * STAGE_BYTES);
        read_hi(smem);
        read_lo(smem);
        for (int kt = 0; kt < nk; ++kt) {
            X6_PASS(ah, bh)
#pragma unroll
            for (int a = 0; a < RM; ++a) slice8_f16_lo(xs[a], ah[a], al[a]);
            X6_PASS(al, bh)
            const bool more = kt + 1 < nk;
            const unsigned char* nstage = smem + ((kt + 1) % NSTAGE) * STAGE_BYTES;
            float4 raw0[RM], raw1[RM];
            if (more) {
                if (NSTAGE == 3 && kt + 2 < nk)
                    wait_vmcnt<PIECES>();
                else
                    wait_vmcnt<0>();
                block_barrier();
                if (kt + NSTAGE < nk) issue(kt + NSTAGE, smem + ((kt + NSTAGE) % NSTAGE) * STAGE_BYTES);
#pragma unroll
                for (int b = 0; b < RN; ++b)
                    bh[b] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(nstage + b_off[b]));
#pragma unroll
                for (int a = 0; a < RM; ++a) {
                    raw0[a] = *reinterpret_cast<const float4*>(nstage + a_off0[a]);
                    raw1[a] = *reinterpret_cast<const float4*>(nstage + a_off1[a]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            X6_PASS(ah, bl)
            __builtin_amdgcn_sched_barrier(0);  // (the LDS latency of the reads above is spent under this pass)
            if (more) {
                read_lo(nstage);
#pragma unroll
                for (int a = 0; a < RM; ++a) slice8_f16_hi(raw0[a], raw1[a], sa, xs[a], ah[a]);
            }
        }
#undef X6_PASS
    } else
    for (int kt = 0; kt < nk; ++kt) {
        if (PERSIST && kt == 0 && !first)
            wait_vmcnt<EPI_STORES>();  // stage 0 was prefetched BEFORE the previous tile's stores: leave those in flight
        else if (NSTAGE == 3 && kt + 1 < nk)
            wait_vmcnt<PIECES>();
        else
            wait_vmcnt<0>();
        block_barrier();
        if (kt == 0) X6_STAMP(1);
        if (kt == 8) X6_STAMP(2);
        // (issuing these DMA pieces between the product passes instead - they cost a wave fewer issue cycles among
        // MFMAs than in front of LDS reads - measured 3.5 % SLOWER at T x 256 x 256)
        if (kt + NSTAGE - 1 < nk) issue(kt + NSTAGE - 1, smem + ((kt + NSTAGE - 1) % NSTAGE) * STAGE_BYTES);
        const unsigned char* stage = smem + (kt % NSTAGE) * STAGE_BYTES;
        if constexpr (F16) {
            // In-order issue: a wave that slices first and multiplies afterwards leaves the matrix pipe idle for the
            // ~65 VALU instructions of the slicing (measured: 88 us of 416 at T x 256 x 256).  So: high slices
            // first, then the ah.bh and ah.bl products with the low-slice arithmetic issued in their shadow (a
            // 32x32x16 MFMA holds the pipe for 8 passes; several VALU fit behind each), al.bh last.  With the source
            // in this order hipcc interleaves them by itself (a forced sched_group_barrier pipeline measured 3 %
            // slower than its choice).
            f16x8 ah[RM], al[RM], bh[RN], bl[RN];
            float xs[RM][8];
#pragma unroll
            for (int a = 0; a < RM; ++a)
                slice8_f16_hi(*reinterpret_cast<const float4*>(stage + a_off0[a]),
                              *reinterpret_cast<const float4*>(stage + a_off1[a]), sa, xs[a], ah[a]);
#pragma unroll
            for (int b = 0; b < RN; ++b) {
                const unsigned char* q = stage + b_off[b];
                bh[b] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(q));
                bl[b] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(q + B_PLANE));
            }
#define X6_PASS(AA, BB)                                                                              \
    _Pragma("unroll") for (int a = 0; a < RM; ++a) _Pragma("unroll") for (int b = 0; b < RN; ++b)    \
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AA[a], BB[b], acc[a][b], 0, 0, 0);
            X6_PASS(ah, bh)
#pragma unroll
            for (int a = 0; a < RM; ++a) slice8_f16_lo(xs[a], ah[a], al[a]);
#if !X6_ABL_ONEMFMA
            X6_PASS(ah, bl)
            X6_PASS(al, bh)
#endif
#undef X6_PASS
        } else {
            bf16x8 ah[RM], am[RM], al[RM], bh[RN], bm[RN], bl[RN];
#pragma unroll
            for (int a = 0; a < RM; ++a)
                slice8(*reinterpret_cast<const float4*>(stage + a_off0[a]),
                       *reinterpret_cast<const float4*>(stage + a_off1[a]), ah[a], am[a], al[a]);
#pragma unroll
            for (int b = 0; b < RN; ++b) {
                const unsigned char* q = stage + b_off[b];
                bh[b] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(q));
                bm[b] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(q + B_PLANE));
                bl[b] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(q + 2 * B_PLANE));
            }
            // six slice products, smallest first; each pass walks the independent accumulators so no MFMA waits
            // on the one issued right before it
#define X6_PASS(AA, BB)                                                                              \
    _Pragma("unroll") for (int a = 0; a < RM; ++a) _Pragma("unroll") for (int b = 0; b < RN; ++b)    \
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AA[a], BB[b], acc[a][b], 0, 0, 0);
#if !X6_ABL_ONEMFMA
            X6_PASS(al, bh)
            X6_PASS(ah, bl)
            X6_PASS(am, bm)
            X6_PASS(am, bh)
            X6_PASS(ah, bm)
#endif
            X6_PASS(ah, bh)
#undef X6_PASS
        }
    }

    X6_STAMP(3);
    // epilogue: per-wave LDS transpose of two 32x32 tiles at a time -> float4 row segments.  Per round: all LDS
    // traffic first, then all addend loads (rows clamped, no per-element branches), then the stores.
    const int64_t next = tile + gridDim.x;
    const bool has_next = PERSIST && next < n_mt;
    if (has_next) {  // every wave has passed the last k-step's barrier, i.e. nobody reads stage 0 any more
        set_rows(next * BM);
        issue(0, smem);
    }
    constexpr int PLD = 64 + 4;
    float* patch = reinterpret_cast<float*>(smem + (PERSIST ? STAGE_BYTES : 0)) + wave * (32 * PLD);
    // EPI == 1: every lane's running column sums (4 columns x its rows, per 64-column half), behind the transpose patches
    float* red_acc = reinterpret_cast<float*>(smem + EPI_BYTES);
    static_assert(EPI_BYTES + (NT / 64) * (RN / 2) * 2 * 256 * 4 == EPI1_LDS, "LDS for the reduction slots (see launch_nt_rm)");
    const int prow = e_prow, pc4 = e_pc4;
#if X6_ABL_NOSTORE == 2
    if (acc[0][0][0] == 12345.678f)  // (keeps the accumulators alive)
#endif
#pragma unroll
    for (int ahb = 0; ahb < RM * (RN / 2); ++ahb) {
        const int a = ahb / (RN / 2), hb = ahb % (RN / 2);
        // the patches are private to a wave (LDS serves one wave's accesses in order): the only cross-wave hazard is
        // the first overwrite of stage memory other waves may still be reading.  PERSIST must not use
        // __syncthreads() here - its vmcnt(0) would drain the prefetch just issued.
#if X6_EPI_NOSYNC
        // one barrier in front of the first overwrite of stage memory; afterwards every wave works on its own patch
        // (LDS serves one wave's accesses in order) and the stores of round r stay in flight under the transposes of
        // round r+1 - __syncthreads() would drain them (it waits vmcnt(0)) four times per tile
        if (ahb == 0) block_barrier();
#else
        if constexpr (PERSIST) {
            if (ahb == 0) block_barrier();
        } else {
            __syncthreads();
        }
#endif
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                patch[((r & 3) + 8 * (r >> 2) + 4 * half) * PLD + b * 32 + il] = acc[a][2 * hb + b][r];
#if !X6_EPI_NOSYNC
        if constexpr (!PERSIST) __syncthreads();
#endif
        const int col = n0 + wn * TN + hb * 64 + pc4;
        const int colc = col < g.N ? col : 0;
        const int64_t row0 = m0 + wm * TM + a * 32 + prow;
        float4 ov[BNRED ? 1 : 8], av[8], xv[BNRED ? 8 : 1];
        if constexpr (PERSIST) {
            // read the patch behind the compiler's back: it would put s_waitcnt vmcnt(0) in front of LDS reads that
            // it cannot tell apart from the destination of the DMA prefetch in flight
            v4f pv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("ds_read_b128 %0, %1"
                             : "=v"(pv[i])
                             : "v"((unsigned)(size_t)(patch + (i * 4 + prow) * PLD + pc4))
                             : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]), "+v"(pv[4]), "+v"(pv[5]), "+v"(pv[6]),
                           "+v"(pv[7])
                         :
                         : "memory");
#pragma unroll
            for (int i = 0; i < 8; ++i) ov[BNRED ? 0 : i] = make_float4(pv[i].x, pv[i].y, pv[i].z, pv[i].w);
        } else if constexpr (!BNRED) {
#pragma unroll
            for (int i = 0; i < 8; ++i) ov[i] = f4_ld(patch + (i * 4 + prow) * PLD + pc4);
        }
        // all global loads of the round are issued before its first store (vmcnt retires in order and counts stores: a
        // load behind a store would make its consumer wait for that store)
        if constexpr (GATHER) {
            int ui[8], vi[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int64_t row = row0 + i * 4;
                if (row >= g.M) row = g.M - 1;
                ui[i] = g.gsrc[row];
                vi[i] = g.gdst[row];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                av[i] = f4_add(f4_ld(g.gp + (int64_t)ui[i] * g.ldgp + colc),
                               f4_ld((g.gp2 ? g.gp2 : g.gp + g.N) + (int64_t)vi[i] * (g.gp2 ? g.ldgp2 : g.ldgp) + colc));
        }
        if (HAS_ADD) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int64_t row = row0 + i * 4;
                if (row >= g.M) row = g.M - 1;
                av[i] = f4_ld(g.addend + row * g.ldadd + colc);
            }
        }
        float4 n_mean, n_sc, n_be, s0, s1;
        if constexpr (STATS) s0 = s1 = f4_zero();
        if constexpr (BNRED) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int64_t row = row0 + i * 4;
                if (row >= g.M) row = g.M - 1;
                xv[i] = f4_lds<true>(g.xn + row * g.ldxn + colc);
            }
            n_mean = f4_ld(g.nstat + colc), n_sc = f4_ld(g.nstat + 2 * g.N + colc), n_be = f4_ld(g.nstat + 3 * g.N + colc);
            s0 = s1 = f4_zero();
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t row = row0 + i * 4;
            if constexpr (BNRED) ov[0] = f4_ld(patch + (i * 4 + prow) * PLD + pc4);  // (LDS: just in time, 4 registers)
            float4 v = ov[BNRED ? 0 : i];
            if constexpr (F16) v = f4_scale(f4_scale(v, inv_sa), inv_sw);
            v = f4_add(v, bias_v[hb]);
            if (HAS_ADD || GATHER) v = f4_add(v, av[i]);
            if (row < g.M && col < g.N && (!X6_ABL_NOSTORE || v.x == 12345.678f)) {
                if (g.stream_out)
                    f4_sts<true>(g.C + row * g.ldc + col, v);
                else
                    f4_st(g.C + row * g.ldc + col, v);
                if constexpr (STATS) {
                    s0 = f4_add(s0, v);
                    s1 = f4_fma(v, v, s1);
                }
                if constexpr (BNRED) {
                    const float4 xc = f4_sub(xv[i], n_mean);
                    const float4 z = f4_fma(xc, n_sc, n_be);
                    const float4 gz = make_float4(v.x * dsilu_f(z.x), v.y * dsilu_f(z.y), v.z * dsilu_f(z.z), v.w * dsilu_f(z.w));
                    s0 = f4_add(s0, gz);
                    s1 = f4_fma(gz, xc, s1);  // (x rstd once, at the end)
                }
            }
        }
        if constexpr (BNRED || STATS) {
            // the lane's running column sums live in LDS between rounds (registers: two waves per SIMD is the budget)
            float* acc_sh = red_acc + ((wave * (RN / 2) + hb) * 2) * 256 + lane * 4;  // [wave][half][2][64 lanes][4]
            if (a != 0) {
                s0 = f4_add(f4_ld(acc_sh), s0);
                s1 = f4_add(f4_ld(acc_sh + 256), s1);
            }
            f4_st(acc_sh, s0);
            f4_st(acc_sh + 256, s1);
        }
    }
    if constexpr (BNRED || STATS) {
        // column sums: the 4 row groups of a wave (lanes l, l+16, l+32, l+48), then the 2 wave rows - fixed order
        // (strip_slabs: one slab per wave row strip instead of one per tile - the slab granularity of the persistent
        // kernel, which takes most variants of the same shape; alignn_gemm_nt_x6_row_tiles counts strips then)
        __syncthreads();
        const bool strips = g.strip_slabs != 0;
        if ((strips ? m0 + wm * TM < g.M : wm == 0) && lane < 16) {
#pragma unroll
            for (int hb = 0; hb < RN / 2; ++hb) {
                const int col = n0 + wn * TN + hb * 64 + lane * 4;
                if (col < g.N) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        float4 sum = f4_zero();
#pragma unroll
                        for (int w = 0; w < WM; ++w)
                            if (!strips || w == wm)
#pragma unroll
                                for (int pr = 0; pr < 4; ++pr)
                                    sum = f4_add(sum, f4_ld(red_acc + (((w * WN + wn) * (RN / 2) + hb) * 2 + k) * 256 +
                                                            (pr * 16 + lane) * 4));
                        if (BNRED && k == 1) sum = f4_mul(sum, f4_ld(g.nstat + g.N + col));  // sum gz*(x-mean) -> sum gz*xhat
                        const size_t slab = strips ? (size_t)tile * WM + wm : (size_t)tile;
                        f4_st(g.red_partial + (slab * 2 + k) * g.N + col, sum);
                    }
                }
            }
        }
    }
    X6_STAMP(4);
#if X6_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    X6_STAMP(5);
#endif
    if (!has_next) break;
    tile = next;
    m0 = tile * BM;
    first = false;
    }  // row tiles
}

template <bool HAS_ADD, bool F16, int RM_, bool PERSIST = false>
__global__ __launch_bounds__(NT) void gemm_nt_x6_kernel(X6Args g) {
    gemm_nt_x6_body<HAS_ADD, F16, RM_, PERSIST, 0>(g);
}
// the EPI == 1 variant needs more registers than the compiler's default target leaves for two waves per SIMD (it would
// settle for one: 170 + 128 accumulator registers); pinning the occupancy makes it allocate within 256 (a handful of
// spills with an addend).  Not applied to the plain kernel: there it measured 2.6 % slower (356 vs 347 us).
template <bool HAS_ADD, int RM_>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2))) void gemm_nt_x6_bnred_kernel(X6Args g) {
    gemm_nt_x6_body<HAS_ADD, true, RM_, false, 1>(g);
}
template <int RM_, bool STATS>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2))) void gemm_nt_x6_gather_kernel(X6Args g) {
    gemm_nt_x6_body<false, true, RM_, false, 2 | (STATS ? 4 : 0)>(g);
}
template <int RM_>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2))) void gemm_nt_x6_stats_kernel(X6Args g) {
    gemm_nt_x6_body<false, true, RM_, false, 4>(g);
}

// ---------------------------------------------------------------------------------------------
// Persistent form of the f16x3 NT kernel for long products (>= 2 row tiles per resident workgroup; 128 x 256 tiles).
// A one-tile workgroup spends its life in four serial phases - in-kernel s_memtime stamps at T x 256 x 256
// (tools/x6_trace.py, profiles/r02_x6_phase_trace_onetile.txt): 11 % from entry until the first k-stage has landed, 62 % in the
// 16 k-steps, 23 % in the epilogue, 3 % waiting for its stores before it may retire - and only the co-resident
// workgroup fills the gaps.  Here 512 workgroups walk the row tiles.  The DMA ring has THREE slots: while step s
// multiplies out of slot s % 3, stages s+1 and s+2 are in flight - across tile boundaries too, so the first two
// stages of the next tile land under the epilogue of the current one, and the slot the last step has just released
// (s % 3) serves as the transpose patches (per wave a private 32 x 32 patch: no barrier inside the epilogue).  The
// stores are fire-and-forget: vmcnt retires in order and counts stores, so the only waits that could meet them are
// the ones on stages issued AFTER them - the first of which is needed two k-steps later, when they are long gone; the
// two prefetched stages are waited for (vmcnt(0), normally already there) BEFORE the first store is issued, so no
// counted wait ever spans the data-dependent number of stores of an epilogue.
// Per tile: nk step barriers + one barrier between the last step and the first patch write.
// Column sums (EPI flags BNRED / STATS): per WAVE strip of 64 rows, reduced over the wave's eight row groups by
// lane shuffles in a fixed order -> one [2][N] slab per 64 rows (alignn_gemm_nt_x6_row_tiles tells the caller).
template <bool HAS_ADD, int EPI>
__device__ __forceinline__ void gemm_nt_f16p_body(const X6Args& g) {
    constexpr bool BNRED = (EPI & 1) != 0, GATHER = (EPI & 2) != 0, STATS = (EPI & 4) != 0;
    static_assert(!(BNRED && STATS), "one set of column sums per launch");
    static_assert(!GATHER || !HAS_ADD, "the gather variant has its own two addends");
    constexpr int RM = 2, BM = Geo<2>::BM, TM = Geo<2>::TM, A_BYTES = Geo<2>::A_BYTES, A_DMA = Geo<2>::A_DMA;
    constexpr int NPL = 2, STAGE_BYTES = A_BYTES + NPL * B_PLANE, B_DMA = NPL * B_PLANE / 1024 / (NT / 64);
    constexpr int NS = 3, PIECES = A_DMA + B_DMA;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int il = lane & 31, half = lane >> 5;
    const int n_nt = g.Npad / BN;
    const int tiles = (int)((g.M + BM - 1) / BM) * n_nt;  // (< 2^31: the launcher checks)
    // Which tiles this workgroup walks: tile_first, tile_first + tile_step, ... (J of them).  Default: blockIdx.x, + grid.
    // XCD-aware map (g.xcd_map; one column tile per row, grid a multiple of 8): workgroup w runs on XCD w % 8 and every XCD
    // has its own L2, so XCD x takes the CONTIGUOUS range of row tiles [x chunk, (x+1) chunk) and its grid / 8 workgroups
    // walk it side by side.  Adjacent tiles of a line graph gather the same node rows (the in-edges of one atom, the Bd
    // row of one bond): with the round-robin map the 8 L2s each fetch all of them (gather with real indices = gather with
    // random rows: 453 vs 448 us, all rows -> row 0: 388 us, tools/gather_probe.py), with this map one L2 does.
    int tile_first = blockIdx.x, tile_step = gridDim.x, tile_end = tiles;
    if (g.xcd_map && n_nt == 1 && (gridDim.x & 7) == 0) {
        const int per = gridDim.x >> 3, xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
        const int chunk = (tiles + 7) >> 3;
        tile_first = xcd * chunk + q;
        tile_step = per;
        tile_end = (xcd + 1) * chunk < tiles ? (xcd + 1) * chunk : tiles;
    }
    const int J = tile_first < tile_end ? (tile_end - tile_first + tile_step - 1) / tile_step : 0;  // tiles of this workgroup
    if (J == 0) return;  // (uniform per workgroup; cannot happen for the shapes the launcher sends here)
    const int nk = g.K / BK;
    const int S = J * nk;  // k-steps of this workgroup, numbered across its tiles

    // ---- DMA side: the stage to issue next is (it_tile, it_kt)
    const int64_t kb_stride = (int64_t)n_nt * (NPL * B_PLANE);
    const unsigned b_lane = lane * 16;
    unsigned a_lane[A_DMA];
    const float* a_base = g.A;
    const unsigned char* b_base = g.Ws;
    int it_tile = tile_first;
    int it_kt = 0, issued = 0;
    auto set_issue_tile = [&](int tile) {
        const int64_t row0 = (int64_t)(tile / n_nt) * BM;
        a_base = g.A + row0 * g.lda;
        b_base = g.Ws + (tile % n_nt) * (NPL * B_PLANE) + (wave * B_DMA) * 1024;
#pragma unroll
        for (int i = 0; i < A_DMA; ++i) {
            const int p = (wave * A_DMA + i) * 64 + lane;
            const int row = p >> 2, c = (p & 3) ^ ((row >> 2) & 3);
            int64_t grow = row0 + row;
            if (grow >= g.M) grow = g.M - 1;  // clamp: rows past the end are computed but never stored
            a_lane[i] = (unsigned)(((grow - row0) * g.lda + c * 4) * 4);
        }
    };
    auto issue_next = [&](int slot) {
        unsigned char* stage = smem + slot * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < A_DMA; ++i) dma16<0>(a_base + it_kt * BK, a_lane[i], stage + (wave * A_DMA + i) * 1024);
#pragma unroll
        for (int i = 0; i < B_DMA; ++i)
            dma16<0>(b_base + it_kt * kb_stride + i * 1024, b_lane, stage + A_BYTES + (wave * B_DMA + i) * 1024);
        ++issued;
        if (++it_kt == nk) {
            it_kt = 0;
            it_tile += tile_step;
            if (it_tile < tile_end) set_issue_tile(it_tile);
        }
    };
    set_issue_tile(it_tile);

    // reader addresses (bytes inside a stage)
    int a_off0[RM], a_off1[RM];
#pragma unroll
    for (int a = 0; a < RM; ++a) {
        const int arow = wm * TM + a * 32 + il;
        const int a_f = (arow >> 2) & 3;
        a_off0[a] = arow * (BK * 4) + (((2 * half) ^ a_f) << 4);
        a_off1[a] = arow * (BK * 4) + (((2 * half + 1) ^ a_f) << 4);
    }
    int b_off[RN];
#pragma unroll
    for (int b = 0; b < RN; ++b) {
        const int n = wn * TN + b * 32 + il;
        b_off[b] = A_BYTES + n * (BK * 2) + ((half ^ ((n >> 3) & 1)) << 4);
    }
    const float sa = f16_scale(*g.a_amax), inv_sa = 1.0f / sa, inv_sw = 1.0f / f16_scale(*g.w_amax);

    if (0 < S) issue_next(0);
    if (1 < S) issue_next(1);

    f32x16 acc[RM][RN];
    int slot = 0;  // of the step about to run
    int tile = tile_first;
    constexpr int PLD = 32 + 4;                // patch row stride (floats)
    constexpr int PATCH_BYTES = STAGE_BYTES / (NT / 64);  // 6 KiB per wave >= 32 * PLD * 4
    static_assert(32 * PLD * 4 <= PATCH_BYTES, "patch fits its share of a slot");
    const int prow = lane >> 3, pc4 = (lane & 7) * 4;  // patch reader: 8 lanes per 32-float row, 8 rows per instruction
    for (int j = 0; j < J; ++j, tile += tile_step) {
        const int64_t m0 = (int64_t)(tile / n_nt) * BM;
        const int n0 = (tile % n_nt) * BN;
#pragma unroll
        for (int a = 0; a < RM; ++a)
#pragma unroll
            for (int b = 0; b < RN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
        // EPI GATHER: the node pair (src, dst) of the strip's row `lane`, requested now and used after the k-loop (one
        // coalesced load per tile instead of a dependent index load in front of every gather)
        int src_l = 0, dst_l = 0;
        if constexpr (GATHER) {
            int64_t row = m0 + wm * TM + lane;
            row = row < g.M ? row : g.M - 1;
            src_l = g.gsrc[row];
            dst_l = g.gdst[row];
        }
        if (j == 1) X6_STAMP(0);
#if X6_TRACE == 2
        unsigned long long ph[5] = {0, 0, 0, 0, 0}, tp = __builtin_readcyclecounter();
#define X6_PH(i)                                                    \
    do {                                                            \
        const unsigned long long tn_ = __builtin_readcyclecounter(); \
        ph[i] += tn_ - tp;                                          \
        tp = tn_;                                                   \
    } while (0)
#else
#define X6_PH(i) \
    do {         \
    } while (0)
#endif
        for (int kt = 0; kt < nk; ++kt) {
            const int s = j * nk + kt;
            // stage s has landed?  Steps 0 and 1 of a later tile were waited for before the previous epilogue.
            if (j == 0 || kt >= 2) {
                if (s + 1 < S)
                    wait_vmcnt<PIECES>();
                else
                    wait_vmcnt<0>();
            }
            X6_PH(0);
            block_barrier();  // (also: every wave has left slot (s+2) % 3 - the reads of step s-1 or the patches)
            X6_PH(1);
            if (j == 2 && kt == 0) X6_STAMP(5);
            if (j == 2 && kt == 2) X6_STAMP(6);
            if (issued < S) issue_next(slot == 0 ? 2 : slot - 1);  // stage s+2 -> slot (s+2) % 3
            X6_PH(2);
            const unsigned char* stage = smem + slot * STAGE_BYTES;
            slot = slot == NS - 1 ? 0 : slot + 1;
            f16x8 ah[RM], al[RM], bh[RN], bl[RN];
            float xs[RM][8];
#pragma unroll
            for (int a = 0; a < RM; ++a)
                slice8_f16_hi(*reinterpret_cast<const float4*>(stage + a_off0[a]),
                              *reinterpret_cast<const float4*>(stage + a_off1[a]), sa, xs[a], ah[a]);
#pragma unroll
            for (int b = 0; b < RN; ++b) {
                const unsigned char* q = stage + b_off[b];
                bh[b] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(q));
                bl[b] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(q + B_PLANE));
            }
#define X6_PASS(AA, BB)                                                                              \
    _Pragma("unroll") for (int a = 0; a < RM; ++a) _Pragma("unroll") for (int b = 0; b < RN; ++b)    \
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AA[a], BB[b], acc[a][b], 0, 0, 0);
#if X6_TRACE == 2
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            X6_PH(3);
#endif
            X6_PASS(ah, bh)
#pragma unroll
            for (int a = 0; a < RM; ++a) slice8_f16_lo(xs[a], ah[a], al[a]);
            X6_PASS(ah, bl)
            X6_PASS(al, bh)
#undef X6_PASS
#if X6_TRACE == 2
            __builtin_amdgcn_sched_barrier(0);
            X6_PH(4);
#endif
        }
#if X6_TRACE == 2
        if (j == 1 && threadIdx.x == 0 && blockIdx.x < 512)
            for (int i = 0; i < 5; ++i) x6_trace_buf[(4096 + blockIdx.x) * 8 + i] = ph[i];
#endif
        if (j == 1) X6_STAMP(1);
        // the two stages in flight belong to the next tile: have them landed before the first store goes out (see above)
        wait_vmcnt<0>();
        if (j == 1) X6_STAMP(2);
        block_barrier();  // every wave has finished reading the last stage: its slot becomes the patches
        if (j == 1) X6_STAMP(3);
        float* patch = reinterpret_cast<float*>(smem + (slot == 0 ? NS - 1 : slot - 1) * STAGE_BYTES + wave * PATCH_BYTES);
        // The epilogue has NO control flow around its memory operations (N is a multiple of 256 here; rows past the end are
        // clamped to the last row, whose values the clamped operand rows reproduce bit for bit, so the surplus stores
        // rewrite that row with what it already holds): hipcc counts vmcnt exactly only through straight-line code -
        // with predicated stores it fell back to s_waitcnt vmcnt(0) before every round's loads AND before the first LDS
        // read of the next k-loop, i.e. it drained the stores and the DMA ring.
        // Eight rounds (b = column group, a = row group) of one 32 x 32 accumulator tile per wave.  vmcnt retires in order
        // and counts stores, so the operands of round r+1 (residual / pre-activation / gathered rows, BatchNorm constants)
        // are requested BEFORE the stores of round r go out: their consumer then waits for vmcnt(4), not for those stores.
        const int last_row = (int)(g.M - 1 - m0 < BM - 1 ? g.M - 1 - m0 : BM - 1);  // last valid row of the tile (relative)
        // (the global offsets below are the same for every tile: left alone, hipcc computes all 32 of them once, in front
        // of the tile loop, and carries them through the k-loop in registers the MFMA operands need)
        int prow_e = prow, pc4_e = pc4;
        asm volatile("" : "+v"(prow_e), "+v"(pc4_e));
        float* c_t = g.C + m0 * g.ldc;
        const float* bd_tab = GATHER ? (g.gp2 ? g.gp2 : g.gp + g.N) : nullptr;  // (scalar selects: no branch in the epilogue)
        const int64_t bd_ld = GATHER ? (g.gp2 ? g.ldgp2 : g.ldgp) : 0;
        const float* add_t = HAS_ADD ? g.addend + m0 * g.ldadd : nullptr;
        const float* xn_t = BNRED ? g.xn + m0 * g.ldxn : nullptr;
        const int col0 = n0 + wn * TN + pc4_e;
        // operands of one HALF round (16 x 32 of the 32 x 32 tile: two float4 per lane and array), requested D half rounds
        // ahead of their use - the latency of these loads, not their volume, is what the fused epilogues cost
        // (tools/x6p_trace.py: 9 k cycles for the plain epilogue, 54 k with a dependent index load + gather per half round)
        constexpr int D = (BNRED && HAS_ADD) ? 1 : 2, NH = 2 * RN * RM;  // look-ahead, half rounds per tile
        float4 av[(HAS_ADD || GATHER) ? D : 1][2], xv[BNRED ? D : 1][2], nst[BNRED ? 2 : 1][3], bias_b[2], s0, s1;
        auto row_of = [&](int a, int i) {
            const int rrel = wm * TM + a * 32 + prow_e + i * 8;
            return rrel < last_row ? rrel : last_row;
        };
        auto load_half = [&](int q) {  // half round q = (b, a, h)
            const int b = q / (2 * RM), a = (q / 2) % RM, h = q & 1, col = col0 + b * 32, sl = q % D;
            if constexpr (GATHER) {
                int ui[2], vi[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {  // (lane l holds the node pair of the strip's row l: no dependent index load here)
                    const int holder = a * 32 + (2 * h + i) * 8 + prow_e;
                    ui[i] = __shfl(src_l, holder);
                    vi[i] = __shfl(dst_l, holder);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    av[sl][i] = f4_add(f4_ld(g.gp + (int64_t)ui[i] * g.ldgp + col), f4_ld(bd_tab + (int64_t)vi[i] * bd_ld + col));
            }
            if constexpr (HAS_ADD) {
#pragma unroll
                for (int i = 0; i < 2; ++i) av[sl][i] = f4_ld(add_t + (row_of(a, 2 * h + i) * (int)g.ldadd + col));
            }
            if constexpr (BNRED) {
#pragma unroll
                for (int i = 0; i < 2; ++i) xv[sl][i] = f4_lds<true>(xn_t + (row_of(a, 2 * h + i) * (int)g.ldxn + col));
            }
            if (a == 0 && h == 0) {  // (constants of a new column group)
                bias_b[b & 1] = g.bias ? f4_ld(g.bias + col) : f4_zero();
                if constexpr (BNRED)
                    nst[b & 1][0] = f4_ld(g.nstat + col), nst[b & 1][1] = f4_ld(g.nstat + 2 * g.N + col),
                              nst[b & 1][2] = f4_ld(g.nstat + 3 * g.N + col);
            }
        };
#pragma unroll
        for (int q = 0; q < D; ++q) load_half(q);
#pragma unroll
        for (int rd = 0; rd < RN * RM; ++rd) {
            const int b = rd / RM, a = rd % RM;
            const int col = col0 + b * 32;
            if (a == 0 && (BNRED || STATS)) s0 = s1 = f4_zero();
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * half) * PLD + il] = acc[a][b][r];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int q = 2 * rd + h, sl = q % D;
                float4 v[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    v[i] = f4_ld(patch + ((2 * h + i) * 8 + prow) * PLD + pc4);
                    v[i] = f4_scale(f4_scale(v[i], inv_sa), inv_sw);
                    v[i] = f4_add(v[i], bias_b[b & 1]);
                    if constexpr (HAS_ADD || GATHER) v[i] = f4_add(v[i], av[sl][i]);
                    const bool valid = wm * TM + a * 32 + prow_e + (2 * h + i) * 8 <= last_row;
                    if constexpr (STATS) {
                        const float4 u = valid ? v[i] : f4_zero();
                        s0 = f4_add(s0, u);
                        s1 = f4_fma(u, u, s1);
                    }
                    if constexpr (BNRED) {
                        const float4 xc = f4_sub(xv[sl][i], nst[b & 1][0]);
                        const float4 z = f4_fma(xc, nst[b & 1][1], nst[b & 1][2]);
                        float4 gz = make_float4(v[i].x * dsilu_f(z.x), v[i].y * dsilu_f(z.y), v[i].z * dsilu_f(z.z), v[i].w * dsilu_f(z.w));
                        gz = valid ? gz : f4_zero();
                        s0 = f4_add(s0, gz);
                        s1 = f4_fma(gz, xc, s1);  // (x rstd once, at the end)
                    }
                }
                __builtin_amdgcn_sched_barrier(0);  // (the look-ahead loads go here, not above the arithmetic: registers)
                if (q + D < NH) load_half(q + D);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i) f4_sts<X6P_NT != 0>(c_t + (row_of(a, 2 * h + i) * (int)g.ldc + col), v[i]);  // (write-once hint)
                if (a == RM - 1 && h == 1 && (BNRED || STATS)) {
                    // the wave's eight row groups (lane >> 3) hold the same four columns: fixed-order butterfly
#pragma unroll
                    for (int d = 8; d < 64; d <<= 1) {
                        s0 = f4_add(s0, make_float4(__shfl_xor(s0.x, d), __shfl_xor(s0.y, d), __shfl_xor(s0.z, d), __shfl_xor(s0.w, d)));
                        s1 = f4_add(s1, make_float4(__shfl_xor(s1.x, d), __shfl_xor(s1.y, d), __shfl_xor(s1.z, d), __shfl_xor(s1.w, d)));
                    }
                    if constexpr (BNRED) s1 = f4_mul(s1, f4_ld(g.nstat + g.N + col));  // sum gz*(x-mean) -> sum gz*xhat
                    // every lane stores (the eight row groups hold the same sums: same bits to the same place); a strip that
                    // lies entirely past the end carries zeros and goes to the scratch slab behind the valid ones (red_partial
                    // holds alignn_gemm_nt_x6_row_tiles() + 1 slabs) - again no branch around a store
                    const int64_t n_strips = (g.M + TM - 1) / TM;
                    int64_t strip = m0 / TM + wm;
                    strip = strip < n_strips ? strip : n_strips;
                    f4_st(g.red_partial + (strip * 2 + 0) * g.N + col, s0);
                    f4_st(g.red_partial + (strip * 2 + 1) * g.N + col, s1);
                }
            }
        }
        if (j == 1) X6_STAMP(4);
    }
}

template <bool HAS_ADD>
__global__ __launch_bounds__(NT) void gemm_nt_f16p_kernel(X6Args g) {
    gemm_nt_f16p_body<HAS_ADD, 0>(g);
}
template <bool HAS_ADD>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2))) void gemm_nt_f16p_bnred_kernel(X6Args g) {
    gemm_nt_f16p_body<HAS_ADD, 1>(g);
}
template <bool STATS>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2))) void gemm_nt_f16p_gather_kernel(X6Args g) {
    gemm_nt_f16p_body<false, 2 | (STATS ? 4 : 0)>(g);
}
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2))) void gemm_nt_f16p_stats_kernel(X6Args g) {
    gemm_nt_f16p_body<false, 4>(g);
}


// ---------------------------------------------------------------------------------------------
// Weight gradient on the same split product:  dW[n,k] = sum_m G[m,n] X[m,k]  (both operands are activations,
// both are sliced in registers).  The reduction index m is the ROW of both inputs, so a 16-row stage is DMA'd
// as it lies in memory ([16][128] of G, [16][256] of X, fp32) and the MFMA operands - 8 consecutive m for one
// output index per lane - are column reads: 8 x ds_read_b32 per 32-wide tile, lanes on consecutive columns
// (conflict-free without any swizzle).  256(n) x 256(k) output tile per workgroup, 8 waves of 64 x 128, the M
// rows split into slabs (fixed-order fp64 slab sum afterwards, as for the fp32 kernel); tiles of one slab are
// launched back-to-back on one XCD so the second reader of the X rows hits L2.
// ---------------------------------------------------------------------------------------------
namespace tn {
constexpr int TBN = 256, TBK = 256, TSTEP = 16;        // output tile (n x k), reduction rows per stage
constexpr int TWM = 4, TWN = 2, TNT = TWM * TWN * 64;  // 8 waves of 64(n) x 128(k): every input row is read ONCE
constexpr int TRM = 2, TRN = 4;
constexpr int TPLANE = TBN * TSTEP * 2;                // one 16-bit slice plane of one operand stage: [256][16] = 8 KiB
constexpr int TEPI = (TNT / 64) * 32 * (64 + 4) * 4;   // 68 KiB of transpose patches
template <bool F16>
struct TSch {
    static constexpr int NPL = F16 ? 2 : 3;
    static constexpr int BUF = 2 * NPL * TPLANE;       // sliced G planes + sliced X planes of one stage: 32 / 48 KiB
    static constexpr int LDS = 2 * BUF > TEPI ? 2 * BUF : TEPI;  // 68 / 96 KiB
};

struct TnArgs {
    const float* G;
    int64_t ldg;
    const float* X;
    int64_t ldx;
    float* ws;  // [splits][N][K]
    int64_t M;
    int N, K;
    int64_t chunk;  // reduction rows per slab (multiple of 16)
    int tiles_n, tiles_k, splits;
    const float* g_amax;  // f16x3 only: device scalars max|G|, max|X|
    const float* x_amax;
};

// dW tile = G_slab^T X_slab.  Both operands are activations, and the MFMA wants, per lane, 8 consecutive reduction
// rows m of ONE column - a column access of the row-major inputs.  Every thread therefore owns one column of the
// stage (threads 0-255: G, 256-511: X), fetches its 16 rows with 16 coalesced global_load_dword (a wave reads 256
// contiguous bytes of one row per instruction; two stages stay in flight in registers), slices them ONCE and writes
// the 16-bit slices to LDS k-contiguous - the layout the weight image of the NT kernel has - so that every wave then
// picks up its operands with ds_read_b128.  (The first version DMA'd fp32 rows to LDS and let each wave slice the
// columns it needed: 3x redundant slicing, and at 4 cycles per wave64 VALU instruction the slicing, not the
// matrix pipe or HBM, bound the kernel - SQ_ACTIVE_INST_VALU 29 % of all wave cycles.)
template <bool F16>
__global__ __launch_bounds__(TNT) void gemm_tn_x6_kernel(TnArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NPL = TSch<F16>::NPL, BUF = TSch<F16>::BUF;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / TWN, wn = wave % TWN;
    const int il = lane & 31, half = lane >> 5;
    // XCD grouping: workgroup L runs on XCD L % 8; give XCD c the slabs c, c+8, ... with their tiles consecutive
    const int tiles = g.tiles_n * g.tiles_k;
    const int L = blockIdx.x;
    const int z = (L & 7) + 8 * ((L >> 3) / tiles);
    if (z >= g.splits) return;
    const int tile = (L >> 3) % tiles;
    const int n0 = (tile % g.tiles_n) * TBN, k0 = (tile / g.tiles_n) * TBK;
    const int64_t rbeg = (int64_t)z * g.chunk;
    int64_t rend = rbeg + g.chunk;
    if (rend > g.M) rend = g.M;
    const int nst = rend > rbeg ? (int)((rend - rbeg + TSTEP - 1) / TSTEP) : 0;

    f32x16 acc[TRM][TRN];
#pragma unroll
    for (int a = 0; a < TRM; ++a)
#pragma unroll
        for (int b = 0; b < TRN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    // ---- loader role: one column of G (waves 0-3) or of X (waves 4-7)
    const bool is_x = wave >= 4;  // wave-uniform
    const int col = t & 255;
    const float* src = is_x ? g.X + k0 + col : g.G + n0 + col;
    const int64_t ld = is_x ? g.ldx : g.ldg;
    float scale = 1.0f, inv_scale = 1.0f;
    if constexpr (F16) {
        const float sg = f16_scale(*g.g_amax), sx = f16_scale(*g.x_amax);
        scale = is_x ? sx : sg;
        inv_scale = (1.0f / sg) * (1.0f / sx);
    }
    // sliced-stage LDS image, per operand: [plane][column 256][chunk 2, swizzled][8 x 16 bit]
    const int w_off = (is_x ? NPL * TPLANE : 0) + col * (TSTEP * 2);
    const int w_swz = (col >> 3) & 1;
    // ---- consumer role: operand addresses of this wave's tiles
    int a_off[TRM], b_off[TRN];
#pragma unroll
    for (int a = 0; a < TRM; ++a) {
        const int n = wm * 64 + a * 32 + il;
        a_off[a] = n * (TSTEP * 2) + ((half ^ ((n >> 3) & 1)) << 4);
    }
#pragma unroll
    for (int b = 0; b < TRN; ++b) {
        const int k = wn * 128 + b * 32 + il;
        b_off[b] = NPL * TPLANE + k * (TSTEP * 2) + ((half ^ ((k >> 3) & 1)) << 4);
    }

    // A stage that lies entirely inside the matrix (all but the last one of the last slab) is fetched from a wave-uniform
    // row pointer (scalar registers, advanced by one leading dimension per load) + the lane's constant column offset:
    // hipcc otherwise rebuilds a 64-bit vector address per load out of ~6 scalar multiply / add instructions
    // (SQ_ACTIVE_INST_SCA was 54 % of all issued instruction cycles of this kernel).
    const unsigned col_bytes = (unsigned)col * 4u;
    const char* sbase = reinterpret_cast<const char*>(is_x ? g.X + k0 : g.G + n0);
    const int64_t ld_bytes = ld * 4;
#define TN_LOAD(R, ST)                                                                                 \
    if ((ST) < nst && (!X6_ABL_NOALOAD || (ST) < 2)) {                                                 \
        const int64_t r0_ = rbeg + (int64_t)(ST) * TSTEP;                                              \
        if (r0_ + TSTEP <= g.M) {                                                                      \
            const char* p_ = sbase + r0_ * ld_bytes;                                                   \
            _Pragma("unroll") for (int m = 0; m < TSTEP; ++m) {                                        \
                R[m] = __builtin_nontemporal_load(reinterpret_cast<const float*>(p_ + col_bytes));     \
                p_ += ld_bytes;                                                                        \
            }                                                                                          \
        } else {                                                                                       \
            _Pragma("unroll") for (int m = 0; m < TSTEP; ++m) {                                        \
                int64_t row = r0_ + m;                                                                 \
                row = row < g.M ? row : g.M - 1; /* clamped; masked when sliced */                     \
                R[m] = __builtin_nontemporal_load(src + row * ld);                                     \
            }                                                                                          \
        }                                                                                              \
    }
    // slice the landed stage ST (registers R) into the LDS image `buf`
#define TN_SLICE(R, ST, buf)                                                                                    \
    {                                                                                                           \
        const int valid = (int)((rend - (rbeg + (int64_t)(ST) * TSTEP)) < TSTEP ? (rend - (rbeg + (int64_t)(ST) * TSTEP)) : TSTEP); \
        if (valid < TSTEP) { /* (only the last stage of a slab can be short: mask it once, not per use) */      \
            _Pragma("unroll") for (int m = 0; m < TSTEP; ++m) R[m] = m < valid ? R[m] : 0.0f;                   \
        }                                                                                                       \
        _Pragma("unroll") for (int c = 0; c < 2; ++c) {                                                         \
            float x[8];                                                                                         \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) x[j] = R[8 * c + j];                                  \
            unsigned char* dst = (buf) + w_off + ((c ^ w_swz) << 4);                                            \
            if constexpr (F16) {                                                                                \
                f16x8 h, l;                                                                                     \
                slice8_f16(make_float4(x[0], x[1], x[2], x[3]), make_float4(x[4], x[5], x[6], x[7]), scale, h, l); \
                *reinterpret_cast<f16x8*>(dst) = h;                                                             \
                *reinterpret_cast<f16x8*>(dst + TPLANE) = l;                                                    \
            } else {                                                                                            \
                bf16x8 h, mm, l;                                                                                \
                slice8(make_float4(x[0], x[1], x[2], x[3]), make_float4(x[4], x[5], x[6], x[7]), h, mm, l);     \
                *reinterpret_cast<bf16x8*>(dst) = h;                                                            \
                *reinterpret_cast<bf16x8*>(dst + TPLANE) = mm;                                                  \
                *reinterpret_cast<bf16x8*>(dst + 2 * TPLANE) = l;                                               \
            }                                                                                                   \
        }                                                                                                       \
    }
#define TN_PASS(TY, AA, BB)                                                                            \
    _Pragma("unroll") for (int a = 0; a < TRM; ++a) _Pragma("unroll") for (int b = 0; b < TRN; ++b)    \
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_##TY(AA[a], BB[b], acc[a][b], 0, 0, 0);
    // operands of this wave's tiles out of the LDS image `buf` (f16x3), products deferred to TN_MMA_F16
#define TN_READ_F16(buf)                                                                                        \
    f16x8 ah[TRM], al[TRM], bh[TRN], bl[TRN];                                                                   \
    _Pragma("unroll") for (int a = 0; a < TRM; ++a) {                                                           \
        ah[a] = *reinterpret_cast<const f16x8*>((buf) + a_off[a]);                                              \
        al[a] = *reinterpret_cast<const f16x8*>((buf) + a_off[a] + TPLANE);                                     \
    }                                                                                                           \
    _Pragma("unroll") for (int b = 0; b < TRN; ++b) {                                                           \
        bh[b] = *reinterpret_cast<const f16x8*>((buf) + b_off[b]);                                              \
        bl[b] = *reinterpret_cast<const f16x8*>((buf) + b_off[b] + TPLANE);                                     \
    }
#define TN_MMA_F16                                                                                              \
    if (!X6_ABL_ONEMFMA) { TN_PASS(f16, al, bh) TN_PASS(f16, ah, bl) }                                          \
    TN_PASS(f16, ah, bh)
#define TN_MMA_BF16(buf)                                                                                        \
    {                                                                                                           \
        bf16x8 ah[TRM], am[TRM], al[TRM], bh[TRN], bm[TRN], bl[TRN];                                            \
        _Pragma("unroll") for (int a = 0; a < TRM; ++a) {                                                       \
            ah[a] = *reinterpret_cast<const bf16x8*>((buf) + a_off[a]);                                         \
            am[a] = *reinterpret_cast<const bf16x8*>((buf) + a_off[a] + TPLANE);                                \
            al[a] = *reinterpret_cast<const bf16x8*>((buf) + a_off[a] + 2 * TPLANE);                            \
        }                                                                                                       \
        _Pragma("unroll") for (int b = 0; b < TRN; ++b) {                                                       \
            bh[b] = *reinterpret_cast<const bf16x8*>((buf) + b_off[b]);                                         \
            bm[b] = *reinterpret_cast<const bf16x8*>((buf) + b_off[b] + TPLANE);                                \
            bl[b] = *reinterpret_cast<const bf16x8*>((buf) + b_off[b] + 2 * TPLANE);                            \
        }                                                                                                       \
        TN_PASS(bf16, al, bh) TN_PASS(bf16, ah, bl) TN_PASS(bf16, am, bm)                                       \
        TN_PASS(bf16, am, bh) TN_PASS(bf16, ah, bm) TN_PASS(bf16, ah, bh)                                       \
    }

    if constexpr (F16 && TN_PIPE) {
        // Software-pipelined stage loop (f16x3): THREE stages of this thread's column in flight in registers; the slices
        // of stage s+1 are computed and written to the other LDS buffer in the shadow of the products of stage s (VALU
        // and LDS-write instructions issue while the matrix pipe works through the 24 products), one barrier per stage.
        // Same arithmetic in the same order as the two-stage loop: bit-identical.
        float r0[TSTEP], r1[TSTEP], r2[TSTEP];
        TN_LOAD(r0, 0)
        TN_LOAD(r1, 1)
        TN_LOAD(r2, 2)
        if (nst > 0) {
            if (nst > 2) wait_vmcnt<2 * TSTEP>(); else if (nst > 1) wait_vmcnt<TSTEP>(); else wait_vmcnt<0>();
            TN_SLICE(r0, 0, smem)
        }
#define TN_PIPE_STEP(RNEXT, RFREE, ST)                                                                          \
    if ((ST) < nst) {                                                                                           \
        if ((ST) + 2 < nst) wait_vmcnt<TSTEP>(); else wait_vmcnt<0>(); /* stage ST+1 has landed */              \
        block_barrier(); /* slices of stage ST visible; every wave is past its reads of stage ST-1 */          \
        unsigned char* cur = smem + ((ST) & 1) * BUF;                                                           \
        unsigned char* nxt = smem + (((ST) + 1) & 1) * BUF;                                                     \
        TN_READ_F16(cur)                                                                                        \
        if (!X6_ABL_ONEMFMA) { TN_PASS(f16, al, bh) }                                                           \
        if ((ST) + 1 < nst) TN_SLICE(RNEXT, (ST) + 1, nxt)                                                      \
        if (!X6_ABL_ONEMFMA) { TN_PASS(f16, ah, bl) }                                                           \
        TN_PASS(f16, ah, bh)                                                                                    \
        TN_LOAD(RFREE, (ST) + 3)                                                                                \
    }
        for (int st = 0; st < nst; st += 3) {
            TN_PIPE_STEP(r1, r0, st)
            TN_PIPE_STEP(r2, r1, st + 1)
            TN_PIPE_STEP(r0, r2, st + 2)
        }
#undef TN_PIPE_STEP
    } else {
        float ra[TSTEP], rb[TSTEP];  // two stages of this thread's column in flight
        // slice the landed stage into the LDS image, refill the registers with stage ST+2, hand over, multiply
#define TN_STEP(R, ST, BUFI)                                                                                    \
    if ((ST) < nst) {                                                                                           \
        if ((ST) + 1 < nst)                                                                                     \
            wait_vmcnt<TSTEP>();                                                                                \
        else                                                                                                    \
            wait_vmcnt<0>();                                                                                    \
        unsigned char* buf = smem + (BUFI) * BUF;                                                               \
        TN_SLICE(R, ST, buf)                                                                                    \
        TN_LOAD(R, (ST) + 2)                                                                                    \
        block_barrier(); /* slices of stage ST visible; every wave is past its reads of the other buffer */    \
        if constexpr (F16) {                                                                                    \
            TN_READ_F16(buf)                                                                                    \
            TN_MMA_F16                                                                                          \
        } else {                                                                                                \
            TN_MMA_BF16(buf)                                                                                    \
        }                                                                                                       \
    }
        TN_LOAD(ra, 0)
        TN_LOAD(rb, 1)
        for (int st = 0; st < nst; st += 2) {
            TN_STEP(ra, st, 0)
            TN_STEP(rb, st + 1, 1)
        }
#undef TN_STEP
    }
#undef TN_MMA_BF16
#undef TN_MMA_F16
#undef TN_READ_F16
#undef TN_PASS
#undef TN_SLICE
#undef TN_LOAD

    // epilogue: slab z of the workspace, rows n, cols k; per-wave LDS transpose -> float4 row segments
    constexpr int PLD = 64 + 4;
    float* patch = reinterpret_cast<float*>(smem) + wave * (32 * PLD);
    const int prow = lane >> 4, pc4 = (lane & 15) * 4;
    float* out = g.ws + (int64_t)z * g.N * g.K;
#pragma unroll
    for (int ahb = 0; ahb < TRM * (TRN / 2); ++ahb) {
        const int a = ahb / (TRN / 2), hb = ahb % (TRN / 2);
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                patch[((r & 3) + 8 * (r >> 2) + 4 * half) * PLD + b * 32 + il] = acc[a][2 * hb + b][r];
        __syncthreads();
        float4 ov[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ov[i] = f4_ld(patch + (i * 4 + prow) * PLD + pc4);
        const int ocol = k0 + wn * 128 + hb * 64 + pc4;
        const int row0 = n0 + wm * 64 + a * 32 + prow;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float4 v = ov[i];
            if constexpr (F16) v = f4_scale(v, inv_scale);
            f4_st(out + (int64_t)(row0 + i * 4) * g.K + ocol, v);
        }
    }
}

inline int64_t tn_chunk(int64_t M, int N, int K) {
    const int64_t tiles = (int64_t)(N / TBN) * (K / TBK);
    int64_t want = (256 + tiles - 1) / tiles;  // one workgroup per CU (8 waves fill it); fewer slabs to sum
    if (want < 1) want = 1;
    int64_t chunk = (M + want - 1) / want;
    chunk = ((chunk + TSTEP - 1) / TSTEP) * TSTEP;
    if (chunk < 8 * TSTEP) chunk = 8 * TSTEP;
    return chunk;
}
inline int tn_splits(int64_t M, int N, int K) {
    const int64_t c = tn_chunk(M, N, K);
    return (int)((M + c - 1) / c);
}
}  // namespace tn

// Slice W (or W^T) into the kernel's DMA image: out[kb][n/256][plane][n%256][chunk ^ ((n>>3)&1)][8], n < Npad
// (zero rows beyond N), kb = k/16, chunk = (k%16)/8.
__global__ void split_bf16x3_kernel(const float* __restrict__ W, int64_t ldw, int N, int Npad, int K, int transpose,
                                    unsigned short* __restrict__ out) {
    const int64_t total = (int64_t)Npad * K;
    const int ntiles = Npad / BN;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / K), k = (int)(i % K);
        float x = 0.0f;
        if (n < N) x = transpose ? W[(int64_t)k * ldw + n] : W[(int64_t)n * ldw + k];
        const float r1 = x - trunc16(x);
        const float r2 = r1 - trunc16(r1);
        const int kb = k / BK, c = (k % BK) >> 3, e = k & 7;
        const int nt = n / BN, nin = n % BN;
        constexpr int plane = BN * BK;  // bf16 elements per slice plane of one (kb, n-tile)
        const int64_t o = ((int64_t)kb * ntiles + nt) * (3 * plane) + nin * BK + ((c ^ ((n >> 3) & 1)) << 3) + e;
        out[o] = (unsigned short)(__float_as_uint(x) >> 16);
        out[o + plane] = (unsigned short)(__float_as_uint(r1) >> 16);
        out[o + 2 * plane] = (unsigned short)(__float_as_uint(r2) >> 16);
    }
}

// fp16 two-slice image of W * 2^s (s from max|W|): same index map, two planes
__global__ void split_f16x2_kernel(const float* __restrict__ W, int64_t ldw, int N, int Npad, int K, int transpose,
                                   const float* __restrict__ w_amax, _Float16* __restrict__ out) {
    const int64_t total = (int64_t)Npad * K;
    const int ntiles = Npad / BN;
    const float sw = f16_scale(*w_amax);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / K), k = (int)(i % K);
        float x = 0.0f;
        if (n < N) x = transpose ? W[(int64_t)k * ldw + n] : W[(int64_t)n * ldw + k];
        x *= sw;
        const _Float16 h = (_Float16)x;
        const _Float16 l = (_Float16)(x - (float)h);
        const int kb = k / BK, c = (k % BK) >> 3, e = k & 7;
        const int nt = n / BN, nin = n % BN;
        constexpr int plane = BN * BK;
        const int64_t o = ((int64_t)kb * ntiles + nt) * (2 * plane) + nin * BK + ((c ^ ((n >> 3) & 1)) << 3) + e;
        out[o] = h;
        out[o + plane] = l;
    }
}

// fp16 two-slice images of W (forward products) and of W^T (input-gradient products) in one launch: the two index spaces
// of split_f16x2_kernel walked by one grid (same images, bit for bit)
__global__ void split_f16x2_both_kernel(const float* __restrict__ W, int64_t ldw, int N, int K, const float* __restrict__ w_amax,
                                        _Float16* __restrict__ out, _Float16* __restrict__ outT) {
    const float sw = f16_scale(*w_amax);
    constexpr int plane = BN * BK;
#pragma unroll
    for (int tr = 0; tr < 2; ++tr) {  // tr == 1: the image of W^T, a [K, N] matrix
        _Float16* o = tr ? outT : out;
        const int n_ = tr ? K : N, k_ = tr ? N : K, np = ((n_ + BN - 1) / BN) * BN, ntiles = np / BN;
        const int total = np * k_;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
            const int n = i / k_, k = i % k_;
            float x = 0.0f;
            if (n < n_) x = tr ? W[(int64_t)k * ldw + n] : W[(int64_t)n * ldw + k];
            x *= sw;
            const _Float16 h = (_Float16)x;
            const _Float16 l = (_Float16)(x - (float)h);
            const int kb = k / BK, c = (k % BK) >> 3, e = k & 7;
            const int nt = n / BN, nin = n % BN;
            const int64_t q = ((int64_t)kb * ntiles + nt) * (2 * plane) + nin * BK + ((c ^ ((n >> 3) & 1)) << 3) + e;
            o[q] = h;
            o[q + plane] = l;
        }
    }
}

// max|X| of a row-major [rows, F] matrix (F % 4 == 0): one atomicMax per workgroup on the bit pattern
// (non-negative floats order like unsigned ints; max is order independent, so this is deterministic)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ X, int64_t ldx, int64_t rows, int F,
                                                      float* __restrict__ amax) {
    const int Q = F >> 2;
    const RowQuad rq(Q);
    const int64_t total = rows * Q;
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t r;
        int q;
        rq.split(i, total, r, q);
        const float4 v = f4_ld(X + r * ldx + q * 4);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    block_amax_commit(m, amax);
}

// ---- all split-product weights of a model in two launches (alignn_prepare_weights): one descriptor per weight
struct WeightDesc {
    const float* W;      // [N, K], leading dimension ldw
    int64_t ldw;
    int32_t N, K;
    float* amax;         // max|W| slot: slot i of the `amax` array alignn_prepare_weights zeroes first
    _Float16* out;       // image of W    (alignn_split_f16x2_bytes(N, K))
    _Float16* outT;      // image of W^T  (alignn_split_f16x2_bytes(K, N))
};
static_assert(sizeof(WeightDesc) == 48, "descriptor layout is part of the C ABI (alignn_prepare_weights)");

__global__ void zero_floats_kernel(float* __restrict__ p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0f;
}

__global__ __launch_bounds__(256) void absmax_batched_kernel(const WeightDesc* __restrict__ descs) {
    const WeightDesc d = descs[blockIdx.y];
    const int Q = d.K >> 2;
    const int64_t total = (int64_t)d.N * Q;
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / Q;
        const int q = (int)(i - r * Q);
        const float4 v = f4_ld(d.W + r * d.ldw + q * 4);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    block_amax_commit(m, d.amax);
}

__global__ __launch_bounds__(256) void split_f16x2_both_batched_kernel(const WeightDesc* __restrict__ descs) {
    const WeightDesc d = descs[blockIdx.y];
    const float sw = f16_scale(*d.amax);
    constexpr int plane = BN * BK;
    const int N = d.N, K = d.K;
#pragma unroll
    for (int tr = 0; tr < 2; ++tr) {  // same element map as split_f16x2_both_kernel: bit-identical images
        _Float16* o = tr ? d.outT : d.out;
        const int n_ = tr ? K : N, k_ = tr ? N : K, np = ((n_ + BN - 1) / BN) * BN, ntiles = np / BN;
        const int total = np * k_;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
            const int n = i / k_, k = i % k_;
            float x = 0.0f;
            if (n < n_) x = tr ? d.W[(int64_t)k * d.ldw + n] : d.W[(int64_t)n * d.ldw + k];
            x *= sw;
            const _Float16 h = (_Float16)x;
            const _Float16 l = (_Float16)(x - (float)h);
            const int kb = k / BK, c = (k % BK) >> 3, e = k & 7;
            const int nt = n / BN, nin = n % BN;
            const int64_t q = ((int64_t)kb * ntiles + nt) * (2 * plane) + nin * BK + ((c ^ ((n >> 3) & 1)) << 3) + e;
            o[q] = h;
            o[q + plane] = l;
        }
    }
}

inline bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int npad(int N) { return ((N + BN - 1) / BN) * BN; }

}  // namespace

namespace {
template <bool F16, int RM_>
int launch_nt_rm(const X6Args& g, hipStream_t st) {
    constexpr int lds = Sch<F16, RM_>::LDS;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_x6_kernel<false, F16, RM_>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)gemm_nt_x6_kernel<true, F16, RM_>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid(alignn_ceil_div(g.M, Geo<RM_>::BM), g.Npad / BN);
#if X6_PERSIST
    // long products (>= 4 tiles per resident workgroup): 512 persistent workgroups (two per CU) walk the row tiles
    if constexpr (RM_ == 2 && Sch<F16, RM_>::NSTAGE == 2) {
        constexpr int kResident = 512;
        const int ny = g.Npad / BN;
        // (without addend only: the addend variant needs 292 registers per lane in this form - one wave per SIMD)
        if (!g.addend && !g.red_partial && !g.gp && g.N == g.Npad && ((g.K / BK) & 1) == 0 && (int64_t)grid.x * ny >= 4 * kResident &&
            ny <= kResident) {
            constexpr int plds = Sch<F16, RM_>::LDS_PERSIST;
            static bool pattr_set = false;
            if (!pattr_set) {
                hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_x6_kernel<false, F16, RM_, true>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, plds);
                if (e != hipSuccess) return (int)e;
                pattr_set = true;
            }
            hipLaunchKernelGGL((gemm_nt_x6_kernel<false, F16, RM_, true>), dim3(kResident / ny, ny), dim3(NT), plds,
                               st, g);
            ALIGNN_CHECK_LAUNCH();
            return 0;
        }
    }
#endif
    if constexpr (F16) {
        constexpr int slds = Sch<F16, RM_>::LDS > EPI1_LDS ? Sch<F16, RM_>::LDS : EPI1_LDS;  // + the column-sum slots
        static bool fattr_set = false;
        if (!fattr_set) {
            hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_x6_gather_kernel<RM_, false>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)gemm_nt_x6_gather_kernel<RM_, true>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, slds);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)gemm_nt_x6_stats_kernel<RM_>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, slds);
            if (e != hipSuccess) return (int)e;
            fattr_set = true;
        }
        if (g.gp != nullptr) {  // gather-add of the two node rows in the epilogue (+ column statistics of the result)
            if (g.red_partial)
                hipLaunchKernelGGL((gemm_nt_x6_gather_kernel<RM_, true>), grid, dim3(NT), slds, st, g);
            else
                hipLaunchKernelGGL((gemm_nt_x6_gather_kernel<RM_, false>), grid, dim3(NT), lds, st, g);
            ALIGNN_CHECK_LAUNCH();
            return 0;
        }
        if (g.red_partial != nullptr && g.xn == nullptr) {  // column statistics of the output only
            if (g.addend) return (int)hipErrorInvalidValue;
            hipLaunchKernelGGL((gemm_nt_x6_stats_kernel<RM_>), grid, dim3(NT), slds, st, g);
            ALIGNN_CHECK_LAUNCH();
            return 0;
        }
    }
    if constexpr (F16) {
        if (g.red_partial != nullptr && g.xn != nullptr) {  // BatchNorm-backward reductions in the epilogue (EPI == 1)
            constexpr int lds = Sch<F16, RM_>::LDS > EPI1_LDS ? Sch<F16, RM_>::LDS : EPI1_LDS;
            static bool eattr_set = false;
            if (!eattr_set) {
                hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_x6_bnred_kernel<false, RM_>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                if (e == hipSuccess)
                    e = hipFuncSetAttribute((const void*)gemm_nt_x6_bnred_kernel<true, RM_>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                if (e != hipSuccess) return (int)e;
                eattr_set = true;
            }
            if (g.addend)
                hipLaunchKernelGGL((gemm_nt_x6_bnred_kernel<true, RM_>), grid, dim3(NT), lds, st, g);
            else
                hipLaunchKernelGGL((gemm_nt_x6_bnred_kernel<false, RM_>), grid, dim3(NT), lds, st, g);
            ALIGNN_CHECK_LAUNCH();
            return 0;
        }
    }
    if (g.addend)
        hipLaunchKernelGGL((gemm_nt_x6_kernel<true, F16, RM_>), grid, dim3(NT), lds, st, g);
    else
        hipLaunchKernelGGL((gemm_nt_x6_kernel<false, F16, RM_>), grid, dim3(NT), lds, st, g);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}
// ---- persistent f16x3 kernel (gemm_nt_f16p_body): long products only
#ifndef X6_NO_PERSIST
#define X6_NO_PERSIST 0  // 1: never dispatch the persistent kernel (A/B in tools/ablate_x6.py)
#endif
constexpr int kResidentP = 512;  // two workgroups per CU
inline bool nt_persistent(int64_t M, int N, int K) {
#if X6_NO_PERSIST || defined(X6_FORCE_RM)
    return false;
#endif
    const char* e = getenv("ALIGNN_AMD_X6_PERSIST");  // =0: one-tile kernels everywhere (read per call: A/B runs, tests)
    if (e != nullptr && e[0] == '0') return false;
    const int64_t tiles = alignn_ceil_div(M, 128) * (int64_t)(npad(N) / BN);
    return K > 64 && K / BK >= 3 && N % BN == 0 && tiles >= 2 * kResidentP && tiles < ((int64_t)1 << 30);
}
int launch_nt_p(const X6Args& g_in, hipStream_t st) {
    constexpr int lds = 3 * (Geo<2>::A_BYTES + 2 * B_PLANE);  // 72 KiB: two workgroups per CU
    static bool attr_set = false;
    if (!attr_set) {
        const void* fns[] = {(const void*)gemm_nt_f16p_kernel<false>,        (const void*)gemm_nt_f16p_kernel<true>,
                             (const void*)gemm_nt_f16p_bnred_kernel<false>,  (const void*)gemm_nt_f16p_bnred_kernel<true>,
                             (const void*)gemm_nt_f16p_gather_kernel<false>, (const void*)gemm_nt_f16p_gather_kernel<true>,
                             (const void*)gemm_nt_f16p_stats_kernel};
        for (const void* fn : fns) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return (int)e;
        }
        attr_set = true;
    }
    if (g_in.ldc >= (1 << 20) || g_in.ldadd >= (1 << 20) || g_in.ldxn >= (1 << 20)) return (int)hipErrorInvalidValue;
    X6Args g = g_in;
    g.xcd_map = 1;  // (each XCD a contiguous range of row tiles; round-robin measured the same: profiles/README.md round 3)
    const int64_t tiles = alignn_ceil_div(g.M, 128) * (int64_t)(g.Npad / BN);
    // (fewer resident workgroups than two per CU - leaving room for the other lane's small kernels - measured slower in
    // round 3: 256 workgroups 16.05 vs 15.68 ms per step, 384: 15.93)
    constexpr int resident = kResidentP;
    const dim3 grid((unsigned)(tiles < resident ? tiles : resident)), block(NT);
    if (g.gp != nullptr) {
        if (g.red_partial)
            hipLaunchKernelGGL(gemm_nt_f16p_gather_kernel<true>, grid, block, lds, st, g);
        else
            hipLaunchKernelGGL(gemm_nt_f16p_gather_kernel<false>, grid, block, lds, st, g);
    } else if (g.red_partial != nullptr && g.xn == nullptr) {
        if (g.addend) return (int)hipErrorInvalidValue;
        hipLaunchKernelGGL(gemm_nt_f16p_stats_kernel, grid, block, lds, st, g);
    } else if (g.red_partial != nullptr) {
        if (g.addend)
            hipLaunchKernelGGL(gemm_nt_f16p_bnred_kernel<true>, grid, block, lds, st, g);
        else
            hipLaunchKernelGGL(gemm_nt_f16p_bnred_kernel<false>, grid, block, lds, st, g);
    } else if (g.addend) {
        hipLaunchKernelGGL(gemm_nt_f16p_kernel<true>, grid, block, lds, st, g);
    } else {
        hipLaunchKernelGGL(gemm_nt_f16p_kernel<false>, grid, block, lds, st, g);
    }
    ALIGNN_CHECK_LAUNCH();
    return 0;
}
// rows per reduction slab of the f16x3 kernels' column sums (EPI flags BNRED / STATS): the block tile of the one-tile
// kernels, the 64-row wave strip of the persistent one
// a product with fewer 128-row tiles than this runs on 64-row tiles: fewer than one per CU (512 - the bond-row products, 397
// tiles = one thin generation, on 793 half tiles - measured slower on every variant in round 3)
inline int64_t rm1_below() { return 256; }
inline int nt_block_rows(int64_t M, int N, int K) {
#ifdef X6_FORCE_RM
    return 64 * X6_FORCE_RM;
#endif
    if (nt_persistent(M, N, K)) return 64;
    const int64_t tiles128 = alignn_ceil_div(M, 128) * (int64_t)(npad(N) / BN);
    return (K <= 64 || tiles128 < rm1_below()) ? 64 : 128;
}
template <bool F16>
int launch_nt(const X6Args& g, hipStream_t st) {
    // 64-row tiles for shallow products and for products too short to give every CU one 128-row tile
    const int64_t tiles128 = alignn_ceil_div(g.M, 128) * (int64_t)(g.Npad / BN);
#ifdef X6_FORCE_RM  // (tools/ablate_x6.py)
    return launch_nt_rm<F16, X6_FORCE_RM>(g, st);
#endif
    if constexpr (F16)
        if (nt_persistent(g.M, g.N, g.K)) {
            // measured per variant at T x 256 x 256, kernels interleaved (tools/x6_family_check.py): persistent -5 % plain,
            // -4 % statistics, -7..-9 % gather (+ statistics), -2 % BatchNorm-backward sums; +3 % with an addend and +10 %
            // for sums + addend (its operand look-ahead spills) - those two stay on the one-tile kernel, with strip slabs
            if (g.addend == nullptr) return launch_nt_p(g, st);
            X6Args gs = g;
            gs.strip_slabs = 1;
            return launch_nt_rm<F16, 2>(gs, st);
        }
    if (g.K <= 64 || tiles128 < rm1_below()) return launch_nt_rm<F16, 1>(g, st);
    return launch_nt_rm<F16, 2>(g, st);
}
inline bool nt_args_ok(const float* A, int64_t lda, const void* Wsplit, const float* bias, const float* addend,
                       int64_t ldadd, float* C, int64_t ldc) {
    return !((lda & 3) || (ldc & 3) || !a16(A) || !a16(C) || !a16(Wsplit) || (bias && !a16(bias)) ||
             (addend && ((ldadd & 3) || !a16(addend))));
}
}  // namespace

#if X6_TRACE
extern "C" int alignn_x6_trace_read(void* host, size_t bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(x6_trace_buf), bytes < sizeof(x6_trace_buf) ? bytes : sizeof(x6_trace_buf));
}
#endif

extern "C" {

size_t alignn_split_bf16x3_bytes(int N, int K) { return (size_t)3 * npad(N) * (size_t)K * 2; }

int alignn_split_bf16x3(const float* W, int64_t ldw, int N, int K, int transpose, void* out, alignn_stream_t stream) {
    if (N <= 0 || K <= 0 || (K % BK) != 0 || out == nullptr) return (int)hipErrorInvalidValue;
    const int64_t total = (int64_t)npad(N) * K;
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, W, ldw, N, npad(N), K,
                       transpose, (unsigned short*)out);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

size_t alignn_split_f16x2_bytes(int N, int K) { return (size_t)2 * npad(N) * (size_t)K * 2; }

int alignn_split_f16x2(const float* W, int64_t ldw, int N, int K, int transpose, const float* w_amax, void* out,
                       alignn_stream_t stream) {
    if (N <= 0 || K <= 0 || (K % BK) != 0 || out == nullptr || w_amax == nullptr) return (int)hipErrorInvalidValue;
    const int64_t total = (int64_t)npad(N) * K;
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(split_f16x2_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, W, ldw, N, npad(N), K,
                       transpose, w_amax, (_Float16*)out);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_split_f16x2_both(const float* W, int64_t ldw, int N, int K, const float* w_amax, void* out, void* out_t,
                            alignn_stream_t stream) {
    if (N <= 0 || K <= 0 || (K % BK) != 0 || (N % BK) != 0 || out == nullptr || out_t == nullptr || w_amax == nullptr ||
        (int64_t)npad(N) * K > ((int64_t)1 << 28) || (int64_t)npad(K) * N > ((int64_t)1 << 28))
        return (int)hipErrorInvalidValue;
    const int64_t total = (int64_t)npad(N) * K > (int64_t)npad(K) * N ? (int64_t)npad(N) * K : (int64_t)npad(K) * N;
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(split_f16x2_both_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, W, ldw, N, K, w_amax,
                       (_Float16*)out, (_Float16*)out_t);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

/* max|W| and both slice images (W for the forward products, W^T for the input gradients) of EVERY split-product weight of a
 * model in three stream operations (a memset of the max|W| slots + two launches) instead of two launches per weight:
 * `descs` = n_weights device records {const float* W; int64 ldw; int32 N, K; float* amax; void* out; void* out_t}
 * (48 bytes each), `amax_slots` = the n_weights floats the records' `amax` point into.  Images are bit-identical to
 * alignn_split_f16x2_both's. */
int alignn_prepare_weights(const void* descs, int n_weights, float* amax_slots, alignn_stream_t stream) {
    if (n_weights == 0) return 0;
    if (descs == nullptr || n_weights < 0 || amax_slots == nullptr) return (int)hipErrorInvalidValue;
    // (a kernel, not hipMemsetAsync: inside a hipGraph capture the memset node was observed to race with the launch behind
    // it - force training replayed with stale maxima, profiles/README.md round 3)
    hipLaunchKernelGGL(zero_floats_kernel, dim3(alignn_ceil_div(n_weights, 256)), dim3(256), 0, (hipStream_t)stream, amax_slots,
                       n_weights);
    hipLaunchKernelGGL(absmax_batched_kernel, dim3(32, n_weights), dim3(256), 0, (hipStream_t)stream, (const WeightDesc*)descs);
    hipLaunchKernelGGL(split_f16x2_both_batched_kernel, dim3(256, n_weights), dim3(256), 0, (hipStream_t)stream,
                       (const WeightDesc*)descs);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

/* *amax = max(*amax, max|X|): alignn_absmax without the reset - for a slot the caller knows to hold 0 (or a bound to keep) */
int alignn_absmax_raise(const float* X, int64_t ldx, int64_t rows, int F, float* amax, alignn_stream_t stream) {
    if (F <= 0 || (F & 3) || (ldx & 3) || rows < 0 || amax == nullptr || !a16(X)) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    int64_t blocks = (rows * (F >> 2) + 256 * 8 - 1) / (256 * 8);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(absmax_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, X, ldx, rows, F, amax);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_absmax(const float* X, int64_t ldx, int64_t rows, int F, float* amax, alignn_stream_t stream) {
    if (F <= 0 || (F & 3) || (ldx & 3) || rows < 0 || amax == nullptr || !a16(X)) return (int)hipErrorInvalidValue;
    // (zeroed by a kernel, not hipMemsetAsync: see alignn_prepare_weights)
    hipLaunchKernelGGL(zero_floats_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, amax, 1);
    if (rows == 0) return 0;
    int64_t blocks = (rows * (F >> 2) + 256 * 8 - 1) / (256 * 8);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(absmax_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, X, ldx, rows, F, amax);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_gemm_tn_x6_supported(int64_t M, int N, int K) {
    return (M >= 4096 && N % tn::TBN == 0 && K % tn::TBK == 0) ? 1 : 0;
}

size_t alignn_gemm_tn_x6_workspace(int64_t M, int N, int K) {
    return (size_t)tn::tn_splits(M, N, K) * (size_t)N * (size_t)K * sizeof(float);
}

int alignn_gemm_tn_x6_splits(int64_t M, int N, int K) { return tn::tn_splits(M, N, K); }

/* slab partials only: ws[z][N][K] for z < alignn_gemm_tn_x6_splits(); the caller sums the slabs */
int alignn_gemm_tn_x6_partials(const float* G, int64_t ldg, const float* g_amax, const float* X, int64_t ldx,
                               const float* x_amax, int64_t M, int N, int K, void* workspace, size_t workspace_bytes,
                               alignn_stream_t stream) {
    if (!alignn_gemm_tn_x6_supported(M, N, K)) return (int)hipErrorInvalidValue;
    if ((ldg & 3) || (ldx & 3) || !a16(G) || !a16(X) || !a16(workspace) ||
        workspace_bytes < alignn_gemm_tn_x6_workspace(M, N, K))
        return (int)hipErrorInvalidValue;
    if ((g_amax == nullptr) != (x_amax == nullptr)) return (int)hipErrorInvalidValue;
    tn::TnArgs g{G, ldg, X, ldx, (float*)workspace, M, N, K, tn::tn_chunk(M, N, K), N / tn::TBN, K / tn::TBK,
                 tn::tn_splits(M, N, K), g_amax, x_amax};
    static bool tn_attr = false;
    if (!tn_attr && true) {
        hipError_t e = hipFuncSetAttribute((const void*)tn::gemm_tn_x6_kernel<false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, tn::TSch<false>::LDS);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)tn::gemm_tn_x6_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    tn::TSch<true>::LDS);
        if (e != hipSuccess) return (int)e;
        tn_attr = true;
    }
    const int tiles = g.tiles_n * g.tiles_k;
    const int per_xcd = alignn_ceil_div(g.splits, 8) * tiles;
    if (g_amax)
        hipLaunchKernelGGL(tn::gemm_tn_x6_kernel<true>, dim3(per_xcd * 8), dim3(tn::TNT), tn::TSch<true>::LDS, (hipStream_t)stream, g);
    else
        hipLaunchKernelGGL(tn::gemm_tn_x6_kernel<false>, dim3(per_xcd * 8), dim3(tn::TNT), tn::TSch<false>::LDS, (hipStream_t)stream, g);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_gemm_nt_x6_supported(int64_t M, int N, int K) { return (M > 0 && N >= 128 && N % 4 == 0 && K % BK == 0) ? 1 : 0; }

int alignn_gemm_nt_x6(const float* A, int64_t lda, const void* Wsplit, const float* bias, const float* addend,
                      int64_t ldadd, float* C, int64_t ldc, int64_t M, int N, int K, alignn_stream_t stream) {
    if (!alignn_gemm_nt_x6_supported(M, N, K)) return (int)hipErrorInvalidValue;
    if (!nt_args_ok(A, lda, Wsplit, bias, addend, ldadd, C, ldc)) return (int)hipErrorInvalidValue;
    X6Args g{A, lda, (const unsigned char*)Wsplit, nullptr, nullptr, bias, addend, ldadd, C, ldc, M, N, npad(N), K,
             M * (int64_t)N * 4 >= ((int64_t)128 << 20), nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr};
    return launch_nt<false>(g, (hipStream_t)stream);
}

int alignn_gemm_nt_f16x3(const float* A, int64_t lda, const float* a_amax, const void* Wsplit, const float* w_amax,
                         const float* bias, const float* addend, int64_t ldadd, float* C, int64_t ldc, int64_t M, int N,
                         int K, alignn_stream_t stream) {
    if (!alignn_gemm_nt_x6_supported(M, N, K) || a_amax == nullptr || w_amax == nullptr) return (int)hipErrorInvalidValue;
    if (!nt_args_ok(A, lda, Wsplit, bias, addend, ldadd, C, ldc)) return (int)hipErrorInvalidValue;
    X6Args g{A, lda, (const unsigned char*)Wsplit, a_amax, w_amax, bias, addend, ldadd, C, ldc, M, N, npad(N), K,
             M * (int64_t)N * 4 >= ((int64_t)128 << 20), nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr};
    return launch_nt<true>(g, (hipStream_t)stream);
}

int alignn_gemm_nt_f16x3_gather(const float* A, int64_t lda, const float* a_amax, const void* Wsplit, const float* w_amax,
                                const float* bias, float* C, int64_t ldc, int64_t M, int N, int K, const float* P, int64_t ldp,
                                const int32_t* src, const int32_t* dst, float* stats_partial, alignn_stream_t stream) {
    if (!alignn_gemm_nt_x6_supported(M, N, K) || a_amax == nullptr || w_amax == nullptr) return (int)hipErrorInvalidValue;
    if (!nt_args_ok(A, lda, Wsplit, bias, nullptr, 0, C, ldc)) return (int)hipErrorInvalidValue;
    if (P == nullptr || src == nullptr || dst == nullptr || (ldp & 3) || !a16(P) || (N & 3) || ldp < 2 * (int64_t)N ||
        (stats_partial && !a16(stats_partial)))
        return (int)hipErrorInvalidValue;
    X6Args g{A, lda, (const unsigned char*)Wsplit, a_amax, w_amax, bias, nullptr, 0, C, ldc, M, N, npad(N), K,
             M * (int64_t)N * 4 >= ((int64_t)128 << 20), nullptr, 0, nullptr, stats_partial, P, ldp, src, dst};
    return launch_nt<true>(g, (hipStream_t)stream);
}

/* alignn_gemm_nt_f16x3_gather with the second gathered row taken from its own table: C[e] = A[e] W^T + bias +
 * P[src[e]][0:N] + Bd2[rank[e]][0:N].  For a line graph the host passes Bd2 = the Bd rows in SEGMENT order and rank[e] = the
 * segment of row e: rows of one segment share one table row and consecutive segments read consecutive rows, where
 * P[dst[e]][N:2N] jumps through the table once per segment (measured: 65 of the kernel's 450 us at T rows). */
int alignn_gemm_nt_f16x3_gather2(const float* A, int64_t lda, const float* a_amax, const void* Wsplit, const float* w_amax,
                                 const float* bias, float* C, int64_t ldc, int64_t M, int N, int K, const float* P, int64_t ldp,
                                 const int32_t* src, const float* Bd2, int64_t ldbd2, const int32_t* rank,
                                 float* stats_partial, alignn_stream_t stream) {
    if (!alignn_gemm_nt_x6_supported(M, N, K) || a_amax == nullptr || w_amax == nullptr) return (int)hipErrorInvalidValue;
    if (!nt_args_ok(A, lda, Wsplit, bias, nullptr, 0, C, ldc)) return (int)hipErrorInvalidValue;
    if (P == nullptr || src == nullptr || Bd2 == nullptr || rank == nullptr || (ldp & 3) || (ldbd2 & 3) || !a16(P) || !a16(Bd2) ||
        (N & 3) || (stats_partial && !a16(stats_partial)))
        return (int)hipErrorInvalidValue;
    X6Args g{A, lda, (const unsigned char*)Wsplit, a_amax, w_amax, bias, nullptr, 0, C, ldc, M, N, npad(N), K,
             M * (int64_t)N * 4 >= ((int64_t)128 << 20), nullptr, 0, nullptr, stats_partial, P, ldp, src, rank};
    g.gp2 = Bd2;
    g.ldgp2 = ldbd2;
    return launch_nt<true>(g, (hipStream_t)stream);
}

int alignn_gemm_nt_f16x3_stats(const float* A, int64_t lda, const float* a_amax, const void* Wsplit, const float* w_amax,
                               const float* bias, float* C, int64_t ldc, int64_t M, int N, int K, float* stats_partial,
                               alignn_stream_t stream) {
    if (!alignn_gemm_nt_x6_supported(M, N, K) || a_amax == nullptr || w_amax == nullptr) return (int)hipErrorInvalidValue;
    if (!nt_args_ok(A, lda, Wsplit, bias, nullptr, 0, C, ldc)) return (int)hipErrorInvalidValue;
    if (stats_partial == nullptr || !a16(stats_partial) || (N & 3)) return (int)hipErrorInvalidValue;
    X6Args g{A, lda, (const unsigned char*)Wsplit, a_amax, w_amax, bias, nullptr, 0, C, ldc, M, N, npad(N), K,
             M * (int64_t)N * 4 >= ((int64_t)128 << 20), nullptr, 0, nullptr, stats_partial, nullptr, 0, nullptr, nullptr};
    return launch_nt<true>(g, (hipStream_t)stream);
}

int alignn_gemm_nt_x6_row_tiles(int64_t M, int N, int K) { return alignn_ceil_div(M, nt_block_rows(M, N, K)); }

int alignn_gemm_nt_f16x3_bnred(const float* A, int64_t lda, const float* a_amax, const void* Wsplit, const float* w_amax,
                               const float* bias, const float* addend, int64_t ldadd, float* C, int64_t ldc, int64_t M,
                               int N, int K, const float* Xn, int64_t ldxn, const float* nstat, float* red_partial,
                               alignn_stream_t stream) {
    if (!alignn_gemm_nt_x6_supported(M, N, K) || a_amax == nullptr || w_amax == nullptr) return (int)hipErrorInvalidValue;
    if (!nt_args_ok(A, lda, Wsplit, bias, addend, ldadd, C, ldc)) return (int)hipErrorInvalidValue;
    if (Xn == nullptr || nstat == nullptr || red_partial == nullptr || (ldxn & 3) || !a16(Xn) || !a16(nstat) ||
        !a16(red_partial))
        return (int)hipErrorInvalidValue;
    X6Args g{A, lda, (const unsigned char*)Wsplit, a_amax, w_amax, bias, addend, ldadd, C, ldc, M, N, npad(N), K,
             M * (int64_t)N * 4 >= ((int64_t)128 << 20), Xn, ldxn, nstat, red_partial, nullptr, 0, nullptr, nullptr};
    return launch_nt<true>(g, (hipStream_t)stream);
}

}  // extern "C"
