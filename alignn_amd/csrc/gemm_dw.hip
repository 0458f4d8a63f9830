// Input gradient AND weight gradient of a [M,256] x [256,256] projection in ONE pass over the projection's output gradient:
//     C[M,256]    = G[M,256] W[256,256] (+ addend)          (the backward of  m = y W_eg^T + ...,  alignn/models/alignn.py:101)
//     dW[256,256] = G[M,256]^T Y[M,256]
// (+ optionally the BatchNorm-backward column sums of C against the pre-activation `xn`, as alignn_gemm_nt_f16x3_bnred).
// Until round 5 these were two launches that each streamed G from HBM (gemm_nt_f16p_* / gemm_nt_x6_bnred on lane T and
// tn::gemm_tn_x6_kernel on the side stream: 4 x 692 MB per headline step, 14 x 575 MB per force-training step).  Here a
// workgroup owns a 64-row tile of G at a time and uses it for both products while it is in LDS.
//
// Arithmetic: the f16x3 split product of gemm_x6.hip (two fp16 slices per fp32 operand after a power-of-two scale, products
// hh + hl + lh on v_mfma_f32_32x32x16_f16, fp32 accumulation) - G is sliced ONCE and both products read the same slices.
// The input gradient repeats gemm_nt_f16p_body's operation order per output element (same bits); the weight gradient sums
// its rows in a different order than tn::gemm_tn_x6_kernel (tiles instead of slabs; fixed, so still run-to-run reproducible).
//
// Layout.  256 persistent workgroups of 8 waves (one per CU, two waves per SIMD, <= 256 registers each):
//   * dW accumulators: the workgroup's 256 x 256 partial lives in registers for the whole walk, 64 x 128 per wave (128 registers);
//     written once at the end as slab `blockIdx.x`, slabs summed in fp64 in slab order (dw_slab_reduce_kernel).
//   * G tile: 64 rows x 1 KiB arrive by global_load_lds DMA as they lie in memory (one instruction per row), are sliced IN PLACE
//     by the wave that requested them into [row][hi 512 B | lo 512 B] fp16 planes (16-byte slots XOR-swizzled by the row so that
//     both access patterns below are conflict-free) and stay there for the tile:
//       - input gradient: A operand = rows of G, 8 consecutive features per lane: ds_read_b128;
//       - weight gradient: A operand = G^T (lane = feature, registers = 8 consecutive rows): gfx950's transposing LDS read
//         ds_read_b64_tr_b16, two per operand (tools/tr_probe.hip checks the addressing on the device).
//   * everything else streams through ONE ring of four 16 KiB slots, three stages in flight, counted vmcnt waits:
//       Y0..Y3    16 rows of Y each (sliced in place like G; read with the transposing read as the B operand of the weight gradient)
//       W0..W15   the k-blocks of the pre-sliced W^T image (alignn_split_f16x2, transpose = 1), the input gradient's B operand
//       [A_q X_q] q = 0..3: the residual addend and the BatchNorm pre-activation for the epilogue's rows, 16 rows each
//     so the epilogue's operands are in LDS before it starts (in the two-launch form they were global loads with a look-ahead
//     of one or two half rounds: 59 k of the 100 k cycles of a tile, profiles/r02_x6_phase_trace_persistent.txt).
//   * the next tile's G rows are requested when the last k-step has left the tile's planes, and land under the epilogue.
// The order of a wave's memory operations per tile is fixed (phantom tiles past the end re-request the last tile), so every
// wait is an exact count (Sched below computes them at compile time by simulating the operation sequence).
#include <type_traits>
#include <utility>

#include "common.h"
#include "../../include/alignn_hip.h"

// Ablation / tracing switches of tools/dw_ablate.py (never defined in the shipped build).  DW_ABL bits: 1 no W-stage DMA,
// 2 no epilogue-stage DMA, 4 no input-gradient products, 8 no weight-gradient products, 16 no stores of C, 32 no Y-stage DMA,
// 64 no G-row DMA (timing only: results are garbage).  DW_TRACE: wave 0 of every workgroup stamps the phases of its third tile.
#ifndef DW_ABL
#define DW_ABL 0
#endif
#ifndef DW_TRACE
#define DW_TRACE 0
#endif
#ifndef DW_NS
#define DW_NS 4
#endif
#ifndef DW_IL
#define DW_IL 0  // 1 (lock-step form): the DMA requests of a step are issued BETWEEN its products, one per three MFMAs (measured
                 // slower: 714 vs 660 us, profiles/r06_dw_ablate.txt)
#endif
#ifndef DW_PP
#define DW_PP 0  // 1: two teams of four waves alternate matrix and load segments (see the kernel; measured slower: 721 vs 660 us);
                 // 0: all eight waves in lock-step
#endif
#if DW_TRACE
__device__ unsigned long long dw_trace_buf[256 * 16];
#endif

#ifndef DW_GRID_DEFAULT
#define DW_GRID_DEFAULT 224  // (profiles/r06_dw_grid_ab.txt - headline step, eager / replayed ms, three boxes: 256: 14.56-14.97 / 14.56-14.67; 224: 14.37-14.58 / 14.54-14.68; 192: 14.41-14.53 / 14.81-14.93; 160: 14.60)
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef short s16x8 __attribute__((__vector_size__(8 * sizeof(short))));

constexpr int H = 256;                   // features: N = K = 256
constexpr int R = 64;                    // rows per tile
constexpr int NW = 8, NTH = NW * 64;     // waves, threads
constexpr int NS = DW_NS;                // ring slots
constexpr int SLOT = 16384;
constexpr int GBUF = 0;                                  // 64 rows x 1 KiB
constexpr int RING = GBUF + R * 1024;                    // NS x 16 KiB
constexpr int PATCH = RING + NS * SLOT;                  // per wave an [8][64 + 4] fp32 transpose patch
constexpr int PLD = 68, PATCH_W = 8 * PLD * 4;
constexpr int NSTAT = PATCH + NW * PATCH_W;              // [4][256]: mean, rstd, gamma rstd, beta
constexpr int LDS_BYTES = NSTAT + 4 * H * 4;             // 152 576 B
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");

// ---- the order of one wave's vector-memory operations, simulated at compile time -------------------------------------
// Steps of a tile: 0-3 the Y stages (one ring slot each), 4-11 PAIRS of W stages (two slots each: two k-steps per barrier),
// then EST epilogue steps of ONE stage each (NE == 1: four quarters of the one operand; NE == 2: eight eighths of both operands,
// so that a step never needs two stages - the ring would then prefetch only one step ahead; NE == 0: four quarters, no stage).
// After the barrier of step s the wave requests the stages that take the ring positions step s-1 released (two DMA
// instructions per stage and wave), at the first epilogue step also the next tile's G rows (8 instructions), and every epilogue
// step issues its stores (2 or 1 per lane).  wait[s] = the number of operations issued after the last DMA of the last stage
// step s consumes = the immediate of its s_waitcnt vmcnt.
template <int NE>
struct Sched {
    static constexpr int EST = NE == 2 ? 8 : 4, E0 = 12, STEPS = E0 + EST, NSL = 20 + (NE > 0 ? EST : 0), STORES = 8 / EST;
    static_assert(NSL % NS == 0, "ring positions are compile-time constants");
    int w0[STEPS], ws[STEPS];  // first tile / every later tile
    int g0, gs;                // ... for the tile's G rows
    int prologue;              // stages requested before the first tile
    static constexpr int cons(int s) { return s < 4 ? 1 : s < E0 ? 2 : (NE > 0 ? 1 : 0); }
    static constexpr int first(int s) { return s <= 4 ? s : s <= E0 ? 4 + 2 * (s - 4) : 20 + (s - E0) * cons(E0); }
    static constexpr int fill_lo(int s) { return (s == 0 ? first(STEPS - 1) - NSL : first(s - 1)) + NS; }
    static constexpr int fill_n(int s) { return s == 0 ? cons(STEPS - 1) : cons(s - 1); }
    static constexpr int slot_ops(int n) {  // DMA instructions per wave for stage n (2; 0 under the ablation switches)
        n = ((n % NSL) + NSL) % NSL;
        if ((DW_ABL & 32) && n < 4) return 0;
        if ((DW_ABL & 1) && n >= 4 && n < 20) return 0;
        if ((DW_ABL & 2) && n >= 20) return 0;
        return 2;
    }
    static constexpr int g_ops() { return (DW_ABL & 64) ? 0 : 8; }
    static constexpr int store_ops() { return (DW_ABL & 16) ? 0 : STORES; }
    static constexpr Sched make() {
        Sched r{};
        constexpr int TILES = 3;
        int end_op[(TILES + 1) * NSL + NS] = {};
        int g_end[TILES + 2] = {};
        int ops = 0;
        g_end[0] = (ops += g_ops());
        r.prologue = fill_lo(0);
        for (int n = 0; n < r.prologue; ++n) end_op[n] = (ops += slot_ops(n));
        int w[TILES][STEPS] = {};
        int gw[TILES] = {};
        for (int t = 0; t < TILES; ++t) {
            gw[t] = ops - g_end[t];
            for (int s = 0; s < STEPS; ++s) {
                if (cons(s) > 0) w[t][s] = ops - end_op[t * NSL + first(s) + cons(s) - 1];
                for (int k = 0; k < fill_n(s); ++k) end_op[t * NSL + fill_lo(s) + k] = (ops += slot_ops(fill_lo(s) + k));
                if (s == E0) g_end[t + 1] = (ops += g_ops());
                if (s >= E0) ops += store_ops();
            }
        }
        for (int s = 0; s < STEPS; ++s) r.w0[s] = w[0][s], r.ws[s] = w[2][s];
        r.g0 = gw[0], r.gs = gw[2];
        // (steady from the second tile on)
        for (int s = 0; s < STEPS; ++s)
            if (w[1][s] != w[2][s]) r.gs = -1;
        if (gw[1] != gw[2]) r.gs = -1;
        return r;
    }
};

template <int NE>
inline constexpr Sched<NE> kSched = Sched<NE>::make();

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void block_barrier() {  // s_barrier without the vmcnt(0) drain of __syncthreads()
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// 64 lanes x 16 B -> 1 KiB of LDS at lds_wave_base (wave-uniform) + lane * 16, from sbase (wave-uniform) + lane_off
__device__ __forceinline__ void dma16(const void* sbase, unsigned lane_off, unsigned char* lds_wave_base) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :
                 : "s"((unsigned)(size_t)lds_wave_base), "v"(lane_off), "s"(sbase)
                 : "memory");
}
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// (as gemm_x6.hip) power-of-two scale that puts a tensor with the given max|x| just below 2^15
__device__ __forceinline__ float f16_scale(float amax) {
    const int e = (int)((__float_as_uint(amax) >> 23) & 255u);
    if (e == 0 || e == 255) return 1.0f;
    int se = 268 - e;
    se = se > 254 ? 254 : se;
    return __uint_as_float((unsigned)se << 23);
}
// (as gemm_x6.hip's slice8_f16) 8 floats -> high and low fp16 slices of x s: h = RN(x s), l = RN(x s - h)
__device__ __forceinline__ void slice8(const float4& a, const float4& b, float s, uint4& hp, uint4& lp) {
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    float xs[8];
    f16x8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        xs[j] = x[j] * s;
        h[j] = (_Float16)xs[j];
    }
    hp = __builtin_bit_cast(uint4, h);
    const unsigned hw[4] = {hp.x, hp.y, hp.z, hp.w};
    unsigned lw[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(lw[p]) : "v"(xs[2 * p]), "v"(hw[p]));
        asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lw[p]) : "v"(xs[2 * p + 1]), "v"(hw[p]));
    }
    lp = make_uint4(lw[0], lw[1], lw[2], lw[3]);
}

// 16-byte slot swizzle of a plane row (both halves): conflict-free for ds_read_b128 over 16 consecutive rows at one slot and
// for the transposing read's 4 rows x 32 B (tools/tr_probe.hip, mode 2)
__host__ __device__ constexpr int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// one MFMA operand (lane = column index, registers = 8 consecutive rows) out of row-major fp16 planes: two transposing reads
template <int OFF>
__device__ __forceinline__ f16x8 tr_operand(const unsigned char* p0, const unsigned char* p1) {
    typedef __attribute__((address_space(3))) s16x4* lp;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p0 + OFF));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p1 + OFF));
    return __builtin_bit_cast(f16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

struct DwArgs {
    const float* G;
    int64_t ldg;
    const float* Y;
    int64_t ldy;
    const unsigned char* Wt;  // alignn_split_f16x2 image of W^T ([256 out, 256 red]): 16 k-blocks of 16 KiB
    const float *g_amax, *y_amax, *w_amax;
    const float* addend;
    int64_t ldadd;
    float* C;
    int64_t ldc;
    const float* xn;
    int64_t ldxn;
    const float* nstat;   // [4][256]
    float* red_partial;   // [2 gridDim.x][2][256]
    float* dw_ws;         // [gridDim.x][256][256]
    int64_t M;
    int tiles;
};

template <bool HAS_ADD, bool BNRED>
__global__ __launch_bounds__(NTH) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_dw_kernel(DwArgs g) {
    constexpr int NE = (HAS_ADD ? 1 : 0) + (BNRED ? 1 : 0);
    using S = Sched<NE>;
    static_assert(kSched<NE>.gs >= 0, "the operation sequence is periodic from the second tile on");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int il = lane & 31, half = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;  // input gradient: rows 32 wm .., columns 64 wn ..
    const int wf = wave >> 1, wc = wave & 1;  // weight gradient: dW rows (features of G) 64 wf .., columns (features of Y) 128 wc ..
    const int grid = gridDim.x;
    const int J = (g.tiles - (int)blockIdx.x + grid - 1) / grid;  // tiles of this workgroup (>= 1: the launcher sees to it)

    const float sg = f16_scale(*g.g_amax), sy = f16_scale(*g.y_amax);
    const float inv_sg = 1.0f / sg, inv_sw = 1.0f / f16_scale(*g.w_amax);
    const float inv_dw = (1.0f / sg) * (1.0f / sy);

    if constexpr (BNRED) {  // BatchNorm constants -> LDS (read per epilogue round; visible after the first barrier)
        for (int i = t; i < 4 * H / 4; i += NTH) *reinterpret_cast<float4*>(smem + NSTAT + i * 16) = f4_ld(g.nstat + i * 4);
    }

    // ---- DMA side ----------------------------------------------------------------------------------------------------
    // A tile is addressed as (scalar base of its first row, 32-bit row offsets): rows past the end of the matrix re-read the
    // last valid row.  `opaque` keeps hipcc from hoisting per-lane / per-tile address arithmetic out of the tile loop (it would
    // carry dozens of ready-made 64-bit addresses through the k-loops in registers the accumulators need).
    const unsigned lane16 = lane * 16;
    struct TileRef {
        int64_t m0;
        int last;  // last valid row of the tile, relative
    };
    auto tile_ref = [&](int tl) {  // (phantom tiles past the end: the last tile again)
        tl = tl < g.tiles ? tl : g.tiles - 1;
        const int64_t m0 = (int64_t)tl * R;
        return TileRef{m0, (int)(g.M - 1 - m0 < R - 1 ? g.M - 1 - m0 : R - 1)};
    };
    auto row_dma = [&](const float* base, int64_t ld, const TileRef& tr, int row, unsigned char* dst) {
        row = row < tr.last ? row : tr.last;
        dma16(base + tr.m0 * ld, lane16 + (unsigned)(row * (int)ld) * 4u, dst);
    };
    // ... one row per 16 lanes: lane-dependent row (clamped like row_dma's), byte offset col_b inside the row
    auto lane_dma = [&](const float* base, int64_t ld, const TileRef& tr, int row, unsigned col_b, unsigned char* dst) {
        row = row < tr.last ? row : tr.last;
        dma16(base + tr.m0 * ld, col_b + (unsigned)(row * (int)ld) * 4u, dst);
    };
    auto issue_G = [&](const TileRef& tr) {
        if constexpr (S::g_ops() == 0) return;
#pragma unroll
        for (int i = 0; i < 8; ++i) row_dma(g.G, g.ldg, tr, wave * 8 + i, smem + GBUF + (wave * 8 + i) * 1024);
    };
    // stage n (0 .. NSL-1) of tile tr -> ring position n % NS
    auto issue_slot = [&](auto nc, const TileRef& tr, int only = -1) {  // only: one of the stage's two instructions (-1: both)
        constexpr int n = decltype(nc)::value;
        if constexpr (S::slot_ops(n) == 0) return;
        unsigned char* dst = smem + RING + (n % NS) * SLOT;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (only >= 0 && only != i) continue;
            const int lr = 2 * wave + i;  // row of the stage / KiB piece of the weight block
            if constexpr (n < 4) {
                row_dma(g.Y, g.ldy, tr, 16 * n + lr, dst + lr * 1024);
            } else if constexpr (n < 20) {
                dma16(g.Wt + (n - 4) * SLOT, lane16 + (unsigned)lr * 1024u, dst + lr * 1024);
            } else {
                // epilogue stages: every wave requests exactly the rows it will read itself, into ITS 2 KiB of the slot - no
                // barrier in the epilogue (the waves drift apart there: stores, transcendentals), and a refill never touches
                // bytes another wave still reads.  One instruction = 4 rows x 256 B (lane: row l >> 4, 16-byte chunk l & 15).
                constexpr int e = n - 20;
                const int r4 = lane >> 4;
                const unsigned col_b = (unsigned)(64 * (wave & 3)) * 4u + (unsigned)(lane & 15) * 16u;
                if constexpr (NE == 1) {  // quarter e: rows 32 wm + 8 e + 4 i + r4 of the one operand
                    const int trow = 32 * (wave >> 2) + 8 * e + 4 * i + r4;
                    if constexpr (HAS_ADD)
                        lane_dma(g.addend, g.ldadd, tr, trow, col_b, dst + wave * 2048 + i * 1024);
                    else
                        lane_dma(g.xn, g.ldxn, tr, trow, col_b, dst + wave * 2048 + i * 1024);
                } else {  // eighth e: rows 32 wm + 4 e + r4; i = 0 the addend, i = 1 the pre-activation
                    const int trow = 32 * (wave >> 2) + 4 * e + r4;
                    if (i == 0)
                        lane_dma(g.addend, g.ldadd, tr, trow, col_b, dst + wave * 2048);
                    else
                        lane_dma(g.xn, g.ldxn, tr, trow, col_b, dst + wave * 2048 + 1024);
                }
            }
        }
    };

    auto opaque_lane = [&]() {
        int l = lane;
        asm volatile("" : "+v"(l));
        return l;
    };

    f32x16 acc_dw[2][4], acc_c[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_dw[a][b][r] = 0.0f;
    float4 s0 = f4_zero(), s1 = f4_zero();

    // ---- prologue: the first tile's G rows and the first stages
    int tile = blockIdx.x;
    {
        const TileRef tr0 = tile_ref(tile);
        issue_G(tr0);
        static_for<0, kSched<NE>.prologue>([&](auto nc) { issue_slot(nc, tr0); });
    }

#if DW_TRACE
    unsigned long long tr_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_w[3] = {0, 0, 0}, tr_b[3] = {0, 0, 0};
    int tr_j = 0;
#define DW_STAMP(i) do { if (tr_j == 2) tr_t[i] = __builtin_readcyclecounter(); } while (0)
#define DW_WAIT(ph, stmt) do { const unsigned long long t0_ = __builtin_readcyclecounter(); stmt; if (tr_j == 2) tr_w[ph] += __builtin_readcyclecounter() - t0_; } while (0)
#define DW_BARRIER(ph) do { const unsigned long long t0_ = __builtin_readcyclecounter(); block_barrier(); if (tr_j == 2) tr_b[ph] += __builtin_readcyclecounter() - t0_; } while (0)
#else
#define DW_STAMP(i) do { } while (0)
#define DW_WAIT(ph, stmt) stmt
#define DW_BARRIER(ph) block_barrier()
#endif
    f16x8 yah[2], yal[2], ybh[4], ybl[4];  // operands of a weight-gradient step (16 rows): G^T blocks, Y blocks
    f16x8 wah[2], wal[2], wbh[2][2], wbl[2][2];  // operands of a pair of input-gradient k-steps
    int ta_off[2], tb_off[2], a_base, b_off[2];
    auto y_reads = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        int ta0 = ta_off[0], ta1 = ta_off[1], tb0 = tb_off[0], tb1 = tb_off[1];
        asm volatile("" : "+v"(ta0), "+v"(ta1), "+v"(tb0), "+v"(tb1));  // (derive the 12 operand addresses here, not per tile)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const unsigned char* p0 = smem + (ta0 ^ (a << 6));
            const unsigned char* p1 = smem + (ta1 ^ (a << 6));
            yah[a] = tr_operand<s * SLOT>(p0, p1);
            yal[a] = tr_operand<s * SLOT + 512>(p0, p1);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const unsigned char* p0 = smem + (tb0 ^ (b << 6));
            const unsigned char* p1 = smem + (tb1 ^ (b << 6));
            ybh[b] = tr_operand<(s % NS) * SLOT>(p0, p1);
            ybl[b] = tr_operand<(s % NS) * SLOT + 512>(p0, p1);
        }
    };
    auto no_il = [](auto) {};
    // (`between(k)`, k = 0, 1: called after the first and second of the three passes - the step's DMA requests go there, DW_IL)
    auto y_mfma = [&](auto&& between) {
        if constexpr (!(DW_ABL & 8)) {
#define DW_PASS(AA, BB)                                                                       \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 4; ++b) \
        acc_dw[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AA[a], BB[b], acc_dw[a][b], 0, 0, 0);
            DW_PASS(yal, ybh)
            between(std::integral_constant<int, 0>{});
            DW_PASS(yah, ybl)
            between(std::integral_constant<int, 1>{});
            DW_PASS(yah, ybh)
#undef DW_PASS
        } else {
            acc_dw[0][0][0] += (float)yah[0][0] + (float)ybh[0][0] + (float)yal[1][0] + (float)ybl[3][0];  // (keeps the operand reads)
            between(std::integral_constant<int, 0>{});
            between(std::integral_constant<int, 1>{});
        }
    };
    auto w_reads = [&](auto ktc) {  // both k-steps of the pair that starts at kt
        constexpr int kt0 = decltype(ktc)::value;
        int ab = a_base;
        asm volatile("" : "+v"(ab));  // (one XOR per k-step instead of 16 addresses carried through the kernel)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned char* slot = smem + RING + ((kt0 + j + 4) % NS) * SLOT;
            const unsigned char* ap = smem + (ab ^ ((kt0 + j) << 5));
            wah[j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(ap));
            wal[j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(ap + 512));
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                wbh[j][b] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(slot + b_off[b]));
                wbl[j][b] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(slot + b_off[b] + 8192));
            }
        }
    };
    // (`between(k)`, k = 0 .. 3: called after every third product of the twelve)
    auto w_mfma = [&](auto&& between) {
        static_for<0, 2>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (!(DW_ABL & 4)) {
                acc_c[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wah[j], wbh[j][0], acc_c[0], 0, 0, 0);
                acc_c[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wah[j], wbh[j][1], acc_c[1], 0, 0, 0);
                acc_c[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wah[j], wbl[j][0], acc_c[0], 0, 0, 0);
                between(std::integral_constant<int, 2 * j>{});
                acc_c[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wah[j], wbl[j][1], acc_c[1], 0, 0, 0);
                acc_c[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wal[j], wbh[j][0], acc_c[0], 0, 0, 0);
                acc_c[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wal[j], wbh[j][1], acc_c[1], 0, 0, 0);
                between(std::integral_constant<int, 2 * j + 1>{});
            } else {
                acc_c[0][0] += (float)wah[j][0] + (float)wal[j][0] + (float)wbh[j][0][0] + (float)wbl[j][1][0];  // (keeps the reads)
                between(std::integral_constant<int, 2 * j>{});
                between(std::integral_constant<int, 2 * j + 1>{});
            }
        });
    };
    // Two teams (DW_PP): waves 0-3 lead, waves 4-7 - their partners on the four SIMDs (tools/simd_map.hip: waves w and w + 4
    // share one) - follow half an interval behind.  An interval has two barriers: in its first half the leaders fetch the
    // operands of step s while the others multiply what they fetched in step s - 1; in its second half the leaders multiply and the
    // others fetch.  On every SIMD one wave's matrix segment runs beside its partner's LDS segment (with all eight waves in
    // lock-step a k-step took 720 cycles for 384 of matrix work: profiles/r06_dw_ablate_v3.txt).
    // (the two teams run two COPIES of the tile loop, chosen once: with per-step team branches around the products hipcc copies
    // the accumulators at every join - 2 500 registers spilled)

    // one tile; FIRST: the waits of the first tile count the prologue's operations instead of the previous tile's
    auto tile_body = [&](auto first_c, auto lead_c) {
        constexpr bool FIRST = decltype(first_c)::value, lead = decltype(lead_c)::value;
        DW_STAMP(0);
        const TileRef cur = tile_ref(tile), nxt = tile_ref(tile + grid);
        const int64_t m0 = cur.m0;
        // ---- the tile's G rows: the wave slices the 8 rows it requested itself, in place
        DW_WAIT(0, (wait_vmcnt<FIRST ? kSched<NE>.g0 : kSched<NE>.gs>()));
        DW_STAMP(1);
        {
            const int l = opaque_lane();  // (the eight swizzled write addresses are derived here, not carried through the kernel)
            const int wslot = l >> 1, wsub = (l & 1) * 8;
#pragma unroll
            for (int i0 = 0; i0 < 8; i0 += 4) {  // (four rows at a time: 16 + 16 registers)
                float4 v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float4*>(smem + GBUF + (wave * 8 + i0 + i) * 1024 + l * 16);
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    uint4 hp, lp;
                    slice8(v[i], v[i + 1], sg, hp, lp);
                    const int ra = wave * 8 + i0 + i, rb = ra + 1;
                    unsigned char* pa = smem + GBUF + ra * 1024 + ((wslot ^ swz(ra)) << 4) + wsub;
                    unsigned char* pb = smem + GBUF + rb * 1024 + ((wslot ^ swz(rb)) << 4) + wsub;
                    *reinterpret_cast<uint2*>(pa) = make_uint2(hp.x, hp.y);
                    *reinterpret_cast<uint2*>(pa + 512) = make_uint2(lp.x, lp.y);
                    *reinterpret_cast<uint2*>(pb) = make_uint2(hp.z, hp.w);
                    *reinterpret_cast<uint2*>(pb + 512) = make_uint2(lp.z, lp.w);
                }
            }
        }
        // ---- weight gradient: four stages of 16 rows of Y
        // transposing reads: lane (grp, i16): column 16 (grp & 1) + 4 (i16 & 3) .. + 3 of a 32-wide block, rows
        // 8 (grp >> 1) + 4 h + (i16 >> 2) of a 16-row stage
        {
            const int l = opaque_lane(), grp = l >> 4, i16 = l & 15;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = 8 * (grp >> 1) + 4 * h + (i16 >> 2);
                const int sl = 2 * (grp & 1) + ((i16 >> 1) & 1);
                ta_off[h] = GBUF + row * 1024 + (((8 * wf + sl) ^ swz(row)) << 4) + (i16 & 1) * 8;   // ^ (a << 6); + 16 KiB per stage
                tb_off[h] = RING + row * 1024 + (((16 * wc + sl) ^ swz(row)) << 4) + (i16 & 1) * 8;  // ^ (b << 6); + ring position
            }
        }
        // slice the two rows of Y stage s this wave requested, in place (rows past the end of the matrix: zeros)
        auto y_convert = [&](auto sc) {
            constexpr int s = decltype(sc)::value;
            DW_WAIT(0, (wait_vmcnt<FIRST ? kSched<NE>.w0[s] : kSched<NE>.ws[s]>()));
            unsigned char* slot = smem + RING + (s % NS) * SLOT;
            const int ra = 2 * wave, rb = ra + 1;
            const int l = opaque_lane();
            float4 va = *reinterpret_cast<const float4*>(slot + ra * 1024 + l * 16);
            float4 vb = *reinterpret_cast<const float4*>(slot + rb * 1024 + l * 16);
            const float ka = m0 + 16 * s + ra < g.M ? 1.0f : 0.0f, kb = m0 + 16 * s + rb < g.M ? 1.0f : 0.0f;
            va = f4_scale(va, ka);
            vb = f4_scale(vb, kb);
            uint4 hp, lp;
            slice8(va, vb, sy, hp, lp);
            unsigned char* pa = slot + ra * 1024 + ((((l >> 1) ^ swz(ra))) << 4) + (l & 1) * 8;
            unsigned char* pb = slot + rb * 1024 + ((((l >> 1) ^ swz(rb))) << 4) + (l & 1) * 8;
            *reinterpret_cast<uint2*>(pa) = make_uint2(hp.x, hp.y);
            *reinterpret_cast<uint2*>(pa + 512) = make_uint2(lp.x, lp.y);
            *reinterpret_cast<uint2*>(pb) = make_uint2(hp.z, hp.w);
            *reinterpret_cast<uint2*>(pb + 512) = make_uint2(lp.z, lp.w);
        };
        auto fills = [&](auto sc) {  // the stages that take the ring positions step s - 1 released
            constexpr int s = decltype(sc)::value;
            static_for<0, S::fill_n(s)>([&](auto kc) {
                constexpr int n = S::fill_lo(s) + decltype(kc)::value;
                issue_slot(std::integral_constant<int, n % S::NSL>{}, n >= S::NSL ? nxt : cur);
            });
        };
        // ... one DMA instruction of them at a time (piece k of 2 fill_n(s)), fenced so that it stays between the products
        auto fill_piece = [&](auto sc, auto kc) {
            constexpr int s = decltype(sc)::value, k = decltype(kc)::value;
            if constexpr (k < 2 * S::fill_n(s)) {
                constexpr int n = S::fill_lo(s) + k / 2;
                __builtin_amdgcn_sched_barrier(0);
                issue_slot(std::integral_constant<int, n % S::NSL>{}, n >= S::NSL ? nxt : cur, k % 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        DW_STAMP(2);
        y_convert(std::integral_constant<int, 0>{});
        static_for<0, 4>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            DW_BARRIER(0);  // the planes of this stage (and, s == 0, of G) are complete; step s-1's ring position is free
#if DW_PP
            // the load segment of a team: DMA requests, operand reads, slicing of the next stage - beside the other team's products
            if constexpr (lead) {
                fills(sc);
                y_reads(sc);
                if constexpr (s < 3) y_convert(std::integral_constant<int, s + 1>{});
            } else {
                if constexpr (s > 0) y_mfma(no_il);  // (step s - 1's operands)
            }
            block_barrier();
            if constexpr (lead) {
                y_mfma(no_il);
            } else {
                fills(sc);
                y_reads(sc);
                if constexpr (s < 3) y_convert(std::integral_constant<int, s + 1>{});
            }
#elif DW_IL
            y_reads(sc);
            y_mfma([&](auto kc) { fill_piece(sc, kc); });
            if constexpr (s < 3) y_convert(std::integral_constant<int, s + 1>{});
#else
            fills(sc);
            y_reads(sc);
            y_mfma(no_il);
            if constexpr (s < 3) y_convert(std::integral_constant<int, s + 1>{});
#endif
        });
        // ---- input gradient: 16 k-steps over the weight image
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_c[b][r] = 0.0f;
        // A operand: row 32 wm + il of the planes, k-step kt: slot 2 kt + half -> (base ^ (kt << 5)); B operand (weight image:
        // [plane][n 256][2 chunks, swizzled][8]): n = 64 wn + 32 b + il
        {
            const int l = opaque_lane(), il = l & 31, half = l >> 5;
            const int a_row = 32 * wm + il;
            a_base = GBUF + a_row * 1024 + ((half ^ swz(a_row)) << 4);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int n = 64 * wn + 32 * b + il;
                b_off[b] = n * 32 + ((half ^ ((n >> 3) & 1)) << 4);
            }
        }
        static_for<4, S::E0>([&](auto sc) {
            constexpr int s = decltype(sc)::value, kt0 = 2 * (s - 4);  // two k-steps per barrier
            if (s == 4) DW_STAMP(3);
            DW_WAIT(1, (wait_vmcnt<FIRST ? kSched<NE>.w0[s] : kSched<NE>.ws[s]>()));
            DW_BARRIER(1);
#if DW_PP
            if constexpr (lead) {
                fills(sc);
                w_reads(std::integral_constant<int, kt0>{});
            } else {
                if constexpr (kt0 == 0)
                    y_mfma(no_il);  // (the last weight-gradient step's operands)
                else
                    w_mfma(no_il);  // (the previous pair's operands)
            }
            block_barrier();
            if constexpr (lead) {
                w_mfma(no_il);
            } else {
                fills(sc);
                w_reads(std::integral_constant<int, kt0>{});
            }
#elif DW_IL
            w_reads(std::integral_constant<int, kt0>{});
            w_mfma([&](auto kc) { fill_piece(sc, kc); });
#else
            fills(sc);
            w_reads(std::integral_constant<int, kt0>{});
            w_mfma(no_il);
#endif
        });
        // ---- epilogue: the operands of each step in the ring
        // a step = 8 (4) rows x 64 columns per wave through its patch; lane (prow, pc4) takes rows prow (and prow + 4) at the
        // columns 64 wn + pc4 .. + 3 - the SAME four columns in every round, so the BatchNorm-backward sums are 8 registers
        const int last_row = cur.last;
        float* c_t = g.C + m0 * g.ldc;
        const int le = opaque_lane();
        const int prow_e = le >> 4, pc4 = (le & 15) * 4, ecol = 64 * wn + pc4, il = le & 31, half = le >> 5;
        float* patch = reinterpret_cast<float*>(smem + PATCH + wave * PATCH_W);
        static_for<S::E0, S::STEPS>([&](auto sc) {
            constexpr int s = decltype(sc)::value, e = s - S::E0;
            constexpr int QR = 32 / S::EST;                      // rows per wave and step: 8 (quarters) or 4 (eighths)
            constexpr int q = QR == 8 ? e : e / 2, hf = e & 1;   // accumulator registers 4 q .. 4 q + 3 (eighths: of half-wave hf)
            constexpr int n_slot = 20 + e;                       // this step's stage (NE > 0)
            if (e == 0) DW_STAMP(4);
            if constexpr (NE > 0) DW_WAIT(2, (wait_vmcnt<FIRST ? kSched<NE>.w0[s] : kSched<NE>.ws[s]>()));
            if constexpr (e == 0) DW_BARRIER(2);  // every wave has left the tile's planes and the last W stages
            if constexpr (e == 0 && DW_PP && !lead) w_mfma(no_il);  // (the last pair's operands)
            fills(sc);
            if constexpr (e == 0) issue_G(nxt);
            const unsigned char* mine = smem + RING + (n_slot % NS) * SLOT + wave * 2048;  // this wave's rows of the stage
            // (eighths: the other half-wave's registers go to the patch's unused rows 4 .. 7 - no branch around the writes)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    patch[(k + 4 * (QR == 8 ? half : (half ^ hf))) * PLD + 32 * nb + il] = acc_c[nb][4 * q + k];
#pragma unroll
            for (int i = 0; i < QR / 4; ++i) {
                const int pr = prow_e + 4 * i;
                const int rrel = 32 * wm + QR * e + pr;
                const int rowc = rrel < last_row ? rrel : last_row;
                float4 v = f4_ld(patch + pr * PLD + pc4);
                v = f4_scale(f4_scale(v, inv_sg), inv_sw);
                // the wave's part of the stage: quarters [2 x 4 rows][256 B] of the one operand, eighths [addend | xn][4 rows][256 B]
                const int e_off = (NE == 2 ? prow_e : pr) * 256 + pc4 * 4;
                if constexpr (HAS_ADD) v = f4_add(v, *reinterpret_cast<const float4*>(mine + e_off));
                if constexpr (BNRED) {
                    const float4 xv = *reinterpret_cast<const float4*>(mine + e_off + (NE == 2 ? 1024 : 0));
                    const float4 xc = f4_sub(xv, *reinterpret_cast<const float4*>(smem + NSTAT + ecol * 4));
                    const float4 z = f4_fma(xc, *reinterpret_cast<const float4*>(smem + NSTAT + 2 * H * 4 + ecol * 4),
                                            *reinterpret_cast<const float4*>(smem + NSTAT + 3 * H * 4 + ecol * 4));
                    float4 gz = make_float4(v.x * dsilu_f(z.x), v.y * dsilu_f(z.y), v.z * dsilu_f(z.z), v.w * dsilu_f(z.w));
                    gz = rrel <= last_row ? gz : f4_zero();
                    s0 = f4_add(s0, gz);
                    s1 = f4_fma(gz, xc, s1);
                }
                if (!(DW_ABL & 16) || v.x == 12345.678f) f4_sts<true>(c_t + (rowc * (int)g.ldc + ecol), v);
                if constexpr (BNRED) __builtin_amdgcn_sched_barrier(0);  // (one row's transcendental chain at a time: registers)
            }
        });
        DW_STAMP(5);
#if DW_TRACE
        ++tr_j;
#endif
    };
    auto run_tiles = [&](auto lead_c) {
        tile_body(std::true_type{}, lead_c);
        tile += grid;
        for (int j = 1; j < J; ++j, tile += grid) tile_body(std::false_type{}, lead_c);
    };
    if (!DW_PP || wave < 4)  // (wave-uniform: one scalar branch per kernel)
        run_tiles(std::true_type{});
    else
        run_tiles(std::false_type{});
#if DW_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 256) {
        unsigned long long* o = dw_trace_buf + blockIdx.x * 16;
        for (int i = 0; i < 6; ++i) o[i] = tr_t[i];
        for (int i = 0; i < 3; ++i) o[6 + i] = tr_w[i], o[9 + i] = tr_b[i];
    }
#endif

    wait_vmcnt<0>();  // (the phantom stages requested past the last tile: no DMA may outlive the workgroup's LDS allocation)
    // ---- the workgroup's dW partial -> slab blockIdx.x (accumulator layout: 32 consecutive columns per row and register)
    {
        float* out = g.dw_ws + (size_t)blockIdx.x * (H * H);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = 64 * wf + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * half, c = 128 * wc + 32 * b + il;
                    out[f * H + c] = acc_dw[a][b][r] * inv_dw;
                }
    }
    if constexpr (BNRED) {
        // column sums: the wave's four row groups (lane >> 4) hold the same four columns - fixed-order butterfly; one slab per
        // (workgroup, wave row)
        const int ecol = 64 * wn + (lane & 15) * 4;
        float4 a0 = s0, a1 = s1;
#pragma unroll
        for (int d = 16; d < 64; d <<= 1) {
            a0 = f4_add(a0, make_float4(__shfl_xor(a0.x, d), __shfl_xor(a0.y, d), __shfl_xor(a0.z, d), __shfl_xor(a0.w, d)));
            a1 = f4_add(a1, make_float4(__shfl_xor(a1.x, d), __shfl_xor(a1.y, d), __shfl_xor(a1.z, d), __shfl_xor(a1.w, d)));
        }
        if (lane < 16) {
            a1 = f4_mul(a1, f4_ld(g.nstat + H + ecol));  // sum gz (x - mean) -> sum gz xhat
            const size_t slab = (size_t)blockIdx.x * 2 + wm;
            f4_st(g.red_partial + (slab * 2 + 0) * H + ecol, a0);
            f4_st(g.red_partial + (slab * 2 + 1) * H + ecol, a1);
        }
    }
}

// dW[256][256] (leading dimension ldo) = sum over the slabs, fp64, slab order (the scheme of gemm_f32.hip's slab_reduce4_kernel)
__global__ __launch_bounds__(1024) void dw_slab_reduce_kernel(const float* __restrict__ ws, int splits, float* __restrict__ out,
                                                              int64_t ldo) {
    __shared__ double sh[16][64][4];
    const int64_t count = H * H;
    const int64_t idx = ((int64_t)blockIdx.x * 64 + threadIdx.x) * 4;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int k = threadIdx.y; k < splits; k += 16) {
        const float4 v = f4_ld(ws + (int64_t)k * count + idx);
        s[0] += (double)v.x, s[1] += (double)v.y, s[2] += (double)v.z, s[3] += (double)v.w;
    }
    double* mine = sh[threadIdx.y][threadIdx.x];
    mine[0] = s[0], mine[1] = s[1], mine[2] = s[2], mine[3] = s[3];
    __syncthreads();
    if (threadIdx.y == 0) {
        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll 4
        for (int k = 0; k < 16; ++k) {
            const double* p = sh[k][threadIdx.x];
            t0 += p[0], t1 += p[1], t2 += p[2], t3 += p[3];
        }
        f4_st(out + (idx / H) * ldo + (idx % H), make_float4((float)t0, (float)t1, (float)t2, (float)t3));
    }
}

// One workgroup per compute unit - and each takes ALL of its compute unit (8 waves x 256 registers, 149 KiB of LDS): whatever
// another stream launches while this kernel runs waits for a compute unit the launch does not use.  With all 256 taken the small
// kernels of the bond-graph chain beside it (column reductions, slab sums: 10-20 us alone) were measured at 470-740 us in the
// step - they ran when this kernel ended.  ALIGNN_AMD_DW_GRID (a multiple of 8: the XCDs stay balanced) leaves the rest to them.
constexpr int kGridMax = 256;
inline int dw_grid_cap() {
    static const int cap = [] {
        const char* e = getenv("ALIGNN_AMD_DW_GRID");
        int v = e ? atoi(e) : DW_GRID_DEFAULT;
        v = v < 8 ? 8 : (v > kGridMax ? kGridMax : v);
        return v & ~7;
    }();
    return cap;
}
inline int dw_grid(int64_t M) {
    const int64_t tiles = (M + R - 1) / R;
    const int cap = dw_grid_cap();
    return (int)(tiles < cap ? tiles : cap);
}
inline bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

#if DW_TRACE
extern "C" int alignn_dw_trace_read(void* host, size_t bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dw_trace_buf), bytes < sizeof(dw_trace_buf) ? bytes : sizeof(dw_trace_buf));
}
#endif

extern "C" {

int alignn_gemm_dgrad_wgrad_supported(int64_t M, int N, int K) {
    return (N == H && K == H && M >= 4096 && (M + R - 1) / R < ((int64_t)1 << 30)) ? 1 : 0;
}
int alignn_gemm_dgrad_wgrad_slabs(int64_t M) { return dw_grid(M); }
size_t alignn_gemm_dgrad_wgrad_workspace(int64_t M) { return (size_t)dw_grid(M) * H * H * sizeof(float); }

int alignn_gemm_dgrad_wgrad_f16x3(const float* G, int64_t ldg, const float* g_amax, const float* Y, int64_t ldy, const float* y_amax,
                                  const void* Wt_split, const float* w_amax, const float* addend, int64_t ldadd, float* C,
                                  int64_t ldc, const float* Xn, int64_t ldxn, const float* nstat, float* red_partial, float* dW,
                                  int64_t lddw, int64_t M, void* workspace, size_t workspace_bytes, alignn_stream_t stream) {
    if (!alignn_gemm_dgrad_wgrad_supported(M, H, H) || G == nullptr || Y == nullptr || Wt_split == nullptr || C == nullptr ||
        dW == nullptr || g_amax == nullptr || y_amax == nullptr || w_amax == nullptr)
        return (int)hipErrorInvalidValue;
    if ((ldg & 3) || (ldy & 3) || (ldc & 3) || (lddw & 3) || !a16(G) || !a16(Y) || !a16(C) || !a16(dW) || !a16(Wt_split) ||
        (addend && ((ldadd & 3) || !a16(addend))))
        return (int)hipErrorInvalidValue;
    const bool bnred = Xn != nullptr;
    if (bnred && (nstat == nullptr || red_partial == nullptr || (ldxn & 3) || !a16(Xn) || !a16(nstat) || !a16(red_partial)))
        return (int)hipErrorInvalidValue;
    if (workspace == nullptr || !a16(workspace) || workspace_bytes < alignn_gemm_dgrad_wgrad_workspace(M))
        return (int)hipErrorInvalidValue;
    const int grid = dw_grid(M);
    DwArgs g{G, ldg, Y, ldy, (const unsigned char*)Wt_split, g_amax, y_amax, w_amax, addend, ldadd, C, ldc, Xn, ldxn, nstat,
             red_partial, (float*)workspace, M, (int)((M + R - 1) / R)};
    static bool attr_set = false;
    if (!attr_set) {
        const void* fns[] = {(const void*)gemm_dw_kernel<false, false>, (const void*)gemm_dw_kernel<true, false>,
                             (const void*)gemm_dw_kernel<false, true>, (const void*)gemm_dw_kernel<true, true>};
        for (const void* fn : fns) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
            if (e != hipSuccess) return (int)e;
        }
        attr_set = true;
    }
    hipStream_t st = (hipStream_t)stream;
    if (addend != nullptr && bnred)
        hipLaunchKernelGGL((gemm_dw_kernel<true, true>), dim3(grid), dim3(NTH), LDS_BYTES, st, g);
    else if (addend != nullptr)
        hipLaunchKernelGGL((gemm_dw_kernel<true, false>), dim3(grid), dim3(NTH), LDS_BYTES, st, g);
    else if (bnred)
        hipLaunchKernelGGL((gemm_dw_kernel<false, true>), dim3(grid), dim3(NTH), LDS_BYTES, st, g);
    else
        hipLaunchKernelGGL((gemm_dw_kernel<false, false>), dim3(grid), dim3(NTH), LDS_BYTES, st, g);
    ALIGNN_CHECK_LAUNCH();
    hipLaunchKernelGGL(dw_slab_reduce_kernel, dim3(H * H / 256), dim3(64, 16), 0, st, (const float*)workspace, grid, dW, lddw);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
