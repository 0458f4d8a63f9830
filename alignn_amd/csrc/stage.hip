// Batch staging in ONE call: canonical CSR of the bond graph g, its line graph L(g), the by-source views of both, the
// canonically ordered bond vectors and the bond-angle cosines, from the COO bond list a loader ships.
//
// Replaces, per batch: dgl.batch + g.to(device) + lg.to(device) of alignn/train.py:264-270 / alignn/lmdb_dataset.py:87-108
// (the reference moves the T-sized COO list of L(g) and its cosines over PCIe every step) and the ~60 torch index
// operations alignn_amd/graph.py (build_csr, line_graph_of) needed on the consumer's thread for the same result.
// Everything is index arithmetic on the canonical layout (include/alignn_hip.h, "Conventions"):
//   g     slots sorted by destination atom (stable: caller's order within a segment); by-source view = stable sort of
//         the slots by source atom;
//   L(g)  node i = g's slot i; one segment per bond e2 (in by-source order of g: all bonds leaving atom j are consecutive),
//         listing the in-edges e1 of j in ascending slot order minus e2 itself - a dense, source-sorted block per atom, so
//         rows, segment ranks and the by-source view follow from prefix sums without sorting anything T-sized.
// T (rows of L(g)) and the largest in-degree are functions of the bond list alone and are computed on the HOST when the
// batch is packed (alignn_amd/loader.pack), so no device value is ever read back.
// Two stable radix sorts (rocPRIM) of E keys + two prefix sums of E + 1 counts; ~15 launches, no atomics, no allocation.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "../../include/alignn_hip.h"
#include "common.h"

namespace {

constexpr int kT = 256;
inline unsigned blocks(int64_t n) { return (unsigned)((n + kT - 1) / kT); }
inline size_t al256(size_t b) { return (b + 255) / 256 * 256; }

__global__ void iota_kernel(int32_t* __restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (i < n) p[i] = (int32_t)i;
}

// first index k in [0, n) with keys[k] >= x (keys ascending)
__device__ __forceinline__ int32_t lower_bound(const int32_t* __restrict__ keys, int32_t n, int32_t x) {
    int32_t lo = 0, hi = n;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (keys[mid] < x) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
// last index s in [0, n) with ptr[s] <= t (ptr ascending, ptr[0] = 0 <= t < ptr[n])
__device__ __forceinline__ int32_t segment_of(const int32_t* __restrict__ ptr, int32_t n, int32_t t) {
    int32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        const int32_t mid = (lo + hi) >> 1;
        if (ptr[mid] <= t) lo = mid;
        else hi = mid;
    }
    return lo;
}

__global__ void ptr_from_sorted_kernel(const int32_t* __restrict__ keys, int32_t n_keys, int32_t n_nodes,
                                       int32_t* __restrict__ ptr) {
    const int32_t i = blockIdx.x * kT + threadIdx.x;
    if (i <= n_nodes) ptr[i] = lower_bound(keys, n_keys, i);
}

// canonical slots of g: slot k holds the caller's edge p = perm32[k]
__global__ void fill_g_kernel(const int32_t* __restrict__ u, const int32_t* __restrict__ perm32, const float* __restrict__ r,
                              int32_t E, int32_t* __restrict__ src, int64_t* __restrict__ perm, int64_t* __restrict__ inv,
                              float* __restrict__ r_canon) {
    const int32_t k = blockIdx.x * kT + threadIdx.x;
    if (k >= E) return;
    const int32_t p = perm32[k];
    src[k] = u[p];
    perm[k] = p;
    inv[p] = k;
    if (r != nullptr) {
        r_canon[3 * (int64_t)k + 0] = r[3 * (int64_t)p + 0];
        r_canon[3 * (int64_t)k + 1] = r[3 * (int64_t)p + 1];
        r_canon[3 * (int64_t)k + 2] = r[3 * (int64_t)p + 2];
    }
}

// rows per segment of L(g) (segment s = bond e2 = out_slot[s] leaving atom j: the in-edges of j minus e2 itself when it is
// a self-image bond), rows per source node of L(g) (bond e1 arriving at j: the bonds leaving j minus e1 itself), and the rank
// of every slot in the by-source order
__global__ void lg_counts_kernel(const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ src,
                                 const int32_t* __restrict__ dst, const int32_t* __restrict__ out_ptr,
                                 const int32_t* __restrict__ out_slot, int32_t E, int32_t* __restrict__ cnt_seg,
                                 int32_t* __restrict__ cnt_out, int32_t* __restrict__ out_rank) {
    const int32_t s = blockIdx.x * kT + threadIdx.x;
    if (s > E) return;
    if (s == E) {  // (closing zeros: the exclusive scans then end with the totals)
        cnt_seg[E] = 0;
        cnt_out[E] = 0;
        return;
    }
    const int32_t e2 = out_slot[s];
    const int32_t j = src[e2];
    cnt_seg[s] = seg_ptr[j + 1] - seg_ptr[j] - (dst[e2] == j ? 1 : 0);
    out_rank[e2] = s;
    const int32_t e1 = s;  // (the same thread index doubles as a slot id for the by-source counts)
    const int32_t a = dst[e1];
    cnt_out[e1] = out_ptr[a + 1] - out_ptr[a] - (src[e1] == a ? 1 : 0);
}

__global__ void lg_rows_kernel(const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ src,
                               const int32_t* __restrict__ dst, const int32_t* __restrict__ out_slot,
                               const int32_t* __restrict__ lg_seg_ptr, int32_t E, int64_t T, int32_t* __restrict__ lg_src,
                               int32_t* __restrict__ lg_dst, int32_t* __restrict__ seg_rank, int64_t* __restrict__ ident) {
    const int64_t t = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (t >= T) return;
    const int32_t s = segment_of(lg_seg_ptr, E, (int32_t)t);
    const int32_t pos = (int32_t)t - lg_seg_ptr[s];
    const int32_t e2 = out_slot[s];
    const int32_t j = src[e2];
    const int32_t base = seg_ptr[j];
    const bool has_self = dst[e2] == j;
    const int32_t e1 = base + pos + ((has_self && pos >= e2 - base) ? 1 : 0);
    lg_src[t] = e1;
    lg_dst[t] = e2;
    seg_rank[t] = s;
    if (ident != nullptr) ident[t] = t;
}

// by-source view of L(g): the q-th row in (source e1, row) order - what a stable sort of the rows by source would give
__global__ void lg_out_slot_kernel(const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ src,
                                   const int32_t* __restrict__ dst, const int32_t* __restrict__ out_ptr,
                                   const int32_t* __restrict__ out_slot, const int32_t* __restrict__ out_rank,
                                   const int32_t* __restrict__ lg_seg_ptr, const int32_t* __restrict__ lg_out_ptr, int32_t E,
                                   int64_t T, int32_t* __restrict__ lg_out_slot) {
    const int64_t q = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (q >= T) return;
    const int32_t e1 = segment_of(lg_out_ptr, E, (int32_t)q);
    const int32_t pos = (int32_t)q - lg_out_ptr[e1];
    const int32_t j = dst[e1];  // centre atom
    const int32_t first = out_ptr[j];
    const bool is_out = src[e1] == j;  // e1 is itself a bond leaving j (self image): its own segment is skipped
    const int32_t s = first + pos + ((is_out && pos >= out_rank[e1] - first) ? 1 : 0);
    const int32_t e2 = out_slot[s];
    const int32_t base = seg_ptr[j];
    const bool seg_has_self = dst[e2] == j;  // segment s lists the in-edges of j minus e2
    lg_out_slot[q] = lg_seg_ptr[s] + (e1 - base) - ((seg_has_self && e2 < e1) ? 1 : 0);
}

// A caller's OWN edge list of L(g) (DGL: g.line_graph(shared=True); lg_u[k] -> lg_v[k] in the caller's ids of g's edges) ->
// for every caller edge k the canonical row t(k) it is, by index arithmetic on the canonical layout (no T-sized sort):
// e1 = inv[lg_u[k]], e2 = inv[lg_v[k]] are g's slots; the row lies in the segment of e2 (rank out_rank[e2]) at the position
// of e1 among the in-edges of the centre atom j = src[e2], minus one if e2 itself precedes it there.  perm[t] = k (the
// caller pre-fills perm with -1), inv_lg[k] = t; *bad counts edges that are not edges of the line graph at all
// (dst[e1] != src[e2], or e1 == e2) or that repeat an earlier one: n_triplets edges with bad == 0 are a bijection.
__global__ void map_lg_rows_kernel(const int64_t* __restrict__ lg_u, const int64_t* __restrict__ lg_v,
                                   const int64_t* __restrict__ inv, const int32_t* __restrict__ seg_ptr,
                                   const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                   const int32_t* __restrict__ out_rank, const int32_t* __restrict__ lg_seg_ptr, int64_t E,
                                   int64_t T, int64_t* __restrict__ perm, int64_t* __restrict__ inv_lg,
                                   int32_t* __restrict__ bad) {
    const int64_t k = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (k >= T) return;
    const int64_t a = lg_u[k], b = lg_v[k];
    bool ok = a >= 0 && a < E && b >= 0 && b < E;
    int64_t t = 0;
    if (ok) {
        const int32_t e1 = (int32_t)inv[a], e2 = (int32_t)inv[b];
        const int32_t j = src[e2];
        ok = dst[e1] == j && e1 != e2;
        if (ok) {
            const int32_t base = seg_ptr[j];
            t = (int64_t)lg_seg_ptr[out_rank[e2]] + (e1 - base) - ((dst[e2] == j && e2 < e1) ? 1 : 0);
            ok = t >= 0 && t < T;
        }
    }
    if (ok) {  // a row claimed twice (a duplicated caller edge) is counted as bad: T edges, none bad => a bijection
        const unsigned long long before = atomicExch(reinterpret_cast<unsigned long long*>(perm + t), (unsigned long long)k);
        ok = before == ~0ull;
    }
    if (ok) {
        inv_lg[k] = t;
    } else {
        inv_lg[k] = -1;
        atomicAdd(bad, 1);  // (integer count: order independent)
    }
}

struct Scratch {
    size_t iota, keys, vals, cnt_seg, cnt_out, out_rank, prim, total, prim_bytes;
};

Scratch scratch_for(int64_t N, int64_t E) {
    Scratch s{};
    size_t off = 0;
    const size_t e4 = al256((size_t)(E + 1) * 4);
    s.iota = off, off += e4;
    s.keys = off, off += e4;
    s.vals = off, off += e4;
    s.cnt_seg = off, off += e4;
    s.cnt_out = off, off += e4;
    s.out_rank = off, off += e4;
    size_t sort_b = 0, scan_b = 0;
    int32_t* ip = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, sort_b, ip, ip, ip, ip, (size_t)E, 0, 32, (hipStream_t)0);
    (void)rocprim::exclusive_scan(nullptr, scan_b, ip, ip, (int32_t)0, (size_t)(E + 1), rocprim::plus<int32_t>(), (hipStream_t)0);
    s.prim_bytes = sort_b > scan_b ? sort_b : scan_b;
    s.prim = off, off += al256(s.prim_bytes);
    s.total = off;
    (void)N;
    return s;
}

inline unsigned bits_for(int64_t n) {
    unsigned b = 1;
    while (((int64_t)1 << b) < n && b < 31) ++b;
    return b;
}

}  // namespace

extern "C" {

size_t alignn_stage_batch_workspace(int64_t n_nodes, int64_t n_edges) { return scratch_for(n_nodes, n_edges).total; }

int alignn_stage_batch(const int32_t* u, const int32_t* v, const float* r, int64_t N, int64_t E, int64_t T,
                       int32_t* seg_ptr, int32_t* src, int32_t* dst, int32_t* out_ptr, int32_t* out_slot, int64_t* perm,
                       int64_t* inv, float* r_canon, int32_t* lg_seg_ptr, int32_t* lg_src, int32_t* lg_dst,
                       int32_t* lg_out_ptr, int32_t* lg_out_slot, int32_t* lg_seg_rank, int64_t* lg_ident, float* h,
                       int32_t* out_rank_out, void* workspace, size_t workspace_bytes, alignn_stream_t stream) {
    if (N <= 0 || E <= 0 || T < 0 || E >= ((int64_t)1 << 31) - 1 || T >= ((int64_t)1 << 31) - 1 || N >= ((int64_t)1 << 31) - 1)
        return (int)hipErrorInvalidValue;
    if (!u || !v || !seg_ptr || !src || !dst || !out_ptr || !out_slot || !perm || !inv || !lg_seg_ptr || !lg_src || !lg_dst ||
        !lg_out_ptr || !lg_out_slot || !lg_seg_rank || !workspace || (r != nullptr && r_canon == nullptr) ||
        (h != nullptr && r == nullptr))
        return (int)hipErrorInvalidValue;
    const Scratch sc = scratch_for(N, E);
    if (workspace_bytes < sc.total) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    char* ws = static_cast<char*>(workspace);
    int32_t* iota = reinterpret_cast<int32_t*>(ws + sc.iota);
    int32_t* keys = reinterpret_cast<int32_t*>(ws + sc.keys);
    int32_t* vals = reinterpret_cast<int32_t*>(ws + sc.vals);
    int32_t* cnt_seg = reinterpret_cast<int32_t*>(ws + sc.cnt_seg);
    int32_t* cnt_out = reinterpret_cast<int32_t*>(ws + sc.cnt_out);
    int32_t* out_rank = out_rank_out != nullptr ? out_rank_out : reinterpret_cast<int32_t*>(ws + sc.out_rank);
    void* prim = ws + sc.prim;
    size_t pb = sc.prim_bytes;
    const unsigned nbits = bits_for(N);
    const int32_t E32 = (int32_t)E, N32 = (int32_t)N;
    hipError_t e;
    hipLaunchKernelGGL(iota_kernel, dim3(blocks(E)), dim3(kT), 0, st, iota, E);
    // ---- g: stable sort of the bonds by destination atom -> slots; segments by binary search in the sorted keys
    e = rocprim::radix_sort_pairs(prim, pb, v, dst, iota, vals, (size_t)E, 0, nbits, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(fill_g_kernel, dim3(blocks(E)), dim3(kT), 0, st, u, vals, r, E32, src, perm, inv, r_canon);
    hipLaunchKernelGGL(ptr_from_sorted_kernel, dim3(blocks(N + 1)), dim3(kT), 0, st, dst, E32, N32, seg_ptr);
    // ---- by-source view: stable sort of the slots by source atom
    pb = sc.prim_bytes;
    e = rocprim::radix_sort_pairs(prim, pb, src, keys, iota, out_slot, (size_t)E, 0, nbits, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(ptr_from_sorted_kernel, dim3(blocks(N + 1)), dim3(kT), 0, st, keys, E32, N32, out_ptr);
    // ---- L(g): row counts per segment / per source node, their prefix sums, then every row by index arithmetic
    hipLaunchKernelGGL(lg_counts_kernel, dim3(blocks(E + 1)), dim3(kT), 0, st, seg_ptr, src, dst, out_ptr, out_slot, E32, cnt_seg,
                       cnt_out, out_rank);
    pb = sc.prim_bytes;
    e = rocprim::exclusive_scan(prim, pb, cnt_seg, lg_seg_ptr, (int32_t)0, (size_t)(E + 1), rocprim::plus<int32_t>(), st);
    if (e != hipSuccess) return (int)e;
    pb = sc.prim_bytes;
    e = rocprim::exclusive_scan(prim, pb, cnt_out, lg_out_ptr, (int32_t)0, (size_t)(E + 1), rocprim::plus<int32_t>(), st);
    if (e != hipSuccess) return (int)e;
    if (T > 0) {
        hipLaunchKernelGGL(lg_rows_kernel, dim3(blocks(T)), dim3(kT), 0, st, seg_ptr, src, dst, out_slot, lg_seg_ptr, E32, T,
                           lg_src, lg_dst, lg_seg_rank, lg_ident);
        hipLaunchKernelGGL(lg_out_slot_kernel, dim3(blocks(T)), dim3(kT), 0, st, seg_ptr, src, dst, out_ptr, out_slot, out_rank,
                           lg_seg_ptr, lg_out_ptr, E32, T, lg_out_slot);
    }
    ALIGNN_CHECK_LAUNCH();
    // ---- bond-angle cosines on the canonical rows (compute_bond_cosines, alignn/graphs.py:847-864)
    if (h != nullptr && T > 0) return alignn_bond_cosine_fwd(r_canon, lg_src, lg_dst, h, T, stream);
    return 0;
}

int alignn_map_line_graph_rows(const int64_t* lg_u, const int64_t* lg_v, const int64_t* inv, const int32_t* seg_ptr,
                               const int32_t* src, const int32_t* dst, const int32_t* out_rank, const int32_t* lg_seg_ptr,
                               int64_t n_edges, int64_t n_triplets, int64_t* perm, int64_t* inv_lg, int32_t* bad,
                               alignn_stream_t stream) {
    if (n_triplets <= 0) return 0;
    if (!lg_u || !lg_v || !inv || !seg_ptr || !src || !dst || !out_rank || !lg_seg_ptr || !perm || !inv_lg || !bad || n_edges <= 0)
        return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(map_lg_rows_kernel, dim3(blocks(n_triplets)), dim3(kT), 0, (hipStream_t)stream, lg_u, lg_v, inv, seg_ptr, src,
                       dst, out_rank, lg_seg_ptr, n_edges, n_triplets, perm, inv_lg, bad);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
