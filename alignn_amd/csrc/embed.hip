// Featurisation and readout kernels: RBF expansion, bond length, per-graph mean pooling, row gather.
// All HBM-bound; one float4 (or one scalar) per lane, grid-stride.
//
// Reference: RBFExpansion.forward alignn/models/utils.py:40-44; torch.norm(r, dim=1)
// alignn/models/alignn.py:313; dgl.nn.AvgPooling alignn/models/alignn.py:325.
#include "common.h"
#include "../../include/alignn_hip.h"

namespace {

inline int grid_for(int64_t total, int threads = 256, int cap = 2048) {
    int64_t g = (total + threads - 1) / threads;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

__global__ void rbf_fwd_kernel(const float* __restrict__ d, const float* __restrict__ centers, float gamma,
                               float* __restrict__ out, int64_t rows, int bins) {
    const int64_t total = rows * bins;
    const RowQuad rq(bins);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r;
        int k;
        rq.split(i, total, r, k);
        float t = d[r] - centers[k];
        out[i] = __expf(-gamma * t * t);
    }
}

__global__ void norm3_fwd_kernel(const float* __restrict__ v, float* __restrict__ out, int64_t rows) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
        float x = v[3 * r], y = v[3 * r + 1], z = v[3 * r + 2];
        out[r] = sqrtf(x * x + y * y + z * z);
    }
}

// one block of four waves per graph; lane l owns feature quads l, l + 64, ...; wave w takes the graph's nodes beg + w, beg + w + 4,
// ... (two interleaved partial sums each: eight independent chains instead of one walk over all nodes - 50 -> ~10 us for a
// 200-atom cell), the partial sums are added in a fixed order: ((w0 + w1) + (w2 + w3)) of (even + odd)
__global__ __launch_bounds__(256) void segment_mean_fwd_kernel(const float* __restrict__ X, const int32_t* __restrict__ gptr,
                                                               float* __restrict__ out, int H) {
    __shared__ float4 sh[4][64];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int beg = gptr[b], end = gptr[b + 1];
    const float inv = end > beg ? 1.0f / (float)(end - beg) : 0.0f;
    for (int q0 = 0; q0 < (H >> 2); q0 += 64) {
        const int q = q0 + lane;
        float4 a0 = f4_zero(), a1 = f4_zero();
        if (q < (H >> 2)) {
            int i = beg + wave;
            for (; i + 4 < end; i += 8) {
                a0 = f4_add(a0, f4_ld(X + (int64_t)i * H + q * 4));
                a1 = f4_add(a1, f4_ld(X + (int64_t)(i + 4) * H + q * 4));
            }
            if (i < end) a0 = f4_add(a0, f4_ld(X + (int64_t)i * H + q * 4));
        }
        sh[wave][lane] = f4_add(a0, a1);
        __syncthreads();
        if (wave == 0 && q < (H >> 2)) {
            const float4 acc = f4_add(f4_add(sh[0][lane], sh[1][lane]), f4_add(sh[2][lane], sh[3][lane]));
            f4_st(out + (int64_t)b * H + q * 4, make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv));
        }
        __syncthreads();
    }
}

__global__ void segment_mean_bwd_kernel(const float* __restrict__ G, const int32_t* __restrict__ gptr,
                                        float* __restrict__ GX, int H) {
    const int b = blockIdx.x;
    const int beg = gptr[b], end = gptr[b + 1];
    const float inv = end > beg ? 1.0f / (float)(end - beg) : 0.0f;
    const int Q = H >> 2;
    const int total = (end - beg) * Q;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        int r = i / Q, q = i - r * Q;
        float4 g = f4_ld(G + (int64_t)b * H + q * 4);
        f4_st(GX + (int64_t)(beg + r) * H + q * 4, make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv));
    }
}

// compute_bond_cosines (alignn/graphs.py:847-864) on L(g): h = clamp(-r[e1].r[e2] / (|r[e1]||r[e2]|), -1, 1)
__global__ void bond_cosine_kernel(const float* __restrict__ r, const int32_t* __restrict__ e1,
                                   const int32_t* __restrict__ e2, float* __restrict__ h, int64_t T) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < T; k += (int64_t)gridDim.x * blockDim.x) {
        const float* a = r + 3 * (int64_t)e1[k];
        const float* b = r + 3 * (int64_t)e2[k];
        const float ax = -a[0], ay = -a[1], az = -a[2];
        const float dot = ax * b[0] + ay * b[1] + az * b[2];
        const float na = sqrtf(ax * ax + ay * ay + az * az), nb = sqrtf(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
        float c = dot / (na * nb);
        h[k] = fminf(fmaxf(c, -1.0f), 1.0f);
    }
}

// ---- first derivatives of the featurisation w.r.t. the geometry (force-field inference: F = -dE/dr) ----
// gd[r] = sum_k G[r,k] * d/dd exp(-gamma (d-c_k)^2) = sum_k G[r,k] * rbf_k * (-2 gamma (d - c_k))
__global__ void rbf_bwd_kernel(const float* __restrict__ d, const float* __restrict__ centers, float gamma,
                               const float* __restrict__ G, float* __restrict__ gd, int64_t rows, int bins) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
        const float x = d[r];
        float acc = 0.0f;
        if ((bins & 3) == 0) {  // rows are 16-byte aligned: four bins per load (the scalar form reads 4 of every 16 bytes it fetches)
            for (int k = 0; k < bins; k += 4) {
                const float4 g = f4_ld(G + r * bins + k);
                const float t0 = x - centers[k], t1 = x - centers[k + 1], t2 = x - centers[k + 2], t3 = x - centers[k + 3];
                acc += g.x * __expf(-gamma * t0 * t0) * (-2.0f * gamma * t0);
                acc += g.y * __expf(-gamma * t1 * t1) * (-2.0f * gamma * t1);
                acc += g.z * __expf(-gamma * t2 * t2) * (-2.0f * gamma * t2);
                acc += g.w * __expf(-gamma * t3 * t3) * (-2.0f * gamma * t3);
            }
        } else {
            for (int k = 0; k < bins; ++k) {
                const float t = x - centers[k];
                acc += G[r * bins + k] * __expf(-gamma * t * t) * (-2.0f * gamma * t);
            }
        }
        gd[r] = acc;
    }
}

// gv[r,:] = g[r] * v[r,:] / |v[r]|
__global__ void norm3_bwd_kernel(const float* __restrict__ v, const float* __restrict__ g, float* __restrict__ gv,
                                 int64_t rows) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
        const float x = v[3 * r], y = v[3 * r + 1], z = v[3 * r + 2];
        const float s = g[r] / sqrtf(x * x + y * y + z * z);
        gv[3 * r] = s * x, gv[3 * r + 1] = s * y, gv[3 * r + 2] = s * z;
    }
}

// per triplet k (a = r[e1], b = r[e2], c = -a.b/(|a||b|)):  ga[k] = gh[k] dc/da,  gb[k] = gh[k] dc/db
//   dc/da = -b/(|a||b|) - c a/|a|^2 ,  dc/db = -a/(|a||b|) - c b/|b|^2 ;  zero where the clamp is active
__global__ void bond_cosine_bwd_kernel(const float* __restrict__ r, const int32_t* __restrict__ e1,
                                       const int32_t* __restrict__ e2, const float* __restrict__ gh,
                                       float* __restrict__ ga, float* __restrict__ gb, int64_t T) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < T; k += (int64_t)gridDim.x * blockDim.x) {
        const float* a = r + 3 * (int64_t)e1[k];
        const float* b = r + 3 * (int64_t)e2[k];
        const float na2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2], nb2 = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
        const float inv = 1.0f / (sqrtf(na2) * sqrtf(nb2));
        const float c = -(a[0] * b[0] + a[1] * b[1] + a[2] * b[2]) * inv;
        const float g = (c >= -1.0f && c <= 1.0f) ? gh[k] : 0.0f;
        const float ca = c / na2, cb = c / nb2;
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            ga[3 * k + x] = g * (-b[x] * inv - ca * a[x]);
            gb[3 * k + x] = g * (-a[x] * inv - cb * b[x]);
        }
    }
}

// out[node(s), f] = sum_{k in [ptr[s], ptr[s+1])} vals[slot ? slot[k] : k, f]   (generic width F, fixed order)
__global__ void segment_sum_kernel(const float* __restrict__ vals, int64_t ldv, const int32_t* __restrict__ ptr,
                                   const int32_t* __restrict__ slot, const int32_t* __restrict__ node,
                                   float* __restrict__ out, int64_t ldo, int64_t n_seg, int F) {
    const int64_t total = n_seg * F;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i / F;
        const int f = (int)(i - s * F);
        float acc = 0.0f;
        for (int k = ptr[s]; k < ptr[s + 1]; ++k) acc += vals[(int64_t)(slot ? slot[k] : k) * ldv + f];
        out[(int64_t)(node ? node[s] : s) * ldo + f] = acc;
    }
}

// The same sum for FEW, LONG segments (per-crystal reductions: 16 crystals x 9 stress components over thousands of
// bonds each - the one-thread-per-output kernel above walks such a segment serially: 745 us at 16 x 200-atom crystals).
// One workgroup per segment: thread t owns feature t % F and walks the rows t / F, t / F + G, ... (G = 256 / F row
// groups); the G partial sums are added in group order through LDS - fixed order, bit-reproducible.
__global__ __launch_bounds__(256) void segment_sum_long_kernel(const float* __restrict__ vals, int64_t ldv,
                                                               const int32_t* __restrict__ ptr,
                                                               const int32_t* __restrict__ slot,
                                                               const int32_t* __restrict__ node, float* __restrict__ out,
                                                               int64_t ldo, int F) {
    __shared__ float sh[256];
    const int s = blockIdx.x;
    const int G = 256 / F;
    const int f = threadIdx.x % F, g = threadIdx.x / F;
    float acc = 0.0f;
    if (g < G) {
        const int beg = ptr[s], end = ptr[s + 1];
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int k = beg + g;
        for (; k + 3 * G < end; k += 4 * G) {
            a0 += vals[(int64_t)(slot ? slot[k] : k) * ldv + f];
            a1 += vals[(int64_t)(slot ? slot[k + G] : k + G) * ldv + f];
            a2 += vals[(int64_t)(slot ? slot[k + 2 * G] : k + 2 * G) * ldv + f];
            a3 += vals[(int64_t)(slot ? slot[k + 3 * G] : k + 3 * G) * ldv + f];
        }
        for (; k < end; k += G) a0 += vals[(int64_t)(slot ? slot[k] : k) * ldv + f];
        acc = (a0 + a1) + (a2 + a3);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (g == 0) {
        for (int q = 1; q < G; ++q) acc += sh[q * F + f];
        out[(int64_t)(node ? node[s] : s) * ldo + f] = acc;
    }
}

__global__ void gather_rows_kernel(const float* __restrict__ in, const int32_t* __restrict__ perm,
                                   float* __restrict__ out, int64_t rows, int F) {
    const int64_t total = rows * F;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i / F;
        int f = (int)(i - r * F);
        out[i] = in[(int64_t)perm[r] * F + f];
    }
}

// out[r, 0:F] = in[perm[r], 0:F] with leading dimensions (a column block of a wider matrix); float4 per thread
__global__ void gather_rows_ld_kernel(const float* __restrict__ in, int64_t ld_in, const int32_t* __restrict__ perm,
                                      float* __restrict__ out, int64_t ld_out, int64_t rows, int F) {
    const int Q = F >> 2;
    const RowQuad rq(Q);
    const int64_t total = rows * Q;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r;
        int q;
        rq.split(i, total, r, q);
        f4_st(out + r * ld_out + q * 4, f4_ld(in + (int64_t)perm[r] * ld_in + q * 4));
    }
}

}  // namespace

extern "C" {

const char* alignn_version(void) { return "alignn_hip 0.1 gfx950"; }

int alignn_rbf_fwd(const float* d, const float* centers, float gamma, float* out, int64_t rows, int bins,
                   alignn_stream_t stream) {
    if (bins <= 0 || rows < 0) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(rbf_fwd_kernel, dim3(grid_for(rows * bins)), dim3(256), 0, (hipStream_t)stream, d, centers,
                       gamma, out, rows, bins);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_norm3_fwd(const float* v, float* out, int64_t rows, alignn_stream_t stream) {
    if (rows == 0) return 0;
    hipLaunchKernelGGL(norm3_fwd_kernel, dim3(grid_for(rows)), dim3(256), 0, (hipStream_t)stream, v, out, rows);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_segment_mean_fwd(const float* X, const int32_t* graph_ptr, float* out, int B, int H,
                            alignn_stream_t stream) {
    if (B <= 0 || (H & 3)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(segment_mean_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, X, graph_ptr, out, H);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_segment_mean_bwd(const float* G, const int32_t* graph_ptr, float* GX, int B, int H,
                            alignn_stream_t stream) {
    if (B <= 0 || (H & 3)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(segment_mean_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, G, graph_ptr, GX, H);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_bond_cosine_fwd(const float* r, const int32_t* e1, const int32_t* e2, float* h, int64_t T,
                            alignn_stream_t stream) {
    if (T == 0) return 0;
    hipLaunchKernelGGL(bond_cosine_kernel, dim3(grid_for(T)), dim3(256), 0, (hipStream_t)stream, r, e1, e2, h, T);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_rbf_bwd(const float* d, const float* centers, float gamma, const float* G, float* gd, int64_t rows, int bins,
                   alignn_stream_t stream) {
    if (bins <= 0) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(rbf_bwd_kernel, dim3(grid_for(rows)), dim3(256), 0, (hipStream_t)stream, d, centers, gamma, G, gd,
                       rows, bins);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_norm3_bwd(const float* v, const float* g, float* gv, int64_t rows, alignn_stream_t stream) {
    if (rows == 0) return 0;
    hipLaunchKernelGGL(norm3_bwd_kernel, dim3(grid_for(rows)), dim3(256), 0, (hipStream_t)stream, v, g, gv, rows);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_bond_cosine_bwd(const float* r, const int32_t* e1, const int32_t* e2, const float* gh, float* ga, float* gb,
                           int64_t T, alignn_stream_t stream) {
    if (T == 0) return 0;
    hipLaunchKernelGGL(bond_cosine_bwd_kernel, dim3(grid_for(T)), dim3(256), 0, (hipStream_t)stream, r, e1, e2, gh, ga,
                       gb, T);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_segment_sum(const float* vals, int64_t ldv, const int32_t* ptr, const int32_t* slot, const int32_t* node,
                       float* out, int64_t ldo, int64_t n_seg, int F, alignn_stream_t stream) {
    if (F <= 0 || n_seg < 0) return (int)hipErrorInvalidValue;
    if (n_seg == 0) return 0;
    if (n_seg * F <= 4096 && F <= 64 && n_seg <= 4096) {
        // few outputs: whatever the segments' lengths, a workgroup per segment is never slower than a thread per output
        hipLaunchKernelGGL(segment_sum_long_kernel, dim3((int)n_seg), dim3(256), 0, (hipStream_t)stream, vals, ldv, ptr, slot, node,
                           out, ldo, F);
        ALIGNN_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(segment_sum_kernel, dim3(grid_for(n_seg * F)), dim3(256), 0, (hipStream_t)stream, vals, ldv, ptr,
                       slot, node, out, ldo, n_seg, F);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_gather_rows_ld(const float* in, int64_t ld_in, const int32_t* perm, float* out, int64_t ld_out, int64_t rows, int F,
                          alignn_stream_t stream) {
    if (F <= 0 || (F & 3) || (ld_in & 3) || (ld_out & 3)) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(gather_rows_ld_kernel, dim3(grid_for(rows * (F >> 2), 256, 4096)), dim3(256), 0, (hipStream_t)stream, in,
                       ld_in, perm, out, ld_out, rows, F);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

int alignn_gather_rows(const float* in, const int32_t* perm, float* out, int64_t rows, int F,
                       alignn_stream_t stream) {
    if (F <= 0) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(rows * F)), dim3(256), 0, (hipStream_t)stream, in, perm, out,
                       rows, F);
    ALIGNN_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
