// Single-head GAT aggregation for gfx950: edge softmax + weighted neighbour sum, fused.
//
// Replaces PyG GATConv(heads=1, concat=False) as pygda's GNNBase(gnn='gat') uses it
// (pygda/nn/gnn_base.py:80-87).  Same mapping as the SpMM kernel: a lane group of G lanes owns
// one destination row and VEC feature columns per lane; the attention logits are scalars per
// node, so the softmax statistics of a row cost one 4-byte gather per neighbour.
//   forward   pass 1: m = max_k e_k;  pass 2: z = sum_k exp(e_k - m);  pass 3: out = sum_k alpha_k h[col_k]
//             (three walks over the row's neighbour list; the list is L2-resident after the first)
//   backward  k_gat_bwd_dst (group per destination):  dalpha_k = gout_i . h[col_k] (group butterfly),
//             s = sum_k alpha_k dalpha_k,  dpre_k = alpha_k (dalpha_k - s) * LeakyReLU'(pre_k),
//             ga_dst[i] = sum_k dpre_k
//             k_gat_bwd_src (group per source, transposed CSR + edge map):
//             gh[j] = sum_k' alpha[map k'] gout[dst_k'],  ga_src[j] = sum_k' dpre[map k']
// All sums are sequential in CSR order: deterministic, no atomics.
#include "gda_common.h"

namespace {

constexpr int TB = 256;

template <int G>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, G));
    return v;
}
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, G);
    return v;
}

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : slope * v; }

template <int G>
__global__ void __launch_bounds__(TB)
k_gat_fwd(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx, int64_t n_rows, int d,
          const float* __restrict__ h, const float* __restrict__ a_src, const float* __restrict__ a_dst,
          float slope, float* __restrict__ out, float* __restrict__ alpha) {
    const int lane = threadIdx.x % G;
    const int64_t row = (int64_t)blockIdx.x * (TB / G) + threadIdx.x / G;
    const bool live = row < n_rows;
    const int32_t start = live ? rowptr[row] : 0, end = live ? rowptr[row + 1] : 0;
    const float ad = live ? a_dst[row] : 0.f;
    // softmax statistics over the incoming edges
    float m = -INFINITY;
    for (int32_t k = start + lane; k < end; k += G) m = fmaxf(m, leaky(a_src[colidx[k]] + ad, slope));
    m = group_max<G>(m);
    float z = 0.f;
    for (int32_t k = start + lane; k < end; k += G) z += __expf(leaky(a_src[colidx[k]] + ad, slope) - m);
    z = group_sum<G>(z);
    const float inv_z = z > 0.f ? 1.f / z : 0.f;
    for (int32_t k = start + lane; k < end; k += G)
        alpha[k] = __expf(leaky(a_src[colidx[k]] + ad, slope) - m) * inv_z;
    // weighted neighbour sum, columns strided over the group
    for (int c = lane; c < d; c += G) {
        float acc = 0.f;
        for (int32_t k = start; k < end; ++k) {
            const int32_t j = colidx[k];
            const float a = __expf(leaky(a_src[j] + ad, slope) - m) * inv_z;
            acc = fmaf(a, h[(int64_t)j * d + c], acc);
        }
        if (live) out[row * d + c] = acc;
    }
}

template <int G>
__global__ void __launch_bounds__(TB)
k_gat_bwd_dst(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx, int64_t n_rows, int d,
              const float* __restrict__ h, const float* __restrict__ a_src, const float* __restrict__ a_dst,
              float slope, const float* __restrict__ alpha, const float* __restrict__ gout,
              float* __restrict__ ga_dst, float* __restrict__ dpre) {
    const int lane = threadIdx.x % G;
    const int64_t row = (int64_t)blockIdx.x * (TB / G) + threadIdx.x / G;
    const bool live = row < n_rows;
    const int32_t start = live ? rowptr[row] : 0, end = live ? rowptr[row + 1] : 0;
    const float ad = live ? a_dst[row] : 0.f;
    // pass 1: dalpha_k = gout_i . h[col_k]; s = sum alpha dalpha   (dalpha parked in dpre)
    float s = 0.f;
    for (int32_t k = start; k < end; ++k) {
        const int32_t j = colidx[k];
        float part = 0.f;
        for (int c = lane; c < d; c += G) part = fmaf(gout[row * d + c], h[(int64_t)j * d + c], part);
        const float da = group_sum<G>(part);
        if (lane == 0) dpre[k] = da;
        s = fmaf(alpha[k], da, s);
    }
    // pass 2: softmax + LeakyReLU backward
    float gad = 0.f;
    for (int32_t k = start; k < end; ++k) {
        const float da = __shfl(lane == 0 ? dpre[k] : 0.f, 0, G);
        const float pre = a_src[colidx[k]] + ad;
        const float g = alpha[k] * (da - s) * (pre > 0.f ? 1.f : slope);
        if (lane == 0) dpre[k] = g;
        gad += g;
    }
    if (live && lane == 0) ga_dst[row] = gad;
}

template <int G>
__global__ void __launch_bounds__(TB)
k_gat_bwd_src(const int32_t* __restrict__ t_rowptr, const int32_t* __restrict__ t_colidx,
              const int32_t* __restrict__ t_to_fwd, int64_t n_rows, int d,
              const float* __restrict__ alpha, const float* __restrict__ dpre, const float* __restrict__ gout,
              float* __restrict__ gh, float* __restrict__ ga_src) {
    const int lane = threadIdx.x % G;
    const int64_t row = (int64_t)blockIdx.x * (TB / G) + threadIdx.x / G;
    const bool live = row < n_rows;
    const int32_t start = live ? t_rowptr[row] : 0, end = live ? t_rowptr[row + 1] : 0;
    float gas = 0.f;
    for (int32_t k = start; k < end; ++k) gas += dpre[t_to_fwd[k]];
    if (live && lane == 0) ga_src[row] = gas;
    for (int c = lane; c < d; c += G) {
        float acc = 0.f;
        for (int32_t k = start; k < end; ++k)
            acc = fmaf(alpha[t_to_fwd[k]], gout[(int64_t)t_colidx[k] * d + c], acc);
        if (live) gh[row * d + c] = acc;
    }
}

int pick_group(int64_t d) { return d > 32 ? 64 : d > 16 ? 32 : d > 8 ? 16 : 8; }

}  // namespace

#define GAT_SWITCH(d, CALL)                  \
    do {                                     \
        switch (pick_group(d)) {             \
            case 64: CALL(64); break;        \
            case 32: CALL(32); break;        \
            case 16: CALL(16); break;        \
            default: CALL(8); break;         \
        }                                    \
    } while (0)

extern "C" int gda_gat_fwd_f32(const int32_t* rowptr, const int32_t* colidx, int64_t n_rows, int64_t d,
                               const float* h, const float* a_src, const float* a_dst, float slope,
                               float* out, float* alpha, gda_stream_t stream_) {
    if (n_rows < 0 || d < 0 || n_rows >= INT32_MAX || d >= INT32_MAX) return GDA_E_SIZE;
    if (n_rows == 0 || d == 0) return GDA_OK;
    if (!rowptr || !colidx || !h || !a_src || !a_dst || !out || !alpha) return GDA_E_NULL;
    hipStream_t s = (hipStream_t)stream_;
#define CALL(G) k_gat_fwd<G><<<(unsigned)gda_cdiv(n_rows, TB / G), TB, 0, s>>>(rowptr, colidx, n_rows, (int)d, h, \
                                                                              a_src, a_dst, slope, out, alpha)
    GAT_SWITCH(d, CALL);
#undef CALL
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_gat_bwd_f32(const int32_t* rowptr, const int32_t* colidx,
                               const int32_t* t_rowptr, const int32_t* t_colidx, const int32_t* t_to_fwd,
                               int64_t n_rows, int64_t d, const float* h, const float* a_src,
                               const float* a_dst, float slope, const float* alpha, const float* gout,
                               float* gh, float* ga_src, float* ga_dst, float* dpre, gda_stream_t stream_) {
    if (n_rows < 0 || d < 0 || n_rows >= INT32_MAX || d >= INT32_MAX) return GDA_E_SIZE;
    if (n_rows == 0 || d == 0) return GDA_OK;
    if (!rowptr || !colidx || !t_rowptr || !t_colidx || !t_to_fwd || !h || !a_src || !a_dst || !alpha || !gout ||
        !gh || !ga_src || !ga_dst || !dpre) return GDA_E_NULL;
    hipStream_t s = (hipStream_t)stream_;
#define CALL(G) k_gat_bwd_dst<G><<<(unsigned)gda_cdiv(n_rows, TB / G), TB, 0, s>>>(                        \
        rowptr, colidx, n_rows, (int)d, h, a_src, a_dst, slope, alpha, gout, ga_dst, dpre)
    GAT_SWITCH(d, CALL);
#undef CALL
    GDA_LAUNCH_CHECK();
#define CALL(G) k_gat_bwd_src<G><<<(unsigned)gda_cdiv(n_rows, TB / G), TB, 0, s>>>(                        \
        t_rowptr, t_colidx, t_to_fwd, n_rows, (int)d, alpha, dpre, gout, gh, ga_src)
    GAT_SWITCH(d, CALL);
#undef CALL
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}
