// Degree-normalised Laplacian smoothness of node features over an edge list (fp32, gfx950):
//     loss = 1/2 * sum_e || f[row_e] * dinv[row_e] - f[col_e] * dinv[col_e] ||^2 ,
//     dinv[i] = (#edges with row == i)^-1/2, inf -> 0.
// Replaces TDSS.compute_laplacian_loss (pygda/models/tdss.py:435-454), which materialises four
// [E, d] temporaries (two gathers, the difference, its square) on a 2-hop / random-walk graph
// with ~10-100x the edges of the input graph, and keeps three of them for autograd.
//
// Same gather pattern as the aggregation kernel (gda_spmm.hip): a lane group owns one node, walks
// its CSR entry list, every lane holds VEC feature columns, a neighbour row is one coalesced
// G*VEC*4-byte read.  Forward reads f once per entry and writes one float per node; backward is
//     grad_f[i] = gl * dinv[i] * ( sum_{c in R(i)} (g_i - g_c) + sum_{r in C(i)} (g_i - g_r) ),  g = f * dinv,
// walking the by-row list R(i) and the by-column list C(i) -- nothing of size [E, d] exists.
// HBM bound: algorithmic bytes fwd = nnz*4 + N*(d*4 + 12), bwd = 2*nnz*4 + 2*N*d*4.
#include <algorithm>

#include "gda_common.h"

namespace {

constexpr int TB = 256;
constexpr int UNROLL = 4;

template <int VEC>
__device__ __forceinline__ void ld(float (&r)[VEC], const float* __restrict__ p) {
    if constexpr (VEC == 4) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
    } else {
        r[0] = *p;
    }
}

// pygda/models/tdss.py:441-444: scatter_add of ones over `row`, pow(-0.5), inf -> 0
__global__ void k_dinv(const int32_t* __restrict__ rowptr_r, int64_t N, float* __restrict__ dinv) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int32_t deg = rowptr_r[i + 1] - rowptr_r[i];
    dinv[i] = deg > 0 ? 1.0f / sqrtf((float)deg) : 0.0f;
}

// accumulate sign * sum_k (g_i - g_{col[k]}) (MODE 1, per column) or sum_k |g_i - g_col|^2 (MODE 0)
template <int G, int VEC, int MODE>
__device__ __forceinline__ void walk(const int32_t* __restrict__ colidx, int32_t start, int32_t end,
                                     const float* __restrict__ f, int64_t ldf,
                                     const float* __restrict__ dinv, int c, bool col_ok,
                                     const float (&gi)[VEC], float (&acc)[VEC], int lane_in_group) {
    for (int32_t base = start; base < end; base += G) {
        const int32_t k = base + lane_in_group;
        const int32_t my_col = k < end ? colidx[k] : 0;
        const float my_d = k < end ? dinv[my_col] : 0.0f;
        const int cnt = min((int32_t)G, end - base);
        int e = 0;
        for (; e + UNROLL <= cnt; e += UNROLL) {
            float xv[UNROLL][VEC], dc[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int32_t cu = __shfl(my_col, e + u, G);
                dc[u] = __shfl(my_d, e + u, G);
                if (col_ok) ld<VEC>(xv[u], f + (int64_t)cu * ldf + c);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const float diff = col_ok ? gi[v] - xv[u][v] * dc[u] : 0.0f;
                    acc[v] += MODE == 0 ? diff * diff : diff;
                }
        }
        for (; e < cnt; ++e) {
            const int32_t cu = __shfl(my_col, e, G);
            const float dc = __shfl(my_d, e, G);
            if (col_ok) {
                float xv[VEC];
                ld<VEC>(xv, f + (int64_t)cu * ldf + c);
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const float diff = gi[v] - xv[v] * dc;
                    acc[v] += MODE == 0 ? diff * diff : diff;
                }
            }
        }
    }
}

template <int G, int VEC>
__global__ void __launch_bounds__(TB)
k_lap_fwd(const int32_t* __restrict__ rowptr_r, const int32_t* __restrict__ colidx_r, int64_t N, int d,
          const float* __restrict__ f, int64_t ldf, const float* __restrict__ dinv,
          float* __restrict__ node_loss) {
    const int lane_in_group = threadIdx.x % G;
    const int64_t i = (int64_t)blockIdx.x * (TB / G) + threadIdx.x / G;
    const bool live = i < N;
    const int32_t start = live ? rowptr_r[i] : 0, end = live ? rowptr_r[i + 1] : 0;
    const float di = live ? dinv[i] : 0.0f;
    float total = 0.0f;
    for (int c0 = 0; c0 < d; c0 += G * VEC) {
        const int c = c0 + lane_in_group * VEC;
        const bool col_ok = live && c < d;
        float gi[VEC], acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) { gi[v] = 0.0f; acc[v] = 0.0f; }
        if (col_ok) {
            ld<VEC>(gi, f + i * ldf + c);
#pragma unroll
            for (int v = 0; v < VEC; ++v) gi[v] *= di;
        }
        walk<G, VEC, 0>(colidx_r, start, end, f, ldf, dinv, c, col_ok, gi, acc, lane_in_group);
#pragma unroll
        for (int v = 0; v < VEC; ++v) total += acc[v];
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) total += __shfl_xor(total, o, G);
    if (live && lane_in_group == 0) node_loss[i] = total;
}

// fixed-shape two-stage sum (double accumulators): same result whatever the launch timing
__global__ void __launch_bounds__(256)
k_sum_stage(const float* __restrict__ v, int64_t n, double* __restrict__ partial) {
    __shared__ double sh[256];
    double a = 0.0;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) a += (double)v[k];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

__global__ void __launch_bounds__(256)
k_sum_final(const double* __restrict__ partial, int n, float* __restrict__ loss) {
    __shared__ double sh[256];
    double a = 0.0;
    for (int k = threadIdx.x; k < n; k += 256) a += partial[k];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (float)(0.5 * sh[0]);              // tdss.py:454
}

template <int G, int VEC>
__global__ void __launch_bounds__(TB)
k_lap_bwd(const int32_t* __restrict__ rowptr_r, const int32_t* __restrict__ colidx_r,
          const int32_t* __restrict__ rowptr_c, const int32_t* __restrict__ colidx_c, int64_t N, int d,
          const float* __restrict__ f, int64_t ldf, const float* __restrict__ dinv,
          const float* __restrict__ grad_loss, float* __restrict__ grad_f, int64_t ldg) {
    const int lane_in_group = threadIdx.x % G;
    const int64_t i = (int64_t)blockIdx.x * (TB / G) + threadIdx.x / G;
    const bool live = i < N;
    const int32_t rs = live ? rowptr_r[i] : 0, re = live ? rowptr_r[i + 1] : 0;
    const int32_t cs = live ? rowptr_c[i] : 0, ce = live ? rowptr_c[i + 1] : 0;
    const float di = live ? dinv[i] : 0.0f;
    const float scale = *grad_loss * di;
    for (int c0 = 0; c0 < d; c0 += G * VEC) {
        const int c = c0 + lane_in_group * VEC;
        const bool col_ok = live && c < d;
        float gi[VEC], acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) { gi[v] = 0.0f; acc[v] = 0.0f; }
        if (col_ok) {
            ld<VEC>(gi, f + i * ldf + c);
#pragma unroll
            for (int v = 0; v < VEC; ++v) gi[v] *= di;
        }
        walk<G, VEC, 1>(colidx_r, rs, re, f, ldf, dinv, c, col_ok, gi, acc, lane_in_group);
        walk<G, VEC, 1>(colidx_c, cs, ce, f, ldf, dinv, c, col_ok, gi, acc, lane_in_group);
        if (col_ok) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) grad_f[i * ldg + c + v] = acc[v] * scale;
        }
    }
}

struct Ws {
    float* node_loss;
    double* partial;
};
constexpr int N_PARTIAL = 512;

Ws carve(void* ws, int64_t N) {
    Ws w;
    w.partial = (double*)ws;
    w.node_loss = (float*)((char*)ws + N_PARTIAL * sizeof(double));
    (void)N;
    return w;
}

bool vec4_ok(int d, const float* f, int64_t ldf) {
    return d % 4 == 0 && ldf % 4 == 0 && ((uintptr_t)f & 15) == 0;
}

}  // namespace

extern "C" size_t gda_laplacian_workspace_bytes(int64_t N) {
    if (N < 0) return 0;
    return N_PARTIAL * sizeof(double) + gda_align_up((size_t)N * sizeof(float), 256);
}

extern "C" int gda_laplacian_fwd_f32(const int32_t* rowptr_r, const int32_t* colidx_r, int64_t N, int d,
                                     const float* f, int64_t ldf, float* loss, float* dinv,
                                     void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    if (!loss || !dinv || !rowptr_r || (N > 0 && d > 0 && !f)) return GDA_E_NULL;
    if (N < 0 || N >= INT32_MAX || d < 0 || ldf < d) return GDA_E_SIZE;
    if (workspace_bytes < gda_laplacian_workspace_bytes(N) || !workspace) return GDA_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    if (N == 0 || d == 0) {
        GDA_HIP_TRY(hipMemsetAsync(loss, 0, sizeof(float), stream));
        if (N > 0) { k_dinv<<<(unsigned)gda_cdiv(N, 256), 256, 0, stream>>>(rowptr_r, N, dinv); GDA_LAUNCH_CHECK(); }
        return GDA_OK;
    }
    if (!colidx_r) return GDA_E_NULL;
    const Ws w = carve(workspace, N);
    k_dinv<<<(unsigned)gda_cdiv(N, 256), 256, 0, stream>>>(rowptr_r, N, dinv);
    GDA_LAUNCH_CHECK();
    if (vec4_ok(d, f, ldf) && d >= 64)
        k_lap_fwd<32, 4><<<(unsigned)gda_cdiv(N, TB / 32), TB, 0, stream>>>(rowptr_r, colidx_r, N, d, f, ldf, dinv, w.node_loss);
    else
        k_lap_fwd<16, 1><<<(unsigned)gda_cdiv(N, TB / 16), TB, 0, stream>>>(rowptr_r, colidx_r, N, d, f, ldf, dinv, w.node_loss);
    GDA_LAUNCH_CHECK();
    const int blocks = (int)std::min<int64_t>(N_PARTIAL, gda_cdiv(N, 256));
    k_sum_stage<<<blocks, 256, 0, stream>>>(w.node_loss, N, w.partial);
    GDA_LAUNCH_CHECK();
    k_sum_final<<<1, 256, 0, stream>>>(w.partial, blocks, loss);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_laplacian_bwd_f32(const int32_t* rowptr_r, const int32_t* colidx_r,
                                     const int32_t* rowptr_c, const int32_t* colidx_c, int64_t N, int d,
                                     const float* f, int64_t ldf, const float* dinv, const float* grad_loss,
                                     float* grad_f, int64_t ldg, gda_stream_t stream_) {
    if (N < 0 || d < 0) return GDA_E_SIZE;
    if (N == 0 || d == 0) return GDA_OK;
    if (!rowptr_r || !colidx_r || !rowptr_c || !colidx_c || !f || !dinv || !grad_loss || !grad_f) return GDA_E_NULL;
    if (N < 0 || N >= INT32_MAX || d < 0 || ldf < d || ldg < d) return GDA_E_SIZE;
    if (grad_f == f) return GDA_E_ALIAS;
    hipStream_t stream = (hipStream_t)stream_;
    if (vec4_ok(d, f, ldf) && d >= 64)
        k_lap_bwd<32, 4><<<(unsigned)gda_cdiv(N, TB / 32), TB, 0, stream>>>(rowptr_r, colidx_r, rowptr_c, colidx_c, N, d, f, ldf, dinv, grad_loss, grad_f, ldg);
    else
        k_lap_bwd<16, 1><<<(unsigned)gda_cdiv(N, TB / 16), TB, 0, stream>>>(rowptr_r, colidx_r, rowptr_c, colidx_c, N, d, f, ldf, dinv, grad_loss, grad_f, ldg);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}
