// View attention of UDAGCN's dual-view encoder, fused (gfx950).
//
// Replaces pygda/nn/attention.py:51-54
//     stacked = torch.stack(inputs, dim=1); weights = F.softmax(self.dense_weight(stacked), dim=1)
//     outputs = torch.sum(stacked * weights, dim=1)
// (K = 2 views in UDAGCN: the GCN stack's and the PPMI stack's node embeddings) and its autograd graph: six library
// launches forward (stack copy, the [N*K, h] x [h, 1] product through the BLAS, softmax, multiply, sum) and a dozen
// backward, all over [N, K, h] temporaries.  Here a lane group of 32 lanes owns a row: it reads the K view rows once
// (16-byte pieces, kept in registers), forms the K scores with a group butterfly, the softmax in registers and the
// weighted sum -- one pass forward, one pass backward, nothing of size [N, K, h] exists.
//
//   forward : s_k = x_k . w + b,  a = softmax_k(s),  out = sum_k a_k x_k            (a [N, K] kept for the backward)
//   backward: t_k = g . x_k,  ds_k = a_k (t_k - sum_j a_j t_j),  gx_k = a_k g + ds_k w,
//             gw = sum_rows sum_k ds_k x_k,  gb = sum_rows sum_k ds_k   (fixed-order two-stage sums: deterministic)
// HBM-bound: forward reads K h + writes h floats per row, backward reads (K + 1) h and writes K h.
// Envelope: 2 <= K <= 4 views, h % 4 == 0, h <= 512 (four 16-byte pieces per lane), 16-byte aligned rows.
#include "gda_common.h"

namespace {

constexpr int TB = 256;
constexpr int G = 32;                 // lanes per row
constexpr int MAXK = 4, MAXP = 4;     // views; 16-byte pieces per lane (h <= 32 * 4 * MAXP)

struct Views { const float* x[MAXK]; float* gx[MAXK]; int64_t ld[MAXK]; };

__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, G);
    return v;
}

template <int K>
__global__ void __launch_bounds__(TB)
k_attn_fwd(Views V, const float* __restrict__ w, const float* __restrict__ b, int64_t n, int h,
           float* __restrict__ out, int64_t ldo, float* __restrict__ att) {
    const int lane = threadIdx.x % G;
    const int64_t row = (int64_t)blockIdx.x * (TB / G) + threadIdx.x / G;
    if (row >= n) return;
    const int pieces = (h / 4 + G - 1) / G;
    float4 xv[K][MAXP];
    float s[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float dot = 0.f;
#pragma unroll
        for (int p = 0; p < MAXP; ++p) {
            const int c = (p * G + lane) * 4;
            xv[k][p] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p < pieces && c < h) {
                xv[k][p] = *reinterpret_cast<const float4*>(V.x[k] + row * V.ld[k] + c);
                const float4 wv = *reinterpret_cast<const float4*>(w + c);
                dot = fmaf(xv[k][p].x, wv.x, dot); dot = fmaf(xv[k][p].y, wv.y, dot);
                dot = fmaf(xv[k][p].z, wv.z, dot); dot = fmaf(xv[k][p].w, wv.w, dot);
            }
        }
        s[k] = group_sum(dot) + b[0];
    }
    float mx = s[0];
#pragma unroll
    for (int k = 1; k < K; ++k) mx = fmaxf(mx, s[k]);
    float den = 0.f, a[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { a[k] = __expf(s[k] - mx); den += a[k]; }
#pragma unroll
    for (int k = 0; k < K; ++k) a[k] = a[k] / den;
    if (lane < K) {
        float mine = a[0];
#pragma unroll
        for (int k = 1; k < K; ++k) mine = lane == k ? a[k] : mine;
        att[row * K + lane] = mine;
    }
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
        const int c = (p * G + lane) * 4;
        if (p < pieces && c < h) {
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < K; ++k) {           // sum over the views in view order, as torch.sum(dim=1) does
                o.x = fmaf(xv[k][p].x, a[k], o.x); o.y = fmaf(xv[k][p].y, a[k], o.y);
                o.z = fmaf(xv[k][p].z, a[k], o.z); o.w = fmaf(xv[k][p].w, a[k], o.w);
            }
            *reinterpret_cast<float4*>(out + row * ldo + c) = o;
        }
    }
}

// part: [blocks, h + 1] per-workgroup partial sums of gw | gb (rows of a workgroup added in row order)
template <int K>
__global__ void __launch_bounds__(TB)
k_attn_bwd(Views V, const float* __restrict__ w, const float* __restrict__ att, const float* __restrict__ g, int64_t ldg,
           int64_t n, int h, float* __restrict__ part) {
    extern __shared__ float sh[];                       // [TB / G][h + 1]: each row group's gw | gb contribution
    const int lane = threadIdx.x % G, grp = threadIdx.x / G;
    const int64_t row = (int64_t)blockIdx.x * (TB / G) + grp;
    const int pieces = (h / 4 + G - 1) / G;
    float* mine = sh + (size_t)grp * (h + 1);
    const bool live = row < n;
    float4 gv[MAXP], xv[K][MAXP];
    float t[K], a[K];
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
        const int c = (p * G + lane) * 4;
        gv[p] = (live && p < pieces && c < h) ? *reinterpret_cast<const float4*>(g + row * ldg + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float dot = 0.f;
#pragma unroll
        for (int p = 0; p < MAXP; ++p) {
            const int c = (p * G + lane) * 4;
            xv[k][p] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live && p < pieces && c < h) {
                xv[k][p] = *reinterpret_cast<const float4*>(V.x[k] + row * V.ld[k] + c);
                dot = fmaf(xv[k][p].x, gv[p].x, dot); dot = fmaf(xv[k][p].y, gv[p].y, dot);
                dot = fmaf(xv[k][p].z, gv[p].z, dot); dot = fmaf(xv[k][p].w, gv[p].w, dot);
            }
        }
        t[k] = group_sum(dot);
        a[k] = live ? att[row * K + k] : 0.f;
    }
    float mean = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) mean = fmaf(a[k], t[k], mean);
    float ds[K], dsum = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) { ds[k] = a[k] * (t[k] - mean); dsum += ds[k]; }
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
        const int c = (p * G + lane) * 4;
        if (p < pieces && c < h) {
            const float4 wv = *reinterpret_cast<const float4*>(w + c);
            float4 gw = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (live && V.gx[k]) {
                    float4 o;
                    o.x = fmaf(ds[k], wv.x, a[k] * gv[p].x); o.y = fmaf(ds[k], wv.y, a[k] * gv[p].y);
                    o.z = fmaf(ds[k], wv.z, a[k] * gv[p].z); o.w = fmaf(ds[k], wv.w, a[k] * gv[p].w);
                    *reinterpret_cast<float4*>(V.gx[k] + row * (int64_t)h + c) = o;      // gx_k: contiguous [n, h]
                }
                gw.x = fmaf(ds[k], xv[k][p].x, gw.x); gw.y = fmaf(ds[k], xv[k][p].y, gw.y);
                gw.z = fmaf(ds[k], xv[k][p].z, gw.z); gw.w = fmaf(ds[k], xv[k][p].w, gw.w);
            }
            mine[c + 0] = gw.x; mine[c + 1] = gw.y; mine[c + 2] = gw.z; mine[c + 3] = gw.w;
        }
    }
    if (lane == 0) mine[h] = dsum;
    __syncthreads();
    float* dst = part + (size_t)blockIdx.x * (h + 1);
    for (int c = threadIdx.x; c <= h; c += TB) {
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < TB / G; ++r) acc += sh[(size_t)r * (h + 1) + c];       // row order: fixed
        dst[c] = acc;
    }
}

// gw[c] = sum over the workgroups' partials: 16 chains per column (chain q the blocks q, q + 16, ... in order, eight loads in
// flight at a time), combined in a fixed tree.  (Four chains of dependent loads took 137 us at UDAGCN's 2,043 blocks.)
constexpr int FOLD_TB = 1024;
__global__ void __launch_bounds__(FOLD_TB)
k_attn_fold(const float* __restrict__ part, int blocks, int h, float* __restrict__ gw, float* __restrict__ gb) {
    __shared__ float red[FOLD_TB / 64][64];
    const int l = threadIdx.x & 63, c = blockIdx.x * 64 + l, q = threadIdx.x >> 6;
    constexpr int NQ = FOLD_TB / 64;
    float acc = 0.f;
    if (c <= h) {
        const float* src = part + c;
        const size_t st = (size_t)(h + 1);
        int b = q;
        for (; b + 7 * NQ < blocks; b += 8 * NQ) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(b + u * NQ) * st];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; b < blocks; b += NQ) acc += src[(size_t)b * st];
    }
    red[q][l] = acc;
    __syncthreads();
    for (int half = NQ / 2; half > 0; half >>= 1) {
        if (q < half) red[q][l] += red[q + half][l];
        __syncthreads();
    }
    if (q == 0 && c <= h) {
        if (c < h) gw[c] = red[0][l]; else gb[0] = red[0][l];
    }
}

int check(int n_views, const float* const* x, const int64_t* ld, int64_t n, int64_t h) {
    if (n_views < 2 || n_views > MAXK) return GDA_E_UNSUPPORTED;
    if (n < 0 || h <= 0) return GDA_E_SIZE;
    if (h % 4 || h > G * 4 * MAXP) return GDA_E_UNSUPPORTED;
    if (!x || !ld) return GDA_E_NULL;
    for (int k = 0; k < n_views; ++k) {
        if (!x[k]) return GDA_E_NULL;
        if (ld[k] < h) return GDA_E_SIZE;
        if (ld[k] % 4 || ((uintptr_t)x[k] & 15)) return GDA_E_UNSUPPORTED;
    }
    return GDA_OK;
}

}  // namespace

extern "C" size_t gda_attention_workspace_bytes(int64_t n, int64_t h) {
    if (n <= 0 || h <= 0) return 0;
    return (size_t)gda_cdiv(n, TB / G) * (size_t)(h + 1) * sizeof(float);
}

extern "C" int gda_attention_fuse_fwd_f32(int n_views, const float* const* x, const int64_t* ld, int64_t n, int64_t h,
                                          const float* w, const float* b, float* out, int64_t ldo, float* att,
                                          gda_stream_t stream_) {
    int st = check(n_views, x, ld, n, h);
    if (st != GDA_OK) return st;
    if (n == 0) return GDA_OK;
    if (!w || !b || !out || !att) return GDA_E_NULL;
    if (ldo < h) return GDA_E_SIZE;
    if (ldo % 4 || ((uintptr_t)out & 15) || ((uintptr_t)w & 15)) return GDA_E_UNSUPPORTED;
    Views V{};
    for (int k = 0; k < n_views; ++k) { V.x[k] = x[k]; V.ld[k] = ld[k]; if (x[k] == out) return GDA_E_ALIAS; }
    const unsigned blocks = (unsigned)gda_cdiv(n, TB / G);
    hipStream_t s = (hipStream_t)stream_;
    if (n_views == 2) k_attn_fwd<2><<<blocks, TB, 0, s>>>(V, w, b, n, (int)h, out, ldo, att);
    else if (n_views == 3) k_attn_fwd<3><<<blocks, TB, 0, s>>>(V, w, b, n, (int)h, out, ldo, att);
    else k_attn_fwd<4><<<blocks, TB, 0, s>>>(V, w, b, n, (int)h, out, ldo, att);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_attention_fuse_bwd_f32(int n_views, const float* const* x, const int64_t* ld, int64_t n, int64_t h,
                                          const float* w, const float* att, const float* gout, int64_t ldg,
                                          float* const* gx /* entries may be NULL: no gradient for that view */,
                                          float* gw, float* gb, void* workspace, size_t workspace_bytes,
                                          gda_stream_t stream_) {
    int st = check(n_views, x, ld, n, h);
    if (st != GDA_OK) return st;
    if (!w || !att || !gout || !gx || !gw || !gb) return GDA_E_NULL;
    if (ldg < h) return GDA_E_SIZE;
    if (ldg % 4 || ((uintptr_t)gout & 15) || ((uintptr_t)w & 15)) return GDA_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream_;
    if (n == 0) {
        GDA_HIP_TRY(hipMemsetAsync(gw, 0, sizeof(float) * h, s));
        GDA_HIP_TRY(hipMemsetAsync(gb, 0, sizeof(float), s));
        return GDA_OK;
    }
    if (!workspace || workspace_bytes < gda_attention_workspace_bytes(n, h)) return GDA_E_WORKSPACE;
    Views V{};
    for (int k = 0; k < n_views; ++k) {
        V.x[k] = x[k]; V.ld[k] = ld[k]; V.gx[k] = gx[k];
        if (gx[k] && ((uintptr_t)gx[k] & 15)) return GDA_E_UNSUPPORTED;
    }
    const unsigned blocks = (unsigned)gda_cdiv(n, TB / G);
    const size_t lds = sizeof(float) * (TB / G) * (size_t)(h + 1);
    float* part = (float*)workspace;
    if (n_views == 2) k_attn_bwd<2><<<blocks, TB, lds, s>>>(V, w, att, gout, ldg, n, (int)h, part);
    else if (n_views == 3) k_attn_bwd<3><<<blocks, TB, lds, s>>>(V, w, att, gout, ldg, n, (int)h, part);
    else k_attn_bwd<4><<<blocks, TB, lds, s>>>(V, w, att, gout, ldg, n, (int)h, part);
    GDA_LAUNCH_CHECK();
    k_attn_fold<<<(unsigned)gda_cdiv(h + 1, 64), FOLD_TB, 0, s>>>(part, (int)blocks, (int)h, gw, gb);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}
