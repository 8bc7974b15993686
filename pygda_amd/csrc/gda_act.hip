// Fused ReLU + inverted dropout for the activation that follows every conv layer
// (pygda/nn/a2gnn_base.py:135-138: x = act(x); x = F.dropout(x, p, training)), gfx950.
//
//   forward : y = (x > 0 && keep) ? x / (1 - p) : 0        one pass, 16-byte accesses
//   backward: gx = (y > 0) ? gy / (1 - p) : 0              (y > 0 <=> x > 0 and kept: no mask stored,
//                                                            no random numbers needed again)
// keep-bits come from a counter-based generator (Philox-4x32-10 keyed on the caller's seed, counter =
// (step, call site, element/4)): reproducible, and safe under hipGraph replay because `step` is
// read from device memory (the trainer bumps it once per step inside the captured graph) while
// the call-site id is a launch constant.
#include "gda_common.h"
#include "gda_philox.h"

namespace {

constexpr int TB = 256;

using Philox = GdaPhilox;

__global__ void __launch_bounds__(TB)
k_relu_dropout_fwd(const float* __restrict__ x, float* __restrict__ y, int64_t n, float p, float scale,
                   uint64_t seed, const int64_t* __restrict__ step, uint32_t site) {
    const uint64_t st = (uint64_t)step[0];
    const uint32_t thresh = (uint32_t)((double)p * 4294967296.0 > 4294967295.0 ? 4294967295.0 : (double)p * 4294967296.0);
    const int64_t quads = (n + 3) / 4;
    for (int64_t q = (int64_t)blockIdx.x * TB + threadIdx.x; q < quads; q += (int64_t)gridDim.x * TB) {
        uint32_t r[4];
        Philox::gen(seed, (st << 20) ^ site, (uint64_t)q, r);
        const int64_t i = q * 4;
        if (i + 3 < n) {
            const float4 v = *reinterpret_cast<const float4*>(x + i);
            float4 o;
            o.x = (v.x > 0.f && r[0] >= thresh) ? v.x * scale : 0.f;
            o.y = (v.y > 0.f && r[1] >= thresh) ? v.y * scale : 0.f;
            o.z = (v.z > 0.f && r[2] >= thresh) ? v.z * scale : 0.f;
            o.w = (v.w > 0.f && r[3] >= thresh) ? v.w * scale : 0.f;
            *reinterpret_cast<float4*>(y + i) = o;
        } else {
            for (int e = 0; e < 4 && i + e < n; ++e)
                y[i + e] = (x[i + e] > 0.f && r[e] >= thresh) ? x[i + e] * scale : 0.f;
        }
    }
}

__global__ void __launch_bounds__(TB)
k_relu_dropout_bwd(const float* __restrict__ gy, const float* __restrict__ y, float* __restrict__ gx,
                   int64_t n, float scale) {
    const int64_t quads = (n + 3) / 4;
    for (int64_t q = (int64_t)blockIdx.x * TB + threadIdx.x; q < quads; q += (int64_t)gridDim.x * TB) {
        const int64_t i = q * 4;
        if (i + 3 < n) {
            const float4 g = *reinterpret_cast<const float4*>(gy + i);
            const float4 v = *reinterpret_cast<const float4*>(y + i);
            float4 o;
            o.x = v.x > 0.f ? g.x * scale : 0.f;
            o.y = v.y > 0.f ? g.y * scale : 0.f;
            o.z = v.z > 0.f ? g.z * scale : 0.f;
            o.w = v.w > 0.f ? g.w * scale : 0.f;
            *reinterpret_cast<float4*>(gx + i) = o;
        } else {
            for (int e = 0; e < 4 && i + e < n; ++e) gx[i + e] = y[i + e] > 0.f ? gy[i + e] * scale : 0.f;
        }
    }
}

// The same activation reading a COLUMN-MAJOR input xT [d, ldT] (what the LDS-resident K-step kernel of
// gda_kstep.hip leaves) and writing the row-major y [n, d]: the transposition rides through a 64 x 64 LDS
// tile, so no separate transpose pass exists.  Keep-bits are keyed on the row-major element index exactly
// as in k_relu_dropout_fwd: same masks, same values as transpose + k_relu_dropout_fwd.
constexpr int TT = 64;

__global__ void __launch_bounds__(TB)
k_relu_dropout_fwd_T(const float* __restrict__ xT, int64_t ldT, float* __restrict__ y, int n, int d, float p,
                     float scale, uint64_t seed, const int64_t* __restrict__ step, uint32_t site) {
    __shared__ float tile[TT][TT + 1];                       // tile[c][i]
    const int i0 = blockIdx.x * TT, c0 = blockIdx.y * TT;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
    for (int cc = ty; cc < TT; cc += 4)
        if (c0 + cc < d && i0 + tx < n) tile[cc][tx] = xT[(int64_t)(c0 + cc) * ldT + i0 + tx];
    __syncthreads();
    const uint64_t st = (uint64_t)step[0];
    const uint32_t thresh = (uint32_t)((double)p * 4294967296.0 > 4294967295.0 ? 4294967295.0 : (double)p * 4294967296.0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = threadIdx.x + TB * k;
        const int r = idx >> 4, c4 = (idx & 15) * 4;
        const int i = i0 + r, c = c0 + c4;
        if (i < n && c < d) {                                  // d % 4 == 0: whole quads
            uint32_t rn[4];
            Philox::gen(seed, (st << 20) ^ site, (uint64_t)(((int64_t)i * d + c) >> 2), rn);
            float4 o;
            const float v0 = tile[c4 + 0][r], v1 = tile[c4 + 1][r], v2 = tile[c4 + 2][r], v3 = tile[c4 + 3][r];
            o.x = (v0 > 0.f && rn[0] >= thresh) ? v0 * scale : 0.f;
            o.y = (v1 > 0.f && rn[1] >= thresh) ? v1 * scale : 0.f;
            o.z = (v2 > 0.f && rn[2] >= thresh) ? v2 * scale : 0.f;
            o.w = (v3 > 0.f && rn[3] >= thresh) ? v3 * scale : 0.f;
            *reinterpret_cast<float4*>(y + (int64_t)i * d + c) = o;
        }
    }
}

// backward: gy, y row-major [n, d] -> gxT column-major [d, ldT] (the layout the backward K-step kernel reads)
__global__ void __launch_bounds__(TB)
k_relu_dropout_bwd_T(const float* __restrict__ gy, const float* __restrict__ y, float* __restrict__ gxT,
                     int64_t ldT, int n, int d, float scale) {
    __shared__ float tile[TT][TT + 1];                       // tile[c][i]
    const int i0 = blockIdx.x * TT, c0 = blockIdx.y * TT;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = threadIdx.x + TB * k;
        const int r = idx >> 4, c4 = (idx & 15) * 4;
        const int i = i0 + r, c = c0 + c4;
        if (i < n && c < d) {
            const float4 g = *reinterpret_cast<const float4*>(gy + (int64_t)i * d + c);
            const float4 v = *reinterpret_cast<const float4*>(y + (int64_t)i * d + c);
            tile[c4 + 0][r] = v.x > 0.f ? g.x * scale : 0.f;
            tile[c4 + 1][r] = v.y > 0.f ? g.y * scale : 0.f;
            tile[c4 + 2][r] = v.z > 0.f ? g.z * scale : 0.f;
            tile[c4 + 3][r] = v.w > 0.f ? g.w * scale : 0.f;
        }
    }
    __syncthreads();
#pragma unroll 4
    for (int cc = ty; cc < TT; cc += 4)
        if (c0 + cc < d && i0 + tx < n) gxT[(int64_t)(c0 + cc) * ldT + i0 + tx] = tile[cc][tx];
}

unsigned grid_for(int64_t n) {
    int64_t g = gda_cdiv((n + 3) / 4, TB);
    if (g > 256 * 8) g = 256 * 8;
    return (unsigned)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" int gda_relu_dropout_fwd_f32(const float* x, float* y, int64_t n, float p, uint64_t seed,
                                        const int64_t* step, uint32_t site, gda_stream_t stream) {
    if (n < 0 || !(p >= 0.f && p < 1.f)) return GDA_E_SIZE;
    if (n == 0) return GDA_OK;
    if (!x || !y || !step) return GDA_E_NULL;
    if (((uintptr_t)x | (uintptr_t)y) % 16 != 0) return GDA_E_UNSUPPORTED;
    k_relu_dropout_fwd<<<grid_for(n), TB, 0, (hipStream_t)stream>>>(x, y, n, p, 1.f / (1.f - p), seed, step, site);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_relu_dropout_bwd_f32(const float* gy, const float* y, float* gx, int64_t n, float p,
                                        gda_stream_t stream) {
    if (n < 0 || !(p >= 0.f && p < 1.f)) return GDA_E_SIZE;
    if (n == 0) return GDA_OK;
    if (!gy || !y || !gx) return GDA_E_NULL;
    if (((uintptr_t)gy | (uintptr_t)y | (uintptr_t)gx) % 16 != 0) return GDA_E_UNSUPPORTED;
    k_relu_dropout_bwd<<<grid_for(n), TB, 0, (hipStream_t)stream>>>(gy, y, gx, n, 1.f / (1.f - p));
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_relu_dropout_fwd_cm_f32(const float* xT, int64_t ldT, float* y, int64_t n, int64_t d, float p,
                                          uint64_t seed, const int64_t* step, uint32_t site, gda_stream_t stream) {
    if (n < 0 || d < 0 || n >= INT32_MAX || d >= INT32_MAX || ldT < n || !(p >= 0.f && p < 1.f)) return GDA_E_SIZE;
    if (n == 0 || d == 0) return GDA_OK;
    if (!xT || !y || !step) return GDA_E_NULL;
    if (d % 4 != 0 || (uintptr_t)y % 16 != 0) return GDA_E_UNSUPPORTED;
    const dim3 grid((unsigned)gda_cdiv(n, TT), (unsigned)gda_cdiv(d, TT));
    if (grid.y > 65535) return GDA_E_SIZE;
    k_relu_dropout_fwd_T<<<grid, TB, 0, (hipStream_t)stream>>>(xT, ldT, y, (int)n, (int)d, p, 1.f / (1.f - p), seed,
                                                               step, site);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_relu_dropout_bwd_cm_f32(const float* gy, const float* y, float* gxT, int64_t ldT, int64_t n,
                                          int64_t d, float p, gda_stream_t stream) {
    if (n < 0 || d < 0 || n >= INT32_MAX || d >= INT32_MAX || ldT < n || !(p >= 0.f && p < 1.f)) return GDA_E_SIZE;
    if (n == 0 || d == 0) return GDA_OK;
    if (!gy || !y || !gxT) return GDA_E_NULL;
    if (d % 4 != 0 || ((uintptr_t)gy | (uintptr_t)y) % 16 != 0) return GDA_E_UNSUPPORTED;
    const dim3 grid((unsigned)gda_cdiv(n, TT), (unsigned)gda_cdiv(d, TT));
    if (grid.y > 65535) return GDA_E_SIZE;
    k_relu_dropout_bwd_T<<<grid, TB, 0, (hipStream_t)stream>>>(gy, y, gxT, ldT, (int)n, (int)d, 1.f / (1.f - p));
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}
