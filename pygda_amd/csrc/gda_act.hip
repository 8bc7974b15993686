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
                   uint64_t seed, const int64_t* __restrict__ step, uint32_t site, int64_t period) {
    const uint64_t st = (uint64_t)step[0];
    const uint32_t thresh = (uint32_t)((double)p * 4294967296.0 > 4294967295.0 ? 4294967295.0 : (double)p * 4294967296.0);
    const int64_t quads = (n + 3) / 4;
    for (int64_t q = (int64_t)blockIdx.x * TB + threadIdx.x; q < quads; q += (int64_t)gridDim.x * TB) {
        uint32_t r[4];
        Philox::gen(seed, (st << 20) ^ site, (uint64_t)q, r);
        const int64_t i = q * 4;
        if (period) {              // y = the activation of `n / period` stacked copies of x: element i reads x[i % period] (period % 4 == 0)
            const float4 v = *reinterpret_cast<const float4*>(x + i % period);
            float4 o;
            o.x = (v.x > 0.f && r[0] >= thresh) ? v.x * scale : 0.f;
            o.y = (v.y > 0.f && r[1] >= thresh) ? v.y * scale : 0.f;
            o.z = (v.z > 0.f && r[2] >= thresh) ? v.z * scale : 0.f;
            o.w = (v.w > 0.f && r[3] >= thresh) ? v.w * scale : 0.f;
            *reinterpret_cast<float4*>(y + i) = o;
        } else if (i + 3 < n) {
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

// Two independent dropout draws of ONE activation: y[i] = drop_a(relu(x[i])), y[n + i] = drop_b(relu(x[i])) --
// the stacked pair of rows on which A2GNN's two source passes (features :192 and logits :181) continue as one
// pass of 2n rows (pygda/models/a2gnn.py); backward sums the two halves (the gradient of the shared layer-0 output).
__global__ void __launch_bounds__(TB)
k_relu_dropout_pair_fwd(const float* __restrict__ x, float* __restrict__ y, int64_t nd, float p, float scale,
                        uint64_t seed, const int64_t* __restrict__ step, uint32_t site_a, uint32_t site_b) {
    const bool drop = p > 0.f;
    const uint64_t st = drop ? (uint64_t)step[0] : 0;
    const uint32_t thresh = (uint32_t)((double)p * 4294967296.0 > 4294967295.0 ? 4294967295.0 : (double)p * 4294967296.0);
    const int64_t quads = nd / 4;
    for (int64_t q = (int64_t)blockIdx.x * TB + threadIdx.x; q < quads; q += (int64_t)gridDim.x * TB) {
        uint32_t ra[4] = {~0u, ~0u, ~0u, ~0u}, rb[4] = {~0u, ~0u, ~0u, ~0u};
        if (drop) {
            Philox::gen(seed, (st << 20) ^ site_a, (uint64_t)q, ra);
            Philox::gen(seed, (st << 20) ^ site_b, (uint64_t)q, rb);
        }
        const float4 v = *reinterpret_cast<const float4*>(x + q * 4);
        const float vv[4] = {v.x, v.y, v.z, v.w};
        float a[4], b[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float r = vv[e] > 0.f ? vv[e] * scale : 0.f;
            a[e] = (!drop || ra[e] >= thresh) ? r : 0.f;
            b[e] = (!drop || rb[e] >= thresh) ? r : 0.f;
        }
        *reinterpret_cast<float4*>(y + q * 4) = make_float4(a[0], a[1], a[2], a[3]);
        *reinterpret_cast<float4*>(y + nd + q * 4) = make_float4(b[0], b[1], b[2], b[3]);
    }
}

__global__ void __launch_bounds__(TB)
k_relu_dropout_pair_bwd(const float* __restrict__ gy, const float* __restrict__ y, float* __restrict__ gx,
                        int64_t nd, float scale) {
    const int64_t quads = nd / 4;
    for (int64_t q = (int64_t)blockIdx.x * TB + threadIdx.x; q < quads; q += (int64_t)gridDim.x * TB) {
        const float4 ga = *reinterpret_cast<const float4*>(gy + q * 4), gb = *reinterpret_cast<const float4*>(gy + nd + q * 4);
        const float4 ya = *reinterpret_cast<const float4*>(y + q * 4), yb = *reinterpret_cast<const float4*>(y + nd + q * 4);
        float4 o;
        o.x = ((ya.x > 0.f ? ga.x : 0.f) + (yb.x > 0.f ? gb.x : 0.f)) * scale;
        o.y = ((ya.y > 0.f ? ga.y : 0.f) + (yb.y > 0.f ? gb.y : 0.f)) * scale;
        o.z = ((ya.z > 0.f ? ga.z : 0.f) + (yb.z > 0.f ? gb.z : 0.f)) * scale;
        o.w = ((ya.w > 0.f ? ga.w : 0.f) + (yb.w > 0.f ? gb.w : 0.f)) * scale;
        *reinterpret_cast<float4*>(gx + q * 4) = o;
    }
}

// The same pass with the column sums of gx as a by-product (the bias gradient of the layer below): a thread owns
// one column quad and every (TB / quads-per-row)-th row of its block's share, block partials in a fixed order.
constexpr int PAIR_BLOCKS = 256;

__global__ void __launch_bounds__(TB)
k_relu_dropout_pair_bwd_colsum(const float* __restrict__ gy, const float* __restrict__ y, float* __restrict__ gx,
                               int64_t n, int d, float scale, float* __restrict__ partial) {
    __shared__ float4 red[TB];
    const int q = d >> 2, rpb = TB / q;
    const int tq = threadIdx.x % q, tr = threadIdx.x / q;
    const int c = tq * 4;
    const int64_t nd = n * d;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tr < rpb) {
        for (int64_t i = (int64_t)blockIdx.x * rpb + tr; i < n; i += (int64_t)gridDim.x * rpb) {
            const int64_t o = i * d + c;
            const float4 ga = *reinterpret_cast<const float4*>(gy + o), gb = *reinterpret_cast<const float4*>(gy + nd + o);
            const float4 ya = *reinterpret_cast<const float4*>(y + o), yb = *reinterpret_cast<const float4*>(y + nd + o);
            float4 v;
            v.x = ((ya.x > 0.f ? ga.x : 0.f) + (yb.x > 0.f ? gb.x : 0.f)) * scale;
            v.y = ((ya.y > 0.f ? ga.y : 0.f) + (yb.y > 0.f ? gb.y : 0.f)) * scale;
            v.z = ((ya.z > 0.f ? ga.z : 0.f) + (yb.z > 0.f ? gb.z : 0.f)) * scale;
            v.w = ((ya.w > 0.f ? ga.w : 0.f) + (yb.w > 0.f ? gb.w : 0.f)) * scale;
            *reinterpret_cast<float4*>(gx + o) = v;
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (tr == 0 && tq < q) {
        float4 s = red[tq];
        for (int r = 1; r < rpb; ++r) {
            const float4 o = red[r * q + tq];
            s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
        }
        *reinterpret_cast<float4*>(partial + (int64_t)blockIdx.x * d + c) = s;
    }
}

// out = [a ; b] (either half NULL -> zeros): the gradient of a stacked pair whose halves went to different consumers
__global__ void __launch_bounds__(TB)
k_stack2(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t half) {
    const int64_t quads = half / 4;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t q = (int64_t)blockIdx.x * TB + threadIdx.x; q < 2 * quads; q += (int64_t)gridDim.x * TB) {
        const float* src = q < quads ? a : b;
        const int64_t k = q < quads ? q : q - quads;
        *reinterpret_cast<float4*>(out + q * 4) = src ? *reinterpret_cast<const float4*>(src + k * 4) : z;
    }
}

// gx = [a ; b] masked by the activation's own output: gx[i] = y[i] > 0 ? ([a ; b])[i] * scale : 0 -- k_stack2 and
// k_relu_dropout_bwd in one pass (the stacked gradient is never written unmasked)
__global__ void __launch_bounds__(TB)
k_relu_dropout_bwd2(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ y,
                    float* __restrict__ gx, int64_t half, float scale) {
    const int64_t quads = half / 4;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t q = (int64_t)blockIdx.x * TB + threadIdx.x; q < 2 * quads; q += (int64_t)gridDim.x * TB) {
        const float* src = q < quads ? a : b;
        const int64_t k = q < quads ? q : q - quads;
        const float4 g = src ? *reinterpret_cast<const float4*>(src + k * 4) : z;
        const float4 v = *reinterpret_cast<const float4*>(y + q * 4);
        float4 o;
        o.x = v.x > 0.f ? g.x * scale : 0.f;
        o.y = v.y > 0.f ? g.y * scale : 0.f;
        o.z = v.z > 0.f ? g.z * scale : 0.f;
        o.w = v.w > 0.f ? g.w * scale : 0.f;
        *reinterpret_cast<float4*>(gx + q * 4) = o;
    }
}

// Column sums of a row-major [n, d] matrix (bias gradients), deterministic two-stage: fixed rows per block lane,
// fixed-order block partials, fixed-order final sum.  d <= 256: TB / d rows in flight per block; wider: one row
// at a time, a thread owning columns tid, tid + TB, ...
constexpr int COLSUM_BLOCKS = 256;
constexpr int COLSUM_MAXD = 4 * TB;

// four rows in flight per lane (independent accumulators, combined in a fixed order): at a few rows per lane the
// kernel is a chain of dependent-latency loads otherwise
__global__ void __launch_bounds__(TB)
k_colsum_partial(const float* __restrict__ x, int64_t ldx, int64_t n, int d, float* __restrict__ partial) {
    __shared__ float red[TB];
    if (d <= TB) {
        const int rpb = TB / d, tc = threadIdx.x % d, tr = threadIdx.x / d;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (tr < rpb) {
            const int64_t stride = (int64_t)gridDim.x * rpb;
            int64_t i = (int64_t)blockIdx.x * rpb + tr;
            for (; i + 7 * stride < n; i += 8 * stride) {            // eight loads in flight
                const float v0 = x[i * ldx + tc], v1 = x[(i + stride) * ldx + tc];
                const float v2 = x[(i + 2 * stride) * ldx + tc], v3 = x[(i + 3 * stride) * ldx + tc];
                const float v4 = x[(i + 4 * stride) * ldx + tc], v5 = x[(i + 5 * stride) * ldx + tc];
                const float v6 = x[(i + 6 * stride) * ldx + tc], v7 = x[(i + 7 * stride) * ldx + tc];
                a0 += v0; a1 += v1; a2 += v2; a3 += v3;
                a0 += v4; a1 += v5; a2 += v6; a3 += v7;
            }
            for (; i < n; i += stride) a0 += x[i * ldx + tc];
        }
        red[threadIdx.x] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (tr == 0) {
            float s = red[tc];
            for (int r = 1; r < rpb; ++r) s += red[r * d + tc];
            partial[(int64_t)blockIdx.x * d + tc] = s;
        }
    } else {
        for (int c = threadIdx.x; c < d; c += TB) {
            float acc = 0.f;
            for (int64_t i = blockIdx.x; i < n; i += gridDim.x) acc += x[i * ldx + c];
            partial[(int64_t)blockIdx.x * d + c] = acc;
        }
    }
}

// out[c] = sum_b partial[b][c]: 8 columns per block, 32 lanes per column over the partials (lane g adds the
// partials g, g + 32, ... in that order, four loads in flight), then a fixed binary tree over the 32 lane sums.
constexpr int CF_COLS = 8, CF_LANES = TB / CF_COLS;

__global__ void __launch_bounds__(TB)
k_colsum_final(const float* __restrict__ partial, int blocks, int d, float* __restrict__ out) {
    __shared__ float red[CF_LANES][CF_COLS];
    const int cl = threadIdx.x % CF_COLS, g = threadIdx.x / CF_COLS;
    const int c = blockIdx.x * CF_COLS + cl;
    float s = 0.f;
    if (c < d) {
        int b = g;
        for (; b + 3 * CF_LANES < blocks; b += 4 * CF_LANES) {
            const float v0 = partial[(int64_t)b * d + c], v1 = partial[(int64_t)(b + CF_LANES) * d + c];
            const float v2 = partial[(int64_t)(b + 2 * CF_LANES) * d + c], v3 = partial[(int64_t)(b + 3 * CF_LANES) * d + c];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; b < blocks; b += CF_LANES) s += partial[(int64_t)b * d + c];
    }
    red[g][cl] = s;
    __syncthreads();
#pragma unroll
    for (int w = CF_LANES / 2; w > 0; w >>= 1) {
        if (g < w) red[g][cl] += red[g + w][cl];
        __syncthreads();
    }
    if (g == 0 && c < d) out[c] = red[0][cl];
}

// Small problems (the classifier's bias gradient: 9,360 rows x 5 classes) in ONE launch of one workgroup: a lane per
// (row slot, column), eight rows in flight per lane, then the row slots of a column folded as a fixed binary tree in
// LDS -- the two-launch form costs a second dependent launch (~5 us in a replayed step) for 187 KB of input.
constexpr int CO_TB = 1024;
constexpr int64_t CO_MAX_ELEMS = 1 << 17;
constexpr int CO_MAX_D = 64;

__global__ void __launch_bounds__(CO_TB)
k_colsum_one(const float* __restrict__ x, int64_t ldx, int64_t n, int d, int slots, float* __restrict__ out) {
    __shared__ float red[CO_TB];
    const int tc = threadIdx.x % d, tr = threadIdx.x / d;          // slots = largest power of two <= CO_TB / d
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (tr < slots) {
        int64_t i = tr;
        const int64_t stride = slots;
        for (; i + 7 * stride < n; i += 8 * stride) {
            const float v0 = x[i * ldx + tc], v1 = x[(i + stride) * ldx + tc];
            const float v2 = x[(i + 2 * stride) * ldx + tc], v3 = x[(i + 3 * stride) * ldx + tc];
            const float v4 = x[(i + 4 * stride) * ldx + tc], v5 = x[(i + 5 * stride) * ldx + tc];
            const float v6 = x[(i + 6 * stride) * ldx + tc], v7 = x[(i + 7 * stride) * ldx + tc];
            a0 += v0; a1 += v1; a2 += v2; a3 += v3;
            a0 += v4; a1 += v5; a2 += v6; a3 += v7;
        }
        for (; i < n; i += stride) a0 += x[i * ldx + tc];
        red[tr * d + tc] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    for (int w = slots >> 1; w > 0; w >>= 1) {
        if (tr < w) red[tr * d + tc] += red[(tr + w) * d + tc];
        __syncthreads();
    }
    if (tr == 0) out[tc] = red[tc];
}

int colsum_blocks(int64_t n, int d) {
    const int rpb = d <= TB ? TB / d : 1;
    const int64_t g = gda_cdiv(n, (int64_t)rpb * 8);                 // >= 8 rows per lane
    return (int)(g > COLSUM_BLOCKS ? COLSUM_BLOCKS : (g < 1 ? 1 : g));
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
    k_relu_dropout_fwd<<<grid_for(n), TB, 0, (hipStream_t)stream>>>(x, y, n, p, 1.f / (1.f - p), seed, step, site, 0);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_relu_dropout_tiled_fwd_f32(const float* x, int64_t period, int64_t copies, float* y, float p, uint64_t seed,
                                              const int64_t* step, uint32_t site, gda_stream_t stream) {
    if (period < 0 || copies < 0 || !(p >= 0.f && p < 1.f) || period % 4 != 0) return GDA_E_SIZE;
    if (period == 0 || copies == 0) return GDA_OK;
    if (!x || !y || !step) return GDA_E_NULL;
    if (((uintptr_t)x | (uintptr_t)y) % 16 != 0) return GDA_E_UNSUPPORTED;
    const int64_t n = period * copies;
    k_relu_dropout_fwd<<<grid_for(n), TB, 0, (hipStream_t)stream>>>(x, y, n, p, 1.f / (1.f - p), seed, step, site, period);
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
    GDA_UNLESS_SKIPPED("k_relu_dropout_fwd_T") k_relu_dropout_fwd_T<<<grid, TB, 0, (hipStream_t)stream>>>(xT, ldT, y, (int)n, (int)d, p, 1.f / (1.f - p), seed,
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
    GDA_UNLESS_SKIPPED("k_relu_dropout_bwd_T") k_relu_dropout_bwd_T<<<grid, TB, 0, (hipStream_t)stream>>>(gy, y, gxT, ldT, (int)n, (int)d, 1.f / (1.f - p));
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_relu_dropout_pair_fwd_f32(const float* x, float* y, int64_t n, int64_t d, float p, uint64_t seed,
                                             const int64_t* step, uint32_t site_a, uint32_t site_b,
                                             gda_stream_t stream) {
    if (n < 0 || d < 0 || !(p >= 0.f && p < 1.f)) return GDA_E_SIZE;
    if (n == 0 || d == 0) return GDA_OK;
    if (!x || !y || (p > 0.f && !step)) return GDA_E_NULL;
    if (d % 4 != 0 || ((uintptr_t)x | (uintptr_t)y) % 16 != 0) return GDA_E_UNSUPPORTED;
    k_relu_dropout_pair_fwd<<<grid_for(n * d), TB, 0, (hipStream_t)stream>>>(x, y, n * d, p, 1.f / (1.f - p), seed, step,
                                                                           site_a, site_b);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" size_t gda_relu_dropout_pair_workspace_bytes(int64_t d) {
    return d > 0 ? (size_t)PAIR_BLOCKS * (size_t)d * sizeof(float) : 0;
}

extern "C" int gda_relu_dropout_pair_bwd_f32(const float* gy, const float* y, float* gx, int64_t n, int64_t d, float p,
                                             float* colsum, void* workspace, size_t workspace_bytes,
                                             gda_stream_t stream) {
    if (n < 0 || d < 0 || !(p >= 0.f && p < 1.f)) return GDA_E_SIZE;
    if (n == 0 || d == 0) return GDA_OK;
    if (!gy || !y || !gx) return GDA_E_NULL;
    if (d % 4 != 0 || ((uintptr_t)gy | (uintptr_t)y | (uintptr_t)gx) % 16 != 0) return GDA_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (!colsum) {
        k_relu_dropout_pair_bwd<<<grid_for(n * d), TB, 0, s>>>(gy, y, gx, n * d, 1.f / (1.f - p));
        GDA_LAUNCH_CHECK();
        return GDA_OK;
    }
    if (d > 4 * TB) return GDA_E_UNSUPPORTED;
    if (!workspace) return GDA_E_NULL;
    if (workspace_bytes < gda_relu_dropout_pair_workspace_bytes(d) || (uintptr_t)workspace % 16 != 0) return GDA_E_WORKSPACE;
    const int rpb = TB / (int)(d >> 2);
    const int64_t want = gda_cdiv(n, rpb);
    const int blocks = (int)(want > PAIR_BLOCKS ? PAIR_BLOCKS : want);
    float* partial = static_cast<float*>(workspace);
    k_relu_dropout_pair_bwd_colsum<<<blocks, TB, 0, s>>>(gy, y, gx, n, (int)d, 1.f / (1.f - p), partial);
    GDA_LAUNCH_CHECK();
    GDA_UNLESS_SKIPPED("k_colsum_final") k_colsum_final<<<(unsigned)gda_cdiv(d, CF_COLS), TB, 0, s>>>(partial, blocks, (int)d, colsum);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_stack2_f32(const float* a, const float* b, float* out, int64_t half_elems, gda_stream_t stream) {
    if (half_elems < 0) return GDA_E_SIZE;
    if (half_elems == 0) return GDA_OK;
    if (!out) return GDA_E_NULL;
    if (half_elems % 4 != 0 || ((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) % 16 != 0) return GDA_E_UNSUPPORTED;
    GDA_UNLESS_SKIPPED("k_stack2") k_stack2<<<grid_for(2 * half_elems), TB, 0, (hipStream_t)stream>>>(a, b, out, half_elems);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_relu_dropout_bwd2_f32(const float* ga, const float* gb, const float* y, float* gx, int64_t half_elems,
                                         float p, gda_stream_t stream) {
    if (half_elems < 0 || !(p >= 0.f && p < 1.f)) return GDA_E_SIZE;
    if (half_elems == 0) return GDA_OK;
    if (!y || !gx) return GDA_E_NULL;
    if (half_elems % 4 != 0 || ((uintptr_t)ga | (uintptr_t)gb | (uintptr_t)y | (uintptr_t)gx) % 16 != 0) return GDA_E_UNSUPPORTED;
    k_relu_dropout_bwd2<<<grid_for(2 * half_elems), TB, 0, (hipStream_t)stream>>>(ga, gb, y, gx, half_elems, 1.f / (1.f - p));
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" size_t gda_colsum_workspace_bytes(int64_t n, int64_t d) {
    if (n <= 0 || d <= 0 || d > COLSUM_MAXD) return 0;
    return (size_t)colsum_blocks(n, (int)d) * (size_t)d * sizeof(float);
}

extern "C" int gda_colsum_f32(const float* x, int64_t ldx, int64_t n, int64_t d, float* out, void* workspace,
                              size_t workspace_bytes, gda_stream_t stream) {
    if (n < 0 || d < 0 || ldx < d) return GDA_E_SIZE;
    if (d == 0) return GDA_OK;
    if (d > COLSUM_MAXD) return GDA_E_UNSUPPORTED;
    if (!out) return GDA_E_NULL;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        GDA_HIP_TRY(hipMemsetAsync(out, 0, (size_t)d * sizeof(float), s));
        return GDA_OK;
    }
    if (!x) return GDA_E_NULL;
    if (d <= CO_MAX_D && n * d <= CO_MAX_ELEMS) {                    // one workgroup, one launch
        int slots = 1;
        while (slots * 2 * (int)d <= CO_TB) slots *= 2;
        k_colsum_one<<<1, CO_TB, 0, s>>>(x, ldx, n, (int)d, slots, out);
        GDA_LAUNCH_CHECK();
        return GDA_OK;
    }
    if (!workspace) return GDA_E_NULL;
    if (workspace_bytes < gda_colsum_workspace_bytes(n, d)) return GDA_E_WORKSPACE;
    const int blocks = colsum_blocks(n, (int)d);
    k_colsum_partial<<<blocks, TB, 0, s>>>(x, ldx, n, (int)d, static_cast<float*>(workspace));
    GDA_LAUNCH_CHECK();
    GDA_UNLESS_SKIPPED("k_colsum_final") k_colsum_final<<<(unsigned)gda_cdiv(d, CF_COLS), TB, 0, s>>>(static_cast<const float*>(workspace), blocks, (int)d, out);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}
