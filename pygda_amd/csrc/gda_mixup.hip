// Layer epilogue of StruRW's mixup backbone (pygda/nn/mixup_base.py:146-196), gfx950.
//
// The reference evaluates three MixUpGCNConv calls per layer (mixup_gcnconv.py:194-238:
// out = Agg(lin(x)) + lin_cen(x_cen) + bias):
//     x'      = drop(relu(Agg (lin(x))       + lin_cen(x)     + b))     the plain layer
//     new     =      relu(Agg (lin(x))       + lin_cen(x_mix) + b)      mixed centre, same graph
//     new_b   =      relu(Aggb(lin(x[perm])) + lin_cen(x_mix) + b)      mixed centre, shuffled graph
//     x_mix'  = drop(lam * new + (1 - lam) * new_b)
// The shuffled graph is the same graph with its nodes renumbered (strurw.py:735-758), so
// Aggb(lin(x[perm])) = Agg(lin(x))[perm]: ONE aggregation P = Agg(lin(x)) serves all three, and what is left of
// the layer is this row-wise epilogue over P, P[perm], the centre projections and the bias.  One launch forward,
// one backward (+ a 1-block column-sum finish for the bias gradient) replace ~14 elementwise / gather kernels
// each way.  Activations travel between layers as the stacked pair XX = [x ; x_mix]  ([2n, h], row-major), so the
// centre projection of a layer is one GEMM over 2n rows and so is its weight gradient.
//
//   FIRST layer: the centre projection is linear, lin_cen(x_mix) = lam C + (1-lam) C[perm] with C = lin_cen(x):
//                x_mix of the (wide, possibly sparse) input features is never formed; CC = C is [n, h].
//   SEP        : the caller hands an explicit Pb (= a second aggregation over a foreign edge_index_b that is NOT
//                a renumbering of edge_index) instead of P[perm].
// Backward needs the three ReLU masks and the keep bit of the mixed row: one byte per element written by the
// forward pass (m2 | m3 << 1 | keep << 2); the plain row's mask is (x' > 0) as in gda_act.hip.
#include "gda_common.h"
#include "gda_philox.h"

namespace {

constexpr int TB = 256;
constexpr int MAX_BLOCKS = 512;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float relu(float v) { return v > 0.f ? v : 0.f; }

template <bool FIRST, bool SEP>
__global__ void __launch_bounds__(TB)
k_mixup_fwd(const float* __restrict__ P, const float* __restrict__ Pb, const float* __restrict__ CC,
            const float* __restrict__ bias, const int64_t* __restrict__ perm, int64_t n, int h, float lam, float p,
            float scale, uint64_t seed, const int64_t* __restrict__ step, uint32_t site_x, uint32_t site_m,
            float* __restrict__ XX, uint8_t* __restrict__ mask) {
    const int q = h >> 2, rpb = TB / q;
    const int tq = threadIdx.x % q, tr = threadIdx.x / q;
    if (tr >= rpb) return;
    const int c = tq * 4;
    const float4 b = bias ? ld4(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float mal = 1.f - lam;
    const bool drop = p > 0.f;
    const uint64_t st = drop ? (uint64_t)step[0] : 0;
    const uint32_t thresh = (uint32_t)((double)p * 4294967296.0 > 4294967295.0 ? 4294967295.0 : (double)p * 4294967296.0);
    for (int64_t i = (int64_t)blockIdx.x * rpb + tr; i < n; i += (int64_t)gridDim.x * rpb) {
        const int64_t j = perm[i];
        const float4 pi = ld4(P + i * h + c);
        const float4 pb = SEP ? ld4(Pb + i * h + c) : ld4(P + j * h + c);
        const float4 ci = ld4(CC + i * h + c);
        float4 cm;
        if (FIRST) {
            const float4 cj = ld4(CC + j * h + c);
            cm = make_float4(lam * ci.x + mal * cj.x, lam * ci.y + mal * cj.y, lam * ci.z + mal * cj.z,
                             lam * ci.w + mal * cj.w);
        } else {
            cm = ld4(CC + (n + i) * h + c);
        }
        uint32_t rx[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        uint32_t rm[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        if (drop) {
            const uint64_t quad = (uint64_t)((i * h + c) >> 2);
            GdaPhilox::gen(seed, (st << 20) ^ site_x, quad, rx);
            GdaPhilox::gen(seed, (st << 20) ^ site_m, quad, rm);
        }
        const float a[4] = {pi.x + ci.x + b.x, pi.y + ci.y + b.y, pi.z + ci.z + b.z, pi.w + ci.w + b.w};
        const float u[4] = {pi.x + cm.x + b.x, pi.y + cm.y + b.y, pi.z + cm.z + b.z, pi.w + cm.w + b.w};
        const float v[4] = {pb.x + cm.x + b.x, pb.y + cm.y + b.y, pb.z + cm.z + b.z, pb.w + cm.w + b.w};
        float xn[4], xm[4];
        uint32_t mk = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool kx = !drop || rx[e] >= thresh, km = !drop || rm[e] >= thresh;
            xn[e] = (a[e] > 0.f && kx) ? a[e] * scale : 0.f;
            const float mix = lam * relu(u[e]) + mal * relu(v[e]);
            xm[e] = km ? mix * scale : 0.f;
            mk |= (uint32_t)((u[e] > 0.f ? 1u : 0u) | (v[e] > 0.f ? 2u : 0u) | (km ? 4u : 0u)) << (8 * e);
        }
        st4(XX + i * h + c, make_float4(xn[0], xn[1], xn[2], xn[3]));
        st4(XX + (n + i) * h + c, make_float4(xm[0], xm[1], xm[2], xm[3]));
        *reinterpret_cast<uint32_t*>(mask + i * h + c) = mk;
    }
}

// t2 / t3 of one row: the gradient of the mixed output routed to `new` (lam, mask bit 0) and to `new_b`
// (1 - lam, mask bit 1), zero where the mixed element was dropped (bit 2)
__device__ __forceinline__ void mix_terms(const float4 g, uint32_t mk, float sl, float sm, float (&t2)[4], float (&t3)[4]) {
    const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t m = mk >> (8 * e);
        t2[e] = ((m & 5u) == 5u) ? gv[e] * sl : 0.f;
        t3[e] = ((m & 6u) == 6u) ? gv[e] * sm : 0.f;
    }
}

template <bool FIRST, bool SEP>
__global__ void __launch_bounds__(TB)
k_mixup_bwd(const float* __restrict__ gXX, const float* __restrict__ XX, const uint8_t* __restrict__ mask,
            const int64_t* __restrict__ inv, int64_t n, int h, float lam, float scale, float* __restrict__ gP,
            float* __restrict__ gPb, float* __restrict__ gCC, float* __restrict__ partial) {
    __shared__ float4 red[TB];
    const int q = h >> 2, rpb = TB / q;
    const int tq = threadIdx.x % q, tr = threadIdx.x / q;
    const bool active = tr < rpb;
    const int c = tq * 4;
    const float mal = 1.f - lam, sl = scale * lam, sm = scale * mal;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        for (int64_t i = (int64_t)blockIdx.x * rpb + tr; i < n; i += (int64_t)gridDim.x * rpb) {
            const float4 gxn = ld4(gXX + i * h + c), gxm = ld4(gXX + (n + i) * h + c), y = ld4(XX + i * h + c);
            const uint32_t mk = *reinterpret_cast<const uint32_t*>(mask + i * h + c);
            const float gc[4] = {y.x > 0.f ? gxn.x * scale : 0.f, y.y > 0.f ? gxn.y * scale : 0.f,
                                 y.z > 0.f ? gxn.z * scale : 0.f, y.w > 0.f ? gxn.w * scale : 0.f};
            float t2[4], t3[4], u2[4] = {0.f, 0.f, 0.f, 0.f}, u3[4] = {0.f, 0.f, 0.f, 0.f};
            mix_terms(gxm, mk, sl, sm, t2, t3);
            if (FIRST || !SEP) {                   // the terms of the row that reads THIS row through perm
                const int64_t j = inv[i];
                mix_terms(ld4(gXX + (n + j) * h + c), *reinterpret_cast<const uint32_t*>(mask + j * h + c), sl, sm,
                          u2, u3);
            }
            float gp[4], gb[4], g0[4], g1[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gcm = t2[e] + t3[e];
                acc[e] += gc[e] + gcm;
                gp[e] = gc[e] + t2[e] + (SEP ? 0.f : u3[e]);
                gb[e] = t3[e];
                if (FIRST) g0[e] = gc[e] + lam * gcm + mal * (u2[e] + u3[e]);
                else { g0[e] = gc[e]; g1[e] = gcm; }
            }
            st4(gP + i * h + c, make_float4(gp[0], gp[1], gp[2], gp[3]));
            if (SEP) st4(gPb + i * h + c, make_float4(gb[0], gb[1], gb[2], gb[3]));
            st4(gCC + i * h + c, make_float4(g0[0], g0[1], g0[2], g0[3]));
            if (!FIRST) st4(gCC + (n + i) * h + c, make_float4(g1[0], g1[1], g1[2], g1[3]));
        }
    }
    red[threadIdx.x] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();
    if (active && tr == 0) {                       // fixed order over the block's row lanes: deterministic
        float4 s = red[tq];
        for (int r = 1; r < rpb; ++r) {
            const float4 o = red[r * q + tq];
            s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
        }
        st4(partial + (int64_t)blockIdx.x * h + c, s);
    }
}

// gbias[c] = sum_b partial[b][c]: 8 columns per block, 32 lanes per column over the partials (lane g adds the
// partials g, g + 32, ... in that order, four loads in flight), then a fixed binary tree over the 32 lane sums.
constexpr int CF_COLS = 8, CF_LANES = TB / CF_COLS;

__global__ void __launch_bounds__(TB)
k_mixup_bias(const float* __restrict__ partial, int blocks, int h, float* __restrict__ gbias) {
    __shared__ float red[CF_LANES][CF_COLS];
    const int cl = threadIdx.x % CF_COLS, g = threadIdx.x / CF_COLS;
    const int c = blockIdx.x * CF_COLS + cl;
    float s = 0.f;
    if (c < h) {
        int b = g;
        for (; b + 3 * CF_LANES < blocks; b += 4 * CF_LANES) {
            const float v0 = partial[(int64_t)b * h + c], v1 = partial[(int64_t)(b + CF_LANES) * h + c];
            const float v2 = partial[(int64_t)(b + 2 * CF_LANES) * h + c], v3 = partial[(int64_t)(b + 3 * CF_LANES) * h + c];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; b < blocks; b += CF_LANES) s += partial[(int64_t)b * h + c];
    }
    red[g][cl] = s;
    __syncthreads();
#pragma unroll
    for (int w = CF_LANES / 2; w > 0; w >>= 1) {
        if (g < w) red[g][cl] += red[g + w][cl];
        __syncthreads();
    }
    if (g == 0 && c < h) gbias[c] = red[0][cl];
}

int blocks_for(int64_t n, int h) {
    const int rpb = TB / (h >> 2);
    const int64_t g = gda_cdiv(n, rpb);
    return (int)(g > MAX_BLOCKS ? MAX_BLOCKS : (g < 1 ? 1 : g));
}

int check_shape(int64_t n, int64_t h, float lam, float p) {
    if (n < 0 || h < 0 || !(p >= 0.f && p < 1.f) || !(lam >= 0.f && lam <= 1.f)) return GDA_E_SIZE;
    if (h % 4 != 0 || h > 4 * TB) return GDA_E_UNSUPPORTED;
    return GDA_OK;
}

}  // namespace

extern "C" size_t gda_mixup_combine_workspace_bytes(int64_t n, int64_t h) {
    if (n <= 0 || h <= 0 || h % 4 != 0 || h > 4 * TB) return 0;
    return (size_t)blocks_for(n, (int)h) * (size_t)h * sizeof(float);
}

extern "C" int gda_mixup_combine_fwd_f32(const float* P, const float* Pb, const float* CC, int first, const float* bias,
                                         const int64_t* perm, int64_t n, int64_t h, float lam, float p, uint64_t seed,
                                         const int64_t* step, uint32_t site_x, uint32_t site_m, float* XX,
                                         uint8_t* mask, gda_stream_t stream) {
    const int rc = check_shape(n, h, lam, p);
    if (rc != GDA_OK) return rc;
    if (n == 0 || h == 0) return GDA_OK;
    if (!P || !CC || !perm || !XX || !mask || (p > 0.f && !step)) return GDA_E_NULL;
    if (((uintptr_t)P | (uintptr_t)Pb | (uintptr_t)CC | (uintptr_t)bias | (uintptr_t)XX) % 16 != 0 ||
        (uintptr_t)mask % 4 != 0)
        return GDA_E_UNSUPPORTED;
    const dim3 grid((unsigned)blocks_for(n, (int)h));
    const float scale = 1.f / (1.f - p);
    hipStream_t s = (hipStream_t)stream;
#define GDA_MIXUP_FWD(F, S) \
    k_mixup_fwd<F, S><<<grid, TB, 0, s>>>(P, Pb, CC, bias, perm, n, (int)h, lam, p, scale, seed, step, site_x, site_m, XX, mask)
    if (first) { if (Pb) GDA_MIXUP_FWD(true, true); else GDA_MIXUP_FWD(true, false); }
    else       { if (Pb) GDA_MIXUP_FWD(false, true); else GDA_MIXUP_FWD(false, false); }
#undef GDA_MIXUP_FWD
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_mixup_combine_bwd_f32(const float* gXX, const float* XX, const uint8_t* mask, const int64_t* inv_perm,
                                         int first, int64_t n, int64_t h, float lam, float p, float* gP, float* gPb,
                                         float* gCC, float* gbias, void* workspace, size_t workspace_bytes,
                                         gda_stream_t stream) {
    const int rc = check_shape(n, h, lam, p);
    if (rc != GDA_OK) return rc;
    if (n == 0 || h == 0) return GDA_OK;
    if (!gXX || !XX || !mask || !inv_perm || !gP || !gCC || !gbias || !workspace) return GDA_E_NULL;
    if (workspace_bytes < gda_mixup_combine_workspace_bytes(n, h)) return GDA_E_WORKSPACE;
    if (((uintptr_t)gXX | (uintptr_t)XX | (uintptr_t)gP | (uintptr_t)gPb | (uintptr_t)gCC | (uintptr_t)workspace) % 16 != 0 ||
        (uintptr_t)mask % 4 != 0)
        return GDA_E_UNSUPPORTED;
    const int blocks = blocks_for(n, (int)h);
    const float scale = 1.f / (1.f - p);
    float* partial = static_cast<float*>(workspace);
    hipStream_t s = (hipStream_t)stream;
#define GDA_MIXUP_BWD(F, S) \
    k_mixup_bwd<F, S><<<blocks, TB, 0, s>>>(gXX, XX, mask, inv_perm, n, (int)h, lam, scale, gP, gPb, gCC, partial)
    if (first) { if (gPb) GDA_MIXUP_BWD(true, true); else GDA_MIXUP_BWD(true, false); }
    else       { if (gPb) GDA_MIXUP_BWD(false, true); else GDA_MIXUP_BWD(false, false); }
#undef GDA_MIXUP_BWD
    GDA_LAUNCH_CHECK();
    k_mixup_bias<<<(unsigned)gda_cdiv(h, CF_COLS), TB, 0, s>>>(partial, blocks, (int)h, gbias);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}
