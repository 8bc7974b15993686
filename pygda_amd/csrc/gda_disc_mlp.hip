// Gradient-reversal + two-layer domain discriminator + per-domain softmax cross-entropy, fused (gfx950).
//
// Replaces, for UDAGCN's domain branch (pygda/models/udagcn.py:176-190, pygda/nn/udagcn_base.py:157-162):
//   GradReverse.apply(encoded_*, alpha)                          pygda/nn/reverse_layer.py:39,65-66
//   domain_model = Linear(h, a) - ReLU - Dropout(0.1) - Linear(a, 2)
//   loss_func(source_domain_preds, zeros) + loss_func(target_domain_preds, ones)    (two means, one per domain)
// -- in the reference ten library launches forward per domain (two GEMMs, bias adds, relu, dropout, log-softmax,
// NLL) and their autograd twins; here one row kernel and one fold each way.  No logits, hidden activations, masks
// or reversed copies are materialised: the backward pass recomputes a row's 40 hidden units from the row (a
// 40 x 128 product, cheaper than reading them back) with the same Philox keep-bits.
//
// Row kernel: four rows per wavefront; lane k (k < a) owns hidden unit k -- z1_k from W1 in padded LDS with the row
// values broadcast by v_readlane --, lane l owns columns l, l + 64 of the row for the input gradient and the
// weight-gradient outer product.  Every reduction over rows is fixed-order (per-wave registers -> workgroup LDS in
// wave order -> per-workgroup partials -> sequential fold): deterministic.
#include "gda_common.h"
#include "gda_philox.h"

namespace {

constexpr int TB = 256;
constexpr int WAVES = TB / 64;
constexpr int RB = 4;              // rows per wavefront: every W1 word read from LDS feeds RB rows
constexpr int HC = 2;              // columns per lane (h <= 128)
constexpr int AMAX = 64;           // hidden width limit (one lane per unit)
constexpr int MAX_BLOCKS = 512;    // two workgroups per CU (round 6: 128 left half the chip idle -- forward 73 us, backward 130 us
                                   // at UDAGCN's 16 k rows; the folds below sum their partials on four chains with eight loads in flight)

// head = 0: two logits per row, softmax cross-entropy against the row's domain (UDAGCN's domain model); head = 1: ONE logit per
// row through a sigmoid, the loss term of a domain is the MEAN of its rows' sigmoids (AdaGCN's critic D = Linear - ReLU -
// Dropout - Linear(a, 1) - Sigmoid in the encoder's loss, adagcn.py:190-193: W2 is [1, a], b2 [1])
struct Mlp { const float* W1; const float* b1; const float* W2; const float* b2; int h, a; int head; };
struct Rows2 { const float* es; int64_t ld_s, n_s; const float* et; int64_t ld_t, n_t; };
struct Drop { float p; uint64_t seed; const int64_t* step; uint32_t site; };

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float bcast(float v, int src_lane) {      // src_lane is wave-uniform
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

// keep-factor of hidden unit k of row `row` (numbered inside its domain) at call site `site`: 0 or 1/(1-p)
__device__ __forceinline__ float keep_factor(const Drop& dr, uint64_t st, uint32_t site, int64_t row, int a, int k) {
    if (dr.p <= 0.f) return 1.f;
    const uint64_t e = (uint64_t)row * (uint64_t)a + (uint64_t)k;
    uint32_t r[4];
    GdaPhilox::gen(dr.seed, (st << 20) ^ site, e >> 2, r);
    const double t = (double)dr.p * 4294967296.0;
    const uint32_t thresh = (uint32_t)(t > 4294967295.0 ? 4294967295.0 : t);
    return r[e & 3] >= thresh ? 1.f / (1.f - dr.p) : 0.f;
}

__device__ __forceinline__ void load_w1(const Mlp& M, float* W1s) {
    for (int e = threadIdx.x; e < M.a * M.h; e += TB) W1s[(e / M.h) * (M.h + 1) + e % M.h] = M.W1[e];
}

// what forward and backward share: the rows of a group, their hidden units and logits
struct Group {
    float x[RB][HC];          // lane l: columns l, l + 64
    float mr[RB], hid[RB];    // lane k: keep * relu'(a_k), drop(relu(a_k))
    float z0[RB], z1[RB];     // wave-uniform logits
    bool live[RB];
    int dom[RB];              // 0 source, 1 target
};

__device__ __forceinline__ void eval_group(const Mlp& M, const Rows2& R, const Drop& dr, uint64_t st, const float* W1s,
                                           int64_t base, int lane, float b1k, float w20k, float w21k, float b20, float b21,
                                           Group& G) {
    const int h = M.h, a = M.a;
    const int64_t n = R.n_s + R.n_t;
    float keep[RB];
#pragma unroll
    for (int q = 0; q < RB; ++q) {
        const int64_t r = base + q;
        G.live[q] = r < n;
        G.dom[q] = (G.live[q] && r >= R.n_s) ? 1 : 0;
        const int64_t rd = G.dom[q] ? r - R.n_s : r;
        const float* p = G.dom[q] ? R.et + rd * R.ld_t : R.es + rd * R.ld_s;
#pragma unroll
        for (int c = 0; c < HC; ++c) {
            const int j = c * 64 + lane;
            G.x[q][c] = (G.live[q] && j < h) ? p[j] : 0.f;
        }
        keep[q] = (G.live[q] && lane < a) ? keep_factor(dr, st, dr.site + (uint32_t)G.dom[q], rd, a, lane) : 0.f;
    }
    // lane k: a_k = sum_j W1[k][j] x_j -- one conflict-free LDS read of W1[k][j] per j, the RB row values by readlane
    float ak[RB];
#pragma unroll
    for (int q = 0; q < RB; ++q) ak[q] = 0.f;
    const float* w = W1s + (size_t)(lane < a ? lane : 0) * (h + 1);
#pragma unroll
    for (int c = 0; c < HC; ++c) {
        if (c * 64 >= h) break;
        const int cols = h - c * 64 < 64 ? h - c * 64 : 64;
        for (int jj = 0; jj < cols; ++jj) {
            const float wv = w[c * 64 + jj];
#pragma unroll
            for (int q = 0; q < RB; ++q) ak[q] = fmaf(wv, bcast(G.x[q][c], jj), ak[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < RB; ++q) {
        const float av = ak[q] + b1k;
        G.mr[q] = (lane < a && av > 0.f) ? keep[q] : 0.f;
        G.hid[q] = G.mr[q] * av;
        G.z0[q] = wave_sum(w20k * G.hid[q]) + b20;
        G.z1[q] = wave_sum(w21k * G.hid[q]) + b21;
    }
}

// part[2][blocks]: sum of the rows' cross-entropy terms per domain
__global__ void __launch_bounds__(TB)
k_mlp_ce_fwd(Mlp M, Rows2 R, Drop dr, double* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float W1s[];
    load_w1(M, W1s);
    __syncthreads();
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    const uint64_t st = dr.p > 0.f ? (uint64_t)dr.step[0] : 0;
    const int a = M.a;
    const float b1k = lane < a ? M.b1[lane] : 0.f;
    const float w20k = lane < a ? M.W2[lane] : 0.f, w21k = (lane < a && M.head == 0) ? M.W2[a + lane] : 0.f;
    const float b20 = M.b2[0], b21 = M.head == 0 ? M.b2[1] : 0.f;
    const int64_t groups = (R.n_s + R.n_t + RB - 1) / RB;
    double acc[2] = {0.0, 0.0};
    for (int64_t grp = (int64_t)blockIdx.x * WAVES + wave; grp < groups; grp += (int64_t)gridDim.x * WAVES) {
        Group G;
        eval_group(M, R, dr, st, W1s, grp * RB, lane, b1k, w20k, w21k, b20, b21, G);
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            if (!G.live[q]) continue;
            if (M.head == 1) { acc[G.dom[q]] += (double)(1.f / (1.f + expf(-G.z0[q]))); continue; }
            const float mx = fmaxf(G.z0[q], G.z1[q]);
            const float lse = mx + logf(expf(G.z0[q] - mx) + expf(G.z1[q] - mx));
            acc[G.dom[q]] += (double)(lse - (G.dom[q] ? G.z1[q] : G.z0[q]));
        }
    }
    __shared__ double red[WAVES][2];
    if (lane == 0) { red[wave][0] = acc[0]; red[wave][1] = acc[1]; }
    __syncthreads();
    if (threadIdx.x < 2) {
        double s = 0.0;
        for (int w = 0; w < WAVES; ++w) s += red[w][threadIdx.x];
        part[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = s;
    }
}

__global__ void k_mlp_ce_fwd_final(const double* __restrict__ part, int blocks, int64_t n_s, int64_t n_t,
                                   float* __restrict__ losses) {
    // 64 threads: lanes 0-31 fold the source partials, 32-63 the target's -- lane l the blocks l, l + 32, ... in order,
    // then a fixed shuffle tree over the 32 lanes
    const int dom = threadIdx.x >> 5, l = threadIdx.x & 31;
    double s = 0.0;
    for (int b = l; b < blocks; b += 32) s += part[(int64_t)dom * blocks + b];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) s += __shfl_down(s, off, 32);
    __shared__ float both[2];
    if (l == 0) {
        const int64_t n = dom ? n_t : n_s;
        both[dom] = losses[dom] = n > 0 ? (float)(s / (double)n) : 0.f;   // mean over an empty domain: 0 (no rows, no term)
    }
    __syncthreads();
    if (threadIdx.x == 0) losses[2] = both[0] + both[1];                  // what udagcn.py:183-190 adds to the loss
}

// Per-workgroup partials: pW1 [blocks][a*h], pb1 [blocks][a], pW2 [blocks][2a], pb2 [blocks][2].
// grad[2] = upstream gradients of the two means; alpha (device scalar if alpha_dev) = the reversal's factor.
template <int AK>
__global__ void __launch_bounds__(TB)
k_mlp_ce_bwd(Mlp M, Rows2 R, Drop dr, const float* __restrict__ grad, int grad_stride, float alpha,
             const float* __restrict__ alpha_dev,
             float* __restrict__ gs, float* __restrict__ gt, float* __restrict__ pW1, float* __restrict__ pb1,
             float* __restrict__ pW2, float* __restrict__ pb2) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* W1s = sm;                                         // [a][h + 1]
    float* accW = sm + (size_t)M.a * (M.h + 1);              // [a][h]: this workgroup's gW1, waves added in order
    load_w1(M, W1s);
    __syncthreads();
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    const uint64_t st = dr.p > 0.f ? (uint64_t)dr.step[0] : 0;
    const int h = M.h, a = M.a;
    const float b1k = lane < a ? M.b1[lane] : 0.f;
    const float w20k = lane < a ? M.W2[lane] : 0.f, w21k = (lane < a && M.head == 0) ? M.W2[a + lane] : 0.f;
    const float b20 = M.b2[0], b21 = M.head == 0 ? M.b2[1] : 0.f;
    const float neg_alpha = -(alpha_dev ? alpha_dev[0] : alpha);
    const float sc[2] = {R.n_s > 0 ? grad[0] / (float)R.n_s : 0.f, R.n_t > 0 ? grad[grad_stride] / (float)R.n_t : 0.f};
    const int64_t groups = (R.n_s + R.n_t + RB - 1) / RB;
    float gw[AK][HC];
#pragma unroll
    for (int k = 0; k < AK; ++k)
#pragma unroll
        for (int c = 0; c < HC; ++c) gw[k][c] = 0.f;
    float gb1 = 0.f, gw20 = 0.f, gw21 = 0.f, gb20 = 0.f, gb21 = 0.f;
    for (int64_t grp = (int64_t)blockIdx.x * WAVES + wave; grp < groups; grp += (int64_t)gridDim.x * WAVES) {
        Group G;
        eval_group(M, R, dr, st, W1s, grp * RB, lane, b1k, w20k, w21k, b20, b21, G);
        float u[RB];
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const float mx = fmaxf(G.z0[q], G.z1[q]);
            const float e0 = expf(G.z0[q] - mx), e1 = expf(G.z1[q] - mx);
            const float inv = 1.f / (e0 + e1);
            const float s = G.live[q] ? sc[G.dom[q]] : 0.f;
            const float sgm = 1.f / (1.f + expf(-G.z0[q]));
            const float dz0 = M.head == 1 ? sgm * (1.f - sgm) * s : (e0 * inv - (G.dom[q] == 0 ? 1.f : 0.f)) * s;
            const float dz1 = M.head == 1 ? 0.f : (e1 * inv - (G.dom[q] == 1 ? 1.f : 0.f)) * s;
            u[q] = G.mr[q] * fmaf(w20k, dz0, w21k * dz1);          // d loss / d a_k
            gw20 = fmaf(dz0, G.hid[q], gw20);
            gw21 = fmaf(dz1, G.hid[q], gw21);
            gb20 += dz0;
            gb21 += dz1;
            gb1 += u[q];
        }
        // input gradient dx_j = sum_k W1[k][j] u_k and the outer product gW1[k][j] += u_k x_j: lane l owns columns
        // l, l + 64; u_k broadcast from lane k
        float dx[RB][HC];
#pragma unroll
        for (int q = 0; q < RB; ++q)
#pragma unroll
            for (int c = 0; c < HC; ++c) dx[q][c] = 0.f;
#pragma unroll
        for (int k = 0; k < AK; ++k) {
            if (k >= a) continue;                              // uniform; static indices keep gw[][] in registers
            float uk[RB];
#pragma unroll
            for (int q = 0; q < RB; ++q) uk[q] = bcast(u[q], k);
#pragma unroll
            for (int c = 0; c < HC; ++c) {
                const int j = c * 64 + lane;
                const float wv = j < h ? W1s[(size_t)k * (h + 1) + j] : 0.f;
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    dx[q][c] = fmaf(wv, uk[q], dx[q][c]);
                    gw[k][c] = fmaf(uk[q], G.x[q][c], gw[k][c]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            if (!G.live[q]) continue;
            const int64_t r = grp * RB + q;
            float* g = G.dom[q] ? (gt ? gt + (r - R.n_s) * h : nullptr) : (gs ? gs + r * h : nullptr);
            if (!g) continue;
#pragma unroll
            for (int c = 0; c < HC; ++c) {
                const int j = c * 64 + lane;
                if (j < h) g[j] = neg_alpha * dx[q][c];            // reverse_layer.py:65: grad.neg() * alpha
            }
        }
    }
    // workgroup sums in wave order
    __shared__ float small[WAVES][3 * AMAX + 2];
    if (lane < a) { small[wave][lane] = gb1; small[wave][AMAX + lane] = gw20; small[wave][2 * AMAX + lane] = gw21; }
    if (lane == 0) { small[wave][3 * AMAX] = gb20; small[wave][3 * AMAX + 1] = gb21; }
    for (int w = 0; w < WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int k = 0; k < AK; ++k) {
                if (k >= a) continue;                              // uniform; static indices keep gw[][] in registers
#pragma unroll
                for (int c = 0; c < HC; ++c) {
                    const int j = c * 64 + lane;
                    if (j < h) accW[(size_t)k * h + j] = (w == 0 ? 0.f : accW[(size_t)k * h + j]) + gw[k][c];
                }
            }
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < a * h; e += TB) pW1[(int64_t)blockIdx.x * a * h + e] = accW[e];
    const int t = threadIdx.x;
    if (t < 3 * a + 2) {
        int slot;
        float* dst;
        if (t < a) { slot = t; dst = pb1 + (int64_t)blockIdx.x * a + t; }
        else if (t < 3 * a) { slot = AMAX + (t - a < a ? t - a : AMAX + (t - 2 * a)); dst = pW2 + (int64_t)blockIdx.x * 2 * a + (t - a); }
        else { slot = 3 * AMAX + (t - 3 * a); dst = pb2 + (int64_t)blockIdx.x * 2 + (t - 3 * a); }
        float v = 0.f;
        for (int w = 0; w < WAVES; ++w) v += small[w][slot];
        *dst = v;                                       // (head 1: the second logit's entries are zeros nobody folds)
    }
}

// out[e] = sum over blocks of partial[b][e], blocks in order -- the four partial arrays in ONE launch
struct Fold4 { const float* part[4]; float* out[4]; int64_t elems[4]; int64_t stride[4]; };     // stride: floats per block of a partial array (>= elems)

// 256 threads = 64 outputs x 4 chains: chain q adds the blocks q, q + 4, ... in order, eight loads in flight at a time; the
// four chain sums are combined as (c0 + c1) + (c2 + c3).  (One thread per output adding block after block was a chain of
// `blocks` dependent loads: 35 us at 128 blocks.)
__global__ void __launch_bounds__(256) k_fold4(Fold4 F, int blocks) {
    __shared__ float chain[4][64];
    const int q = threadIdx.x >> 6, l = threadIdx.x & 63;
    int64_t e = (int64_t)blockIdx.x * 64 + l;
    const float* src = nullptr;
    float* dst = nullptr;
    int64_t stride = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (!src && e < F.elems[i]) { src = F.part[i] + e; dst = F.out[i] + e; stride = F.stride[i]; }
        if (!src) e -= F.elems[i];
    }
    float s = 0.f;
    if (src) {
        int b = q;
        for (; b + 28 < blocks; b += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(b + 4 * u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < blocks; b += 4) s += src[(int64_t)b * stride];
    }
    chain[q][l] = s;
    __syncthreads();
    if (q == 0 && dst) *dst = (chain[0][l] + chain[1][l]) + (chain[2][l] + chain[3][l]);
}

int blocks_for(int64_t n) {
    int64_t nb = gda_cdiv(n, (int64_t)WAVES * RB * 2);          // >= 2 groups per wave before adding workgroups
    if (nb < 1) nb = 1;
    if (nb > MAX_BLOCKS) nb = MAX_BLOCKS;
    return (int)nb;
}

struct MlpWs { double* part; float* pW1; float* pb1; float* pW2; float* pb2; size_t total; };

MlpWs carve(void* base, int64_t h, int64_t a) {
    MlpWs w{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void* p = base ? (void*)((char*)base + off) : nullptr;
        off += gda_align_up(bytes, 256);
        return p;
    };
    w.part = (double*)take(sizeof(double) * 2 * MAX_BLOCKS);
    w.pW1 = (float*)take(sizeof(float) * MAX_BLOCKS * a * h);
    w.pb1 = (float*)take(sizeof(float) * MAX_BLOCKS * a);
    w.pW2 = (float*)take(sizeof(float) * MAX_BLOCKS * 2 * a);
    w.pb2 = (float*)take(sizeof(float) * MAX_BLOCKS * 2);
    w.total = off;
    return w;
}

int check(const float* es, int64_t ld_s, int64_t n_s, const float* et, int64_t ld_t, int64_t n_t, int64_t h, int64_t a) {
    if (n_s < 0 || n_t < 0 || n_s + n_t <= 0 || h <= 0 || a <= 0 || ld_s < h || ld_t < h) return GDA_E_SIZE;
    if ((n_s > 0 && !es) || (n_t > 0 && !et)) return GDA_E_NULL;
    if (h > 64 * HC || a > AMAX) return GDA_E_UNSUPPORTED;
    return GDA_OK;
}

}  // namespace

extern "C" size_t gda_grl_mlp_ce_workspace_bytes(int64_t h, int64_t a) {
    if (h <= 0 || a <= 0) return 0;
    return carve(nullptr, h, a).total;
}

extern "C" int gda_grl_mlp_ce_fwd_f32(const float* es, int64_t ld_s, int64_t n_s, const float* et, int64_t ld_t, int64_t n_t,
                                      int64_t h, int64_t a, const float* W1, const float* b1, const float* W2,
                                      const float* b2, float dropout_p, uint64_t seed, const int64_t* step, uint32_t site,
                                      float* losses, void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    return gda_mlp_head_fwd_f32(0, es, ld_s, n_s, et, ld_t, n_t, h, a, W1, b1, W2, b2, dropout_p, seed, step, site, losses,
                                workspace, workspace_bytes, stream_);
}

extern "C" int gda_mlp_head_fwd_f32(int head, const float* es, int64_t ld_s, int64_t n_s, const float* et, int64_t ld_t, int64_t n_t,
                                    int64_t h, int64_t a, const float* W1, const float* b1, const float* W2,
                                    const float* b2, float dropout_p, uint64_t seed, const int64_t* step, uint32_t site,
                                    float* losses, void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    if (head != 0 && head != 1) return GDA_E_UNSUPPORTED;
    int st = check(es, ld_s, n_s, et, ld_t, n_t, h, a);
    if (st != GDA_OK) return st;
    if (!W1 || !b1 || !W2 || !b2 || !losses || !workspace || (dropout_p > 0.f && !step)) return GDA_E_NULL;
    if (dropout_p < 0.f || dropout_p >= 1.f) return GDA_E_SIZE;
    MlpWs ws = carve(workspace, h, a);
    if (workspace_bytes < ws.total) return GDA_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const Mlp M{W1, b1, W2, b2, (int)h, (int)a, head};
    const Rows2 R{es, ld_s, n_s, et, ld_t, n_t};
    const Drop dr{dropout_p, seed, step, site};
    const int nb = blocks_for(n_s + n_t);
    const size_t lds = sizeof(float) * (size_t)a * (h + 1);
    k_mlp_ce_fwd<<<nb, TB, lds, stream>>>(M, R, dr, ws.part);
    GDA_LAUNCH_CHECK();
    k_mlp_ce_fwd_final<<<1, 64, 0, stream>>>(ws.part, nb, n_s, n_t, losses);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_grl_mlp_ce_bwd_f32(const float* es, int64_t ld_s, int64_t n_s, const float* et, int64_t ld_t, int64_t n_t,
                                      int64_t h, int64_t a, const float* W1, const float* b1, const float* W2,
                                      const float* b2, float dropout_p, uint64_t seed, const int64_t* step, uint32_t site,
                                      const float* grad_losses, int grad_stride, float alpha, const float* alpha_dev,
                                      float* g_es, float* g_et, float* gW1, float* gb1, float* gW2, float* gb2,
                                      void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    return gda_mlp_head_bwd_f32(0, es, ld_s, n_s, et, ld_t, n_t, h, a, W1, b1, W2, b2, dropout_p, seed, step, site, grad_losses,
                                grad_stride, alpha, alpha_dev, g_es, g_et, gW1, gb1, gW2, gb2, workspace, workspace_bytes, stream_);
}

extern "C" int gda_mlp_head_bwd_f32(int head, const float* es, int64_t ld_s, int64_t n_s, const float* et, int64_t ld_t, int64_t n_t,
                                    int64_t h, int64_t a, const float* W1, const float* b1, const float* W2,
                                    const float* b2, float dropout_p, uint64_t seed, const int64_t* step, uint32_t site,
                                    const float* grad_losses, int grad_stride, float alpha, const float* alpha_dev,
                                    float* g_es, float* g_et, float* gW1, float* gb1, float* gW2, float* gb2,
                                    void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    if (head != 0 && head != 1) return GDA_E_UNSUPPORTED;
    int st = check(es, ld_s, n_s, et, ld_t, n_t, h, a);
    if (st != GDA_OK) return st;
    if (!W1 || !b1 || !W2 || !b2 || !grad_losses || !gW1 || !gb1 || !gW2 || !gb2 || !workspace ||
        (dropout_p > 0.f && !step)) return GDA_E_NULL;
    if (dropout_p < 0.f || dropout_p >= 1.f || (grad_stride != 0 && grad_stride != 1)) return GDA_E_SIZE;
    MlpWs ws = carve(workspace, h, a);
    if (workspace_bytes < ws.total) return GDA_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const Mlp M{W1, b1, W2, b2, (int)h, (int)a, head};
    const Rows2 R{es, ld_s, n_s, et, ld_t, n_t};
    const Drop dr{dropout_p, seed, step, site};
    const int nb = blocks_for(n_s + n_t);
    const size_t lds = sizeof(float) * ((size_t)a * (h + 1) + (size_t)a * h);
    if (lds > 150 * 1024) return GDA_E_UNSUPPORTED;          // a <= 64, h <= 128: at most 66 KB
#define GDA_MLP_BWD(AK)                                                                                             \
    do {                                                                                                            \
        static size_t configured = 48 * 1024;     /* above the default limit: raise it to what this shape needs */  \
        if (lds > configured) {                                                                                     \
            GDA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp_ce_bwd<AK>),                        \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                 \
            configured = lds;                                                                                       \
        }                                                                                                           \
        k_mlp_ce_bwd<AK><<<nb, TB, lds, stream>>>(M, R, dr, grad_losses, grad_stride, alpha, alpha_dev, g_es, g_et, ws.pW1, \
                                                  ws.pb1, ws.pW2, ws.pb2);                                          \
    } while (0)
    if (a <= 16) GDA_MLP_BWD(16);
    else if (a <= 32) GDA_MLP_BWD(32);
    else if (a <= 48) GDA_MLP_BWD(48);
    else GDA_MLP_BWD(64);
#undef GDA_MLP_BWD
    GDA_LAUNCH_CHECK();
    // (head 1: W2 is [1, a] -- only the first logit's a + 1 entries are folded; the partials keep their [blocks][2 a] / [2] strides)
    const int64_t nz = head == 0 ? 2 : 1;
    const Fold4 F{{ws.pW1, ws.pb1, ws.pW2, ws.pb2}, {gW1, gb1, gW2, gb2}, {a * h, a, nz * a, nz}, {a * h, a, 2 * a, 2}};
    k_fold4<<<(unsigned)gda_cdiv(a * h + a + nz * a + nz, 64), 256, 0, stream>>>(F, nb);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// DANE's LSGAN discriminator head (pygda/models/dane.py:339-350,468-470: Linear(h, h) - ReLU - Linear(h, 1) on 8 x
// sample_size sampled rows per domain, loss mean((D(x) - target)^2)).  The first layer is a [rows, h] x [h, h]
// product -- GEMM-sized, it runs on the matrix-core kernels (csrc/gda_gemm.hip, bias in the epilogue); this is what
// follows it, one pass each way over Z = x W1^T + b1 ([rows, a] fp32, HBM-bound: forward reads Z once, backward
// reads Z and writes gZ):
//   forward : pre[r] = sum_k relu(Z[r,k]) w2[k] + b2 ;  loss = mean_r (pre[r] - target)^2
//   backward: dpre = 2 (pre - target) / rows * g ;  gZ[r,k] = dpre w2[k] [Z[r,k] > 0] ;  gw2[k] = sum_r dpre relu(Z[r,k]) ;
//             gb2 = sum_r dpre          (fixed-order sums; gW1, gb1, gx follow from gZ by GEMMs)
// One wavefront per row, lane l owns units l, l + 64, ... (a <= 256).
namespace {

constexpr int LS_UC = 4;             // units per lane
constexpr int LS_BLOCKS = 256;

__global__ void __launch_bounds__(TB)
k_lsgan_fwd(const float* __restrict__ Z, int64_t ldz, int64_t rows, int a, const float* __restrict__ w2,
            const float* __restrict__ b2, float target, float* __restrict__ pre, double* __restrict__ part) {
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    float w[LS_UC];
#pragma unroll
    for (int c = 0; c < LS_UC; ++c) { const int k = c * 64 + lane; w[c] = k < a ? w2[k] : 0.f; }
    const float bias = b2[0];
    double acc = 0.0;
    for (int64_t r = (int64_t)blockIdx.x * WAVES + wave; r < rows; r += (int64_t)gridDim.x * WAVES) {
        const float* z = Z + r * ldz;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < LS_UC; ++c) { const int k = c * 64 + lane; if (k < a) s = fmaf(fmaxf(z[k], 0.f), w[c], s); }
        const float p = wave_sum(s) + bias;
        if (lane == 0) pre[r] = p;
        acc += (double)((p - target) * (p - target));
    }
    __shared__ double red[WAVES];
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { double v = 0.0; for (int k = 0; k < WAVES; ++k) v += red[k]; part[blockIdx.x] = v; }
}

__global__ void k_lsgan_fwd_final(const double* __restrict__ part, int blocks, int64_t rows, float* __restrict__ loss) {
    double s = 0.0;                                       // 64 threads: lane l the blocks l, l + 64, ..., then a fixed tree
    for (int b = threadIdx.x; b < blocks; b += 64) s += part[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (threadIdx.x == 0) loss[0] = (float)(s / (double)rows);
}

// pw2 [blocks][a], pb2 [blocks]
__global__ void __launch_bounds__(TB)
k_lsgan_bwd(const float* __restrict__ Z, int64_t ldz, int64_t rows, int a, const float* __restrict__ w2,
            const float* __restrict__ pre, float target, const float* __restrict__ grad, float* __restrict__ gZ, int64_t ldg,
            float* __restrict__ pw2, float* __restrict__ pb2) {
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    float w[LS_UC], gw[LS_UC];
#pragma unroll
    for (int c = 0; c < LS_UC; ++c) { const int k = c * 64 + lane; w[c] = k < a ? w2[k] : 0.f; gw[c] = 0.f; }
    const float sc = 2.f * grad[0] / (float)rows;
    float gb = 0.f;
    for (int64_t r = (int64_t)blockIdx.x * WAVES + wave; r < rows; r += (int64_t)gridDim.x * WAVES) {
        const float* z = Z + r * ldz;
        float* g = gZ + r * ldg;
        const float dp = (pre[r] - target) * sc;
        gb += dp;
#pragma unroll
        for (int c = 0; c < LS_UC; ++c) {
            const int k = c * 64 + lane;
            if (k < a) {
                const float zv = z[k];
                g[k] = zv > 0.f ? dp * w[c] : 0.f;
                gw[c] = fmaf(dp, fmaxf(zv, 0.f), gw[c]);
            }
        }
    }
    __shared__ float sh[WAVES][64 * LS_UC + 1];
#pragma unroll
    for (int c = 0; c < LS_UC; ++c) sh[wave][c * 64 + lane] = gw[c];
    if (lane == 0) sh[wave][64 * LS_UC] = gb;
    __syncthreads();
    for (int k = threadIdx.x; k < a; k += TB) {
        float v = 0.f;
        for (int q = 0; q < WAVES; ++q) v += sh[q][k];
        pw2[(int64_t)blockIdx.x * a + k] = v;
    }
    if (threadIdx.x == 0) { float v = 0.f; for (int q = 0; q < WAVES; ++q) v += sh[q][64 * LS_UC]; pb2[blockIdx.x] = v; }
}

int ls_blocks(int64_t rows) {
    int64_t nb = gda_cdiv(rows, (int64_t)WAVES * 8);
    if (nb < 1) nb = 1;
    if (nb > LS_BLOCKS) nb = LS_BLOCKS;
    return (int)nb;
}

struct LsWs { double* part; float* pw2; float* pb2; size_t total; };

LsWs ls_carve(void* base, int64_t a) {
    LsWs w{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void* p = base ? (void*)((char*)base + off) : nullptr;
        off += gda_align_up(bytes, 256);
        return p;
    };
    w.part = (double*)take(sizeof(double) * LS_BLOCKS);
    w.pw2 = (float*)take(sizeof(float) * LS_BLOCKS * (a > 0 ? a : 1));
    w.pb2 = (float*)take(sizeof(float) * LS_BLOCKS);
    w.total = off;
    return w;
}

}  // namespace

extern "C" size_t gda_lsgan_head_workspace_bytes(int64_t a) {
    if (a <= 0) return 0;
    return ls_carve(nullptr, a).total;
}

extern "C" int gda_lsgan_head_fwd_f32(const float* Z, int64_t ldz, int64_t rows, int64_t a, const float* w2, const float* b2,
                                      float target, float* pre, float* loss, void* workspace, size_t workspace_bytes,
                                      gda_stream_t stream_) {
    if (rows <= 0 || a <= 0 || ldz < a) return GDA_E_SIZE;
    if (a > 64 * LS_UC) return GDA_E_UNSUPPORTED;
    if (!Z || !w2 || !b2 || !pre || !loss || !workspace) return GDA_E_NULL;
    LsWs ws = ls_carve(workspace, a);
    if (workspace_bytes < ws.total) return GDA_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const int nb = ls_blocks(rows);
    k_lsgan_fwd<<<nb, TB, 0, stream>>>(Z, ldz, rows, (int)a, w2, b2, target, pre, ws.part);
    GDA_LAUNCH_CHECK();
    k_lsgan_fwd_final<<<1, 64, 0, stream>>>(ws.part, nb, rows, loss);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_lsgan_head_bwd_f32(const float* Z, int64_t ldz, int64_t rows, int64_t a, const float* w2, const float* pre,
                                      float target, const float* grad_loss, float* gZ, int64_t ldg, float* gw2, float* gb2,
                                      void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    if (rows <= 0 || a <= 0 || ldz < a || ldg < a) return GDA_E_SIZE;
    if (a > 64 * LS_UC) return GDA_E_UNSUPPORTED;
    if (!Z || !w2 || !pre || !grad_loss || !gZ || !gw2 || !gb2 || !workspace) return GDA_E_NULL;
    LsWs ws = ls_carve(workspace, a);
    if (workspace_bytes < ws.total) return GDA_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const int nb = ls_blocks(rows);
    k_lsgan_bwd<<<nb, TB, 0, stream>>>(Z, ldz, rows, (int)a, w2, pre, target, grad_loss, gZ, ldg, ws.pw2, ws.pb2);
    GDA_LAUNCH_CHECK();
    const Fold4 F{{ws.pw2, ws.pb2, nullptr, nullptr}, {gw2, gb2, nullptr, nullptr}, {a, 1, 0, 0}, {a, 1, 0, 0}};
    k_fold4<<<(unsigned)gda_cdiv(a + 1, 64), 256, 0, stream>>>(F, nb);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}
