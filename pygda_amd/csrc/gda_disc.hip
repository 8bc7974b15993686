// Gradient-reversal + linear domain discriminator + softmax cross-entropy, fused (gfx950).
//
// Replaces, for the adversarial branch of A2GNN / GRADE(JS):
//   GradReverse.apply            pygda/nn/reverse_layer.py:39 (identity), :65-66 (-alpha*g)
//   Linear(h, 2)                 pygda/nn/a2gnn_base.py:69-70,201 ; grade_base.py:64-73
//   F.cross_entropy(cat(s,t))    pygda/models/a2gnn.py:197-205 ; pygda/models/grade.py:170-176
// without materialising cat(source, target), the logits, or the reversed activations.
//
// One wavefront per feature row (grid-stride); lane l owns columns l, l+64, ...; the C
// dot products are wave butterflies.  The weight gradient is a fixed-order two-stage
// reduction (per-workgroup partials, then a sequential sum per element): deterministic.
#include "gda_common.h"

namespace {

constexpr int TB = 256;
constexpr int WAVES = TB / 64;
constexpr int MAXC = 4;
constexpr int MAX_BLOCKS = 512;

struct Feat {
    const float* src; int64_t ld_src; int64_t n_src;
    const float* tgt; int64_t ld_tgt; int64_t n_tgt;
};

__device__ __forceinline__ const float* feat_row(const Feat& F, int64_t r) {
    return r < F.n_src ? F.src + r * F.ld_src : F.tgt + (r - F.n_src) * F.ld_tgt;
}

__device__ __forceinline__ float wave_sum(float v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int CPL>
__global__ void __launch_bounds__(TB)
k_fwd(Feat F, int h, int C, const float* __restrict__ W, const float* __restrict__ b,
      const int64_t* __restrict__ labels, float* __restrict__ probs, double* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float Ws[];      // [C][h]
    __shared__ double red[WAVES];
    for (int k = threadIdx.x; k < C * h; k += TB) Ws[k] = W[k];
    __syncthreads();
    const int lane = threadIdx.x % 64, wave = threadIdx.x / 64;
    const int64_t n = F.n_src + F.n_tgt;
    double local = 0.0;
    for (int64_t r = (int64_t)blockIdx.x * WAVES + wave; r < n; r += (int64_t)gridDim.x * WAVES) {
        const float* p = feat_row(F, r);
        float f[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) { const int c = lane + 64 * j; f[j] = c < h ? p[c] : 0.f; }
        float z[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            float s = 0.f;
            if (c < C) {
#pragma unroll
                for (int j = 0; j < CPL; ++j) { const int col = lane + 64 * j; if (col < h) s = fmaf(f[j], Ws[c * h + col], s); }
                s = wave_sum(s) + b[c];
            }
            z[c] = s;
        }
        float mx = z[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, z[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(z[c] - mx);
        const float lse = mx + logf(se);
        const int lab = labels ? (int)labels[r] : (r < F.n_src ? 0 : 1);
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C && lane == c) probs[r * C + c] = expf(z[c] - lse);
        float zl = z[0];
#pragma unroll
        for (int c = 1; c < MAXC; ++c) zl = (c == lab) ? z[c] : zl;
        local += (double)(lse - zl);
    }
    // every lane of a wave carries the same `local`; reduce the 4 waves in fixed order
    if (lane == 0) red[wave] = local;
    __syncthreads();
    if (threadIdx.x == 0) { double s = 0.0; for (int w = 0; w < WAVES; ++w) s += red[w]; partial[blockIdx.x] = s; }
}

// 64 threads: lane l adds the partials l, l + 64, ... in order, then a fixed shuffle tree (one thread adding block after
// block was a chain of nblocks dependent loads: 27 us at GRADE's 14 k rows)
__global__ void k_fwd_finalize(const double* __restrict__ partial, int nblocks, int64_t n,
                               float* __restrict__ loss) {
    double s = 0.0;
    for (int k = threadIdx.x; k < nblocks; k += 64) s += partial[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (threadIdx.x == 0) loss[0] = (float)(s / (double)n);
}

template <int CPL>
__global__ void __launch_bounds__(TB)
k_bwd(Feat F, int h, int C, const float* __restrict__ W, const int64_t* __restrict__ labels,
      const float* __restrict__ probs, const float* __restrict__ grad_loss, float alpha,
      float* __restrict__ gsrc, float* __restrict__ gtgt, float* __restrict__ gw_partial,
      float* __restrict__ gb_partial) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Ws = sm;                               // [C][h]
    float* acc_sh = sm + C * h;                   // [WAVES][C][h] wave partials of gW
    __shared__ float gb_sh[WAVES][MAXC];
    for (int k = threadIdx.x; k < C * h; k += TB) Ws[k] = W[k];
    __syncthreads();
    const int lane = threadIdx.x % 64, wave = threadIdx.x / 64;
    const int64_t n = F.n_src + F.n_tgt;
    const float scale = grad_loss[0] / (float)n;
    float gw[MAXC][CPL], gb[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) { gb[c] = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) gw[c][j] = 0.f; }
    for (int64_t r = (int64_t)blockIdx.x * WAVES + wave; r < n; r += (int64_t)gridDim.x * WAVES) {
        const float* p = feat_row(F, r);
        float* g = r < F.n_src ? (gsrc ? gsrc + r * h : nullptr) : (gtgt ? gtgt + (r - F.n_src) * h : nullptr);
        const int lab = labels ? (int)labels[r] : (r < F.n_src ? 0 : 1);
        float dz[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) dz[c] = c < C ? (probs[r * C + c] - (c == lab ? 1.f : 0.f)) * scale : 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int col = lane + 64 * j;
            if (col >= h) continue;
            const float fv = p[col];
            float gx = 0.f;
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < C) { gw[c][j] = fmaf(dz[c], fv, gw[c][j]); gx = fmaf(dz[c], Ws[c * h + col], gx); }
            if (g) g[col] = -alpha * gx;          // reverse_layer.py:65: grad.neg() * alpha
        }
#pragma unroll
        for (int c = 0; c < MAXC; ++c) gb[c] += dz[c];
    }
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int j = 0; j < CPL; ++j) { const int col = lane + 64 * j; if (col < h) acc_sh[(wave * C + c) * h + col] = gw[c][j]; }
    if (lane == 0) for (int c = 0; c < C; ++c) gb_sh[wave][c] = gb[c];
    __syncthreads();
    for (int k = threadIdx.x; k < C * h; k += TB) {
        float s = 0.f;
        for (int w = 0; w < WAVES; ++w) s += acc_sh[w * C * h + k];
        gw_partial[(int64_t)blockIdx.x * C * h + k] = s;
    }
    if (threadIdx.x < C) {
        float s = 0.f;
        for (int w = 0; w < WAVES; ++w) s += gb_sh[w][threadIdx.x];
        gb_partial[blockIdx.x * MAXC + threadIdx.x] = s;
    }
}

// 256 threads = 64 outputs x 4 chains: chain q adds the blocks q, q + 4, ... in order with eight loads in flight, the four
// chain sums are combined as (c0 + c1) + (c2 + c3).  Outputs: the C h weight gradients, then (gb given) the C bias
// gradients.  (One thread per output adding block after block: 164 us at GRADE's shapes.)
__global__ void __launch_bounds__(256)
k_bwd_finalize(const float* __restrict__ gw_partial, const float* __restrict__ gb_partial,
               int nblocks, int h, int C, float* __restrict__ gW, float* __restrict__ gb) {
    __shared__ float chain[4][64];
    const int q = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int k = blockIdx.x * 64 + l;
    const float* src = nullptr;
    float* dst = nullptr;
    int64_t stride = 0;
    if (k < C * h) { src = gw_partial + k; dst = gW + k; stride = (int64_t)C * h; }
    else if (gb && k - C * h < C) { src = gb_partial + (k - C * h); dst = gb + (k - C * h); stride = MAXC; }
    float s = 0.f;
    if (src) {
        int b = q;
        for (; b + 28 < nblocks; b += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(b + 4 * u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < nblocks; b += 4) s += src[(int64_t)b * stride];
    }
    chain[q][l] = s;
    __syncthreads();
    if (q == 0 && dst) *dst = (chain[0][l] + chain[1][l]) + (chain[2][l] + chain[3][l]);
}

int nblocks_for(int64_t n) {
    int64_t nb = gda_cdiv(n, WAVES * 4);           // >= 4 rows per wave before adding workgroups
    if (nb < 1) nb = 1;
    if (nb > MAX_BLOCKS) nb = MAX_BLOCKS;
    return (int)nb;
}

struct DiscWs { double* loss_partial; float* gw_partial; float* gb_partial; size_t total; };

DiscWs carve(void* base, int64_t h, int C) {
    DiscWs w{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void* p = base ? (void*)((char*)base + off) : nullptr;
        off += gda_align_up(bytes, 256);
        return p;
    };
    w.loss_partial = (double*)take(sizeof(double) * MAX_BLOCKS);
    w.gw_partial = (float*)take(sizeof(float) * MAX_BLOCKS * C * h);
    w.gb_partial = (float*)take(sizeof(float) * MAX_BLOCKS * MAXC);
    w.total = off;
    return w;
}

int check(const float* fs, int64_t lds, int64_t ns, const float* ft, int64_t ldt, int64_t nt,
          int64_t h, int C) {
    if (ns < 0 || nt < 0 || ns + nt <= 0 || h <= 0 || lds < h || ldt < h) return GDA_E_SIZE;
    if ((ns > 0 && !fs) || (nt > 0 && !ft)) return GDA_E_NULL;
    if (C < 2 || C > MAXC || h > 64 * 16) return GDA_E_UNSUPPORTED;
    return GDA_OK;
}

}  // namespace

extern "C" size_t gda_grl_disc_workspace_bytes(int64_t n_rows, int64_t h, int C) {
    (void)n_rows;
    if (h <= 0 || C <= 0) return 0;
    return carve(nullptr, h, C).total;
}

#define CPL_SWITCH(h, CALL)                      \
    do {                                         \
        const int cpl_ = (int)gda_cdiv(h, 64);   \
        if (cpl_ <= 1) { CALL(1); }              \
        else if (cpl_ <= 2) { CALL(2); }         \
        else if (cpl_ <= 4) { CALL(4); }         \
        else if (cpl_ <= 8) { CALL(8); }         \
        else { CALL(16); }                       \
    } while (0)

extern "C" int gda_grl_disc_ce_fwd_f32(const float* feat_src, int64_t ld_src, int64_t n_src,
                                       const float* feat_tgt, int64_t ld_tgt, int64_t n_tgt,
                                       int64_t h, int C, const float* W, const float* b,
                                       const int64_t* labels, float* probs, float* loss,
                                       void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    int st = check(feat_src, ld_src, n_src, feat_tgt, ld_tgt, n_tgt, h, C);
    if (st != GDA_OK) return st;
    if (!W || !b || !probs || !loss || !workspace) return GDA_E_NULL;
    DiscWs ws = carve(workspace, h, C);
    if (workspace_bytes < ws.total) return GDA_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const Feat F{feat_src, ld_src, n_src, feat_tgt, ld_tgt, n_tgt};
    const int nb = nblocks_for(n_src + n_tgt);
    const size_t lds = sizeof(float) * C * h;
#define CALL(CPL) k_fwd<CPL><<<nb, TB, lds, stream>>>(F, (int)h, C, W, b, labels, probs, ws.loss_partial)
    CPL_SWITCH(h, CALL);
#undef CALL
    GDA_LAUNCH_CHECK();
    k_fwd_finalize<<<1, 64, 0, stream>>>(ws.loss_partial, nb, n_src + n_tgt, loss);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_grl_disc_ce_bwd_f32(const float* feat_src, int64_t ld_src, int64_t n_src,
                                       const float* feat_tgt, int64_t ld_tgt, int64_t n_tgt,
                                       int64_t h, int C, const float* W, const int64_t* labels,
                                       const float* probs, const float* grad_loss, float alpha,
                                       float* gfeat_src, float* gfeat_tgt, float* gW, float* gb,
                                       void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    int st = check(feat_src, ld_src, n_src, feat_tgt, ld_tgt, n_tgt, h, C);
    if (st != GDA_OK) return st;
    if (!W || !probs || !grad_loss || !gW || !workspace) return GDA_E_NULL;
    DiscWs ws = carve(workspace, h, C);
    if (workspace_bytes < ws.total) return GDA_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const Feat F{feat_src, ld_src, n_src, feat_tgt, ld_tgt, n_tgt};
    const int nb = nblocks_for(n_src + n_tgt);
    const size_t lds = sizeof(float) * C * h * (1 + WAVES);
    if (lds > 60 * 1024) return GDA_E_UNSUPPORTED;
#define CALL(CPL) k_bwd<CPL><<<nb, TB, lds, stream>>>(F, (int)h, C, W, labels, probs, grad_loss, alpha, \
                                                      gfeat_src, gfeat_tgt, ws.gw_partial, ws.gb_partial)
    CPL_SWITCH(h, CALL);
#undef CALL
    GDA_LAUNCH_CHECK();
    k_bwd_finalize<<<(unsigned)gda_cdiv(C * h + C, 64), 256, 0, stream>>>(ws.gw_partial, ws.gb_partial, nb,
                                                                        (int)h, C, gW, gb);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}
