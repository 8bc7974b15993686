// Source classification loss: mean over rows of -log_softmax(logits)[label], forward + backward, fp32.
//
// Replaces `F.nll_loss(F.log_softmax(source_logits, dim=1), source_data.y)` (pygda/models/a2gnn.py:182,
// the same line in every trainer) -- four generic kernels per step (softmax / NLL, each way), of which
// the NLL reductions run as a single workgroup (144 us + 99 us on the 150k-row sampled batches of
// cfg-S).  One pass each way over [N, C] with C small: a thread owns whole rows (row reads are
// contiguous C*4 bytes; neighbouring threads read neighbouring rows), per-row log-sum-exp in
// registers, fixed-order double-precision block partials -> deterministic loss.
//   backward: gx[i, c] = (softmax(x_i)[c] - [c == y_i]) * gl / N.
#include "gda_common.h"

namespace {

constexpr int TB = 256;
constexpr int MAXC = 64;          // classes held in registers per row
constexpr int CE_BLOCKS = 256;    // partials of the loss sum

__device__ __forceinline__ float row_lse(const float* __restrict__ x, int C, float (&v)[MAXC]) {
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) { v[c] = x[c]; mx = fmaxf(mx, v[c]); }
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(v[c] - mx);
    return mx + logf(s);
}

// first index attaining the row maximum, NaN counting as the largest value (torch.argmax)
__device__ __forceinline__ int row_argmax(const float (&v)[MAXC], int C) {
    float best = v[0];
    int bi = 0;
    for (int c = 1; c < C; ++c)
        if (v[c] > best || (v[c] != v[c] && best == best)) { best = v[c]; bi = c; }
    return bi;
}

__device__ __forceinline__ double block_sum(double (&sh)[TB], double acc) {
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = TB / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// partial[b] = loss sum of block b; COUNT: partial[CE_BLOCKS + b] = rows of block b whose argmax is the label
template <bool COUNT>
__global__ void __launch_bounds__(TB)
k_ce_fwd(const float* __restrict__ x, int64_t ldx, const int64_t* __restrict__ y, int64_t N, int C,
         double* __restrict__ partial, const int64_t* __restrict__ n_valid) {
    __shared__ double sh[TB];
    double acc = 0.0, hit = 0.0;
    float v[MAXC];
    if (n_valid) N = min(N, max(*n_valid, (int64_t)0));          // rows from *n_valid on are padding: no loss, no count
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < N; i += (int64_t)gridDim.x * TB) {
        const float lse = row_lse(x + i * ldx, C, v);
        acc += (double)(lse - x[i * ldx + y[i]]);
        if (COUNT) hit += row_argmax(v, C) == (int)y[i] ? 1.0 : 0.0;
    }
    const double a = block_sum(sh, acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = a;
    if (COUNT) {
        const double h = block_sum(sh, hit);
        if (threadIdx.x == 0) partial[CE_BLOCKS + blockIdx.x] = h;
    }
}

__global__ void __launch_bounds__(TB)
k_ce_final(const double* __restrict__ partial, int n, int64_t N, float* __restrict__ loss, double* __restrict__ stats,
           const int64_t* __restrict__ n_valid) {
    __shared__ double sh[TB];
    if (n_valid) N = min(N, max(*n_valid, (int64_t)0));
    double a = 0.0, h = 0.0;
    for (int k = threadIdx.x; k < n; k += TB) a += partial[k];
    if (stats)
        for (int k = threadIdx.x; k < n; k += TB) h += partial[CE_BLOCKS + k];
    a = block_sum(sh, a);
    if (stats) h = block_sum(sh, h);
    if (threadIdx.x == 0) {
        *loss = (float)(a / (double)N);
        if (stats) { stats[0] = (double)*loss; stats[1] = h; }
    }
}

__global__ void __launch_bounds__(TB)
k_ce_bwd(const float* __restrict__ x, int64_t ldx, const int64_t* __restrict__ y, int64_t N, int C,
         const float* __restrict__ grad_loss, float* __restrict__ gx, int64_t ldg, const int64_t* __restrict__ n_valid) {
    const int64_t NV = n_valid ? min(N, max(*n_valid, (int64_t)0)) : N;
    const float scale = *grad_loss / (float)NV;
    float v[MAXC];
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < N; i += (int64_t)gridDim.x * TB) {
        if (i >= NV) {                                               // padding rows: exact zeros
            for (int c = 0; c < C; ++c) gx[i * ldg + c] = 0.f;
            continue;
        }
        const float lse = row_lse(x + i * ldx, C, v);
        const int64_t yi = y[i];
        for (int c = 0; c < C; ++c)
            gx[i * ldg + c] = (expf(v[c] - lse) - (c == yi ? 1.f : 0.f)) * scale;
    }
}

// ---- mean entropy of the clamped softmax (UDAGCN's target term, pygda/models/udagcn.py:193-197) --------------------------
// p = clamp(softmax(z), lo, 1);  loss = mean_i sum_c -p_ic log p_ic.  One row per thread, the probabilities in registers.
__device__ __forceinline__ void row_probs(const float* __restrict__ x, int C, float (&p)[MAXC]) {
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) { p[c] = x[c]; mx = fmaxf(mx, p[c]); }
    float s = 0.f;
    for (int c = 0; c < C; ++c) { p[c] = expf(p[c] - mx); s += p[c]; }
    for (int c = 0; c < C; ++c) p[c] = p[c] / s;
}

__global__ void __launch_bounds__(TB)
k_entropy_fwd(const float* __restrict__ x, int64_t ldx, int64_t N, int C, float lo, double* __restrict__ partial) {
    __shared__ double sh[TB];
    double acc = 0.0;
    float p[MAXC];
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < N; i += (int64_t)gridDim.x * TB) {
        row_probs(x + i * ldx, C, p);
        float e = 0.f;
        for (int c = 0; c < C; ++c) {
            const float q = fminf(fmaxf(p[c], lo), 1.f);
            e += -q * logf(q);
        }
        acc += (double)e;
    }
    const double a = block_sum(sh, acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = a;
}

// d loss / d z_j = p_j (g_j - sum_c g_c p_c) / N,  g_c = -(log q_c + 1) where the clamp passes (lo <= p_c <= 1), else 0
__global__ void __launch_bounds__(TB)
k_entropy_bwd(const float* __restrict__ x, int64_t ldx, int64_t N, int C, float lo, const float* __restrict__ grad_loss,
              float* __restrict__ gx, int64_t ldg) {
    const float scale = *grad_loss / (float)N;
    float p[MAXC], g[MAXC];
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < N; i += (int64_t)gridDim.x * TB) {
        row_probs(x + i * ldx, C, p);
        float dot = 0.f;
        for (int c = 0; c < C; ++c) {
            const bool pass = p[c] >= lo && p[c] <= 1.f;
            g[c] = pass ? -(logf(p[c]) + 1.f) : 0.f;
            dot += g[c] * p[c];
        }
        for (int c = 0; c < C; ++c) gx[i * ldg + c] = p[c] * (g[c] - dot) * scale;
    }
}

}  // namespace

extern "C" int gda_softmax_entropy_fwd_f32(const float* logits, int64_t ld, int64_t N, int C, float clamp_min, float* loss,
                                           void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    if (N <= 0 || C <= 0 || ld < C || !(clamp_min > 0.f && clamp_min < 1.f)) return GDA_E_SIZE;
    if (C > MAXC) return GDA_E_UNSUPPORTED;
    if (!logits || !loss || !workspace) return GDA_E_NULL;
    if (workspace_bytes < 2 * CE_BLOCKS * sizeof(double)) return GDA_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const int blocks = (int)min((int64_t)CE_BLOCKS, gda_cdiv(N, TB));
    double* partial = static_cast<double*>(workspace);
    k_entropy_fwd<<<blocks, TB, 0, stream>>>(logits, ld, N, C, clamp_min, partial);
    GDA_LAUNCH_CHECK();
    k_ce_final<<<1, TB, 0, stream>>>(partial, blocks, N, loss, nullptr, nullptr);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_softmax_entropy_bwd_f32(const float* logits, int64_t ld, int64_t N, int C, float clamp_min,
                                           const float* grad_loss, float* grad_logits, int64_t ldg, gda_stream_t stream_) {
    if (N <= 0 || C <= 0 || ld < C || ldg < C || !(clamp_min > 0.f && clamp_min < 1.f)) return GDA_E_SIZE;
    if (C > MAXC) return GDA_E_UNSUPPORTED;
    if (!logits || !grad_loss || !grad_logits) return GDA_E_NULL;
    hipStream_t stream = (hipStream_t)stream_;
    const int blocks = (int)min((int64_t)1024, gda_cdiv(N, TB));
    k_entropy_bwd<<<blocks, TB, 0, stream>>>(logits, ld, N, C, clamp_min, grad_loss, grad_logits, ldg);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" size_t gda_softmax_nll_workspace_bytes(void) { return 2 * CE_BLOCKS * sizeof(double); }

// n_valid (device int64[1], or NULL = all N rows): only rows [0, *n_valid) are real -- the loss is their mean, the count is
// theirs, and the backward pass writes exact zeros into the rows behind them.  For batches padded to a static capacity
// (the captured sampled step, pygda_amd/sampled_graph.py): the row count is read on the device, the launch stays the same.
extern "C" int gda_softmax_nll_fwd_nv_f32(const float* logits, int64_t ld, const int64_t* labels, int64_t N, int C,
                                          const int64_t* n_valid, float* loss, double* stats, void* workspace,
                                          size_t workspace_bytes, gda_stream_t stream_) {
    if (N < 0 || C < 1 || C > MAXC || ld < C) return C > MAXC ? GDA_E_UNSUPPORTED : GDA_E_SIZE;
    if (!loss || !workspace || (N > 0 && (!logits || !labels))) return GDA_E_NULL;
    if (workspace_bytes < gda_softmax_nll_workspace_bytes()) return GDA_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    if (N == 0) {                                             // mean over nothing: nan, as torch
        const float nanv = __builtin_nanf("");
        GDA_HIP_TRY(hipMemcpyAsync(loss, &nanv, sizeof(float), hipMemcpyHostToDevice, stream));
        if (stats) {
            const double st[2] = {(double)nanv, 0.0};
            GDA_HIP_TRY(hipMemcpyAsync(stats, st, sizeof(st), hipMemcpyHostToDevice, stream));
        }
        return GDA_OK;
    }
    const int blocks = (int)(gda_cdiv(N, TB) < CE_BLOCKS ? gda_cdiv(N, TB) : CE_BLOCKS);
    if (stats) k_ce_fwd<true><<<blocks, TB, 0, stream>>>(logits, ld, labels, N, C, (double*)workspace, n_valid);
    else k_ce_fwd<false><<<blocks, TB, 0, stream>>>(logits, ld, labels, N, C, (double*)workspace, n_valid);
    GDA_LAUNCH_CHECK();
    GDA_UNLESS_SKIPPED("k_ce_final") k_ce_final<<<1, TB, 0, stream>>>((const double*)workspace, blocks, N, loss, stats, n_valid);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_softmax_nll_fwd_ex_f32(const float* logits, int64_t ld, const int64_t* labels, int64_t N, int C,
                                          float* loss, double* stats, void* workspace, size_t workspace_bytes,
                                          gda_stream_t stream_) {
    return gda_softmax_nll_fwd_nv_f32(logits, ld, labels, N, C, nullptr, loss, stats, workspace, workspace_bytes, stream_);
}

extern "C" int gda_softmax_nll_fwd_f32(const float* logits, int64_t ld, const int64_t* labels, int64_t N, int C,
                                       float* loss, void* workspace, size_t workspace_bytes,
                                       gda_stream_t stream_) {
    return gda_softmax_nll_fwd_ex_f32(logits, ld, labels, N, C, loss, nullptr, workspace, workspace_bytes, stream_);
}

extern "C" int gda_softmax_nll_bwd_nv_f32(const float* logits, int64_t ld, const int64_t* labels, int64_t N, int C,
                                          const int64_t* n_valid, const float* grad_loss, float* grad_logits, int64_t ldg,
                                          gda_stream_t stream_) {
    if (N < 0 || C < 1 || C > MAXC || ld < C || ldg < C) return C > MAXC ? GDA_E_UNSUPPORTED : GDA_E_SIZE;
    if (N == 0) return GDA_OK;
    if (!logits || !labels || !grad_loss || !grad_logits) return GDA_E_NULL;
    if (grad_logits == logits) return GDA_E_ALIAS;
    const int64_t blocks = gda_cdiv(N, TB) < 4096 ? gda_cdiv(N, TB) : 4096;
    k_ce_bwd<<<(unsigned)blocks, TB, 0, (hipStream_t)stream_>>>(logits, ld, labels, N, C, grad_loss, grad_logits, ldg, n_valid);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_softmax_nll_bwd_f32(const float* logits, int64_t ld, const int64_t* labels, int64_t N, int C,
                                       const float* grad_loss, float* grad_logits, int64_t ldg,
                                       gda_stream_t stream_) {
    return gda_softmax_nll_bwd_nv_f32(logits, ld, labels, N, C, nullptr, grad_loss, grad_logits, ldg, stream_);
}
