// Wasserstein critic with gradient penalty (WGAN-GP), one critic update's loss and parameter gradients in
// closed form, gfx950.
//
// Replaces the critic loop body of pygda/models/adagcn.py:169-183 with gradient_penalty (:387-454) for the
// critic the trainer builds at :264-270,
//     D(x) = sigmoid(w2 . drop_p(relu(W1 x + b1)) + b2),      W1 [a, h], w2 [a]   (h = hidden, a = adv_dim)
//     L = -| mean_s D(e_s) - mean_t D(e_t) |  +  gp_weight * mean_i (|| grad_x D(x_i) ||_2 - 1)^2
// over x_i in cat(e_s, e_t, interpolates), interpolates_i = e_t[it_i] + alpha_i (e_s[is_i] - e_t[it_i]).
// The reference differentiates twice through torch autograd (create_graph=True): ~60 kernels and a dozen
// rocBLAS GEMMs of [28k x 128 x 40] per update.  For this two-layer critic everything is closed form per row:
//     a = W1 x + b1, r = [a > 0], m = dropout mask / (1 - p), hid = m r a, z = w2.hid + b2, s = sigmoid(z),
//     s' = s (1 - s), u = m r w2, v = W1^T u, grad_x D = s' v, nrm = s' |v|
//     d nrm = nv s'(1 - 2 s) dz + s' d|v|,   dz = hid.dw2 + db2 + u.(dW1 x + db1),   d|v| = vhat.(dW1^T u) + (m r (W1 vhat)).dw2
// so a row contributes  u (cA x + cB vhat)^T  to gW1 and  cA u  to gb1 (cA, cB scalars): the kernel writes
// U [R, a] and Y [R, h + 4] (column h = cA) and ONE matrix-core product gW1|gb1 = U^T Y (gda_gemm_f32, TN,
// deterministic row-slab split) finishes them; gw2, gb2 and the loss are fixed-order block sums.
//   k_critic_gap    D on the gap rows -> sums per domain (the sign of the gap scales those rows' gradients)
//   k_critic_rows   every row (gap rows with their own masks, then the penalty rows): U, Y, partial sums
//   gda_gemm_f32    U^T Y
//   k_critic_final  gw2, gb2, loss; gW1 / gb1 unpacked
// One wavefront per row: lane k owns hidden unit k (a <= 64) for the W1 x and W1 vhat products, lanes own
// columns j, j + 64, ... for v = W1^T u; W1 sits in LDS with a padded leading dimension (conflict free both ways).
#include "gda_common.h"
#include "gda_philox.h"

namespace {

constexpr int TB = 256;
constexpr int WAVES = TB / 64;
constexpr int HMAX = 256;          // input width limit (4 columns per lane)
constexpr int AMAX = 64;           // critic hidden width limit (one lane per unit)

struct Critic {
    const float* W1; const float* b1; const float* w2; const float* b2;
    int h, a;
};

struct RowsIn {
    const float* es; int64_t n_s;
    const float* et; int64_t n_t;
    const int32_t* is; const int32_t* it; const float* alpha; int64_t n_i;     // interpolates
};

struct Drop {
    float p; uint64_t seed; const int64_t* step; uint32_t site;
};

// LDS hand-off between the lanes of ONE wavefront: its DS operations retire in order; this only keeps the
// compiler from moving the dependent reads above the writes
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// keep-factor of hidden unit k of row `row` at call site `site`: 0 or 1/(1-p)
__device__ __forceinline__ float keep_factor(const Drop& dr, uint64_t st, uint32_t site, int64_t row, int a, int k) {
    if (dr.p <= 0.f) return 1.f;
    const uint64_t e = (uint64_t)row * (uint64_t)a + (uint64_t)k;
    uint32_t r[4];
    GdaPhilox::gen(dr.seed, (st << 20) ^ site, e >> 2, r);
    const uint32_t thresh = (uint32_t)((double)dr.p * 4294967296.0 > 4294967295.0 ? 4294967295.0 : (double)dr.p * 4294967296.0);
    return r[e & 3] >= thresh ? 1.f / (1.f - dr.p) : 0.f;
}

// x of penalty row g (cat(e_s, e_t, interpolates)), column j
__device__ __forceinline__ float gp_x(const RowsIn& R, int h, int64_t g, int j) {
    if (g < R.n_s) return R.es[g * h + j];
    if (g < R.n_s + R.n_t) return R.et[(g - R.n_s) * h + j];
    const int64_t i = g - R.n_s - R.n_t;
    const float t = R.et[(int64_t)R.it[i] * h + j], s = R.es[(int64_t)R.is[i] * h + j];
    return t + R.alpha[i] * (s - t);                                   // adagcn.py:438
}

struct Shared {
    float* W1s;      // [a][h + 1]
    float* xs;       // [WAVES][h]
    float* us;       // [WAVES][AMAX]
    float* vs;       // [WAVES][h]
};

__device__ __forceinline__ Shared carve_lds(float* base, int h, int a) {
    Shared s;
    s.W1s = base;
    s.xs = s.W1s + (size_t)a * (h + 1);
    s.us = s.xs + (size_t)WAVES * h;
    s.vs = s.us + (size_t)WAVES * AMAX;
    return s;
}

__host__ __device__ inline size_t lds_floats(int h, int a) {
    return (size_t)a * (h + 1) + (size_t)WAVES * h + (size_t)WAVES * AMAX + (size_t)WAVES * h;
}

__device__ __forceinline__ void load_w1(const Critic& C, float* W1s) {
    for (int e = threadIdx.x; e < C.a * C.h; e += TB) W1s[(e / C.h) * (C.h + 1) + e % C.h] = C.W1[e];
}

// lane k: dot of row k of W1 with the wave's LDS vector `vec`
__device__ __forceinline__ float w1_row_dot(const float* W1s, const float* vec, int h, int a, int lane) {
    float acc = 0.f;
    if (lane < a) {
        const float* w = W1s + (size_t)lane * (h + 1);
#pragma unroll 8
        for (int j = 0; j < h; ++j) acc = fmaf(w[j], vec[j], acc);
    }
    return acc;
}

// forward of one row up to sigmoid: lane k keeps its unit's hid = m r a, mr = m r and u = m r w2; returns s (all lanes)
__device__ __forceinline__ float critic_row_fwd(const Critic& C, const float* W1s, const float* xs, int lane,
                                                float keep, float& hid, float& u, float& mr) {
    wave_lds_sync();                                     // xs was written by the other lanes
    float ak = w1_row_dot(W1s, xs, C.h, C.a, lane);
    hid = 0.f; u = 0.f; mr = 0.f;
    float zk = 0.f;
    if (lane < C.a) {
        ak += C.b1[lane];
        const float w2k = C.w2[lane];
        mr = ak > 0.f ? keep : 0.f;
        hid = mr * ak;                                   // drop(relu(a))
        u = mr * w2k;                                    // d z / d a_k
        zk = w2k * hid;
    }
    const float z = wave_sum(zk) + C.b2[0];
    return 1.f / (1.f + __expf(-z));
}

__global__ void __launch_bounds__(TB)
k_critic_gap(Critic C, RowsIn R, Drop dr, double* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const Shared S = carve_lds(lds, C.h, C.a);
    load_w1(C, S.W1s);
    __syncthreads();
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    float* xs = S.xs + (size_t)wave * C.h;
    const uint64_t st = dr.p > 0.f ? (uint64_t)dr.step[0] : 0;
    const int64_t rows = R.n_s + R.n_t;
    double sum_s = 0.0, sum_t = 0.0;
    for (int64_t r = (int64_t)blockIdx.x * WAVES + wave; r < rows; r += (int64_t)gridDim.x * WAVES) {
        const bool src = r < R.n_s;
        const float* x = src ? R.es + r * C.h : R.et + (r - R.n_s) * C.h;
        for (int j = lane; j < C.h; j += 64) xs[j] = x[j];
        const float keep = lane < C.a ? keep_factor(dr, st, dr.site + (src ? 0u : 1u), src ? r : r - R.n_s, C.a, lane) : 0.f;
        float hid, u, mr;
        const float s = critic_row_fwd(C, S.W1s, xs, lane, keep, hid, u, mr);
        if (src) sum_s += (double)s; else sum_t += (double)s;
        wave_lds_sync();                                 // xs is overwritten by the next row
    }
    __shared__ double red[WAVES][2];
    if (lane == 0) { red[wave][0] = sum_s; red[wave][1] = sum_t; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a0 = 0.0, a1 = 0.0;
        for (int w = 0; w < WAVES; ++w) { a0 += red[w][0]; a1 += red[w][1]; }
        part[(int64_t)blockIdx.x * 2 + 0] = a0;
        part[(int64_t)blockIdx.x * 2 + 1] = a1;
    }
}

// rows 0 .. n_s+n_t-1: gap rows; then the n_s + n_t + n_i penalty rows.
// part_rows[block][a + 2]: sum of the w2 / b2 gradient terms, then the block's sum of (nrm - 1)^2
__global__ void __launch_bounds__(TB)
k_critic_rows(Critic C, RowsIn R, Drop dr, float gp_weight, const double* __restrict__ gap_part, int gap_blocks,
              float* __restrict__ U, float* __restrict__ Y, int ldy, double* __restrict__ part_rows) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const Shared S = carve_lds(lds, C.h, C.a);
    load_w1(C, S.W1s);
    __shared__ float sign_sh;
    if (threadIdx.x == 0) {
        double a0 = 0.0, a1 = 0.0;
        for (int b = 0; b < gap_blocks; ++b) { a0 += gap_part[2 * b]; a1 += gap_part[2 * b + 1]; }
        const double gap = a0 / (double)R.n_s - a1 / (double)R.n_t;
        sign_sh = gap > 0.0 ? 1.f : (gap < 0.0 ? -1.f : 0.f);
    }
    __syncthreads();
    const float sgn = sign_sh;
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    float* xs = S.xs + (size_t)wave * C.h;
    float* us = S.us + (size_t)wave * AMAX;
    float* vs = S.vs + (size_t)wave * C.h;
    const uint64_t st = dr.p > 0.f ? (uint64_t)dr.step[0] : 0;
    const int64_t n_gap = R.n_s + R.n_t, m_gp = R.n_s + R.n_t + R.n_i, rows = n_gap + m_gp;
    double acc_w2 = 0.0, acc_b2 = 0.0, acc_gp = 0.0;              // lane k: its w2 term; lane 0: b2 and penalty
    for (int64_t r = (int64_t)blockIdx.x * WAVES + wave; r < rows; r += (int64_t)gridDim.x * WAVES) {
        const bool is_gap = r < n_gap;
        float keep = 0.f;
        if (is_gap) {
            const bool src = r < R.n_s;
            const float* x = src ? R.es + r * C.h : R.et + (r - R.n_s) * C.h;
            for (int j = lane; j < C.h; j += 64) xs[j] = x[j];
            if (lane < C.a) keep = keep_factor(dr, st, dr.site + (src ? 0u : 1u), src ? r : r - R.n_s, C.a, lane);
        } else {
            const int64_t g = r - n_gap;
            for (int j = lane; j < C.h; j += 64) xs[j] = gp_x(R, C.h, g, j);
            if (lane < C.a) keep = keep_factor(dr, st, dr.site + 2u, g, C.a, lane);
        }
        float hid, u, mr;
        const float s = critic_row_fwd(C, S.W1s, xs, lane, keep, hid, u, mr);
        const float sp = s * (1.f - s);
        float cA, cB = 0.f, wk = 0.f;
        if (is_gap) {
            // d(-|gap|) / d D_i = -sign / n_s (source rows), +sign / n_t (target rows)
            cA = (r < R.n_s ? -sgn / (float)R.n_s : sgn / (float)R.n_t) * sp;
        } else {
            if (lane < C.a) us[lane] = u;
            wave_lds_sync();
            // v = W1^T u: lanes own columns j, j + 64, ...
            float nv2 = 0.f;
            for (int j = lane; j < C.h; j += 64) {
                float v = 0.f;
                for (int k = 0; k < C.a; ++k) v = fmaf(S.W1s[(size_t)k * (C.h + 1) + j], us[k], v);
                vs[j] = v;
                nv2 = fmaf(v, v, nv2);
            }
            nv2 = wave_sum(nv2);
            const float nv = sqrtf(nv2);
            const float inv = nv > 0.f ? 1.f / nv : 0.f;
            for (int j = lane; j < C.h; j += 64) vs[j] *= inv;          // vhat (each lane rescales what it wrote)
            wave_lds_sync();
            const float nrm = sp * nv;
            const float e = gp_weight / (float)m_gp * 2.f * (nrm - 1.f);
            cA = e * nv * sp * (1.f - 2.f * s);
            cB = e * sp;
            wk = w1_row_dot(S.W1s, vs, C.h, C.a, lane);              // (W1 vhat)_k
            if (lane == 0) acc_gp += (double)((nrm - 1.f) * (nrm - 1.f));
        }
        // gradient terms of this row
        float* yrow = Y + r * (int64_t)ldy;
        for (int j = lane; j < C.h; j += 64) yrow[j] = cA * xs[j] + (is_gap ? 0.f : cB * vs[j]);
        if (lane == 0) yrow[C.h] = cA;
        if (lane < C.a) {
            U[r * (int64_t)C.a + lane] = u;
            acc_w2 += (double)(cA * hid + cB * mr * wk);
        }
        if (lane == 0) acc_b2 += (double)cA;
        wave_lds_sync();                                 // xs / us / vs are overwritten by the next row
    }
    __shared__ double red[WAVES][AMAX + 2];
    if (lane < C.a) red[wave][lane] = acc_w2;
    if (lane == 0) { red[wave][AMAX] = acc_b2; red[wave][AMAX + 1] = acc_gp; }
    __syncthreads();
    double* out = part_rows + (int64_t)blockIdx.x * (C.a + 2);
    if ((int)threadIdx.x < C.a) {
        double v = 0.0;
        for (int w = 0; w < WAVES; ++w) v += red[w][threadIdx.x];
        out[threadIdx.x] = v;
    } else if ((int)threadIdx.x == C.a || (int)threadIdx.x == C.a + 1) {
        const int q = threadIdx.x - C.a;
        double v = 0.0;
        for (int w = 0; w < WAVES; ++w) v += red[w][AMAX + q];
        out[C.a + q] = v;
    }
}

__global__ void __launch_bounds__(TB)
k_critic_final(Critic C, RowsIn R, float gp_weight, const double* __restrict__ gap_part, int gap_blocks,
               const double* __restrict__ part_rows, int row_blocks, const float* __restrict__ UtY, int ldc,
               float* __restrict__ loss, float* __restrict__ gW1, float* __restrict__ gb1, float* __restrict__ gw2,
               float* __restrict__ gb2) {
    const int t = threadIdx.x;
    if (blockIdx.x == 0) {
        if (t < C.a + 2) {
            double v = 0.0;
            for (int b = 0; b < row_blocks; ++b) v += part_rows[(int64_t)b * (C.a + 2) + t];
            if (t < C.a) gw2[t] = (float)v;
            else if (t == C.a) gb2[0] = (float)v;
            else {
                double a0 = 0.0, a1 = 0.0;
                for (int b = 0; b < gap_blocks; ++b) { a0 += gap_part[2 * b]; a1 += gap_part[2 * b + 1]; }
                const double gap = a0 / (double)R.n_s - a1 / (double)R.n_t;
                const double m_gp = (double)(R.n_s + R.n_t + R.n_i);
                loss[0] = (float)(-(gap < 0 ? -gap : gap) + (double)gp_weight * v / m_gp);       // adagcn.py:177
            }
        }
        if (t < C.a) gb1[t] = UtY[(int64_t)t * ldc + C.h];
    }
    for (int64_t e = (int64_t)blockIdx.x * TB + t; e < (int64_t)C.a * C.h; e += (int64_t)gridDim.x * TB)
        gW1[e] = UtY[(e / C.h) * ldc + e % C.h];
}

struct Ws { double* gap_part; double* part_rows; float* U; float* Y; float* UtY; void* gemm_ws; size_t gemm_bytes; size_t total; };

constexpr int GAP_BLOCKS = 128, ROW_BLOCKS = 512;

Ws carve(void* base, int64_t n_s, int64_t n_t, int64_t n_i, int h, int a) {
    const int64_t rows = 2 * (n_s + n_t) + n_i;
    const int ldy = h + 4;
    Ws w{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void* p = base ? (void*)((char*)base + off) : nullptr;
        off += gda_align_up(bytes, 256);
        return p;
    };
    w.gap_part = (double*)take(sizeof(double) * GAP_BLOCKS * 2);
    w.part_rows = (double*)take(sizeof(double) * ROW_BLOCKS * (a + 2));
    w.U = (float*)take(sizeof(float) * rows * a);
    w.Y = (float*)take(sizeof(float) * rows * ldy);
    w.UtY = (float*)take(sizeof(float) * a * ldy);
    w.gemm_bytes = gda_gemm_workspace_bytes(GDA_GEMM_TN, a, ldy, rows);
    w.gemm_ws = take(w.gemm_bytes);
    w.total = off;
    return w;
}

}  // namespace

extern "C" size_t gda_wgan_critic_workspace_bytes(int64_t n_s, int64_t n_t, int64_t n_i, int h, int a) {
    if (n_s <= 0 || n_t <= 0 || n_i < 0 || h <= 0 || a <= 0) return 0;
    return carve(nullptr, n_s, n_t, n_i, h, a).total;
}

extern "C" int gda_wgan_critic_f32(const float* es, int64_t n_s, const float* et, int64_t n_t, int h,
                                   const int32_t* idx_s, const int32_t* idx_t, const float* alpha, int64_t n_i,
                                   const float* W1, const float* b1, const float* w2, const float* b2, int a,
                                   float dropout_p, uint64_t seed, const int64_t* step, uint32_t site,
                                   float gp_weight, float* loss, float* gW1, float* gb1, float* gw2, float* gb2,
                                   void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    if (n_s <= 0 || n_t <= 0 || n_i < 0 || h <= 0 || a <= 0 || h > HMAX || a > AMAX || h % 4 != 0) return GDA_E_SIZE;
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) return GDA_E_SIZE;
    if (2 * (n_s + n_t) + n_i >= INT32_MAX / 2) return GDA_E_SIZE;
    if (!es || !et || !W1 || !b1 || !w2 || !b2 || !loss || !gW1 || !gb1 || !gw2 || !gb2 || !workspace) return GDA_E_NULL;
    if (n_i > 0 && (!idx_s || !idx_t || !alpha)) return GDA_E_NULL;
    if (dropout_p > 0.f && !step) return GDA_E_NULL;
    const Ws ws = carve(workspace, n_s, n_t, n_i, h, a);
    if (workspace_bytes < ws.total) return GDA_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const Critic C{W1, b1, w2, b2, h, a};
    const RowsIn R{es, n_s, et, n_t, idx_s, idx_t, alpha, n_i};
    const Drop dr{dropout_p, seed, step, site};
    const size_t lds = lds_floats(h, a) * sizeof(float);
    if (lds > 48 * 1024) {                         // the widest legal critic needs 75 KB of dynamic LDS
        GDA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_critic_gap),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        GDA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_critic_rows),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const int64_t rows = 2 * (n_s + n_t) + n_i;
    const int ldy = h + 4;
    k_critic_gap<<<GAP_BLOCKS, TB, lds, stream>>>(C, R, dr, ws.gap_part);
    GDA_LAUNCH_CHECK();
    k_critic_rows<<<ROW_BLOCKS, TB, lds, stream>>>(C, R, dr, gp_weight, ws.gap_part, GAP_BLOCKS, ws.U, ws.Y, ldy, ws.part_rows);
    GDA_LAUNCH_CHECK();
    // columns h+1 .. h+3 of Y are padding of the 16-byte row stride: never written, multiplied into columns of
    // UtY that nobody reads
    int st = gda_gemm_f32(GDA_GEMM_TN, a, ldy, rows, ws.U, a, ws.Y, ldy, ws.UtY, ldy, ws.gemm_ws, ws.gemm_bytes, stream_);
    if (st != GDA_OK) return st;
    k_critic_final<<<8, TB, 0, stream>>>(C, R, gp_weight, ws.gap_part, GAP_BLOCKS, ws.part_rows, ROW_BLOCKS, ws.UtY, ldy,
                                         loss, gW1, gb1, gw2, gb2);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}
