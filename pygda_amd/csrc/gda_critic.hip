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
//   k_critic_rows   every row (gap rows with their own masks, carrying the UNSIGNED derivative of the gap, then
//                   the penalty rows): U, Y, fixed-order block sums (w2 / b2 terms, penalty, D per domain)
//   gda_gemm_f32    U^T Y for the gap rows and for the penalty rows
//   k_critic_final  sign(gap) from the block sums; gW1, gb1, gw2, gb2 = -sign * gap part + penalty part; loss
// One wavefront takes RB = 4 rows: lane k owns hidden unit k (a <= 64) for the W1 x and W1 vhat products, lanes
// own columns j, j + 64, ... for v = W1^T u; W1 sits in LDS with a padded leading dimension (conflict free both
// ways), every W1 word read feeds the four rows, the rows' values are broadcast with v_readlane.
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

constexpr int RB = 4;              // rows per wavefront: every W1 word read from LDS feeds RB rows

__device__ __forceinline__ float bcast(float v, int src_lane) {      // src_lane is wave-uniform
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

__host__ __device__ inline size_t lds_floats(int h, int a) { return (size_t)a * (h + 1); }

__device__ __forceinline__ void load_w1(const Critic& C, float* W1s) {
    for (int e = threadIdx.x; e < C.a * C.h; e += TB) W1s[(e / C.h) * (C.h + 1) + e % C.h] = C.W1[e];
}

// lane k (k < a): out[q] = sum_j W1[k][j] * vec_q[j], vec_q distributed over the lanes (lane l holds columns
// l, l + 64, ...): one conflict-free LDS read of W1[k][j] per j, broadcast of the RB row values by v_readlane
template <int HC>
__device__ __forceinline__ void w1_rows_dot(const float* W1s, const float (&vec)[RB][HC], int h, int a, int lane,
                                            float (&out)[RB]) {
#pragma unroll
    for (int q = 0; q < RB; ++q) out[q] = 0.f;
    const float* w = W1s + (size_t)(lane < a ? lane : 0) * (h + 1);
#pragma unroll
    for (int c = 0; c < HC; ++c) {
        if (c * 64 >= h) break;
        const int cols = h - c * 64 < 64 ? h - c * 64 : 64;           // h is a multiple of 4
#pragma unroll 4
        for (int jj = 0; jj < cols; ++jj) {
            const float wv = w[c * 64 + jj];
#pragma unroll
            for (int q = 0; q < RB; ++q) out[q] = fmaf(wv, bcast(vec[q][c], jj), out[q]);
        }
    }
}

// part_rows[2a + 5][blocks]: gap rows' w2 terms [a] and b2 term, penalty rows' w2 terms [a] and b2 term, sum of
// (nrm - 1)^2, sum of D over the source gap rows, sum of D over the target gap rows
// Row space: [0, n_gap) gap rows (e_s then e_t, their own masks), [n_gap, n_gap + m_gp) penalty rows.
// Gap rows carry the UNSIGNED derivative of gap = mean_s D - mean_t D; k_critic_final applies -sign(gap).
template <int HC>
__global__ void __launch_bounds__(TB)
k_critic_rows(Critic C, RowsIn R, Drop dr, float gp_weight, float* __restrict__ U, float* __restrict__ Y, int ldy,
              double* __restrict__ part_rows) {
    extern __shared__ __attribute__((aligned(16))) float W1s[];
    load_w1(C, W1s);
    __syncthreads();
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    const uint64_t st = dr.p > 0.f ? (uint64_t)dr.step[0] : 0;
    const int h = C.h, a = C.a;
    const int64_t n_gap = R.n_s + R.n_t, m_gp = R.n_s + R.n_t + R.n_i;
    const int64_t g_gap = (n_gap + RB - 1) / RB, g_gp = (m_gp + RB - 1) / RB;
    const float b1k = lane < a ? C.b1[lane] : 0.f, w2k = lane < a ? C.w2[lane] : 0.f, b2 = C.b2[0];
    double acc_w2[2] = {0.0, 0.0}, acc_b2[2] = {0.0, 0.0}, acc_gp = 0.0, acc_ds = 0.0, acc_dt = 0.0;
    for (int64_t grp = (int64_t)blockIdx.x * WAVES + wave; grp < g_gap + g_gp; grp += (int64_t)gridDim.x * WAVES) {
        const bool is_gap = grp < g_gap;
        const int64_t base = is_gap ? grp * RB : (grp - g_gap) * RB;       // first row of the group in its space
        const int64_t limit = is_gap ? n_gap : m_gp;
        float x[RB][HC], keep[RB];
        bool live[RB];
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int64_t r = base + q;
            live[q] = r < limit;
            keep[q] = 0.f;
#pragma unroll
            for (int c = 0; c < HC; ++c) {
                const int j = c * 64 + lane;
                float v = 0.f;
                if (live[q] && j < h) {
                    if (is_gap) v = r < R.n_s ? R.es[r * h + j] : R.et[(r - R.n_s) * h + j];
                    else v = gp_x(R, h, r, j);
                }
                x[q][c] = v;
            }
            if (live[q] && lane < a) {
                if (is_gap) keep[q] = keep_factor(dr, st, dr.site + (r < R.n_s ? 0u : 1u), r < R.n_s ? r : r - R.n_s, a, lane);
                else keep[q] = keep_factor(dr, st, dr.site + 2u, r, a, lane);
            }
        }
        float ak[RB];
        w1_rows_dot<HC>(W1s, x, h, a, lane, ak);
        float hid[RB], u[RB], mr[RB], s[RB], sp[RB];
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const float av = ak[q] + b1k;
            mr[q] = (lane < a && av > 0.f) ? keep[q] : 0.f;
            hid[q] = mr[q] * av;                                     // drop(relu(a))
            u[q] = mr[q] * w2k;                                      // d z / d a_k
            const float z = wave_sum(w2k * hid[q]) + b2;
            s[q] = 1.f / (1.f + __expf(-z));
            sp[q] = s[q] * (1.f - s[q]);
        }
        float cA[RB], cB[RB], wk[RB], vh[RB][HC];
#pragma unroll
        for (int q = 0; q < RB; ++q) { cB[q] = 0.f; wk[q] = 0.f; }
        if (is_gap) {
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int64_t r = base + q;
                cA[q] = live[q] ? (r < R.n_s ? sp[q] / (float)R.n_s : -sp[q] / (float)R.n_t) : 0.f;   // d gap / d z_i
                if (live[q] && lane == 0) { if (r < R.n_s) acc_ds += (double)s[q]; else acc_dt += (double)s[q]; }
#pragma unroll
                for (int c = 0; c < HC; ++c) vh[q][c] = 0.f;
            }
        } else {
            // v = W1^T u: lane l owns columns l, l + 64, ...; u_k broadcast from lane k
#pragma unroll
            for (int q = 0; q < RB; ++q)
#pragma unroll
                for (int c = 0; c < HC; ++c) vh[q][c] = 0.f;
            for (int k = 0; k < a; ++k) {
                float uk[RB];
#pragma unroll
                for (int q = 0; q < RB; ++q) uk[q] = bcast(u[q], k);
#pragma unroll
                for (int c = 0; c < HC; ++c) {
                    const int j = c * 64 + lane;
                    if (j < h) {
                        const float wv = W1s[(size_t)k * (h + 1) + j];
#pragma unroll
                        for (int q = 0; q < RB; ++q) vh[q][c] = fmaf(wv, uk[q], vh[q][c]);
                    }
                }
            }
            float nv[RB];
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                float n2 = 0.f;
#pragma unroll
                for (int c = 0; c < HC; ++c) n2 = fmaf(vh[q][c], vh[q][c], n2);
                n2 = wave_sum(n2);
                nv[q] = sqrtf(n2);
                const float inv = nv[q] > 0.f ? 1.f / nv[q] : 0.f;
#pragma unroll
                for (int c = 0; c < HC; ++c) vh[q][c] *= inv;                    // vhat
            }
            w1_rows_dot<HC>(W1s, vh, h, a, lane, wk);                            // (W1 vhat)_k
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const float nrm = sp[q] * nv[q];
                const float e = live[q] ? gp_weight / (float)m_gp * 2.f * (nrm - 1.f) : 0.f;
                cA[q] = e * nv[q] * sp[q] * (1.f - 2.f * s[q]);
                cB[q] = e * sp[q];
                if (live[q] && lane == 0) acc_gp += (double)((nrm - 1.f) * (nrm - 1.f));
            }
        }
        const int set = is_gap ? 0 : 1;
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            if (!live[q]) continue;
            const int64_t r = (is_gap ? 0 : n_gap) + base + q;
            float* yrow = Y + r * (int64_t)ldy;
#pragma unroll
            for (int c = 0; c < HC; ++c) {
                const int j = c * 64 + lane;
                if (j < h) yrow[j] = cA[q] * x[q][c] + cB[q] * vh[q][c];
            }
            if (lane == 0) { yrow[h] = cA[q]; acc_b2[set] += (double)cA[q]; }
            if (lane < a) {
                U[r * (int64_t)a + lane] = u[q];
                acc_w2[set] += (double)(cA[q] * hid[q] + cB[q] * mr[q] * wk[q]);
            }
        }
    }
    __shared__ double red[WAVES][2 * AMAX + 5];
    if (lane < a) { red[wave][lane] = acc_w2[0]; red[wave][AMAX + lane] = acc_w2[1]; }
    if (lane == 0) {
        red[wave][2 * AMAX + 0] = acc_b2[0]; red[wave][2 * AMAX + 1] = acc_b2[1]; red[wave][2 * AMAX + 2] = acc_gp;
        red[wave][2 * AMAX + 3] = acc_ds; red[wave][2 * AMAX + 4] = acc_dt;
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < 2 * a + 5) {
        // output order: gap w2 [a], gap b2, gp w2 [a], gp b2, gp sum, D_s sum, D_t sum
        int slot;
        if (t < a) slot = t;
        else if (t == a) slot = 2 * AMAX + 0;
        else if (t < 2 * a + 1) slot = AMAX + (t - a - 1);
        else slot = 2 * AMAX + 1 + (t - (2 * a + 1));
        double v = 0.0;
        for (int w = 0; w < WAVES; ++w) v += red[w][slot];
        part_rows[(int64_t)t * gridDim.x + blockIdx.x] = v;          // [column][block]: the fold reads columns coalesced
    }
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// sum of one column of the block partials by ONE wavefront: coalesced loads, fixed order
__device__ __forceinline__ double column_sum(const double* __restrict__ part_rows, int col, int row_blocks, int lane) {
    double v = 0.0;
    for (int b = lane; b < row_blocks; b += 64) v += part_rows[(int64_t)col * row_blocks + b];
    return wave_sum_d(v);
}

// gW1 | gb1 = -sign(gap) * UtY_gap + UtY_gp;  gw2, gb2 likewise from the block sums;  loss = -|gap| + gp_weight * mean.
// Every wavefront forms sign(gap) itself (two columns); the a + 2 outputs are spread over the grid's wavefronts.
__global__ void __launch_bounds__(TB)
k_critic_final(Critic C, RowsIn R, float gp_weight, const double* __restrict__ part_rows, int row_blocks,
               const float* __restrict__ UtY_gap, const float* __restrict__ UtY_gp, int ldc,
               float* __restrict__ loss, float* __restrict__ gW1, float* __restrict__ gb1, float* __restrict__ gw2,
               float* __restrict__ gb2) {
    const int t = threadIdx.x, a = C.a, lane = t % 64;
    const int gw = blockIdx.x * WAVES + t / 64, nw = gridDim.x * WAVES;
    const double gap = column_sum(part_rows, 2 * a + 3, row_blocks, lane) / (double)R.n_s
                     - column_sum(part_rows, 2 * a + 4, row_blocks, lane) / (double)R.n_t;
    const float sgn = gap > 0.0 ? 1.f : (gap < 0.0 ? -1.f : 0.f);
    for (int k = gw; k <= a + 1; k += nw) {
        if (k <= a) {          // hidden unit k (k < a) or the output bias (k == a): gap part and penalty part
            const double g0 = column_sum(part_rows, k, row_blocks, lane), g1 = column_sum(part_rows, a + 1 + k, row_blocks, lane);
            if (lane == 0) {
                const float v = (float)(-(double)sgn * g0 + g1);
                if (k < a) gw2[k] = v; else gb2[0] = v;
            }
        } else {
            const double gp = column_sum(part_rows, 2 * a + 2, row_blocks, lane);
            const double m_gp = (double)(R.n_s + R.n_t + R.n_i);
            if (lane == 0) loss[0] = (float)(-(gap < 0 ? -gap : gap) + (double)gp_weight * gp / m_gp);     // adagcn.py:177
        }
    }
    if (blockIdx.x == 0 && t < a) gb1[t] = -sgn * UtY_gap[(int64_t)t * ldc + C.h] + UtY_gp[(int64_t)t * ldc + C.h];
    for (int64_t e = (int64_t)blockIdx.x * TB + t; e < (int64_t)a * C.h; e += (int64_t)gridDim.x * TB) {
        const int64_t o = (e / C.h) * ldc + e % C.h;
        gW1[e] = -sgn * UtY_gap[o] + UtY_gp[o];
    }
}

struct Ws { double* part_rows; float* U; float* Y; float* UtY_gap; float* UtY_gp; void* gemm_ws; size_t gemm_bytes; size_t total; };

constexpr int ROW_BLOCKS = 512;

Ws carve(void* base, int64_t n_s, int64_t n_t, int64_t n_i, int h, int a) {
    const int64_t n_gap = n_s + n_t, m_gp = n_s + n_t + n_i, rows = n_gap + m_gp;
    const int ldy = h + 4;
    Ws w{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void* p = base ? (void*)((char*)base + off) : nullptr;
        off += gda_align_up(bytes, 256);
        return p;
    };
    w.part_rows = (double*)take(sizeof(double) * ROW_BLOCKS * (2 * a + 5));
    w.U = (float*)take(sizeof(float) * rows * a);
    w.Y = (float*)take(sizeof(float) * rows * ldy);
    w.UtY_gap = (float*)take(sizeof(float) * a * ldy);
    w.UtY_gp = (float*)take(sizeof(float) * a * ldy);
    const size_t g1 = gda_gemm_workspace_bytes(GDA_GEMM_TN, a, ldy, n_gap), g2 = gda_gemm_workspace_bytes(GDA_GEMM_TN, a, ldy, m_gp);
    w.gemm_bytes = g1 > g2 ? g1 : g2;
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
    const int64_t n_gap = n_s + n_t, m_gp = n_s + n_t + n_i;
    const int ldy = h + 4;
    const bool wide = h > 128;
    if (lds > 48 * 1024) {                         // the widest legal critic needs 66 KB of dynamic LDS
        const void* fn = wide ? reinterpret_cast<const void*>(k_critic_rows<4>) : reinterpret_cast<const void*>(k_critic_rows<2>);
        GDA_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (wide) k_critic_rows<4><<<ROW_BLOCKS, TB, lds, stream>>>(C, R, dr, gp_weight, ws.U, ws.Y, ldy, ws.part_rows);
    else k_critic_rows<2><<<ROW_BLOCKS, TB, lds, stream>>>(C, R, dr, gp_weight, ws.U, ws.Y, ldy, ws.part_rows);
    GDA_LAUNCH_CHECK();
    // gW1 | gb1 of the two row groups: U^T Y on the matrix cores (columns h+1 .. h+3 of Y pad the 16-byte row
    // stride: never written, they only reach columns of UtY that nobody reads)
    int st = gda_gemm_f32(GDA_GEMM_TN, a, ldy, n_gap, ws.U, a, ws.Y, ldy, ws.UtY_gap, ldy, ws.gemm_ws, ws.gemm_bytes, stream_);
    if (st != GDA_OK) return st;
    st = gda_gemm_f32(GDA_GEMM_TN, a, ldy, m_gp, ws.U + n_gap * a, a, ws.Y + n_gap * (int64_t)ldy, ldy, ws.UtY_gp, ldy,
                      ws.gemm_ws, ws.gemm_bytes, stream_);
    if (st != GDA_OK) return st;
    k_critic_final<<<16, TB, 0, stream>>>(C, R, gp_weight, ws.part_rows, ROW_BLOCKS, ws.UtY_gap, ws.UtY_gp, ldy,
                                         loss, gW1, gb1, gw2, gb2);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}
