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
// so a row contributes  u (cA x + cB vhat)^T = u y^T  to gW1 and  cA u  to gb1 (cA, cB scalars); gw2, gb2 and the
// loss are fixed-order block sums.  Two paths:
//  * h in {64, 96, 128}, 16-byte aligned encodings -- the matrix-core path, TWO launches (round 6):
//      k_critic_rows_mfma    32 rows per wavefront; Z^T = W1 X^T, V^T = W1^T U^T, T'^T = W1 Y^T and the tile's U^T Y on
//                            the fp32 MFMAs; the four tiles of a workgroup summed through LDS into ONE partial
//                            [a][h] per workgroup (gap rows and penalty rows separately); per-unit / per-row sums
//                            (w2, b1, b2 terms, penalty, D per domain) as fixed-order block sums.  U and Y never
//                            reach memory (until round 6: 37 MB written, read back by two GEMM launches).
//      k_critic_final_fused  the partials folded in a fixed order; sign(gap); gW1, gb1, gw2, gb2 = -sign * gap part
//                            + penalty part; loss
//  * any other shape -- the readlane path:
//      k_critic_rows   every row (gap rows with their own masks, carrying the UNSIGNED derivative of the gap, then
//                      the penalty rows): U [R, a], Y [R, h + 4] (column h = cA), block sums
//      gda_gemm_f32    U^T Y for the gap rows and for the penalty rows (TN, deterministic row-slab split)
//      k_critic_final  sign(gap) from the block sums; the four gradients and the loss
//    One wavefront takes RB = 4 rows: lane k owns hidden unit k (a <= 64) for the W1 x and W1 vhat products, lanes
//    own columns j, j + 64, ... for v = W1^T u; W1 sits in LDS with a padded leading dimension (conflict free both
//    ways), every W1 word read feeds the four rows, the rows' values are broadcast with v_readlane.
#include "gda_common.h"
#include "gda_philox.h"
#include "gda_adam_rule.h"

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

__device__ __forceinline__ double wave_sum_d_fwd(double v) {
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

// ---------------------------------------------------------------------------------------------------------------
// The same rows on the fp32 MATRIX CORES (h in {64, 96, 128}, a % 4 == 0): a wavefront takes 32 rows; the per-row
// matrix-vector products become 32-row tile products with the rows as the MFMA's column index,
//     Z^T [unit, row] = W1 X^T,      V^T [column, row] = W1^T U^T,      T'^T [unit, row] = W1 Y^T,
// so that a lane pair (l, l + 32) owns ONE row and holds 32 of its 64 (padded) hidden units / half of its columns in
// accumulator registers: every per-row scalar (z, s, |v|, cA, cB) is lane-local plus one cross-half shuffle.  With
// y = cA x + cB vhat formed in place over the row's x in LDS, W1 y = cA (a - b1) + cB W1 vhat: the w2 terms
// m (cA a + cB W1 vhat) are m (T' + cA b1), and the SAME tile is the B operand of the tile's gradient product
//     (U^T Y) [unit, column] = sum over the 32 rows  u[row, unit] y[row, column]            (rows = the MFMA's k).
// LDS per workgroup: W1 [a][h + 1], four X | Y tiles [32][h + 1], four U tiles [32][65] (row strides odd:
// conflict-free operand reads) = 132 KB at h = 128, a = 64.  The readlane version above spends two VALU
// instructions per (row, weight): 119 us for the 35 k rows of an AdaGCN critic step; this one 464 MFMAs per 32
// penalty rows (+ 128 for a tile with gap rows) including the gradient product.
// Same keep-bits as the readlane kernel (keyed on row and unit; one Philox call covers a lane's four adjacent units).
constexpr int RT = 32;

__host__ __device__ inline size_t lds_floats_mfma(int h, int a) {      // W1, four X | Y tiles, four U tiles
    return (size_t)a * (h + 1) + (size_t)WAVES * RT * (h + 1) + (size_t)WAVES * RT * (AMAX + 1);
}

using f32x16 = __attribute__((ext_vector_type(16))) float;

// Where row g of cat(e_s, e_t, interpolates) comes from, WITHOUT branches: x = t + al (s - t) with s = t and al = 0
// for the plain rows (exact: t + 0 * 0).  Loads behind per-lane branches cannot be batched by the compiler -- every one
// of a lane's 16 row pieces then costs its own memory round trip (123 us for this kernel, 2 x 16 round trips per tile).
struct RowSrc { const float* t; const float* s; float al; };

__device__ __forceinline__ RowSrc row_src(const RowsIn& R, int h, int64_t g, bool valid) {
    g = valid ? g : 0;
    const bool is_s = g < R.n_s, is_t = !is_s && g < R.n_s + R.n_t, plain = is_s || is_t;
    const int64_t i = plain ? 0 : g - R.n_s - R.n_t;
    int32_t si = 0, ti = 0;
    float al = 0.f;
    if (R.n_i > 0) { si = R.is[i]; ti = R.it[i]; al = R.alpha[i]; }         // uniform condition
    const float* t = is_s ? R.es + g * h : (is_t ? R.et + (g - R.n_s) * h : R.et + (int64_t)ti * h);
    const float* sp = plain ? t : R.es + (int64_t)si * h;
    return RowSrc{t, sp, plain ? 0.f : al};
}

__device__ __forceinline__ float4 mix4(const float4 t, const float4 s, float al, bool valid) {
    const float4 v = make_float4(t.x + al * (s.x - t.x), t.y + al * (s.y - t.y), t.z + al * (s.z - t.z), t.w + al * (s.w - t.w));
    return valid ? v : make_float4(0.f, 0.f, 0.f, 0.f);
}

__device__ __forceinline__ void lds_settle() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// sum over the 32 rows of column `lane` of a wave's tile [32][ldw] (64 columns in use): conflict-free reads, four chains
__device__ __forceinline__ float tile_column_sum(const float* Ts, int ldw, int lane) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int r = 0; r < RT; r += 4) {
        s0 += Ts[(size_t)r * ldw + lane]; s1 += Ts[(size_t)(r + 1) * ldw + lane];
        s2 += Ts[(size_t)(r + 2) * ldw + lane]; s3 += Ts[(size_t)(r + 3) * ldw + lane];
    }
    return (s0 + s1) + (s2 + s3);
}

#ifdef GDA_CRITIC_TRACE
// tracing build (tools/critic_trace.py): clock stamps of every wavefront's first tile at the phase boundaries
__device__ unsigned long long* gda_critic_trace_buf;
#define CT_STAMP(i) do { if (lane == 0 && gda_critic_trace_buf) gda_critic_trace_buf[((size_t)blockIdx.x * WAVES + wave) * 16 + (i)] = (i) >= 12 ? wall_clock64() : clock64(); } while (0)
#else
#define CT_STAMP(i) do { } while (0)
#endif

// The critic's own optimiser step applied where the gradients are formed (gda_wgan_critic_adam_f32): torch.optim.Adam
// state of W1, b1, w2, b2 (in this order); on = 0: gradients only.
struct CriticAdam {
    float* p[4]; float* m[4]; float* v[4]; float* step[4];
    float lr, beta1, beta2, eps, weight_decay;
    int on;
};

constexpr int LDU = AMAX + 1;      // row stride of a wavefront's U tile [32][AMAX + 1]

// Sum of the four wavefronts' 32 x 32 accumulator blocks acc[NB rd + k] (k < NB; block bi = unit tile bi / HT, column tile
// bi % HT of the [a][h] product) into the workgroup's partial: every wavefront stages its NB blocks in LDS
// (stage[wave][k][32][32]), then the 256 threads add the four copies in wave order and write (or add to) the partial's
// entries.  Barriers inside: called by every wavefront of the workgroup, the staging area must be free on entry.
template <int NB, int HT, int NTOT>
__device__ __forceinline__ void quad_sum_round(const f32x16 (&acc)[NTOT], int rd, float* stage, int wave, int rl, int half, int a, int h,
                                               float* __restrict__ out, bool add) {
#pragma unroll
    for (int k = 0; k < NB; ++k)
#pragma unroll
        for (int q = 0; q < 16; ++q)
            stage[((size_t)(wave * NB + k) * 32 + (q & 3) + 8 * (q >> 2) + 4 * half) * 32 + rl] = acc[NB * rd + k][q];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NB * 4; ++j) {
        const int idx = threadIdx.x + TB * j, k = idx >> 10, rowl = (idx >> 5) & 31, coll = idx & 31;
        const int bi = NB * rd + k, row = 32 * (bi / HT) + rowl;
        float v = stage[idx];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) v += stage[(size_t)w * NB * 1024 + idx];
        if (row < a) {
            float* o = out + (size_t)row * h + 32 * (bi % HT) + coll;
            *o = add ? *o + v : v;
        }
    }
    __syncthreads();
}

template <int HT>
__global__ void __launch_bounds__(TB)
k_critic_rows_mfma(Critic C, RowsIn R, Drop dr, float gp_weight, float* __restrict__ P_gap, float* __restrict__ P_gp,
                   double* __restrict__ part_rows, CriticAdam ad) {
    extern __shared__ __attribute__((aligned(16))) float W1s[];
    __shared__ float b1s[AMAX], w2s[AMAX];
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    const int h = C.h, a = C.a, ldw = h + 1;
    float* Tall = W1s + (size_t)a * ldw;                                    // the four waves' tiles [32][h + 1]; at the end of a trip
    float* Ts = Tall + (size_t)wave * RT * ldw;                             //   the staging area of the U^T Y sum (4 x HT blocks)
    float* Uall = Tall + (size_t)WAVES * RT * ldw;                          // the four waves' U tiles [32][65]; the staging area of
    float* Us = Uall + (size_t)wave * RT * LDU;                             //   the gap rows' sum (4 x 2 blocks)
    if (ad.on && blockIdx.x == 0 && threadIdx.x < 4) *ad.step[threadIdx.x] += 1.0f;      // read by the final launch
    CT_STAMP(0);
    CT_STAMP(12);
#ifdef GDA_CRITIC_TRACE
    if (lane == 0 && gda_critic_trace_buf)     // where the wavefront runs: HW_ID (cu 11:8, sh 12, se 15:13) | XCC_ID << 32
        gda_critic_trace_buf[((size_t)blockIdx.x * WAVES + wave) * 16 + 11] =
            (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
#endif
    {   // W1 -> LDS, row stride h + 1: float4 loads, eight in flight per thread (h = 32 HT: the row / column split is a shift)
        const int n4 = a * (8 * HT);
        const float4* w4 = reinterpret_cast<const float4*>(C.W1);
        for (int e0 = threadIdx.x; e0 < n4; e0 += 8 * TB) {
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = e0 + k * TB < n4 ? w4[e0 + k * TB] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = e0 + k * TB;
                if (e < n4) {
                    float* t = W1s + (size_t)(e / (8 * HT)) * ldw + (e % (8 * HT)) * 4;
                    t[0] = v[k].x; t[1] = v[k].y; t[2] = v[k].z; t[3] = v[k].w;
                }
            }
        }
    }
    if (threadIdx.x < AMAX) {
        b1s[threadIdx.x] = threadIdx.x < a ? C.b1[threadIdx.x] : 0.f;
        w2s[threadIdx.x] = threadIdx.x < a ? C.w2[threadIdx.x] : 0.f;
    }
    __syncthreads();
    const uint64_t st = dr.p > 0.f ? (uint64_t)dr.step[0] : 0;
    const uint32_t thresh = (uint32_t)((double)dr.p * 4294967296.0 > 4294967295.0 ? 4294967295.0 : (double)dr.p * 4294967296.0);
    const float keep_on = 1.f / (1.f - dr.p);
    const int rl = lane & 31, half = lane >> 5;
    const int64_t n_gap = R.n_s + R.n_t, m_gp = R.n_s + R.n_t + R.n_i;
    const int64_t g_gp = (m_gp + RT - 1) / RT, nquad = (g_gp + WAVES - 1) / WAVES;
    const float b2 = C.b2[0];
    const int nit = a > 32 ? 2 : 1;                                         // unit tiles that hold real units
    const int64_t pe = (int64_t)a * h;                                      // entries of one U^T Y partial
    // Per-unit sums over the rows (the w2 terms; u . cA = the b1 terms): the per-(row, unit) values go through the wave's
    // U tile and lane i sums unit i's column over the 32 rows -- one accumulator per lane and sum instead of one per
    // unit register (the kernel has to stay under 256 VGPRs: beyond them values live in AGPRs and every use is a copy).
    // (Until round 6 a shuffle butterfly per unit register: 160 ds_bpermute, 10 k cycles a tile.)
    float w2sum_gap = 0.f, w2sum_gp = 0.f, b1sum_gap = 0.f, b1sum_gp = 0.f;
    double acc_b2[2] = {0.0, 0.0}, acc_gp = 0.0, acc_ds = 0.0, acc_dt = 0.0;

    // W1 as the MFMA's row operand: unit i = rl + 32 it, column k   (zero rows past a)
    // (no per-lane branch around the load: a branch inside the MFMA loops made the compiler carry the accumulators in
    // VGPRs and copy all 32 of them to and from the AGPRs around every MFMA)
    const float* w1row0 = W1s + (size_t)rl * ldw;                           // unit rl      (always < a: a >= 32 or rows read as 0 below)
    const float* w1row1 = W1s + (size_t)(rl + 32 < a ? rl + 32 : 0) * ldw;  // unit rl + 32 (clamped; masked by on1)
    const float on0 = rl < a ? 1.f : 0.f, on1 = rl + 32 < a ? 1.f : 0.f;
    const float* w1row0c = rl < a ? w1row0 : W1s;

    // ONE pass over the tiles of the penalty space cat(e_s, e_t, interpolates): its first n_gap rows ARE the gap rows, so a
    // tile that holds some also evaluates the gap terms of those rows from the same Z^T = W1 X^T (their own dropout draws)
    // instead of a tile of its own recomputing it.  A workgroup takes four consecutive tiles (a "quad") per trip so that
    // the U^T Y products of its wavefronts can be summed through LDS: every wavefront runs every trip (rows past the
    // end are dead: zero contributions), the barriers below are uniform.
    int64_t trip = 0;
    for (int64_t quad = blockIdx.x; quad < nquad; quad += gridDim.x, ++trip) {
        const int64_t grp = quad * WAVES + wave;
        const int64_t base = grp * RT;
        const int64_t limit = m_gp;
        const int64_t r = base + rl;                                        // this lane pair's row in the penalty space
        const bool live = r < limit;
        const bool quad_has_gap = quad * (WAVES * RT) < n_gap;              // workgroup-uniform
        const bool tile_has_gap = base < n_gap;                             // wave-uniform
        const bool live_g = r < n_gap;
        CT_STAMP(1);
        // ---- X tile -> LDS (rows past the limit read as zeros)
        // A wavefront is alone on its SIMD here: every dependent load is a full memory round trip.  So: a tile of source /
        // target rows only (wave-uniform) reads its rows directly, 8 pieces in flight; a tile with interpolates looks up
        // all its 4 HT row pieces' (index, index, alpha) first -- one round trip -- and then loads 8 pieces (16 loads) a pass.
        const bool plain_tile = base + RT <= n_gap;
        {
            constexpr int NP = 4 * HT, PS = HT % 2 == 0 ? 8 : 4;
            if (plain_tile) {
#pragma unroll
                for (int ps = 0; ps < NP / PS; ++ps) {
                    float4 xt[PS];
#pragma unroll
                    for (int q = 0; q < PS; ++q) {
                        const int idx = lane + 64 * (PS * ps + q), c4 = (idx % (8 * HT)) * 4;
                        const int64_t g = base + idx / (8 * HT);
                        xt[q] = *reinterpret_cast<const float4*>((g < R.n_s ? R.es + g * h : R.et + (g - R.n_s) * h) + c4);
                    }
#pragma unroll
                    for (int q = 0; q < PS; ++q) {
                        const int idx = lane + 64 * (PS * ps + q), row = idx / (8 * HT), c4 = (idx % (8 * HT)) * 4;
                        float* t = Ts + (size_t)row * ldw + c4;
                        t[0] = xt[q].x; t[1] = xt[q].y; t[2] = xt[q].z; t[3] = xt[q].w;
                    }
                }
            } else {
                RowSrc src[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const int64_t g = base + (lane + 64 * q) / (8 * HT);
                    src[q] = row_src(R, h, g, g < limit);
                }
#pragma unroll
                for (int ps = 0; ps < NP / PS; ++ps) {
                    float4 xt[PS], xs[PS];
#pragma unroll
                    for (int q = 0; q < PS; ++q) {
                        const int c4 = ((lane + 64 * (PS * ps + q)) % (8 * HT)) * 4;
                        xt[q] = *reinterpret_cast<const float4*>(src[PS * ps + q].t + c4);
                        xs[q] = *reinterpret_cast<const float4*>(src[PS * ps + q].s + c4);
                    }
#pragma unroll
                    for (int q = 0; q < PS; ++q) {                          // gap space = the first rows of the penalty space
                        const int idx = lane + 64 * (PS * ps + q), row = idx / (8 * HT), c4 = (idx % (8 * HT)) * 4;
                        const float4 v = mix4(xt[q], xs[q], src[PS * ps + q].al, base + row < limit);
                        float* t = Ts + (size_t)row * ldw + c4;
                        t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
                    }
                }
            }
        }
        lds_settle();
        CT_STAMP(2);
        // ---- Z^T = W1 X^T
        f32x16 accz[2];
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int q = 0; q < 16; ++q) accz[it][q] = 0.f;
        if (nit > 1) {
#pragma unroll 8
            for (int kk = 0; kk < h; kk += 2) {
                const float xb = Ts[(size_t)rl * ldw + kk + half];
                accz[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(on0 * w1row0c[kk + half], xb, accz[0], 0, 0, 0);
                accz[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(on1 * w1row1[kk + half], xb, accz[1], 0, 0, 0);
            }
        } else {
#pragma unroll 8
            for (int kk = 0; kk < h; kk += 2)
                accz[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(on0 * w1row0c[kk + half], Ts[(size_t)rl * ldw + kk + half], accz[0], 0, 0, 0);
        }
        // accz becomes a = W1 x + b1
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int q = 0; q < 16; ++q) accz[it][q] += b1s[32 * it + (q & 3) + 8 * (q >> 2) + 4 * half];
        CT_STAMP(3);
        // ---- the GAP terms of the rows that are source / target rows (their own keep-bits: sites 0 | 1, row numbered in its
        // domain): z, s, d gap / d z = cAg; the rows' part of gW1 = sum_r (cAg u_r) x_r^T on the matrix cores, summed over
        // the workgroup's four tiles through LDS
        if (quad_has_gap) {
            f32x16 accg[2 * HT];                                            // block it HT + jt
#pragma unroll
            for (int bi = 0; bi < 2 * HT; ++bi)
#pragma unroll
                for (int q = 0; q < 16; ++q) accg[bi][q] = 0.f;
            if (tile_has_gap) {
                float mg[2][16];
                float zg = 0.f;
                const uint32_t site_g = dr.site + (r < R.n_s ? 0u : 1u);
                const int64_t rowkey_g = r < R.n_s ? r : r - R.n_s;
#pragma unroll
                for (int it = 0; it < 2; ++it)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int i0 = 32 * it + 8 * g4 + 4 * half;
                        uint32_t rn[4] = {~0u, ~0u, ~0u, ~0u};
                        if (dr.p > 0.f && i0 < a) GdaPhilox::gen(dr.seed, (st << 20) ^ site_g, ((uint64_t)rowkey_g * (uint64_t)a + (uint64_t)i0) >> 2, rn);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int i = i0 + e, q = 4 * g4 + e;
                            const float av = accz[it][q];
                            const float kf = dr.p > 0.f ? (rn[e] >= thresh ? keep_on : 0.f) : 1.f;
                            const float m = (live_g && i < a && av > 0.f) ? kf : 0.f;
                            mg[it][q] = m;
                            zg = fmaf(w2s[i], m * av, zg);
                        }
                    }
                const float z_g = zg + __shfl_xor(zg, 32, 64) + b2;
                const float sg_g = 1.f / (1.f + __expf(-z_g)), sp_g = sg_g * (1.f - sg_g);
                const float cAg = live_g ? (r < R.n_s ? sp_g / (float)R.n_s : -sp_g / (float)R.n_t) : 0.f;      // d gap / d z_i
                if (live_g && half == 0) {
                    if (r < R.n_s) acc_ds += (double)sg_g; else acc_dt += (double)sg_g;
                    acc_b2[0] += (double)cAg;
                }
                // U tile of the gap rows, scaled by the row's cAg: [row][unit]
#pragma unroll
                for (int it = 0; it < 2; ++it)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int i = 32 * it + (q & 3) + 8 * (q >> 2) + 4 * half;
                        Us[(size_t)rl * LDU + i] = (mg[it][q] * w2s[i]) * cAg;
                    }
                lds_settle();
                b1sum_gap += tile_column_sum(Us, LDU, lane);
                // (cAg U)^T X: units x columns, the 32 rows are the MFMA's k
#pragma unroll
                for (int it = 0; it < 2; ++it)
                    if (it < nit) {
#pragma unroll 4
                        for (int k2 = 0; k2 < RT; k2 += 2) {
                            const float ua = Us[(size_t)(k2 + half) * LDU + 32 * it + rl];
                            const float* xr = Ts + (size_t)(k2 + half) * ldw + rl;
#pragma unroll
                            for (int jt = 0; jt < HT; ++jt) accg[it * HT + jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua, xr[32 * jt], accg[it * HT + jt], 0, 0, 0);
                        }
                    }
                // w2 terms of the gap rows
                lds_settle();
#pragma unroll
                for (int it = 0; it < 2; ++it)
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        Us[(size_t)rl * LDU + 32 * it + (q & 3) + 8 * (q >> 2) + 4 * half] = mg[it][q] * (cAg * accz[it][q]);
                lds_settle();
                w2sum_gap += tile_column_sum(Us, LDU, lane);
            }
            // the four tiles' products summed through the U tiles' LDS, two blocks a round, into the block's partial
            __syncthreads();
#pragma unroll
            for (int rd = 0; rd < HT; ++rd)
                quad_sum_round<2, HT>(accg, rd, Uall, wave, rl, half, a, h, P_gap + (int64_t)blockIdx.x * pe, trip > 0);
        }
        CT_STAMP(4);
        // ---- per unit: relu, keep, hid, u; per row: z, s, s'  (the penalty evaluation: site 2, row numbered in the penalty space)
        float mr[2][16];                                                    // keep / (1 - p) where the unit is on
        float zpart = 0.f;
        const uint32_t site = dr.site + 2u;
        const int64_t rowkey = r;
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int i0 = 32 * it + 8 * g4 + 4 * half;                 // units i0 .. i0 + 3 = registers 4 g4 .. 4 g4 + 3
                uint32_t rn[4] = {~0u, ~0u, ~0u, ~0u};
                if (dr.p > 0.f && i0 < a) GdaPhilox::gen(dr.seed, (st << 20) ^ site, ((uint64_t)rowkey * (uint64_t)a + (uint64_t)i0) >> 2, rn);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = i0 + e, q = 4 * g4 + e;
                    const float av = accz[it][q];
                    const float kf = dr.p > 0.f ? (rn[e] >= thresh ? keep_on : 0.f) : 1.f;
                    const float m = (live && i < a && av > 0.f) ? kf : 0.f;
                    mr[it][q] = m;
                    zpart = fmaf(w2s[i], m * av, zpart);                    // hid = m a, u = m w2
                }
            }
        const float z = zpart + __shfl_xor(zpart, 32, 64) + b2;
        const float sg = 1.f / (1.f + __expf(-z)), sp = sg * (1.f - sg);
        float cA = 0.f, cB = 0.f;
        f32x16 accp[2 * HT];                                                // this tile's U^T Y: block it HT + jt
        {
            f32x16 accv[HT];                                                // V^T, then Vhat^T: this row's columns
#pragma unroll
            for (int jt = 0; jt < HT; ++jt)
#pragma unroll
                for (int q = 0; q < 16; ++q) accv[jt][q] = 0.f;
            // ---- U tile -> LDS, V^T = W1^T U^T
            CT_STAMP(5);
            lds_settle();
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int i = 32 * it + (q & 3) + 8 * (q >> 2) + 4 * half;
                    Us[(size_t)rl * LDU + i] = mr[it][q] * w2s[i];          // u
                }
            lds_settle();
#pragma unroll 4
            for (int kk = 0; kk < a; kk += 2) {                             // a is a multiple of 4
                const int i = kk + half;
                const float ub = Us[(size_t)rl * LDU + i];
                const float* wrow = W1s + (size_t)i * ldw + rl;
#pragma unroll
                for (int jt = 0; jt < HT; ++jt) accv[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[32 * jt], ub, accv[jt], 0, 0, 0);
            }
            float n2 = 0.f;
#pragma unroll
            for (int jt = 0; jt < HT; ++jt)
#pragma unroll
                for (int q = 0; q < 16; ++q) n2 = fmaf(accv[jt][q], accv[jt][q], n2);
            n2 += __shfl_xor(n2, 32, 64);
            const float nv = sqrtf(n2), inv = nv > 0.f ? 1.f / nv : 0.f;
            const float nrm = sp * nv;
            const float e = live ? gp_weight / (float)m_gp * 2.f * (nrm - 1.f) : 0.f;
            cA = e * nv * sp * (1.f - 2.f * sg);
            cB = e * sp;
            if (live && half == 0) { acc_gp += (double)((nrm - 1.f) * (nrm - 1.f)); acc_b2[1] += (double)cA; }
            // ---- the row's Y = cA x + cB vhat, in place over its x in the tile (a lane pair owns its row)
            CT_STAMP(6);
#pragma unroll
            for (int jt = 0; jt < HT; ++jt)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    float* y = Ts + (size_t)rl * ldw + 32 * jt + (q & 3) + 8 * (q >> 2) + 4 * half;
                    *y = cA * *y + cB * (accv[jt][q] * inv);
                }
            lds_settle();
        }
        // ---- T'^T = W1 Y^T = cA (a - b1) + cB W1 vhat: the w2 terms are m (T' + cA b1)
        f32x16 acct[2];
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int q = 0; q < 16; ++q) acct[it][q] = 0.f;
        if (nit > 1) {
#pragma unroll 8
            for (int kk = 0; kk < h; kk += 2) {
                const float vb = Ts[(size_t)rl * ldw + kk + half];
                acct[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(on0 * w1row0c[kk + half], vb, acct[0], 0, 0, 0);
                acct[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(on1 * w1row1[kk + half], vb, acct[1], 0, 0, 0);
            }
        } else {
#pragma unroll 8
            for (int kk = 0; kk < h; kk += 2)
                acct[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(on0 * w1row0c[kk + half], Ts[(size_t)rl * ldw + kk + half], acct[0], 0, 0, 0);
        }
        CT_STAMP(7);
        // ---- U^T Y: units x columns, the 32 rows are the MFMA's k
#pragma unroll
        for (int bi = 0; bi < 2 * HT; ++bi)
#pragma unroll
            for (int q = 0; q < 16; ++q) accp[bi][q] = 0.f;
#pragma unroll
        for (int it = 0; it < 2; ++it)
            if (it < nit) {
#pragma unroll 4
                for (int k2 = 0; k2 < RT; k2 += 2) {
                    const float ua = Us[(size_t)(k2 + half) * LDU + 32 * it + rl];
                    const float* yr = Ts + (size_t)(k2 + half) * ldw + rl;
#pragma unroll
                    for (int jt = 0; jt < HT; ++jt) accp[it * HT + jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua, yr[32 * jt], accp[it * HT + jt], 0, 0, 0);
                }
            }
        // ---- the w2 terms m (T' + cA b1) and the b1 terms cA u: column sums over the tile's rows
        lds_settle();
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int i = 32 * it + (q & 3) + 8 * (q >> 2) + 4 * half;
                Us[(size_t)rl * LDU + i] = mr[it][q] * (acct[it][q] + cA * b1s[i]);
            }
        lds_settle();
        w2sum_gp += tile_column_sum(Us, LDU, lane);
        lds_settle();
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int i = 32 * it + (q & 3) + 8 * (q >> 2) + 4 * half;
                Us[(size_t)rl * LDU + i] = (mr[it][q] * w2s[i]) * cA;
            }
        lds_settle();
        b1sum_gp += tile_column_sum(Us, LDU, lane);
        CT_STAMP(8);
        // ---- the four tiles' U^T Y summed through LDS (over the X | Y tiles), one unit tile a round, into the block's partial
        __syncthreads();
#pragma unroll
        for (int rd = 0; rd < 2; ++rd)
            quad_sum_round<HT, HT>(accp, rd, Tall, wave, rl, half, a, h, P_gp + (int64_t)blockIdx.x * pe, trip > 0);
        // (the last barrier of the round: the tiles are free for the next trip)
    }
    // ---- block partials of the per-unit and per-row sums
    CT_STAMP(9);
    __shared__ double red[WAVES][4 * AMAX + 5];
    if (lane < a) {                                                         // lane = unit
        red[wave][lane] = (double)w2sum_gap; red[wave][AMAX + lane] = (double)w2sum_gp;
        red[wave][2 * AMAX + 5 + lane] = (double)b1sum_gap; red[wave][3 * AMAX + 5 + lane] = (double)b1sum_gp;
    }
    const double b20 = wave_sum_d_fwd(acc_b2[0]), b21 = wave_sum_d_fwd(acc_b2[1]), gps = wave_sum_d_fwd(acc_gp);
    const double dss = wave_sum_d_fwd(acc_ds), dts = wave_sum_d_fwd(acc_dt);
    if (lane == 0) {
        red[wave][2 * AMAX + 0] = b20; red[wave][2 * AMAX + 1] = b21; red[wave][2 * AMAX + 2] = gps;
        red[wave][2 * AMAX + 3] = dss; red[wave][2 * AMAX + 4] = dts;
    }
    __syncthreads();
    // columns of part_rows: [0, a) w2 gap | a: b2 gap | [a + 1, 2a + 1) w2 penalty | 2a + 1: b2 penalty | 2a + 2: penalty
    // | 2a + 3, 2a + 4: sum D over source / target | [2a + 5, 3a + 5) b1 gap | [3a + 5, 4a + 5) b1 penalty
    for (int t = threadIdx.x; t < 4 * a + 5; t += TB) {
        int slot;
        if (t < a) slot = t;
        else if (t == a) slot = 2 * AMAX + 0;
        else if (t < 2 * a + 1) slot = AMAX + (t - a - 1);
        else if (t < 2 * a + 5) slot = 2 * AMAX + 1 + (t - (2 * a + 1));
        else if (t < 3 * a + 5) slot = 2 * AMAX + 5 + (t - (2 * a + 5));
        else slot = 3 * AMAX + 5 + (t - (3 * a + 5));
        double v = 0.0;
        for (int w = 0; w < WAVES; ++w) v += red[w][slot];
        part_rows[(int64_t)t * gridDim.x + blockIdx.x] = v;
    }
    CT_STAMP(10);
    CT_STAMP(13);
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

// The matrix-core path's last launch: the blocks' U^T Y partials [block][a h] folded in a fixed order (16 wavefronts of a
// workgroup = 16 chains over the blocks for 64 adjacent entries, eight loads in flight each, combined in chain order
// through LDS), gW1 = -sign(gap) * gap part + penalty part; gb1, gw2, gb2 and the loss from the blocks' column sums.
constexpr int FIN_TB = 1024, FIN_CH = FIN_TB / 64;

__device__ __forceinline__ float chain_sum(const float* __restrict__ P, int64_t pe, int64_t e, int blocks, int chain) {
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b0 = chain; b0 < blocks; b0 += 8 * FIN_CH) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = b0 + k * FIN_CH < blocks ? P[(int64_t)(b0 + k * FIN_CH) * pe + e] : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += v[k];
    }
    return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

__global__ void __launch_bounds__(FIN_TB)
k_critic_final_fused(Critic C, RowsIn R, float gp_weight, const double* __restrict__ part_rows, int row_blocks, int gap_blocks,
                     const float* __restrict__ P_gap, const float* __restrict__ P_gp, float* __restrict__ loss,
                     float* __restrict__ gW1, float* __restrict__ gb1, float* __restrict__ gw2, float* __restrict__ gb2,
                     CriticAdam ad) {
    __shared__ float cg[FIN_CH][64], cp[FIN_CH][64];
    // the update of element i of tensor k (0 W1, 1 b1, 2 w2, 3 b2) by the thread that formed its gradient; the step
    // counters were incremented by the row kernel
    auto update = [&](int k, int64_t i, float g) {
        if (ad.on)
            gda_adam_element(ad.p[k] + i, g, ad.m[k] + i, ad.v[k] + i, gda_adam_coef(*ad.step[k], ad.lr, ad.beta1, ad.beta2),
                             ad.beta1, ad.beta2, ad.eps, ad.weight_decay);
    };
    const int t = threadIdx.x, a = C.a, lane = t % 64, chain = t / 64;
    const int64_t pe = (int64_t)a * C.h, e = (int64_t)blockIdx.x * 64 + lane;
    // the W1 entry this thread will update, its moments and the bias corrections: fetched beside the partials
    float w_p = 0.f, w_m = 0.f, w_v = 0.f;
    GdaAdamCoef w_c{0.f, 1.f};
    if (ad.on && chain == 0 && e < pe) {
        w_p = ad.p[0][e]; w_m = ad.m[0][e]; w_v = ad.v[0][e];
        w_c = gda_adam_coef(*ad.step[0], ad.lr, ad.beta1, ad.beta2);
    }
    if (e < pe) {
        cg[chain][lane] = chain_sum(P_gap, pe, e, gap_blocks, chain);
        cp[chain][lane] = chain_sum(P_gp, pe, e, row_blocks, chain);
    }
    const int gw = blockIdx.x * FIN_CH + chain, nw = gridDim.x * FIN_CH;
    const double gap = column_sum(part_rows, 2 * a + 3, row_blocks, lane) / (double)R.n_s
                     - column_sum(part_rows, 2 * a + 4, row_blocks, lane) / (double)R.n_t;
    const float sgn = gap > 0.0 ? 1.f : (gap < 0.0 ? -1.f : 0.f);
    for (int k = gw; k <= 2 * a + 1; k += nw) {
        if (k <= a) {          // hidden unit k (k < a) or the output bias (k == a): gap part and penalty part
            const double g0 = column_sum(part_rows, k, row_blocks, lane), g1 = column_sum(part_rows, a + 1 + k, row_blocks, lane);
            if (lane == 0) {
                const float v = (float)(-(double)sgn * g0 + g1);
                if (k < a) { gw2[k] = v; update(2, k, v); } else { gb2[0] = v; update(3, 0, v); }
            }
        } else if (k == a + 1) {
            const double gp = column_sum(part_rows, 2 * a + 2, row_blocks, lane);
            const double m_gp = (double)(R.n_s + R.n_t + R.n_i);
            if (lane == 0) loss[0] = (float)(-(gap < 0 ? -gap : gap) + (double)gp_weight * gp / m_gp);     // adagcn.py:177
        } else {               // first-layer bias of unit k - a - 2
            const int u = k - a - 2;
            const double g0 = column_sum(part_rows, 2 * a + 5 + u, row_blocks, lane), g1 = column_sum(part_rows, 3 * a + 5 + u, row_blocks, lane);
            if (lane == 0) { const float v = (float)(-(double)sgn * g0 + g1); gb1[u] = v; update(1, u, v); }
        }
    }
    __syncthreads();
    if (chain == 0 && e < pe) {
        float g = 0.f, p = 0.f;
#pragma unroll
        for (int c = 0; c < FIN_CH; ++c) { g += cg[c][lane]; p += cp[c][lane]; }
        const float v = -sgn * g + p;
        gW1[e] = v;
        if (ad.on) {
            ad.p[0][e] = gda_adam_value(w_p, v, w_m, w_v, w_c, ad.beta1, ad.beta2, ad.eps, ad.weight_decay);
            ad.m[0][e] = w_m; ad.v[0][e] = w_v;
        }
    }
}

struct Ws {
    double* part_rows; float* U; float* Y; float* UtY_gap; float* UtY_gp; void* gemm_ws; size_t gemm_bytes;
    float* P_gap; float* P_gp;                      // the matrix-core path's block partials (they share the bytes of U | Y ...)
    size_t total;
};

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
    w.part_rows = (double*)take(sizeof(double) * ROW_BLOCKS * (4 * a + 5));
    const size_t shared_from = off;
    w.U = (float*)take(sizeof(float) * rows * a);
    w.Y = (float*)take(sizeof(float) * rows * ldy);
    w.UtY_gap = (float*)take(sizeof(float) * a * ldy);
    w.UtY_gp = (float*)take(sizeof(float) * a * ldy);
    const size_t g1 = gda_gemm_workspace_bytes(GDA_GEMM_TN, a, ldy, n_gap), g2 = gda_gemm_workspace_bytes(GDA_GEMM_TN, a, ldy, m_gp);
    w.gemm_bytes = g1 > g2 ? g1 : g2;
    w.gemm_ws = take(w.gemm_bytes);
    const size_t end_rows = off;
    off = shared_from;                              // the other path's layout over the same bytes
    w.P_gap = (float*)take(sizeof(float) * ROW_BLOCKS * (size_t)a * h);
    w.P_gp = (float*)take(sizeof(float) * ROW_BLOCKS * (size_t)a * h);
    w.total = off > end_rows ? off : end_rows;
    return w;
}

}  // namespace

#ifdef GDA_CRITIC_TRACE
extern "C" int gda_dbg_critic_trace(unsigned long long* buf) {
    GDA_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(gda_critic_trace_buf), &buf, sizeof(buf)));
    return GDA_OK;
}
#endif

extern "C" size_t gda_wgan_critic_workspace_bytes(int64_t n_s, int64_t n_t, int64_t n_i, int h, int a) {
    if (n_s <= 0 || n_t <= 0 || n_i < 0 || h <= 0 || a <= 0) return 0;
    return carve(nullptr, n_s, n_t, n_i, h, a).total;
}

static int critic_update(const float* es, int64_t n_s, const float* et, int64_t n_t, int h,
                         const int32_t* idx_s, const int32_t* idx_t, const float* alpha, int64_t n_i,
                         const float* W1, const float* b1, const float* w2, const float* b2, int a,
                         float dropout_p, uint64_t seed, const int64_t* step, uint32_t site,
                         float gp_weight, float* loss, float* gW1, float* gb1, float* gw2, float* gb2,
                         const CriticAdam& ad, void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
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
    const int64_t n_gap = n_s + n_t, m_gp = n_s + n_t + n_i;
    const int ldy = h + 4;
    // the matrix-core path: h = 64, 96, 128 (LDS: W1, four X | Y tiles, four U tiles)
    const bool mfma = h % 32 == 0 && h >= 64 && h <= 128 && a % 4 == 0 && ((uintptr_t)es | (uintptr_t)et | (uintptr_t)W1) % 16 == 0;
    if (ad.on && !mfma) return GDA_E_UNSUPPORTED;         // the update rides in the matrix-core path's final launch only
    int row_blocks = ROW_BLOCKS;
    if (mfma) {
        // 32 rows per wavefront, four tiles per workgroup and trip: as many workgroups as there are quads (at most ROW_BLOCKS)
        const int64_t tiles = gda_cdiv(m_gp, RT);             // the gap rows ride in the penalty space's first tiles
        const int64_t want = gda_cdiv(tiles, WAVES);
        row_blocks = (int)(want < ROW_BLOCKS ? want : ROW_BLOCKS);
        const int64_t gap_quads = gda_cdiv(n_gap, WAVES * RT);
        const int gap_blocks = (int)(gap_quads < row_blocks ? gap_quads : row_blocks);
        const size_t lds = lds_floats_mfma(h, a) * sizeof(float);
#define GDA_CRITIC_MFMA(HT)                                                                                          \
        {                                                                                                            \
            GDA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_critic_rows_mfma<HT>),                   \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                  \
            k_critic_rows_mfma<HT><<<row_blocks, TB, lds, stream>>>(C, R, dr, gp_weight, ws.P_gap, ws.P_gp, ws.part_rows, ad); \
        }
        switch (h / 32) {
            case 2: GDA_CRITIC_MFMA(2) break;
            case 3: GDA_CRITIC_MFMA(3) break;
            default: GDA_CRITIC_MFMA(4) break;
        }
#undef GDA_CRITIC_MFMA
        GDA_LAUNCH_CHECK();
        k_critic_final_fused<<<(unsigned)gda_cdiv((int64_t)a * h, 64), FIN_TB, 0, stream>>>(
            C, R, gp_weight, ws.part_rows, row_blocks, gap_blocks, ws.P_gap, ws.P_gp, loss, gW1, gb1, gw2, gb2, ad);
        GDA_LAUNCH_CHECK();
        return GDA_OK;
    } else {
        const size_t lds = lds_floats(h, a) * sizeof(float);
        const bool wide = h > 128;
        if (lds > 48 * 1024) {                     // the widest legal critic needs 66 KB of dynamic LDS
            const void* fn = wide ? reinterpret_cast<const void*>(k_critic_rows<4>) : reinterpret_cast<const void*>(k_critic_rows<2>);
            GDA_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        if (wide) k_critic_rows<4><<<ROW_BLOCKS, TB, lds, stream>>>(C, R, dr, gp_weight, ws.U, ws.Y, ldy, ws.part_rows);
        else k_critic_rows<2><<<ROW_BLOCKS, TB, lds, stream>>>(C, R, dr, gp_weight, ws.U, ws.Y, ldy, ws.part_rows);
    }
    GDA_LAUNCH_CHECK();
    // gW1 | gb1 of the two row groups: U^T Y on the matrix cores (columns h+1 .. h+3 of Y pad the 16-byte row
    // stride: never written, they only reach columns of UtY that nobody reads)
    int st = gda_gemm_f32(GDA_GEMM_TN, a, ldy, n_gap, ws.U, a, ws.Y, ldy, ws.UtY_gap, ldy, ws.gemm_ws, ws.gemm_bytes, stream_);
    if (st != GDA_OK) return st;
    st = gda_gemm_f32(GDA_GEMM_TN, a, ldy, m_gp, ws.U + n_gap * a, a, ws.Y + n_gap * (int64_t)ldy, ldy, ws.UtY_gp, ldy,
                      ws.gemm_ws, ws.gemm_bytes, stream_);
    if (st != GDA_OK) return st;
    k_critic_final<<<16, TB, 0, stream>>>(C, R, gp_weight, ws.part_rows, row_blocks, ws.UtY_gap, ws.UtY_gp, ldy,
                                         loss, gW1, gb1, gw2, gb2);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_wgan_critic_f32(const float* es, int64_t n_s, const float* et, int64_t n_t, int h,
                                   const int32_t* idx_s, const int32_t* idx_t, const float* alpha, int64_t n_i,
                                   const float* W1, const float* b1, const float* w2, const float* b2, int a,
                                   float dropout_p, uint64_t seed, const int64_t* step, uint32_t site,
                                   float gp_weight, float* loss, float* gW1, float* gb1, float* gw2, float* gb2,
                                   void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    CriticAdam ad{};
    return critic_update(es, n_s, et, n_t, h, idx_s, idx_t, alpha, n_i, W1, b1, w2, b2, a, dropout_p, seed, step, site,
                         gp_weight, loss, gW1, gb1, gw2, gb2, ad, workspace, workspace_bytes, stream_);
}

extern "C" int gda_wgan_critic_adam_f32(const float* es, int64_t n_s, const float* et, int64_t n_t, int h,
                                        const int32_t* idx_s, const int32_t* idx_t, const float* alpha, int64_t n_i,
                                        int a, float dropout_p, uint64_t seed, const int64_t* step, uint32_t site,
                                        float gp_weight, float* loss, const gda_adam_tensor* params, float lr, float beta1,
                                        float beta2, float eps, float weight_decay,
                                        void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    if (!params) return GDA_E_NULL;
    CriticAdam ad{};
    const int64_t want[4] = {(int64_t)a * h, a, a, 1};
    for (int k = 0; k < 4; ++k) {
        const gda_adam_tensor& t = params[k];
        if (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq || !t.step) return GDA_E_NULL;
        if (t.numel != want[k]) return GDA_E_SIZE;
        ad.p[k] = t.param; ad.m[k] = t.exp_avg; ad.v[k] = t.exp_avg_sq; ad.step[k] = t.step;
    }
    ad.lr = lr; ad.beta1 = beta1; ad.beta2 = beta2; ad.eps = eps; ad.weight_decay = weight_decay; ad.on = 1;
    return critic_update(es, n_s, et, n_t, h, idx_s, idx_t, alpha, n_i, params[0].param, params[1].param, params[2].param,
                         params[3].param, a, dropout_p, seed, step, site, gp_weight, loss, const_cast<float*>(params[0].grad),
                         const_cast<float*>(params[1].grad), const_cast<float*>(params[2].grad), const_cast<float*>(params[3].grad),
                         ad, workspace, workspace_bytes, stream_);
}
