// Multi-kernel Gaussian MMD over sampled source/target rows, forward + backward, gfx950.
//
// Replaces the torch elementwise chain of pygda/utils/mmd.py:4-159 (guassian_kernel,
// get_MMD, MMD), which materialises an [m, m, d] difference tensor per resample
// (m = 2000, d = 128: 2 GB, several live at once and kept for backward).  Here the
// pairwise squared distances (mmd.py:43-46: sum_k (total[j,k]-total[i,k])^2) are produced tile by
// tile on the fp32 matrix cores in Gram form |a'|^2 + |b'|^2 - 2 a'.b' over PIVOT-SHIFTED rows
// a' = a - p, p = the first sampled row of the resample: the statistic only depends on differences,
// (a - p) - (b - p) = a - b, and after the shift the operands are of the size of the spread of the
// batch, not of its common offset -- so the Gram form does not cancel when the domains collapse onto a
// far-away point (features c + eps * noise, c >> eps), the regime the loss drives training into.  The
// result is clamped at 0 (the reference's sum of squares cannot be negative).  Only the [m, m] matrix
// is kept (16 MB per resample, L2/MALL resident) and the backward pass reuses it.
//
//   k_rowstats   per 64-row chunk: sum |t_i|^2 and the column sums of the shifted rows
//   k_bandwidth  bandwidth per resample (mmd.py:50-51) from those sufficient statistics:
//                sum_ij |t_i - t_j|^2 = 2 m sum_i |t_i|^2 - 2 |sum_i t_i|^2  -- O(m d) instead of a pass over
//                the [m, m] matrix, so the kernel weights can be formed where the distances are produced
//   k_pairdist   tile 64x64 of L2 = |t_i|^2 + |t_j|^2 - 2 t_i.t_j on v_mfma_f32_32x32x2_f32 (norms as the
//                same k-ordered fma chain, from the staged tiles)      (MFMA bound: 2*m^2*d flop), and in
//                its epilogue K = sum_q exp(-L2/bw_q) (mmd.py:52-55), the signed block sums
//                XX+YY-XY-YX (mmd.py:100-106) and the backward's weights g = dK/dL2, which is all that is
//                written: the distance matrix itself never reaches memory
//   k_finalize   mean per resample, average over resamples (mmd.py:152-157)
//   k_bwd        grad_total[i,:] = 4 * ((sum_j G[i,j]) total[i,:] - sum_j G[i,j] total[j,:]),
//                G = dloss/dL2 (symmetric; the bandwidth is a constant, mmd.py:50 .data); the
//                G x total product runs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact
//                fp32 fma chains, measured no less accurate than the difference form here)
// All reductions are fixed-order (no atomics): results are run-to-run deterministic.
#include "gda_common.h"

#include <cstdlib>

namespace {

constexpr int TB = 256;
constexpr int TILE = 64;      // rows/cols of the pair tile per workgroup
constexpr int DK = 32;        // feature chunk staged in LDS per iteration (forward)
constexpr int LDT = TILE + 1; // padded LDS leading dimension: 4 * LDT = 4 (mod 32), so the staging writes of a wave -- eight
                              // k-quads x eight rows, four words each -- fall on 32 different banks (TILE + 4: four-way conflicts)
constexpr int DC = 128;       // feature columns per workgroup in the backward kernel
constexpr int MAXQ = 8;       // kernel_num upper bound

struct Rows {
    const float* src; int64_t ld_src;
    const float* tgt; int64_t ld_tgt;
    const int64_t* src_idx; const int64_t* tgt_idx;   // [times, n] or NULL (rows stacked [times, n, d])
    int64_t n;                                         // rows per domain
    bool vec4;                                         // 16-byte loads legal
};

__device__ __forceinline__ const float* row_ptr(const Rows& R, int t, int64_t r) {
    if (r < R.n) {
        const int64_t g = R.src_idx ? R.src_idx[(int64_t)t * R.n + r] : (int64_t)t * R.n + r;
        return R.src + g * R.ld_src;
    }
    const int64_t q = r - R.n;
    const int64_t g = R.tgt_idx ? R.tgt_idx[(int64_t)t * R.n + q] : (int64_t)t * R.n + q;
    return R.tgt + g * R.ld_tgt;
}

// rows stacked [times, n, d] per domain (no index): the address is arithmetic only
__device__ __forceinline__ const float* row_direct(const Rows& R, int t, int64_t r) {
    return r < R.n ? R.src + ((int64_t)t * R.n + r) * R.ld_src : R.tgt + ((int64_t)t * R.n + (r - R.n)) * R.ld_tgt;
}

// four consecutive features k..k+3 of a row, zero beyond d / for a missing row
__device__ __forceinline__ float4 load4(const float* __restrict__ p, int64_t k, int64_t d, bool vec4) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!p) return v;
    if (vec4 && k + 3 < d) return *reinterpret_cast<const float4*>(p + k);
    if (k + 0 < d) v.x = p[k + 0];
    if (k + 1 < d) v.y = p[k + 1];
    if (k + 2 < d) v.z = p[k + 2];
    if (k + 3 < d) v.w = p[k + 3];
    return v;
}

// a - p where the row exists (missing rows stay all-zero)
__device__ __forceinline__ float4 sub4(float4 a, float4 p, bool live) {
    if (!live) return a;
    return make_float4(a.x - p.x, a.y - p.y, a.z - p.z, a.w - p.w);
}

__device__ __forceinline__ int64_t gda_cdiv_dev(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ double block_sum(double v, double* sh) {
    // fixed-order tree: wave shuffle, then 4 wave leaders through LDS
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0) for (int w = 0; w < TB / 64; ++w) s += sh[w];
    return s;   // valid on thread 0
}

// ------------------------------------------------------- bandwidth statistics --
struct KParams {
    float kernel_mul; int kernel_num; float fix_sigma;
};

constexpr int SR = 16;        // rows per k_rowstats workgroup (4 per thread: independent loads, many workgroups)

// chunk (t, blockIdx.x): s1 = sum over its rows of |t_i - p|^2, col[c] = sum over its rows of (t_i - p)[c]
__global__ void __launch_bounds__(TB)
k_rowstats(Rows R, int64_t d, int64_t m, double* __restrict__ part_s1, float* __restrict__ part_col,
           float* __restrict__ rows_src, float* __restrict__ rows_tgt, float* __restrict__ norms,
           float* __restrict__ part_max) {
    __shared__ double red[TB / 64];
    __shared__ float redmax[TB / 64];
    __shared__ float colsh[TB / 64][64];
    extern __shared__ float rs_tile[];                    // [SR][d + 1] shifted rows, only when `norms` is asked for
    const int64_t ldt = d + 1;
    const int t = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * SR;
    const int lane = threadIdx.x % 64, rg = threadIdx.x / 64;
    const float* pivot = row_ptr(R, t, 0);
    float* out = part_col + ((int64_t)t * gridDim.x + blockIdx.x) * d;
    float s1 = 0.f, mx = 0.f;
    for (int64_t c0 = 0; c0 < d; c0 += 64) {
        const int64_t c = c0 + lane;
        float col = 0.f;
        if (c < d) {
            const float pv = pivot[c];
            for (int rr = rg; rr < SR; rr += TB / 64) {
                const int64_t r = r0 + rr;
                if (r < m) {
                    const float raw = row_ptr(R, t, r)[c];
                    if (rows_src)                          // the gather rides along: rows stacked [times, n, d] per domain
                        (r < R.n ? rows_src + ((int64_t)t * R.n + r) * d : rows_tgt + ((int64_t)t * R.n + (r - R.n)) * d)[c] = raw;
                    const float v = raw - pv;
                    s1 = fmaf(v, v, s1);
                    mx = fmaxf(mx, fabsf(v));
                    col += v;
                    if (norms) rs_tile[rr * ldt + c] = v;
                }
            }
        }
        colsh[rg][lane] = col;
        __syncthreads();
        if (rg == 0 && c < d) out[c] = (colsh[0][lane] + colsh[1][lane]) + (colsh[2][lane] + colsh[3][lane]);
        __syncthreads();
    }
    // |t_i - p|^2 per row as ONE k-ascending fmaf chain: the chain the fp32 MFMA of k_pairdist builds for the dot
    // products, so for identical rows (sampling is with replacement) dot == norm bit for bit and their distance is
    // exactly 0, as in the reference's difference form.  (The last barrier of the column loop ordered the tile.)
    if (norms && threadIdx.x < SR && r0 + threadIdx.x < m) {
        const float* row = rs_tile + threadIdx.x * ldt;
        float acc = 0.f;
        for (int64_t k = 0; k < d; ++k) acc = fmaf(row[k], row[k], acc);
        norms[(int64_t)t * m + r0 + threadIdx.x] = acc;
    }
    if (part_max) {                                       // largest |t_i - p| entry of the chunk: the fp16 split's scale (fused path)
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_down(mx, off, 64));
        if (lane == 0) redmax[rg] = mx;
    }
    const double s = block_sum((double)s1, red);           // (its barriers order redmax as well)
    if (threadIdx.x == 0) {
        part_s1[(int64_t)t * gridDim.x + blockIdx.x] = s;
        if (part_max) part_max[(int64_t)t * gridDim.x + blockIdx.x] = fmaxf(fmaxf(redmax[0], redmax[1]), fmaxf(redmax[2], redmax[3]));
    }
}

// bandwidth[t] = (sum_ij L2 + 1e-6) / (m^2 - m) / kernel_mul^(kernel_num/2)     (mmd.py:50-51)
// One workgroup of 1024 threads per resample: a column's chunk partials are summed by EIGHT threads (chunks g, g + 8,
// ..., eight loads in flight each) whose double sums are added in group order -- a 125-deep chain of loads per
// column took 8 us on the step's critical path with 128 threads doing it alone.
constexpr int BW_TB = 1024, BW_G = 8, BW_C = BW_TB / BW_G;
__global__ void __launch_bounds__(BW_TB)
k_bandwidth(const double* __restrict__ part_s1, const float* __restrict__ part_col, int chunks, int64_t d,
            int64_t m, KParams kp, float* __restrict__ bandwidth) {
    __shared__ double colpart[BW_G][BW_C];
    __shared__ double red[2][BW_TB / 64];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int g = tid / BW_C, cl = tid % BW_C;
    double s2 = 0.0;
    for (int64_t c0 = 0; c0 < d; c0 += BW_C) {
        const int64_t c = c0 + cl;
        double cs = 0.0;
        if (c < d) {
            const float* pc = part_col + (int64_t)t * chunks * d + c;
            int k = g;
            for (; k + 7 * BW_G < chunks; k += 8 * BW_G) {     // eight loads in flight, summed in chunk order
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = pc[(int64_t)(k + u * BW_G) * d];
#pragma unroll
                for (int u = 0; u < 8; ++u) cs += (double)v[u];
            }
            for (; k < chunks; k += BW_G) cs += (double)pc[(int64_t)k * d];
        }
        colpart[g][cl] = cs;
        __syncthreads();
        if (g == 0 && c < d) {
            double tot = 0.0;
#pragma unroll
            for (int u = 0; u < BW_G; ++u) tot += colpart[u][cl];
            s2 += tot * tot;
        }
        __syncthreads();
    }
    double s1 = 0.0;
    for (int k = tid; k < chunks; k += BW_TB) s1 += part_s1[(int64_t)t * chunks + k];
    // both sums through one fixed-order tree: wave shuffles, then the 16 wave leaders
    for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_down(s1, off, 64); s2 += __shfl_down(s2, off, 64); }
    if (tid % 64 == 0) { red[0][tid / 64] = s1; red[1][tid / 64] = s2; }
    __syncthreads();
    if (tid == 0) {
        double S1 = 0.0, S2 = 0.0;
        for (int w = 0; w < BW_TB / 64; ++w) { S1 += red[0][w]; S2 += red[1][w]; }
        float bw;
        if (kp.fix_sigma > 0.f) bw = kp.fix_sigma;
        else {
            double tot = 2.0 * (double)m * S1 - 2.0 * S2;
            if (tot < 0.0) tot = 0.0;
            bw = ((float)tot + 1e-6f) / (float)(m * m - m);
        }
        float div = 1.f;
        for (int q = 0; q < kp.kernel_num / 2; ++q) div *= kp.kernel_mul;
        bandwidth[t] = bw / div;
    }
}

// ---------------------------------------------------------------- forward --
using f32x16 = __attribute__((ext_vector_type(16))) float;

// L2[i,j] = (|t_i|^2 + |t_j|^2) - 2 t_i.t_j over the pivot-shifted rows on the fp32 matrix cores: 64x64 tile per workgroup,
// one 32x32 sub-tile per wave, feature chunks of 32 staged k-major in LDS so that both MFMA
// operands are conflict-free ds_read_b32 (A: lane l -> row l&31, k = l>>5; B likewise).  The
// fp32 MFMA is an exact k-ordered fma chain, so the Gram form costs ~1e-7 relative on L2 (far
// inside the 1e-4 loss tolerance) for half the VALU work of the difference form and none of it
// on the VALU.  Tile sums feed the bandwidth (mmd.py:50).
// The matrix is symmetric: only the tiles (I, J) with I <= J are computed (half the matrix-core work and half
// the exponentials); an off-diagonal tile is stored twice, as it is and transposed -- the transposed store is
// the MFMA accumulator's natural 16-byte direction (four consecutive i per lane) -- and counts twice in the
// block sums.  SQ: kernel_mul == 2 with five kernels (every pygda call): the five exponentials
// exp(-L2 / (bw 2^q)) are one __expf and four squarings (e_q = e_{q+1}^2).
// FAST (rows stacked without an index, 16-byte aligned, d a multiple of the 32-feature chunk, row norms left by
// k_rowstats): the staging loads are branch-free (row index clamped, the piece zeroed on its way into LDS), those of
// chunk k + 1 are issued before chunk k's MFMAs, and no thread walks a norm chain between the barrier and the MFMAs.
template <int KN, bool SQ, bool FAST>
// round 4: asked for eight workgroups per CU the compiler fits the kernel into 63 registers, accumulators included, without a
// spill (unhinted: 80 + 16 = five waves per SIMD) -- this kernel lives on occupancy (see the 128-tile experiment, profiles/HISTORY.md 4.3)
__global__ void __launch_bounds__(TB, 8)
k_pairdist(Rows R, int64_t d, int64_t m, int nt, const float* __restrict__ bandwidth, KParams kp,
           float* __restrict__ l2, double* __restrict__ partial, const float* __restrict__ norms) {
    __shared__ __attribute__((aligned(16))) float As[DK][LDT];   // As[k][row i of the tile]
    __shared__ __attribute__((aligned(16))) float Bs[DK][LDT];   // Bs[k][row j of the tile]
    __shared__ float nA[TILE], nB[TILE];                          // |t_i|^2, |t_j|^2 of the tile rows
    __shared__ double red[TB / 64];
    const int t = blockIdx.z;
    // linear index -> (I, J), I <= J, rows of the upper triangle laid end to end
    int I, J;
    {
        const int q = (int)blockIdx.x;
        const float b = (float)(2 * nt + 1);
        I = (int)((b - sqrtf(b * b - 8.f * (float)q)) * 0.5f);
        I = I < 0 ? 0 : (I > nt - 1 ? nt - 1 : I);
        while (I > 0 && q < I * nt - I * (I - 1) / 2) --I;
        while (q >= (I + 1) * nt - (I + 1) * I / 2) ++I;
        J = I + (q - (I * nt - I * (I - 1) / 2));
    }
    const int64_t i0 = (int64_t)I * TILE, j0 = (int64_t)J * TILE;
    const int tid = threadIdx.x, wave = tid / 64, lane = tid % 64;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;         // this wave's 32x32 sub-tile
    const int ka = lane >> 5, la = lane & 31;

    // staging: this thread moves rows (tid/8) and (tid/8 + 32) of both tiles, features kq*4..+3
    const int lr = tid / 8, kq = (tid % 8) * 4;
    const float* pivot = row_ptr(R, t, 0);               // common shift of every row of this resample
    const float* pa[2]; const float* pb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int64_t ri = i0 + lr + 32 * q, rj = j0 + lr + 32 * q;
        pa[q] = ri < m ? row_ptr(R, t, ri) : nullptr;
        pb[q] = rj < m ? row_ptr(R, t, rj) : nullptr;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // Row norms as ONE k-ascending fmaf chain per row, walked on the staged tiles by threads 0..63
    // (A rows) and 64..127 (B rows) -- the chain the fp32 MFMA builds for the dot products, so for
    // identical rows (sampling is with replacement) dot == norm bit for bit and their distance is
    // exactly 0, as in the reference's difference form.  (Was a separate kernel: one thread per row
    // striding through global memory, 20 us on the critical path.)
    float nacc = 0.f;
    float (*Sn)[LDT] = tid < TILE ? As : Bs;
    const int nrow = tid & (TILE - 1);

    if constexpr (FAST) {
        const bool oka[2] = {i0 + lr < m, i0 + lr + 32 < m}, okb[2] = {j0 + lr < m, j0 + lr + 32 < m};
        const float* qa[2]; const float* qb[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int64_t ri = i0 + lr + 32 * q, rj = j0 + lr + 32 * q;
            qa[q] = row_direct(R, t, ri < m ? ri : m - 1) + kq;
            qb[q] = row_direct(R, t, rj < m ? rj : m - 1) + kq;
        }
        if (tid < TILE) nA[tid] = i0 + tid < m ? norms[(int64_t)t * m + i0 + tid] : 0.f;
        else if (tid < 2 * TILE) nB[tid - TILE] = j0 + tid - TILE < m ? norms[(int64_t)t * m + j0 + tid - TILE] : 0.f;
        float4 va[2], vb[2], pv;
        pv = *reinterpret_cast<const float4*>(pivot + kq);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            va[q] = *reinterpret_cast<const float4*>(qa[q]);
            vb[q] = *reinterpret_cast<const float4*>(qb[q]);
        }
        for (int64_t k0 = 0; k0 < d; k0 += DK) {
            __syncthreads();                                   // the previous chunk's MFMAs have read their operands
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = lr + 32 * q;
                As[kq + 0][r] = oka[q] ? va[q].x - pv.x : 0.f; As[kq + 1][r] = oka[q] ? va[q].y - pv.y : 0.f;
                As[kq + 2][r] = oka[q] ? va[q].z - pv.z : 0.f; As[kq + 3][r] = oka[q] ? va[q].w - pv.w : 0.f;
                Bs[kq + 0][r] = okb[q] ? vb[q].x - pv.x : 0.f; Bs[kq + 1][r] = okb[q] ? vb[q].y - pv.y : 0.f;
                Bs[kq + 2][r] = okb[q] ? vb[q].z - pv.z : 0.f; Bs[kq + 3][r] = okb[q] ? vb[q].w - pv.w : 0.f;
            }
            if (k0 + DK < d) {                                 // next chunk: in flight under this chunk's MFMAs
                pv = *reinterpret_cast<const float4*>(pivot + k0 + DK + kq);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    va[q] = *reinterpret_cast<const float4*>(qa[q] + k0 + DK);
                    vb[q] = *reinterpret_cast<const float4*>(qb[q] + k0 + DK);
                }
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < DK; kk += 2)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[kk + ka][wi + la], Bs[kk + ka][wj + la], acc, 0, 0, 0);
        }
        __syncthreads();
    } else {
    // single-buffered on purpose: at 17 KB of LDS nine workgroups share a CU and hide each
    // other's staging; a double-buffered variant (35 KB, four workgroups) measured 8 % slower
    for (int64_t k0 = 0; k0 < d; k0 += DK) {
        float4 va[2], vb[2];
        const float4 pv = load4(pivot, k0 + kq, d, R.vec4);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            va[q] = sub4(load4(pa[q], k0 + kq, d, R.vec4), pv, pa[q] != nullptr);
            vb[q] = sub4(load4(pb[q], k0 + kq, d, R.vec4), pv, pb[q] != nullptr);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = lr + 32 * q;
            As[kq + 0][r] = va[q].x; As[kq + 1][r] = va[q].y; As[kq + 2][r] = va[q].z; As[kq + 3][r] = va[q].w;
            Bs[kq + 0][r] = vb[q].x; Bs[kq + 1][r] = vb[q].y; Bs[kq + 2][r] = vb[q].z; Bs[kq + 3][r] = vb[q].w;
        }
        __syncthreads();
        if (tid < 2 * TILE) {
#pragma unroll
            for (int kk = 0; kk < DK; ++kk) nacc = fmaf(Sn[kk][nrow], Sn[kk][nrow], nacc);
        }
#pragma unroll
        for (int kk = 0; kk < DK; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[kk + ka][wi + la], Bs[kk + ka][wj + la], acc, 0, 0, 0);
    }
    if (tid < TILE) nA[nrow] = nacc;
    else if (tid < 2 * TILE) nB[nrow] = nacc;
    __syncthreads();
    }

    // Epilogue: the distances never leave the registers.  K = sum_q exp(L2 * (-1/bw_q)) feeds the signed block
    // sums of the loss; what is stored is the backward's weight g[i,j] = +-sum_q exp(.) * (-1/bw_q) = dK/dL2,
    // so the backward pass is a pure matrix product and evaluates no exponentials.
    const int kn = KN > 0 ? KN : kp.kernel_num;
    float nib[KN > 0 ? KN : MAXQ];                     // -1 / (bandwidth * kernel_mul^q), mmd.py:52
    {
        const float bw0 = bandwidth[t];
        float f = 1.f;
#pragma unroll
        for (int q = 0; q < kn; ++q) { nib[q] = -1.f / (bw0 * f); f *= kp.kernel_mul; }
    }
    float* out = l2 + (int64_t)t * m * m;
    const int64_t j = j0 + wj + la;
    const float nj = nB[wj + la];
    const int64_t n = R.n;
    const bool offdiag = I != J;
    const bool vecT = (m % 4 == 0) && (((uintptr_t)out & 15) == 0);
    float local = 0.f;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        float gv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = g4 * 4 + e;
            const int li = wi + e + 8 * g4 + 4 * ka;                      // C/D layout of the 32x32 MFMA
            const int64_t i = i0 + li;
            gv[e] = 0.f;
            if (i < m && j < m) {
                float v = (nA[li] + nj) - 2.f * acc[r];
                v = v < 0.f ? 0.f : v;                                 // a sum of squares; NaN stays NaN
                float kv = 0.f, dk = 0.f;
                if constexpr (SQ) {
                    const float e4 = __expf(v * nib[4]);
                    const float e3 = e4 * e4, e2 = e3 * e3, e1 = e2 * e2, e0 = e1 * e1;
                    kv = (((e0 + e1) + e2) + e3) + e4;
                    dk = fmaf(e4, nib[4], fmaf(e3, nib[3], fmaf(e2, nib[2], fmaf(e1, nib[1], e0 * nib[0]))));
                } else {
#pragma unroll
                    for (int q = 0; q < kn; ++q) { const float ex = __expf(v * nib[q]); kv += ex; dk = fmaf(ex, nib[q], dk); }
                }
                const bool same = (i < n) == (j < n);
                local += same ? kv : -kv;
                gv[e] = same ? dk : -dk;
                out[i * m + j] = gv[e];
            }
        }
        if (offdiag && j < m) {                                        // the mirror tile: out[j][i..i+3]
            const int64_t ib = i0 + wi + 8 * g4 + 4 * ka;
            if (vecT && ib + 3 < m) *reinterpret_cast<float4*>(out + j * m + ib) = make_float4(gv[0], gv[1], gv[2], gv[3]);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (ib + e < m) out[j * m + ib + e] = gv[e];
            }
        }
    }
    if (offdiag) local *= 2.f;
    const double s = block_sum((double)local, red);
    if (tid == 0) partial[(int64_t)t * gridDim.x + blockIdx.x] = s;
}

__global__ void __launch_bounds__(TB)
k_finalize(const double* __restrict__ kpartial, int tiles_per_t, int times, int64_t n,
           float scale, const float* __restrict__ add, float* __restrict__ loss) {
    // every resample's tile sums in ONE pass: per-thread partials of all resamples, one shuffle tree over the lot, one
    // barrier (a block sum per resample, one after the other, was 5 x 1.3 us in front of the backward pass)
    constexpr int MAXT = 8;
    __shared__ double red[TB / 64][MAXT];
    float total = 0.f;
    for (int t0 = 0; t0 < times; t0 += MAXT) {
        const int nt_ = times - t0 < MAXT ? times - t0 : MAXT;
        double v[MAXT];
#pragma unroll
        for (int u = 0; u < MAXT; ++u) {
            v[u] = 0.0;
            if (u < nt_)
                for (int k = threadIdx.x; k < tiles_per_t; k += TB) v[u] += kpartial[(int64_t)(t0 + u) * tiles_per_t + k];
        }
#pragma unroll
        for (int u = 0; u < MAXT; ++u)
            for (int off = 32; off > 0; off >>= 1) v[u] += __shfl_down(v[u], off, 64);
        __syncthreads();
        if (threadIdx.x % 64 == 0)
#pragma unroll
            for (int u = 0; u < MAXT; ++u) red[threadIdx.x / 64][u] = v[u];
        __syncthreads();
        if (threadIdx.x == 0)
            for (int u = 0; u < nt_; ++u) {
                double sres = 0.0;
                for (int w = 0; w < TB / 64; ++w) sres += red[w][u];
                total += (float)(sres / ((double)n * (double)n));            // mmd.py:106 mean
            }
    }
    if (threadIdx.x == 0) {                            // mmd.py:157; `add + scale * mmd` in one go when asked for
        const float v = total / (float)times;
        loss[0] = (add ? add[0] : 0.f) + (scale != 1.f ? scale * v : v);
    }
}

// --------------------------------------------------------------- backward --
// With the weights g = dK/dL2 (block-signed) written by the epilogue of k_pairdist, the
// gradient w.r.t. the sampled rows is a matrix product plus a row scaling,
//     grad_total[i,:] = 4 c ((sum_j g[i,j]) total[i,:] - sum_j g[i,j] total[j,:]),   c = dloss / (n^2 times),
// run on the fp32 matrix cores.  Workgroup tile: 64 rows x 128 feature columns; wave w owns rows
// (w>>1)*32.. and two 32-column MFMA tiles at (w&1)*64.  The j dimension is cut into NSEG segments
// (deterministic two-stage sum, no atomics) and walked in chunks of 32 rows with double-buffered
// LDS: the global loads of chunk c+1 are in flight while the 32 MFMAs of chunk c issue.
// g is symmetric, so the A operand tile is read as g[j][i]: 16-byte loads along i, stored k-major,
// and every MFMA operand fetch is a conflict-free ds_read_b32.
constexpr int BI = 64;        // rows i per workgroup (default variant)
constexpr int BJ = 32;        // rows j per chunk (MFMA K = 2 per instruction)

// BI_ x 128 workgroup tile, BI_ / 32 * 2 wavefronts (64 rows: 4 waves, 128 rows: 8 waves -- the T chunk is then shared
// by twice the rows: 0.25 KB of staging per row and chunk instead of 0.375, four waves per SIMD with two workgroups
// per CU).  Dynamic LDS: [Gs 2 x BJ x BI_][Ts 2 x BJ x DC][rowsum BI_].
//
// FAST (m % 4 == 0, 16-byte aligned rows, d a multiple of the 128-column block: every A2GNN call): the staging loads
// are branch-free -- row / column indices clamped into the matrix, out-of-range pieces zeroed when they are written
// to LDS -- and nothing consumes them before the chunk's MFMAs have been issued (the pivot shift happens on the way
// into LDS), so a chunk's global loads really are in flight under the previous chunk's matrix work; the generic
// variant keeps the guarded loads.  Both read a chunk's MFMA operands from LDS ahead of the MFMAs that use them
// (round 3: with read -> wait -> two MFMAs the loop ran at 56 % of the matrix pipe with no global load in it,
// 69 us of the kernel's 82).
template <bool FAST, int BI_>
__global__ void __launch_bounds__(BI_ * 4, BI_ == 128 ? 4 : 3)
k_bwd(Rows R, int64_t d, int64_t m, const float* __restrict__ l2, const float* __restrict__ grad_loss, float scale,
      int times, int nseg, float* __restrict__ part) {
    constexpr int TB_ = BI_ * 4;
    constexpr int GC = BI_ / 4;                   // float4 per G row
    constexpr int TQ = (BJ * DC / 4) / TB_;       // T float4 per thread and chunk
    constexpr int TSTEP = TB_ / (DC / 4);         // rows between a thread's T pieces
    static_assert(TB_ / GC == 16 && BJ == 32, "two G rows per thread");
    extern __shared__ __attribute__((aligned(16))) char bwd_lds[];
    float (*Gs)[BJ][BI_] = reinterpret_cast<float (*)[BJ][BI_]>(bwd_lds);                       // Gs[b][j][i] = g[i][j]
    float (*Ts)[BJ][DC] = reinterpret_cast<float (*)[BJ][DC]>(bwd_lds + sizeof(float) * 2 * BJ * BI_);   // rows j of total
    float* rowsum = reinterpret_cast<float*>(bwd_lds + sizeof(float) * 2 * BJ * (BI_ + DC));
    const int t = blockIdx.z;
    const int seg = blockIdx.y % nseg;
    const int64_t c0 = (int64_t)(blockIdx.y / nseg) * DC;
    const int64_t i0 = (int64_t)blockIdx.x * BI_;
    const int tid = threadIdx.x, wave = tid / 64, lane = tid % 64;
    const int ka = lane >> 5, la = lane & 31;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 64;
    const int64_t n = R.n;
    const float* L = l2 + (int64_t)t * m * m;
    const bool g_vec = (m % 4 == 0);

    // staging roles: G chunk = 32 rows x GC float4 (2 per thread), T chunk = 32 rows x 32 float4 (TQ per thread)
    const int g_c4 = (tid % GC) * 4, g_r = tid / GC;       // rows g_r and g_r + 16
    const int t_c4 = (tid % 32) * 4, t_r = tid / 32;       // rows t_r, t_r + TSTEP, ...
    // the same pivot shift as the forward: sum_j g_ij (t_i - t_j) is unchanged by it, and the two
    // products it is computed from no longer carry the batch's common offset
    const float* pivot = row_ptr(R, t, 0);
    const float4 pv = load4(pivot, c0 + t_c4, d, R.vec4);

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float rsum = 0.f;                             // sum over this lane's j (j = ka mod 2) of g[wr + la][j]

    const int64_t nchunks = gda_cdiv_dev(m, BJ);
    // Two register sets of staged pieces: the loads of chunk k + 3 are issued when chunk k + 1 has been written to
    // LDS, and are consumed two chunks of matrix work later -- the G tiles (80 MB per call) come from the Infinity
    // Cache / HBM at 2-3 us under load, more than one chunk's MFMAs cover (one set, one chunk ahead: 73 us; the
    // loads alone 42, the MFMA loop alone 36).
    float4 gq0[2], tq0[TQ], gq1[2], tq1[TQ];
#define BW_FETCH(GQ, TQV, CH)                                                                                     \
    {                                                                                                             \
        const int64_t j0_ = (CH) * BJ;                                                                            \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                           \
            const int64_t j = j0_ + g_r + 16 * q, i = i0 + g_c4;                                                  \
            if constexpr (FAST) {                                                                                 \
                GQ[q] = *reinterpret_cast<const float4*>(L + (j < m ? j : m - 1) * m + (i < m ? i : m - 4));      \
            } else {                                                                                              \
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                       \
                if (j < m) {                                                                                      \
                    const float* p = L + j * m + i;                                                               \
                    if (g_vec && i + 3 < m) v = *reinterpret_cast<const float4*>(p);                              \
                    else {                                                                                        \
                        if (i + 0 < m) v.x = p[0];                                                                \
                        if (i + 1 < m) v.y = p[1];                                                                \
                        if (i + 2 < m) v.z = p[2];                                                                \
                        if (i + 3 < m) v.w = p[3];                                                                \
                    }                                                                                             \
                }                                                                                                 \
                GQ[q] = v;                                                                                        \
            }                                                                                                     \
        }                                                                                                         \
        _Pragma("unroll") for (int q = 0; q < TQ; ++q) {                                                          \
            const int64_t j = j0_ + t_r + TSTEP * q;                                                              \
            if constexpr (FAST) TQV[q] = *reinterpret_cast<const float4*>(row_direct(R, t, j < m ? j : m - 1) + c0 + t_c4); \
            else TQV[q] = sub4(load4(j < m ? row_ptr(R, t, j) : nullptr, c0 + t_c4, d, R.vec4), pv, j < m);       \
        }                                                                                                         \
    }
#define BW_STASH(GQ, TQV, B, CH)                                                                                  \
    {                                                                                                             \
        const int64_t j0_ = (CH) * BJ;                                                                            \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                           \
            float4 v = GQ[q];                                                                                     \
            if constexpr (FAST) {                                                                                 \
                const bool ok = j0_ + g_r + 16 * q < m && i0 + g_c4 < m;                                          \
                v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;           \
            }                                                                                                     \
            *reinterpret_cast<float4*>(&Gs[B][g_r + 16 * q][g_c4]) = v;                                           \
        }                                                                                                         \
        _Pragma("unroll") for (int q = 0; q < TQ; ++q) {                                                          \
            float4 v = TQV[q];                                                                                    \
            if constexpr (FAST) {                                                                                 \
                const bool ok = j0_ + t_r + TSTEP * q < m;                                                        \
                v.x = ok ? v.x - pv.x : 0.f; v.y = ok ? v.y - pv.y : 0.f;                                         \
                v.z = ok ? v.z - pv.z : 0.f; v.w = ok ? v.w - pv.w : 0.f;                                         \
            }                                                                                                     \
            *reinterpret_cast<float4*>(&Ts[B][t_r + TSTEP * q][t_c4]) = v;                                        \
        }                                                                                                         \
    }
    // operands of group g + 1 (4 k-pairs: 12 LDS words per lane) are read while the 8 MFMAs of group g issue;
    // the scheduling barriers keep the compiler from sinking every read to just in front of its MFMA again
#define BW_LOAD(g)                                                                                   \
        _Pragma("unroll") for (int u = 4 * (g); u < 4 * (g) + 4; ++u) {                              \
            av[u] = Gs[buf][2 * u + ka][wr + la];                                                    \
            b0[u] = Ts[buf][2 * u + ka][wc + la];                                                    \
            b1[u] = Ts[buf][2 * u + ka][wc + 32 + la];                                               \
        }
#define BW_MFMA(g)                                                                                   \
        _Pragma("unroll") for (int u = 4 * (g); u < 4 * (g) + 4; ++u) {                              \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], b0[u], acc0, 0, 0, 0);                \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], b1[u], acc1, 0, 0, 0);                \
            rsum += av[u];                                                                           \
        }
#define BW_COMPUTE()                                                                                 \
    {                                                                                                \
        float av[BJ / 2], b0[BJ / 2], b1[BJ / 2];                                                    \
        BW_LOAD(0) __builtin_amdgcn_sched_barrier(0);                                                \
        BW_LOAD(1) BW_MFMA(0) __builtin_amdgcn_sched_barrier(0);                                     \
        BW_LOAD(2) BW_MFMA(1) __builtin_amdgcn_sched_barrier(0);                                     \
        BW_LOAD(3) BW_MFMA(2) __builtin_amdgcn_sched_barrier(0);                                     \
        BW_MFMA(3)                                                                                   \
    }

    // this segment's chunks: seg, seg + nseg, ...; chunk k + 1 waits in set (k mod 2) while chunk k is multiplied
    int64_t ch = seg;
    const int64_t step = nseg;
    int buf = 0;
    if (ch < nchunks) BW_FETCH(gq1, tq1, ch)
    if (ch + step < nchunks) BW_FETCH(gq0, tq0, ch + step)
    if (ch < nchunks) BW_STASH(gq1, tq1, 0, ch)
    if (ch + 2 * step < nchunks) BW_FETCH(gq1, tq1, ch + 2 * step)
    __syncthreads();
    while (ch < nchunks) {
        BW_COMPUTE()
        if (ch + step < nchunks) BW_STASH(gq0, tq0, buf ^ 1, ch + step)
        if (ch + 3 * step < nchunks) BW_FETCH(gq0, tq0, ch + 3 * step)
        __syncthreads();
        buf ^= 1;
        ch += step;
        if (ch >= nchunks) break;
        BW_COMPUTE()
        if (ch + step < nchunks) BW_STASH(gq1, tq1, buf ^ 1, ch + step)
        if (ch + 3 * step < nchunks) BW_FETCH(gq1, tq1, ch + 3 * step)
        __syncthreads();
        buf ^= 1;
        ch += step;
    }
#undef BW_FETCH
#undef BW_STASH
#undef BW_LOAD
#undef BW_MFMA
#undef BW_COMPUTE

    // row sums of g over this segment: the A operands a wave walked ARE its 32 rows' entries -- the two lane halves
    // (even / odd j) added, in a fixed order
    rsum += __shfl_xor(rsum, 32, 64);
    if ((wave & 1) == 0 && lane < 32) rowsum[wr + la] = rsum;
    __syncthreads();

    // epilogue: part[t][seg][i][c] = c * (rowsum[i] * total[i][c] - acc)   (factor 4 in the reduce)
    const float coef = grad_loss[0] * scale / ((float)n * (float)n) / (float)times;
    float* out = part + (((int64_t)t * nseg + seg) * m) * d;
    if constexpr (FAST) {
        // the 32 row values a lane needs are 32 INDEPENDENT loads (row index clamped, no branch around them): one
        // memory round trip for the lot.  Guarded one by one -- load, wait, store, next -- they were ~30 us of the
        // kernel: every workgroup reaches its epilogue at the same time (the grid is one round of workgroups)
        const float pv0 = pivot[c0 + wc + la], pv1 = pivot[c0 + wc + 32 + la];
        float tiv[2][16];
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t i = i0 + wr + (r & 3) + 8 * (r >> 2) + 4 * ka;
                tiv[half][r] = row_direct(R, t, i < m ? i : m - 1)[c0 + wc + 32 * half + la];
            }
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int il = wr + (r & 3) + 8 * (r >> 2) + 4 * ka;     // C/D layout of the 32x32 MFMA
                const int64_t i = i0 + il;
                const float ti = tiv[half][r] - (half == 0 ? pv0 : pv1);
                const float a = half == 0 ? acc0[r] : acc1[r];
                if (i < m) out[i * d + c0 + wc + 32 * half + la] = coef * fmaf(rowsum[il], ti, -a);
            }
        return;
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int64_t c = c0 + wc + 32 * half + la;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int il = wr + (r & 3) + 8 * (r >> 2) + 4 * ka;     // C/D layout of the 32x32 MFMA
            const int64_t i = i0 + il;
            if (i < m && c < d) {
                const float ti = row_ptr(R, t, i)[c] - pivot[c];
                const float a = half == 0 ? acc0[r] : acc1[r];
                out[i * d + c] = coef * fmaf(rowsum[il], ti, -a);
            }
        }
    }
}

// grad_rows = 4 * sum over segments (fixed order)
// the same fold for partials whose rows are padded to ldp >= d floats (the chunked one-pass kernel): grad_rows is [times, m, d]
__global__ void __launch_bounds__(TB)
k_bwd_reduce_ld(const float* __restrict__ part, int64_t m, int64_t d, int64_t ldp, int nseg, int times,
                float* __restrict__ grad_rows, const float* __restrict__ grad_loss, float cmul) {
    const float c4 = grad_loss ? 4.f * (grad_loss[0] * cmul) : 4.f;
    const int64_t total = m * d * times;
    for (int64_t k = (int64_t)blockIdx.x * TB + threadIdx.x; k < total; k += (int64_t)gridDim.x * TB) {
        const int64_t t = k / (m * d), r = k % (m * d), i = r / d, c = r % d;
        float s = 0.f;
        for (int g = 0; g < nseg; ++g) s += part[((t * nseg + g) * m + i) * ldp + c];
        grad_rows[k] = c4 * s;
    }
}

__global__ void __launch_bounds__(TB)
k_bwd_reduce(const float* __restrict__ part, int64_t per_t, int nseg, int times,
             float* __restrict__ grad_rows, const float* __restrict__ grad_loss, float cmul) {
    const float c4 = grad_loss ? 4.f * (grad_loss[0] * cmul) : 4.f;       // see k_bwd_scatter
    const int64_t total = per_t * times;
    for (int64_t k = (int64_t)blockIdx.x * TB + threadIdx.x; k < total; k += (int64_t)gridDim.x * TB) {
        const int64_t t = k / per_t, r = k % per_t;
        float s = 0.f;
        for (int g = 0; g < nseg; ++g) s += part[(t * nseg + g) * per_t + r];
        grad_rows[k] = c4 * s;
    }
}

// Segment reduce AND scatter onto the sampled feature rows in one pass: feature row r of a domain receives
//     sum over its selection entries p (positions t*m + i in the [times, m, d] row-gradient array, CSR order)
//         of 4 * (sum over the nseg segment partials of position p, in segment order)
// -- the values k_bwd_reduce followed by the selection-matrix SpMM produce, bit for bit; rows nobody sampled get 0.
template <int NSEG>
__global__ void __launch_bounds__(TB)
k_bwd_scatter(const float* __restrict__ part, int64_t m, int64_t d, int64_t ldp, bool vec_out, int nseg_rt,
              const int32_t* __restrict__ s_rowptr, const int32_t* __restrict__ s_col, int64_t n_src_rows,
              float* __restrict__ gsrc, const int32_t* __restrict__ t_rowptr, const int32_t* __restrict__ t_col,
              int64_t n_tgt_rows, float* __restrict__ gtgt, const float* __restrict__ grad_loss, float cmul,
              const float* __restrict__ mask_s = nullptr, float mscale_s = 1.f,
              const float* __restrict__ mask_t = nullptr, float mscale_t = 1.f) {
    // mask_* (may be NULL): the OUTPUT y of the activation dropout(relu(.)) that produced this domain's feature rows -- the
    // row gradient is then written as y > 0 ? g * mscale : 0, i.e. already through that activation's backward
    // (gda_relu_dropout_bwd_f32's values bit for bit), and the activation's own backward launch disappears.
    // the fused forward leaves UNSCALED partials: 4 * dloss * cmul is applied here (grad_loss NULL: k_bwd scaled them, 4 is left)
    const float c4 = grad_loss ? 4.f * (grad_loss[0] * cmul) : 4.f;
    const int nseg = NSEG > 0 ? NSEG : nseg_rt;
    const int64_t rows_per_block = TB / 32;               // 32 lanes x float4 = one 128-wide row slab
    const int lane = threadIdx.x % 32;
    const int64_t r = (int64_t)blockIdx.x * rows_per_block + threadIdx.x / 32;
    const bool tgt = blockIdx.y == 1;
    const int64_t n_rows = tgt ? n_tgt_rows : n_src_rows;
    if (r >= n_rows) return;
    const int32_t* rp = tgt ? t_rowptr : s_rowptr;
    const int32_t* ci = tgt ? t_col : s_col;
    float* out = (tgt ? gtgt : gsrc) + r * d;
    const float* mk = tgt ? mask_t : mask_s;
    const float msc = tgt ? mscale_t : mscale_s;
    if (mk) mk += r * d;
    const int32_t b = rp[r], e = rp[r + 1];
    if constexpr (NSEG > 0) {
        // ldp % 4 == 0 and a 16-byte aligned partial array (the launcher checks): the NSEG partial quads of an entry are NSEG
        // independent 16-byte loads in flight at once (the generic loop below issues one dependent 4-byte load after
        // the other: 34 us at the A2GNN shapes against 8) -- added in the same segment order, entry after entry
        const int mi = (int)m;
        for (int64_t c = (int64_t)lane * 4; c < d; c += 128) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int32_t pn = b < e ? ci[b] : 0;
            for (int32_t k = b; k < e; ++k) {
                const int p = pn;
                if (k + 1 < e) pn = ci[k + 1];
                const int t = p / mi, i = p - t * mi;
                const float* q = part + (((int64_t)t * NSEG) * m + i) * ldp + c;
                float4 v[NSEG];
#pragma unroll
                for (int g = 0; g < NSEG; ++g) v[g] = *reinterpret_cast<const float4*>(q + (int64_t)g * m * ldp);
                float4 sg = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int g = 0; g < NSEG; ++g) { sg.x += v[g].x; sg.y += v[g].y; sg.z += v[g].z; sg.w += v[g].w; }
                acc.x = __fadd_rn(acc.x, c4 * sg.x); acc.y = __fadd_rn(acc.y, c4 * sg.y);
                acc.z = __fadd_rn(acc.z, c4 * sg.z); acc.w = __fadd_rn(acc.w, c4 * sg.w);
            }
            if (!vec_out) {                                        // output rows of any width / alignment (d = 645: GRADE)
                const float a4[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
                for (int v2 = 0; v2 < 4; ++v2)
                    if (c + v2 < d) out[c + v2] = mk ? (mk[c + v2] > 0.f ? a4[v2] * msc : 0.f) : a4[v2];
                continue;
            }
            if (mk) {
                const float4 y = *reinterpret_cast<const float4*>(mk + c);
                acc.x = y.x > 0.f ? acc.x * msc : 0.f; acc.y = y.y > 0.f ? acc.y * msc : 0.f;
                acc.z = y.z > 0.f ? acc.z * msc : 0.f; acc.w = y.w > 0.f ? acc.w * msc : 0.f;
            }
            *reinterpret_cast<float4*>(out + c) = acc;
        }
        return;
    }
    for (int64_t c = (int64_t)lane * 4; c < d; c += 128) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int32_t k = b; k < e; ++k) {
            const int64_t p = ci[k], t = p / m, i = p % m;
            float sg[4] = {0.f, 0.f, 0.f, 0.f};
            for (int g = 0; g < nseg; ++g) {
                const float* q = part + (((int64_t)t * nseg + g) * m + i) * ldp + c;
#pragma unroll
                for (int v = 0; v < 4; ++v) if (c + v < d) sg[v] += q[v];
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[v] = __fadd_rn(acc[v], __fmul_rn(1.0f, c4 * sg[v]));
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) if (c + v < d) out[c + v] = mk ? (mk[c + v] > 0.f ? acc[v] * msc : 0.f) : acc[v];
    }
}

constexpr int BWD_NSEG = 4;        // 64-row tiles, measured: 2 -> 141+5 us, 4 -> 97+7, 8 -> 102+10, 16 -> 119+18 (k_bwd + k_bwd_reduce)
constexpr int BWD_NSEG_MAX = 8;    // workspace bound for the segment count of any variant

// Row tile and j-segment count of k_bwd.  Measured at the A2GNN shapes (times = 5, m = 2000, d = 128; k_bwd + the fold
// behind it, us): 64 rows x 4 segments 102.6 + 6.7 (round 1's choice), 128 x 4 109.5 + 6.6, 128 x 5 93.4 + 7.6,
// 128 x 6 83.0 + 8.4, 128 x 8 100.1 + 10.3 -- 16 row tiles x 6 segments x 5 resamples = 480 workgroups of 8 waves,
// two per CU on 240 of the 256 CUs at once; the 128-row tile shares every staged T chunk between twice the rows.
// PYGDA_AMD_MMD_BWD_TILE / PYGDA_AMD_MMD_BWD_NSEG override (experiments).
struct BwdVariant { int tile, nseg; };
BwdVariant bwd_variant(int64_t m) {
    static const BwdVariant env = [] {
        BwdVariant r{0, 0};
        if (const char* e = std::getenv("PYGDA_AMD_MMD_BWD_TILE")) r.tile = std::atoi(e) == 64 ? 64 : (std::atoi(e) == 128 ? 128 : 0);
        if (const char* e = std::getenv("PYGDA_AMD_MMD_BWD_NSEG")) { const int k = std::atoi(e); if (k >= 1 && k <= BWD_NSEG_MAX) r.nseg = k; }
        return r;
    }();
    BwdVariant v = m >= 512 ? BwdVariant{128, 6} : BwdVariant{64, BWD_NSEG};
    if (env.tile) { v.tile = env.tile; v.nseg = env.tile == 128 ? 6 : BWD_NSEG; }
    if (env.nseg) v.nseg = env.nseg;
    return v;
}

template <bool FAST, int BI_>
int launch_bwd(dim3 grid, hipStream_t stream, Rows R, int64_t d, int64_t m, const float* l2, const float* grad_loss, float scale,
               int times, int nseg, float* part) {
    const size_t lds = sizeof(float) * (2 * BJ * (BI_ + DC) + BI_);
    GDA_LDS_ATTR_ONCE((k_bwd<FAST, BI_>), lds);
    k_bwd<FAST, BI_><<<grid, BI_ * 4, lds, stream>>>(R, d, m, l2, grad_loss, scale, times, nseg, part);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

#include "gda_mmd_fused.inc"
#include "gda_mmd_chunked.inc"

struct MmdWs {
    double* kpartial; double* part_s1; float* part_col; float* bwd_part; float* norms;
    float* part_max; unsigned char* images;      // the fused pass (NULL / empty when it does not cover the shape)
    size_t total;
};

MmdWs carve(void* base, int times, int64_t n, int64_t d) {
    const int64_t m = 2 * n, nt = gda_cdiv(m, TILE);
    MmdWs w{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void* p = base ? (void*)((char*)base + off) : nullptr;
        off += gda_align_up(bytes, 256);
        return p;
    };
    const int64_t chunks = gda_cdiv(m, SR);
    FusedPlan fp;
    const bool fused = fused_plan(times, n, d, 2.0f, 5, &fp);
    const int64_t per_t = fused && (int64_t)fp.njb * fp.nseg > nt * nt ? (int64_t)fp.njb * fp.nseg : nt * nt;
    w.kpartial = (double*)take(sizeof(double) * times * per_t);
    w.part_s1 = (double*)take(sizeof(double) * times * chunks);
    w.part_col = (float*)take(sizeof(float) * times * chunks * (d > 0 ? d : 1));
    w.bwd_part = (float*)take(sizeof(float) * times * BWD_NSEG_MAX * m * (d > 0 ? d : 1));
    w.norms = (float*)take(sizeof(float) * times * m);
    if (fused) {
        w.part_max = (float*)take(sizeof(float) * times * chunks);
        w.images = (unsigned char*)take((size_t)times * fp.ntiles * fp.img);
    }
    w.total = off;
    return w;
}

int check_common(const float* src, int64_t ld_src, const float* tgt, int64_t ld_tgt, int64_t d,
                 const int64_t* src_idx, const int64_t* tgt_idx, int times, int64_t n, int kernel_num) {
    if (!src || !tgt) return GDA_E_NULL;
    if (d <= 0 || n <= 0 || times <= 0 || ld_src < d || ld_tgt < d) return GDA_E_SIZE;
    if (2 * n >= 46340 * 2) return GDA_E_SIZE;                 // m*m*times must stay well inside int64 / fp32 counts
    if (kernel_num < 1 || kernel_num > MAXQ) return GDA_E_UNSUPPORTED;
    if ((src_idx == nullptr) != (tgt_idx == nullptr)) return GDA_E_NULL;
    return GDA_OK;
}

Rows make_rows(const float* src, int64_t ld_src, const float* tgt, int64_t ld_tgt,
               const int64_t* src_idx, const int64_t* tgt_idx, int64_t n) {
    Rows R{src, ld_src, tgt, ld_tgt, src_idx, tgt_idx, n, false};
    R.vec4 = (ld_src % 4 == 0) && (ld_tgt % 4 == 0) && ((uintptr_t)src % 16 == 0) && ((uintptr_t)tgt % 16 == 0);
    return R;
}

}  // namespace

extern "C" size_t gda_mmd_workspace_bytes(int times, int64_t n, int64_t d) {
    if (times <= 0 || n <= 0 || d <= 0) return 0;
    return carve(nullptr, times, n, d).total;
}

extern "C" int gda_mmd_fwd_f32(const float* src, int64_t ld_src, const float* tgt, int64_t ld_tgt,
                               int64_t d, const int64_t* src_idx, const int64_t* tgt_idx,
                               int times, int64_t n, float kernel_mul, int kernel_num, float fix_sigma,
                               float* loss, float* bandwidth, float* l2_saved,
                               void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    return gda_mmd_fwd_ex_f32(src, ld_src, tgt, ld_tgt, d, src_idx, tgt_idx, times, n, kernel_mul, kernel_num, fix_sigma,
                              1.0f, nullptr, loss, bandwidth, l2_saved, workspace, workspace_bytes, stream_);
}

extern "C" int gda_mmd_fwd_ex_f32(const float* src, int64_t ld_src, const float* tgt, int64_t ld_tgt,
                                  int64_t d, const int64_t* src_idx, const int64_t* tgt_idx,
                                  int times, int64_t n, float kernel_mul, int kernel_num, float fix_sigma,
                                  float scale, const float* add, float* loss, float* bandwidth, float* l2_saved,
                                  void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    return gda_mmd_fwd_gather_f32(src, ld_src, tgt, ld_tgt, d, src_idx, tgt_idx, times, n, kernel_mul, kernel_num,
                                  fix_sigma, scale, add, nullptr, nullptr, loss, bandwidth, l2_saved, workspace,
                                  workspace_bytes, stream_);
}

extern "C" int gda_mmd_fwd_gather_f32(const float* src, int64_t ld_src, const float* tgt, int64_t ld_tgt,
                                      int64_t d, const int64_t* src_idx, const int64_t* tgt_idx,
                                      int times, int64_t n, float kernel_mul, int kernel_num, float fix_sigma,
                                      float scale, const float* add, float* rows_src, float* rows_tgt,
                                      float* loss, float* bandwidth, float* l2_saved,
                                      void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    int st = check_common(src, ld_src, tgt, ld_tgt, d, src_idx, tgt_idx, times, n, kernel_num);
    if (st != GDA_OK) return st;
    if (!loss || !bandwidth || !l2_saved || !workspace) return GDA_E_NULL;
    if ((rows_src == nullptr) != (rows_tgt == nullptr) || (rows_src && !src_idx)) return GDA_E_NULL;
    if (rows_src && (rows_src == src || rows_tgt == tgt)) return GDA_E_ALIAS;
    MmdWs ws = carve(workspace, times, n, d);
    if (workspace_bytes < ws.total) return GDA_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t m = 2 * n;
    const unsigned nt = (unsigned)gda_cdiv(m, TILE);
    const KParams kp{kernel_mul, kernel_num, fix_sigma};
    const unsigned chunks = (unsigned)gda_cdiv(m, SR);
    Rows R = make_rows(src, ld_src, tgt, ld_tgt, src_idx, tgt_idx, n);
    // the branch-free k_pairdist: rows without an index after this kernel (stacked as given, or gathered by it),
    // whole 32-feature chunks of 16-byte pieces, and a row tile of k_rowstats that fits its LDS
    const bool fast = (rows_src || !src_idx) && d % DK == 0 && d <= 2048 && m >= 1 &&
                      (rows_src ? (d % 4 == 0 && (uintptr_t)rows_src % 16 == 0 && (uintptr_t)rows_tgt % 16 == 0) : R.vec4) &&
                      kernel_num == 5 && kernel_mul == 2.0f;
    const size_t rs_lds = fast ? sizeof(float) * SR * (size_t)(d + 1) : 0;
    if (fast && rs_lds > 48 * 1024) {
        GDA_LDS_ATTR_ONCE(k_rowstats, sizeof(float) * SR * 2049);
    }
    k_rowstats<<<dim3(chunks, (unsigned)times), TB, rs_lds, stream>>>(R, d, m, ws.part_s1, ws.part_col, rows_src, rows_tgt,
                                                                    fast ? ws.norms : nullptr, nullptr);
    GDA_LAUNCH_CHECK();
    if (rows_src) R = make_rows(rows_src, d, rows_tgt, d, nullptr, nullptr, n);      // gathered: no index from here on
    k_bandwidth<<<(unsigned)times, BW_TB, 0, stream>>>(ws.part_s1, ws.part_col, (int)chunks, d, m, kp, bandwidth);
    GDA_LAUNCH_CHECK();
    const unsigned ntri = nt * (nt + 1) / 2;
    const dim3 grid(ntri, 1, (unsigned)times);
    if (fast)
        k_pairdist<5, true, true><<<grid, TB, 0, stream>>>(R, d, m, (int)nt, bandwidth, kp, l2_saved, ws.kpartial, ws.norms);
    else if (kernel_num == 5 && kernel_mul == 2.0f)
        k_pairdist<5, true, false><<<grid, TB, 0, stream>>>(R, d, m, (int)nt, bandwidth, kp, l2_saved, ws.kpartial, nullptr);
    else if (kernel_num == 5)
        k_pairdist<5, false, false><<<grid, TB, 0, stream>>>(R, d, m, (int)nt, bandwidth, kp, l2_saved, ws.kpartial, nullptr);
    else
        k_pairdist<0, false, false><<<grid, TB, 0, stream>>>(R, d, m, (int)nt, bandwidth, kp, l2_saved, ws.kpartial, nullptr);
    GDA_LAUNCH_CHECK();
    GDA_UNLESS_SKIPPED("k_finalize") k_finalize<<<1, TB, 0, stream>>>(ws.kpartial, (int)ntri, times, n, scale, add, loss);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_mmd_bwd_f32(const float* src, int64_t ld_src, const float* tgt, int64_t ld_tgt,
                               int64_t d, const int64_t* src_idx, const int64_t* tgt_idx,
                               int times, int64_t n, float kernel_mul, int kernel_num,
                               const float* bandwidth, const float* l2_saved, const float* grad_loss,
                               float* grad_rows, void* workspace, size_t workspace_bytes,
                               gda_stream_t stream_) {
    if (!grad_rows) return GDA_E_NULL;
    return gda_mmd_bwd_ex_f32(src, ld_src, tgt, ld_tgt, d, src_idx, tgt_idx, times, n, kernel_mul, kernel_num, bandwidth,
                              l2_saved, grad_loss, 1.0f, grad_rows, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0,
                              nullptr, workspace, workspace_bytes, stream_);
}

extern "C" int gda_mmd_bwd_ex_f32(const float* src, int64_t ld_src, const float* tgt, int64_t ld_tgt,
                                  int64_t d, const int64_t* src_idx, const int64_t* tgt_idx,
                                  int times, int64_t n, float kernel_mul, int kernel_num,
                                  const float* bandwidth, const float* l2_saved, const float* grad_loss, float scale,
                                  float* grad_rows,
                                  const int32_t* sel_s_rowptr, const int32_t* sel_s_col, int64_t n_src_rows, float* gsrc,
                                  const int32_t* sel_t_rowptr, const int32_t* sel_t_col, int64_t n_tgt_rows, float* gtgt,
                                  void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    int st = check_common(src, ld_src, tgt, ld_tgt, d, src_idx, tgt_idx, times, n, kernel_num);
    if (st != GDA_OK) return st;
    const bool scatter = sel_s_rowptr != nullptr;
    if (!bandwidth || !l2_saved || !grad_loss || !workspace || (!scatter && !grad_rows)) return GDA_E_NULL;
    if (scatter && (!sel_s_col || !sel_t_rowptr || !sel_t_col || !gsrc || !gtgt || n_src_rows < 0 || n_tgt_rows < 0))
        return GDA_E_NULL;
    MmdWs ws = carve(workspace, times, n, d);
    if (workspace_bytes < ws.total) return GDA_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t m = 2 * n;
    const Rows R = make_rows(src, ld_src, tgt, ld_tgt, src_idx, tgt_idx, n);
    const KParams kp{kernel_mul, kernel_num, 0.f};
    const int64_t ntiles = gda_cdiv(m, BJ);
    const BwdVariant var = bwd_variant(m);
    const int nseg = (int)(ntiles < var.nseg ? ntiles : var.nseg);
    const dim3 grid((unsigned)gda_cdiv(m, var.tile), (unsigned)(gda_cdiv(d, DC) * nseg), (unsigned)times);
    // branch-free staging (see k_bwd): whole 16-byte pieces everywhere
    const bool fast = R.vec4 && !R.src_idx && m % 4 == 0 && m >= 4 && d % DC == 0 && ((uintptr_t)l2_saved % 16 == 0);
    (void)kp;
    if (var.tile == 128)
        st = fast ? launch_bwd<true, 128>(grid, stream, R, d, m, l2_saved, grad_loss, scale, times, nseg, ws.bwd_part)
                  : launch_bwd<false, 128>(grid, stream, R, d, m, l2_saved, grad_loss, scale, times, nseg, ws.bwd_part);
    else
        st = fast ? launch_bwd<true, 64>(grid, stream, R, d, m, l2_saved, grad_loss, scale, times, nseg, ws.bwd_part)
                  : launch_bwd<false, 64>(grid, stream, R, d, m, l2_saved, grad_loss, scale, times, nseg, ws.bwd_part);
    if (st != GDA_OK) return st;
    if (scatter) {
        const int64_t most = n_src_rows > n_tgt_rows ? n_src_rows : n_tgt_rows;
        if (most > 0) {
            const dim3 sg((unsigned)gda_cdiv(most, TB / 32), 2);
            const bool quads = d % 4 == 0 && ((uintptr_t)ws.bwd_part % 16 == 0) && ((uintptr_t)gsrc % 16 == 0) &&
                               ((uintptr_t)gtgt % 16 == 0);
#define GDA_SCATTER(NS) k_bwd_scatter<NS><<<sg, TB, 0, stream>>>(ws.bwd_part, m, d, d, true, nseg, sel_s_rowptr, sel_s_col, \
                                                              n_src_rows, gsrc, sel_t_rowptr, sel_t_col, n_tgt_rows, gtgt, nullptr, 1.f)
            if (quads && nseg == 6) GDA_SCATTER(6);
            else if (quads && nseg == 4) GDA_SCATTER(4);
            else GDA_SCATTER(0);
#undef GDA_SCATTER
            GDA_LAUNCH_CHECK();
        }
        return GDA_OK;
    }
    const int64_t total = (int64_t)times * m * d;
    int64_t rg = gda_cdiv(total, TB);
    if (rg > 4096) rg = 4096;
    k_bwd_reduce<<<(unsigned)rg, TB, 0, stream>>>(ws.bwd_part, m * d, nseg, times, grad_rows, nullptr, 1.f);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

// ------------------------------------------------------------------ fused pass (gda_mmd_fused.inc) --
extern "C" int gda_mmd_fused_nseg(int times, int64_t n, int64_t d, float kernel_mul, int kernel_num) {
    FusedPlan fp;
    return fused_plan(times, n, d, kernel_mul, kernel_num, &fp) ? fp.nseg : 0;
}

// What the fused pass lays out for (times, n, d) -- host arithmetic only, no device needed: the CPU test suite checks
// the image layout, the plan and the workspace carve with it.
extern "C" int gda_mmd_fused_layout(int times, int64_t n, int64_t d, int64_t* out, int n_out) {
    if (!out) return GDA_E_NULL;
    if (n_out < 16) return GDA_E_SIZE;
    FusedPlan fp;
    if (!fused_plan(times, n, d, 2.0f, 5, &fp)) return GDA_E_UNSUPPORTED;
    const MmdWs w = carve((void*)(uintptr_t)4096, times, n, d);       // offsets relative to a fake base
    const int di = (int)d;
    const int64_t v[16] = {fp.nb, fp.ntiles, fp.njb, fp.nseg, fp.total, (int64_t)fp.img,
                           f_off_rl(di), f_off_th(di), f_off_tl(di), f_off_n(di), F_OFF_XS, F_TAIL_FLOATS,
                           (int64_t)((uintptr_t)w.images - 4096), (int64_t)((uintptr_t)w.part_max - 4096),
                           (int64_t)((uintptr_t)w.kpartial - 4096), (int64_t)w.total};
    for (int i = 0; i < 16; ++i) out[i] = v[i];
    return GDA_OK;
}

template <int NB>
static int launch_fused(hipStream_t stream, const Rows& Rin, float* rows_src, float* rows_tgt, int64_t m, int64_t d, const MmdWs& ws,
                        const FusedPlan& fp, int times, float* bandwidth, float* grad_part) {
    GDA_UNLESS_SKIPPED("k_tile_split") k_tile_split<NB><<<dim3((unsigned)fp.ntiles, (unsigned)times), TB, 0, stream>>>(Rin, m, ws.part_s1, ws.part_col, ws.part_max,
                                                                               rows_src, rows_tgt, ws.images);
    GDA_LAUNCH_CHECK();
    const Rows R = rows_src ? make_rows(rows_src, d, rows_tgt, d, nullptr, nullptr, Rin.n) : Rin;   // gathered: no index from here on
    const size_t lds = 2 * fp.img + sizeof(float) * 32 * (TB / 64);
    GDA_LDS_ATTR_ONCE((k_mmd_fused<NB>), lds);
    const unsigned grid = 8u * (unsigned)gda_cdiv(fp.total, 8);
    GDA_UNLESS_SKIPPED("k_mmd_fused") k_mmd_fused<NB><<<grid, TB, lds, stream>>>(R, m, fp.ntiles, fp.njb, fp.nseg, fp.total, ws.images, ws.part_s1, ws.part_col,
                                              ws.part_max, bandwidth, grad_part, ws.kpartial);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_mmd_fused_fwd_f32(const float* src, int64_t ld_src, const float* tgt, int64_t ld_tgt,
                                     int64_t d, const int64_t* src_idx, const int64_t* tgt_idx,
                                     int times, int64_t n, float kernel_mul, int kernel_num, float fix_sigma,
                                     float scale, const float* add, float* rows_src, float* rows_tgt,
                                     float* loss, float* bandwidth, float* grad_part, int nseg,
                                     void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    int st = check_common(src, ld_src, tgt, ld_tgt, d, src_idx, tgt_idx, times, n, kernel_num);
    if (st != GDA_OK) return st;
    if (!loss || !bandwidth || !grad_part || !workspace) return GDA_E_NULL;
    if ((rows_src == nullptr) != (rows_tgt == nullptr) || (rows_src && !src_idx)) return GDA_E_NULL;
    if (rows_src && (rows_src == src || rows_tgt == tgt)) return GDA_E_ALIAS;
    FusedPlan fp;
    if (!fused_plan(times, n, d, kernel_mul, kernel_num, &fp)) return GDA_E_UNSUPPORTED;
    if (fix_sigma > 0.f) return GDA_E_UNSUPPORTED;                // (a NaN row reaches the loss through the data-dependent bandwidth here)
    if (nseg != fp.nseg) return GDA_E_SIZE;
    if (src_idx && !rows_src) return GDA_E_UNSUPPORTED;           // the pass reads rows without an index: gather them
    Rows R = make_rows(src, ld_src, tgt, ld_tgt, src_idx, tgt_idx, n);
    if (!R.vec4 || (rows_src && ((uintptr_t)rows_src % 16 != 0 || (uintptr_t)rows_tgt % 16 != 0))) return GDA_E_UNSUPPORTED;   // 16-byte loads
    if ((uintptr_t)grad_part % 16 != 0) return GDA_E_UNSUPPORTED;
    MmdWs ws = carve(workspace, times, n, d);
    if (workspace_bytes < ws.total) return GDA_E_WORKSPACE;
    const int64_t m = 2 * n;
    hipStream_t stream = (hipStream_t)stream_;
    switch (fp.nb) {
        case 1: st = launch_fused<1>(stream, R, rows_src, rows_tgt, m, d, ws, fp, times, bandwidth, grad_part); break;
        case 2: st = launch_fused<2>(stream, R, rows_src, rows_tgt, m, d, ws, fp, times, bandwidth, grad_part); break;
        case 3: st = launch_fused<3>(stream, R, rows_src, rows_tgt, m, d, ws, fp, times, bandwidth, grad_part); break;
        default: st = launch_fused<4>(stream, R, rows_src, rows_tgt, m, d, ws, fp, times, bandwidth, grad_part); break;
    }
    if (st != GDA_OK) return st;
    GDA_UNLESS_SKIPPED("k_finalize") k_finalize<<<1, TB, 0, stream>>>(ws.kpartial, fp.njb * fp.nseg, times, n, scale, add, loss);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

// ------------------------------------------------------------------ chunked one-pass (gda_mmd_chunked.inc) --
extern "C" int gda_mmd_chunked_plan(int times, int64_t n, int64_t d, float kernel_mul, int kernel_num, int64_t* out, int n_out) {
    if (!out) return GDA_E_NULL;
    if (n_out < 8) return GDA_E_SIZE;
    ChunkPlan fp;
    if (!chunk_plan(times, n, d, kernel_mul, kernel_num, &fp)) return GDA_E_UNSUPPORTED;
    const int64_t v[8] = {fp.nseg, fp.dp, fp.nb, fp.nc, fp.ntiles, fp.njb, fp.total, (int64_t)fp.img};
    for (int i = 0; i < 8; ++i) out[i] = v[i];
    return GDA_OK;
}

#ifdef GDA_MMD_TRACE
extern "C" int gda_dbg_mmd_trace(unsigned long long* buf) {
    GDA_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(gda_mmd_trace_buf), &buf, sizeof(buf)));
    return GDA_OK;
}
#endif

extern "C" size_t gda_mmd_chunked_workspace_bytes(int times, int64_t n, int64_t d) {
    ChunkPlan fp;
    if (!chunk_plan(times, n, d, 2.0f, 5, &fp)) return 0;
    return chunk_carve(nullptr, times, fp).total;
}

template <int NB>
static int launch_chunked(hipStream_t stream, const Rows& Rin, float* rows_src, float* rows_tgt, int64_t ldr, int64_t m, int64_t d,
                          const ChunkWs& ws, const ChunkPlan& fp, int times, float* bandwidth, float* grad_part, int64_t ldp) {
    GDA_UNLESS_SKIPPED("k_tile_split") k_tile_split_c<NB><<<dim3((unsigned)fp.ntiles, (unsigned)times), TB, 0, stream>>>(
        Rin, m, d, fp.nc, ws.part_s1, ws.part_col, ws.part_max, rows_src, rows_tgt, ldr, ws.images);
    GDA_LAUNCH_CHECK();
    k_bw_fold<<<dim3((unsigned)gda_cdiv(fp.dp, 64), (unsigned)times), TB, 0, stream>>>(fp.ntiles, fp.dp, ws.part_s1, ws.part_col, ws.part_max, ws.bwstat);
    GDA_LAUNCH_CHECK();
    const Rows R = make_rows(rows_src, ldr, rows_tgt, ldr, nullptr, nullptr, Rin.n);          // the padded copy from here on
    constexpr int D = 32 * NB, PR = 2 * FT * f_rstride(D), PC = 2 * D * F_TSTRIDE;
    const size_t lds = FC_NBUF * (size_t)(PC > PR ? PC : PR) + sizeof(float) * (FC_NI * 36 + 32 * (TB / 64) + FC_STG * (TB / 64));
    GDA_LDS_ATTR_ONCE((k_mmd_chunked<NB>), lds);
    const unsigned grid = 8u * (unsigned)gda_cdiv(fp.total, 8);
    GDA_UNLESS_SKIPPED("k_mmd_fused") k_mmd_chunked<NB><<<grid, TB, lds, stream>>>(R, m, fp.ntiles, fp.nc, fp.njb, fp.nseg, fp.total, ws.images,
                                                                                  ws.bwstat, bandwidth, grad_part, ldp, ws.kpartial);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_mmd_chunked_fwd_f32(const float* src, int64_t ld_src, const float* tgt, int64_t ld_tgt,
                                       int64_t d, const int64_t* src_idx, const int64_t* tgt_idx,
                                       int times, int64_t n, float kernel_mul, int kernel_num, float fix_sigma,
                                       float scale, const float* add, float* rows_src, float* rows_tgt, int64_t ld_rows,
                                       float* loss, float* bandwidth, float* grad_part, int64_t ld_part, int nseg,
                                       void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    int st = check_common(src, ld_src, tgt, ld_tgt, d, src_idx, tgt_idx, times, n, kernel_num);
    if (st != GDA_OK) return st;
    if (!loss || !bandwidth || !grad_part || !workspace || !rows_src || !rows_tgt) return GDA_E_NULL;
    if (rows_src == src || rows_tgt == tgt || rows_src == rows_tgt) return GDA_E_ALIAS;
    ChunkPlan fp;
    if (!chunk_plan(times, n, d, kernel_mul, kernel_num, &fp)) return GDA_E_UNSUPPORTED;
    if (fix_sigma > 0.f) return GDA_E_UNSUPPORTED;
    if (nseg != fp.nseg || ld_rows != fp.dp || ld_part != fp.dp) return GDA_E_SIZE;
    if (((uintptr_t)rows_src | (uintptr_t)rows_tgt | (uintptr_t)grad_part) % 16 != 0) return GDA_E_UNSUPPORTED;
    Rows R = make_rows(src, ld_src, tgt, ld_tgt, src_idx, tgt_idx, n);
    ChunkWs ws = chunk_carve(workspace, times, fp);
    if (workspace_bytes < ws.total) return GDA_E_WORKSPACE;
    if ((uintptr_t)ws.images % 16 != 0) return GDA_E_UNSUPPORTED;
    const int64_t m = 2 * n;
    hipStream_t stream = (hipStream_t)stream_;
    switch (fp.nb) {
        case 1: st = launch_chunked<1>(stream, R, rows_src, rows_tgt, ld_rows, m, d, ws, fp, times, bandwidth, grad_part, ld_part); break;
        case 2: st = launch_chunked<2>(stream, R, rows_src, rows_tgt, ld_rows, m, d, ws, fp, times, bandwidth, grad_part, ld_part); break;
        case 3: st = launch_chunked<3>(stream, R, rows_src, rows_tgt, ld_rows, m, d, ws, fp, times, bandwidth, grad_part, ld_part); break;
        default: st = launch_chunked<4>(stream, R, rows_src, rows_tgt, ld_rows, m, d, ws, fp, times, bandwidth, grad_part, ld_part); break;
    }
    if (st != GDA_OK) return st;
    GDA_UNLESS_SKIPPED("k_finalize") k_finalize<<<1, TB, 0, stream>>>(ws.kpartial, fp.njb * fp.nseg, times, n, scale, add, loss);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_mmd_fused_bwd_mask_f32(const float* grad_part, int nseg, int times, int64_t n, int64_t d,
                                          const float* grad_loss, float scale, float* grad_rows,
                                          const int32_t* sel_s_rowptr, const int32_t* sel_s_col, int64_t n_src_rows, float* gsrc,
                                          const int32_t* sel_t_rowptr, const int32_t* sel_t_col, int64_t n_tgt_rows, float* gtgt,
                                          const float* mask_src, float p_src, const float* mask_tgt, float p_tgt,
                                          gda_stream_t stream_) {
    return gda_mmd_fused_bwd_ld_f32(grad_part, d, nseg, times, n, d, grad_loss, scale, grad_rows, sel_s_rowptr, sel_s_col,
                                    n_src_rows, gsrc, sel_t_rowptr, sel_t_col, n_tgt_rows, gtgt, mask_src, p_src, mask_tgt, p_tgt,
                                    stream_);
}

// The same with the partials' rows padded to ld_part >= d floats (the chunked one-pass kernel pads every width to whole
// 32-column blocks: 645 -> 672); the feature-row gradients gsrc / gtgt and grad_rows stay d wide, of any alignment.
extern "C" int gda_mmd_fused_bwd_ld_f32(const float* grad_part, int64_t ld_part, int nseg, int times, int64_t n, int64_t d,
                                        const float* grad_loss, float scale, float* grad_rows,
                                        const int32_t* sel_s_rowptr, const int32_t* sel_s_col, int64_t n_src_rows, float* gsrc,
                                        const int32_t* sel_t_rowptr, const int32_t* sel_t_col, int64_t n_tgt_rows, float* gtgt,
                                        const float* mask_src, float p_src, const float* mask_tgt, float p_tgt,
                                        gda_stream_t stream_) {
    if (ld_part < d) return GDA_E_SIZE;
    if ((mask_src && !(p_src >= 0.f && p_src < 1.f)) || (mask_tgt && !(p_tgt >= 0.f && p_tgt < 1.f))) return GDA_E_SIZE;
    if ((mask_src || mask_tgt) && !sel_s_rowptr) return GDA_E_UNSUPPORTED;        // masks ride on the scatter only
    if ((((uintptr_t)mask_src | (uintptr_t)mask_tgt) % 16) && d % 4 == 0) return GDA_E_UNSUPPORTED;
    const float ms_s = mask_src ? 1.f / (1.f - p_src) : 1.f, ms_t = mask_tgt ? 1.f / (1.f - p_tgt) : 1.f;
    if (!grad_part || !grad_loss) return GDA_E_NULL;
    if (times <= 0 || n <= 0 || d <= 0 || nseg < 1 || nseg > F_NSEG_MAX) return GDA_E_SIZE;
    const bool scatter = sel_s_rowptr != nullptr;
    if (!scatter && !grad_rows) return GDA_E_NULL;
    if (scatter && (!sel_s_col || !sel_t_rowptr || !sel_t_col || !gsrc || !gtgt || n_src_rows < 0 || n_tgt_rows < 0))
        return GDA_E_NULL;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t m = 2 * n;
    const float cmul = scale / ((float)n * (float)n) / (float)times;       // dloss x this x 4: k_bwd's coefficient
    if (scatter) {
        const int64_t most = n_src_rows > n_tgt_rows ? n_src_rows : n_tgt_rows;
        if (most <= 0) return GDA_OK;
        const dim3 sg((unsigned)gda_cdiv(most, TB / 32), 2);
        // partial quads by 16-byte loads when their rows allow it; the feature-row gradients by 16-byte stores when THEIR rows do
        const bool vec_out = d % 4 == 0 && ((uintptr_t)gsrc % 16 == 0) && ((uintptr_t)gtgt % 16 == 0);
        const bool quads = ld_part % 4 == 0 && ((uintptr_t)grad_part % 16 == 0) && (vec_out || ld_part > d);
#define GDA_SCATTER(NS) k_bwd_scatter<NS><<<sg, TB, 0, stream>>>(grad_part, m, d, ld_part, vec_out, nseg, sel_s_rowptr, sel_s_col, n_src_rows, gsrc, \
                                                              sel_t_rowptr, sel_t_col, n_tgt_rows, gtgt, grad_loss, cmul,    \
                                                              mask_src, ms_s, mask_tgt, ms_t)
        if (!quads) GDA_SCATTER(0);
        else switch (nseg) {
            case 1: GDA_SCATTER(1); break;
            case 2: GDA_SCATTER(2); break;
            case 3: GDA_SCATTER(3); break;
            case 4: GDA_SCATTER(4); break;
            case 5: GDA_SCATTER(5); break;
            case 6: GDA_SCATTER(6); break;
            case 7: GDA_SCATTER(7); break;
            default: GDA_SCATTER(8); break;
        }
#undef GDA_SCATTER
        GDA_LAUNCH_CHECK();
        return GDA_OK;
    }
    const int64_t total = (int64_t)times * m * d;
    int64_t rg = gda_cdiv(total, TB);
    if (rg > 4096) rg = 4096;
    if (ld_part == d) k_bwd_reduce<<<(unsigned)rg, TB, 0, stream>>>(grad_part, m * d, nseg, times, grad_rows, grad_loss, cmul);
    else k_bwd_reduce_ld<<<(unsigned)rg, TB, 0, stream>>>(grad_part, m, d, ld_part, nseg, times, grad_rows, grad_loss, cmul);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_mmd_fused_bwd_f32(const float* grad_part, int nseg, int times, int64_t n, int64_t d,
                                     const float* grad_loss, float scale, float* grad_rows,
                                     const int32_t* sel_s_rowptr, const int32_t* sel_s_col, int64_t n_src_rows, float* gsrc,
                                     const int32_t* sel_t_rowptr, const int32_t* sel_t_col, int64_t n_tgt_rows, float* gtgt,
                                     gda_stream_t stream_) {
    return gda_mmd_fused_bwd_mask_f32(grad_part, nseg, times, n, d, grad_loss, scale, grad_rows, sel_s_rowptr, sel_s_col,
                                      n_src_rows, gsrc, sel_t_rowptr, sel_t_col, n_tgt_rows, gtgt, nullptr, 0.f, nullptr, 0.f,
                                      stream_);
}

