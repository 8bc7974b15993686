// Multi-kernel Gaussian MMD over sampled source/target rows, forward + backward, gfx950.
//
// Replaces the torch elementwise chain of pygda/utils/mmd.py:4-159 (guassian_kernel,
// get_MMD, MMD), which materialises an [m, m, d] difference tensor per resample
// (m = 2000, d = 128: 2 GB, several live at once and kept for backward).  Here the
// pairwise squared distances are produced tile by tile in LDS in the reference's direct
// difference form (mmd.py:43-46: sum_k (total[j,k]-total[i,k])^2 -- no |a|^2+|b|^2-2ab
// cancellation), only the [m, m] distance matrix is kept (16 MB per resample, L2/MALL
// resident) and the backward pass recomputes the kernel weights from it.
//
//   k_pairdist   tile 64x64 of L2 + per-tile partial sums; only tiles on/above the diagonal are
//                computed, the mirror tile is written from registers (L2 is exactly symmetric:
//                (a-b)^2 == (b-a)^2 in fp32)                          (VALU bound: 1.5*m^2*d flop)
//   k_ksum       bandwidth from the partials (mmd.py:50-51), K = sum_q exp(-L2/bw_q)
//                (mmd.py:52-55), signed block sums XX+YY-XY-YX (mmd.py:100-106)
//   k_finalize   mean per resample, average over resamples (mmd.py:152-157)
//   k_bwd        grad_total[i,:] = 4 * ((sum_j G[i,j]) total[i,:] - sum_j G[i,j] total[j,:]),
//                G = dloss/dL2 (symmetric; the bandwidth is a constant, mmd.py:50 .data); the
//                G x total product runs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact
//                fp32 fma chains, measured no less accurate than the difference form here)
// All reductions are fixed-order (no atomics): results are run-to-run deterministic.
#include "gda_common.h"

namespace {

constexpr int TB = 256;
constexpr int TILE = 64;      // rows/cols of the pair tile per workgroup
constexpr int DK = 32;        // feature chunk staged in LDS per iteration (forward)
constexpr int LDT = TILE + 4; // padded LDS leading dimension, keeps 16-byte row alignment
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

// ---------------------------------------------------------------- forward --
__global__ void __launch_bounds__(TB)
k_pairdist(Rows R, int64_t d, int64_t m, float* __restrict__ l2, double* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float As[DK][LDT];   // rows i of the tile
    __shared__ __attribute__((aligned(16))) float Bs[DK][LDT];   // rows j of the tile
    __shared__ double red[TB / 64];
    const int t = blockIdx.z;
    const int64_t i0 = (int64_t)blockIdx.y * TILE, j0 = (int64_t)blockIdx.x * TILE;
    const int tid = threadIdx.x, ty = tid / 16, tx = tid % 16;
    if (blockIdx.x < blockIdx.y) {                       // mirror tile: written by its twin
        if (tid == 0) partial[((int64_t)t * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = 0.0;
        return;
    }
    const bool diag = blockIdx.x == blockIdx.y;

    // this thread stages rows (tid/8) and (tid/8 + 32) of both tiles, features kq*4..+3
    const int lr = tid / 8, kq = (tid % 8) * 4;
    const float* pa[2]; const float* pb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int64_t ri = i0 + lr + 32 * q, rj = j0 + lr + 32 * q;
        pa[q] = ri < m ? row_ptr(R, t, ri) : nullptr;
        pb[q] = rj < m ? row_ptr(R, t, rj) : nullptr;
    }

    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;

    for (int64_t k0 = 0; k0 < d; k0 += DK) {
        float4 va[2], vb[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            va[q] = load4(pa[q], k0 + kq, d, R.vec4);
            vb[q] = load4(pb[q], k0 + kq, d, R.vec4);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = lr + 32 * q;
            As[kq + 0][r] = va[q].x; As[kq + 1][r] = va[q].y; As[kq + 2][r] = va[q].z; As[kq + 3][r] = va[q].w;
            Bs[kq + 0][r] = vb[q].x; Bs[kq + 1][r] = vb[q].y; Bs[kq + 2][r] = vb[q].z; Bs[kq + 3][r] = vb[q].w;
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < DK; ++kk) {
            const float4 a4 = *reinterpret_cast<const float4*>(&As[kk][ty * 4]);
            const float4 b4 = *reinterpret_cast<const float4*>(&Bs[kk][tx * 4]);
            const float av[4] = {a4.x, a4.y, a4.z, a4.w};
            const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float df = bv[b] - av[a];          // total0 - total1 = total[j] - total[i]
                    acc[a][b] = fmaf(df, df, acc[a][b]);
                }
        }
    }

    float local = 0.f;
    float* out = l2 + (int64_t)t * m * m;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int64_t i = i0 + ty * 4 + a;
        if (i >= m) continue;
        const int64_t j = j0 + tx * 4;
        if (j + 3 < m && (m % 4 == 0)) {
            *reinterpret_cast<float4*>(out + i * m + j) = make_float4(acc[a][0], acc[a][1], acc[a][2], acc[a][3]);
            local += (acc[a][0] + acc[a][1]) + (acc[a][2] + acc[a][3]);
        } else {
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (j + b < m) { out[i * m + j + b] = acc[a][b]; local += acc[a][b]; }
        }
    }
    if (!diag) {                                         // mirror tile L2[j][i] = L2[i][j], from registers
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int64_t j = j0 + tx * 4 + b;
            if (j >= m) continue;
            const int64_t i = i0 + ty * 4;
            if (i + 3 < m && (m % 4 == 0)) {
                *reinterpret_cast<float4*>(out + j * m + i) = make_float4(acc[0][b], acc[1][b], acc[2][b], acc[3][b]);
            } else {
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    if (i + a < m) out[j * m + i + a] = acc[a][b];
            }
        }
    }
    const double s = block_sum((double)local, red);
    if (tid == 0) partial[((int64_t)t * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = diag ? s : 2.0 * s;
}

struct KParams {
    float kernel_mul; int kernel_num; float fix_sigma;
};

// bandwidth of resample t from the pairdist partials (every workgroup recomputes it in the
// same fixed order: a few hundred L2-resident doubles)
__device__ float bandwidth_of(const double* __restrict__ partial, int tiles, int t, int64_t m,
                              KParams kp, double* sh) {
    double v = 0.0;
    for (int k = threadIdx.x; k < tiles; k += TB) v += partial[(int64_t)t * tiles + k];
    const double s = block_sum(v, sh);
    __shared__ float bw_sh;
    if (threadIdx.x == 0) {
        float bw;
        if (kp.fix_sigma > 0.f) bw = kp.fix_sigma;
        else bw = ((float)s + 1e-6f) / (float)(m * m - m);          // mmd.py:50
        float div = 1.f;
        for (int q = 0; q < kp.kernel_num / 2; ++q) div *= kp.kernel_mul;
        bw_sh = bw / div;                                            // mmd.py:51
    }
    __syncthreads();
    return bw_sh;
}

// one workgroup per resample: bandwidth[t] from the pairdist partials (mmd.py:50-51)
__global__ void __launch_bounds__(TB)
k_bandwidth(const double* __restrict__ partial, int tiles_per_t, int64_t m, KParams kp,
            float* __restrict__ bandwidth) {
    __shared__ double red[TB / 64];
    const float bw0 = bandwidth_of(partial, tiles_per_t, blockIdx.x, m, kp, red);
    if (threadIdx.x == 0) bandwidth[blockIdx.x] = bw0;
}

constexpr int KS_ROWS = 8;    // rows of L2 per workgroup pass in k_ksum

// K = sum_q exp(-L2 / bw_q) (mmd.py:52-55) and the signed block sums XX + YY - XY - YX
// (mmd.py:100-106).  Streams the [m, m] matrix row-wise with 16-byte loads: one workgroup takes
// KS_ROWS consecutive rows per pass (grid-stride), HBM/Infinity-Cache bound.
// KN = compile-time kernel_num (5 in every pygda call: fully unrolled, exponents in registers);
// KN = 0 keeps the run-time count.
template <int KN>
__global__ void __launch_bounds__(TB)
k_ksum(const float* __restrict__ l2, int64_t m, int64_t n, KParams kp,
       const float* __restrict__ bandwidth, double* __restrict__ kpartial) {
    __shared__ double red[TB / 64];
    const int t = blockIdx.y;
    const float bw0 = bandwidth[t];
    const int kn = KN > 0 ? KN : kp.kernel_num;
    float nib[KN > 0 ? KN : MAXQ];                     // -1 / (bandwidth * kernel_mul^q), mmd.py:52
    {
        float f = 1.f;
#pragma unroll
        for (int q = 0; q < kn; ++q) { nib[q] = -1.f / (bw0 * f); f *= kp.kernel_mul; }
    }
    const float* L = l2 + (int64_t)t * m * m;
    const bool v4 = (m % 4 == 0);
    float local = 0.f;
    for (int64_t r0 = (int64_t)blockIdx.x * KS_ROWS; r0 < m; r0 += (int64_t)gridDim.x * KS_ROWS) {
        const int64_t r1 = r0 + KS_ROWS < m ? r0 + KS_ROWS : m;
        if (v4) {
            const int64_t per_row = m / 4;
            for (int64_t f = threadIdx.x; f < (r1 - r0) * per_row; f += TB) {
                const int64_t i = r0 + f / per_row, j = (f % per_row) * 4;
                const float4 dv = *reinterpret_cast<const float4*>(L + i * m + j);
                const float dd[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float kv = 0.f;
#pragma unroll
                    for (int q = 0; q < kn; ++q) kv += __expf(dd[c] * nib[q]);
                    local += ((i < n) == (j + c < n)) ? kv : -kv;
                }
            }
        } else {
            for (int64_t f = threadIdx.x; f < (r1 - r0) * m; f += TB) {
                const int64_t i = r0 + f / m, j = f % m;
                float kv = 0.f;
#pragma unroll
                for (int q = 0; q < kn; ++q) kv += __expf(L[i * m + j] * nib[q]);
                local += ((i < n) == (j < n)) ? kv : -kv;
            }
        }
    }
    const double s = block_sum((double)local, red);
    if (threadIdx.x == 0) kpartial[(int64_t)t * gridDim.x + blockIdx.x] = s;
}

__global__ void __launch_bounds__(TB)
k_finalize(const double* __restrict__ kpartial, int tiles_per_t, int times, int64_t n,
           float* __restrict__ loss) {
    __shared__ double red[TB / 64];
    float total = 0.f;
    for (int t = 0; t < times; ++t) {
        double v = 0.0;
        for (int k = threadIdx.x; k < tiles_per_t; k += TB) v += kpartial[(int64_t)t * tiles_per_t + k];
        const double s = block_sum(v, red);
        if (threadIdx.x == 0) total += (float)(s / ((double)n * (double)n));            // mmd.py:106 mean
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = total / (float)times;                               // mmd.py:157
}

// --------------------------------------------------------------- backward --
// Work decomposition: 32 rows x 128 feature columns per workgroup, the j range cut into
// NSEG segments -> (m/32) * NSEG * times workgroups (1260 at m=2000, times=5) so that every
// CU holds several workgroups; per-segment partial sums are combined in a fixed order by
// k_bwd_reduce (deterministic, no atomics).
//
// Per j tile of 64 rows: G[32 x 64] is rebuilt from the saved L2 into LDS (as Gs[j][i]) and the
// tile's rows of `total` are staged in LDS; each of the 4 waves then owns 32 feature columns and
// accumulates  acc[32 x 32] += G[32 x 64] * total[64 x 32]  with 32 v_mfma_f32_32x32x2_f32
// (A = G: lane l holds G[i = l&31][k = l>>5]; B: lane l holds total[k = l>>5][c = l&31]; both
// are single conflict-free ds_read_b32).  Row sums of G accumulate beside it; the epilogue forms
// 4 * (rowsum * total[i] - acc).
constexpr int BI = 32;        // rows i per workgroup
constexpr int BJ = 64;        // rows j per LDS tile
constexpr int LDG = BI + 4;   // padded leading dimension of the G tile (16-byte aligned rows)
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int KN>
__global__ void __launch_bounds__(TB)
k_bwd(Rows R, int64_t d, int64_t m, const float* __restrict__ l2, const float* __restrict__ bandwidth,
      KParams kp, const float* __restrict__ grad_loss, int times, int nseg,
      float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float Gs[BJ][LDG];     // Gs[j][i] = G[i][j] (G symmetric)
    __shared__ __attribute__((aligned(16))) float Ts[BJ][DC];      // rows j of total, this column chunk
    __shared__ float rowsum[BI];
    const int t = blockIdx.z;
    const int seg = blockIdx.y % nseg;
    const int64_t c0 = (int64_t)(blockIdx.y / nseg) * DC;
    const int64_t i0 = (int64_t)blockIdx.x * BI;
    const int tid = threadIdx.x, wave = tid / 64, lane = tid % 64;
    const int64_t n = R.n;

    const int kn = KN > 0 ? KN : kp.kernel_num;
    float nib[KN > 0 ? KN : MAXQ];                                  // -1 / bw_q
    {
        float f = 1.f;
        const float b0 = bandwidth[t];
#pragma unroll
        for (int q = 0; q < kn; ++q) { nib[q] = -1.f / (b0 * f); f *= kp.kernel_mul; }
    }
    // d loss / d K[i,j] = +-1 / (n^2 * times) * upstream
    const float coef = grad_loss[0] / ((float)n * (float)n) / (float)times;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float rs = 0.f;                                                 // threads 0..31: row sum of G for row tid
    if (tid < BI) rowsum[tid] = 0.f;

    // this segment's j tiles: tiles seg, seg + nseg, ...
    const float* L = l2 + (int64_t)t * m * m;
    const int64_t ntiles = gda_cdiv_dev(m, BJ);
    const int ka = lane >> 5, la = lane & 31;
    for (int64_t jt = seg; jt < ntiles; jt += nseg) {
        const int64_t j0 = jt * BJ;
        __syncthreads();
        // G tile, read as L2[j][i] (== L2[i][j]): coalesced along i, stored row-contiguous
        for (int f = tid; f < BJ * BI; f += TB) {
            const int jj = f / BI, ii = f % BI;
            const int64_t j = j0 + jj, i = i0 + ii;
            float g = 0.f;
            if (i < m && j < m) {
                const float dist = L[j * m + i];
                float dk = 0.f;
#pragma unroll
                for (int q = 0; q < kn; ++q) dk = fmaf(__expf(dist * nib[q]), nib[q], dk);
                g = (((i < n) == (j < n)) ? coef : -coef) * dk;
            }
            Gs[jj][ii] = g;
        }
        for (int f = tid; f < BJ * (DC / 4); f += TB) {
            const int jj = f / (DC / 4), c4 = (f % (DC / 4)) * 4;
            const int64_t j = j0 + jj;
            const float* p = j < m ? row_ptr(R, t, j) : nullptr;
            *reinterpret_cast<float4*>(&Ts[jj][c4]) = load4(p, c0 + c4, d, R.vec4);
        }
        __syncthreads();
        if (tid < BI) {
#pragma unroll 8
            for (int jj = 0; jj < BJ; ++jj) rs += Gs[jj][tid];
        }
#pragma unroll 8
        for (int jj = 0; jj < BJ; jj += 2) {
            const float av = Gs[jj + ka][la];
            const float bv = Ts[jj + ka][wave * 32 + la];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    }
    if (tid < BI) rowsum[tid] = rs;
    __syncthreads();

    // epilogue: part[t][seg][i][c] = rowsum[i] * total[i][c] - acc   (the factor 4 is applied by the reduce)
    float* out = part + (((int64_t)t * nseg + seg) * m) * d;
    const int64_t c = c0 + wave * 32 + la;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int il = (r & 3) + 8 * (r >> 2) + 4 * ka;             // C/D layout of the 32x32 MFMA
        const int64_t i = i0 + il;
        if (i < m && c < d) {
            const float ti = row_ptr(R, t, i)[c];
            out[i * d + c] = fmaf(rowsum[il], ti, -acc[r]);
        }
    }
}

// grad_rows = 4 * sum over segments (fixed order)
__global__ void __launch_bounds__(TB)
k_bwd_reduce(const float* __restrict__ part, int64_t per_t, int nseg, int times,
             float* __restrict__ grad_rows) {
    const int64_t total = per_t * times;
    for (int64_t k = (int64_t)blockIdx.x * TB + threadIdx.x; k < total; k += (int64_t)gridDim.x * TB) {
        const int64_t t = k / per_t, r = k % per_t;
        float s = 0.f;
        for (int g = 0; g < nseg; ++g) s += part[(t * nseg + g) * per_t + r];
        grad_rows[k] = 4.f * s;
    }
}

constexpr int BWD_NSEG = 4;

struct MmdWs { double* partial; double* kpartial; float* bwd_part; size_t total; };

MmdWs carve(void* base, int times, int64_t n, int64_t d) {
    const int64_t m = 2 * n, nt = gda_cdiv(m, TILE);
    MmdWs w{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void* p = base ? (void*)((char*)base + off) : nullptr;
        off += gda_align_up(bytes, 256);
        return p;
    };
    w.partial = (double*)take(sizeof(double) * times * nt * nt);
    w.kpartial = (double*)take(sizeof(double) * times * nt * nt);
    w.bwd_part = (float*)take(sizeof(float) * times * BWD_NSEG * m * (d > 0 ? d : 1));
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
    int st = check_common(src, ld_src, tgt, ld_tgt, d, src_idx, tgt_idx, times, n, kernel_num);
    if (st != GDA_OK) return st;
    if (!loss || !bandwidth || !l2_saved || !workspace) return GDA_E_NULL;
    MmdWs ws = carve(workspace, times, n, d);
    if (workspace_bytes < ws.total) return GDA_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t m = 2 * n;
    const unsigned nt = (unsigned)gda_cdiv(m, TILE);
    const Rows R = make_rows(src, ld_src, tgt, ld_tgt, src_idx, tgt_idx, n);
    const KParams kp{kernel_mul, kernel_num, fix_sigma};
    const dim3 grid(nt, nt, (unsigned)times);
    k_pairdist<<<grid, TB, 0, stream>>>(R, d, m, l2_saved, ws.partial);
    GDA_LAUNCH_CHECK();
    k_bandwidth<<<(unsigned)times, TB, 0, stream>>>(ws.partial, (int)(nt * nt), m, kp, bandwidth);
    GDA_LAUNCH_CHECK();
    const unsigned kgrid = (unsigned)(gda_cdiv(m, KS_ROWS) < 256 ? gda_cdiv(m, KS_ROWS) : 256);
    if (kernel_num == 5) k_ksum<5><<<dim3(kgrid, (unsigned)times), TB, 0, stream>>>(l2_saved, m, n, kp, bandwidth, ws.kpartial);
    else k_ksum<0><<<dim3(kgrid, (unsigned)times), TB, 0, stream>>>(l2_saved, m, n, kp, bandwidth, ws.kpartial);
    GDA_LAUNCH_CHECK();
    k_finalize<<<1, TB, 0, stream>>>(ws.kpartial, (int)kgrid, times, n, loss);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_mmd_bwd_f32(const float* src, int64_t ld_src, const float* tgt, int64_t ld_tgt,
                               int64_t d, const int64_t* src_idx, const int64_t* tgt_idx,
                               int times, int64_t n, float kernel_mul, int kernel_num,
                               const float* bandwidth, const float* l2_saved, const float* grad_loss,
                               float* grad_rows, void* workspace, size_t workspace_bytes,
                               gda_stream_t stream_) {
    int st = check_common(src, ld_src, tgt, ld_tgt, d, src_idx, tgt_idx, times, n, kernel_num);
    if (st != GDA_OK) return st;
    if (!bandwidth || !l2_saved || !grad_loss || !grad_rows || !workspace) return GDA_E_NULL;
    MmdWs ws = carve(workspace, times, n, d);
    if (workspace_bytes < ws.total) return GDA_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t m = 2 * n;
    const Rows R = make_rows(src, ld_src, tgt, ld_tgt, src_idx, tgt_idx, n);
    const KParams kp{kernel_mul, kernel_num, 0.f};
    const int64_t ntiles = gda_cdiv(m, BJ);
    const int nseg = (int)(ntiles < BWD_NSEG ? ntiles : BWD_NSEG);
    const dim3 grid((unsigned)gda_cdiv(m, BI), (unsigned)(gda_cdiv(d, DC) * nseg), (unsigned)times);
    if (kernel_num == 5) k_bwd<5><<<grid, TB, 0, stream>>>(R, d, m, l2_saved, bandwidth, kp, grad_loss, times, nseg, ws.bwd_part);
    else k_bwd<0><<<grid, TB, 0, stream>>>(R, d, m, l2_saved, bandwidth, kp, grad_loss, times, nseg, ws.bwd_part);
    GDA_LAUNCH_CHECK();
    const int64_t total = (int64_t)times * m * d;
    int64_t rg = gda_cdiv(total, TB);
    if (rg > 4096) rg = 4096;
    k_bwd_reduce<<<(unsigned)rg, TB, 0, stream>>>(ws.bwd_part, m * d, nseg, times, grad_rows);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}
