// fp32 GEMMs of the hidden / classifier projections on the gfx950 matrix cores, for the shapes this
// path has: one TALL operand (rows = nodes, 10^3..10^7) against a SMALL weight (<= 256 x 256).
//
// Replaces the dense part of `self.lin(x)` in PropGCNConv / CachedGCNConv / GCNConv
// (pygda/nn/prop_gcn_conv.py:204, cached_gcn_conv.py:129) and its two autograd GEMMs.  The BLAS
// picks 128x128 macro-tiles for these shapes (74 workgroups for 9360 rows: under a third of the
// chip, 14 us for a 0.3 GFLOP product that moves 10 MB); here a workgroup owns a 64x64 tile
// (294 workgroups), one 32x32 v_mfma_f32_32x32x2_f32 accumulator per wave, 32-deep K chunks staged
// k-major in LDS so that both operands are conflict-free ds_read_b32 -- the tiling of k_pairdist
// (gda_mmd.hip).  The fp32 MFMA is an exact k-ordered fma chain: results differ from the BLAS only
// by summation order.
//
//   NT  C[i,n] = sum_k A[i,k] B[n,k]   forward  y  = x W^T      (A tall)
//   NN  C[i,n] = sum_k A[i,k] B[k,n]   dgrad    gx = gy W       (A tall)
//   TN  C[m,n] = sum_i A[i,m] B[i,n]   wgrad    gW = gy^T x     (reduction over the tall dimension:
//                                      split into row slabs, partial tiles summed in slab order --
//                                      a deterministic split-K)
#include "gda_common.h"

namespace {

constexpr int TB = 256;
constexpr int TILE = 64;
constexpr int DK = 32;
constexpr int LDT = TILE + 4;
constexpr int PF = 4;           // K chunks in flight per workgroup

using f32x16 = __attribute__((ext_vector_type(16))) float;

// element (r, c) of a row-major matrix with bounds (zero outside), 4 consecutive columns
__device__ __forceinline__ float4 ld4(const float* __restrict__ p, int64_t ld, int64_t r, int64_t c,
                                      int64_t rows, int64_t cols, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r >= rows) return v;
    const float* q = p + r * ld + c;
    if (vec && c + 3 < cols) return *reinterpret_cast<const float4*>(q);
    if (c + 0 < cols) v.x = q[0];
    if (c + 1 < cols) v.y = q[1];
    if (c + 2 < cols) v.z = q[2];
    if (c + 3 < cols) v.w = q[3];
    return v;
}

// branch-free variant for the common case (16-byte aligned rows, column count a multiple of 4, so a
// float4 is inside or outside as a whole): out-of-range addresses are clamped, the value zeroed
__device__ __forceinline__ float4 ld4_fast(const float* __restrict__ p, int64_t ld, int64_t r, int64_t c,
                                           int64_t rows, int64_t cols) {
    const int64_t rr = min(r, rows - 1), cc = min(c, cols - 4);
    float4 v = *reinterpret_cast<const float4*>(p + rr * ld + cc);
    const bool in = r < rows && c < cols;
    v.x = in ? v.x : 0.f; v.y = in ? v.y : 0.f; v.z = in ? v.z : 0.f; v.w = in ? v.w : 0.f;
    return v;
}

// Operand tile into S[k][x] (k-major).  TRANS = false: the matrix is [x, k] row-major (k contiguous):
// each thread loads 4 consecutive k of one x and scatters them; TRANS = true: the matrix is [k, x]
// row-major: rows of the tile are copied as they are.
template <bool TRANS, bool FAST>
__device__ __forceinline__ void stage_load(float4 (&v)[2], const float* __restrict__ p, int64_t ld,
                                           int64_t x0, int64_t k0, int64_t nx, int64_t nk, bool vec) {
    const int tid = threadIdx.x;
    if constexpr (!TRANS) {
        const int lr = tid / 8, kq = (tid % 8) * 4;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            v[q] = FAST ? ld4_fast(p, ld, x0 + lr + 32 * q, k0 + kq, nx, nk)
                        : ld4(p, ld, x0 + lr + 32 * q, k0 + kq, nx, nk, vec);
    } else {
        const int kk = tid / 16, c4 = (tid % 16) * 4;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            v[q] = FAST ? ld4_fast(p, ld, k0 + kk + 16 * q, x0 + c4, nk, nx)
                        : ld4(p, ld, k0 + kk + 16 * q, x0 + c4, nk, nx, vec);
    }
}

template <bool TRANS>
__device__ __forceinline__ void stage_store(float (&S)[DK][LDT], const float4 (&v)[2]) {
    const int tid = threadIdx.x;
    if constexpr (!TRANS) {
        const int lr = tid / 8, kq = (tid % 8) * 4;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = lr + 32 * q;
            S[kq + 0][r] = v[q].x; S[kq + 1][r] = v[q].y; S[kq + 2][r] = v[q].z; S[kq + 3][r] = v[q].w;
        }
    } else {
        const int kk = tid / 16, c4 = (tid % 16) * 4;
#pragma unroll
        for (int q = 0; q < 2; ++q) *reinterpret_cast<float4*>(&S[kk + 16 * q][c4]) = v[q];
    }
}

// C tile (i0.., j0..) = sum over k in [kbeg, kend) of Aop[i][k] * Bop[k][j]
//   TA = false: A is [M, K] (NT / NN);  TA = true: A is [K, M] (TN)
//   TB_ = false: B is [N, K] (NT);      TB_ = true: B is [K, N] (NN / TN)
template <bool TA, bool TB_, bool FAST>
__global__ void __launch_bounds__(TB)
k_gemm(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
       float* __restrict__ C, int64_t ldc, int64_t M, int64_t N, int64_t K, int64_t k_slab, bool vec_a,
       bool vec_b, const float* __restrict__ bias, float* __restrict__ colsum) {
    __shared__ __attribute__((aligned(16))) float As[DK][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[DK][LDT];
    const int64_t i0 = (int64_t)blockIdx.x * TILE, j0 = (int64_t)blockIdx.y * TILE;   // rows on grid.x: 2^31 tiles
    const int64_t kbeg = (int64_t)blockIdx.z * k_slab, kend = min(K, kbeg + k_slab);
    const int tid = threadIdx.x, wave = tid / 64, lane = tid % 64;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
    const int ka = lane >> 5, la = lane & 31;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float csum = 0.f;                  // TN only: column sum of A (= the bias gradient beside the weight gradient)
    const bool do_colsum = TA && colsum != nullptr && blockIdx.y == 0 && tid < TILE;
    // A block is alone on its CU at these grid sizes, so nothing hides a chunk's global-load latency
    // but the block itself: PF chunks are kept in flight in registers (all of K for the 128-wide
    // hidden layers), refilled as they are consumed.
    float4 va[PF][2], vb[PF][2];
#pragma unroll
    for (int c = 0; c < PF; ++c)
        if (kbeg + c * DK < kend) {
            stage_load<TA, FAST>(va[c], A, lda, i0, kbeg + c * DK, M, kend, vec_a);
            stage_load<TB_, FAST>(vb[c], B, ldb, j0, kbeg + c * DK, N, kend, vec_b);
        }
    for (int64_t kb = kbeg; kb < kend; kb += PF * DK) {
#pragma unroll
        for (int c = 0; c < PF; ++c) {
            const int64_t k0 = kb + c * DK;
            if (k0 >= kend) break;                                   // block-uniform
            __syncthreads();
            stage_store<TA>(As, va[c]);
            stage_store<TB_>(Bs, vb[c]);
            if (k0 + PF * DK < kend) {
                stage_load<TA, FAST>(va[c], A, lda, i0, k0 + PF * DK, M, kend, vec_a);
                stage_load<TB_, FAST>(vb[c], B, ldb, j0, k0 + PF * DK, N, kend, vec_b);
            }
            __syncthreads();
            if (do_colsum) {
#pragma unroll
                for (int kk = 0; kk < DK; ++kk) csum += As[kk][tid];      // rows beyond K were staged as zeros
            }
#pragma unroll
            for (int kk = 0; kk < DK; kk += 2)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[kk + ka][wi + la], Bs[kk + ka][wj + la], acc, 0, 0, 0);
        }
    }
    float* out = C + (int64_t)blockIdx.z * M * ldc;              // split-K: slab z writes its own [M, ldc] plane
    const int64_t j = j0 + wj + la;
    const float bj = (bias != nullptr && j < N) ? bias[j] : 0.f;  // forward epilogue: + bias along N
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t i = i0 + wi + (r & 3) + 8 * (r >> 2) + 4 * ka;   // C/D layout of the 32x32 MFMA
        if (i < M && j < N) out[i * ldc + j] = acc[r] + bj;
    }
    if (do_colsum && i0 + tid < M) colsum[(int64_t)blockIdx.z * M + i0 + tid] = csum;   // slab z's partial
}

// C[m, n] = sum_z partial[z][m][n] in a fixed order; the extra M entries (idx >= M*N) are the column sums.
// 64 outputs per workgroup: wave w adds the slabs z = w, w + 4, ... in increasing z (256-byte coalesced reads per
// slab), the four wave sums are combined as (a0 + a1) + (a2 + a3) -- the values of the one-thread-per-output loop
// this replaces, on four times as many workgroups and a quarter of the dependent loads per thread (the kernel
// is a latency chain: 15 us for 4 MB at 64 slabs before).
constexpr int SS_OUT = 64;

__global__ void __launch_bounds__(TB)
k_slab_sum(const float* __restrict__ partial, int slabs, int64_t M, int64_t N, float* __restrict__ C,
           int64_t ldc, const float* __restrict__ cs_partial, float* __restrict__ colsum) {
    __shared__ float part[4][SS_OUT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t idx = (int64_t)blockIdx.x * SS_OUT + lane;
    const int64_t plane = M * N;
    const bool is_c = idx < plane;
    const bool is_cs = !is_c && colsum != nullptr && idx - plane < M;
    const float* src = is_c ? partial + idx : cs_partial + (idx - plane);
    const int64_t step = is_c ? plane : M;
    float a = 0.f;
    if (is_c || is_cs) {
        int z = wave;
        for (; z + 12 < slabs; z += 16) {                             // four loads in flight, added in z order
            const float v0 = src[(int64_t)z * step], v1 = src[(int64_t)(z + 4) * step];
            const float v2 = src[(int64_t)(z + 8) * step], v3 = src[(int64_t)(z + 12) * step];
            a += v0; a += v1; a += v2; a += v3;
        }
        for (; z < slabs; z += 4) a += src[(int64_t)z * step];
    }
    part[wave][lane] = a;
    __syncthreads();
    if (wave == 0) {
        const float r = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        if (is_c) C[(idx / N) * ldc + idx % N] = r;
        else if (is_cs) colsum[idx - plane] = r;
    }
}

bool vec_ok(const float* p, int64_t ld) { return ld % 4 == 0 && ((uintptr_t)p & 15) == 0; }

int slabs_for(int64_t K) {
    int64_t s = gda_cdiv(K, 256);
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return (int)s;
}

}  // namespace

// the reduction dimension is the tall one (nodes) and the output small: TN always, NN when its K is the tall
// side (the weight gradient from a column-major gradient: gW = gT x with gT [out, nodes])
bool split_k(int mode, int64_t M, int64_t N, int64_t K) {
    return mode == GDA_GEMM_TN || (mode == GDA_GEMM_NN && K >= 1024 && M <= 512 && N <= 512);
}

extern "C" size_t gda_gemm_workspace_bytes(int mode, int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0 || !split_k(mode, M, N, K)) return 0;
    const int s = slabs_for(K);
    return s > 1 ? (size_t)s * (size_t)M * (size_t)(N + 1) * sizeof(float) : 0;     // + one column-sum partial per slab
}

extern "C" int gda_gemm_f32(int mode, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                            const float* B, int64_t ldb, float* C, int64_t ldc,
                            void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    return gda_gemm_ex_f32(mode, M, N, K, A, lda, B, ldb, C, ldc, nullptr, nullptr, workspace, workspace_bytes, stream_);
}

extern "C" int gda_gemm_ex_f32(int mode, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                               const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias,
                               float* colsum, void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    if (bias != nullptr && mode != GDA_GEMM_NT) return GDA_E_UNSUPPORTED;
    if (colsum != nullptr && mode != GDA_GEMM_TN) return GDA_E_UNSUPPORTED;
    if (mode != GDA_GEMM_NT && mode != GDA_GEMM_NN && mode != GDA_GEMM_TN) return GDA_E_UNSUPPORTED;
    if (M < 0 || N < 0 || K < 0 || ldc < N) return GDA_E_SIZE;
    if ((mode == GDA_GEMM_TN ? lda < M : lda < K) || (mode == GDA_GEMM_NT ? ldb < K : ldb < N)) return GDA_E_SIZE;
    if (M == 0 || N == 0) return GDA_OK;
    if (!C || (K > 0 && (!A || !B))) return GDA_E_NULL;
    if (C == A || C == B) return GDA_E_ALIAS;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t gx = gda_cdiv(N, TILE), gy = gda_cdiv(M, TILE);
    if (gx > 65535 || gy > INT32_MAX) return GDA_E_SIZE;          // row tiles ride on grid.x, column tiles on grid.y
    if (K == 0) {
        GDA_HIP_TRY(hipMemset2DAsync(C, sizeof(float) * ldc, 0, sizeof(float) * N, (size_t)M, stream));
        return GDA_OK;
    }
    const bool va = vec_ok(A, lda), vb = vec_ok(B, ldb);
    // branch-free loads need whole float4s: the contiguous extent of each operand a multiple of 4
    const int64_t a_cols = mode == GDA_GEMM_TN ? M : K, b_cols = mode == GDA_GEMM_NT ? K : N;
    const bool fast = va && vb && a_cols % 4 == 0 && b_cols % 4 == 0 && a_cols >= 4 && b_cols >= 4;
#define GDA_GEMM_LAUNCH(TA_, TB__, grid, Cp, ldc_, kslab)                                                   \
    do {                                                                                                  \
        if (fast) k_gemm<TA_, TB__, true><<<grid, TB, 0, stream>>>(A, lda, B, ldb, Cp, ldc_, M, N, K, kslab, va, vb, bias, csp); \
        else k_gemm<TA_, TB__, false><<<grid, TB, 0, stream>>>(A, lda, B, ldb, Cp, ldc_, M, N, K, kslab, va, vb, bias, csp);    \
    } while (0)
    float* csp = colsum;                         // where the kernel writes its column-sum (partials)
    if (split_k(mode, M, N, K)) {
        const bool tn = mode == GDA_GEMM_TN;
        const int s = slabs_for(K);
        const int64_t k_slab = gda_cdiv(gda_cdiv(K, s), DK) * DK;        // whole chunks per slab
        if (s > 1) {
            if (!workspace || workspace_bytes < gda_gemm_workspace_bytes(mode, M, N, K)) return GDA_E_WORKSPACE;
            float* cs_part = (float*)workspace + (size_t)s * M * N;
            if (colsum) csp = cs_part;
            if (tn) GDA_GEMM_LAUNCH(true, true, dim3((unsigned)gy, (unsigned)gx, s), (float*)workspace, N, k_slab);
            else GDA_GEMM_LAUNCH(false, true, dim3((unsigned)gy, (unsigned)gx, s), (float*)workspace, N, k_slab);
            GDA_LAUNCH_CHECK();
            k_slab_sum<<<(unsigned)gda_cdiv(M * N + (colsum ? M : 0), SS_OUT), TB, 0, stream>>>(
                (const float*)workspace, s, M, N, C, ldc, cs_part, colsum);
        } else if (tn) {
            GDA_GEMM_LAUNCH(true, true, dim3((unsigned)gy, (unsigned)gx, 1), C, ldc, K);
        } else {
            GDA_GEMM_LAUNCH(false, true, dim3((unsigned)gy, (unsigned)gx, 1), C, ldc, K);
        }
    } else if (mode == GDA_GEMM_NT) {
        GDA_GEMM_LAUNCH(false, false, dim3((unsigned)gy, (unsigned)gx, 1), C, ldc, K);
    } else {
        GDA_GEMM_LAUNCH(false, true, dim3((unsigned)gy, (unsigned)gx, 1), C, ldc, K);
    }
#undef GDA_GEMM_LAUNCH
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}
