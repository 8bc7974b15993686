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
#include "gda_philox.h"

#include <cstdlib>

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

// ---- tall products at 10^5 .. 10^7 rows (sampled sub-graphs of the cfg-S regime) -------------------------------------
// Above ~50 k rows the 64 x 64-tile kernel above loses to the BLAS's 128 x 128 macro-tiles by 20-35 %: both operands go
// through LDS for a single 32 x 32 accumulator per wave.  At these shapes one operand is a WEIGHT of at most 256 x 256:
// it fits the register file of a workgroup.
//
// k_tall_fwd16 (NT forward  y = x W^T,  NN dgrad  gx = gy W): a persistent workgroup of 8 waves owns 128 output columns,
// wave w the 16 columns [16 w, 16 w + 16); its slice of the weight -- every group of 4 reduction indices x 16 columns --
// lives in K/4 registers per lane as ready MFMA B operands for the whole kernel.  Only the tall operand moves: 128-row
// tiles in 32-deep chunks through a double-buffered, row-padded LDS image, one barrier per chunk, the next chunk's
// global loads issued before the current chunk's 64 MFMAs per wave (8 row sub-tiles x 8 k-groups) and stored
// mid-chunk, the pipeline running on across tile boundaries.  Per MFMA: ONE ds_read_b32 (the 64 x 64-tile kernel: two,
// the 2 x 2-tile probe of round 2: one, plus the weight's staging).
//
// k_tall_wgrad (TN  gW = gy^T x, reduction over the rows): both operands are consumed along their rows -- lane (c, kk)
// of the A operand is gy[m + kk][n0 + c], of the B operand x[m + kk][32 t + c] -- so 32-row chunks of gy and x are
// staged in LDS exactly as they lie in memory (16-byte loads and stores, no transposition; the operand fetch "one row,
// 32 consecutive columns" is conflict-free), double-buffered, and the whole 128 x K' output of a row slab stays in
// accumulators: 8 waves, wave w = output rows [32 (w & 3), +32) x one half of the K' columns.  One workgroup per slab
// (256 slabs at 10^5 rows), slab partials summed in slab order by k_slab_sum (deterministic), the bias gradient
// (column sums of gy) as a by-product of the A operand registers.  (First version: operands straight from global
// memory, 9 dword loads per 8 MFMAs and wave: 160 us at 150 k x 256 x 128; staged: 117; 8 waves: 101; the BLAS: 89.)
//
// Measured (tools/gemm_bench.py, profiles/r3_gemm_bench.jsonl; fraction of the 157.3 TF fp32 MFMA peak; a register-only
// MFMA loop reaches 95-99 % of it on this part, tools/ubench/mfma_peak.hip): forward 54-67 %, data gradient 59-62 %,
// weight gradient 51-66 % between 150 k and 300 k rows -- the BLAS 49-69 / 48-62 / 63-70 %; ahead of it at 300 k rows
// (forward, data gradient), 3-12 % behind at 150 k.  By the counters on the first (32x32x2, one wave per SIMD) forward
// kernel (rocprofv3 --pmc, profiles/r3_gemm_pmc.txt) the matrix pipe was busy 64 % of the time, its wave parked at the
// chunk barrier / waitcnt 13 % and issuing LDS reads and register moves 16 % of the time.  Tried on the 16x16x4 kernel
// and dropped: contiguous row ranges per workgroup with a partial last tile + loads two chunks ahead (117 us at
// 150 k x 256 x 128 against 100 for the round-robin tiles with loads one chunk ahead).
constexpr int TALL_BM = 128;
constexpr int TALL_TB = 512;          // k_tall_wgrad: 8 waves (2 per SIMD; its accumulators are half the forward kernel's):
                                      // 117 -> 101 us at 150 k x 256 x 128.  The forward kernel stays at 4 waves: with the weight
                                      // slice, 4 accumulators and the operand prefetch a wave needs > 256 registers, and two
                                      // workgroups per CU spill (measured: 119 us with 8 waves against 106 with 4)

// k_tall_fwd16 runs on v_mfma_f32_16x16x4_f32: a wave owns 16 output columns, so its weight slice is K/4 registers
// (64 at K = 256) and a 16-row sub-tile's accumulator 4 -- 8 waves = TWO per SIMD fit the register file.  (First
// version on v_mfma_f32_32x32x2_f32, 4 waves x 32 columns: K/2 weight registers + 64 accumulators = one wave per SIMD,
// whose matrix pipe idles while it issues LDS reads and waits: 106 us at 150 k x 256 x 128 against 100 for this one.)
// LDS image row-major with a row stride == 2 mod 32 words: the A-operand fetch (16 rows x 2 k per 32-lane group) is
// conflict-free.
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int KQ, bool BT, int CQ /* k-groups of 4 per chunk: 8 (32-deep) or 16 (64-deep: half the barriers) */>
__global__ void __launch_bounds__(TALL_TB, 1)
k_tall_fwd16(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb, float* __restrict__ C,
             int64_t ldc, int64_t M, const float* __restrict__ bias) {
    constexpr int NC = KQ / CQ;                       // chunks of the reduction
    constexpr int CK = 4 * CQ;                        // reduction depth of a chunk
    constexpr int LD = CK + 2;                        // row stride of the LDS image (== 2 mod 32: conflict-free fetches)
    constexpr int PQ = TALL_BM * CK / 4 / TALL_TB;    // 16-byte pieces per thread and chunk
    constexpr int RSTEP = TALL_TB / (CK / 4);         // rows between a thread's pieces
    static_assert(NC % 2 == 0, "the buffer index of a chunk is its parity");
    extern __shared__ __attribute__((aligned(16))) float tf_lds[];
    float* As = tf_lds;                               // [2][TALL_BM * LD]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kk = lane >> 4, lc = lane & 15;         // operand layouts: A[row = lc][k = kk], B[k = kk][col = lc]
    const int64_t j0 = (int64_t)blockIdx.y * 128 + 16 * wave;
    float breg[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q)
        breg[q] = BT ? B[(int64_t)(4 * q + kk) * ldb + j0 + lc] : B[(j0 + lc) * ldb + 4 * q + kk];
    const float bv = bias ? bias[j0 + lc] : 0.f;
    const int64_t ntiles = (M + TALL_BM - 1) / TALL_BM;
    const int sr = tid / (CK / 4), sk = (tid % (CK / 4)) * 4;      // staging role: rows sr + RSTEP q; k piece sk .. sk + 3
    float4 v[PQ];
    int64_t tile = blockIdx.x;
    if (tile >= ntiles) return;
#define TF_FETCH(TILE, CH)                                                                             \
    _Pragma("unroll") for (int q = 0; q < PQ; ++q) {                                                  \
        int64_t r = (TILE) * TALL_BM + sr + RSTEP * q;                                                \
        r = r < M ? r : M - 1;      /* rows past the end: any valid address; their outputs are not stored */ \
        v[q] = *reinterpret_cast<const float4*>(A + r * lda + (CH) * CK + sk);                        \
    }
#define TF_STASH(BUF)                                                                                 \
    _Pragma("unroll") for (int q = 0; q < PQ; ++q) {                                                  \
        float2* d = reinterpret_cast<float2*>(As + (BUF) * TALL_BM * LD + (sr + RSTEP * q) * LD + sk); \
        d[0] = make_float2(v[q].x, v[q].y); d[1] = make_float2(v[q].z, v[q].w);                       \
    }
    TF_FETCH(tile, 0)
    TF_STASH(0)
    __syncthreads();
    const float* ar = As + lc * LD + kk;
    for (; tile < ntiles; tile += gridDim.x) {
        f32x4 acc[8];
#pragma unroll
        for (int ms = 0; ms < 8; ++ms)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[ms][r] = 0.f;
        const bool more_tiles = tile + gridDim.x < ntiles;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const bool next = c + 1 < NC || more_tiles;
            if (next) { TF_FETCH(c + 1 < NC ? tile : tile + gridDim.x, c + 1 < NC ? c + 1 : 0) }
            __builtin_amdgcn_sched_barrier(0);        // the loads go out before the chunk's MFMAs, not between them
            const float* a = ar + (c & 1) * (TALL_BM * LD);
#pragma unroll
            for (int q = 0; q < CQ; ++q) {
#pragma unroll
                for (int ms = 0; ms < 8; ++ms)
                    acc[ms] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ms * 16 * LD + 4 * q], breg[c * CQ + q], acc[ms], 0, 0, 0);
                if (q == CQ / 2 - 1) {                // mid-chunk: the other image is free since the last barrier
                    __builtin_amdgcn_sched_barrier(0);
                    if (next) { TF_STASH((c + 1) & 1) }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int ms = 0; ms < 8; ++ms)
#pragma unroll
            for (int r = 0; r < 4; ++r) {             // C/D layout of the 16x16 MFMA: column lane % 16, rows 4 (lane / 16) + r
                const int64_t i = tile * TALL_BM + 16 * ms + 4 * kk + r;
                if (i < M) C[i * ldc + j0 + lc] = acc[ms][r] + bv;
            }
    }
#undef TF_FETCH
#undef TF_STASH
}

#include "gda_gemm_split.inc"

template <int NT8>
__global__ void __launch_bounds__(TALL_TB, NT8 == 4 ? 2 : 1)
k_tall_wgrad(const float* __restrict__ G, int64_t ldg, const float* __restrict__ X, int64_t ldx, float* __restrict__ P,
             float* __restrict__ CS, int64_t M, int64_t rows_per_slab, const int64_t* __restrict__ xrow = nullptr) {
    // xrow (may be NULL): row r of x is X[xrow[r]] (the batch's feature gather done by the operand fetch)
    constexpr int XC = 32 * NT8, RC = 32;             // x columns; rows per chunk
    constexpr int HT = NT8 / 2;                       // column tiles per wave (the two waves of a row group split them)
    constexpr int XQ = RC * XC / 4 / TALL_TB, GQ = RC * 128 / 4 / TALL_TB;      // 16-byte pieces per thread and chunk
    extern __shared__ __attribute__((aligned(16))) float tw_lds[];
    float* Xs = tw_lds;                               // [2][RC][XC]  rows as they lie in memory: the B-operand fetch
    float* Gs = tw_lds + 2 * RC * XC;                 // [2][RC][128] "one row, 32 consecutive columns" is conflict-free
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int ka = lane >> 5, la = lane & 31;
    const int n0 = 32 * (wave & 3), t0 = HT * (wave >> 2);
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_slab, r1 = min(M, r0 + rows_per_slab);
    f32x16 acc[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float csum = 0.f;
    float4 vx[XQ], vg[GQ];
    // (macros, not lambdas: with the staging arrays captured by reference the compiler kept them in scratch memory)
#define TW_FETCH(M0)                                                                                       \
    _Pragma("unroll") for (int q = 0; q < XQ; ++q) {                                                       \
        const int idx = tid + TALL_TB * q, row = idx / (XC / 4), c4 = idx % (XC / 4);                      \
        int64_t r = (M0) + row;                                                                            \
        r = r < r1 ? r : r1 - 1;              /* past the slab: the last row, with gy as zero */           \
        if (xrow) r = xrow[r];                                                                             \
        vx[q] = *reinterpret_cast<const float4*>(X + r * ldx + 4 * c4);                                    \
    }                                                                                                      \
    _Pragma("unroll") for (int q = 0; q < GQ; ++q) {                                                       \
        const int idx = tid + TALL_TB * q, row = idx / 32, c4 = idx % 32;                                  \
        const int64_t r = (M0) + row;                                                                      \
        const bool in = r < r1;                                                                            \
        float4 g = *reinterpret_cast<const float4*>(G + (in ? r : r1 - 1) * ldg + 4 * c4);                 \
        g.x = in ? g.x : 0.f; g.y = in ? g.y : 0.f; g.z = in ? g.z : 0.f; g.w = in ? g.w : 0.f;            \
        vg[q] = g;                                                                                         \
    }
#define TW_STASH(BUF)                                                                                      \
    _Pragma("unroll") for (int q = 0; q < XQ; ++q)                                                         \
        *reinterpret_cast<float4*>(Xs + (BUF) * RC * XC + 4 * (tid + TALL_TB * q)) = vx[q];                \
    _Pragma("unroll") for (int q = 0; q < GQ; ++q)                                                         \
        *reinterpret_cast<float4*>(Gs + (BUF) * RC * 128 + 4 * (tid + TALL_TB * q)) = vg[q];
    if (r0 < r1) {
        TW_FETCH(r0)
        TW_STASH(0)
        __syncthreads();
        const float* xb = Xs + ka * XC + 32 * t0 + la;
        const float* gb = Gs + ka * 128 + n0 + la;
        int buf = 0;
        for (int64_t m0 = r0; m0 < r1; m0 += RC) {
            TW_FETCH(m0 + RC)                         // unconditional: the loads go out before the chunk's MFMAs
            __builtin_amdgcn_sched_barrier(0);
            const float* xs = xb + buf * RC * XC;
            const float* gs = gb + buf * RC * 128;
#pragma unroll
            for (int p = 0; p < RC / 2; ++p) {
                const float a = gs[2 * p * 128];
                csum += a;
#pragma unroll
                for (int t = 0; t < HT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xs[2 * p * XC + 32 * t], acc[t], 0, 0, 0);
                if (p == RC / 4 - 1) {                // mid-chunk: the other image is free since the last barrier
                    __builtin_amdgcn_sched_barrier(0);
                    TW_STASH(buf ^ 1)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();
            buf ^= 1;
        }
    }
#undef TW_FETCH
#undef TW_STASH
    float* out = P + (int64_t)blockIdx.x * 128 * XC;
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            out[(int64_t)(n0 + (r & 3) + 8 * (r >> 2) + 4 * ka) * XC + 32 * (t0 + t) + la] = acc[t][r];
    if (CS && t0 == 0) {                               // rows m + 0 and m + 1 of every pair sit in the two lane halves
        const float other = __shfl_down(csum, 32, 64);
        if (ka == 0) CS[(int64_t)blockIdx.x * 128 + n0 + la] = csum + other;
    }
}

// k_tall_wgrad with its operand chunks arriving by LDS-DMA (round 6).  The register-staged kernel above issues ONE chunk of
// loads per iteration and stashes it half a chunk of MFMAs later: whenever the loads take longer than that half (they do:
// ~1.2 us against 0.85 us at 128 columns) every wave of the workgroup stands at the stash, and the compiler's vmcnt(0)
// there defeats any deeper register prefetch (measured: tall_wgrad_launch).  tools/ubench/wgrad_stream.hip streams the same
// operands through the same launch shape, LDS and barrier included, in 24 us -- the 60 us of the real kernels are this
// exposed latency plus the MFMAs, not the memory system.  The LDS image of a chunk is the chunk's rows exactly as they lie
// in memory, which is what global_load_lds_dwordx4 writes (1 KB per wave instruction, LDS address = M0 + 16 lane): no
// staging registers, no stash phase, and the prefetch depth is a matter of LDS -- THREE chunk buffers, two chunks of loads
// in flight while the third is multiplied; one barrier per chunk.  The loads are inline assembly with one explicit
// s_waitcnt per chunk (behind the builtin the compiler waits for every transfer before the next LDS read).  A partial last
// chunk (rows past the slab: gy must read as zero) goes through the register path.  Same accumulation order as
// k_tall_wgrad: bit-identical results.
template <int NT8>
__global__ void __launch_bounds__(TALL_TB, 1)
k_tall_wgrad_dma(const float* __restrict__ G, int64_t ldg, const float* __restrict__ X, int64_t ldx, float* __restrict__ P,
                 float* __restrict__ CS, int64_t M, int64_t rows_per_slab, const int64_t* __restrict__ xrow) {
    constexpr int XC = 32 * NT8, RC = 32, HT = NT8 / 2;
    constexpr int XB = RC * XC * 4, GB = RC * 128 * 4, BUFB = XB + GB;          // bytes: x part, gy part, one chunk buffer
    constexpr int NI = BUFB / 1024, IPW = NI / 8;                              // DMA instructions per chunk / per wave
    constexpr int XQ = RC * XC / 4 / TALL_TB, GQ = RC * 128 / 4 / TALL_TB;
    static_assert(NI % 8 == 0, "every wave issues the same number of transfers");
    extern __shared__ __attribute__((aligned(16))) char td_lds[];               // [3][ x [RC][XC] | gy [RC][128] ]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int ka = lane >> 5, la = lane & 31;
    const int n0 = 32 * (wave & 3), t0 = HT * (wave >> 2);
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_slab, r1 = min(M, r0 + rows_per_slab);
    const int64_t nfull = r0 < r1 ? (r1 - r0) / RC : 0;                         // whole chunks; the rest takes the register path
    f32x16 acc[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float csum = 0.f;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)td_lds;
    // gathered x: the slab's row ids are staged in LDS ONCE (behind the three buffers) -- read per transfer from global memory
    // they would sit in the same in-order vmcnt queue as the transfers and every wait for an id would drain the prefetch
    int64_t* ids = reinterpret_cast<int64_t*>(td_lds + 3 * BUFB);
    if (xrow) {
        for (int64_t i = tid; i < r1 - r0; i += TALL_TB) ids[i] = xrow[r0 + i];
        __syncthreads();
    }
    // transfer j of a chunk covers bytes [1024 j, 1024 j + 1024) of its buffer; this lane's 16 bytes sit at 1024 j + 16 lane
#define TD_ISSUE(CH, B_)                                                                                     \
    _Pragma("unroll") for (int k = 0; k < IPW; ++k) {                                                        \
        const int j = wave + 8 * k;                                                                          \
        const int b = 1024 * j + 16 * lane;                                                                  \
        const float* gp;                                                                                     \
        if (b < XB) {                                                                                        \
            int64_t r = r0 + (CH) * RC + b / (XC * 4);                                                       \
            if (xrow) r = ids[(CH) * RC + b / (XC * 4)];                                                     \
            gp = X + r * ldx + (b % (XC * 4)) / 4;                                                           \
        } else {                                                                                             \
            const int bb = b - XB;                                                                           \
            gp = G + (r0 + (CH) * RC + bb / 512) * ldg + (bb % 512) / 4;                                     \
        }                                                                                                    \
        const uint32_t lp = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)((B_) * BUFB + 1024 * j));       \
        uint32_t keep;                                                                                       \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep) : "v"(gp), "s"(lp) : "memory");                                           \
    }
#define TD_MFMA(B_)                                                                                          \
    {                                                                                                        \
        const float* xs = reinterpret_cast<const float*>(td_lds + (B_) * BUFB) + ka * XC + 32 * t0 + la;     \
        const float* gs = reinterpret_cast<const float*>(td_lds + (B_) * BUFB + XB) + ka * 128 + n0 + la;    \
        _Pragma("unroll") for (int p = 0; p < RC / 2; ++p) {                                                 \
            const float a = gs[2 * p * 128];                                                                 \
            csum += a;                                                                                       \
            _Pragma("unroll") for (int t = 0; t < HT; ++t)                                                   \
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xs[2 * p * XC + 32 * t], acc[t], 0, 0, 0);  \
        }                                                                                                    \
    }
    if (nfull > 0) { TD_ISSUE(0, 0) }
    if (nfull > 1) { TD_ISSUE(1, 1) }
    int buf = 0;
    for (int64_t c = 0; c < nfull; ++c) {
        // this wave's transfers of chunk c have landed when at most the IPW newer ones (chunk c + 1) are outstanding
        if (c + 1 < nfull) { if constexpr (IPW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // everybody's transfers of chunk c; everybody done with chunk c - 1's buffer
        if (c + 2 < nfull) { const int nb = buf == 0 ? 2 : buf - 1; TD_ISSUE(c + 2, nb) }
        TD_MFMA(buf)
        buf = buf == 2 ? 0 : buf + 1;
    }
    if (r0 < r1 && r0 + nfull * RC < r1) {             // the partial last chunk, through registers (rows past the end: gy = 0)
        const int64_t m0 = r0 + nfull * RC;
        float4 vx[XQ], vg[GQ];
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int idx = tid + TALL_TB * q, row = idx / (XC / 4), c4 = idx % (XC / 4);
            int64_t r = m0 + row;
            r = r < r1 ? r : r1 - 1;
            if (xrow) r = xrow[r];
            vx[q] = *reinterpret_cast<const float4*>(X + r * ldx + 4 * c4);
        }
#pragma unroll
        for (int q = 0; q < GQ; ++q) {
            const int idx = tid + TALL_TB * q, row = idx / 32, c4 = idx % 32;
            const int64_t r = m0 + row;
            const bool in = r < r1;
            float4 g = *reinterpret_cast<const float4*>(G + (in ? r : r1 - 1) * ldg + 4 * c4);
            g.x = in ? g.x : 0.f; g.y = in ? g.y : 0.f; g.z = in ? g.z : 0.f; g.w = in ? g.w : 0.f;
            vg[q] = g;
        }
        __syncthreads();                               // every wave is past its last MFMAs on the buffers
        float* xw = reinterpret_cast<float*>(td_lds);
        float* gw = reinterpret_cast<float*>(td_lds + XB);
#pragma unroll
        for (int q = 0; q < XQ; ++q) *reinterpret_cast<float4*>(xw + 4 * (tid + TALL_TB * q)) = vx[q];
#pragma unroll
        for (int q = 0; q < GQ; ++q) *reinterpret_cast<float4*>(gw + 4 * (tid + TALL_TB * q)) = vg[q];
        __syncthreads();
        TD_MFMA(0)
    }
#undef TD_ISSUE
#undef TD_MFMA
    float* out = P + (int64_t)blockIdx.x * 128 * XC;
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            out[(int64_t)(n0 + (r & 3) + 8 * (r >> 2) + 4 * ka) * XC + 32 * (t0 + t) + la] = acc[t][r];
    if (CS && t0 == 0) {
        const float other = __shfl_down(csum, 32, 64);
        if (ka == 0) CS[(int64_t)blockIdx.x * 128 + n0 + la] = csum + other;
    }
}

// ---- skinny products: the classifier projection h -> C (C <= 8 classes) at 10^5 rows ---------------------------------
// On the 64 x 64-tile kernel 92 % of such a product's matrix-core work is padding (5 of 64 columns) and its weight
// gradient ran 179 us per cfg-S step; the products are memory-bound streams (read x once / write gx once), so they run on
// the vector ALUs: W (<= 8 x 256) in LDS or registers, 16-byte coalesced accesses of the node matrix, fixed-order sums.
constexpr int SK_N = 8;

// y[M, N] = x[M, K] W[N, K]^T (+ bias): 8 lanes per row, lane j covers the 16-byte pieces j, j + 8, ... of the row
__global__ void __launch_bounds__(TB)
k_skinny_fwd(const float* __restrict__ X, int64_t ldx, const float* __restrict__ W, int64_t ldw, const float* __restrict__ bias,
             float* __restrict__ Y, int64_t ldy, int64_t M, int N, int K) {
    __shared__ __attribute__((aligned(16))) float Ws[SK_N * 256];
    for (int i = threadIdx.x; i < N * K; i += TB) Ws[i] = W[(int64_t)(i / K) * ldw + i % K];
    __syncthreads();
    const int j = threadIdx.x & 7;
    const int pieces = K / 4;
    for (int64_t row = (int64_t)blockIdx.x * (TB / 8) + (threadIdx.x >> 3); row < M; row += (int64_t)gridDim.x * (TB / 8)) {
        float acc[SK_N];
#pragma unroll
        for (int n = 0; n < SK_N; ++n) acc[n] = 0.f;
        for (int p = j; p < pieces; p += 8) {
            const float4 v = *reinterpret_cast<const float4*>(X + row * ldx + 4 * p);
#pragma unroll
            for (int n = 0; n < SK_N; ++n)
                if (n < N) {
                    const float4 w = *reinterpret_cast<const float4*>(Ws + n * K + 4 * p);
                    acc[n] = fmaf(v.x, w.x, acc[n]); acc[n] = fmaf(v.y, w.y, acc[n]);
                    acc[n] = fmaf(v.z, w.z, acc[n]); acc[n] = fmaf(v.w, w.w, acc[n]);
                }
        }
#pragma unroll
        for (int n = 0; n < SK_N; ++n) {                   // butterfly over the row's 8 lanes: every lane ends with the sum
            acc[n] += __shfl_xor(acc[n], 1, 8);
            acc[n] += __shfl_xor(acc[n], 2, 8);
            acc[n] += __shfl_xor(acc[n], 4, 8);
        }
        float out = 0.f;
#pragma unroll
        for (int n = 0; n < SK_N; ++n) out = (j == n) ? acc[n] : out;
        if (j < N) Y[row * ldy + j] = out + (bias ? bias[j] : 0.f);
    }
}

// gx[M, K] = gy[M, N] W[N, K]: K / 4 lanes per row, a 16-byte piece each
__global__ void __launch_bounds__(TB)
k_skinny_dgrad(const float* __restrict__ GY, int64_t ldg, const float* __restrict__ W, int64_t ldw, float* __restrict__ GX,
               int64_t ldx, int64_t M, int N, int K,
               const float* __restrict__ mask = nullptr, int64_t ldm = 0, float mscale = 1.f) {
    const int pieces = K / 4, p = threadIdx.x % pieces, rows_per_block = TB / pieces;
    float4 w[SK_N];
#pragma unroll
    for (int n = 0; n < SK_N; ++n)
        w[n] = n < N ? *reinterpret_cast<const float4*>(W + (int64_t)n * ldw + 4 * p) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int64_t row = (int64_t)blockIdx.x * rows_per_block + threadIdx.x / pieces; row < M;
         row += (int64_t)gridDim.x * rows_per_block) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int n = 0; n < SK_N; ++n)
            if (n < N) {
                const float g = GY[row * ldg + n];
                o.x = fmaf(g, w[n].x, o.x); o.y = fmaf(g, w[n].y, o.y); o.z = fmaf(g, w[n].z, o.z); o.w = fmaf(g, w[n].w, o.w);
            }
        if (mask) {              // the activation's backward in the epilogue (see k_tall_fwd_h)
            const float4 y = *reinterpret_cast<const float4*>(mask + row * ldm + 4 * p);
            o.x = y.x > 0.f ? o.x * mscale : 0.f; o.y = y.y > 0.f ? o.y * mscale : 0.f;
            o.z = y.z > 0.f ? o.z * mscale : 0.f; o.w = y.w > 0.f ? o.w * mscale : 0.f;
        }
        *reinterpret_cast<float4*>(GX + row * ldx + 4 * p) = o;
    }
}

// slab partial of gW[N, K] = gy^T x and of colsum(gy): K / 4 lanes across a row, TB / (K / 4) row lanes, rows in order
__global__ void __launch_bounds__(TB)
k_skinny_wgrad(const float* __restrict__ GY, int64_t ldg, const float* __restrict__ X, int64_t ldx, float* __restrict__ P,
               float* __restrict__ CS, int64_t M, int64_t rows_per_slab, int N, int K) {
    __shared__ float red[TB * 4];
    const int pieces = K / 4, p = threadIdx.x % pieces, rl = threadIdx.x / pieces, rlanes = TB / pieces;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_slab, r1 = min(M, r0 + rows_per_slab);
    float4 acc[SK_N];
    float cs[SK_N];
#pragma unroll
    for (int n = 0; n < SK_N; ++n) { acc[n] = make_float4(0.f, 0.f, 0.f, 0.f); cs[n] = 0.f; }
    // four rows' loads in flight per lane (same accumulation order): one row at a time left every iteration waiting for
    // its own loads -- 55-118 us for an 80 MB operand at 157 k rows
#pragma unroll 4
    for (int64_t row = r0 + rl; row < r1; row += rlanes) {
        const float4 v = *reinterpret_cast<const float4*>(X + row * ldx + 4 * p);
#pragma unroll
        for (int n = 0; n < SK_N; ++n)
            if (n < N) {
                const float g = GY[row * ldg + n];
                acc[n].x = fmaf(g, v.x, acc[n].x); acc[n].y = fmaf(g, v.y, acc[n].y);
                acc[n].z = fmaf(g, v.z, acc[n].z); acc[n].w = fmaf(g, v.w, acc[n].w);
                cs[n] += g;
            }
    }
    // the row lanes' partials are added in row-lane order through LDS (one class at a time)
    float* out = P + (int64_t)blockIdx.x * N * K;
    for (int n = 0; n < N; ++n) {
        __syncthreads();
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        float c = 0.f;
#pragma unroll
        for (int q = 0; q < SK_N; ++q) if (q == n) { a = acc[q]; c = cs[q]; }
        *reinterpret_cast<float4*>(red + 4 * threadIdx.x) = a;
        __syncthreads();
        if (rl == 0) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int r = 0; r < rlanes; ++r) {
                const float4 u = *reinterpret_cast<const float4*>(red + 4 * (r * pieces + p));
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            *reinterpret_cast<float4*>(out + (int64_t)n * K + 4 * p) = t;
        }
        if (CS) {
            __syncthreads();
            if (p == 0) red[rl] = c;
            __syncthreads();
            if (threadIdx.x == 0) {
                float t = 0.f;
                for (int r = 0; r < rlanes; ++r) t += red[r];
                CS[(int64_t)blockIdx.x * N + n] = t;
            }
        }
    }
}

bool vec_ok(const float* p, int64_t ld) { return ld % 4 == 0 && ((uintptr_t)p & 15) == 0; }

// Row slabs of a split-K product: 128 rows each, at most 256 (round 6; 256 rows / at most 64 before: a workgroup then walks
// 8 - 15 chunks of 32 rows one after the other -- AdaGCN's two critic products of 16 k / 31 k rows took 17 / 31 us ten times per
// step: 2.42 -> 2.32 ms/epoch with shorter slabs, cfg-A 0.370 -> 0.367; 64-row slabs cost cfg-A 4 %)
int slabs_for(int64_t K) {
    static const int per = [] { const char* e = std::getenv("PYGDA_AMD_GEMM_SLAB_ROWS"); const int v = e ? std::atoi(e) : 0; return v >= 32 ? v : 128; }();
    static const int cap = [] { const char* e = std::getenv("PYGDA_AMD_GEMM_SLABS_MAX"); const int v = e ? std::atoi(e) : 0; return v >= 1 ? v : 256; }();
    int64_t s = gda_cdiv(K, per);
    if (s < 1) s = 1;
    if (s > cap) s = cap;
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
    if (gda_dbg_skip(mode == GDA_GEMM_TN ? "gemm_ex_tn" : mode == GDA_GEMM_NN ? "gemm_ex_nn" : "gemm_ex_nt")) return GDA_OK;
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
        if (colsum) GDA_HIP_TRY(hipMemsetAsync(colsum, 0, sizeof(float) * (size_t)M, stream));      // an empty sum
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
            GDA_UNLESS_SKIPPED("k_slab_sum") k_slab_sum<<<(unsigned)gda_cdiv(M * N + (colsum ? M : 0), SS_OUT), TB, 0, stream>>>(
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

// ---- tall entry point ------------------------------------------------------------------------------------------------
// Same products as gda_gemm_ex_f32 for the shapes of sampled sub-graphs: M (the node count) large, the weight at most
// 256 x 256 with both extents multiples of 128 resp. 32:
//   NT  C[M, N] = A[M, K] B[N, K]^T (+ bias[N])   N in {128, 256}, K in {128, 256}
//   NN  C[M, N] = A[M, K] B[K, N]                 N in {128, 256}, K in {128, 256}
//   TN  C[128, N] = A[Mrows, 128]^T B[Mrows, N]   (mode TN: M = 128 output rows, K = the tall reduction), N in {128, 256};
//       colsum[128] = column sums of A.  Workspace: gda_gemm_tall_workspace_bytes.
// GDA_E_UNSUPPORTED outside that envelope (callers use gda_gemm_ex_f32).
// Row slabs of the tall weight gradient: one workgroup each, 256 = one per CU.  Measured in round 6 (tools/wgrad_sweep.py,
// 158,720 rows, us for x 128 / 256 wide): 256 slabs 71.3 / 128.5, 384 79.9 / 155.6, 512 (two workgroups per CU for the
// 128-wide form) 72.8 / 146.8, 768 83.0 / 159.1, 1024 86.1 / 164.6 -- the fp32 kernel is bound by its MFMAs (time doubles
// with the width), not by load latency: more workgroups buy nothing.  PYGDA_AMD_WGRAD_SLABS overrides (experiments).
static int64_t tall_wgrad_slabs(int64_t rows, int64_t N) {
    static const int64_t forced = [] { const char* e = std::getenv("PYGDA_AMD_WGRAD_SLABS"); return e ? (int64_t)std::atoi(e) : (int64_t)0; }();
    (void)N;
    return min(forced > 0 ? forced : (int64_t)256, gda_cdiv(rows, 64));
}

// The slab kernel of the tall weight gradient: the fp32-MFMA form (k_tall_wgrad, the default) or split-fp16 MFMAs
// (k_tall_wgrad_h, PYGDA_AMD_WGRAD_SPLIT_F16=1).  Round 6 built the second one expecting the first to be MFMA-bound (its time
// doubles with the width) -- and measured, at 158,720 rows x 128 x {128, 256} (tools/wgrad_sweep.py, us incl. the slab sum):
//   fp32 MFMA 71.0 / 127.9 | split fp16 (5.3 x less matrix time) 72.0 / 122.9 | + chunks dealt round-robin 71.9 / 114.8 |
//   + two chunks of loads in flight 73.3 / - | fp32 with two workgroups per CU 72.8 / 146.8
// and in the cfg-S step 2.26 ms with the fp16 form against 2.17 with the fp32 one.  Every variant streams its 160 - 240 MB at
// 2.3 - 2.7 TB/s while two torch reductions over the same arrays reach 3.7 - 4.5 (tools/stream_probe.py): the kernel is bound
// neither by its MFMAs, nor by load latency, nor by the slabs' address pattern -- what the forms share is 256 - 512
// workgroups of 8 waves staging through LDS behind one barrier per 32 rows.  tools/ubench/wgrad_stream.hip then streamed the
// same operands through the same launch shape, LDS and barrier included, in 24 us: the shape is innocent; what costs is the
// exposed latency of a one-chunk register prefetch (the compiler's vmcnt(0) at the stash defeats a deeper one) on top of the
// arithmetic.  k_tall_wgrad_dma (LDS-DMA, three buffers, fp32 MFMAs; the default, PYGDA_AMD_WGRAD_DMA=0 restores the
// register-staged kernel) hides it: 66.7 / 122.6 us, cfg-S 2.20 -> 2.13 ms/step -- and is then bound by its MFMAs (58 % of
// the fp32 roof).  The fp16 form cannot take the same route cheaply: with DMA the operands arrive as fp32 and every wave
// that consumes an element converts it again (~350 VALU per chunk and wave: the matrix time saved).  It stays in as the
// opt-in it was measured as (same results to 3e-7 of the fp64 product).
static int tall_wgrad_launch(int64_t N, const float* A, int64_t lda, const float* X, int64_t ldx, const int64_t* xrow, float* part,
                             float* cs_part, int64_t Krows, int64_t rows, int64_t slabs, hipStream_t stream) {
    // (rows: contiguous rows per slab; 0 = chunks dealt round-robin -- k_tall_wgrad_h only)
    static const bool split16 = [] {
        const char* e = std::getenv("PYGDA_AMD_GEMM_SPLIT_F16");
        const char* w = std::getenv("PYGDA_AMD_WGRAD_SPLIT_F16");
        return !(e && e[0] == '0') && (w && w[0] == '1');
    }();
    static const bool dealt = [] { const char* e = std::getenv("PYGDA_AMD_WGRAD_DEALT"); return e && e[0] == '1'; }();
    static const bool dma = [] { const char* e = std::getenv("PYGDA_AMD_WGRAD_DMA"); return !(e && e[0] == '0'); }();
    if (split16) {
        if (dealt) rows = 0;
        const size_t lds4 = 2 * (size_t)(2 * 128 * TWH_CS + 2 * 128 * TWH_CS + 128 * 4 + 128 * 4);
        const size_t lds8 = 2 * (size_t)(2 * 128 * TWH_CS + 2 * 256 * TWH_CS + 128 * 4 + 256 * 4);
        GDA_LDS_ATTR_ONCE(k_tall_wgrad_h<4>, 160 * 1024);
        GDA_LDS_ATTR_ONCE(k_tall_wgrad_h<8>, 160 * 1024);
        if (N == 128) k_tall_wgrad_h<4><<<(unsigned)slabs, TALL_TB, lds4, stream>>>(A, lda, X, ldx, part, cs_part, Krows, rows, xrow);
        else k_tall_wgrad_h<8><<<(unsigned)slabs, TALL_TB, lds8, stream>>>(A, lda, X, ldx, part, cs_part, Krows, rows, xrow);
    } else if (dma && lda % 4 == 0 && ldx % 4 == 0 && (!xrow || rows <= 1024)) {
        const size_t lds = (size_t)3 * 32 * (N + 128) * sizeof(float) + (xrow ? (size_t)rows * 8 : 0);   // three chunk buffers (+ row ids)
        GDA_LDS_ATTR_ONCE(k_tall_wgrad_dma<4>, 160 * 1024);
        GDA_LDS_ATTR_ONCE(k_tall_wgrad_dma<8>, 160 * 1024);
        if (N == 128) k_tall_wgrad_dma<4><<<(unsigned)slabs, TALL_TB, lds, stream>>>(A, lda, X, ldx, part, cs_part, Krows, rows, xrow);
        else k_tall_wgrad_dma<8><<<(unsigned)slabs, TALL_TB, lds, stream>>>(A, lda, X, ldx, part, cs_part, Krows, rows, xrow);
    } else {
        const size_t lds = (size_t)2 * 32 * (N + 128) * sizeof(float);           // both images of x and gy chunks
        GDA_LDS_ATTR_ONCE(k_tall_wgrad<4>, 160 * 1024);
        GDA_LDS_ATTR_ONCE(k_tall_wgrad<8>, 160 * 1024);
        if (N == 128) k_tall_wgrad<4><<<(unsigned)slabs, TALL_TB, lds, stream>>>(A, lda, X, ldx, part, cs_part, Krows, rows, xrow);
        else k_tall_wgrad<8><<<(unsigned)slabs, TALL_TB, lds, stream>>>(A, lda, X, ldx, part, cs_part, Krows, rows, xrow);
    }
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" size_t gda_gemm_tall_workspace_bytes(int mode, int64_t M, int64_t N, int64_t K) {
    if (mode != GDA_GEMM_TN || M != 128 || (N != 128 && N != 256) || K <= 0) return 0;
    const int64_t slabs = tall_wgrad_slabs(K, N);
    return (size_t)slabs * 128 * (size_t)(N + 1) * sizeof(float);
}

extern "C" int gda_gemm_tall_f32(int mode, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                                 const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, float* colsum,
                                 void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    if (mode != GDA_GEMM_NT && mode != GDA_GEMM_NN && mode != GDA_GEMM_TN) return GDA_E_UNSUPPORTED;
    if (M <= 0 || N <= 0 || K <= 0 || ldc < N) return GDA_E_SIZE;
    if (!A || !B || !C) return GDA_E_NULL;
    if (C == A || C == B) return GDA_E_ALIAS;
    hipStream_t stream = (hipStream_t)stream_;
    if (mode == GDA_GEMM_TN) {
        if (bias) return GDA_E_UNSUPPORTED;
        if (M != 128 || (N != 128 && N != 256) || lda < 128 || ldb < N) return GDA_E_UNSUPPORTED;
        const int64_t slabs = tall_wgrad_slabs(K, N);
        const int64_t rows = gda_cdiv(gda_cdiv(K, slabs), 8) * 8;
        if (!workspace || workspace_bytes < gda_gemm_tall_workspace_bytes(mode, M, N, K)) return GDA_E_WORKSPACE;
        float* part = (float*)workspace;
        float* cs_part = part + (size_t)slabs * 128 * N;
        if (lda % 4 || ldb % 4 || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return GDA_E_UNSUPPORTED;
        int st = tall_wgrad_launch(N, A, lda, B, ldb, nullptr, part, colsum ? cs_part : nullptr, K, rows, slabs, stream);
        if (st != GDA_OK) return st;
        GDA_UNLESS_SKIPPED("k_slab_sum") k_slab_sum<<<(unsigned)gda_cdiv(128 * N + (colsum ? 128 : 0), SS_OUT), TB, 0, stream>>>(
            part, (int)slabs, 128, N, C, ldc, cs_part, colsum);
        GDA_LAUNCH_CHECK();
        return GDA_OK;
    }
    if (colsum) return GDA_E_UNSUPPORTED;
    if ((N != 128 && N != 256) || (K != 128 && K != 256) || lda < K || lda % 4 || ((uintptr_t)A & 15)) return GDA_E_UNSUPPORTED;
    if (mode == GDA_GEMM_NT ? ldb < K : ldb < N) return GDA_E_SIZE;
    if (bias && mode != GDA_GEMM_NT) return GDA_E_UNSUPPORTED;
    // forward / data gradient on the 16-bit matrix cores with split operands (gda_gemm_split.inc): the default
    static const bool split16 = [] { const char* e = std::getenv("PYGDA_AMD_GEMM_SPLIT_F16"); return !(e && e[0] == '0'); }();
    if (split16 && ldc % 4 == 0 && ((uintptr_t)C & 15) == 0 && ldb % 4 == 0 && ((uintptr_t)B & 15) == 0) {
        const int64_t bm = K == 128 ? 128 : 64;
        const int64_t nt = gda_cdiv(M, bm);
        const dim3 g((unsigned)min(nt, (int64_t)256), (unsigned)(N / 128));
        const size_t img = (size_t)bm * (K + 8) * 2;
        const size_t lds = 4 * img + 2 * (size_t)bm * sizeof(float);
#ifdef GDA_MEASUREMENT_AIDS
        static const int dbg = [] { const char* e = std::getenv("PYGDA_AMD_GEMM_DBG"); return e ? std::atoi(e) : 0; }();
#define TH_DBG(D_)                                                                                            \
    do {                                                                                                      \
        GDA_LDS_ATTR_ONCE((k_tall_fwd_h<128, false, D_>), 160 * 1024);                                        \
        k_tall_fwd_h<128, false, D_><<<g, TALL_TB, lds, stream>>>(A, lda, B, ldb, C, ldc, M, bias);           \
        GDA_LAUNCH_CHECK();                                                                                   \
        return GDA_OK;                                                                                        \
    } while (0)
        if (dbg && mode == GDA_GEMM_NT && K == 128) {          // phase-elimination probes (tools/gemm_probe.py): not a product
            if (dbg == 1) TH_DBG(1); if (dbg == 2) TH_DBG(2); if (dbg == 4) TH_DBG(4); if (dbg == 3) TH_DBG(3);
            if (dbg == 6) TH_DBG(6); if (dbg == 7) TH_DBG(7);
        }
#undef TH_DBG
#endif
#define TH_LAUNCH(K_, BT_)                                                                                    \
    do {                                                                                                      \
        GDA_LDS_ATTR_ONCE((k_tall_fwd_h<K_, BT_>), 160 * 1024);                                               \
        k_tall_fwd_h<K_, BT_><<<g, TALL_TB, lds, stream>>>(A, lda, B, ldb, C, ldc, M, bias);                  \
    } while (0)
        if (mode == GDA_GEMM_NT) {
            if (K == 128) TH_LAUNCH(128, false); else TH_LAUNCH(256, false);
        } else {
            if (K == 128) TH_LAUNCH(128, true); else TH_LAUNCH(256, true);
        }
#undef TH_LAUNCH
        GDA_LAUNCH_CHECK();
        return GDA_OK;
    }
    const int64_t ntiles = gda_cdiv(M, TALL_BM);
    const dim3 grid((unsigned)min(ntiles, (int64_t)256), (unsigned)(N / 128));
    // 64-deep chunks (one barrier per 128 MFMAs of a wave; 32-deep: 186 -> 183 us forward, 201 -> 187 us data gradient
    // at 300 k x 256 x 128)
#define TF_LAUNCH(KQ_, BT_)                                                                                   \
    do {                                                                                                      \
        const size_t lds = (size_t)2 * TALL_BM * (4 * 16 + 2) * sizeof(float);                                \
        GDA_LDS_ATTR_ONCE((k_tall_fwd16<KQ_, BT_, 16>), 160 * 1024);                                          \
        k_tall_fwd16<KQ_, BT_, 16><<<grid, TALL_TB, lds, stream>>>(A, lda, B, ldb, C, ldc, M, bias);          \
    } while (0)
    if (mode == GDA_GEMM_NT) {
        if (K == 128) TF_LAUNCH(32, false); else TF_LAUNCH(64, false);
    } else {
        if (K == 128) TF_LAUNCH(32, true); else TF_LAUNCH(64, true);
    }
#undef TF_LAUNCH
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

// ---- skinny entry point ----------------------------------------------------------------------------------------------
// The classifier projection's three products (at most 8 classes against a width of 32 / 64 / 128 / 256):
//   NT  C[M, N] = A[M, K] B[N, K]^T (+ bias)     N <= 8, K in {32, 64, 128, 256}
//   NN  C[M, N] = A[M, K] B[K, N]                K <= 8, N in {32, 64, 128, 256}      (data gradient)
//   TN  C[M, N] = A[rows, M]^T B[rows, N]        M <= 8, N in {32, 64, 128, 256}, K = rows; colsum[M] = A's column sums
// 16-byte aligned operands with leading dimensions % 4 == 0 where rows are read in 16-byte pieces.
// GDA_E_UNSUPPORTED outside that envelope.
// row slabs of the weight gradient: >= 64 rows each, at most 512 (at citation size -- 9,360 rows -- 146 workgroups instead
// of the 36 that 256-row slabs would leave on 256 CUs)
static int64_t skinny_slabs(int64_t rows) { return min((int64_t)512, max((int64_t)1, rows / 64)); }

extern "C" size_t gda_gemm_skinny_workspace_bytes(int mode, int64_t M, int64_t N, int64_t K) {
    if (mode != GDA_GEMM_TN || M <= 0 || M > SK_N || K <= 0) return 0;
    return (size_t)skinny_slabs(K) * (size_t)M * (size_t)(N + 1) * sizeof(float);
}

extern "C" int gda_gemm_skinny_f32(int mode, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                                   const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, float* colsum,
                                   void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    if (M <= 0 || N <= 0 || K <= 0 || ldc < N) return GDA_E_SIZE;
    if (!A || !B || !C) return GDA_E_NULL;
    if (C == A || C == B) return GDA_E_ALIAS;
    hipStream_t stream = (hipStream_t)stream_;
    auto wide_ok = [](int64_t w) { return w == 32 || w == 64 || w == 128 || w == 256; };
    auto al = [](const float* p, int64_t ld) { return ld % 4 == 0 && ((uintptr_t)p & 15) == 0; };
    if (mode == GDA_GEMM_NT) {
        if (N > SK_N || !wide_ok(K) || colsum || !al(A, lda) || lda < K || ldb < K) return GDA_E_UNSUPPORTED;
        const int64_t blocks = min(gda_cdiv(M, TB / 8), (int64_t)256 * 16);
        k_skinny_fwd<<<(unsigned)blocks, TB, 0, stream>>>(A, lda, B, ldb, bias, C, ldc, M, (int)N, (int)K);
    } else if (mode == GDA_GEMM_NN) {
        if (K > SK_N || !wide_ok(N) || bias || colsum || !al(B, ldb) || !al(C, ldc) || lda < K || ldb < N) return GDA_E_UNSUPPORTED;
        const int64_t rpb = TB / (N / 4);
        const int64_t blocks = min(gda_cdiv(M, rpb), (int64_t)256 * 16);
        k_skinny_dgrad<<<(unsigned)blocks, TB, 0, stream>>>(A, lda, B, ldb, C, ldc, M, (int)K, (int)N);
    } else if (mode == GDA_GEMM_TN) {
        if (M > SK_N || !wide_ok(N) || bias || !al(B, ldb) || lda < M || ldb < N) return GDA_E_UNSUPPORTED;
        const int64_t slabs = skinny_slabs(K);
        const int64_t rows = gda_cdiv(K, slabs);
        if (!workspace || workspace_bytes < gda_gemm_skinny_workspace_bytes(mode, M, N, K)) return GDA_E_WORKSPACE;
        float* part = (float*)workspace;
        float* cs_part = part + (size_t)slabs * M * N;
        k_skinny_wgrad<<<(unsigned)slabs, TB, 0, stream>>>(A, lda, B, ldb, part, colsum ? cs_part : nullptr, K, rows, (int)M, (int)N);
        GDA_LAUNCH_CHECK();
        GDA_UNLESS_SKIPPED("k_slab_sum") k_slab_sum<<<(unsigned)gda_cdiv(M * N + (colsum ? M : 0), SS_OUT), TB, 0, stream>>>(part, (int)slabs, M, N, C, ldc, cs_part, colsum);
    } else {
        return GDA_E_UNSUPPORTED;
    }
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

// Data gradient with the upstream activation's backward in its epilogue:
//   C[M, N] = mask(y) * (A[M, K] B[K, N]) / (1 - p),  mask = (y > 0),  y [M, N] (ld = ldm) = the output of dropout(relu(.))
// -- gda_relu_dropout_bwd_f32(gda_gemm NN(A, B), y) in one launch, same values.  Two envelopes: the tall split-fp16 kernel
// (N, K in {128, 256}; 16-byte aligned operands) and the skinny classifier kernel (K <= 8, N in {32, 64, 128, 256});
// GDA_E_UNSUPPORTED elsewhere (the caller composes).
extern "C" int gda_gemm_nn_mask_f32(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                                    float* C, int64_t ldc, const float* y, int64_t ldm, float p, gda_stream_t stream_) {
    if (M <= 0 || N <= 0 || K <= 0 || ldc < N || ldm < N || lda < K || ldb < N) return GDA_E_SIZE;
    if (!A || !B || !C || !y) return GDA_E_NULL;
    if (C == A || C == B || C == y) return GDA_E_ALIAS;
    if (!(p >= 0.f && p < 1.f)) return GDA_E_SIZE;
    if (ldc % 4 || ldm % 4 || ldb % 4 || (((uintptr_t)C | (uintptr_t)y | (uintptr_t)B) & 15)) return GDA_E_UNSUPPORTED;
    hipStream_t stream = (hipStream_t)stream_;
    const float mscale = 1.f / (1.f - p);
    if (K <= SK_N && (N == 32 || N == 64 || N == 128 || N == 256)) {
        const int64_t rpb = TB / (N / 4);
        const int64_t blocks = min(gda_cdiv(M, rpb), (int64_t)256 * 16);
        k_skinny_dgrad<<<(unsigned)blocks, TB, 0, stream>>>(A, lda, B, ldb, C, ldc, M, (int)K, (int)N, y, ldm, mscale);
        GDA_LAUNCH_CHECK();
        return GDA_OK;
    }
    static const bool split16 = [] { const char* e = std::getenv("PYGDA_AMD_GEMM_SPLIT_F16"); return !(e && e[0] == '0'); }();
    if (!split16 || (N != 128 && N != 256) || (K != 128 && K != 256) || lda % 4 || ((uintptr_t)A & 15)) return GDA_E_UNSUPPORTED;
    const int64_t bm = K == 128 ? 128 : 64;
    const int64_t nt = gda_cdiv(M, bm);
    const dim3 g((unsigned)min(nt, (int64_t)256), (unsigned)(N / 128));
    const size_t img = (size_t)bm * (K + 8) * 2;
    const size_t lds = 4 * img + 2 * (size_t)bm * sizeof(float);
    if (K == 128) {
        GDA_LDS_ATTR_ONCE((k_tall_fwd_h<128, true>), 160 * 1024);
        k_tall_fwd_h<128, true><<<g, TALL_TB, lds, stream>>>(A, lda, B, ldb, C, ldc, M, nullptr, y, ldm, mscale);
    } else {
        GDA_LDS_ATTR_ONCE((k_tall_fwd_h<256, true>), 160 * 1024);
        k_tall_fwd_h<256, true><<<g, TALL_TB, lds, stream>>>(A, lda, B, ldb, C, ldc, M, nullptr, y, ldm, mscale);
    }
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

// ---- the sampled batch's first projection without its gather pass, and projections with the activation in the epilogue ----
// Forward (NT):  C = act(A[arow] B^T + bias)
//   arow (may be NULL): row i of the tall operand is A[arow[i]] (int64 [M], device) -- x[n_id] of a sampled batch read by
//     the operand fetch; A then has any number of rows, lda >= K.
//   act_mode 0: none.  1: C [M, N] = dropout_site0(relu(.)).  2: C [2M, N], row i = dropout_site0, row M + i = an independent
//     draw dropout_site1 of the same pre-activation (gda_relu_dropout_pair_fwd_f32's stacked pair).  Keep-bits exactly as
//     gda_relu_dropout_fwd_f32 / _pair_fwd_f32 would draw them on the [M, N] pre-activation; p <= 0: relu only.
// Envelope: N, K in {128, 256}, ldc == N, 16-byte aligned operands, the split-fp16 kernel (PYGDA_AMD_GEMM_SPLIT_F16 != 0);
// GDA_E_UNSUPPORTED elsewhere.
extern "C" int gda_gemm_tall_fwd_ex_f32(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const int64_t* arow,
                                        const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias,
                                        int act_mode, float p, uint64_t seed, const int64_t* step_dev, uint32_t site0,
                                        uint32_t site1, gda_stream_t stream_) {
    if (M <= 0 || N <= 0 || K <= 0 || lda < K || ldb < K) return GDA_E_SIZE;
    if (!A || !B || !C) return GDA_E_NULL;
    if (act_mode < 0 || act_mode > 2 || !(p < 1.f) || (act_mode && p > 0.f && !step_dev)) return GDA_E_SIZE;
    if (C == A || C == B) return GDA_E_ALIAS;
    static const bool split16 = [] { const char* e = std::getenv("PYGDA_AMD_GEMM_SPLIT_F16"); return !(e && e[0] == '0'); }();
    if (!split16 || (N != 128 && N != 256) || (K != 128 && K != 256) || ldc != N || lda % 4 || ldb % 4 ||
        (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15))
        return GDA_E_UNSUPPORTED;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t bm = K == 128 ? 128 : 64;
    const int64_t nt = gda_cdiv(M, bm);
    const dim3 g((unsigned)min(nt, (int64_t)256), (unsigned)(N / 128));
    const size_t img = (size_t)bm * (K + 8) * 2;
    const size_t lds = 4 * img + 2 * (size_t)bm * sizeof(float);
    const TallAct act{p > 0.f ? p : 0.f, p > 0.f ? 1.f / (1.f - p) : 1.f, seed, step_dev, site0, site1};
#define TFX(K_, ACT_)                                                                                              \
    do {                                                                                                           \
        GDA_LDS_ATTR_ONCE((k_tall_fwd_h<K_, false, 0, ACT_>), 160 * 1024);                                         \
        k_tall_fwd_h<K_, false, 0, ACT_><<<g, TALL_TB, lds, stream>>>(A, lda, B, ldb, C, ldc, M, bias, nullptr, 0, 1.f, arow, act); \
    } while (0)
    if (K == 128) { if (act_mode == 0) TFX(128, 0); else if (act_mode == 1) TFX(128, 1); else TFX(128, 2); }
    else          { if (act_mode == 0) TFX(256, 0); else if (act_mode == 1) TFX(256, 1); else TFX(256, 2); }
#undef TFX
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

// Weight gradient (TN) with the gathered operand:  C[128, N] = A[Krows, 128]^T X[xrow][Krows, N], colsum[128] = column sums of
// A -- gda_gemm_tall_f32's TN form reading x through the batch's node ids.  Same envelope, same workspace.
extern "C" int gda_gemm_tall_wgrad_gather_f32(int64_t N, int64_t Krows, const float* A, int64_t lda, const float* X, int64_t ldx,
                                              const int64_t* xrow, float* C, int64_t ldc, float* colsum,
                                              void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    if (Krows <= 0 || (N != 128 && N != 256) || lda < 128 || ldx < N || ldc < N) return GDA_E_UNSUPPORTED;
    if (!A || !X || !C || !xrow) return GDA_E_NULL;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t slabs = tall_wgrad_slabs(Krows, N);
    const int64_t rows = gda_cdiv(gda_cdiv(Krows, slabs), 8) * 8;
    if (!workspace || workspace_bytes < gda_gemm_tall_workspace_bytes(GDA_GEMM_TN, 128, N, Krows)) return GDA_E_WORKSPACE;
    float* part = (float*)workspace;
    float* cs_part = part + (size_t)slabs * 128 * N;
    if (lda % 4 || ldx % 4 || ((uintptr_t)A & 15) || ((uintptr_t)X & 15)) return GDA_E_UNSUPPORTED;
    int st = tall_wgrad_launch(N, A, lda, X, ldx, xrow, part, colsum ? cs_part : nullptr, Krows, rows, slabs, stream);
    if (st != GDA_OK) return st;
    k_slab_sum<<<(unsigned)gda_cdiv(128 * N + (colsum ? 128 : 0), SS_OUT), TB, 0, stream>>>(part, (int)slabs, 128, N, C, ldc, cs_part, colsum);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}
