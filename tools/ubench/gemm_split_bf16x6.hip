// EXPERIMENT, NOT PART OF libgda_hip.so (round 2).  fp32-accurate tall GEMMs on the bf16 matrix cores by an exact
// three-way bf16 split of both operands (six bf16 MFMAs per term).  Correct (error at the level of the BLAS's fp32
// result against float64; bit-exact on bf16-representable integer inputs; NT / NN / TN) but NOT faster than the BLAS's
// fp32 MFMA kernels on MI355X, in either of two structures:
//   v1  every wave loads, splits, multiplies and stores; 2 workgroups per CU:   150k x 256 x 128 forward 109-123 us
//   v2  persistent, wave-specialised (4 MFMA waves + 4 staging waves, double-buffered LDS, per-wave chunk ownership,
//       16-byte stores through transposed accumulators):                         120-135 us          (BLAS: 90-95 us)
// Bisecting v2 by switching phases off (GDA_SPLIT_DBG): launch + weight split 19 us, MFMA waves alone +31, staging
// waves alone (global loads + split + LDS stores) +65, tile stores +19 -- the phases add up instead of overlapping,
// and the staging side alone already costs what the BLAS needs for the whole product: splitting 4096 fp32 values
// into 3 x bf16 and pushing 48 KB through the LDS store path (~80 B/clk/CU) per 32-deep chunk is ~1-2 us against
// 0.64 us of MFMA time.  The 6/16 matrix-core advantage is eaten by the conversion; pre-splitting the tall operand in
// HBM would cost more traffic than the product itself.  Kept for the record; the product path uses the BLAS above
// 50 k rows (pygda_amd/nn/linear.py) and the 64x64 fp32 MFMA kernels (csrc/gda_gemm.hip) below.
// Build stand-alone: hipcc --offload-arch=gfx950 -O3 -I pygda_amd/csrc -shared -fPIC tools/ubench/gemm_split_bf16x6.hip
//
// fp32 GEMM on the bf16 matrix cores for TALL operands (rows = 10^5 nodes of a sampled sub-graph), gfx950.
//
// The fp32-input MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 MFMA rate on CDNA4, and a 150 k x 256 x 128
// projection is matrix-core bound on it (9.8 GFLOP: 63 us at the fp32 MFMA peak against 29 us of HBM time; the BLAS
// reaches 68 % of that peak).  Every fp32 number is EXACTLY the sum of three bf16 numbers (8 + 8 + 8 significand
// bits: x = x0 + x1 + x2 by two truncate-and-subtract steps, no rounding anywhere), so
//     a * b = sum_{i+j<=2} a_i b_j  +  (a1 b2 + a2 b1 + a2 b2),      the bracket <= 3 * 2^-24 |a b|,
// and the six kept products are exact in the MFMA's fp32 accumulator.  Six bf16 MFMAs (32x32x16: 8x the k-depth of
// the fp32 instruction in half its cycles) cost 6/16 of the fp32 MFMA time -- the product keeps fp32 accuracy (error
// of the order of the fp32 accumulation's own rounding; tests compare against float64) and the kernel lands on the
// HBM side of the roofline.
//
//   macro-tile 128 x 128 per workgroup (4 waves, each a 64 x 64 quadrant = 2 x 2 MFMA tiles), k chunks of 32;
//   A (the tall fp32 operand, k contiguous) is split on its way into LDS; B (the small weight) is split ONCE by
//   gda_gemm_split_prepare into three bf16 planes [3][N][K] (transposing if the caller's B is [K, N]: that is the
//   data-gradient product) and copied into LDS as it is.  LDS rows are 80 bytes (32 bf16 + pad): the 16-byte
//   operand reads of 16 consecutive lanes cover all 64 banks once.
//   TN (the weight gradient, reduction over the rows): both operands are tall and k-strided; a thread loads 8
//   consecutive rows x 4 columns, transposes in registers and writes 16-byte k-runs; the row range is cut into slabs
//   whose partial tiles are summed in a fixed order (deterministic split-K).
#include "../../pygda_amd/csrc/gda_common.h"
#include <stdlib.h>

namespace {

constexpr int TB = 256;
constexpr int BM = 128, BN = 128, KC = 32;
constexpr int LDR = 40;                  // bf16 per LDS row: 32 + 8 pad (80 bytes)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;      // plain vector: HIP's uint4 (a struct of unions) kept an
                                                                  // array of staged runs in scratch memory

// x = h + m + l exactly, each a bf16 (kept in the high halves of three fp32 patterns)
__device__ __forceinline__ void split3(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(h);
    m = __float_as_uint(r1) & 0xffff0000u;
    l = __float_as_uint(r1 - __uint_as_float(m));
}
__device__ __forceinline__ uint32_t pack2(uint32_t lo_elem, uint32_t hi_elem) { return (lo_elem >> 16) | (hi_elem & 0xffff0000u); }

// split four consecutive-k values and store them as 8 bytes per plane at S[plane][row][k]
__device__ __forceinline__ void store_split4(unsigned short (*S)[BM][LDR], int row, int k, const float4 v) {
    uint32_t h[4], m[4], l[4];
    split3(v.x, h[0], m[0], l[0]); split3(v.y, h[1], m[1], l[1]);
    split3(v.z, h[2], m[2], l[2]); split3(v.w, h[3], m[3], l[3]);
    *reinterpret_cast<uint2*>(&S[0][row][k]) = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
    *reinterpret_cast<uint2*>(&S[1][row][k]) = make_uint2(pack2(m[0], m[1]), pack2(m[2], m[3]));
    *reinterpret_cast<uint2*>(&S[2][row][k]) = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
}

// the 24 MFMAs of one 16-deep k-step on a wave's 2 x 2 tiles.  The B fragment is handed to the MFMA as its ROW operand:
// the accumulator tile is the transposed product, D[i = output column][j = output row], so that a lane ends up with
// four CONSECUTIVE COLUMNS of one output row per register group -- 16-byte stores (a quarter of the store
// instructions of the natural orientation, which took 35 us of a 120 us kernel).
__device__ __forceinline__ void kstep16(const unsigned short (*As)[BM][LDR], const unsigned short (*Bs)[BN][LDR], int wr,
                                        int wc, int lane, int k0, f32x16 (&acc)[2][2]) {
    const int r = lane & 31, kk = k0 + (lane >> 5) * 8;
    bf16x8 a[2][3], b[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            a[t][p] = *reinterpret_cast<const bf16x8*>(&As[p][wr + 32 * t + r][kk]);
            b[t][p] = *reinterpret_cast<const bf16x8*>(&Bs[p][wc + 32 * t + r][kk]);
        }
    // products outermost (smallest terms first), the four tiles innermost: consecutive MFMAs never share an accumulator
#define GDA_SPLIT_PRODUCT(PA, PB)                                                                          \
    _Pragma("unroll") for (int ti = 0; ti < 2; ++ti)                                                        \
    _Pragma("unroll") for (int tj = 0; tj < 2; ++tj)                                                        \
        acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[tj][PB], a[ti][PA], acc[ti][tj], 0, 0, 0);
    GDA_SPLIT_PRODUCT(2, 0)
    GDA_SPLIT_PRODUCT(1, 1)
    GDA_SPLIT_PRODUCT(0, 2)
    GDA_SPLIT_PRODUCT(1, 0)
    GDA_SPLIT_PRODUCT(0, 1)
    GDA_SPLIT_PRODUCT(0, 0)
#undef GDA_SPLIT_PRODUCT
}

// C tile store.  acc[ti][tj] = D[i][j] with j = lane & 31 the output row (inside tile ti) and
// i = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) the output column (inside tile tj): four consecutive columns per group
__device__ __forceinline__ void store_tile(const f32x16 (&acc)[2][2], float* __restrict__ C, int64_t ldc, int64_t M,
                                           int64_t N, int64_t m0, int64_t n0, int lane, const float* __restrict__ bias) {
    const bool vec = (ldc % 4 == 0) && ((uintptr_t)C % 16 == 0);
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        const int64_t row = m0 + 32 * ti + (lane & 31);
        if (row >= M) continue;
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t col = n0 + 32 * tj + 8 * g + 4 * (lane >> 5);
                if (col >= N) continue;
                float v[4] = {acc[ti][tj][4 * g], acc[ti][tj][4 * g + 1], acc[ti][tj][4 * g + 2], acc[ti][tj][4 * g + 3]};
                if (bias) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (col + e < N) v[e] += bias[col + e];
                }
                float* out = C + row * ldc + col;
                if (vec && col + 3 < N) *reinterpret_cast<float4*>(out) = make_float4(v[0], v[1], v[2], v[3]);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (col + e < N) out[e] = v[e];
                }
            }
    }
}

// B [N, K] fp32 (ldb) or, transposed, B [K, N] -> planes [3][Npad][K] bf16, Npad = round_up(N, 128), zero padded
__global__ void __launch_bounds__(TB)
k_split_prepare(const float* __restrict__ B, int64_t ldb, int transposed, int64_t N, int64_t K, int64_t Npad,
                unsigned short* __restrict__ planes) {
    const int64_t total = Npad * K;
    for (int64_t e = (int64_t)blockIdx.x * TB + threadIdx.x; e < total; e += (int64_t)gridDim.x * TB) {
        const int64_t n = e / K, k = e % K;
        const float v = n < N ? (transposed ? B[k * ldb + n] : B[n * ldb + k]) : 0.f;
        uint32_t h, m, l;
        split3(v, h, m, l);
        planes[e] = (unsigned short)(h >> 16);
        planes[total + e] = (unsigned short)(m >> 16);
        planes[2 * total + e] = (unsigned short)(l >> 16);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Both kernels are PERSISTENT and WAVE-SPECIALISED: a workgroup is 8 waves -- waves 0-3 only read LDS operand
// fragments and issue MFMAs (and store a finished tile), waves 4-7 only move data (global loads two chunks ahead in a
// two-set register ring, the bf16 split, LDS stores) -- over a double-buffered LDS image, one barrier per 32-deep
// chunk, and the chunk counter runs on across the tiles a workgroup owns, so the loads of the next tile are in flight
// while the last chunks of this one are multiplied and its result is stored.  (The first version gave every wave both
// jobs, two workgroups per CU: load, split, multiply and store phases simply added up -- 123 us at 150 k x 256 x 128,
// of which the MFMAs were 42.)
// ---------------------------------------------------------------------------------------------------------------
constexpr int WG = 512;

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// C[M, N] = A[M, K] . Bplanes[N, K]^T (+ bias): A fp32 row-major (k contiguous), K % 32 == 0
__global__ void __launch_bounds__(WG)
k_gemm_split_nt(const float* __restrict__ A, int64_t lda, const unsigned short* __restrict__ planes, int64_t Npad,
                int64_t M, int64_t N, int64_t K, float* __restrict__ C, int64_t ldc, const float* __restrict__ bias,
                int tiles_n, int64_t ntiles, int dbg) {
    __shared__ __attribute__((aligned(16))) unsigned short As[2][3][BM][LDR];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][3][BN][LDR];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nchunks = (int)(K / KC);
    const int64_t my_tiles = (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const int64_t total = my_tiles * nchunks;
    const int64_t plane_sz = Npad * K;

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        // Waves 4,5 stage the even chunks, waves 6,7 the odd ones (half a chunk each): a wave's loads belong to ONE
        // chunk, issued two iterations before it stores them, so the wait for them (per-wave vmcnt) never covers
        // younger loads -- with one ring shared by all producer waves every store waited for the newest loads too.
        const int pw = wave - 4, parity = pw >> 1, half = pw & 1;
        const int ak = (lane & 7) * 4, ar = half * 64 + (lane >> 3);     // A half chunk: 64 rows x 8 float4 -> 8 per lane
        float4 av[8];
        u32x4 bv[12];                                                  // B half chunk: 768 runs of 16 bytes -> 12 per lane
        auto fetch = [&](int64_t j) {
            if (j >= total || (dbg & 2)) return;
            const int64_t t = blockIdx.x + (j / nchunks) * gridDim.x;
            const int64_t m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN, k0 = (j % nchunks) * (int64_t)KC;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int64_t row = m0 + ar + 8 * q;
                av[q] = row < M ? *reinterpret_cast<const float4*>(A + row * lda + k0 + ak) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                const int u = half * 768 + lane + 64 * q;      // plane = u / 512, row = (u % 512) / 4, run = u % 4
                const int p = u >> 9, row = (u & 511) >> 2, run = u & 3;
                bv[q] = *reinterpret_cast<const u32x4*>(planes + p * plane_sz + (n0 + row) * K + k0 + run * 8);
            }
        };
        auto stash = [&](int64_t j, int buf) {
            if (j >= total || (dbg & 4)) return;
#pragma unroll
            for (int q = 0; q < 8; ++q) store_split4(As[buf], ar + 8 * q, ak, av[q]);
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                const int u = half * 768 + lane + 64 * q;
                const int p = u >> 9, row = (u & 511) >> 2, run = u & 3;
                *reinterpret_cast<u32x4*>(&Bs[buf][p][row][run * 8]) = bv[q];
            }
        };
        if (parity == 0) { fetch(0); stash(0, 0); fetch(2); }
        else fetch(1);
        __syncthreads();
        for (int64_t g = 0; g < total; ++g) {
            if (((g + 1) & 1) == parity) {                     // this pair's turn: chunk g+1 into the buffer the MFMA waves
                stash(g + 1, (int)((g + 1) & 1));              // are not reading, then its next chunk's loads
                fetch(g + 3);
            }
            __syncthreads();
        }
        return;
    }
    // ---------------------------------------------------------------------- matrix-core waves
    const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;
    f32x16 acc[2][2];
    zero_acc(acc);
    auto finish = [&](int64_t g) {                             // last chunk of a tile: write it out, start the next one
        if ((g + 1) % nchunks != 0) return;
        const int64_t t = blockIdx.x + (g / nchunks) * gridDim.x;
        if (!(dbg & 8)) store_tile(acc, C, ldc, M, N, (t / tiles_n) * BM + wr, (t % tiles_n) * BN + wc, lane, bias);
        zero_acc(acc);
    };
    __syncthreads();
    for (int64_t g = 0; g < total; ++g) {
        const int buf = (int)(g & 1);
        if (!(dbg & 1)) {
            kstep16(As[buf], Bs[buf], wr, wc, lane, 0, acc);
            kstep16(As[buf], Bs[buf], wr, wc, lane, 16, acc);
        }
        finish(g);
        __syncthreads();
    }
}

// partial[z][M, N] = sum over the rows of slab z of A[row, m] * B[row, n]   (A [R, M], B [R, N] fp32 row-major).
// A job = (output tile, row slab); every slab has `cps` chunks (rows past R read as zeros).
__global__ void __launch_bounds__(WG)
k_gemm_split_tn(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb, int64_t R,
                int64_t M, int64_t N, int cps, int tiles_n, int64_t ntiles, int64_t njobs, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) unsigned short As[2][3][BM][LDR];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][3][BN][LDR];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t my_jobs = (njobs - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const int64_t total = my_jobs * cps;

    if (wave >= 4) {
        // A unit of staging = 4 columns (c4) x one run of 8 consecutive rows (the k direction); a chunk is 32 rows x 128
        // columns = 128 units per operand.  Waves 4,5 stage the even chunks, waves 6,7 the odd ones; a lane of the
        // pair (0..127) owns unit `ul` of A and unit `ul` of B.
        const int pw = wave - 4, parity = pw >> 1, ul = (pw & 1) * 64 + lane;
        const int c4 = (ul & 31) * 4, run = ul >> 5;
        float4 va[8], vb[8];
        auto fetch1 = [&](const float* __restrict__ P, int64_t ld, int64_t c, int64_t lim, int64_t k0, float4 (&v)[8]) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t row = k0 + run * 8 + i;
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < R) {
                    const float* p = P + row * ld + c;
                    if (c + 3 < lim) x = *reinterpret_cast<const float4*>(p);
                    else { if (c < lim) x.x = p[0]; if (c + 1 < lim) x.y = p[1]; if (c + 2 < lim) x.z = p[2]; }
                }
                v[i] = x;
            }
        };
        auto fetch = [&](int64_t j) {
            if (j >= total) return;
            const int64_t job = blockIdx.x + (j / cps) * gridDim.x;
            const int64_t t = job % ntiles, z = job / ntiles;
            const int64_t k0 = (z * cps + j % cps) * KC;
            fetch1(A, lda, (t / tiles_n) * BM + c4, M, k0, va);
            fetch1(B, ldb, (t % tiles_n) * BN + c4, N, k0, vb);
        };
        auto stash1 = [&](unsigned short (*S)[BM][LDR], const float4 (&v)[8]) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {                      // column c4 + e: its 8 consecutive k as one 16-byte run per plane
                uint32_t h[8], m[8], l[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float x = e == 0 ? v[i].x : e == 1 ? v[i].y : e == 2 ? v[i].z : v[i].w;
                    split3(x, h[i], m[i], l[i]);
                }
                u32x4 w0 = {pack2(h[0], h[1]), pack2(h[2], h[3]), pack2(h[4], h[5]), pack2(h[6], h[7])};
                u32x4 w1 = {pack2(m[0], m[1]), pack2(m[2], m[3]), pack2(m[4], m[5]), pack2(m[6], m[7])};
                u32x4 w2 = {pack2(l[0], l[1]), pack2(l[2], l[3]), pack2(l[4], l[5]), pack2(l[6], l[7])};
                *reinterpret_cast<u32x4*>(&S[0][c4 + e][run * 8]) = w0;
                *reinterpret_cast<u32x4*>(&S[1][c4 + e][run * 8]) = w1;
                *reinterpret_cast<u32x4*>(&S[2][c4 + e][run * 8]) = w2;
            }
        };
        auto stash = [&](int64_t j, int buf) {
            if (j >= total) return;
            stash1(As[buf], va);
            stash1(Bs[buf], vb);
        };
        if (parity == 0) { fetch(0); stash(0, 0); fetch(2); }
        else fetch(1);
        __syncthreads();
        for (int64_t g = 0; g < total; ++g) {
            if (((g + 1) & 1) == parity) {
                stash(g + 1, (int)((g + 1) & 1));
                fetch(g + 3);
            }
            __syncthreads();
        }
        return;
    }
    const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;
    f32x16 acc[2][2];
    zero_acc(acc);
    auto finish = [&](int64_t g) {
        if ((g + 1) % cps != 0) return;
        const int64_t job = blockIdx.x + (g / cps) * gridDim.x;
        const int64_t t = job % ntiles, z = job / ntiles;
        store_tile(acc, partial + z * M * N, N, M, N, (t / tiles_n) * BM + wr, (t % tiles_n) * BN + wc, lane, nullptr);
        zero_acc(acc);
    };
    __syncthreads();
    for (int64_t g = 0; g < total; ++g) {
        const int buf = (int)(g & 1);
        kstep16(As[buf], Bs[buf], wr, wc, lane, 0, acc);
        kstep16(As[buf], Bs[buf], wr, wc, lane, 16, acc);
        finish(g);
        __syncthreads();
    }
}

// C[m, n] = sum_z partial[z][m][n]: 64 outputs per workgroup, slabs over the four waves, fixed order
__global__ void __launch_bounds__(TB)
k_split_slab_sum(const float* __restrict__ partial, int slabs, int64_t MN, int64_t N, float* __restrict__ C, int64_t ldc) {
    __shared__ float part[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t idx = (int64_t)blockIdx.x * 64 + lane;
    float a = 0.f;
    if (idx < MN) {
        int z = wave;
        for (; z + 12 < slabs; z += 16) {
            const float v0 = partial[(int64_t)z * MN + idx], v1 = partial[(int64_t)(z + 4) * MN + idx];
            const float v2 = partial[(int64_t)(z + 8) * MN + idx], v3 = partial[(int64_t)(z + 12) * MN + idx];
            a += v0; a += v1; a += v2; a += v3;
        }
        for (; z < slabs; z += 4) a += partial[(int64_t)z * MN + idx];
    }
    part[wave][lane] = a;
    __syncthreads();
    if (wave == 0 && idx < MN) C[(idx / N) * ldc + idx % N] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

constexpr int PERSISTENT_WGS = 256;      // one 8-wave workgroup per CU (122 KB of LDS each)

// slabs of the row range for the weight gradient: ~PERSISTENT_WGS jobs in all, at least eight chunks per slab
int tn_slabs(int64_t R, int64_t M, int64_t N) {
    const int64_t tiles = gda_cdiv(M, BM) * gda_cdiv(N, BN);
    int64_t s = gda_cdiv(PERSISTENT_WGS, tiles);
    const int64_t most = gda_cdiv(R, 8 * KC);
    if (s > most) s = most;
    return (int)(s < 1 ? 1 : s);
}

}  // namespace

extern "C" size_t gda_gemm_split_workspace_bytes(int mode, int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    if (mode == GDA_GEMM_TN) return (size_t)tn_slabs(K, M, N) * (size_t)M * (size_t)N * sizeof(float);
    const int64_t npad = gda_cdiv(N, BN) * BN;              // NT / NN: the three bf16 planes of the small operand
    return (size_t)3 * (size_t)npad * (size_t)K * sizeof(unsigned short);
}

// mode NT: C[M, N] = A[M, K] . B[N, K]^T (+ bias)     B = the weight as stored, [out, in]
// mode NN: C[M, N] = A[M, K] . B[K, N]                 B = the weight as stored (data gradient: K = out, N = in)
// mode TN: C[M, N] = A[K, M]^T . B[K, N]               reduction over the K rows of two tall operands (weight gradient)
extern "C" int gda_gemm_split_f32(int mode, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                                  int64_t ldb, float* C, int64_t ldc, const float* bias, void* workspace,
                                  size_t workspace_bytes, gda_stream_t stream_) {
    if (M < 0 || N < 0 || K < 0 || ldc < N) return GDA_E_SIZE;
    if (mode != GDA_GEMM_NT && mode != GDA_GEMM_NN && mode != GDA_GEMM_TN) return GDA_E_UNSUPPORTED;
    if (M == 0 || N == 0) return GDA_OK;
    if (K == 0) return GDA_E_UNSUPPORTED;
    if (!A || !B || !C || !workspace) return GDA_E_NULL;
    if (workspace_bytes < gda_gemm_split_workspace_bytes(mode, M, N, K) || (uintptr_t)workspace % 16 != 0) return GDA_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t tiles_m = gda_cdiv(M, BM), tiles_n = gda_cdiv(N, BN), ntiles = tiles_m * tiles_n;
    if (tiles_n > INT32_MAX || ntiles > INT32_MAX) return GDA_E_SIZE;
    if (mode == GDA_GEMM_TN) {
        if (lda < M || ldb < N || bias) return bias ? GDA_E_UNSUPPORTED : GDA_E_SIZE;
        if (lda % 4 != 0 || ldb % 4 != 0 || ((uintptr_t)A | (uintptr_t)B) % 16 != 0) return GDA_E_UNSUPPORTED;
        const int slabs = tn_slabs(K, M, N);
        const int64_t cps = gda_cdiv(gda_cdiv(K, slabs), KC);            // chunks per slab
        if (cps > INT32_MAX) return GDA_E_SIZE;
        const int64_t used = gda_cdiv(K, cps * KC);                      // slabs that hold rows (<= slabs)
        const int64_t njobs = ntiles * used;
        float* partial = static_cast<float*>(workspace);
        const unsigned grid = (unsigned)(njobs < PERSISTENT_WGS ? njobs : PERSISTENT_WGS);
        k_gemm_split_tn<<<grid, WG, 0, stream>>>(A, lda, B, ldb, K, M, N, (int)cps, (int)tiles_n, ntiles, njobs, partial);
        GDA_LAUNCH_CHECK();
        k_split_slab_sum<<<(unsigned)gda_cdiv(M * N, 64), TB, 0, stream>>>(partial, (int)used, M * N, N, C, ldc);
        GDA_LAUNCH_CHECK();
        return GDA_OK;
    }
    const bool nn = mode == GDA_GEMM_NN;
    if (lda < K || ldb < (nn ? N : K)) return GDA_E_SIZE;
    if (K % KC != 0 || lda % 4 != 0 || (uintptr_t)A % 16 != 0) return GDA_E_UNSUPPORTED;
    const int64_t npad = tiles_n * BN;
    unsigned short* planes = static_cast<unsigned short*>(workspace);
    const int64_t total = npad * K;
    k_split_prepare<<<(unsigned)(gda_cdiv(total, TB) < 1024 ? gda_cdiv(total, TB) : 1024), TB, 0, stream>>>(
        B, ldb, nn ? 1 : 0, N, K, npad, planes);
    GDA_LAUNCH_CHECK();
    const unsigned grid = (unsigned)(ntiles < PERSISTENT_WGS ? ntiles : PERSISTENT_WGS);
    const char* e = getenv("GDA_SPLIT_DBG");
    k_gemm_split_nt<<<grid, WG, 0, stream>>>(A, lda, planes, npad, M, N, K, C, ldc, bias, (int)tiles_n, ntiles, e ? atoi(e) : 0);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}
