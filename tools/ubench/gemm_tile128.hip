// Stand-alone probe (NOT part of libgda_hip.so): the tall projection  C[M, N] = A[M, K] * B[N, K]^T  (x W^T, fp32) on
// 128 x 128 macro-tiles with 2 x 2 MFMA tiles per wave -- the shape class where the library's 64 x 64-tile kernel
// (csrc/gda_gemm.hip: one 32x32 accumulator per wave) loses to the BLAS by 20-35 % (75 k - 300 k rows, profiles/HISTORY.md 4.6) and
// nn/linear.py hands the product to F.linear.  Written at the end of round 2 and run ONCE with the last GPU seconds
// (profiles/r2_gemm_tile128_ubench.txt): self-checks pass; 150,000 x 256 -> 128 in 122.4 us = 80.3 TF = 51 % of the fp32
// MFMA peak, 150,000 x 128 -> 128 in 73.7 us (42 %), 9,360 x 128 -> 128 in 20.3 us -- the BLAS does the first shape in
// 90-95 us (68 %), so this structure alone is NOT the win either; untuned (no wave-level software pipelining of the
// operand fetches, one barrier per 32-deep chunk, 2 waves per SIMD).  To rebuild and run:
//
//     hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/gemm_tile128.hip -o /tmp/gemm_tile128 && /tmp/gemm_tile128
//
// It checks itself against a float64 host product on sampled entries, then times M = 150,000 with (K, N) =
// (256, 128) and (128, 128) -- the cfg-S hidden projections -- and prints TFLOP/s next to the 157.3 TF fp32 MFMA peak
// (the BLAS measured 68 % of it at 150 k x 256 x 128, tools/gemm_bench.py).
//
// Design.  Workgroup = 4 waves, tile 128 (rows) x 128 (cols); wave w owns the 64 x 64 quadrant (w >> 1, w & 1) as
// 2 x 2 v_mfma_f32_32x32x2_f32 tiles (4 x 16 accumulator registers): per K pair two A fetches and two B fetches feed
// four MFMAs (the 64 x 64-tile kernel: two fetches per MFMA).  K runs in chunks of 32: a chunk of A (128 x 32) and of B
// (128 x 32) is read as 16-byte pieces along K (128-byte row segments: full cache lines), transposed into k-major LDS
// images S[k][row] (row stride 129: the operand fetch `S[k][32 consecutive rows]` is a conflict-free ds_read_b32 for
// both lane groups, and so are the transposing stores), double-buffered so that a chunk costs ONE barrier; the next chunk's global loads are issued
// before the current chunk's 64 MFMAs per wave and stored to the other LDS image after them.
// LDS: 2 buffers x 2 operands x 32 x 129 x 4 B = 66 KB -> two workgroups per CU.
#include <hip/hip_runtime.h>

#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s failed: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

namespace {

constexpr int TB = 256;
constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LD = BM + 1;                    // row stride of the k-major LDS images: odd, so that the transposing stores
                                              // (8 k-pieces x 4 rows per lane group -> banks 4 kp + row) are conflict-free
                                              // too; the operand fetches read 32 consecutive words of one k row either way

using f32x16 = __attribute__((ext_vector_type(16))) float;

// 16-byte piece (row r, columns c .. c+3) of a row-major [rows, K] matrix; out-of-range rows / columns read as zero
// through clamped addresses (branch-free: the loads of a chunk are issued back to back)
__device__ __forceinline__ float4 piece(const float* __restrict__ p, int64_t ld, int64_t r, int64_t c, int64_t rows, int64_t K) {
    const int64_t rr = r < rows ? r : rows - 1, cc = c + 4 <= K ? c : K - 4;
    float4 v = *reinterpret_cast<const float4*>(p + rr * ld + cc);
    const bool in = r < rows && c + 4 <= K;
    v.x = in ? v.x : 0.f; v.y = in ? v.y : 0.f; v.z = in ? v.z : 0.f; v.w = in ? v.w : 0.f;
    return v;
}

// K % 4 == 0, 16-byte aligned rows (lda, ldb % 4 == 0)
__global__ void __launch_bounds__(TB)
k_gemm_nt_128(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb, float* __restrict__ C,
              int64_t ldc, int64_t M, int64_t N, int64_t K, const float* __restrict__ bias) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float (*As)[BK][LD] = reinterpret_cast<float (*)[BK][LD]>(lds);                       // As[buf][k][row]
    float (*Bs)[BK][LD] = reinterpret_cast<float (*)[BK][LD]>(lds + 2 * BK * LD);         // Bs[buf][k][col]
    const int64_t i0 = (int64_t)blockIdx.x * BM, j0 = (int64_t)blockIdx.y * BN;
    const int tid = threadIdx.x, wave = tid / 64, lane = tid % 64;
    const int wi = (wave >> 1) * 64, wj = (wave & 1) * 64;
    const int ka = lane >> 5, la = lane & 31;
    // staging role: row = tid / 8 (+ 32 per piece), k piece = (tid % 8) * 4 -- 8 lanes cover one 128-byte row segment
    const int sr = tid / 8, sk = (tid % 8) * 4;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float4 va[4], vb[4];
    auto fetch = [&](int64_t k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            va[q] = piece(A, lda, i0 + sr + 32 * q, k0 + sk, M, K);
            vb[q] = piece(B, ldb, j0 + sr + 32 * q, k0 + sk, N, K);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = sr + 32 * q;
            As[buf][sk + 0][r] = va[q].x; As[buf][sk + 1][r] = va[q].y; As[buf][sk + 2][r] = va[q].z; As[buf][sk + 3][r] = va[q].w;
            Bs[buf][sk + 0][r] = vb[q].x; Bs[buf][sk + 1][r] = vb[q].y; Bs[buf][sk + 2][r] = vb[q].z; Bs[buf][sk + 3][r] = vb[q].w;
        }
    };
    fetch(0);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int64_t k0 = 0; k0 < K; k0 += BK) {
        const bool more = k0 + BK < K;
        if (more) fetch(k0 + BK);                         // in flight under the MFMAs below
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a0 = As[buf][kk + ka][wi + la], a1 = As[buf][kk + ka][wi + 32 + la];
            const float b0 = Bs[buf][kk + ka][wj + la], b1 = Bs[buf][kk + ka][wj + 32 + la];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) stash(buf ^ 1);                          // nobody reads that image before the barrier
        __syncthreads();
        buf ^= 1;
    }
    // C/D layout of the 32x32 MFMA: register r of lane l holds row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int64_t j = j0 + wj + 32 * b + la;
            const float bv = (bias && j < N) ? bias[j] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t i = i0 + wi + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * ka;
                if (i < M && j < N) C[i * ldc + j] = acc[a][b][r] + bv;
            }
        }
}

void launch(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t K, const float* bias, hipStream_t s) {
    const size_t lds = sizeof(float) * 4 * BK * LD;
    static bool configured = false;
    if (!configured) {
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_nt_128), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    const dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((N + BN - 1) / BN));
    k_gemm_nt_128<<<grid, TB, lds, s>>>(A, K, B, K, C, N, M, N, K, bias);
}

double check(int64_t M, int64_t N, int64_t K) {
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K), hb((size_t)N), hC((size_t)M * N);
    unsigned st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (float)((st >> 8) & 0xFFFF) / 65536.f - 0.5f; };
    for (auto& v : hA) v = rnd();
    for (auto& v : hB) v = rnd();
    for (auto& v : hb) v = rnd();
    float *dA, *dB, *db, *dC;
    HIP_OK(hipMalloc(&dA, hA.size() * 4)); HIP_OK(hipMalloc(&dB, hB.size() * 4)); HIP_OK(hipMalloc(&db, hb.size() * 4));
    HIP_OK(hipMalloc(&dC, hC.size() * 4));
    HIP_OK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    launch(dA, dB, dC, M, N, K, db, nullptr);
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (int64_t t = 0; t < 4000; ++t) {                       // sampled entries incl. the ragged last tile
        const int64_t i = t < 2000 ? (int64_t)(((uint64_t)t * 2654435761u) % (uint64_t)M) : M - 1 - (t % 200);
        const int64_t j = (int64_t)(((uint64_t)t * 40503u) % (uint64_t)N);
        double ref = hb[j];
        for (int64_t k = 0; k < K; ++k) ref += (double)hA[i * K + k] * (double)hB[j * K + k];
        worst = std::fmax(worst, std::fabs(ref - (double)hC[i * N + j]));
    }
    HIP_OK(hipFree(dA)); HIP_OK(hipFree(dB)); HIP_OK(hipFree(db)); HIP_OK(hipFree(dC));
    return worst;
}

void time_it(int64_t M, int64_t N, int64_t K) {
    float *dA, *dB, *dC;
    HIP_OK(hipMalloc(&dA, (size_t)M * K * 4)); HIP_OK(hipMalloc(&dB, (size_t)N * K * 4)); HIP_OK(hipMalloc(&dC, (size_t)M * N * 4));
    HIP_OK(hipMemset(dA, 0, (size_t)M * K * 4)); HIP_OK(hipMemset(dB, 0, (size_t)N * K * 4));
    for (int w = 0; w < 5; ++w) launch(dA, dB, dC, M, N, K, nullptr, nullptr);
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    const int iters = 50;
    HIP_OK(hipEventRecord(e0, nullptr));
    for (int w = 0; w < iters; ++w) launch(dA, dB, dC, M, N, K, nullptr, nullptr);
    HIP_OK(hipEventRecord(e1, nullptr));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, tf = 2.0 * (double)M * (double)N * (double)K / (us * 1e-6) / 1e12;
    std::printf("{\"kernel\": \"gemm_nt_128x128\", \"M\": %lld, \"N\": %lld, \"K\": %lld, \"us\": %.2f, \"TFLOPs\": %.2f, \"frac_of_157.3\": %.3f}\n",
                (long long)M, (long long)N, (long long)K, us, tf, tf / 157.3);
    HIP_OK(hipFree(dA)); HIP_OK(hipFree(dB)); HIP_OK(hipFree(dC));
}

}  // namespace

int main() {
    for (auto s : {std::array<int64_t, 3>{1000, 128, 128}, std::array<int64_t, 3>{777, 72, 100}, std::array<int64_t, 3>{4099, 256, 256}}) {
        const double err = check(s[0], s[1], s[2]);
        std::printf("check M=%lld N=%lld K=%lld: max |err| = %.3g %s\n", (long long)s[0], (long long)s[1], (long long)s[2], err,
                    err < 1e-4 ? "ok" : "FAILED");
        if (!(err < 1e-4)) return 1;
    }
    time_it(150000, 128, 256);
    time_it(150000, 128, 128);
    time_it(9360, 128, 128);
    return 0;
}
