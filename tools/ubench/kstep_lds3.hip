// Perf/bit-exactness harness for pygda_amd/csrc/gda_kstep.hip (the product file is included as is).
#include "../../pygda_amd/csrc/gda_kstep.hip"
#include <cstdio>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define CG(x) do { int s_ = (x); if (s_ != 0) { printf("gda status %d at %d\n", s_, __LINE__); return 1; } } while (0)

template <int TB, int G>
__global__ void __launch_bounds__(TB)
k_chain(const int* __restrict__ rowptr, const int* __restrict__ colidx, const float* __restrict__ val, long n_rows,
        const float* __restrict__ x, float* __restrict__ y) {
    constexpr int VEC = 4, d = 128;
    const int lane = threadIdx.x % G;
    const long row = (long)blockIdx.x * (TB / G) + threadIdx.x / G;
    const bool live = row < n_rows;
    const int start = live ? rowptr[row] : 0, end = live ? rowptr[row + 1] : 0;
    const int c = lane * VEC;
    float acc[VEC] = {0, 0, 0, 0};
    for (int base = start; base < end; base += G) {
        const int kx = base + lane;
        const int my_col = kx < end ? colidx[kx] : 0;
        const float my_val = kx < end ? val[kx] : 0.f;
        const int cnt = min(G, end - base);
        for (int e = 0; e < cnt; ++e) {
            const int cu = __shfl(my_col, e, G); const float w = __shfl(my_val, e, G);
            const float4 xv = *reinterpret_cast<const float4*>(x + (long)cu * d + c);
            acc[0] = __fadd_rn(acc[0], __fmul_rn(w, xv.x)); acc[1] = __fadd_rn(acc[1], __fmul_rn(w, xv.y));
            acc[2] = __fadd_rn(acc[2], __fmul_rn(w, xv.z)); acc[3] = __fadd_rn(acc[3], __fmul_rn(w, xv.w));
        }
    }
    if (live) *reinterpret_cast<float4*>(y + row * d + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

static float time_us(hipEvent_t e0, hipEvent_t e1, int iters) { float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms * 1000.f / iters; }

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 5484;
    const int und = argc > 2 ? atoi(argv[2]) : 8117;
    const int d = 128, K = 10;
    const int npad = (n + 3) / 4 * 4;
    std::mt19937 g(1);
    std::vector<std::vector<int>> adj(n);
    for (int e = 0; e < und; ++e) { int a = g() % n, b = g() % n; if (a == b) continue; adj[a].push_back(b); adj[b].push_back(a); }
    for (int i = 0; i < n; ++i) if (i % 997 != 5) adj[i].push_back(i);
    adj[7].clear();                                                    // an empty row
    for (int q = 0; q < 19; ++q) adj[11].push_back((q * 37) % n);      // a longer row (several slots)
    std::vector<int> rp(n + 1, 0), ci; std::vector<float> va;
    std::uniform_real_distribution<float> U(0.1f, 0.5f);
    for (int i = 0; i < n; ++i) { rp[i + 1] = rp[i] + (int)adj[i].size(); for (int c : adj[i]) { ci.push_back(c); va.push_back(U(g)); } }
    const int nnz = (int)ci.size();
    std::vector<float> hx((size_t)n * d), hb(d);
    std::normal_distribution<float> Nrm(0.f, 1.f);
    for (auto& v : hx) v = Nrm(g);
    for (auto& v : hb) v = Nrm(g);
    std::vector<float> a = hx, b((size_t)n * d);
    for (int s = 0; s < K; ++s) {
        for (int i = 0; i < n; ++i)
            for (int c = 0; c < d; ++c) {
                volatile float acc = 0.f;
                for (int k = rp[i]; k < rp[i + 1]; ++k) { volatile float p = va[k] * a[(size_t)ci[k] * d + c]; acc = acc + p; }
                b[(size_t)i * d + c] = (s == K - 1) ? acc + hb[c] : acc;
            }
        a.swap(b);
    }
    std::vector<char> plan(gda_kstep_plan_bytes(12));
    const int S = gda_kstep_plan_host(rp.data(), ci.data(), va.data(), n, plan.data(), plan.size());
    printf("n %d nnz %d d %d K %d -> slots %d (R = %d)\n", n, nnz, d, K, S, S * 4);
    if (S <= 0) return 0;
    int *drp, *dci; float *dva, *x, *y, *y2, *dbias, *scratch; void* dplan;
    CK(hipMalloc(&drp, (n + 1) * 4)); CK(hipMalloc(&dci, nnz * 4)); CK(hipMalloc(&dva, nnz * 4));
    CK(hipMalloc(&x, (size_t)n * d * 4)); CK(hipMalloc(&y, (size_t)n * d * 4)); CK(hipMalloc(&y2, (size_t)n * d * 4));
    CK(hipMalloc(&scratch, (size_t)2 * npad * d * 4)); CK(hipMalloc(&dbias, d * 4)); CK(hipMalloc(&dplan, gda_kstep_plan_bytes(S)));
    CK(hipMemcpy(drp, rp.data(), (n + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dci, ci.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dva, va.data(), nnz * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, hb.data(), d * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dplan, plan.data(), gda_kstep_plan_bytes(S), hipMemcpyHostToDevice));
    CK(hipMemset(scratch, 0, (size_t)2 * npad * d * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 200;
    {
        const unsigned grid = (unsigned)((n + 7) / 8);
        for (int w = 0; w < 5; ++w) k_chain<256, 32><<<grid, 256>>>(drp, dci, dva, n, x, y);
        CK(hipEventRecord(e0));
        for (int it = 0; it < iters; ++it) {
            const float* in = x;
            for (int s = 0; s < K; ++s) { float* out = (s & 1) ? y2 : y; k_chain<256, 32><<<grid, 256>>>(drp, dci, dva, n, in, out); in = out; }
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        printf("chain of K launches      : %8.2f us (%.2f per step)\n", time_us(e0, e1, iters), time_us(e0, e1, iters) / K);
    }
    // row-major wrapper (transpose + kernel + transpose): correctness + time
    for (int w = 0; w < 5; ++w) CG(gda_kstep_lds_f32(dplan, S, n, d, K, x, d, 0, y, d, 0, dbias, nullptr, scratch, nullptr));
    CK(hipEventRecord(e0));
    for (int it = 0; it < iters; ++it) CG(gda_kstep_lds_f32(dplan, S, n, d, K, x, d, 0, y, d, 0, dbias, nullptr, scratch, nullptr));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    std::vector<float> got((size_t)n * d);
    CK(hipMemcpy(got.data(), y, got.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < got.size(); ++i) if (memcmp(&got[i], &a[i], 4) != 0) ++bad;
    printf("row-major wrapper K=%d    : %8.2f us   mismatches %zu\n", K, time_us(e0, e1, iters), bad);
    float* xT = scratch; float* yT = scratch + (size_t)npad * d;
    for (int Kx : {10, 0, 1, 2, 30}) {
        for (int w = 0; w < 3; ++w) CG(gda_kstep_lds_colmajor_f32(dplan, S, n, d, Kx, xT, npad, yT, npad, dbias, nullptr, nullptr));
        CK(hipEventRecord(e0));
        for (int it = 0; it < iters; ++it) CG(gda_kstep_lds_colmajor_f32(dplan, S, n, d, Kx, xT, npad, yT, npad, dbias, nullptr, nullptr));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        printf("column-major kernel K=%2d : %8.2f us\n", Kx, time_us(e0, e1, iters));
    }
    CK(hipEventRecord(e0));
    for (int it = 0; it < iters; ++it) CG(gda_transpose_f32(x, d, xT, npad, n, d, nullptr));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    printf("transpose in             : %8.2f us\n", time_us(e0, e1, iters));
    CK(hipEventRecord(e0));
    for (int it = 0; it < iters; ++it) CG(gda_transpose_f32(yT, npad, y, d, d, n, nullptr));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    printf("transpose out            : %8.2f us\n", time_us(e0, e1, iters));
    return 0;
}
