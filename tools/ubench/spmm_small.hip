// Stand-alone probe: per-launch time of the aggregation kernel at the citation size (N=5484,
// ~4 entries/row, d=128) in a back-to-back K-step chain, for different workgroup shapes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>

template <int TB, int G, int UNROLL>
__global__ void __launch_bounds__(TB)
k(const int* __restrict__ rowptr, const int* __restrict__ colidx, const float* __restrict__ val, long n_rows,
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
        int e = 0;
        for (; e + UNROLL <= cnt; e += UNROLL) {
            float4 xv[UNROLL]; float w[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int cu = __shfl(my_col, e + u, G); w[u] = __shfl(my_val, e + u, G);
                xv[u] = *reinterpret_cast<const float4*>(x + (long)cu * d + c);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { acc[0] += w[u] * xv[u].x; acc[1] += w[u] * xv[u].y; acc[2] += w[u] * xv[u].z; acc[3] += w[u] * xv[u].w; }
        }
        for (; e < cnt; ++e) {
            const int cu = __shfl(my_col, e, G); const float w = __shfl(my_val, e, G);
            const float4 xv = *reinterpret_cast<const float4*>(x + (long)cu * d + c);
            acc[0] += w * xv.x; acc[1] += w * xv.y; acc[2] += w * xv.z; acc[3] += w * xv.w;
        }
    }
    if (live) *reinterpret_cast<float4*>(y + row * d + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

template <int TB, int G, int UNROLL>
float run(const int* rp, const int* ci, const float* va, long n, float* a, float* b) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const unsigned grid = (unsigned)((n + TB / G - 1) / (TB / G));
    for (int w = 0; w < 10; ++w) { k<TB, G, UNROLL><<<grid, TB>>>(rp, ci, va, n, a, b); k<TB, G, UNROLL><<<grid, TB>>>(rp, ci, va, n, b, a); }
    (void)hipEventRecord(e0);
    for (int it = 0; it < 200; ++it) { k<TB, G, UNROLL><<<grid, TB>>>(rp, ci, va, n, a, b); k<TB, G, UNROLL><<<grid, TB>>>(rp, ci, va, n, b, a); }
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 400 * 1000;
}

int main() {
    const long n = 5484; std::mt19937 g(1);
    std::vector<std::vector<int>> adj(n);
    for (int e = 0; e < 8117; ++e) { int a = g() % n, b = g() % n; adj[a].push_back(b); adj[b].push_back(a); }
    for (long i = 0; i < n; ++i) adj[i].push_back((int)i);
    std::vector<int> rp(n + 1, 0), ci; std::vector<float> va;
    for (long i = 0; i < n; ++i) { rp[i + 1] = rp[i] + (int)adj[i].size(); for (int c : adj[i]) { ci.push_back(c); va.push_back(0.25f); } }
    int *drp, *dci; float *dva, *a, *b;
    (void)hipMalloc(&drp, (n + 1) * 4); (void)hipMalloc(&dci, ci.size() * 4); (void)hipMalloc(&dva, va.size() * 4);
    (void)hipMalloc(&a, n * 128 * 4); (void)hipMalloc(&b, n * 128 * 4);
    (void)hipMemcpy(drp, rp.data(), (n + 1) * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dci, ci.data(), ci.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dva, va.data(), va.size() * 4, hipMemcpyHostToDevice); (void)hipMemset(a, 0, n * 128 * 4);
    printf("nnz %zu\n", ci.size());
    printf("TB256 G32 U4: %.2f us\n", run<256, 32, 4>(drp, dci, dva, n, a, b));
    printf("TB128 G32 U4: %.2f us\n", run<128, 32, 4>(drp, dci, dva, n, a, b));
    printf("TB64  G32 U4: %.2f us\n", run<64, 32, 4>(drp, dci, dva, n, a, b));
    printf("TB512 G32 U4: %.2f us\n", run<512, 32, 4>(drp, dci, dva, n, a, b));
    printf("TB256 G32 U2: %.2f us\n", run<256, 32, 2>(drp, dci, dva, n, a, b));
    printf("TB256 G32 U8: %.2f us\n", run<256, 32, 8>(drp, dci, dva, n, a, b));
    printf("TB64  G32 U8: %.2f us\n", run<64, 32, 8>(drp, dci, dva, n, a, b));
    return 0;
}
