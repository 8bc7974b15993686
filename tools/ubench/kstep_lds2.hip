// Probe v2 of the LDS-resident K-step aggregation: the graph is compiled ONCE into a per-lane
// register program (wave-transposed, so the per-launch load is coalesced): thread t owns a contiguous
// row range chosen so that it holds at most R entries; entry j of lane l of wave w sits at
// sched[(w * R + j) * 64 + l] = {col | flags, val}.  Inside the launch: column slab of x in LDS
// (ping-pong), all R LDS gathers issued first, then the sequential (bit-exact) row sums, flushes at
// row ends.  No global memory access inside the step loop.  I/O here is column-major [d, n]
// (contiguous per column); a tiled transpose kernel is timed next to it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr unsigned F_LAST = 0x80000000u, F_SKIP = 0x40000000u, COL_MASK = 0x3fffffffu;

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

template <int R, int TB>
__global__ void __launch_bounds__(TB)
k_kstep(const int2* __restrict__ sched, const int* __restrict__ rowstart, int n_rows, int K,
        const float* __restrict__ xT, long ldx, float* __restrict__ yT, long ldy, const float* __restrict__ bias) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* cur = lds;
    float* nxt = lds + n_rows;
    const int c = blockIdx.x;
    const int t = threadIdx.x;
    const float* xc = xT + (long)c * ldx;
    for (int i = t * 4; i < n_rows; i += TB * 4) {          // n_rows % 4 == 0 is arranged by the host (padding)
        *reinterpret_cast<float4*>(cur + i) = *reinterpret_cast<const float4*>(xc + i);
    }
    unsigned ecol[R];
    float ew[R];
    const int2* sp = sched + (size_t)(t / 64) * R * 64 + (t % 64);
#pragma unroll
    for (int j = 0; j < R; ++j) { const int2 e = sp[j * 64]; ecol[j] = (unsigned)e.x; ew[j] = __int_as_float(e.y); }
    const int row0 = rowstart[t];
    const float bv = bias ? bias[c] : 0.f;
    __syncthreads();
    for (int step = 0; step < K; ++step) {
        const bool last = step == K - 1;
        float xv[R];
#pragma unroll
        for (int j = 0; j < R; ++j) xv[j] = cur[ecol[j] & COL_MASK];
        float acc = 0.f;
        int orow = row0;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const float nacc = __fadd_rn(acc, __fmul_rn(ew[j], xv[j]));
            acc = (ecol[j] & F_SKIP) ? acc : nacc;
            if (ecol[j] & F_LAST) {
                nxt[orow] = last ? __fadd_rn(acc, bv) : acc;
                ++orow;
                acc = 0.f;
            }
        }
        __syncthreads();
        float* tmp = cur; cur = nxt; nxt = tmp;
    }
    float* yc = yT + (long)c * ldy;
    for (int i = t * 4; i < n_rows; i += TB * 4) *reinterpret_cast<float4*>(yc + i) = *reinterpret_cast<const float4*>(cur + i);
}

// [rows, cols] -> [cols, rows], 64x64 tiles through LDS
__global__ void __launch_bounds__(256)
k_transpose(const float* __restrict__ in, long ldi, float* __restrict__ out, long ldo, int rows, int cols) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x % 64, ty = threadIdx.x / 64;
    for (int r = ty; r < 64; r += 4) if (r0 + r < rows && c0 + tx < cols) tile[r][tx] = in[(long)(r0 + r) * ldi + c0 + tx];
    __syncthreads();
    for (int cc = ty; cc < 64; cc += 4) if (c0 + cc < cols && r0 + tx < rows) out[(long)(c0 + cc) * ldo + r0 + tx] = tile[tx][cc];
}

static float time_us(hipEvent_t e0, hipEvent_t e1, int iters) { float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms * 1000.f / iters; }

template <int R, int TB>
int run(const std::vector<int>& rp, const std::vector<int>& ci, const std::vector<float>& va, int n, int npad, int d, int K,
        const float* xT, float* yT, const float* bias, const std::vector<float>& want) {
    // ---- compile the graph: contiguous row ranges with <= R entries per thread
    std::vector<int> rowstart(TB + 1, n);
    std::vector<int2> sched((size_t)TB * R, int2{(int)F_SKIP, 0});
    int row = 0;
    bool ok = true;
    for (int t = 0; t < TB; ++t) {
        rowstart[t] = row;
        int used = 0;
        const int target = (int)(((long)rp[n] + n / 8) * (t + 1) / TB);      // even spread, never beyond R
        while (row < n) {
            const int len = std::max(rp[row + 1] - rp[row], 1);              // an empty row costs one (skip|last) slot
            if (used + len > R) break;
            if (used > 0 && rp[row] + len > target && t + 1 < TB) break;
            const int w = t / 64, l = t % 64;
            if (rp[row + 1] == rp[row]) sched[((size_t)w * R + used) * 64 + l] = int2{(int)(F_SKIP | F_LAST), 0};
            for (int k = rp[row]; k < rp[row + 1]; ++k) {
                unsigned e = (unsigned)ci[k] | (k + 1 == rp[row + 1] ? F_LAST : 0u);
                int vb; memcpy(&vb, &va[k], 4);
                sched[((size_t)w * R + used + (k - rp[row])) * 64 + l] = int2{(int)e, vb};
            }
            used += len; ++row;
        }
    }
    if (row < n) ok = false;
    rowstart[TB] = n;
    if (!ok) { printf("R=%d TB=%d: graph does not fit the register program\n", R, TB); return 0; }
    int2* dsched; int* drs;
    CK(hipMalloc(&dsched, sched.size() * 8)); CK(hipMalloc(&drs, (TB + 1) * 4));
    CK(hipMemcpy(dsched, sched.data(), sched.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(drs, rowstart.data(), (TB + 1) * 4, hipMemcpyHostToDevice));
    const size_t lds = (size_t)2 * npad * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_kstep<R, TB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int Kx : {K, 0, 1, 30}) {
        for (int w = 0; w < 5; ++w) k_kstep<R, TB><<<d, TB, lds>>>(dsched, drs, npad, Kx, xT, npad, yT, npad, bias);
        CK(hipGetLastError());
        const int iters = 200;
        CK(hipEventRecord(e0));
        for (int it = 0; it < iters; ++it) k_kstep<R, TB><<<d, TB, lds>>>(dsched, drs, npad, Kx, xT, npad, yT, npad, bias);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        const float us = time_us(e0, e1, iters);
        size_t bad = 0;
        if (Kx == K) {
            std::vector<float> got((size_t)npad * d);
            CK(hipMemcpy(got.data(), yT, got.size() * 4, hipMemcpyDeviceToHost));
            for (int c = 0; c < d; ++c) for (int i = 0; i < n; ++i) if (memcmp(&got[(size_t)c * npad + i], &want[(size_t)i * d + c], 4) != 0) ++bad;
        }
        printf("kstep R=%d TB=%d K=%2d : %8.2f us/launch  mismatches %zu\n", R, TB, Kx, us, bad);
    }
    return 0;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 5484;
    const int und = argc > 2 ? atoi(argv[2]) : 8117;
    const int d = 128, K = 10;
    const int npad = (n + 3) / 4 * 4;
    std::mt19937 g(1);
    std::vector<std::vector<int>> adj(n);
    for (int e = 0; e < und; ++e) { int a = g() % n, b = g() % n; if (a == b) continue; adj[a].push_back(b); adj[b].push_back(a); }
    for (int i = 0; i < n; ++i) if (i % 997 != 5) adj[i].push_back(i);          // a few rows without a self loop; row 5 etc. may be empty
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
    int empties = 0; for (int i = 0; i < n; ++i) empties += rp[i + 1] == rp[i];
    printf("n %d nnz %d d %d K %d empty rows %d\n", n, nnz, d, K, empties);
    int *drp, *dci; float *dva, *x, *xT, *yT, *y, *y2, *dbias;
    CK(hipMalloc(&drp, (n + 1) * 4)); CK(hipMalloc(&dci, nnz * 4)); CK(hipMalloc(&dva, nnz * 4));
    CK(hipMalloc(&x, (size_t)n * d * 4)); CK(hipMalloc(&y, (size_t)n * d * 4)); CK(hipMalloc(&y2, (size_t)n * d * 4));
    CK(hipMalloc(&xT, (size_t)npad * d * 4)); CK(hipMalloc(&yT, (size_t)npad * d * 4)); CK(hipMalloc(&dbias, d * 4));
    CK(hipMemcpy(drp, rp.data(), (n + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dci, ci.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dva, va.data(), nnz * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, hb.data(), d * 4, hipMemcpyHostToDevice));
    CK(hipMemset(xT, 0, (size_t)npad * d * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    {   // baseline chain
        const unsigned grid = (unsigned)((n + 7) / 8);
        const int iters = 200;
        for (int w = 0; w < 5; ++w) { k_chain<256, 32><<<grid, 256>>>(drp, dci, dva, n, x, y); }
        CK(hipEventRecord(e0));
        for (int it = 0; it < iters; ++it) {
            const float* in = x;
            for (int s = 0; s < K; ++s) { float* out = (s & 1) ? y2 : y; k_chain<256, 32><<<grid, 256>>>(drp, dci, dva, n, in, out); in = out; }
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        printf("chain of K launches: %.2f us (%.2f per step)\n", time_us(e0, e1, iters), time_us(e0, e1, iters) / K);
    }
    {   // transposes
        dim3 grid((n + 63) / 64, d / 64);
        const int iters = 200;
        for (int w = 0; w < 5; ++w) k_transpose<<<grid, 256>>>(x, d, xT, npad, n, d);
        CK(hipEventRecord(e0));
        for (int it = 0; it < iters; ++it) k_transpose<<<grid, 256>>>(x, d, xT, npad, n, d);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        printf("transpose [n,128] -> [128,n]: %.2f us\n", time_us(e0, e1, iters));
        dim3 grid2(d / 64, (n + 63) / 64);
        CK(hipEventRecord(e0));
        for (int it = 0; it < iters; ++it) k_transpose<<<grid2, 256>>>(xT, npad, y, d, d, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        printf("transpose [128,n] -> [n,128]: %.2f us\n", time_us(e0, e1, iters));
        k_transpose<<<grid, 256>>>(x, d, xT, npad, n, d);
    }
    run<32, 1024>(rp, ci, va, n, npad, d, K, xT, yT, dbias, a);
    run<40, 1024>(rp, ci, va, n, npad, d, K, xT, yT, dbias, a);
    run<48, 1024>(rp, ci, va, n, npad, d, K, xT, yT, dbias, a);
    run<64, 512>(rp, ci, va, n, npad, d, K, xT, yT, dbias, a);
    run<24, 1024>(rp, ci, va, n, npad, d, K, xT, yT, dbias, a);
    return 0;
}
