// Stand-alone probe for the LDS-resident K-step aggregation (citation-graph regime).
//
// Baseline: K dependent launches of the lane-group CSR kernel (what gda_spmm_csr_kstep_f32 does).
// Candidate: ONE launch; a workgroup owns CS feature columns of ALL rows, keeps that column slab in
// LDS (ping-pong), keeps its threads' (col, val) lists in REGISTERS, and runs all K steps with
// workgroup barriers only -- the K-step product is independent per feature column, so no
// inter-workgroup exchange exists.  Sequential per-row sums in CSR order (bit-exact vs the chain).
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o kstep_lds_ubench kstep_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// ------------------------------------------------------------------ baseline chain kernel
template <int TB, int G, int UNROLL>
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

// ------------------------------------------------------------------ LDS-resident K-step
// Thread t owns the contiguous rows [t*rpt, (t+1)*rpt) and holds their first R entries in registers:
// ecol[j] = LDS float index of the neighbour (col * CS) | (1 << 31 if last entry of its row).
// Entries beyond R are read from global memory every step (rare; sized by the host).
template <int CS> struct Vec;
template <> struct Vec<1> { using T = float; };
template <> struct Vec<2> { using T = float2; };
template <> struct Vec<4> { using T = float4; };

template <int CS, int R, int TB>
__global__ void __launch_bounds__(TB)
k_kstep_lds(const int* __restrict__ rowptr, const int* __restrict__ colidx, const float* __restrict__ val,
            int n_rows, int rpt, int K, const float* __restrict__ x, long ldx, float* __restrict__ y, long ldy,
            const float* __restrict__ bias, int io_mode) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* cur = lds;
    float* nxt = lds + (size_t)n_rows * CS;
    const int c0 = blockIdx.x * CS;
    const int t = threadIdx.x;
    const int row0 = min(t * rpt, n_rows), row1 = min(row0 + rpt, n_rows);
    // ---- entry: column slab -> LDS
    if (io_mode == 0) {
        for (int i = t; i < n_rows; i += TB) {
#pragma unroll
            for (int v = 0; v < CS; ++v) cur[i * CS + v] = x[(long)i * ldx + c0 + v];
        }
    } else {        // x given column-major [d, n]: contiguous per column
        for (int i = t; i < n_rows; i += TB) {
#pragma unroll
            for (int v = 0; v < CS; ++v) cur[i * CS + v] = x[(long)(c0 + v) * ldx + i];
        }
    }
    // ---- this thread's entries -> registers
    const int e0 = row0 < n_rows ? rowptr[row0] : 0;
    const int e1 = row0 < n_rows ? rowptr[row1] : 0;
    const int cnt = e1 - e0;
    int ecol[R];
    float ew[R];
    {
        int r = row0;
        int rend = row0 < n_rows ? rowptr[row0 + 1] : 0;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            ecol[j] = 0; ew[j] = 0.f;
            if (j < cnt) {
                const int k = e0 + j;
                while (k >= rend) { ++r; rend = rowptr[r + 1]; }      // skips empty rows (they are zero-filled below)
                ecol[j] = (colidx[k] * CS) | ((k + 1 == rend) ? 0x80000000 : 0);
                ew[j] = val[k];
            }
        }
    }
    float bv[CS];
#pragma unroll
    for (int v = 0; v < CS; ++v) bv[v] = bias ? bias[c0 + v] : 0.f;
    __syncthreads();
    for (int step = 0; step < K; ++step) {
        const bool last = step == K - 1;
        // rows of this thread: walk the register list, flush at row ends
        int orow = row0;
        int rend = row0 < n_rows ? rowptr[row0 + 1] - e0 : 0;      // end of the current row, relative
        float acc[CS];
#pragma unroll
        for (int v = 0; v < CS; ++v) acc[v] = 0.f;
        // empty rows in front / between: handled by the generic flush below (rare): zero result
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (j < cnt) {
                const int a = ecol[j] & 0x7fffffff;
                typename Vec<CS>::T xv = *reinterpret_cast<const typename Vec<CS>::T*>(cur + a);
                const float* xp = reinterpret_cast<const float*>(&xv);
#pragma unroll
                for (int v = 0; v < CS; ++v) acc[v] = __fadd_rn(acc[v], __fmul_rn(ew[j], xp[v]));
                if (ecol[j] < 0) {
                    while (rend <= j) {          // empty rows before this one (rend == start of a later row)
#pragma unroll
                        for (int v = 0; v < CS; ++v) nxt[orow * CS + v] = last ? bv[v] : 0.f;
                        ++orow; rend = rowptr[orow + 1] - e0;
                    }
#pragma unroll
                    for (int v = 0; v < CS; ++v) nxt[orow * CS + v] = last ? __fadd_rn(acc[v], bv[v]) : acc[v];
#pragma unroll
                    for (int v = 0; v < CS; ++v) acc[v] = 0.f;
                    ++orow;
                    rend = orow < row1 ? rowptr[orow + 1] - e0 : 0x7fffffff;
                }
            }
        }
        // overflow entries (beyond the register list) and trailing empty rows: generic path
        if (cnt > R || orow < row1) {
            const int consumed = min(cnt, R);
            int k = e0 + consumed;
            for (; orow < row1; ++orow) {
                const int re = rowptr[orow + 1];
                if (re <= e0 + consumed) {                 // an empty row inside the register range
#pragma unroll
                    for (int v = 0; v < CS; ++v) nxt[orow * CS + v] = last ? bv[v] : 0.f;
                    continue;
                }
                for (; k < re; ++k) {
                    const int a = colidx[k] * CS; const float w = val[k];
#pragma unroll
                    for (int v = 0; v < CS; ++v) acc[v] = __fadd_rn(acc[v], __fmul_rn(w, cur[a + v]));
                }
#pragma unroll
                for (int v = 0; v < CS; ++v) { nxt[orow * CS + v] = last ? __fadd_rn(acc[v], bv[v]) : acc[v]; acc[v] = 0.f; }
            }
        }
        __syncthreads();
        float* tmp = cur; cur = nxt; nxt = tmp;
    }
    // ---- exit
    if (io_mode == 0) {
        for (int i = t; i < n_rows; i += TB) {
#pragma unroll
            for (int v = 0; v < CS; ++v) y[(long)i * ldy + c0 + v] = cur[i * CS + v];
        }
    } else {
        for (int i = t; i < n_rows; i += TB) {
#pragma unroll
            for (int v = 0; v < CS; ++v) y[(long)(c0 + v) * ldy + i] = cur[i * CS + v];
        }
    }
}

static float time_us(hipEvent_t e0, hipEvent_t e1, int iters) {
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms * 1000.f / iters;
}

template <int CS, int R, int TB>
int run_lds(const char* tag, const int* drp, const int* dci, const float* dva, int n, int d, int K, const float* x,
            float* y, const float* bias, int io_mode, const std::vector<float>& want, int iters) {
    const int rpt = (n + TB - 1) / TB;
    const size_t lds = (size_t)2 * n * CS * sizeof(float);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_kstep_lds<CS, R, TB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long ld = io_mode == 0 ? d : n;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 5; ++w) k_kstep_lds<CS, R, TB><<<d / CS, TB, lds>>>(drp, dci, dva, n, rpt, K, x, ld, y, ld, bias, io_mode);
    CK(hipGetLastError());
    CK(hipEventRecord(e0));
    for (int it = 0; it < iters; ++it) k_kstep_lds<CS, R, TB><<<d / CS, TB, lds>>>(drp, dci, dva, n, rpt, K, x, ld, y, ld, bias, io_mode);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    const float us = time_us(e0, e1, iters);
    std::vector<float> got((size_t)n * d);
    CK(hipMemcpy(got.data(), y, got.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    if (!want.empty()) {
        for (size_t i = 0; i < got.size(); ++i) {
            const size_t r = io_mode == 0 ? i / d : i % n, c = io_mode == 0 ? i % d : i / n;
            if (memcmp(&got[i], &want[r * d + c], 4) != 0) ++bad;
        }
    }
    printf("%-34s K=%2d  %8.2f us/launch  (%.2f us/step)  lds %zu KB  mismatches %zu\n", tag, K, us, K ? us / K : 0.f, lds / 1024, bad);
    return 0;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 5484;
    const int und = argc > 2 ? atoi(argv[2]) : 8117;
    const int d = 128, K = 10;
    std::mt19937 g(1);
    std::vector<std::vector<int>> adj(n);
    for (int e = 0; e < und; ++e) { int a = g() % n, b = g() % n; if (a == b) continue; adj[a].push_back(b); adj[b].push_back(a); }
    for (int i = 0; i < n; ++i) adj[i].push_back(i);
    std::vector<int> rp(n + 1, 0), ci; std::vector<float> va;
    std::uniform_real_distribution<float> U(0.1f, 0.5f);
    for (int i = 0; i < n; ++i) { rp[i + 1] = rp[i] + (int)adj[i].size(); for (int c : adj[i]) { ci.push_back(c); va.push_back(U(g)); } }
    const int nnz = (int)ci.size();
    std::vector<float> hx((size_t)n * d), hb(d);
    std::normal_distribution<float> Nrm(0.f, 1.f);
    for (auto& v : hx) v = Nrm(g);
    for (auto& v : hb) v = Nrm(g);
    // CPU reference (same sequential order, separately rounded)
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
    int *drp, *dci; float *dva, *x, *xt, *y, *y2, *dbias;
    CK(hipMalloc(&drp, (n + 1) * 4)); CK(hipMalloc(&dci, nnz * 4)); CK(hipMalloc(&dva, nnz * 4));
    CK(hipMalloc(&x, (size_t)n * d * 4)); CK(hipMalloc(&xt, (size_t)n * d * 4)); CK(hipMalloc(&y, (size_t)n * d * 4)); CK(hipMalloc(&y2, (size_t)n * d * 4));
    CK(hipMalloc(&dbias, d * 4));
    CK(hipMemcpy(drp, rp.data(), (n + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dci, ci.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dva, va.data(), nnz * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, hb.data(), d * 4, hipMemcpyHostToDevice));
    std::vector<float> hxt((size_t)n * d);
    for (int i = 0; i < n; ++i) for (int c = 0; c < d; ++c) hxt[(size_t)c * n + i] = hx[(size_t)i * d + c];
    CK(hipMemcpy(xt, hxt.data(), hxt.size() * 4, hipMemcpyHostToDevice));
    printf("n %d nnz %d d %d K %d  max row %d\n", n, nnz, d, K, (int)std::max_element(adj.begin(), adj.end(), [](auto& p, auto& q) { return p.size() < q.size(); })->size());

    // baseline chain
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const unsigned grid = (unsigned)((n + 7) / 8);
        const int iters = 200;
        for (int w = 0; w < 5; ++w) { k_chain<256, 32, 4><<<grid, 256>>>(drp, dci, dva, n, x, y); k_chain<256, 32, 4><<<grid, 256>>>(drp, dci, dva, n, y, y2); }
        CK(hipEventRecord(e0));
        for (int it = 0; it < iters; ++it) {
            const float* in = x;
            for (int s = 0; s < K; ++s) { float* out = (s & 1) ? y2 : y; k_chain<256, 32, 4><<<grid, 256>>>(drp, dci, dva, n, in, out); in = out; }
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        const float us = time_us(e0, e1, iters);
        printf("%-34s K=%2d  %8.2f us/chain   (%.2f us/step)\n", "chain of K launches (G=32)", K, us, us / K);
    }
    const std::vector<float> none;
    run_lds<1, 32, 1024>("lds cs=1 R=32 tb=1024 rowmajor", drp, dci, dva, n, d, K, x, y, dbias, 0, a, 100);
    run_lds<1, 32, 1024>("lds cs=1 R=32 tb=1024 K=0 (I/O)", drp, dci, dva, n, d, 0, x, y, dbias, 0, none, 100);
    run_lds<1, 32, 1024>("lds cs=1 R=32 tb=1024 colmajor", drp, dci, dva, n, d, K, xt, y, dbias, 1, a, 100);
    run_lds<1, 32, 1024>("lds cs=1 colmajor K=0 (I/O)", drp, dci, dva, n, d, 0, xt, y, dbias, 1, none, 100);
    run_lds<2, 32, 1024>("lds cs=2 R=32 tb=1024 rowmajor", drp, dci, dva, n, d, K, x, y, dbias, 0, a, 100);
    run_lds<2, 32, 1024>("lds cs=2 K=0 (I/O)", drp, dci, dva, n, d, 0, x, y, dbias, 0, none, 100);
    run_lds<4, 32, 1024>("lds cs=4 R=32 tb=1024 rowmajor", drp, dci, dva, n, d, K, x, y, dbias, 0, a, 100);
    run_lds<4, 32, 1024>("lds cs=4 K=0 (I/O)", drp, dci, dva, n, d, 0, x, y, dbias, 0, none, 100);
    run_lds<1, 48, 1024>("lds cs=1 R=48 tb=1024 rowmajor", drp, dci, dva, n, d, K, x, y, dbias, 0, a, 100);
    run_lds<1, 64, 512>("lds cs=1 R=64 tb=512  rowmajor", drp, dci, dva, n, d, K, x, y, dbias, 0, a, 100);
    run_lds<2, 64, 512>("lds cs=2 R=64 tb=512  rowmajor", drp, dci, dva, n, d, K, x, y, dbias, 0, a, 100);
    run_lds<1, 32, 1024>("lds cs=1 R=32 K=1", drp, dci, dva, n, d, 1, x, y, dbias, 0, none, 100);
    run_lds<1, 32, 1024>("lds cs=1 R=32 K=30", drp, dci, dva, n, d, 30, x, y, dbias, 0, none, 100);
    return 0;
}
