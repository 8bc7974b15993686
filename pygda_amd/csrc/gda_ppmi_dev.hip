// PPMI graph construction on the device (UDAGCN / AdaGCN(ppmi) / SpecReg start-up).
//
// Same estimator as the native host builder (gda_ppmi.cpp) and the reference's Python loop
// (pygda/nn/ppmi_conv.py:98-172), and the SAME counter-based walks: a walk is a pure function of
// (seed, pass, start node), so host and device builders produce identical visit counts.
//   1. symmetrised, de-duplicated adjacency: 64-bit keys (a << 32 | b), radix sort, unique
//   2. one thread per (start node, pass): walk of 1..path_len steps, every visited node emitted as
//      a key (start << 32 | visited); unused slots carry a sentinel that sorts last
//   3. radix sort of the visits; run heads = the distinct (start, visited) pairs, run lengths =
//      the counts; p = count / visits(start)                                        (:109-117,150)
//   4. pairs re-sorted by visited node; one thread per node adds its column in the host builder's
//      order (start ascending)                                                       (:152-155)
//   5. w = max(log(p / colsum * |targets| / path_len), 0)                            (:157-163)
// Output: (src, dst, w) sorted by (src, dst) + the pair count in device memory; all reductions are
// fixed-order, so a run is reproducible bit for bit.
#include "gda_common.h"

#include <hipcub/hipcub.hpp>

namespace {

constexpr int TB = 256;
constexpr uint64_t SENT = ~0ull;

struct Mix {
    uint64_t s;
    __device__ explicit Mix(uint64_t seed) : s(seed) {}
    __device__ uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    __device__ uint64_t below(uint64_t n) { return __umul64hi(next(), n); }
};

__device__ __forceinline__ int64_t lower_bound_u64(const uint64_t* __restrict__ a, int64_t n, uint64_t key) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ void k_sym_keys(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int64_t E,
                           uint64_t* __restrict__ keys) {
    const int64_t e = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (e >= E) return;
    const uint64_t a = (uint64_t)src[e], b = (uint64_t)dst[e];
    keys[2 * e] = (a << 32) | b;
    keys[2 * e + 1] = (b << 32) | a;
}

// head[i] = 1 for the first key of every run of equal non-sentinel keys
__global__ void k_heads(const uint64_t* __restrict__ keys, int64_t n, int32_t* __restrict__ head) {
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = keys[i];
    head[i] = (k != SENT && (i == 0 || keys[i - 1] != k)) ? 1 : 0;
}

// compact the run heads: out_key[pos] = key, out_at[pos] = index of the head (run start)
__global__ void k_compact(const uint64_t* __restrict__ keys, const int32_t* __restrict__ head,
                          const int32_t* __restrict__ pos, int64_t n, uint64_t* __restrict__ out_key,
                          int32_t* __restrict__ out_at) {
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n || !head[i]) return;
    out_key[pos[i]] = keys[i];
    if (out_at) out_at[pos[i]] = (int32_t)i;
}

__global__ void k_adj(const uint64_t* __restrict__ uniq, const int32_t* __restrict__ count, int64_t N,
                      int32_t* __restrict__ ptr, int32_t* __restrict__ adj) {
    const int64_t nu = *count;
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i <= N) ptr[i] = (int32_t)lower_bound_u64(uniq, nu, (uint64_t)i << 32);
    if (i < nu) adj[i] = (int32_t)(uniq[i] & 0xFFFFFFFFull);
}

__global__ void k_walk(const int32_t* __restrict__ ptr, const int32_t* __restrict__ adj, int64_t N, int path_len,
                       int passes, uint64_t seed, uint64_t* __restrict__ vis) {
    const int64_t idx = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (idx >= N * passes) return;
    const int64_t a = idx / passes;
    const int p = (int)(idx % passes);
    uint64_t* out = vis + idx * path_len;
    int steps = 0;
    if (ptr[a + 1] > ptr[a]) {
        Mix rng(seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(p + 1)) ^ (0x9E3779B97F4A7C15ull * (uint64_t)(a + 1)));
        steps = 1 + (int)rng.below((uint64_t)path_len);
        int64_t cur = a;
        for (int s = 0; s < steps; ++s) {
            const int64_t d = ptr[cur + 1] - ptr[cur];
            const int32_t b = adj[ptr[cur] + (int64_t)rng.below((uint64_t)d)];
            out[s] = ((uint64_t)a << 32) | (uint64_t)(uint32_t)b;
            cur = b;
        }
    }
    for (int s = steps; s < path_len; ++s) out[s] = SENT;
}

// p = run length / visits of the start node; second sort key (visited << 32 | pair index)
__global__ void k_prob(const uint64_t* __restrict__ vis_sorted, int64_t nv, const uint64_t* __restrict__ pair,
                       const int32_t* __restrict__ at, const int32_t* __restrict__ n_pairs_p,
                       double* __restrict__ prob, uint64_t* __restrict__ key2, int64_t cap) {
    const int64_t n_pairs = *n_pairs_p;
    const int64_t k = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (k >= cap) return;
    if (k >= n_pairs) { key2[k] = SENT; return; }
    const int64_t n_valid = lower_bound_u64(vis_sorted, nv, SENT);
    const int64_t run_end = k + 1 < n_pairs ? at[k + 1] : n_valid;
    const uint64_t a = pair[k] >> 32;
    const int64_t total = lower_bound_u64(vis_sorted, nv, (a + 1) << 32) - lower_bound_u64(vis_sorted, nv, a << 32);
    prob[k] = (double)(run_end - at[k]) / (double)total;
    key2[k] = ((pair[k] & 0xFFFFFFFFull) << 32) | (uint64_t)k;
}

__global__ void k_colsum(const uint64_t* __restrict__ key2_sorted, const int32_t* __restrict__ n_pairs_p,
                         const double* __restrict__ prob, int64_t N, double* __restrict__ colsum,
                         int32_t* __restrict__ n_targets) {
    const int64_t b = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (b >= N) return;
    const int64_t n_pairs = *n_pairs_p;
    const int64_t lo = lower_bound_u64(key2_sorted, n_pairs, (uint64_t)b << 32);
    const int64_t hi = lower_bound_u64(key2_sorted, n_pairs, (uint64_t)(b + 1) << 32);
    double s = 0.0;
    for (int64_t k = lo; k < hi; ++k) s += prob[key2_sorted[k] & 0xFFFFFFFFull];     // pair index ascending = start ascending
    colsum[b] = s;
    if (hi > lo) atomicAdd(n_targets, 1);                                           // integer count: order-free
}

__global__ void k_weights(const uint64_t* __restrict__ pair, const int32_t* __restrict__ n_pairs_p,
                          const double* __restrict__ prob, const double* __restrict__ colsum,
                          const int32_t* __restrict__ n_targets, int path_len, int64_t* __restrict__ out_src,
                          int64_t* __restrict__ out_dst, float* __restrict__ out_w, int64_t* __restrict__ out_count) {
    const int64_t n_pairs = *n_pairs_p;
    const int64_t k = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (k == 0) *out_count = n_pairs;
    if (k >= n_pairs) return;
    const int64_t a = (int64_t)(pair[k] >> 32), b = (int64_t)(pair[k] & 0xFFFFFFFFull);
    const double v = log(prob[k] / colsum[b] * (double)(*n_targets) / (double)path_len);
    out_src[k] = a;
    out_dst[k] = b;
    out_w[k] = (float)(v > 0.0 ? v : 0.0);
}

struct Ws {
    uint64_t *k0, *k1;      // [nk]   symmetrised keys (sort in / out)
    uint64_t* uniq;         // [nk]
    int32_t *adj, *ptr;     // [nk], [N + 1]
    uint64_t *v0, *v1;      // [nv]   visits (sort in / out)
    uint64_t* pair;         // [nv]   distinct (start, visited)
    int32_t* at;            // [nv]   run starts
    double* prob;           // [nv]
    uint64_t *q0, *q1;      // [nv]   (visited, pair index) keys
    int32_t *head, *pos;    // [max(nk, nv) + 1]
    double* colsum;         // [N]
    int32_t* scalars;       // [4]: n_uniq, n_pairs, n_targets
    void* cub; size_t cub_bytes;
    size_t total;
};

size_t cub_needed(int64_t n) {
    size_t a = 0, b = 0;
    uint64_t* k = nullptr;
    int32_t* p = nullptr;
    hipcub::DeviceRadixSort::SortKeys(nullptr, a, k, k, (int)n);
    hipcub::DeviceScan::ExclusiveSum(nullptr, b, p, p, (int)n + 1);
    return a > b ? a : b;
}

Ws carve(void* base, int64_t E, int64_t N, int path_len, int passes) {
    const int64_t nk = 2 * E > 0 ? 2 * E : 1, nv = N * passes * path_len > 0 ? N * passes * path_len : 1;
    const int64_t nm = nk > nv ? nk : nv;
    Ws w{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void* p = base ? (void*)((char*)base + off) : nullptr;
        off += gda_align_up(bytes, 256);
        return p;
    };
    w.k0 = (uint64_t*)take(8 * nk); w.k1 = (uint64_t*)take(8 * nk); w.uniq = (uint64_t*)take(8 * nk);
    w.adj = (int32_t*)take(4 * nk); w.ptr = (int32_t*)take(4 * (N + 2));
    w.v0 = (uint64_t*)take(8 * nv); w.v1 = (uint64_t*)take(8 * nv); w.pair = (uint64_t*)take(8 * nv);
    w.at = (int32_t*)take(4 * (nv + 1)); w.prob = (double*)take(8 * nv);
    w.q0 = (uint64_t*)take(8 * nv); w.q1 = (uint64_t*)take(8 * nv);
    w.head = (int32_t*)take(4 * (nm + 2)); w.pos = (int32_t*)take(4 * (nm + 2));
    w.colsum = (double*)take(8 * (N + 1));
    w.scalars = (int32_t*)take(64);
    w.cub_bytes = cub_needed(nm);
    w.cub = take(w.cub_bytes);
    w.total = off;
    return w;
}

unsigned blocks(int64_t n) { return (unsigned)gda_cdiv(n > 0 ? n : 1, TB); }

// heads of `keys[n]` -> compacted (out_key, out_at), number of heads in *count
int unique_runs(const uint64_t* keys, int64_t n, Ws& w, uint64_t* out_key, int32_t* out_at, int32_t* count,
                hipStream_t stream) {
    k_heads<<<blocks(n), TB, 0, stream>>>(keys, n, w.head);
    GDA_LAUNCH_CHECK();
    GDA_HIP_TRY(hipMemsetAsync(w.head + n, 0, sizeof(int32_t), stream));
    size_t cb = w.cub_bytes;
    GDA_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(w.cub, cb, w.head, w.pos, (int)n + 1, stream));
    GDA_HIP_TRY(hipMemcpyAsync(count, w.pos + n, sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
    k_compact<<<blocks(n), TB, 0, stream>>>(keys, w.head, w.pos, n, out_key, out_at);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

}  // namespace

extern "C" size_t gda_ppmi_workspace_bytes(int64_t E, int64_t N, int path_len, int passes) {
    if (E < 0 || N < 0 || path_len < 1 || passes < 1) return 0;
    if (2 * E >= INT32_MAX || N * passes * path_len >= INT32_MAX) return 0;
    return carve(nullptr, E, N, path_len, passes).total;
}

extern "C" int gda_ppmi_build(const int64_t* src, const int64_t* dst, int64_t E, int64_t N, int path_len,
                              int passes, uint64_t seed, int64_t* out_src, int64_t* out_dst, float* out_w,
                              int64_t* out_count, void* workspace, size_t workspace_bytes,
                              gda_stream_t stream_) {
    if (E < 0 || N < 0 || N >= INT32_MAX || path_len < 1 || passes < 1) return GDA_E_SIZE;
    if (2 * E >= INT32_MAX || N * passes * path_len >= INT32_MAX) return GDA_E_UNSUPPORTED;   // host builder
    if (!out_count || !workspace || (E > 0 && (!src || !dst))) return GDA_E_NULL;
    hipStream_t stream = (hipStream_t)stream_;
    if (E == 0 || N == 0) { GDA_HIP_TRY(hipMemsetAsync(out_count, 0, sizeof(int64_t), stream)); return GDA_OK; }
    if (!out_src || !out_dst || !out_w) return GDA_E_NULL;
    Ws w = carve(workspace, E, N, path_len, passes);
    if (workspace_bytes < w.total) return GDA_E_WORKSPACE;
    const int64_t nk = 2 * E, nv = N * passes * path_len;
    int32_t* n_uniq = w.scalars, *n_pairs = w.scalars + 1, *n_targets = w.scalars + 2;
    GDA_HIP_TRY(hipMemsetAsync(w.scalars, 0, 64, stream));
    // 1. adjacency
    k_sym_keys<<<blocks(E), TB, 0, stream>>>(src, dst, E, w.k0);
    GDA_LAUNCH_CHECK();
    size_t cb = w.cub_bytes;
    GDA_HIP_TRY(hipcub::DeviceRadixSort::SortKeys(w.cub, cb, w.k0, w.k1, (int)nk, 0, 64, stream));
    int st = unique_runs(w.k1, nk, w, w.uniq, nullptr, n_uniq, stream);
    if (st != GDA_OK) return st;
    k_adj<<<blocks(nk > N + 1 ? nk : N + 1), TB, 0, stream>>>(w.uniq, n_uniq, N, w.ptr, w.adj);
    GDA_LAUNCH_CHECK();
    // 2. walks
    k_walk<<<blocks(N * passes), TB, 0, stream>>>(w.ptr, w.adj, N, path_len, passes, seed, w.v0);
    GDA_LAUNCH_CHECK();
    // 3. counts
    cb = w.cub_bytes;
    GDA_HIP_TRY(hipcub::DeviceRadixSort::SortKeys(w.cub, cb, w.v0, w.v1, (int)nv, 0, 64, stream));
    st = unique_runs(w.v1, nv, w, w.pair, w.at, n_pairs, stream);
    if (st != GDA_OK) return st;
    k_prob<<<blocks(nv), TB, 0, stream>>>(w.v1, nv, w.pair, w.at, n_pairs, w.prob, w.q0, nv);
    GDA_LAUNCH_CHECK();
    // 4. column sums in a fixed order
    cb = w.cub_bytes;
    GDA_HIP_TRY(hipcub::DeviceRadixSort::SortKeys(w.cub, cb, w.q0, w.q1, (int)nv, 0, 64, stream));
    k_colsum<<<blocks(N), TB, 0, stream>>>(w.q1, n_pairs, w.prob, N, w.colsum, n_targets);
    GDA_LAUNCH_CHECK();
    // 5. weights
    k_weights<<<blocks(nv), TB, 0, stream>>>(w.pair, n_pairs, w.prob, w.colsum, n_targets, path_len, out_src, out_dst,
                                             out_w, out_count);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}
