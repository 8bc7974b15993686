// Graph ingestion for the aggregation path: COO edge list -> self-loop merge ->
// (optional) symmetric GCN normalisation -> CSR by destination + CSR by source.
//
// Restates gcn_norm (pygda/nn/prop_gcn_conv.py:64-81) and CachedGCNConv.norm
// (pygda/nn/cached_gcn_conv.py:88-103).  One-off per graph / mini-batch; the result is
// cached by the host side, so the only requirement here is determinism and that the
// edges of each row stay in original edge order (stable radix sort), which makes the
// sequential row sums of the SpMM kernel reproduce the CPU scatter-add order.
//
// Data-dependent sizes never come back to the host: the compacted edge list is padded
// to its capacity E+N with sentinel keys (= N) that sort to the end, and rowptr[N]
// carries the true nnz.
#include "gda_common.h"

#include <hipcub/hipcub.hpp>

namespace {

constexpr int TB = 256;

struct GraphWs {
    int32_t* flag;       // [E]
    int32_t* pos;        // [E]
    int32_t* loop_last;  // [N]
    int32_t* nsrc;       // [cap]
    int32_t* ndst;       // [cap]
    float* nw;           // [cap]
    int32_t* iota;       // [cap]
    int32_t* keys_out;   // [cap]
    int32_t* perm;       // [cap]
    int32_t* inv_perm;   // [cap] list index -> position in the by-destination CSR
    float* deg;          // [N]
    float* dis;          // [N]
    void* cub;           // hipcub temp storage
    size_t cub_bytes;
    size_t total;
};

size_t cub_bytes_needed(int64_t E, int64_t cap) {
    size_t a = 0, b = 0;
    int32_t* p = nullptr;
    hipcub::DeviceScan::ExclusiveSum(nullptr, a, p, p, (int)(E > 0 ? E : 1));
    hipcub::DeviceRadixSort::SortPairs(nullptr, b, p, p, p, p, (int)cap, 0, 32);
    return a > b ? a : b;
}

GraphWs carve(void* base, int64_t E, int64_t N) {
    const int64_t cap = E + N;
    GraphWs w{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void* p = base ? (void*)((char*)base + off) : nullptr;
        off += gda_align_up(bytes, 256);
        return p;
    };
    w.flag = (int32_t*)take(sizeof(int32_t) * (E + 1));
    w.pos = (int32_t*)take(sizeof(int32_t) * (E + 1));
    w.loop_last = (int32_t*)take(sizeof(int32_t) * (N + 1));
    w.nsrc = (int32_t*)take(sizeof(int32_t) * (cap + 1));
    w.ndst = (int32_t*)take(sizeof(int32_t) * (cap + 1));
    w.nw = (float*)take(sizeof(float) * (cap + 1));
    w.iota = (int32_t*)take(sizeof(int32_t) * (cap + 1));
    w.keys_out = (int32_t*)take(sizeof(int32_t) * (cap + 1));
    w.perm = (int32_t*)take(sizeof(int32_t) * (cap + 1));
    w.inv_perm = (int32_t*)take(sizeof(int32_t) * (cap + 1));
    w.deg = (float*)take(sizeof(float) * (N + 1));
    w.dis = (float*)take(sizeof(float) * (N + 1));
    w.cub_bytes = cub_bytes_needed(E, cap);
    w.cub = take(w.cub_bytes);
    w.total = off;
    return w;
}

// prop_gcn_conv.py:72 / cached_gcn_conv.py:95 (add_remaining_self_loops): mask = row != col;
// an existing loop hands its weight to the appended loop of that node (last one wins).
__global__ void k_flag(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                       int64_t E, int add_self_loops, int32_t* __restrict__ flag,
                       int32_t* __restrict__ loop_last) {
    int64_t e = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (e >= E) return;
    const int64_t s = src[e], t = dst[e];
    const bool loop = (s == t);
    flag[e] = (add_self_loops && loop) ? 0 : 1;
    if (add_self_loops && loop) atomicMax(&loop_last[s], (int32_t)e);
}

__global__ void k_fill(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                       const float* __restrict__ w, int64_t E, int64_t N, float fill_value,
                       int add_self_loops, const int32_t* __restrict__ flag,
                       const int32_t* __restrict__ pos, const int32_t* __restrict__ loop_last,
                       int32_t* __restrict__ nsrc, int32_t* __restrict__ ndst,
                       float* __restrict__ nw, int32_t* __restrict__ iota) {
    const int64_t cap = E + N;
    int64_t idx = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (idx >= cap) return;
    iota[idx] = (int32_t)idx;
    const int32_t n_kept = E > 0 ? pos[E - 1] + flag[E - 1] : 0;
    const int64_t total = n_kept + (add_self_loops == 1 ? N : 0);
    if (idx < E && flag[idx]) {                      // a kept edge, compacted in edge order
        const int32_t p = pos[idx];
        nsrc[p] = (int32_t)src[idx];
        ndst[p] = (int32_t)dst[idx];
        nw[p] = w ? w[idx] : 1.0f;
    }
    if (add_self_loops == 1 && idx < N) {            // appended loops, node order, LAST (mode 2: dropped, none added)
        const int64_t p = n_kept + idx;
        const int32_t l = loop_last[idx];
        nsrc[p] = (int32_t)idx;
        ndst[p] = (int32_t)idx;
        nw[p] = l >= 0 ? (w ? w[l] : 1.0f) : fill_value;
    }
    if (idx >= total) {                              // padding: sentinel key sorts to the end
        nsrc[idx] = (int32_t)N;
        ndst[idx] = (int32_t)N;
        nw[idx] = 0.0f;
    }
}

// after the stable sort by `key`: rows of the CSR are runs of equal keys
__global__ void k_csr_fill(const int32_t* __restrict__ keys_sorted, const int32_t* __restrict__ perm,
                           const int32_t* __restrict__ other, const float* __restrict__ nw,
                           int64_t cap, int64_t N, int32_t* __restrict__ colidx,
                           float* __restrict__ val) {
    int64_t k = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (k >= cap) return;
    if (keys_sorted[k] >= N) return;
    const int32_t p = perm[k];
    colidx[k] = other[p];
    val[k] = nw[p];
}

// inv[perm[k]] = k   (list index -> CSR position), for the transpose-to-forward edge map
__global__ void k_invert(const int32_t* __restrict__ perm, int64_t cap, int32_t* __restrict__ inv) {
    int64_t k = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (k < cap) inv[perm[k]] = (int32_t)k;
}

// t_to_fwd[k'] = position in the by-destination CSR of the edge stored at k' of the by-source CSR
__global__ void k_edge_map(const int32_t* __restrict__ keys_sorted, const int32_t* __restrict__ perm,
                           const int32_t* __restrict__ inv, int64_t cap, int64_t N,
                           int32_t* __restrict__ t_to_fwd) {
    int64_t k = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (k >= cap || keys_sorted[k] >= N) return;
    t_to_fwd[k] = inv[perm[k]];
}

__global__ void k_rowptr(const int32_t* __restrict__ keys_sorted, int64_t cap, int64_t N,
                         int32_t* __restrict__ rowptr) {
    int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i > N) return;
    int64_t lo = 0, hi = cap;                        // lower_bound(keys, i)
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys_sorted[mid] < (int32_t)i) lo = mid + 1; else hi = mid;
    }
    rowptr[i] = (int32_t)lo;
}

// prop_gcn_conv.py:78-80: deg = scatter_add(w, col); deg^-1/2; inf -> 0.
// One thread per node, sequential in edge order (the CPU scatter-add order).
__global__ void k_degree(const int32_t* __restrict__ rowptr, const float* __restrict__ val,
                         int64_t N, float* __restrict__ deg, float* __restrict__ dis) {
    int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= N) return;
    float s = 0.0f;
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) s = __fadd_rn(s, val[k]);
    deg[i] = s;
    float r = 1.0f / sqrtf(s);   // pow(-0.5) == IEEE 1/sqrt on the CPU path (sqrtf and / are correctly rounded; __fsqrt_rn is not)
    if (isinf(r)) r = 0.0f;
    dis[i] = r;
}

__device__ __forceinline__ int32_t row_of(const int32_t* __restrict__ rowptr, int64_t N, int32_t k) {
    int64_t lo = 0, hi = N;                          // largest i with rowptr[i] <= k
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (rowptr[mid] <= k) lo = mid; else hi = mid - 1;
    }
    return (int32_t)lo;
}

// prop_gcn_conv.py:81: deg_inv_sqrt[row] * edge_weight * deg_inv_sqrt[col]   (row=src, col=dst)
// key_is_dst: rows of this CSR are destinations (colidx = src), else sources (colidx = dst).
__global__ void k_normalize(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                            int64_t N, const float* __restrict__ dis, int key_is_dst,
                            float* __restrict__ val) {
    int64_t k = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (k >= rowptr[N]) return;
    const int32_t r = row_of(rowptr, N, (int32_t)k);
    const int32_t c = colidx[k];
    const int32_t s = key_is_dst ? c : r;
    const int32_t t = key_is_dst ? r : c;
    val[k] = __fmul_rn(__fmul_rn(dis[s], val[k]), dis[t]);
}

__global__ void k_csr_to_coo(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                             int64_t N, int64_t* __restrict__ src_out, int64_t* __restrict__ dst_out) {
    int64_t k = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (k >= rowptr[N]) return;
    src_out[k] = colidx[k];
    dst_out[k] = row_of(rowptr, N, (int32_t)k);
}

// ---- long-row discovery for the load-balanced SpMM (gda_row_split_build) ----
__global__ void k_split_count(const int32_t* __restrict__ rowptr, int64_t n_rows, int32_t T,
                              int32_t* __restrict__ is_long, int32_t* __restrict__ nchunks) {
    int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n_rows) return;
    const int32_t deg = rowptr[i + 1] - rowptr[i];
    const bool lg = deg > T;
    is_long[i] = lg ? 1 : 0;
    nchunks[i] = lg ? (deg + T - 1) / T : 0;
}

__global__ void k_split_emit(const int32_t* __restrict__ is_long, const int32_t* __restrict__ nchunks,
                             const int32_t* __restrict__ long_off, const int32_t* __restrict__ chunk_off,
                             int64_t n_rows, int32_t* __restrict__ long_rows,
                             int32_t* __restrict__ long_chunk_ptr, int32_t* __restrict__ chunk_long,
                             int32_t* __restrict__ counts_out) {
    int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n_rows) return;
    if (i == n_rows - 1) {
        const int32_t nl = long_off[i] + is_long[i], nc = chunk_off[i] + nchunks[i];
        counts_out[0] = nl;
        counts_out[1] = nc;
        long_chunk_ptr[nl] = nc;
    }
    if (!is_long[i]) return;
    const int32_t li = long_off[i], c0 = chunk_off[i];
    long_rows[li] = (int32_t)i;
    long_chunk_ptr[li] = c0;
    for (int32_t c = 0; c < nchunks[i]; ++c) chunk_long[c0 + c] = li;
}

struct SplitWs { int32_t* is_long; int32_t* nchunks; int32_t* long_off; int32_t* chunk_off; void* cub; size_t cub_bytes; size_t total; };

SplitWs carve_split(void* base, int64_t n_rows) {
    SplitWs w{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void* p = base ? (void*)((char*)base + off) : nullptr;
        off += gda_align_up(bytes, 256);
        return p;
    };
    w.is_long = (int32_t*)take(sizeof(int32_t) * (n_rows + 1));
    w.nchunks = (int32_t*)take(sizeof(int32_t) * (n_rows + 1));
    w.long_off = (int32_t*)take(sizeof(int32_t) * (n_rows + 1));
    w.chunk_off = (int32_t*)take(sizeof(int32_t) * (n_rows + 1));
    size_t cb = 0;
    int32_t* p = nullptr;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, cb, p, p, (int)(n_rows > 0 ? n_rows : 1));
    w.cub_bytes = cb;
    w.cub = take(cb);
    w.total = off;
    return w;
}

int bits_for(int64_t n) {
    int b = 1;
    while (((int64_t)1 << b) <= n) ++b;
    return b;
}

}  // namespace

extern "C" size_t gda_graph_workspace_bytes(int64_t E, int64_t N) {
    if (E < 0 || N < 0) return 0;
    return carve(nullptr, E, N).total;
}

extern "C" int gda_build_csr_norm(const int64_t* src, const int64_t* dst, const float* w,
                                  int64_t E, int64_t N, float fill_value, int add_self_loops,
                                  int normalize, int degree_side,
                                  int32_t* rowptr, int32_t* colidx, float* val,
                                  int32_t* t_rowptr, int32_t* t_colidx, float* t_val,
                                  void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    return gda_build_csr_norm_map(src, dst, w, E, N, fill_value, add_self_loops, normalize, degree_side,
                                  rowptr, colidx, val, t_rowptr, t_colidx, t_val, nullptr, workspace,
                                  workspace_bytes, stream_);
}

extern "C" int gda_build_csr_norm_map(const int64_t* src, const int64_t* dst, const float* w,
                                      int64_t E, int64_t N, float fill_value, int add_self_loops,
                                      int normalize, int degree_side,
                                      int32_t* rowptr, int32_t* colidx, float* val,
                                      int32_t* t_rowptr, int32_t* t_colidx, float* t_val,
                                      int32_t* t_to_fwd,
                                      void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    if (E < 0 || N < 0 || E + N >= INT32_MAX) return GDA_E_SIZE;
    if ((E > 0 && (!src || !dst)) || !rowptr || !t_rowptr || !workspace) return GDA_E_NULL;
    const int64_t cap = E + N;
    if (cap > 0 && (!colidx || !val || !t_colidx || !t_val)) return GDA_E_NULL;
    GraphWs ws = carve(workspace, E, N);
    if (workspace_bytes < ws.total) return GDA_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    if (cap == 0) {
        GDA_HIP_TRY(hipMemsetAsync(rowptr, 0, sizeof(int32_t), stream));
        GDA_HIP_TRY(hipMemsetAsync(t_rowptr, 0, sizeof(int32_t), stream));
        return GDA_OK;
    }
    const unsigned gE = (unsigned)gda_cdiv(E > 0 ? E : 1, TB), gC = (unsigned)gda_cdiv(cap, TB),
                   gN = (unsigned)gda_cdiv(N + 1, TB);
    GDA_HIP_TRY(hipMemsetAsync(ws.loop_last, 0xFF, sizeof(int32_t) * (N + 1), stream));
    if (E > 0) {
        k_flag<<<gE, TB, 0, stream>>>(src, dst, E, add_self_loops, ws.flag, ws.loop_last);
        GDA_LAUNCH_CHECK();
        size_t cb = ws.cub_bytes;
        GDA_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(ws.cub, cb, ws.flag, ws.pos, (int)E, stream));
    }
    k_fill<<<gC, TB, 0, stream>>>(src, dst, w, E, N, fill_value, add_self_loops, ws.flag, ws.pos,
                                  ws.loop_last, ws.nsrc, ws.ndst, ws.nw, ws.iota);
    GDA_LAUNCH_CHECK();
    const int end_bit = bits_for(N);
    for (int pass = 0; pass < 2; ++pass) {
        const int32_t* key = pass == 0 ? ws.ndst : ws.nsrc;
        const int32_t* other = pass == 0 ? ws.nsrc : ws.ndst;
        int32_t* rp = pass == 0 ? rowptr : t_rowptr;
        int32_t* ci = pass == 0 ? colidx : t_colidx;
        float* vv = pass == 0 ? val : t_val;
        size_t cb = ws.cub_bytes;
        GDA_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(ws.cub, cb, key, ws.keys_out, ws.iota, ws.perm,
                                                      (int)cap, 0, end_bit, stream));
        k_csr_fill<<<gC, TB, 0, stream>>>(ws.keys_out, ws.perm, other, ws.nw, cap, N, ci, vv);
        GDA_LAUNCH_CHECK();
        k_rowptr<<<gN, TB, 0, stream>>>(ws.keys_out, cap, N, rp);
        GDA_LAUNCH_CHECK();
        if (t_to_fwd) {
            if (pass == 0) k_invert<<<gC, TB, 0, stream>>>(ws.perm, cap, ws.inv_perm);
            else k_edge_map<<<gC, TB, 0, stream>>>(ws.keys_out, ws.perm, ws.inv_perm, cap, N, t_to_fwd);
            GDA_LAUNCH_CHECK();
        }
    }
    if (normalize) {
        if (degree_side == 0) k_degree<<<gN, TB, 0, stream>>>(rowptr, val, N, ws.deg, ws.dis);
        else k_degree<<<gN, TB, 0, stream>>>(t_rowptr, t_val, N, ws.deg, ws.dis);
        GDA_LAUNCH_CHECK();
        k_normalize<<<gC, TB, 0, stream>>>(rowptr, colidx, N, ws.dis, 1, val);
        GDA_LAUNCH_CHECK();
        k_normalize<<<gC, TB, 0, stream>>>(t_rowptr, t_colidx, N, ws.dis, 0, t_val);
        GDA_LAUNCH_CHECK();
    }
    return GDA_OK;
}

extern "C" int gda_csr_to_coo(const int32_t* rowptr, const int32_t* colidx, int64_t N,
                              int64_t nnz_cap, int64_t* src_out, int64_t* dst_out,
                              gda_stream_t stream_) {
    if (!rowptr || !colidx || !src_out || !dst_out) return GDA_E_NULL;
    if (N < 0 || nnz_cap < 0) return GDA_E_SIZE;
    if (nnz_cap == 0) return GDA_OK;
    k_csr_to_coo<<<(unsigned)gda_cdiv(nnz_cap, TB), TB, 0, (hipStream_t)stream_>>>(
        rowptr, colidx, N, src_out, dst_out);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" size_t gda_row_split_workspace_bytes(int64_t n_rows) {
    if (n_rows < 0) return 0;
    return carve_split(nullptr, n_rows).total;
}

extern "C" int gda_row_split_build(const int32_t* rowptr, int64_t n_rows, int32_t threshold,
                                   int32_t* long_rows, int32_t* long_chunk_ptr, int32_t* chunk_long,
                                   int32_t* counts_out, void* workspace, size_t workspace_bytes,
                                   gda_stream_t stream_) {
    if (n_rows < 0 || n_rows >= INT32_MAX || threshold < 1) return GDA_E_SIZE;
    if (!rowptr || !long_rows || !long_chunk_ptr || !chunk_long || !counts_out || !workspace) return GDA_E_NULL;
    SplitWs ws = carve_split(workspace, n_rows);
    if (workspace_bytes < ws.total) return GDA_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    if (n_rows == 0) {
        GDA_HIP_TRY(hipMemsetAsync(counts_out, 0, 2 * sizeof(int32_t), stream));
        GDA_HIP_TRY(hipMemsetAsync(long_chunk_ptr, 0, sizeof(int32_t), stream));
        return GDA_OK;
    }
    const unsigned g = (unsigned)gda_cdiv(n_rows, TB);
    k_split_count<<<g, TB, 0, stream>>>(rowptr, n_rows, threshold, ws.is_long, ws.nchunks);
    GDA_LAUNCH_CHECK();
    size_t cb = ws.cub_bytes;
    GDA_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(ws.cub, cb, ws.is_long, ws.long_off, (int)n_rows, stream));
    cb = ws.cub_bytes;
    GDA_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(ws.cub, cb, ws.nchunks, ws.chunk_off, (int)n_rows, stream));
    k_split_emit<<<g, TB, 0, stream>>>(ws.is_long, ws.nchunks, ws.long_off, ws.chunk_off, n_rows, long_rows,
                                       long_chunk_ptr, chunk_long, counts_out);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}
