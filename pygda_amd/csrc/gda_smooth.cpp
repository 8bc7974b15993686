// Native host construction of the TDSS "smoothness" graphs (pygda/models/tdss.py:314-388).
//
//   K-hop  (tdss.py:374-386, TwoHopNeighbor :67-87): every round replaces the edge set by
//          coalesce(E  U  pattern(A.A) minus self loops)  -- spspmm + remove_self_loops + coalesce in
//          the reference.  One marker-array pass per node here, nodes spread over threads; the
//          result is sorted by (row, col) and duplicate-free, exactly what coalesce returns.
//   RW     (tdss.py:367-373): one uniform random walk of `walk_len` steps from every node along
//          row -> col (torch_cluster.random_walk semantics: a node without out-edges stays put,
//          parallel edges keep their multiplicity); the edge (visited, start) is set for every
//          visited node, start included.  The reference goes through a dense N x N matrix and
//          dense_to_sparse; here the pairs are bucketed by `visited` and de-duplicated, which gives
//          the same (row, col)-sorted list without the N^2 buffer.  Own counter-based generator
//          keyed on (seed, start node): statistical, not bit-wise, parity with torch_cluster.
#include <algorithm>
#include <cstdint>
#include <new>
#include <thread>
#include <vector>

#include "../../include/gda_hip.h"
#include "gda_edge_list.h"

namespace {

struct Mix64 {
    uint64_t s;
    explicit Mix64(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    uint64_t below(uint64_t n) { return (uint64_t)(((unsigned __int128)next() * n) >> 64); }
};

template <class F>
void parallel_nodes(int64_t N, int threads, F&& body) {
    threads = (int)std::max<int64_t>(1, std::min<int64_t>(threads, N / 1024 + 1));
    if (threads == 1) { body(0, 0, N); return; }
    std::vector<std::thread> pool;
    const int64_t per = (N + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
        const int64_t lo = t * per, hi = std::min<int64_t>(N, lo + per);
        if (lo >= hi) break;
        pool.emplace_back([&, t, lo, hi] { body(t, lo, hi); });
    }
    for (auto& th : pool) th.join();
}

// CSR by row; `dedup` drops parallel edges (pattern of A), otherwise multiplicity is kept
int build_rows(const int64_t* src, const int64_t* dst, int64_t E, int64_t N, bool dedup,
               std::vector<int64_t>& ptr, std::vector<int32_t>& nb) {
    ptr.assign(N + 1, 0);
    for (int64_t e = 0; e < E; ++e) {
        const int64_t a = src[e], b = dst[e];
        if (a < 0 || a >= N || b < 0 || b >= N) return GDA_E_SIZE;
        ++ptr[a + 1];
    }
    for (int64_t v = 0; v < N; ++v) ptr[v + 1] += ptr[v];
    nb.resize(ptr[N]);
    std::vector<int64_t> cur(ptr.begin(), ptr.end() - 1);
    for (int64_t e = 0; e < E; ++e) nb[cur[src[e]]++] = (int32_t)dst[e];
    for (int64_t v = 0; v < N; ++v) std::sort(nb.begin() + ptr[v], nb.begin() + ptr[v + 1]);
    if (!dedup) return GDA_OK;
    int64_t w = 0;
    for (int64_t v = 0; v < N; ++v) {
        const int64_t lo = ptr[v], hi = ptr[v + 1];
        ptr[v] = w;
        for (int64_t k = lo; k < hi; ++k)
            if (k == lo || nb[k] != nb[k - 1]) nb[w++] = nb[k];
    }
    ptr[N] = w;
    nb.resize(w);
    return GDA_OK;
}

}  // namespace

extern "C" int gda_two_hop_host(const int64_t* src_host, const int64_t* dst_host, int64_t E, int64_t N,
                                int rounds, int threads, gda_edge_list** out) {
    if (!out || (E > 0 && (!src_host || !dst_host))) return GDA_E_NULL;
    if (E < 0 || N < 0 || N >= INT32_MAX || rounds < 0) return GDA_E_SIZE;
    std::vector<int64_t> ptr;
    std::vector<int32_t> nb;
    if (int st = build_rows(src_host, dst_host, E, N, true, ptr, nb)) return st;
    for (int r = 0; r < rounds; ++r) {
        std::vector<std::vector<int32_t>> grown(N);
        parallel_nodes(N, threads, [&](int, int64_t lo, int64_t hi) {
            std::vector<int64_t> stamp(N, -1);
            for (int64_t i = lo; i < hi; ++i) {
                auto& mine = grown[i];
                for (int64_t k = ptr[i]; k < ptr[i + 1]; ++k) {          // E itself (self loops kept)
                    const int32_t j = nb[k];
                    if (stamp[j] != i) { stamp[j] = i; mine.push_back(j); }
                }
                for (int64_t k = ptr[i]; k < ptr[i + 1]; ++k) {          // A.A, self loops removed
                    const int32_t j = nb[k];
                    for (int64_t q = ptr[j]; q < ptr[j + 1]; ++q) {
                        const int32_t t = nb[q];
                        if (t != i && stamp[t] != i) { stamp[t] = i; mine.push_back(t); }
                    }
                }
                std::sort(mine.begin(), mine.end());
            }
        });
        int64_t total = 0;
        for (int64_t i = 0; i < N; ++i) { ptr[i] = total; total += (int64_t)grown[i].size(); }
        ptr[N] = total;
        if (total >= INT32_MAX) return GDA_E_SIZE;
        nb.resize(total);
        for (int64_t i = 0; i < N; ++i) std::copy(grown[i].begin(), grown[i].end(), nb.begin() + ptr[i]);
    }
    gda_edge_list* L = new (std::nothrow) gda_edge_list();
    if (!L) return GDA_E_WORKSPACE;
    L->src.resize(ptr[N]);
    L->dst.resize(ptr[N]);
    for (int64_t i = 0; i < N; ++i)
        for (int64_t k = ptr[i]; k < ptr[i + 1]; ++k) { L->src[k] = i; L->dst[k] = nb[k]; }
    *out = L;
    return GDA_OK;
}

extern "C" int gda_walk_smooth_host(const int64_t* src_host, const int64_t* dst_host, int64_t E, int64_t N,
                                    int walk_len, uint64_t seed, int threads, gda_edge_list** out) {
    if (!out || (E > 0 && (!src_host || !dst_host))) return GDA_E_NULL;
    if (E < 0 || N < 0 || N >= INT32_MAX || walk_len < 0) return GDA_E_SIZE;
    std::vector<int64_t> ptr;
    std::vector<int32_t> nb;
    if (int st = build_rows(src_host, dst_host, E, N, false, ptr, nb)) return st;
    const int64_t L1 = (int64_t)walk_len + 1;
    std::vector<int32_t> visited(N * L1);                     // walk[start][t]
    parallel_nodes(N, threads, [&](int, int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            Mix64 rng(seed * 0xD1342543DE82EF95ull + (uint64_t)i);
            int64_t v = i;
            visited[i * L1] = (int32_t)i;
            for (int64_t t = 1; t < L1; ++t) {
                const int64_t deg = ptr[v + 1] - ptr[v];
                if (deg > 0) v = nb[ptr[v] + (int64_t)rng.below((uint64_t)deg)];
                visited[i * L1 + t] = (int32_t)v;
            }
        }
    });
    // bucket the (visited, start) pairs by `visited`, starts ascending inside a bucket
    std::vector<int64_t> bptr(N + 1, 0);
    for (int64_t k = 0; k < N * L1; ++k) ++bptr[visited[k] + 1];
    for (int64_t v = 0; v < N; ++v) bptr[v + 1] += bptr[v];
    std::vector<int32_t> starts(N * L1);
    {
        std::vector<int64_t> cur(bptr.begin(), bptr.end() - 1);
        for (int64_t i = 0; i < N; ++i)
            for (int64_t t = 0; t < L1; ++t) starts[cur[visited[i * L1 + t]]++] = (int32_t)i;
    }
    gda_edge_list* L = new (std::nothrow) gda_edge_list();
    if (!L) return GDA_E_WORKSPACE;
    L->src.reserve(N * L1);
    L->dst.reserve(N * L1);
    for (int64_t v = 0; v < N; ++v)
        for (int64_t k = bptr[v]; k < bptr[v + 1]; ++k)
            if (k == bptr[v] || starts[k] != starts[k - 1]) { L->src.push_back(v); L->dst.push_back(starts[k]); }
    *out = L;
    return GDA_OK;
}

// ---------------------------------------------------------------------------------------------
// C = A * A for a CSR operator (host, threaded over rows): the two-step aggregation operator of a
// STATIC full-batch graph, so that a K-step propagation (pygda/nn/prop_gcn_conv.py:208-210 runs K
// dependent scatter passes) needs K/2 dependent launches.  Row i of C accumulates a_ij * a_jk in
// the fixed order (j in row order of A, k in row order of row j) into a dense per-thread
// accumulator; entries come out sorted by column.  Returned as an edge list (src = row, dst =
// column, w = value).
extern "C" int gda_csr_square_host(const int32_t* rowptr, const int32_t* colidx, const float* val, int64_t N,
                                   int threads, int64_t max_nnz, gda_edge_list** out) {
    if (!out || !rowptr || (rowptr && N > 0 && rowptr[N] > 0 && (!colidx || !val))) return GDA_E_NULL;
    if (N < 0 || N >= INT32_MAX) return GDA_E_SIZE;
    std::vector<std::vector<int32_t>> cols(N);
    std::vector<std::vector<float>> vals(N);
    parallel_nodes(N, threads, [&](int, int64_t lo, int64_t hi) {
        std::vector<float> acc(N, 0.f);
        std::vector<int64_t> stamp(N, -1);
        std::vector<int32_t> touched;
        for (int64_t i = lo; i < hi; ++i) {
            touched.clear();
            for (int32_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
                const int32_t j = colidx[p];
                const float a = val[p];
                for (int32_t q = rowptr[j]; q < rowptr[j + 1]; ++q) {
                    const int32_t k = colidx[q];
                    if (stamp[k] != i) { stamp[k] = i; acc[k] = 0.f; touched.push_back(k); }
                    acc[k] += a * val[q];
                }
            }
            std::sort(touched.begin(), touched.end());
            cols[i] = touched;
            vals[i].resize(touched.size());
            for (size_t t = 0; t < touched.size(); ++t) vals[i][t] = acc[touched[t]];
        }
    });
    int64_t total = 0;
    for (int64_t i = 0; i < N; ++i) total += (int64_t)cols[i].size();
    if (total >= INT32_MAX) return GDA_E_SIZE;
    gda_edge_list* L = new (std::nothrow) gda_edge_list();
    if (!L) return GDA_E_WORKSPACE;
    if (max_nnz >= 0 && total > max_nnz) { *out = L; return GDA_OK; }        // too dense to pay off: empty list
    L->src.reserve(total); L->dst.reserve(total); L->w.reserve(total);
    for (int64_t i = 0; i < N; ++i)
        for (size_t t = 0; t < cols[i].size(); ++t) {
            L->src.push_back(i); L->dst.push_back(cols[i][t]); L->w.push_back(vals[i][t]);
        }
    *out = L;
    return GDA_OK;
}
