// Native host neighbour sampler (mini-batch assembly for the data-parallel path).
//
// Replaces the C++ sampler behind PyG's NeighborLoader (pyg-lib / torch-sparse), which
// pygda's trainers construct at pygda/models/a2gnn.py:260-277 (same block in every trainer):
// for a batch of seed nodes and fan-outs [k_1..k_L], hop l samples, for every node first
// reached in hop l-1, up to k_l of its in-neighbours WITHOUT replacement (k = -1: all) and
// keeps the sampled edges (neighbour -> node).  Output nodes are the seeds first, then newly
// reached nodes in discovery order; edges are relabelled to that local numbering, grouped
// by destination in frontier order.  PyG's RNG stream cannot be matched (SURVEY §7 "sampler
// fidelity"); parity is structural + exact for fan-out -1.
//
// The per-node draws come from a counter-based generator keyed on (seed, hop, node), so a
// batch is reproducible and independent of traversal order.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "../../include/gda_hip.h"

namespace {

struct SplitMix {
    uint64_t s;
    explicit SplitMix(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    // unbiased enough for sampling: 64-bit multiply-shift
    uint64_t below(uint64_t n) { return (uint64_t)(((unsigned __int128)next() * n) >> 64); }
};

}  // namespace

// fork-join parallel-for over [0, n): contiguous chunks, one per worker (the picks of a frontier
// node depend only on (seed, hop, node), so any partition gives the same batch)
template <class F>
static void parallel_for(int64_t n, int workers, F&& body) {
    if (workers <= 1 || n < 4096) { body(0, n, 0); return; }
    std::vector<std::thread> pool;
    const int64_t chunk = (n + workers - 1) / workers;
    for (int w = 0; w < workers; ++w) {
        const int64_t b = w * chunk, e = std::min<int64_t>(n, b + chunk);
        if (b >= e) break;
        pool.emplace_back([&body, b, e, w] { body(b, e, w); });
    }
    for (auto& t : pool) t.join();
}

// Batch-local map global node id -> local id: open addressing over a power-of-two table that holds the batch only
// (~160 k nodes at cfg-S: 2^19 slots x 12 B = 6 MB, cache resident), instead of N-sized `local` / claim arrays
// (60 MB per loader at 5 M nodes: every probe a DRAM miss, 160 k scattered resets per batch).
struct NodeMap {
    std::vector<int64_t> key;          // -1 = empty
    std::vector<int32_t> val;
    int bits = 0;
    int64_t count = 0;
    void reset(int64_t expect) {
        int b = 12;
        while (((int64_t)1 << b) < 2 * expect) ++b;
        if (b != bits) { bits = b; key.assign((size_t)1 << b, -1); val.assign((size_t)1 << b, -1); }
        else std::fill(key.begin(), key.end(), (int64_t)-1);
        count = 0;
    }
    inline size_t slot(int64_t v) const { return (size_t)(((uint64_t)v * 0x9E3779B97F4A7C15ull) >> (64 - bits)); }
    void grow() {
        std::vector<int64_t> ok; std::vector<int32_t> ov;
        ok.swap(key); ov.swap(val);
        ++bits;
        key.assign((size_t)1 << bits, -1); val.assign((size_t)1 << bits, -1);
        const size_t mask = ((size_t)1 << bits) - 1;
        for (size_t i = 0; i < ok.size(); ++i)
            if (ok[i] >= 0) { size_t p = slot(ok[i]); while (key[p] >= 0) p = (p + 1) & mask; key[p] = ok[i]; val[p] = ov[i]; }
    }
    // local id of v; a new node gets `next` (the caller appends it to its node list when the returned id == next)
    inline int32_t find_or_insert(int64_t v, int32_t next) {
        const size_t mask = ((size_t)1 << bits) - 1;
        size_t p = slot(v);
        while (true) {
            const int64_t k = key[p];
            if (k == v) return val[p];
            if (k < 0) break;
            p = (p + 1) & mask;
        }
        if (2 * (count + 1) > ((int64_t)1 << bits)) { grow(); return find_or_insert(v, next); }
        key[p] = v; val[p] = next; ++count;
        return next;
    }
    inline void prefetch(int64_t v) const { __builtin_prefetch(&key[slot(v)]); }
};

struct gda_sampler {
    int64_t N = 0;
    int workers = 1;
    std::vector<int64_t> pick_off, picks;   // per-hop scratch: offsets / picked global ids of the frontier
    std::vector<int64_t> in_ptr;      // [N+1]  in-neighbour lists (sources of edges into v), edge order
    std::vector<int64_t> in_src;      // [E]
    NodeMap map;                      // global -> local id of the batch being built
    int64_t last_nodes = 4096;        // size hint for the next batch's map
    // last sampled batch
    std::vector<int64_t> nodes, esrc, edst;
};

extern "C" int gda_sampler_create(const int64_t* src_host, const int64_t* dst_host, int64_t E,
                                  int64_t N, gda_sampler** out) {
    if (!out || (E > 0 && (!src_host || !dst_host))) return GDA_E_NULL;
    if (E < 0 || N < 0) return GDA_E_SIZE;
    gda_sampler* s = new (std::nothrow) gda_sampler();
    if (!s) return GDA_E_SIZE;
    s->N = N;
    s->in_ptr.assign(N + 1, 0);
    for (int64_t e = 0; e < E; ++e) {
        if (src_host[e] < 0 || src_host[e] >= N || dst_host[e] < 0 || dst_host[e] >= N) { delete s; return GDA_E_SIZE; }
        ++s->in_ptr[dst_host[e] + 1];
    }
    for (int64_t v = 0; v < N; ++v) s->in_ptr[v + 1] += s->in_ptr[v];
    s->in_src.resize(E);
    std::vector<int64_t> cur(s->in_ptr.begin(), s->in_ptr.end() - 1);
    for (int64_t e = 0; e < E; ++e) s->in_src[cur[dst_host[e]]++] = src_host[e];   // stable: edge order kept
    *out = s;
    return GDA_OK;
}

extern "C" void gda_sampler_destroy(gda_sampler* s) { delete s; }

extern "C" int gda_sampler_set_threads(gda_sampler* s, int workers) {
    if (!s) return GDA_E_NULL;
    if (workers < 1 || workers > 256) return GDA_E_SIZE;
    s->workers = workers;
    return GDA_OK;
}

extern "C" int gda_sampler_sample(gda_sampler* s, const int64_t* seeds_host, int64_t n_seeds,
                                  const int32_t* fanouts, int L, uint64_t rng_seed,
                                  int64_t* n_nodes_out, int64_t* n_edges_out) {
    if (!s || !n_nodes_out || !n_edges_out || (n_seeds > 0 && !seeds_host) || (L > 0 && !fanouts)) return GDA_E_NULL;
    if (n_seeds < 0 || L < 0) return GDA_E_SIZE;
    s->nodes.clear(); s->esrc.clear(); s->edst.clear();
    s->map.reset(std::max<int64_t>(s->last_nodes, n_seeds));
    for (int64_t i = 0; i < n_seeds; ++i) {
        const int64_t v = seeds_host[i];
        if (v < 0 || v >= s->N) return GDA_E_SIZE;
        const int32_t next = (int32_t)s->nodes.size();
        if (s->map.find_or_insert(v, next) == next) s->nodes.push_back(v);
    }
    int64_t frontier_begin = 0;
    for (int hop = 0; hop < L; ++hop) {
        const int64_t frontier_end = (int64_t)s->nodes.size();
        const int64_t nf = frontier_end - frontier_begin;
        const int32_t k = fanouts[hop];
        // 1. how many neighbours each frontier node contributes -> offsets (sequential; the in_ptr reads are random)
        s->pick_off.resize(nf + 1);
        s->pick_off[0] = 0;
        for (int64_t f = 0; f < nf; ++f) {
            if (f + 16 < nf) __builtin_prefetch(&s->in_ptr[s->nodes[frontier_begin + f + 16]]);
            const int64_t v = s->nodes[frontier_begin + f];
            const int64_t deg = s->in_ptr[v + 1] - s->in_ptr[v];
            s->pick_off[f + 1] = s->pick_off[f] + ((k < 0 || deg <= k) ? deg : k);
        }
        s->picks.resize(s->pick_off[nf]);
        // 2. the picks themselves, in parallel: the cache-missing part (random reads of the edge array)
        parallel_for(nf, s->workers, [&](int64_t fb, int64_t fe, int) {
            std::vector<int64_t> scratch;
            for (int64_t f = fb; f < fe; ++f) {
                if (f + 8 < fe) __builtin_prefetch(s->in_src.data() + s->in_ptr[s->nodes[frontier_begin + f + 8]]);
                const int64_t v = s->nodes[frontier_begin + f];
                const int64_t b = s->in_ptr[v], deg = s->in_ptr[v + 1] - b;
                int64_t* out = s->picks.data() + s->pick_off[f];
                if (k < 0 || deg <= k) {
                    for (int64_t j = 0; j < deg; ++j) out[j] = s->in_src[b + j];
                } else {
                    // k distinct positions out of deg: partial Fisher-Yates on an index scratch, then
                    // restore list order so the kept edges stay in original edge order
                    SplitMix rng(rng_seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(hop + 1)) ^
                                 (0x9E3779B97F4A7C15ull * (uint64_t)(v + 1)));
                    scratch.resize(deg);
                    for (int64_t j = 0; j < deg; ++j) scratch[j] = j;
                    for (int32_t j = 0; j < k; ++j) {
                        const int64_t r = j + (int64_t)rng.below((uint64_t)(deg - j));
                        std::swap(scratch[j], scratch[r]);
                    }
                    std::sort(scratch.begin(), scratch.begin() + k);
                    for (int32_t j = 0; j < k; ++j) out[j] = s->in_src[b + scratch[j]];
                }
            }
        });
        // 3. relabel in pick order: the first occurrence of a node among the picks gives it the next local id
        //    (discovery order), every pick becomes an edge (local source -> local frontier node).  One pass over a
        //    cache-resident map -- sequential by nature, ~20 ns per pick.
        const int64_t np = s->pick_off[nf];
        const int64_t e0 = (int64_t)s->esrc.size();
        s->esrc.resize(e0 + np);
        s->edst.resize(e0 + np);
        for (int64_t f = 0; f < nf; ++f) {
            const int64_t lv = frontier_begin + f;                 // frontier nodes are nodes[frontier_begin ..]: their local ids
            for (int64_t q = s->pick_off[f]; q < s->pick_off[f + 1]; ++q) {
                if (q + 8 < np) s->map.prefetch(s->picks[q + 8]);
                const int64_t u = s->picks[q];
                const int32_t next = (int32_t)s->nodes.size();
                const int32_t lu = s->map.find_or_insert(u, next);
                if (lu == next) s->nodes.push_back(u);
                s->esrc[e0 + q] = lu;
                s->edst[e0 + q] = lv;
            }
        }
        frontier_begin = frontier_end;
    }
    s->last_nodes = (int64_t)s->nodes.size();
    *n_nodes_out = (int64_t)s->nodes.size();
    *n_edges_out = (int64_t)s->esrc.size();
    return GDA_OK;
}

extern "C" int gda_sampler_fetch(const gda_sampler* s, int64_t* nodes_out, int64_t* esrc_out,
                                 int64_t* edst_out) {
    if (!s) return GDA_E_NULL;
    if (!s->nodes.empty()) { if (!nodes_out) return GDA_E_NULL; std::memcpy(nodes_out, s->nodes.data(), s->nodes.size() * sizeof(int64_t)); }
    if (!s->esrc.empty()) {
        if (!esrc_out || !edst_out) return GDA_E_NULL;
        std::memcpy(esrc_out, s->esrc.data(), s->esrc.size() * sizeof(int64_t));
        std::memcpy(edst_out, s->edst.data(), s->edst.size() * sizeof(int64_t));
    }
    return GDA_OK;
}


// The normalised adjacency of the LAST sampled batch as the two CSRs the aggregation kernels consume -- built where
// the structure is known instead of re-derived on the device.  gda_build_csr_norm (csrc/gda_graph.hip) gets the
// relabelled edge list back from the host and spends two radix sorts, a scan and five small kernels per batch and
// domain on it (0.5 ms of a 7.6 ms cfg-S step); the sampler already holds the edges grouped by destination.
// Output = exactly what gda_build_csr_norm(esrc, edst, w = NULL, E, N = n_nodes, fill 1, add_self_loops = 1,
// normalize = 1, degree_side = col) writes (pygda/nn/prop_gcn_conv.py:64-81, gcn_norm with unit weights):
//   rows of the by-destination CSR: the kept (non-loop) edges into the node in edge order, then its self loop;
//   rows of the by-source CSR: the kept edges out of the node in edge order, then its self loop;
//   deg[i] = entries of by-destination row i, dis = 1 / sqrt(deg), val = (dis[src] * 1) * dis[dst];
// arrays of n_nodes + 1 and n_edges + n_nodes entries (the capacity layout of the device builder; rowptr[n_nodes] =
// number of stored entries).
extern "C" int gda_sampler_csr_norm(const gda_sampler* s, int32_t* rowptr, int32_t* colidx, float* val,
                                    int32_t* t_rowptr, int32_t* t_colidx, float* t_val) {
    if (!s || !rowptr || !t_rowptr) return GDA_E_NULL;
    const int64_t n = (int64_t)s->nodes.size(), E = (int64_t)s->esrc.size();
    if (E + n >= INT32_MAX) return GDA_E_SIZE;
    if (E + n > 0 && (!colidx || !val || !t_colidx || !t_val)) return GDA_E_NULL;
    std::vector<int32_t> cnt_d((size_t)n + 1, 0), cnt_s((size_t)n + 1, 0);
    for (int64_t e = 0; e < E; ++e) {
        if (s->esrc[e] == s->edst[e]) continue;                       // add_remaining_self_loops: loops are re-appended
        ++cnt_d[s->edst[e] + 1];
        ++cnt_s[s->esrc[e] + 1];
    }
    rowptr[0] = t_rowptr[0] = 0;
    for (int64_t i = 0; i < n; ++i) {
        rowptr[i + 1] = rowptr[i] + cnt_d[i + 1] + 1;
        t_rowptr[i + 1] = t_rowptr[i] + cnt_s[i + 1] + 1;
    }
    std::vector<float> dis((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const float deg = (float)(rowptr[i + 1] - rowptr[i]);             // a sum of ones: exact
        dis[i] = 1.0f / std::sqrt(deg);
    }
    std::vector<int32_t> cur_d(rowptr, rowptr + n), cur_s(t_rowptr, t_rowptr + n);
    for (int64_t e = 0; e < E; ++e) {
        const int64_t u = s->esrc[e], v = s->edst[e];
        if (u == v) continue;
        const float w = (dis[u] * 1.0f) * dis[v];
        colidx[cur_d[v]] = (int32_t)u; val[cur_d[v]++] = w;
        t_colidx[cur_s[u]] = (int32_t)v; t_val[cur_s[u]++] = w;
    }
    for (int64_t i = 0; i < n; ++i) {                                     // the appended loops come last in their rows
        const float w = (dis[i] * 1.0f) * dis[i];
        colidx[cur_d[i]] = (int32_t)i; val[cur_d[i]] = w;
        t_colidx[cur_s[i]] = (int32_t)i; t_val[cur_s[i]] = w;
    }
    return GDA_OK;
}


// Host side of the MMD sample-gradient scatter (pygda/utils/mmd.py:148-153 draws the row samples
// on the host): CSR of the 0/1 selection matrix for idx [times, n] -- rows = feature rows,
// columns = positions t*m + offset + r of the [times, m, d] sample-gradient buffer, stable in
// sample order.  A counting sort: O(times*n + num_rows), microseconds.
extern "C" int gda_selection_csr_host(const int64_t* idx_host, int times, int64_t n, int64_t num_rows,
                                      int64_t offset, int64_t m, int32_t* rowptr_out,
                                      int32_t* colidx_out) {
    if (!idx_host || !rowptr_out || (times * n > 0 && !colidx_out)) return GDA_E_NULL;
    if (times < 0 || n < 0 || num_rows < 0 || (int64_t)times * m >= INT32_MAX) return GDA_E_SIZE;
    const int64_t total = (int64_t)times * n;
    std::memset(rowptr_out, 0, sizeof(int32_t) * (size_t)(num_rows + 1));
    for (int64_t p = 0; p < total; ++p) {
        const int64_t r = idx_host[p];
        if (r < 0 || r >= num_rows) return GDA_E_SIZE;
        ++rowptr_out[r + 1];
    }
    for (int64_t r = 0; r < num_rows; ++r) rowptr_out[r + 1] += rowptr_out[r];
    std::vector<int32_t> cur(rowptr_out, rowptr_out + num_rows);
    for (int t = 0; t < times; ++t)
        for (int64_t r = 0; r < n; ++r)
            colidx_out[cur[idx_host[t * n + r]]++] = (int32_t)(t * m + offset + r);
    return GDA_OK;
}
