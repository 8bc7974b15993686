// Native host construction of the PPMI graph (UDAGCN / AdaGCN(ppmi) start-up).
//
// Replaces the Python random-walk loop of PPMIConv.norm (pygda/nn/ppmi_conv.py:98-172),
// which walks `passes`(=40) times from every node over the symmetrised neighbour sets with
// np.random and takes minutes on the citation graphs.  Same estimator, own generator
// (counter-based, keyed on (seed, pass, start node): reproducible and order-independent), so
// parity with the reference is statistical; the reference's exact np.random stream is
// replayed only by the test oracle.
//   counts[a][b] = visits of b on walks started at a (walk length ~ U{1..path_len})
//   p[a][b]      = counts / sum_b counts                          (ppmi_conv.py:109-117,150)
//   colsum[b]    = sum_a p[a][b]                                  (:152-155)
//   w[a][b]      = max(log(p / colsum[b] * |{b}| / path_len), 0)  (:157-163), zero weights kept
// The edge list comes back sorted by (a, b); self loops and normalisation are applied by
// gda_build_csr_norm afterwards (degree over the SOURCE side, cached_gcn_conv.py:98-103).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/gda_hip.h"
#include "gda_edge_list.h"

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
    uint64_t below(uint64_t n) { return (uint64_t)(((unsigned __int128)next() * n) >> 64); }
};

}  // namespace


extern "C" int gda_ppmi_build_host(const int64_t* src_host, const int64_t* dst_host, int64_t E,
                                   int64_t N, int path_len, int passes, uint64_t seed,
                                   gda_edge_list** out) {
    if (!out || (E > 0 && (!src_host || !dst_host))) return GDA_E_NULL;
    if (E < 0 || N < 0 || N >= INT32_MAX || path_len < 1 || passes < 1) return GDA_E_SIZE;
    // symmetrised, de-duplicated, sorted neighbour lists (the reference's adj_dict of sets)
    std::vector<int64_t> deg(N + 1, 0);
    for (int64_t e = 0; e < E; ++e) {
        const int64_t a = src_host[e], b = dst_host[e];
        if (a < 0 || a >= N || b < 0 || b >= N) return GDA_E_SIZE;
        ++deg[a + 1]; ++deg[b + 1];
    }
    for (int64_t v = 0; v < N; ++v) deg[v + 1] += deg[v];
    std::vector<int32_t> nb(deg[N]);
    {
        std::vector<int64_t> cur(deg.begin(), deg.end() - 1);
        for (int64_t e = 0; e < E; ++e) {
            nb[cur[src_host[e]]++] = (int32_t)dst_host[e];
            nb[cur[dst_host[e]]++] = (int32_t)src_host[e];
        }
    }
    std::vector<int64_t> ptr(N + 1, 0);
    for (int64_t v = 0; v < N; ++v) {
        auto b = nb.begin() + deg[v], e = nb.begin() + deg[v + 1];
        std::sort(b, e);
        ptr[v + 1] = ptr[v] + (std::unique(b, e) - b);
    }
    std::vector<int32_t> adj(ptr[N]);
    for (int64_t v = 0; v < N; ++v)
        std::copy(nb.begin() + deg[v], nb.begin() + deg[v] + (ptr[v + 1] - ptr[v]), adj.begin() + ptr[v]);

    gda_edge_list* L = new (std::nothrow) gda_edge_list();
    if (!L) return GDA_E_SIZE;
    std::vector<double> colsum(N, 0.0);
    std::vector<char> seen(N, 0);
    std::vector<int32_t> visits;
    std::vector<double> prob;                     // parallel to L->src/dst until weights are final
    for (int64_t a = 0; a < N; ++a) {
        if (ptr[a + 1] == ptr[a]) continue;        // no neighbours: never a key of adj_dict
        visits.clear();
        for (int p = 0; p < passes; ++p) {
            SplitMix rng(seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(p + 1)) ^ (0x9E3779B97F4A7C15ull * (uint64_t)(a + 1)));
            const int steps = 1 + (int)rng.below((uint64_t)path_len);
            int64_t cur = a;
            for (int s = 0; s < steps; ++s) {
                const int64_t d = ptr[cur + 1] - ptr[cur];
                const int32_t b = adj[ptr[cur] + (int64_t)rng.below((uint64_t)d)];
                visits.push_back(b);
                cur = b;
            }
        }
        std::sort(visits.begin(), visits.end());
        const double total = (double)visits.size();
        for (size_t i = 0; i < visits.size();) {
            size_t j = i;
            while (j < visits.size() && visits[j] == visits[i]) ++j;
            const double pr = (double)(j - i) / total;
            L->src.push_back(a);
            L->dst.push_back(visits[i]);
            prob.push_back(pr);
            colsum[visits[i]] += pr;
            seen[visits[i]] = 1;
            i = j;
        }
    }
    int64_t n_targets = 0;
    for (int64_t v = 0; v < N; ++v) n_targets += seen[v];
    L->w.resize(prob.size());
    for (size_t k = 0; k < prob.size(); ++k) {
        const double v = std::log(prob[k] / colsum[L->dst[k]] * (double)n_targets / (double)path_len);
        L->w[k] = (float)(v > 0.0 ? v : 0.0);
    }
    *out = L;
    return GDA_OK;
}

extern "C" int64_t gda_edge_list_size(const gda_edge_list* l) { return l ? (int64_t)l->src.size() : 0; }

extern "C" int gda_edge_list_fetch(const gda_edge_list* l, int64_t* src_out, int64_t* dst_out,
                                   float* w_out) {
    if (!l) return GDA_E_NULL;
    if (l->src.empty()) return GDA_OK;
    if (!src_out || !dst_out || (!w_out && !l->w.empty())) return GDA_E_NULL;
    std::memcpy(src_out, l->src.data(), l->src.size() * sizeof(int64_t));
    std::memcpy(dst_out, l->dst.data(), l->dst.size() * sizeof(int64_t));
    if (!l->w.empty()) std::memcpy(w_out, l->w.data(), l->w.size() * sizeof(float));
    return GDA_OK;
}

extern "C" void gda_edge_list_destroy(gda_edge_list* l) { delete l; }
