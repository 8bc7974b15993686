// Device-side neighbour sampler: the mini-batch assembly of pygda's NeighborLoader call sites
// (pygda/models/a2gnn.py:260-277, same block in every trainer) without the host.
//
// Same contract and -- bit for bit -- the same batches as the native host sampler (gda_sampler.cpp): for seeds and
// fan-outs [k_1..k_L], hop l picks, for every node first reached in hop l-1, up to k_l of its in-neighbours without
// replacement (k < 0: all of them), in edge order; nodes are numbered seeds first, then in discovery order (the
// position of a node's first occurrence in the hop's pick list); every pick is an edge (local source -> local
// frontier node).  The per-node draws come from the same counter-based generator keyed on (seed, hop, node): a
// partial Fisher-Yates over the positions of the in-list, emulated here on a sparse position map (only the <= 2k
// touched positions are materialised), so a node picks the same positions on the host and on the device.
//
// What the host did sequentially is data-parallel here:
//   * discovery order: every candidate takes part in an atomicMin on (BASE + its index in the pick list) in an
//     open-addressing table keyed by global id; the candidate that owns the minimum is the first occurrence, an
//     exclusive scan over those flags is the discovery rank;
//   * the batch's GCN-normalised CSR pair (what gda_build_csr_norm would derive from the batch's edge list with two
//     radix sorts of E + N keys) falls out of the sampler's own structure: picks are already grouped by destination
//     in local-id order, so the by-destination CSR is one scan; the by-source CSR is ONE stable sort of the edges
//     by source.  Values: deg = row length (a sum of ones: exact), dis = 1/sqrt(deg), w = (dis[src]*1)*dis[dst] --
//     the arithmetic of csrc/gda_graph.hip, same bits.
// Sizes are data dependent: every kernel is launched on capacity-sized grids and reads the live counts from device
// memory, the caller reads {n_nodes, n_edges, nnz} back once per batch.
#include "gda_common.h"

#include <algorithm>

#include <hipcub/hipcub.hpp>

namespace {

constexpr int DS_TB = 256;
constexpr int DS_MAXK = 64;                 // largest positive fan-out (the sparse Fisher-Yates map is 2 * DS_MAXK entries)
constexpr int32_t DS_BASE = 1 << 30;        // table values >= DS_BASE: "first seen at pick index value - DS_BASE"
constexpr int32_t DS_EMPTY_VAL = 0x7F7F7F7F;

struct DsCnt {            // device-resident counters of the batch being built
    long long n_nodes, n_edges, fb, fe, m, nnz, overflow, nonloop;
};

struct SplitMix {
    uint64_t s;
    __device__ explicit SplitMix(uint64_t seed) : s(seed) {}
    __device__ uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    __device__ uint64_t below(uint64_t n) { return __umul64hi(next(), n); }
};

__device__ __forceinline__ uint32_t ds_hash(int32_t u, int bits) { return ((uint32_t)u * 0x9E3779B1u) >> (32 - bits); }

// ---- graph: in-neighbour lists ------------------------------------------------------------------------------
__global__ void k_ds_edges32(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int64_t E, int64_t N,
                             int32_t* __restrict__ s32, int32_t* __restrict__ d32, unsigned long long* __restrict__ deg,
                             int32_t* __restrict__ status) {
    const int64_t e = (int64_t)blockIdx.x * DS_TB + threadIdx.x;
    if (e >= E) return;
    const int64_t u = src[e], v = dst[e];
    if (u < 0 || u >= N || v < 0 || v >= N) { atomicAdd(&status[0], 1); s32[e] = 0; d32[e] = 0; return; }
    s32[e] = (int32_t)u; d32[e] = (int32_t)v;
    atomicAdd(&deg[v], 1ull);
}

__global__ void k_ds_maxdeg(const int64_t* __restrict__ in_ptr, int64_t N, int32_t* __restrict__ status) {
    const int64_t v = (int64_t)blockIdx.x * DS_TB + threadIdx.x;
    int d = 0;
    if (v < N) d = (int)min((long long)(in_ptr[v + 1] - in_ptr[v]), (long long)INT32_MAX);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) d = max(d, __shfl_down(d, off, 64));
    if ((threadIdx.x & 63) == 0 && d > 0) atomicMax(&status[1], d);
}

// ---- one batch ----------------------------------------------------------------------------------------------
__global__ void k_ds_init(const int64_t* __restrict__ seeds, int64_t n_seeds, int64_t N, int32_t* __restrict__ cand,
                          DsCnt* __restrict__ cnt) {
    const int64_t q = (int64_t)blockIdx.x * DS_TB + threadIdx.x;
    if (q == 0) { cnt->n_nodes = 0; cnt->n_edges = 0; cnt->fb = 0; cnt->fe = 0; cnt->m = n_seeds; cnt->nnz = 0; cnt->nonloop = 0; }
    if (q >= n_seeds) return;
    const int64_t v = seeds[q];
    if (v < 0 || v >= N) { cnt->overflow = 2; cand[q] = 0; return; }
    cand[q] = (int32_t)v;
}

// candidates cand[0 .. m): claim a table slot per distinct global id, remember the smallest pick index per id
__global__ void k_ds_insert(const int32_t* __restrict__ cand, const DsCnt* __restrict__ cnt, int32_t* __restrict__ keys,
                            int32_t* __restrict__ vals, int bits, int32_t* __restrict__ slot_of) {
    const int64_t q = (int64_t)blockIdx.x * DS_TB + threadIdx.x;
    if (q >= cnt->m) return;
    const int32_t u = cand[q];
    const uint32_t mask = (1u << bits) - 1u;
    uint32_t p = ds_hash(u, bits);
    while (true) {
        int32_t k = keys[p];
        if (k == -1) k = atomicCAS(&keys[p], -1, u);
        if (k == -1 || k == u) break;
        p = (p + 1) & mask;
    }
    slot_of[q] = (int32_t)p;
    atomicMin(&vals[p], DS_BASE + (int32_t)q);
}

__global__ void k_ds_flag(const DsCnt* __restrict__ cnt, const int32_t* __restrict__ vals, const int32_t* __restrict__ slot_of,
                          int64_t cap, int32_t* __restrict__ flag) {
    const int64_t q = (int64_t)blockIdx.x * DS_TB + threadIdx.x;
    if (q > cap) return;
    flag[q] = (q < cnt->m && vals[slot_of[q]] == DS_BASE + (int32_t)q) ? 1 : 0;
}

// local ids: an id seen before this hop is in the table, a new one gets n_nodes + (discovery rank)
__global__ void k_ds_relabel(const int32_t* __restrict__ cand, const DsCnt* cntp, const int32_t* __restrict__ vals,
                             const int32_t* __restrict__ slot_of, const int32_t* __restrict__ flag, const int32_t* __restrict__ rank,
                             const int32_t* __restrict__ dst_of, int with_edges, int64_t node_cap, int64_t edge_cap,
                             int64_t* __restrict__ nodes, int64_t* __restrict__ esrc, int64_t* __restrict__ edst,
                             DsCnt* cnt) {
    const int64_t q = (int64_t)blockIdx.x * DS_TB + threadIdx.x;
    if (q >= cntp->m) return;
    const int64_t n0 = cntp->n_nodes, e0 = cntp->n_edges;
    const int32_t v = vals[slot_of[q]];
    const int64_t lu = v >= DS_BASE ? n0 + rank[v - DS_BASE] : (int64_t)v;
    if (flag[q]) {
        if (lu < node_cap) nodes[lu] = cand[q]; else cnt->overflow = 1;
    }
    if (with_edges) {
        if (e0 + q < edge_cap) { esrc[e0 + q] = lu; edst[e0 + q] = dst_of[q]; } else cnt->overflow = 1;
    }
}

__global__ void k_ds_finalize(const DsCnt* __restrict__ cnt, int32_t* __restrict__ vals, const int32_t* __restrict__ slot_of,
                              const int32_t* __restrict__ flag, const int32_t* __restrict__ rank) {
    const int64_t q = (int64_t)blockIdx.x * DS_TB + threadIdx.x;
    if (q >= cnt->m || !flag[q]) return;
    vals[slot_of[q]] = (int32_t)(cnt->n_nodes + rank[q]);
}

__global__ void k_ds_advance(DsCnt* __restrict__ cnt, const int32_t* __restrict__ rank, int64_t cap, int with_edges,
                             int64_t node_cap, int64_t edge_cap) {
    long long n = cnt->n_nodes + rank[cap];
    if (n > node_cap) { n = node_cap; cnt->overflow = 1; }
    if (with_edges) {
        long long e = cnt->n_edges + cnt->m;
        if (e > edge_cap) { e = edge_cap; cnt->overflow = 1; }
        cnt->n_edges = e;
    }
    cnt->fb = cnt->fe;
    cnt->fe = n;
    cnt->n_nodes = n;
}

// how many neighbours each frontier node contributes
__global__ void k_ds_count(const int64_t* __restrict__ in_ptr, const int64_t* __restrict__ nodes, const DsCnt* __restrict__ cnt,
                           int k, int64_t fcap, int32_t* __restrict__ c) {
    const int64_t f = (int64_t)blockIdx.x * DS_TB + threadIdx.x;
    if (f > fcap) return;
    int32_t out = 0;
    if (f < cnt->fe - cnt->fb) {
        const int64_t v = nodes[cnt->fb + f];
        const int64_t deg = in_ptr[v + 1] - in_ptr[v];
        out = (int32_t)((k < 0 || deg <= k) ? deg : k);
    }
    c[f] = out;
}

__global__ void k_ds_setm(DsCnt* __restrict__ cnt, const int32_t* __restrict__ pick_off, int64_t fcap, int64_t pcap) {
    long long m = pick_off[fcap];
    if (m > pcap) { m = pcap; cnt->overflow = 1; }
    cnt->m = m;
}

// nodes with more in-neighbours than the fan-out: k distinct positions by the host sampler's partial Fisher-Yates
// (scratch[j] <-> scratch[j + below(deg - j)] for j < k over scratch = identity), then in list order.  The array is
// never materialised: positions < k live in a dense k-entry table, the touched positions >= k (at most k) in a small
// key / value list -- 3 k words per lane, in LDS (lane-interleaved: conflict-free), one wave per workgroup.
constexpr int DS_PICK_TB = 64;

__global__ void __launch_bounds__(DS_PICK_TB)
k_ds_pick_rng(const int64_t* __restrict__ in_ptr, const int32_t* __restrict__ in_src,
              const int64_t* __restrict__ nodes, const DsCnt* __restrict__ cnt, int k, int hop, uint64_t rng_seed,
              const int32_t* __restrict__ pick_off, int64_t pcap, int32_t* __restrict__ cand) {
    extern __shared__ int32_t ds_sm[];
    const int lane = threadIdx.x;
    const int64_t f = (int64_t)blockIdx.x * DS_PICK_TB + lane;
    if (f >= cnt->fe - cnt->fb) return;
    const int64_t v = nodes[cnt->fb + f];
    const int64_t b = in_ptr[v], deg = in_ptr[v + 1] - b;
    if (k < 0 || deg <= k) return;
    int32_t* dense = ds_sm + lane;                           // entry j at dense[j * 64]
    int32_t* skey = ds_sm + (size_t)k * DS_PICK_TB + lane;
    int32_t* sval = ds_sm + (size_t)2 * k * DS_PICK_TB + lane;
    for (int j = 0; j < k; ++j) dense[j * DS_PICK_TB] = j;
    int used = 0;
    SplitMix rng(rng_seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(hop + 1)) ^ (0x9E3779B97F4A7C15ull * (uint64_t)(v + 1)));
    for (int j = 0; j < k; ++j) {
        const int32_t r = (int32_t)(j + (int64_t)rng.below((uint64_t)(deg - j)));
        const int32_t aj = dense[j * DS_PICK_TB];
        int32_t ar;
        if (r < k) {
            ar = dense[r * DS_PICK_TB];
            dense[r * DS_PICK_TB] = aj;
        } else {
            int ir = -1;
            for (int t = 0; t < used; ++t) if (skey[t * DS_PICK_TB] == r) ir = t;
            if (ir >= 0) { ar = sval[ir * DS_PICK_TB]; sval[ir * DS_PICK_TB] = aj; }
            else { ar = r; skey[used * DS_PICK_TB] = r; sval[used * DS_PICK_TB] = aj; ++used; }
        }
        dense[j * DS_PICK_TB] = ar;
    }
    for (int j = 1; j < k; ++j) {                             // insertion sort: the kept edges stay in edge order
        const int32_t a = dense[j * DS_PICK_TB];
        int i = j;
        while (i > 0 && dense[(i - 1) * DS_PICK_TB] > a) { dense[i * DS_PICK_TB] = dense[(i - 1) * DS_PICK_TB]; --i; }
        dense[i * DS_PICK_TB] = a;
    }
    const int64_t o = pick_off[f];
    for (int j = 0; j < k; ++j)
        if (o + j < pcap) cand[o + j] = in_src[b + dense[j * DS_PICK_TB]];
}

// per pick: its frontier node (binary search in the offsets); nodes that keep all their neighbours copy them here
__global__ void k_ds_pick_copy(const int64_t* __restrict__ in_ptr, const int32_t* __restrict__ in_src,
                               const int64_t* __restrict__ nodes, const DsCnt* __restrict__ cnt, int k,
                               const int32_t* __restrict__ pick_off, int32_t* __restrict__ cand, int32_t* __restrict__ dst_of) {
    const int64_t q = (int64_t)blockIdx.x * DS_TB + threadIdx.x;
    if (q >= cnt->m) return;
    int64_t lo = 0, hi = cnt->fe - cnt->fb;          // the last f with pick_off[f] <= q
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (pick_off[mid] <= q) lo = mid; else hi = mid;
    }
    const int64_t f = lo;
    dst_of[q] = (int32_t)(cnt->fb + f);
    const int64_t v = nodes[cnt->fb + f];
    const int64_t b = in_ptr[v], deg = in_ptr[v + 1] - b;
    if (k < 0 || deg <= k) cand[q] = in_src[b + (q - pick_off[f])];
}

// ---- the batch's normalised CSR pair ---------------------------------------------------------------------------
__global__ void k_ds_csr_count(const int64_t* __restrict__ esrc, const int64_t* __restrict__ edst, const DsCnt* __restrict__ cnt,
                               int64_t ecap, int32_t* __restrict__ nl, int32_t* __restrict__ cntd, int32_t* __restrict__ cnts) {
    const int64_t e = (int64_t)blockIdx.x * DS_TB + threadIdx.x;
    if (e > ecap) return;
    int32_t f = 0;
    if (e < cnt->n_edges) {
        const int64_t u = esrc[e], v = edst[e];
        if (u != v) { f = 1; atomicAdd(&cntd[v], 1); atomicAdd(&cnts[u], 1); }     // add_remaining_self_loops: loops are re-appended
    }
    nl[e] = f;
}

__global__ void k_ds_rowlen(const DsCnt* __restrict__ cnt, const int32_t* __restrict__ cntd, const int32_t* __restrict__ cnts,
                            int64_t ncap, int32_t* __restrict__ rl, int32_t* __restrict__ trl, float* __restrict__ dis) {
    const int64_t i = (int64_t)blockIdx.x * DS_TB + threadIdx.x;
    if (i > ncap) return;
    const bool live = i < cnt->n_nodes;
    rl[i] = live ? cntd[i] + 1 : 0;
    trl[i] = live ? cnts[i] + 1 : 0;
    if (live) dis[i] = 1.0f / sqrtf((float)(cntd[i] + 1));       // the degree is a sum of ones: exact
}

__global__ void k_ds_fill_fwd(const int64_t* __restrict__ esrc, const int64_t* __restrict__ edst, DsCnt* __restrict__ cnt,
                              const int32_t* __restrict__ nl, const int32_t* __restrict__ g, const float* __restrict__ dis,
                              int64_t ecap, int64_t ncap, int32_t* __restrict__ colidx, float* __restrict__ val,
                              int32_t* __restrict__ skey, int32_t* __restrict__ sval) {
    const int64_t e = (int64_t)blockIdx.x * DS_TB + threadIdx.x;
    if (e >= ecap) return;
    if (e == 0) { cnt->nonloop = g[ecap]; cnt->nnz = g[ecap] + cnt->n_nodes; }
    int32_t key = (int32_t)ncap, v32 = 0;                         // beyond the live edges and loops: sorted to the end
    if (e < cnt->n_edges && nl[e]) {
        const int64_t u = esrc[e], v = edst[e];
        const int64_t p = g[e] + v;                               // row start = (kept edges into smaller ids) + (their loops)
        colidx[p] = (int32_t)u;
        val[p] = __fmul_rn(__fmul_rn(dis[u], 1.0f), dis[v]);
        key = (int32_t)u; v32 = (int32_t)v;
    }
    skey[e] = key; sval[e] = v32;
}

__global__ void k_ds_fill_loops(const DsCnt* __restrict__ cnt, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ t_rowptr,
                                const float* __restrict__ dis, int32_t* __restrict__ colidx, float* __restrict__ val,
                                int32_t* __restrict__ t_colidx, float* __restrict__ t_val) {
    const int64_t i = (int64_t)blockIdx.x * DS_TB + threadIdx.x;
    if (i >= cnt->n_nodes) return;
    const float w = __fmul_rn(__fmul_rn(dis[i], 1.0f), dis[i]);
    const int32_t p = rowptr[i + 1] - 1, tp = t_rowptr[i + 1] - 1;       // the appended loops come last in their rows
    colidx[p] = (int32_t)i; val[p] = w;
    t_colidx[tp] = (int32_t)i; t_val[tp] = w;
}

__global__ void k_ds_fill_t(const DsCnt* __restrict__ cnt, const int32_t* __restrict__ skey, const int32_t* __restrict__ sval,
                            const float* __restrict__ dis, int32_t* __restrict__ t_colidx, float* __restrict__ t_val) {
    const int64_t p = (int64_t)blockIdx.x * DS_TB + threadIdx.x;
    if (p >= cnt->nonloop) return;
    const int32_t u = skey[p], v = sval[p];
    t_colidx[p + u] = v;
    t_val[p + u] = __fmul_rn(__fmul_rn(dis[u], 1.0f), dis[v]);
}

__global__ void k_ds_counts_out(const DsCnt* __restrict__ cnt, int64_t* __restrict__ out) {
    out[0] = cnt->n_nodes; out[1] = cnt->n_edges; out[2] = cnt->nnz; out[3] = cnt->overflow;
    out[4] = cnt->fb;          // nodes before the last hop's discoveries: rows [fb, n_nodes) are never expanded
}

inline unsigned ds_grid(int64_t n) { return (unsigned)gda_cdiv(n > 0 ? n : 1, DS_TB); }

struct DsWs {
    DsCnt* cnt;
    int32_t *keys, *vals;                                  // [1 << bits]
    int32_t *c, *pick_off;                                 // [fcap + 1]
    int32_t *cand, *dst_of, *slot_of, *flag, *rank;        // [pcap + 1]
    int32_t *cntd, *cnts, *rl, *trl;                       // [ncap + 1]
    float* dis;                                            // [ncap]
    int32_t *nl, *g, *skey, *sval, *skey_out, *sval_out;   // [ecap + 1]
    void* cub;
    size_t cub_bytes, total;
    int bits;
};

DsWs ds_carve(void* base, int64_t fcap, int64_t pcap, int64_t ncap, int64_t ecap) {
    DsWs w{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void* p = base ? (void*)((char*)base + off) : nullptr;
        off += gda_align_up(bytes, 256);
        return p;
    };
    int bits = 10;
    while (((int64_t)1 << bits) < 2 * ncap && bits < 31) ++bits;
    w.bits = bits;
    w.cnt = (DsCnt*)take(sizeof(DsCnt));
    w.keys = (int32_t*)take(sizeof(int32_t) << bits);
    w.vals = (int32_t*)take(sizeof(int32_t) << bits);
    w.c = (int32_t*)take(4 * (size_t)(fcap + 1));
    w.pick_off = (int32_t*)take(4 * (size_t)(fcap + 1));
    w.cand = (int32_t*)take(4 * (size_t)(pcap + 1));
    w.dst_of = (int32_t*)take(4 * (size_t)(pcap + 1));
    w.slot_of = (int32_t*)take(4 * (size_t)(pcap + 1));
    w.flag = (int32_t*)take(4 * (size_t)(pcap + 1));
    w.rank = (int32_t*)take(4 * (size_t)(pcap + 1));
    w.cntd = (int32_t*)take(4 * (size_t)(ncap + 1));
    w.cnts = (int32_t*)take(4 * (size_t)(ncap + 1));
    w.rl = (int32_t*)take(4 * (size_t)(ncap + 1));
    w.trl = (int32_t*)take(4 * (size_t)(ncap + 1));
    w.dis = (float*)take(4 * (size_t)(ncap + 1));
    w.nl = (int32_t*)take(4 * (size_t)(ecap + 1));
    w.g = (int32_t*)take(4 * (size_t)(ecap + 1));
    w.skey = (int32_t*)take(4 * (size_t)(ecap + 1));
    w.sval = (int32_t*)take(4 * (size_t)(ecap + 1));
    w.skey_out = (int32_t*)take(4 * (size_t)(ecap + 1));
    w.sval_out = (int32_t*)take(4 * (size_t)(ecap + 1));
    size_t a = 0, b = 0;
    int32_t* p = nullptr;
    const int64_t big = std::max(std::max(fcap, pcap), std::max(ncap, ecap)) + 1;
    hipcub::DeviceScan::ExclusiveSum(nullptr, a, p, p, (int)big);
    hipcub::DeviceRadixSort::SortPairs(nullptr, b, p, p, p, p, (int)(ecap > 0 ? ecap : 1), 0, 32);
    w.cub_bytes = std::max(a, b);
    w.cub = take(w.cub_bytes);
    w.total = off;
    return w;
}

// per-hop capacities: frontier, picks
struct DsCaps { int64_t fcap, pcap, ncap, ecap; };

bool ds_caps(int64_t n_seeds, const int32_t* fan, int L, int64_t max_deg, int64_t E, int64_t N, DsCaps* out) {
    int64_t frontier = std::min(n_seeds, N), nodes = frontier, edges = 0, fmax = std::max<int64_t>(n_seeds, 1), pmax = std::max<int64_t>(n_seeds, 1);
    for (int h = 0; h < L; ++h) {
        const int64_t k = fan[h];
        if (k == 0 || k > DS_MAXK) return false;
        const int64_t per = k < 0 ? max_deg : std::min<int64_t>(k, max_deg);
        int64_t picks = frontier * per;                                  // frontier <= N < 2^31, per < 2^31: no overflow
        if (picks > E) picks = E;                                        // in-lists of distinct nodes are disjoint
        fmax = std::max(fmax, frontier);
        pmax = std::max(pmax, picks);
        edges += picks;
        frontier = std::min(picks, N);
        nodes = std::min(nodes + picks, N);
    }
    if (pmax >= DS_BASE || nodes >= DS_BASE || edges >= INT32_MAX - nodes) return false;
    *out = DsCaps{fmax, pmax, std::max<int64_t>(nodes, 1), std::max<int64_t>(edges, 1)};
    return true;
}

}  // namespace

extern "C" size_t gda_dsampler_graph_workspace_bytes(int64_t E, int64_t N) {
    if (E < 0 || N < 0 || E >= INT32_MAX || N >= INT32_MAX) return 0;
    size_t a = 0, b = 0;
    int32_t* p = nullptr;
    unsigned long long* q = nullptr;
    hipcub::DeviceRadixSort::SortPairs(nullptr, a, p, p, p, p, (int)(E > 0 ? E : 1), 0, 32);
    hipcub::DeviceScan::ExclusiveSum(nullptr, b, q, q, (int)(N + 1));
    return gda_align_up(std::max(a, b), 256) + 3 * gda_align_up(4 * (size_t)(E + 1), 256) + gda_align_up(8 * (size_t)(N + 1), 256);
}

// In-neighbour lists of a graph on the device: in_src[in_ptr[v] .. in_ptr[v+1]) = sources of the edges into v, in
// edge order (stable sort by destination) -- the layout gda_sampler_create builds on the host.
// status (device int32[2]): [0] = number of edges with an endpoint outside [0, N) (must be 0), [1] = largest in-degree.
extern "C" int gda_dsampler_build_graph(const int64_t* src, const int64_t* dst, int64_t E, int64_t N,
                                        int64_t* in_ptr, int32_t* in_src, int32_t* status,
                                        void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    if (E < 0 || N < 0 || E >= INT32_MAX || N >= INT32_MAX) return GDA_E_SIZE;
    if (!in_ptr || !status || !workspace || (E > 0 && (!src || !dst || !in_src))) return GDA_E_NULL;
    if (workspace_bytes < gda_dsampler_graph_workspace_bytes(E, N)) return GDA_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    size_t a = 0, b = 0;
    int32_t* p = nullptr;
    unsigned long long* q = nullptr;
    hipcub::DeviceRadixSort::SortPairs(nullptr, a, p, p, p, p, (int)(E > 0 ? E : 1), 0, 32);
    hipcub::DeviceScan::ExclusiveSum(nullptr, b, q, q, (int)(N + 1));
    size_t cub_bytes = std::max(a, b);
    char* base = (char*)workspace;
    void* cub = base; base += gda_align_up(cub_bytes, 256);
    int32_t* s32 = (int32_t*)base; base += gda_align_up(4 * (size_t)(E + 1), 256);
    int32_t* d32 = (int32_t*)base; base += gda_align_up(4 * (size_t)(E + 1), 256);
    int32_t* d32o = (int32_t*)base; base += gda_align_up(4 * (size_t)(E + 1), 256);
    unsigned long long* deg = (unsigned long long*)base;
    GDA_HIP_TRY(hipMemsetAsync(status, 0, 8, stream));
    GDA_HIP_TRY(hipMemsetAsync(deg, 0, 8 * (size_t)(N + 1), stream));
    if (E > 0) {
        k_ds_edges32<<<ds_grid(E), DS_TB, 0, stream>>>(src, dst, E, N, s32, d32, deg, status);
        GDA_LAUNCH_CHECK();
    }
    int bits = 1;
    while (((int64_t)1 << bits) < N && bits < 31) ++bits;
    GDA_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(cub, cub_bytes, deg, (unsigned long long*)in_ptr, (int)(N + 1), stream));
    if (E > 0)
        GDA_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(cub, cub_bytes, d32, d32o, s32, in_src, (int)E, 0, bits, stream));
    if (N > 0) {
        k_ds_maxdeg<<<ds_grid(N), DS_TB, 0, stream>>>(in_ptr, N, status);
        GDA_LAUNCH_CHECK();
    }
    return GDA_OK;
}

// Capacity of a batch (every array of gda_dsampler_sample is sized by these): nodes <= node_cap, edges <= edge_cap.
// Returns GDA_E_UNSUPPORTED for a fan-out of 0 or beyond 64, or a batch beyond the int32 range.
extern "C" int gda_dsampler_caps(int64_t n_seeds, const int32_t* fanouts_host, int L, int64_t max_in_degree, int64_t E,
                                 int64_t N, int64_t* node_cap, int64_t* edge_cap) {
    if (!node_cap || !edge_cap || (L > 0 && !fanouts_host)) return GDA_E_NULL;
    if (n_seeds < 0 || L < 0 || max_in_degree < 0 || E < 0 || N < 0) return GDA_E_SIZE;
    DsCaps c;
    if (!ds_caps(n_seeds, fanouts_host, L, max_in_degree, E, N, &c)) return GDA_E_UNSUPPORTED;
    *node_cap = c.ncap; *edge_cap = c.ecap;
    return GDA_OK;
}

extern "C" size_t gda_dsampler_workspace_bytes(int64_t n_seeds, const int32_t* fanouts_host, int L, int64_t max_in_degree,
                                               int64_t E, int64_t N) {
    DsCaps c;
    if (n_seeds < 0 || L < 0 || (L > 0 && !fanouts_host) || !ds_caps(n_seeds, fanouts_host, L, max_in_degree, E, N, &c)) return 0;
    return ds_carve(nullptr, c.fcap, c.pcap, c.ncap, c.ecap).total;
}

// One batch.  Device arrays: nodes [node_cap] int64 (global ids, seeds first), esrc / edst [edge_cap] int64 (local
// ids), the CSR pair in the capacity layout of gda_build_csr_norm for N = node_cap, E = edge_cap (rowptr [node_cap+1],
// colidx / val [edge_cap + node_cap]; all six NULL: no CSR), counts int64[5] = {n_nodes, n_edges, nnz, status,
// n_interior} (status 0 = ok, 1 = a capacity was exceeded, 2 = a seed outside [0, N); n_interior = nodes before the last
// hop's discoveries: the rows from there on hold their self loop only).  rowptr[i] for i >= n_nodes equals nnz.
extern "C" int gda_dsampler_sample(const int64_t* in_ptr, const int32_t* in_src, int64_t N, int64_t E, int64_t max_in_degree,
                                   const int64_t* seeds, int64_t n_seeds, const int32_t* fanouts_host, int L, uint64_t rng_seed,
                                   int64_t* nodes, int64_t* esrc, int64_t* edst,
                                   int32_t* rowptr, int32_t* colidx, float* val,
                                   int32_t* t_rowptr, int32_t* t_colidx, float* t_val,
                                   int64_t* counts, void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    if (!in_ptr || !nodes || !counts || !workspace || (n_seeds > 0 && !seeds) || (L > 0 && !fanouts_host)) return GDA_E_NULL;
    if (N < 0 || E < 0 || n_seeds < 0 || L < 0) return GDA_E_SIZE;
    DsCaps cp;
    if (!ds_caps(n_seeds, fanouts_host, L, max_in_degree, E, N, &cp)) return GDA_E_UNSUPPORTED;
    if (L > 0 && E > 0 && (!in_src || !esrc || !edst)) return GDA_E_NULL;
    const bool csr = rowptr != nullptr;
    if (csr && (!colidx || !val || !t_rowptr || !t_colidx || !t_val)) return GDA_E_NULL;
    DsWs w = ds_carve(workspace, cp.fcap, cp.pcap, cp.ncap, cp.ecap);
    if (workspace_bytes < w.total) return GDA_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream_;
    GDA_HIP_TRY(hipMemsetAsync(w.cnt, 0, sizeof(DsCnt), s));
    GDA_HIP_TRY(hipMemsetAsync(w.keys, 0xFF, sizeof(int32_t) << w.bits, s));
    GDA_HIP_TRY(hipMemsetAsync(w.vals, 0x7F, sizeof(int32_t) << w.bits, s));
    static_assert(DS_EMPTY_VAL == 0x7F7F7F7F, "memset pattern");
    k_ds_init<<<ds_grid(n_seeds), DS_TB, 0, s>>>(seeds, n_seeds, N, w.cand, w.cnt);
    GDA_LAUNCH_CHECK();
    // the seeds, then every hop: candidates -> first occurrences -> discovery ranks -> local ids
    auto discover = [&](int64_t cap, int with_edges) -> int {
        k_ds_insert<<<ds_grid(cap), DS_TB, 0, s>>>(w.cand, w.cnt, w.keys, w.vals, w.bits, w.slot_of);
        k_ds_flag<<<ds_grid(cap + 1), DS_TB, 0, s>>>(w.cnt, w.vals, w.slot_of, cap, w.flag);
        GDA_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(w.cub, w.cub_bytes, w.flag, w.rank, (int)(cap + 1), s));
        k_ds_relabel<<<ds_grid(cap), DS_TB, 0, s>>>(w.cand, w.cnt, w.vals, w.slot_of, w.flag, w.rank, w.dst_of, with_edges,
                                                    cp.ncap, cp.ecap, nodes, esrc, edst, w.cnt);
        k_ds_finalize<<<ds_grid(cap), DS_TB, 0, s>>>(w.cnt, w.vals, w.slot_of, w.flag, w.rank);
        k_ds_advance<<<1, 1, 0, s>>>(w.cnt, w.rank, cap, with_edges, cp.ncap, cp.ecap);
        GDA_LAUNCH_CHECK();
        return GDA_OK;
    };
    int st = discover(std::max<int64_t>(n_seeds, 1), 0);
    if (st != GDA_OK) return st;
    int64_t frontier = std::min(n_seeds, N);
    for (int hop = 0; hop < L; ++hop) {
        const int k = fanouts_host[hop];
        const int64_t per = k < 0 ? max_in_degree : std::min<int64_t>(k, max_in_degree);
        const int64_t fcap = std::max<int64_t>(frontier, 1);
        const int64_t pcap = std::max<int64_t>(std::min(frontier * per, E), 1);
        k_ds_count<<<ds_grid(fcap + 1), DS_TB, 0, s>>>(in_ptr, nodes, w.cnt, k, fcap, w.c);
        GDA_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(w.cub, w.cub_bytes, w.c, w.pick_off, (int)(fcap + 1), s));
        k_ds_setm<<<1, 1, 0, s>>>(w.cnt, w.pick_off, fcap, pcap);
        if (k >= 0)
            k_ds_pick_rng<<<(unsigned)gda_cdiv(fcap, DS_PICK_TB), DS_PICK_TB, (size_t)3 * k * DS_PICK_TB * sizeof(int32_t), s>>>(
                in_ptr, in_src, nodes, w.cnt, k, hop, rng_seed, w.pick_off, pcap, w.cand);
        k_ds_pick_copy<<<ds_grid(pcap), DS_TB, 0, s>>>(in_ptr, in_src, nodes, w.cnt, k, w.pick_off, w.cand, w.dst_of);
        GDA_LAUNCH_CHECK();
        st = discover(pcap, 1);
        if (st != GDA_OK) return st;
        frontier = std::min(std::min(frontier * per, E), N);
    }
    if (csr) {
        const int64_t nc = cp.ncap, ec = cp.ecap;
        GDA_HIP_TRY(hipMemsetAsync(w.cntd, 0, 4 * (size_t)(nc + 1), s));
        GDA_HIP_TRY(hipMemsetAsync(w.cnts, 0, 4 * (size_t)(nc + 1), s));
        k_ds_csr_count<<<ds_grid(ec + 1), DS_TB, 0, s>>>(esrc, edst, w.cnt, ec, w.nl, w.cntd, w.cnts);
        k_ds_rowlen<<<ds_grid(nc + 1), DS_TB, 0, s>>>(w.cnt, w.cntd, w.cnts, nc, w.rl, w.trl, w.dis);
        GDA_LAUNCH_CHECK();
        GDA_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(w.cub, w.cub_bytes, w.rl, rowptr, (int)(nc + 1), s));
        GDA_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(w.cub, w.cub_bytes, w.trl, t_rowptr, (int)(nc + 1), s));
        GDA_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(w.cub, w.cub_bytes, w.nl, w.g, (int)(ec + 1), s));
        k_ds_fill_fwd<<<ds_grid(ec), DS_TB, 0, s>>>(esrc, edst, w.cnt, w.nl, w.g, w.dis, ec, nc, colidx, val, w.skey, w.sval);
        GDA_LAUNCH_CHECK();
        int bits = 1;
        while (((int64_t)1 << bits) <= nc && bits < 31) ++bits;        // keys 0 .. nc (nc = "not an edge")
        GDA_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(w.cub, w.cub_bytes, w.skey, w.skey_out, w.sval, w.sval_out, (int)ec, 0, bits, s));
        k_ds_fill_loops<<<ds_grid(nc), DS_TB, 0, s>>>(w.cnt, rowptr, t_rowptr, w.dis, colidx, val, t_colidx, t_val);
        k_ds_fill_t<<<ds_grid(ec), DS_TB, 0, s>>>(w.cnt, w.skey_out, w.sval_out, w.dis, t_colidx, t_val);
        GDA_LAUNCH_CHECK();
    }
    k_ds_counts_out<<<1, 1, 0, s>>>(w.cnt, counts);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

// ---- one call per batch for a loader that recycles its batch blocks ------------------------------------------------
// Events the loader's producer and consumer threads order themselves with (plain HIP events, no timing), owned by the
// library so that a batch costs the producer thread ONE foreign call: no tensor views, no allocator, no event objects
// under the interpreter lock the training thread is waiting for.
extern "C" int gda_event_create(void** event_out) {
    if (!event_out) return GDA_E_NULL;
    hipEvent_t ev = nullptr;
    GDA_HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    *event_out = (void*)ev;
    return GDA_OK;
}
extern "C" int gda_event_destroy(void* event) {
    if (!event) return GDA_OK;
    GDA_HIP_TRY(hipEventDestroy((hipEvent_t)event));
    return GDA_OK;
}
extern "C" int gda_event_record(void* event, gda_stream_t stream) {
    if (!event) return GDA_E_NULL;
    GDA_HIP_TRY(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return GDA_OK;
}
extern "C" int gda_event_synchronize(void* event) {
    if (!event) return GDA_E_NULL;
    GDA_HIP_TRY(hipEventSynchronize((hipEvent_t)event));
    return GDA_OK;
}
extern "C" int gda_stream_wait_event(gda_stream_t stream, void* event) {
    if (!event) return GDA_E_NULL;
    GDA_HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return GDA_OK;
}

// gda_dsampler_sample + the two interior K-step plans + the counts' trip home, as ONE call on `stream`:
//   [wait_event] -> seeds (host or device, n_seeds int64) -> seeds_dev -> the batch -> plan_fwd / plan_bwd (both or
//   neither; counts[5:7] / counts[7:9] <- their {q, T}) -> counts (device int64[12], zeroed first) -> counts_host (pinned
//   int64[12]) -> [done_event].  The arrays may be a block the caller overwrote before: wait_event is what makes that safe.
namespace {
// counts[4] (n_interior) <- min(max(n_interior, rows), n_nodes): a batch whose consumer runs at STATIC shapes (the captured
// sampled step) declares a fixed number of leading rows "interior".  Rows between the sampled n_interior and `rows` are
// last-hop discoveries -- their only entry is the unit self loop -- so every interior-rows kernel computes the same
// values for them (1 * x + 0 + 0) that the leaf copy would have produced.
__global__ void k_ds_round_interior(int64_t* counts, int64_t rows) {
    const int64_t n = counts[0], ni = counts[4];
    if (counts[3] == 0 && rows > ni) counts[4] = rows < n ? rows : n;
}
}  // namespace

extern "C" int gda_dsampler_batch_ex(const int64_t* in_ptr, const int32_t* in_src, int64_t N, int64_t E, int64_t max_in_degree,
                                     const int64_t* seeds, int64_t n_seeds, int64_t* seeds_dev,
                                     const int32_t* fanouts_host, int L, uint64_t rng_seed,
                                     int64_t* nodes, int64_t* esrc, int64_t* edst,
                                     int32_t* rowptr, int32_t* colidx, float* val,
                                     int32_t* t_rowptr, int32_t* t_colidx, float* t_val,
                                     int64_t* counts, void* plan_fwd, void* plan_bwd, size_t plan_bytes,
                                     int64_t* counts_host, void* wait_event, void* done_event, int64_t interior_rows,
                                     void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    if (!counts || !counts_host || (n_seeds > 0 && (!seeds || !seeds_dev))) return GDA_E_NULL;
    if ((plan_fwd == nullptr) != (plan_bwd == nullptr)) return GDA_E_NULL;
    if (plan_fwd && !rowptr) return GDA_E_NULL;
    if (n_seeds < 0) return GDA_E_SIZE;
    hipStream_t s = (hipStream_t)stream_;
    if (wait_event) GDA_HIP_TRY(hipStreamWaitEvent(s, (hipEvent_t)wait_event, 0));
    if (n_seeds > 0) GDA_HIP_TRY(hipMemcpyAsync(seeds_dev, seeds, sizeof(int64_t) * (size_t)n_seeds, hipMemcpyDefault, s));
    GDA_HIP_TRY(hipMemsetAsync(counts, 0, 12 * sizeof(int64_t), s));
    int st = gda_dsampler_sample(in_ptr, in_src, N, E, max_in_degree, seeds_dev, n_seeds, fanouts_host, L, rng_seed, nodes, esrc,
                                 edst, rowptr, colidx, val, t_rowptr, t_colidx, t_val, counts, workspace, workspace_bytes, stream_);
    if (st != GDA_OK) return st;
    if (interior_rows > 0) {
        k_ds_round_interior<<<1, 1, 0, s>>>(counts, interior_rows);
        GDA_LAUNCH_CHECK();
    }
    if (plan_fwd) {
        st = gda_interior_plan_build(rowptr, colidx, val, counts + 4, plan_fwd, plan_bytes, counts + 5, stream_);
        if (st != GDA_OK) return st;
        st = gda_interior_plan_build(t_rowptr, t_colidx, t_val, counts + 4, plan_bwd, plan_bytes, counts + 7, stream_);
        if (st != GDA_OK) return st;
    }
    GDA_HIP_TRY(hipMemcpyAsync(counts_host, counts, 12 * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    if (done_event) GDA_HIP_TRY(hipEventRecord((hipEvent_t)done_event, s));
    return GDA_OK;
}

extern "C" int gda_dsampler_batch(const int64_t* in_ptr, const int32_t* in_src, int64_t N, int64_t E, int64_t max_in_degree,
                                  const int64_t* seeds, int64_t n_seeds, int64_t* seeds_dev,
                                  const int32_t* fanouts_host, int L, uint64_t rng_seed,
                                  int64_t* nodes, int64_t* esrc, int64_t* edst,
                                  int32_t* rowptr, int32_t* colidx, float* val,
                                  int32_t* t_rowptr, int32_t* t_colidx, float* t_val,
                                  int64_t* counts, void* plan_fwd, void* plan_bwd, size_t plan_bytes,
                                  int64_t* counts_host, void* wait_event, void* done_event,
                                  void* workspace, size_t workspace_bytes, gda_stream_t stream_) {
    return gda_dsampler_batch_ex(in_ptr, in_src, N, E, max_in_degree, seeds, n_seeds, seeds_dev, fanouts_host, L, rng_seed, nodes,
                                 esrc, edst, rowptr, colidx, val, t_rowptr, t_colidx, t_val, counts, plan_fwd, plan_bwd, plan_bytes,
                                 counts_host, wait_event, done_event, 0, workspace, workspace_bytes, stream_);
}
