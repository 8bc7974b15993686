// Neighbour aggregation y = A_hat * x as a CSR SpMM for gfx950 (fp32).
//
// Replaces PyG MessagePassing.propagate + message + scatter-add aggregate as pygda
// calls it (pygda/nn/prop_gcn_conv.py:208-210,238; pygda/nn/cached_gcn_conv.py:138,156).
//
// Mapping.  A wavefront (64 lanes) is cut into lane groups of G lanes; one group owns
// one destination row and walks its neighbour list sequentially, so the per-element
// accumulation order is the CSR (= original edge) order and multiply and add are
// rounded separately -- the CPU gather/scale/scatter-add result bit for bit.  Each
// lane owns VEC consecutive feature columns (16-byte loads when the width allows),
// so a group reads one neighbour's feature row as one fully coalesced
// G*VEC*4-byte segment (512 B for the hid=128 layers: G=32, VEC=4 -> two rows per wave).
// Neighbour (col, val) pairs are fetched G at a time, one pair per lane (coalesced),
// and broadcast inside the group with DPP/bpermute shuffles; the gather loads of four
// neighbours are issued back to back before the dependent adds so four 512 B
// requests per group are in flight.
// HBM bound: algorithmic bytes per call = nnz*8 + (N+1)*4 + 2*N*d*4.
#include "gda_common.h"
#include "gda_philox.h"

#include <cstdlib>

namespace {

constexpr int TB = 256;           // 4 wavefronts per workgroup
constexpr int UNROLL = 4;

template <int VEC> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<2> { using type = float2; };
template <> struct VecT<4> { using type = float4; };

template <int VEC>
__device__ __forceinline__ void vload(float (&r)[VEC], const float* __restrict__ p) {
    using V = typename VecT<VEC>::type;
    const V v = *reinterpret_cast<const V*>(p);
    if constexpr (VEC == 1) { r[0] = v; }
    if constexpr (VEC == 2) { r[0] = v.x; r[1] = v.y; }
    if constexpr (VEC == 4) { r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w; }
}

// the same with the non-temporal hint (streamed once: do not keep the line in L2); the builtins take native vectors
template <int VEC> struct NatT { using type = float __attribute__((ext_vector_type(VEC))); };
template <> struct NatT<1> { using type = float; };

template <int VEC>
__device__ __forceinline__ void vload_nt(float (&r)[VEC], const float* __restrict__ p) {
    using V = typename NatT<VEC>::type;
    const V v = __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
    if constexpr (VEC == 1) { r[0] = v; }
    else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) r[k] = v[k];
    }
}

template <int VEC>
__device__ __forceinline__ void vstore_nt(float* __restrict__ p, const float (&r)[VEC]) {
    using V = typename NatT<VEC>::type;
    V v;
    if constexpr (VEC == 1) { v = r[0]; }
    else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[k] = r[k];
    }
    __builtin_nontemporal_store(v, reinterpret_cast<V*>(p));
}

template <int VEC>
__device__ __forceinline__ void vstore(float* __restrict__ p, const float (&r)[VEC]) {
    using V = typename VecT<VEC>::type;
    V v;
    if constexpr (VEC == 1) { v = r[0]; }
    if constexpr (VEC == 2) { v.x = r[0]; v.y = r[1]; }
    if constexpr (VEC == 4) { v.x = r[0]; v.y = r[1]; v.z = r[2]; v.w = r[3]; }
    *reinterpret_cast<V*>(p) = v;
}

// G lanes per row, VEC columns per lane.  Columns beyond G*VEC are covered by an outer
// chunk loop (re-walking the neighbour list), so any d works with any (G, VEC) whose
// alignment holds.
// Long rows (power-law hubs).  A row with more than `threshold` entries is cut into chunks of
// `threshold` entries; each chunk is a unit of work like a row (same lane-group code path) whose
// partial sum goes to scratch[chunk][d]; k_long_reduce then adds a row's chunks in order.  The
// row blocks and the chunk blocks share ONE launch (blockIdx ranges), so a graph without long
// rows pays nothing and a hub no longer serialises 10^5 gathers on one lane group.  Fixed
// chunking + ordered reduce: deterministic; only split rows lose the bit-for-bit CPU order.
struct RowSplit {
    int32_t threshold;              // 0: no splitting
    int32_t n_long, n_chunks;
    const int32_t* long_rows;       // [n_long]     row id of each long row
    const int32_t* long_chunk_ptr;  // [n_long + 1] chunk range of each long row
    const int32_t* chunk_long;      // [n_chunks]   index into long_rows
    float* scratch;                 // [n_chunks, d]
    const int32_t* counts;          // device {n_long, n_chunks} when the host never read them back:
                                    // n_long / n_chunks above are then CAPACITIES (grid bounds only)
};

// Optional epilogue (polynomial graph filters, e.g. the Bernstein filter of DGSDA):
//     y[i] = alpha * x[i] + beta * (A x)[i] + gamma * z[i],   gamma = gamma_imm * (gamma_dev ? *gamma_dev : 1)
// so I +- A_hat steps and Horner accumulations cost no extra pass over [N, d].
struct Epilogue {
    float alpha, beta, gamma_imm;
    const float* gamma_dev;         // device scalar (a learnable filter coefficient) or NULL
    const float* z;                 // [n_rows, d] (ld = ldz) or NULL
    int64_t ldz;
    int32_t hub_rows;               // HUB kernels: source rows below it are kept in L2, everything else is streamed
};

template <int VEC>
__device__ __forceinline__ void apply_epilogue(float (&acc)[VEC], const Epilogue& ep, const float* __restrict__ x,
                                               int64_t ldx, int64_t row, int c) {
    float xs[VEC];
    vload<VEC>(xs, x + row * ldx + c);
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = ep.alpha * xs[v] + ep.beta * acc[v];
    if (ep.z) {
        const float g = ep.gamma_dev ? ep.gamma_imm * *ep.gamma_dev : ep.gamma_imm;
        float zs[VEC];
        vload<VEC>(zs, ep.z + row * ep.ldz + c);
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += g * zs[v];
    }
}

// STG: the (col, val) tile of a lane group is staged in LDS (one ds_write_b64 per lane, one ds_read_b64 per neighbour,
// a broadcast inside the group) instead of being broadcast with two cross-lane shuffles per neighbour -- the
// "LDS staging of per-wavefront neighbour tiles" of the path's specification.  Same values in the same order.
// Measured against the shuffle form (tools/spmm_sweep.py with PYGDA_AMD_SPMM_LDS=1, profiles/r3_spmm_lds_vs_shuffle.jsonl).
// TOUT: y is written TRANSPOSED, yT[c * ldy + row] (ldy >= n_rows) -- the column-major hand-over to the LDS-resident
// K-step kernel (gda_kstep.hip) for a projection of sparse input features: no transpose launch between the two.
// HUB (opt-in measurement variant; degree-ordered graphs: data.auto_reorder): the hub rows'
// gathers -- source row below ep.hub_rows, a prefix that fits one XCD's L2 -- are ordinary loads, every other byte the
// kernel touches (the neighbour lists, the tail rows' gathers, the result) carries the non-temporal hint, so that the
// once-used tail does not push the hubs out of the L2 (R-MAT 2^22: a third of all edge endpoints lie in the first 8,192
// rows).  What it buys is in the comment at hub_rows() below: not enough to be the default.
template <int G, int VEC, bool EPI, bool STG = false, bool TOUT = false, bool HUB = false>
__global__ void __launch_bounds__(TB)
k_spmm(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
       const float* __restrict__ val, int64_t n_rows, int d,
       const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy,
       const float* __restrict__ bias, RowSplit sp, unsigned row_blocks, Epilogue ep) {
    constexpr int ROWS_PER_BLOCK = TB / G;
    const int lane_in_group = threadIdx.x % G;
    __shared__ int2 stg_tile[STG ? TB : 1];                // STG: one (col, val bits) pair per lane
    const int stg_base = threadIdx.x - lane_in_group;       // the group's first slot (groups never straddle a wave)
    const bool chunk_mode = blockIdx.x >= row_blocks;       // block-uniform
    int64_t row;                                             // output row (or chunk id in chunk mode)
    bool live;
    int32_t start = 0, end = 0;
    if (!chunk_mode) {
        row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + threadIdx.x / G;
        live = row < n_rows;
        // keep whole groups together for the shuffles; dead groups just run empty loops
        if (live) {
            start = rowptr[row];
            end = rowptr[row + 1];
            if (sp.threshold > 0 && end - start > sp.threshold) { live = false; end = start; }   // done by chunks
        }
    } else {
        row = (int64_t)(blockIdx.x - row_blocks) * ROWS_PER_BLOCK + threadIdx.x / G;
        live = row < (sp.counts ? sp.counts[1] : sp.n_chunks);
        if (live) {
            const int32_t li = sp.chunk_long[row];
            const int32_t r = sp.long_rows[li];
            start = rowptr[r] + (int32_t)(row - sp.long_chunk_ptr[li]) * sp.threshold;
            end = min(start + sp.threshold, rowptr[r + 1]);
        }
        y = sp.scratch;                                      // partial sums, no bias
        ldy = d;
        bias = nullptr;
    }

    for (int c0 = 0; c0 < d; c0 += G * VEC) {
        const int c = c0 + lane_in_group * VEC;
        const bool col_ok = c < d;                    // d % VEC == 0 is guaranteed by dispatch
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.0f;

        for (int32_t base = start; base < end; base += G) {
            // one (col, val) pair per lane: a coalesced G*8-byte tile of the neighbour list
            const int32_t k = base + lane_in_group;
            const int32_t my_col = k < end ? (HUB ? __builtin_nontemporal_load(colidx + k) : colidx[k]) : 0;
            const float my_val = k < end ? (HUB ? __builtin_nontemporal_load(val + k) : val[k]) : 0.0f;
            const int cnt = min((int32_t)G, end - base);
            if constexpr (STG) stg_tile[threadIdx.x] = make_int2(my_col, __float_as_int(my_val));   // same wave reads it back
            int e = 0;
            for (; e + UNROLL <= cnt; e += UNROLL) {
                float xv[UNROLL][VEC];
                float w[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    int32_t cu;
                    if constexpr (STG) { const int2 p = stg_tile[stg_base + e + u]; cu = p.x; w[u] = __int_as_float(p.y); }
                    else { cu = __shfl(my_col, e + u, G); w[u] = __shfl(my_val, e + u, G); }
                    if constexpr (HUB) {
                        if (col_ok) { if (cu < ep.hub_rows) vload<VEC>(xv[u], x + (int64_t)cu * ldx + c); else vload_nt<VEC>(xv[u], x + (int64_t)cu * ldx + c); }
                    } else {
                        if (col_ok) vload<VEC>(xv[u], x + (int64_t)cu * ldx + c);
                    }
                }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u)
#pragma unroll
                    for (int v = 0; v < VEC; ++v)
                        acc[v] = __fadd_rn(acc[v], __fmul_rn(w[u], col_ok ? xv[u][v] : 0.0f));
            }
            for (; e < cnt; ++e) {
                int32_t cu; float w;
                if constexpr (STG) { const int2 p = stg_tile[stg_base + e]; cu = p.x; w = __int_as_float(p.y); }
                else { cu = __shfl(my_col, e, G); w = __shfl(my_val, e, G); }
                float xv[VEC];
                if (col_ok) {
                    if (HUB && cu >= ep.hub_rows) vload_nt<VEC>(xv, x + (int64_t)cu * ldx + c);
                    else vload<VEC>(xv, x + (int64_t)cu * ldx + c);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[v] = __fadd_rn(acc[v], __fmul_rn(w, xv[v]));
                }
            }
        }
        if (live && col_ok) {
            if constexpr (EPI) {
                if (!chunk_mode) apply_epilogue<VEC>(acc, ep, x, ldx, row, c);     // chunks: done by k_long_reduce
            }
            if (bias) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] = __fadd_rn(acc[v], bias[c + v]);
            }
            if constexpr (TOUT) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) y[(int64_t)(c + v) * ldy + row] = acc[v];
            } else if constexpr (HUB) {
                vstore_nt<VEC>(y + row * ldy + c, acc);
            } else {
                vstore<VEC>(y + row * ldy + c, acc);
            }
        }
    }
}

// y[long_row] = sum over its chunks, in chunk order (+ bias)
template <bool EPI>
__global__ void __launch_bounds__(TB)
k_long_reduce(RowSplit sp, int d, float* __restrict__ y, int64_t ldy, const float* __restrict__ bias,
              const float* __restrict__ x, int64_t ldx, Epilogue ep) {
    const int64_t total = (int64_t)(sp.counts ? sp.counts[0] : sp.n_long) * d;
    for (int64_t k = (int64_t)blockIdx.x * TB + threadIdx.x; k < total; k += (int64_t)gridDim.x * TB) {
        const int32_t li = (int32_t)(k / d);
        const int c = (int)(k % d);
        float acc = 0.f;
        for (int32_t q = sp.long_chunk_ptr[li]; q < sp.long_chunk_ptr[li + 1]; ++q)
            acc = __fadd_rn(acc, sp.scratch[(int64_t)q * d + c]);
        if constexpr (EPI) {
            float a1[1] = {acc};
            apply_epilogue<1>(a1, ep, x, ldx, sp.long_rows[li], c);
            acc = a1[0];
        }
        if (bias) acc = __fadd_rn(acc, bias[c]);
        y[(int64_t)sp.long_rows[li] * ldy + c] = acc;
    }
}

// ---- sampled sub-graphs: the rows that can change ---------------------------------------------------------------
// A NeighborLoader batch numbers its nodes seeds first, then in discovery order, and the nodes found in the LAST hop
// are never expanded: rows [n_int, n) of the normalised adjacency hold their self loop only (weight exactly 1), ~90 %
// of a fan-out [15, 10] batch.  K aggregation steps leave those rows unchanged, so only rows [0, n_int) are
// recomputed per step; and in the transposed operator (backward) the interior rows never read a leaf.  This kernel is
// the plain row walk of k_spmm (same accumulation order, separately rounded multiply and add) on a row RANGE, with two
// gather sources split at a column index -- columns < col_split come from `xa`, the others from `xb` -- and an
// optional running sum of the step inputs (sacc_mode 1: sacc[row] = xa[row]; 2: sacc[row] += xa[row]), which the
// backward pass needs once at the end:  (A^T)^K g  restricted to the leaves is  g_L + A_IL^T (h_0 + ... + h_{K-1}).
// Rows of a sampled batch hold at most fan-out + 1 entries: no hub handling here.
template <int G, int VEC>
__global__ void __launch_bounds__(TB)
k_spmm_range(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx, const float* __restrict__ val,
             int64_t row0, int64_t n_rows, int d, const float* __restrict__ xa, int64_t lda,
             const float* __restrict__ xb, int64_t ldb, int32_t col_split, float* __restrict__ y, int64_t ldy,
             const float* __restrict__ bias, float* __restrict__ sacc, int sacc_mode, int src_mode,
             const float* __restrict__ cadd) {
    // src_mode 0: every entry (columns < col_split from xa, the others from xb); 1: only the entries at or beyond
    // col_split (from xb); 2: only the entries below it (from xa), and `cadd[row]` is added to the row's sum -- the
    // forward K-step of a sampled batch forms the leaves' contribution c = A_IL x_L ONCE (mode 1) and then iterates
    // y_I <- A_II y_I + c (mode 2): a step gathers the few interior neighbours instead of every neighbour.  A skipped
    // entry contributes w * 0 at its place in the chain.
    constexpr int ROWS_PER_BLOCK = TB / G;
    const int lane_in_group = threadIdx.x % G;
    const int64_t row = row0 + (int64_t)blockIdx.x * ROWS_PER_BLOCK + threadIdx.x / G;
    const bool live = row < row0 + n_rows;
    int32_t start = 0, end = 0;
    if (live) { start = rowptr[row]; end = rowptr[row + 1]; }
    for (int c0 = 0; c0 < d; c0 += G * VEC) {
        const int c = c0 + lane_in_group * VEC;
        const bool col_ok = c < d;
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.0f;
        for (int32_t base = start; base < end; base += G) {
            const int32_t k = base + lane_in_group;
            const int32_t my_col = k < end ? colidx[k] : 0;
            const float my_val = k < end ? val[k] : 0.0f;
            const int cnt = min((int32_t)G, end - base);
            int e = 0;
            for (; e + UNROLL <= cnt; e += UNROLL) {
                float xv[UNROLL][VEC];
                float w[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const int32_t cu = __shfl(my_col, e + u, G);
                    w[u] = __shfl(my_val, e + u, G);
                    const float* src = cu < col_split ? xa + (int64_t)cu * lda : xb + (int64_t)cu * ldb;
                    const bool take = src_mode == 0 || (src_mode == 1) == (cu >= col_split);
                    if (col_ok && take) vload<VEC>(xv[u], src + c);
                    else {
#pragma unroll
                        for (int v = 0; v < VEC; ++v) xv[u][v] = 0.0f;
                    }
                }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u)
#pragma unroll
                    for (int v = 0; v < VEC; ++v)
                        acc[v] = __fadd_rn(acc[v], __fmul_rn(w[u], xv[u][v]));
            }
            for (; e < cnt; ++e) {
                const int32_t cu = __shfl(my_col, e, G);
                const float w = __shfl(my_val, e, G);
                const float* src = cu < col_split ? xa + (int64_t)cu * lda : xb + (int64_t)cu * ldb;
                const bool take = src_mode == 0 || (src_mode == 1) == (cu >= col_split);
                float xv[VEC];
                if (col_ok && take) {
                    vload<VEC>(xv, src + c);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[v] = __fadd_rn(acc[v], __fmul_rn(w, xv[v]));
                }
            }
        }
        if (live && col_ok) {
            if (sacc_mode) {                              // running sum of the step inputs (own row: no race)
                float own[VEC], run[VEC];
                vload<VEC>(own, xa + row * lda + c);
                if (sacc_mode == 2) {
                    vload<VEC>(run, sacc + row * (int64_t)d + c);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) own[v] = __fadd_rn(run[v], own[v]);
                }
                vstore<VEC>(sacc + row * (int64_t)d + c, own);
            }
            if (cadd) {
                float cv[VEC];
                vload<VEC>(cv, cadd + row * (int64_t)d + c);
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] = __fadd_rn(acc[v], cv[v]);
            }
            if (bias) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] = __fadd_rn(acc[v], bias[c + v]);
            }
            vstore<VEC>(y + row * ldy + c, acc);
        }
    }
}

// y[row] = x[row] (+ bias) for rows [row0, row0 + n_rows): the leaf rows of a forward K-step
__global__ void __launch_bounds__(TB)
k_rows_copy_bias(const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy, int64_t row0, int64_t n_rows,
                 int d, const float* __restrict__ bias) {
    const int64_t total = n_rows * d;
    for (int64_t k = (int64_t)blockIdx.x * TB + threadIdx.x; k < total; k += (int64_t)gridDim.x * TB) {
        const int64_t r = row0 + k / d;
        const int c = (int)(k % d);
        // 0 + 1 * x: what the self-loop row computes (a negative zero becomes +0, like there)
        float v = __fadd_rn(0.0f, __fmul_rn(1.0f, x[r * ldx + c]));
        if (bias) v = __fadd_rn(v, bias[c]);
        y[r * ldy + c] = v;
    }
}

template <int G, int VEC>
int launch_range(const int32_t* rowptr, const int32_t* colidx, const float* val, int64_t row0, int64_t n_rows, int d,
                 const float* xa, int64_t lda, const float* xb, int64_t ldb, int32_t col_split, float* y, int64_t ldy,
                 const float* bias, float* sacc, int sacc_mode, hipStream_t s, int src_mode = 0, const float* cadd = nullptr) {
    if (n_rows <= 0) return GDA_OK;
    constexpr int ROWS_PER_BLOCK = TB / G;
    const int64_t blocks = gda_cdiv(n_rows, ROWS_PER_BLOCK);
    if (blocks > INT32_MAX) return GDA_E_SIZE;
    k_spmm_range<G, VEC><<<(unsigned)blocks, TB, 0, s>>>(rowptr, colidx, val, row0, n_rows, d, xa, lda, xb, ldb, col_split,
                                                         y, ldy, bias, sacc, sacc_mode, src_mode, cadd);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

int dispatch_range(const int32_t* rowptr, const int32_t* colidx, const float* val, int64_t row0, int64_t n_rows, int64_t d,
                   const float* xa, int64_t lda, const float* xb, int64_t ldb, int32_t col_split, float* y, int64_t ldy,
                   const float* bias, float* sacc, int sacc_mode, hipStream_t s, int src_mode = 0, const float* cadd = nullptr) {
    const bool a16 = (d % 4 == 0) && (lda % 4 == 0) && (ldb % 4 == 0) && (ldy % 4 == 0) && ((uintptr_t)xa % 16 == 0) &&
                     ((uintptr_t)xb % 16 == 0) && ((uintptr_t)y % 16 == 0) && (!sacc || (uintptr_t)sacc % 16 == 0) &&
                     (!cadd || (uintptr_t)cadd % 16 == 0);
    const int di = (int)d;
#define GO(G, V) return launch_range<G, V>(rowptr, colidx, val, row0, n_rows, di, xa, lda, xb, ldb, col_split, y, ldy, bias, sacc, sacc_mode, s, src_mode, cadd)
    if (a16) {
        const int64_t lanes = d / 4;
        if (lanes >= 64) GO(64, 4);
        if (lanes > 16) GO(32, 4);
        if (lanes > 8) GO(16, 4);
        if (lanes > 4) GO(8, 4);
        GO(4, 4);
    }
    if (d > 32) GO(64, 1);
    if (d > 16) GO(32, 1);
    if (d > 8) GO(16, 1);
    if (d > 4) GO(8, 1);
    GO(4, 1);
#undef GO
}

// PYGDA_AMD_SPMM_HUB_ROWS = H > 0: the HUB kernel with source rows below H as the hubs (d = 128 launches).  OFF by default
// -- measured on R-MAT 2^22, degree-ordered, d = 128 (profiles/r6_experiments.txt 19): L2-side traffic 32.9 -> 28.9 GB at
// H = 6,144 (launch 4.52 -> 4.84 ms), 29.4 GB at H = 16,384 (4.55 ms): the hint steers a tenth of the bytes and no time.
static int32_t hub_rows() {
    static const int32_t v = getenv("PYGDA_AMD_SPMM_HUB_ROWS") ? (int32_t)atoi(getenv("PYGDA_AMD_SPMM_HUB_ROWS")) : 0;
    return v > 0 ? v : 0;
}

template <int G, int VEC>
int launch(const int32_t* rowptr, const int32_t* colidx, const float* val, int64_t n_rows, int d,
           const float* x, int64_t ldx, float* y, int64_t ldy, const float* bias, const RowSplit& sp,
           hipStream_t s, const Epilogue* ep) {
    constexpr int ROWS_PER_BLOCK = TB / G;
    const int64_t row_blocks = gda_cdiv(n_rows, ROWS_PER_BLOCK);
    const int64_t chunk_blocks = sp.n_chunks > 0 ? gda_cdiv(sp.n_chunks, ROWS_PER_BLOCK) : 0;
    if (row_blocks + chunk_blocks > INT32_MAX) return GDA_E_SIZE;
    static const bool staged = getenv("PYGDA_AMD_SPMM_LDS") != nullptr;     // measured alternative (see k_spmm): off by default
    if (ep)
        k_spmm<G, VEC, true><<<(unsigned)(row_blocks + chunk_blocks), TB, 0, s>>>(
            rowptr, colidx, val, n_rows, d, x, ldx, y, ldy, bias, sp, (unsigned)row_blocks, *ep);
    else if (staged)
        k_spmm<G, VEC, false, true><<<(unsigned)(row_blocks + chunk_blocks), TB, 0, s>>>(
            rowptr, colidx, val, n_rows, d, x, ldx, y, ldy, bias, sp, (unsigned)row_blocks, Epilogue{});
    else if (const int32_t hub = (G == 32 && VEC == 4) ? hub_rows() : 0) {
        Epilogue e{};
        e.hub_rows = hub;
        if constexpr (G == 32 && VEC == 4)
            k_spmm<G, VEC, false, false, false, true><<<(unsigned)(row_blocks + chunk_blocks), TB, 0, s>>>(
                rowptr, colidx, val, n_rows, d, x, ldx, y, ldy, bias, sp, (unsigned)row_blocks, e);
    } else
        k_spmm<G, VEC, false><<<(unsigned)(row_blocks + chunk_blocks), TB, 0, s>>>(
            rowptr, colidx, val, n_rows, d, x, ldx, y, ldy, bias, sp, (unsigned)row_blocks, Epilogue{});
    GDA_LAUNCH_CHECK();
    if (sp.n_long > 0) {
        int64_t g = gda_cdiv((int64_t)sp.n_long * d, TB);
        if (g > 2048) g = 2048;
        if (ep) k_long_reduce<true><<<(unsigned)g, TB, 0, s>>>(sp, d, y, ldy, bias, x, ldx, *ep);
        else k_long_reduce<false><<<(unsigned)g, TB, 0, s>>>(sp, d, y, ldy, bias, x, ldx, Epilogue{});
        GDA_LAUNCH_CHECK();
    }
    return GDA_OK;
}

int dispatch(const int32_t* rowptr, const int32_t* colidx, const float* val, int64_t n_rows, int64_t d,
             const float* x, int64_t ldx, float* y, int64_t ldy, const float* bias, const RowSplit& sp,
             hipStream_t s, const Epilogue* ep = nullptr) {
    // vector width: widest that keeps every row start and column offset aligned
    const bool z16 = !ep || !ep->z || ((ep->ldz % 4 == 0) && ((uintptr_t)ep->z % 16 == 0));
    const bool z8 = !ep || !ep->z || ((ep->ldz % 2 == 0) && ((uintptr_t)ep->z % 8 == 0));
    const bool a16 = (d % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && z16 &&
                     ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0);
    const bool a8 = (d % 2 == 0) && (ldx % 2 == 0) && (ldy % 2 == 0) && z8 &&
                    ((uintptr_t)x % 8 == 0) && ((uintptr_t)y % 8 == 0);
    const int di = (int)d;
#define GO(G, V) return launch<G, V>(rowptr, colidx, val, n_rows, di, x, ldx, y, ldy, bias, sp, s, ep)
    if (a16) {
        const int64_t lanes = d / 4;
        if (lanes >= 64) GO(64, 4);
        if (lanes > 16) GO(32, 4);
        if (lanes > 8) GO(16, 4);
        if (lanes > 4) GO(8, 4);
        GO(4, 4);
    }
    if (a8) {
        const int64_t lanes = d / 2;
        if (lanes > 32) GO(64, 2);
        if (lanes > 16) GO(32, 2);
        if (lanes > 8) GO(16, 2);
        if (lanes > 4) GO(8, 2);
        GO(4, 2);
    }
    if (d > 32) GO(64, 1);
    if (d > 16) GO(32, 1);
    if (d > 8) GO(16, 1);
    if (d > 4) GO(8, 1);
    GO(4, 1);
#undef GO
}

template <int G, int VEC>
int launch_tout(const int32_t* rowptr, const int32_t* colidx, const float* val, int64_t n_rows, int d,
                const float* x, int64_t ldx, float* yT, int64_t ldyT, const float* bias, hipStream_t s) {
    const int64_t row_blocks = gda_cdiv(n_rows, TB / G);
    if (row_blocks > INT32_MAX) return GDA_E_SIZE;
    k_spmm<G, VEC, false, false, true><<<(unsigned)row_blocks, TB, 0, s>>>(
        rowptr, colidx, val, n_rows, d, x, ldx, yT, ldyT, bias, RowSplit{}, (unsigned)row_blocks, Epilogue{});
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

int check(const int32_t* rowptr, const int32_t* colidx, const float* val, int64_t n_rows, int64_t d,
          const float* x, int64_t ldx, float* y, int64_t ldy) {
    if (n_rows < 0 || d < 0 || n_rows >= INT32_MAX || d >= INT32_MAX || ldx < d || ldy < d) return GDA_E_SIZE;
    if (n_rows == 0 || d == 0) return GDA_OK;
    if (!rowptr || !x || !y) return GDA_E_NULL;
    (void)colidx; (void)val;   // may be NULL only when nnz == 0, which the host cannot see here
    if ((const float*)y == x) return GDA_E_ALIAS;
    return GDA_OK;
}

RowSplit make_split(const gda_row_split* h) {
    RowSplit sp{};
    if (h && h->threshold > 0 && h->n_long > 0) {
        sp.threshold = h->threshold; sp.n_long = h->n_long; sp.n_chunks = h->n_chunks;
        sp.long_rows = h->long_rows; sp.long_chunk_ptr = h->long_chunk_ptr; sp.chunk_long = h->chunk_long;
        sp.scratch = h->scratch; sp.counts = h->counts_dev;
    }
    return sp;
}

int check_split(const gda_row_split* h) {
    if (!h || h->n_long <= 0) return GDA_OK;
    if (h->threshold <= 0 || h->n_chunks < h->n_long) return GDA_E_SIZE;
    if (!h->long_rows || !h->long_chunk_ptr || !h->chunk_long || !h->scratch) return GDA_E_NULL;
    return GDA_OK;
}

}  // namespace

extern "C" int gda_spmm_csr_f32(const int32_t* rowptr, const int32_t* colidx, const float* val,
                                int64_t n_rows, int64_t d, const float* x, int64_t ldx,
                                float* y, int64_t ldy, const float* bias, gda_stream_t stream) {
    return gda_spmm_csr_split_f32(rowptr, colidx, val, n_rows, d, 1, x, ldx, y, ldy, nullptr, bias, nullptr,
                                  stream);
}

extern "C" int gda_spmm_csr_tout_f32(const int32_t* rowptr, const int32_t* colidx, const float* val,
                                     int64_t n_rows, int64_t d, const float* x, int64_t ldx,
                                     float* yT, int64_t ldyT, const float* bias, gda_stream_t stream) {
    if (n_rows < 0 || d < 0 || n_rows >= INT32_MAX || d >= INT32_MAX || ldx < d || ldyT < n_rows) return GDA_E_SIZE;
    if (n_rows == 0 || d == 0) return GDA_OK;
    if (!rowptr || !x || !yT) return GDA_E_NULL;
    if ((const float*)yT == x) return GDA_E_ALIAS;
    hipStream_t s = (hipStream_t)stream;
    const int di = (int)d;
    const bool a16 = (d % 4 == 0) && (ldx % 4 == 0) && ((uintptr_t)x % 16 == 0);
#define GO(G, V) return launch_tout<G, V>(rowptr, colidx, val, n_rows, di, x, ldx, yT, ldyT, bias, s)
    if (a16) {
        const int64_t lanes = d / 4;
        if (lanes >= 64) GO(64, 4);
        if (lanes > 16) GO(32, 4);
        if (lanes > 8) GO(16, 4);
        if (lanes > 4) GO(8, 4);
        GO(4, 4);
    }
    if (d > 32) GO(64, 1);
    if (d > 16) GO(32, 1);
    if (d > 8) GO(16, 1);
    if (d > 4) GO(8, 1);
    GO(4, 1);
#undef GO
}

extern "C" int gda_spmm_csr_kstep_f32(const int32_t* rowptr, const int32_t* colidx, const float* val,
                                      int64_t n_rows, int64_t d, int K, const float* x, int64_t ldx,
                                      float* y, int64_t ldy, float* tmp, const float* bias,
                                      gda_stream_t stream) {
    return gda_spmm_csr_split_f32(rowptr, colidx, val, n_rows, d, K, x, ldx, y, ldy, tmp, bias, nullptr, stream);
}

extern "C" int gda_spmm_csr_split_f32(const int32_t* rowptr, const int32_t* colidx, const float* val,
                                      int64_t n_rows, int64_t d, int K, const float* x, int64_t ldx,
                                      float* y, int64_t ldy, float* tmp, const float* bias,
                                      const gda_row_split* split, gda_stream_t stream) {
    if (K < 1) return GDA_E_SIZE;
    int st = check(rowptr, colidx, val, n_rows, d, x, ldx, y, ldy);
    if (st != GDA_OK || n_rows == 0 || d == 0) return st;
    if ((st = check_split(split)) != GDA_OK) return st;
    if (K > 1 && !tmp) return GDA_E_NULL;
    if (K > 1 && (tmp == y || (const float*)tmp == x)) return GDA_E_ALIAS;
    const RowSplit sp = make_split(split);
    const float* in = x;
    int64_t ld_in = ldx;
    for (int j = 1; j <= K; ++j) {                   // step j writes y when K-j is even
        float* out = ((K - j) % 2 == 0) ? y : tmp;
        const int r = dispatch(rowptr, colidx, val, n_rows, d, in, ld_in, out, ldy,
                               j == K ? bias : nullptr, sp, (hipStream_t)stream);
        if (r != GDA_OK) return r;
        in = out;
        ld_in = ldy;
    }
    return GDA_OK;
}

extern "C" int gda_spmm_csr_axpby_f32(const int32_t* rowptr, const int32_t* colidx, const float* val,
                                      int64_t n_rows, int64_t d, const float* x, int64_t ldx,
                                      float* y, int64_t ldy, float alpha, float beta,
                                      const float* z, int64_t ldz, float gamma, const float* gamma_dev,
                                      const gda_row_split* split, gda_stream_t stream) {
    int st = check(rowptr, colidx, val, n_rows, d, x, ldx, y, ldy);
    if (st != GDA_OK || n_rows == 0 || d == 0) return st;
    if ((st = check_split(split)) != GDA_OK) return st;
    if (z && ldz < d) return GDA_E_SIZE;
    if (z == y) return GDA_E_ALIAS;
    const RowSplit sp = make_split(split);
    const Epilogue ep{alpha, beta, gamma, gamma_dev, z, ldz};
    return dispatch(rowptr, colidx, val, n_rows, d, x, ldx, y, ldy, nullptr, sp, (hipStream_t)stream, &ep);
}

// K aggregation steps on a sampled sub-graph whose rows [n_int, n_rows) hold their unit self loop only (the nodes a
// NeighborLoader batch discovered in its last hop; csrc/gda_dsampler.hip reports n_int).
//   transposed = 0 (rows by destination):  y = A^K x (+ bias): K steps over rows [0, n_int) -- gathers of leaf columns
//     read x itself -- and one copy of the leaf rows.  Same values as gda_spmm_csr_kstep_f32 (signed zeros aside).
//     With `sacc` given and K >= 2 the leaf columns' contribution (the same in every step) is formed once and a step
//     gathers interior columns only: same result up to fp32 summation order (a row's leaf entries are summed apart).
//   transposed = 1 (rows by source; the backward pass):  y = (A^T)^K x: K steps over the interior rows (they read interior
//     columns only), their inputs summed on the way, then ONE pass over the leaf rows  y_L = x_L + A_IL^T (h_0 + .. + h_{K-1}).
//     Same result as K full steps up to fp32 summation order (the leaves' sums are re-associated).
// tmp: [n_int, d] (K > 1); sacc: [n_int, d] (transposed: required; forward: optional, see above).  x, y contiguous rows (ld = d).
extern "C" int gda_spmm_csr_interior_kstep_f32(const int32_t* rowptr, const int32_t* colidx, const float* val,
                                               int64_t n_rows, int64_t n_int, int64_t d, int K, int transposed,
                                               const float* x, float* y, float* tmp, float* sacc, const float* bias,
                                               gda_stream_t stream) {
    if (K < 1 || n_int < 0 || n_int > n_rows) return GDA_E_SIZE;
    int st = check(rowptr, colidx, val, n_rows, d, x, d, y, d);
    if (st != GDA_OK || n_rows == 0 || d == 0) return st;
    if ((K > 1 && n_int > 0 && !tmp) || (transposed && n_int > 0 && !sacc)) return GDA_E_NULL;
    if (tmp && (tmp == y || (const float*)tmp == x)) return GDA_E_ALIAS;
    if (transposed && bias) return GDA_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int32_t split = (int32_t)n_int;
    const float* in = x;
    // forward with a scratch and K >= 2: the leaf columns read x in EVERY step and x does not change -- their
    // contribution c = A_IL x_L is formed once into `sacc`, the steps gather interior columns only
    const bool hoist = !transposed && K >= 2 && n_int > 0 && sacc != nullptr;
    if (hoist) {
        const int r = dispatch_range(rowptr, colidx, val, 0, n_int, d, x, d, x, d, split, sacc, d, nullptr, nullptr, 0, s, 1, nullptr);
        if (r != GDA_OK) return r;
    }
    for (int j = 1; j <= K && n_int > 0; ++j) {          // step j writes y when K - j is even
        float* out = ((K - j) % 2 == 0) ? y : tmp;
        // forward: leaf columns always come from x; transposed: interior rows have no leaf columns at all
        const int r = hoist
            ? dispatch_range(rowptr, colidx, val, 0, n_int, d, in, d, x, d, split, out, d, j == K ? bias : nullptr, nullptr, 0, s, 2, sacc)
            : dispatch_range(rowptr, colidx, val, 0, n_int, d, in, d, x, d, split, out, d,
                             (j == K && !transposed) ? bias : nullptr, transposed ? sacc : nullptr,
                             transposed ? (j == 1 ? 1 : 2) : 0, s);
        if (r != GDA_OK) return r;
        in = out;
    }
    const int64_t n_leaf = n_rows - n_int;
    if (n_leaf == 0) return GDA_OK;
    if (!transposed) {
        int64_t g = gda_cdiv(n_leaf * d, TB);
        if (g > 256 * 16) g = 256 * 16;
        k_rows_copy_bias<<<(unsigned)g, TB, 0, s>>>(x, d, y, d, n_int, n_leaf, (int)d, bias);
        GDA_LAUNCH_CHECK();
        return GDA_OK;
    }
    // leaves of the transposed operator: interior columns read the summed step inputs, the self loop reads x
    return dispatch_range(rowptr, colidx, val, n_int, n_leaf, d, n_int > 0 ? sacc : x, d, x, d, split, y, d, nullptr, nullptr, 0, s);
}

#include "gda_interior.inc"
