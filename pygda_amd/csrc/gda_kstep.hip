// K-step neighbour aggregation  y = A_hat^K x (+ bias)  in ONE launch for graphs whose feature
// columns fit the LDS of a CU (the citation-graph regime: N <= 16320 nodes).
//
// Replaces the prop_nums loop of pygda/nn/prop_gcn_conv.py:208-210 (one PyG propagate per step) and
// the K dependent launches gda_spmm_csr_kstep_f32 makes of it.  At N ~ 5-10 k nodes a step moves
// 6 MB and costs a launch's latency (6-7 us); the product is independent per feature column, so:
//
//   * a workgroup owns ONE feature column of ALL rows and keeps it in LDS (ping-pong, N x 4 B each):
//     no inter-workgroup exchange, K steps separated by workgroup barriers only;
//   * the graph is compiled once (gda_kstep_plan_host) into a per-lane register program: thread t owns a
//     contiguous row range cut into S slots of L entries (a row of length len takes ceil(len/L) consecutive
//     slots, padded with zero-weight entries that read a dedicated zero word), stored wave-transposed so
//     the per-launch load is coalesced.  Inside the step loop a thread touches no global memory: R = S*L
//     LDS gathers, the sequential separately-rounded multiply-add chain of each row in CSR (= edge) order
//     -- bit for bit the CPU scatter-add result -- and one LDS store per slot at a compile-time position;
//   * activations cross the kernel in column-major [d, ldT] (contiguous per column): the row-major
//     wrapper below transposes through 64x64 LDS tiles before and after.
//
// Bound: LDS gather rate / VALU issue of one CU per column (not HBM: the working set never leaves LDS).
#include "gda_common.h"

#include <algorithm>
#include <cstring>
#include <vector>

namespace {

constexpr int KS_TB = 1024;          // 16 wavefronts: 4 per SIMD, <= 128 VGPRs each
constexpr int KS_L = 4;              // entries per slot
constexpr int KS_OFFB = 65528;       // byte offset of the second LDS buffer: fits the 16-bit DS offset field
constexpr int KS_MAX_ROWS = (KS_OFFB / 4 - 32) / 32 * 32;    // 16,320: 510 node words on each of the 32 banks
constexpr int KS_BANKS = 32;         // ds_read_b32 / ds_write_b32: bank = (byte address / 4) mod 32, lanes 0-31 and 32-63 apart
constexpr int KS_ZERO_W = 0;         // word 0 of each buffer reads as 0 (padding entries), word 1 takes the writes of
constexpr int KS_DUMP_W = 1;         // slots that do not end a row; node words start at KS_NODE_W0
constexpr int KS_NODE_W0 = KS_BANKS;
constexpr int KS_POS_WORDS = (KS_MAX_ROWS + 2 + 3) / 4 * 4;      // the plan's node -> LDS address table (fixed size)

// ds_read / ds_write at (LDS byte address held in a register) + (compile-time offset): the offset folds
// into the instruction's 16-bit offset field, the register holds an absolute LDS address (the kernel adds
// the dynamic-LDS base to the plan's addresses once, at load time), so a gather costs no address VALU
typedef __attribute__((address_space(3))) float lds_float;
template <int OFF>
__device__ __forceinline__ float lds_ld(unsigned a) {
    return *reinterpret_cast<const lds_float*>((uintptr_t)(a + (unsigned)OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_st(unsigned a, float v) {
    *reinterpret_cast<lds_float*>((uintptr_t)(a + (unsigned)OFF)) = v;
}

template <int S, int RD, int WR>
__device__ __forceinline__ void ks_step(const unsigned (&ea)[S * KS_L], const float (&ew)[S * KS_L],
                                        const unsigned (&oa)[S], unsigned keep) {
    // the gathers run D slots ahead of the multiply-add chain (a ring of D * L registers): enough LDS
    // latency cover next to the other three waves of the SIMD, without holding all R values at once
    constexpr int D = S <= 10 ? 4 : 2;
    float xv[D][KS_L];
#pragma unroll
    for (int s = 0; s < D && s < S; ++s)
#pragma unroll
        for (int l = 0; l < KS_L; ++l) xv[s][l] = lds_ld<RD>(ea[s * KS_L + l]);
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
#pragma unroll
        for (int l = 0; l < KS_L; ++l) acc = __fadd_rn(acc, __fmul_rn(ew[s * KS_L + l], xv[s % D][l]));
        if (s + D < S) {
#pragma unroll
            for (int l = 0; l < KS_L; ++l) xv[s % D][l] = lds_ld<RD>(ea[(s + D) * KS_L + l]);
        }
        lds_st<WR>(oa[s], acc);                       // rows' last slots hit their row, the others a dump word
        acc = (keep >> s) & 1u ? acc : 0.f;                // a row continues in the next slot
    }
}

template <int S>
__global__ void __launch_bounds__(KS_TB)
k_kstep_lds(const int2* __restrict__ ent, const unsigned* __restrict__ outa, const unsigned* __restrict__ keepm,
            const unsigned* __restrict__ pos, int n_rows, int n_pad, int K, const float* __restrict__ xT, int64_t ldx, float* __restrict__ yT, int64_t ldy,
            const float* __restrict__ bias, float* __restrict__ colsum) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int R = S * KS_L;
    const int c = blockIdx.x;
    const int t = threadIdx.x;
    float* bufA = reinterpret_cast<float*>(lds);
    float* bufB = reinterpret_cast<float*>(lds + KS_OFFB);
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const float* xc = xT + (int64_t)c * ldx;
    float csum = 0.f;
    for (int i = t * 4; i < n_pad; i += KS_TB * 4) {
        const float4 v = *reinterpret_cast<const float4*>(xc + i);
        const uint4 p = *reinterpret_cast<const uint4*>(pos + i);      // node i lives at LDS byte address pos[i]
        lds_st<0>(p.x + base, v.x); lds_st<0>(p.y + base, v.y); lds_st<0>(p.z + base, v.z); lds_st<0>(p.w + base, v.w);
        if (colsum) {                                        // column sum of the INPUT over the real rows
            csum += (i + 0 < n_rows ? v.x : 0.f); csum += (i + 1 < n_rows ? v.y : 0.f);
            csum += (i + 2 < n_rows ? v.z : 0.f); csum += (i + 3 < n_rows ? v.w : 0.f);
        }
    }
    if (t < 1) { bufA[KS_ZERO_W] = 0.f; bufB[KS_ZERO_W] = 0.f; }        // the zero word of each buffer
    if (colsum) {      // fixed-order block reduction (wave butterflies, then 16 leaders through the spare LDS above both buffers)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) csum += __shfl_down(csum, off, 64);
        float* red = reinterpret_cast<float*>(lds + 2 * KS_OFFB);
        if ((t & 63) == 0) red[t >> 6] = csum;
        __syncthreads();
        if (t == 0) {
            float sres = 0.f;
#pragma unroll
            for (int k = 0; k < KS_TB / 64; ++k) sres += red[k];
            colsum[c] = sres;
        }
    }
    unsigned ea[R], oa[S];
    float ew[R];
    const int w = t >> 6, lane = t & 63;
    const int2* ep = ent + ((size_t)w * R) * 64 + lane;
#pragma unroll
    for (int j = 0; j < R; ++j) { const int2 e = ep[(size_t)j * 64]; ea[j] = (unsigned)e.x + base; ew[j] = __int_as_float(e.y);
        asm volatile("" : "+v"(ea[j])); }       // pin the sum in the register: no re-derivation from the symbol inside the loop
    const unsigned* op = outa + ((size_t)w * S) * 64 + lane;
#pragma unroll
    for (int s = 0; s < S; ++s) { oa[s] = op[(size_t)s * 64] + base; asm volatile("" : "+v"(oa[s])); }
    const unsigned keep = keepm[t];
    __syncthreads();
    int step = 0;
    for (; step + 2 <= K; step += 2) {
        ks_step<S, 0, KS_OFFB>(ea, ew, oa, keep);
        __syncthreads();
        ks_step<S, KS_OFFB, 0>(ea, ew, oa, keep);
        __syncthreads();
    }
    unsigned rbase = base;
    if (step < K) {
        ks_step<S, 0, KS_OFFB>(ea, ew, oa, keep);
        __syncthreads();
        rbase = base + KS_OFFB;
    }
    const float bv = bias ? bias[c] : 0.f;
    float* yc = yT + (int64_t)c * ldy;
    for (int i = t * 4; i < n_pad; i += KS_TB * 4) {
        const uint4 p = *reinterpret_cast<const uint4*>(pos + i);
        float4 v;
        v.x = lds_ld<0>(p.x + rbase); v.y = lds_ld<0>(p.y + rbase); v.z = lds_ld<0>(p.z + rbase); v.w = lds_ld<0>(p.w + rbase);
        if (bias) { v.x = __fadd_rn(v.x, bv); v.y = __fadd_rn(v.y, bv); v.z = __fadd_rn(v.z, bv); v.w = __fadd_rn(v.w, bv); }
        *reinterpret_cast<float4*>(yc + i) = v;
    }
}

// [rows, cols] (ld = ldi) -> [cols, rows] (ld = ldo); 64 x 64 tiles through LDS, 16-byte accesses along the
// contiguous side of both matrices when the shapes allow
__global__ void __launch_bounds__(256)
k_transpose(const float* __restrict__ in, int64_t ldi, float* __restrict__ out, int64_t ldo, int rows, int cols) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
    for (int r = ty; r < 64; r += 4)
        if (r0 + r < rows && c0 + tx < cols) tile[r][tx] = in[(int64_t)(r0 + r) * ldi + c0 + tx];
    __syncthreads();
#pragma unroll 4
    for (int cc = ty; cc < 64; cc += 4)
        if (c0 + cc < cols && r0 + tx < rows) out[(int64_t)(c0 + cc) * ldo + r0 + tx] = tile[tx][cc];
}

template <int S>
int ks_launch(const int2* ent, const unsigned* outa, const unsigned* keep, const unsigned* pos, int n_rows, int n_pad, int d,
              int K, const float* xT, int64_t ldx, float* yT, int64_t ldy, const float* bias, float* colsum, hipStream_t s) {
    const size_t lds = (size_t)2 * KS_OFFB + (size_t)(KS_TB / 64) * 4;       // both buffers in full + the column-sum scratch
    static bool configured = false;          // idempotent attribute; racing first calls set the same value
    if (!configured) {
        GDA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_kstep_lds<S>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        configured = true;
    }
    k_kstep_lds<S><<<(unsigned)d, KS_TB, lds, s>>>(ent, outa, keep, pos, n_rows, n_pad, K, xT, ldx, yT, ldy, bias, colsum);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

// ---- bank-aware node placement -------------------------------------------------------------------------
// The step loop is bound by its LDS gathers: every wave instruction reads 2 x 32 unrelated words, and 32 random
// words over 32 banks put ~3.4 distinct addresses on the busiest bank (measured: 1.6 us per step at the cfg-A
// target graph, 2.7 LDS cycles per lane group where 1 is the floor).  WHERE a node's word lives in LDS is free
// -- only the plan's addresses and the column's load / store go through the table -- so the nodes are assigned
// to banks such that the nodes one lane group touches in one instruction (32 neighbours of a ds_read, 32 row
// ends of a ds_write, 32 nodes of the column load / store) sit on different banks as far as possible: a greedy
// pass in order of decreasing incidence, then local-search sweeps, on  sum_groups weight * (distinct addresses on
// the group's busiest bank).  The sums are untouched (same entries, same order): results stay bit-identical.
struct KsGroups {
    std::vector<int> gptr{0}, gnode, gweight;
    std::vector<unsigned char> gfix;         // bit 0: the zero word (bank of KS_ZERO_W) is read, bit 1: the dump word is written
    void add(std::vector<int>& nodes, int weight, unsigned char fix) {
        std::sort(nodes.begin(), nodes.end());
        nodes.erase(std::unique(nodes.begin(), nodes.end()), nodes.end());
        if (nodes.size() + (fix ? 1 : 0) < 2) { nodes.clear(); return; }          // nothing can collide
        gnode.insert(gnode.end(), nodes.begin(), nodes.end());
        gptr.push_back((int)gnode.size());
        gweight.push_back(weight);
        gfix.push_back(fix);
        nodes.clear();
    }
};

void ks_place(int n, int S, const std::vector<int>& ent_node, const std::vector<int>& out_node, bool bank_aware,
              std::vector<unsigned>& pos_word) {
    pos_word.assign((size_t)n, 0u);
    if (!bank_aware) {
        for (int i = 0; i < n; ++i) pos_word[i] = (unsigned)(KS_NODE_W0 + i);
        return;
    }
    const int R = S * KS_L, NB = KS_BANKS;
    constexpr int W_STEP = 4, W_IO = 1;          // a step's groups run K times per launch, the column load / store once each
    KsGroups G;
    std::vector<int> tmp;
    for (int w = 0; w < KS_TB / 64; ++w)
        for (int j = 0; j < R; ++j)
            for (int h = 0; h < 2; ++h) {
                unsigned char fix = 0;
                for (int l = 0; l < 32; ++l) {
                    const int v = ent_node[((size_t)w * R + j) * 64 + h * 32 + l];
                    if (v >= 0) tmp.push_back(v); else fix = 1;
                }
                G.add(tmp, W_STEP, fix);
            }
    for (int w = 0; w < KS_TB / 64; ++w)
        for (int sl = 0; sl < S; ++sl)
            for (int h = 0; h < 2; ++h) {
                unsigned char fix = 0;
                for (int l = 0; l < 32; ++l) {
                    const int v = out_node[((size_t)w * S + sl) * 64 + h * 32 + l];
                    if (v >= 0) tmp.push_back(v); else fix = 2;
                }
                G.add(tmp, W_STEP, fix);
            }
    for (int i0 = 0; i0 < n; i0 += 128)          // the column load / store: lane l of a 32-lane group holds nodes i0 + 4l .. 4l+3
        for (int j = 0; j < 4; ++j) {
            for (int l = 0; l < 32; ++l) if (i0 + 4 * l + j < n) tmp.push_back(i0 + 4 * l + j);
            G.add(tmp, 2 * W_IO, 0);
        }
    const int ng = (int)G.gweight.size();
    std::vector<int> iptr((size_t)n + 1, 0), inc(G.gnode.size());
    for (int v : G.gnode) ++iptr[v + 1];
    for (int i = 0; i < n; ++i) iptr[i + 1] += iptr[i];
    {
        std::vector<int> fill(iptr.begin(), iptr.end() - 1);
        for (int g = 0; g < ng; ++g)
            for (int k = G.gptr[g]; k < G.gptr[g + 1]; ++k) inc[fill[G.gnode[k]]++] = g;
    }
    std::vector<unsigned short> cnt((size_t)ng * NB, 0), gmax((size_t)ng, 0);
    for (int g = 0; g < ng; ++g) {
        if (G.gfix[g] & 1) cnt[(size_t)g * NB + KS_ZERO_W % NB] = 1;
        if (G.gfix[g] & 2) cnt[(size_t)g * NB + KS_DUMP_W % NB] = 1;
        gmax[g] = G.gfix[g] ? 1 : 0;
    }
    const int cap = (KS_OFFB / 4 - KS_NODE_W0) / NB;
    std::vector<int> bank((size_t)n, -1), load(NB, 0);
    auto put = [&](int v, int b) {
        bank[v] = b; ++load[b];
        for (int k = iptr[v]; k < iptr[v + 1]; ++k) {
            const int g = inc[k];
            const unsigned short c = ++cnt[(size_t)g * NB + b];
            if (c > gmax[g]) gmax[g] = c;
        }
    };
    auto take = [&](int v) {
        const int b = bank[v];
        --load[b];
        for (int k = iptr[v]; k < iptr[v + 1]; ++k) {
            const int g = inc[k];
            const unsigned short c = cnt[(size_t)g * NB + b]--;
            if (c == gmax[g]) gmax[g] = *std::max_element(cnt.begin() + (size_t)g * NB, cnt.begin() + (size_t)(g + 1) * NB);
        }
    };
    // (growth of the groups' maxima, crowding of the bank inside the node's groups) of putting v on bank b
    auto score = [&](int v, int b, int64_t& hard, int64_t& soft) {
        hard = soft = 0;
        for (int k = iptr[v]; k < iptr[v + 1]; ++k) {
            const int g = inc[k];
            const int c = cnt[(size_t)g * NB + b];
            hard += (c + 1 > gmax[g]) ? G.gweight[g] : 0;
            soft += (int64_t)c * G.gweight[g];
        }
    };
    std::vector<int> order((size_t)n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::vector<int64_t> wsum((size_t)n, 0);
    for (int v = 0; v < n; ++v) for (int k = iptr[v]; k < iptr[v + 1]; ++k) wsum[v] += G.gweight[inc[k]];
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return wsum[a] > wsum[b]; });
    for (int v : order) {
        int best = -1; int64_t bh = 0, bs = 0;
        for (int b = 0; b < NB; ++b) {
            if (load[b] >= cap) continue;
            int64_t h, sft; score(v, b, h, sft);
            if (best < 0 || h < bh || (h == bh && (sft < bs || (sft == bs && load[b] < load[best])))) { best = b; bh = h; bs = sft; }
        }
        put(v, best);
    }
    for (int sweep = 0; sweep < 8; ++sweep) {
        int moved = 0;
        for (int v = 0; v < n; ++v) {
            if (iptr[v] == iptr[v + 1]) continue;
            const int b0 = bank[v];
            take(v);
            int best = b0; int64_t bh, bs; score(v, b0, bh, bs);
            for (int b = 0; b < NB; ++b) {
                if (b == b0 || load[b] >= cap) continue;
                int64_t h, sft; score(v, b, h, sft);
                if (h < bh || (h == bh && sft < bs)) { best = b; bh = h; bs = sft; }
            }
            put(v, best);
            moved += best != b0;
        }
        if (!moved) break;
    }
    std::vector<int> next(NB, 0);
    for (int v = 0; v < n; ++v) pos_word[v] = (unsigned)(KS_NODE_W0 + bank[v] + NB * next[bank[v]]++);
}

}  // namespace

extern "C" int gda_kstep_max_rows(void) { return KS_MAX_ROWS; }

extern "C" size_t gda_kstep_plan_bytes(int slots) {
    return (size_t)KS_TB * slots * KS_L * sizeof(int2) + (size_t)KS_TB * slots * 4 + (size_t)KS_TB * 4 + (size_t)KS_POS_WORDS * 4;
}

// Compile a CSR (HOST arrays) into the register program.  Tries S = 6, 8, 10, 12 slots per thread and
// writes the first that fits into `plan_host` (gda_kstep_plan_bytes(12) bytes are always enough):
//   [ent: int2[16 waves][S*L][64]] [outa: u32[16][S][64]] [keep: u32[1024]] [pos: u32[KS_POS_WORDS]]
// ent.x / outa / pos are LDS byte addresses inside one buffer: word 0 = the zero word, word 1 = the dump word,
// node i at pos[i] (rows n_rows .. round_up(n_rows, 4) - 1 of the column park on the dump word).
// flags bit 0: bank-aware placement of the nodes (ks_place); without it node i sits at word 32 + i.
// Returns the S chosen (>0), 0 if the graph is not eligible (too many rows, a row longer than S*L entries,
// or more slots than 1024 threads hold), <0 on invalid arguments.
extern "C" int gda_kstep_plan_host_ex(const int32_t* rowptr_host, const int32_t* colidx_host, const float* val_host,
                                      int64_t n_rows, int flags, void* plan_host, size_t plan_bytes) {
    if (n_rows < 0) return GDA_E_SIZE;
    if (n_rows == 0) return 0;
    if (!rowptr_host || !plan_host) return GDA_E_NULL;
    if (n_rows > KS_MAX_ROWS) return 0;
    const int n = (int)n_rows;
    const int n_pad = (n + 3) / 4 * 4;
    int64_t total_slots = 0;
    int max_len = 0;
    for (int i = 0; i < n; ++i) {
        const int len = rowptr_host[i + 1] - rowptr_host[i];
        if (len < 0) return GDA_E_SIZE;
        max_len = std::max(max_len, len);
        total_slots += std::max((len + KS_L - 1) / KS_L, 1);
    }
    if (rowptr_host[n] > 0 && (!colidx_host || !val_host)) return GDA_E_NULL;
    for (int S : {6, 8, 10, 12}) {
        if (max_len > S * KS_L || total_slots > (int64_t)S * KS_TB) continue;
        if (plan_bytes < gda_kstep_plan_bytes(S)) return GDA_E_WORKSPACE;
        const int R = S * KS_L;
        // pass 1: the program on node ids (-1 = zero word / dump word)
        std::vector<int> ent_node((size_t)KS_TB * R, -1), out_node((size_t)KS_TB * S, -1);
        std::vector<int> ent_w((size_t)KS_TB * R, 0);
        std::vector<unsigned> keepv((size_t)KS_TB, 0u);
        int row = 0;
        int64_t placed = 0;
        for (int t = 0; t < KS_TB && row < n; ++t) {
            // even spread of the slots over the threads (prefix target), never beyond S per thread
            const int64_t target = (total_slots * (t + 1) + KS_TB - 1) / KS_TB;
            const int w = t >> 6, lane = t & 63;
            int used = 0;
            while (row < n) {
                const int len = rowptr_host[row + 1] - rowptr_host[row];
                const int need = std::max((len + KS_L - 1) / KS_L, 1);
                if (used + need > S) break;
                if (used > 0 && placed + need > target) break;
                for (int q = 0; q < len; ++q) {
                    const int k = rowptr_host[row] + q;
                    const int32_t col = colidx_host[k];
                    if (col < 0 || col >= n) return GDA_E_SIZE;
                    const size_t at = ((size_t)w * R + (size_t)used * KS_L + q) * 64 + lane;
                    ent_node[at] = col;
                    std::memcpy(&ent_w[at], &val_host[k], 4);
                }
                for (int q = 0; q + 1 < need; ++q) keepv[t] |= 1u << (used + q);
                out_node[((size_t)w * S + used + need - 1) * 64 + lane] = row;
                used += need;
                placed += need;
                ++row;
            }
        }
        if (row != n) continue;
        // pass 2: where the nodes live, then the program on LDS byte addresses
        std::vector<unsigned> pos_word;
        ks_place(n, S, ent_node, out_node, (flags & 1) != 0, pos_word);
        int2* ent = static_cast<int2*>(plan_host);
        unsigned* outa = reinterpret_cast<unsigned*>(ent + (size_t)KS_TB * R);
        unsigned* keep = outa + (size_t)KS_TB * S;
        unsigned* pos = keep + KS_TB;
        for (size_t i = 0; i < (size_t)KS_TB * R; ++i)
            ent[i] = int2{(int)((ent_node[i] >= 0 ? pos_word[ent_node[i]] : (unsigned)KS_ZERO_W) * 4u), ent_w[i]};
        for (size_t i = 0; i < (size_t)KS_TB * S; ++i)
            outa[i] = (out_node[i] >= 0 ? pos_word[out_node[i]] : (unsigned)KS_DUMP_W) * 4u;
        std::memcpy(keep, keepv.data(), (size_t)KS_TB * 4);
        for (int i = 0; i < KS_POS_WORDS; ++i) pos[i] = (i < n ? pos_word[i] : (unsigned)KS_DUMP_W) * 4u;
        (void)n_pad;
        return S;
    }
    return 0;
}

extern "C" int gda_kstep_plan_host(const int32_t* rowptr_host, const int32_t* colidx_host, const float* val_host,
                                   int64_t n_rows, void* plan_host, size_t plan_bytes) {
    return gda_kstep_plan_host_ex(rowptr_host, colidx_host, val_host, n_rows, 1, plan_host, plan_bytes);
}

// Column-major entry point: xT, yT are [d, ld*] with ld* >= n_pad = round_up(n_rows, 4) and 16-byte aligned
// columns; rows n_rows..n_pad-1 of xT must be readable (their values are never used).  plan = device copy of
// the gda_kstep_plan_host output for S = slots.  colsum ([d] or NULL) receives the column sums of the INPUT
// over the n_rows real rows (the bias gradient when the call is the backward pass), fixed order.
extern "C" int gda_kstep_lds_colmajor_f32(const void* plan, int slots, int64_t n_rows, int64_t d, int K,
                                          const float* xT, int64_t ldx, float* yT, int64_t ldy,
                                          const float* bias, float* colsum, gda_stream_t stream) {
    if (n_rows < 0 || d < 0 || K < 0 || n_rows > KS_MAX_ROWS || d > 65535) return GDA_E_SIZE;
    if (n_rows == 0 || d == 0) return GDA_OK;
    if (!plan || !xT || !yT) return GDA_E_NULL;
    const int n_pad = ((int)n_rows + 3) / 4 * 4;
    if (ldx < n_pad || ldy < n_pad || ldx % 4 || ldy % 4 || ((uintptr_t)xT % 16) || ((uintptr_t)yT % 16)) return GDA_E_SIZE;
    const int R = slots * KS_L;
    const int2* ent = static_cast<const int2*>(plan);
    const unsigned* outa = reinterpret_cast<const unsigned*>(ent + (size_t)KS_TB * R);
    const unsigned* keep = outa + (size_t)KS_TB * slots;
    const unsigned* pos = keep + KS_TB;
    hipStream_t s = (hipStream_t)stream;
    const int n = (int)n_rows;
    switch (slots) {
        case 6: return ks_launch<6>(ent, outa, keep, pos, n, n_pad, (int)d, K, xT, ldx, yT, ldy, bias, colsum, s);
        case 8: return ks_launch<8>(ent, outa, keep, pos, n, n_pad, (int)d, K, xT, ldx, yT, ldy, bias, colsum, s);
        case 10: return ks_launch<10>(ent, outa, keep, pos, n, n_pad, (int)d, K, xT, ldx, yT, ldy, bias, colsum, s);
        case 12: return ks_launch<12>(ent, outa, keep, pos, n, n_pad, (int)d, K, xT, ldx, yT, ldy, bias, colsum, s);
        default: return GDA_E_UNSUPPORTED;
    }
}

extern "C" int gda_transpose_f32(const float* in, int64_t ldi, float* out, int64_t ldo, int64_t rows, int64_t cols,
                                 gda_stream_t stream) {
    if (rows < 0 || cols < 0 || rows >= INT32_MAX || cols >= INT32_MAX || ldi < cols || ldo < rows) return GDA_E_SIZE;
    if (rows == 0 || cols == 0) return GDA_OK;
    if (!in || !out) return GDA_E_NULL;
    if (in == out) return GDA_E_ALIAS;
    const dim3 grid((unsigned)gda_cdiv(rows, 64), (unsigned)gda_cdiv(cols, 64));
    if (grid.y > 65535) return GDA_E_SIZE;
    k_transpose<<<grid, 256, 0, (hipStream_t)stream>>>(in, ldi, out, ldo, (int)rows, (int)cols);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

// General wrapper.  x is row-major [n_rows, ldx] (x_colmajor = 0) or column-major [d, ldx] (x_colmajor = 1,
// ldx >= n_pad, 16-byte aligned); y likewise.  Row-major operands are transposed through scratchT
// (2 * d * n_pad floats; may be NULL when both operands are column-major).  K >= 1.
extern "C" int gda_kstep_lds_f32(const void* plan, int slots, int64_t n_rows, int64_t d, int K,
                                 const float* x, int64_t ldx, int x_colmajor, float* y, int64_t ldy, int y_colmajor,
                                 const float* bias, float* colsum, float* scratchT, gda_stream_t stream) {
    if (n_rows < 0 || d < 0 || K < 1 || n_rows > KS_MAX_ROWS) return GDA_E_SIZE;
    if ((!x_colmajor && ldx < d) || (!y_colmajor && ldy < d)) return GDA_E_SIZE;
    if (n_rows == 0 || d == 0) return GDA_OK;
    if (!plan || !x || !y || (!scratchT && (!x_colmajor || !y_colmajor))) return GDA_E_NULL;
    const int64_t n_pad = (n_rows + 3) / 4 * 4;
    const float* xT = x;
    int64_t ldxT = ldx;
    if (!x_colmajor) {
        int st = gda_transpose_f32(x, ldx, scratchT, n_pad, n_rows, d, stream);
        if (st != GDA_OK) return st;
        xT = scratchT; ldxT = n_pad;
    }
    float* yT = y_colmajor ? y : scratchT + d * n_pad;
    const int64_t ldyT = y_colmajor ? ldy : n_pad;
    int st = gda_kstep_lds_colmajor_f32(plan, slots, n_rows, d, K, xT, ldxT, yT, ldyT, bias, colsum, stream);
    if (st != GDA_OK || y_colmajor) return st;
    return gda_transpose_f32(yT, n_pad, y, ldy, d, n_rows, stream);
}
