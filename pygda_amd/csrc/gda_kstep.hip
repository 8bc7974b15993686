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
//   * long rows (power-law hubs: real citation graphs have nodes with hundreds of neighbours) do not fit one
//     lane's S*L entries.  Such a row is cut into segments of S*L entries; a segment is an ordinary program row
//     whose sum lands in a PARTIAL word (an extra word of the ping-pong buffer) instead of the node's word, and
//     a second phase per step -- one more workgroup barrier, executed only when the plan has hubs, and only by
//     the few waves that hold combine lanes -- adds a hub's partials with a fixed-shape butterfly inside an
//     aligned lane group of one wave and stores the node's word.  Rows of up to S*L entries keep the sequential
//     CSR-order chain (bit-exact against the CPU scatter-add); a hub row's sum is its segments' sequential sums
//     added as a balanced tree: deterministic, and within fp32 summation tolerance of the sequential order.
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
constexpr int KS_MAX_PARTS = 64;     // segments of one hub row: its combine group is a lane group of ONE wave
constexpr int KS_RED_BYTES = (KS_TB / 64) * 4;                     // column-sum scratch above the two buffers

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

// Phase 2 of a step on a graph with hub rows.  Lane t of the first `hub_waves` waves owns one table entry
// {x = LDS address of a partial word (or of the zero word), y = (group size - 1) << 24 | LDS address of the hub
// node's word in the lane that leads a group (0: none)}.  A hub's P partials sit in an aligned group of
// G = pow2ceil(P) lanes (padding lanes read the zero word); the butterfly adds lane + off for off = 1, 2, .. G/2,
// so the group's first lane ends with ((p0 + p1) + (p2 + p3)) + ... -- a fixed tree, the same on every launch.
// Everything it needs is re-derived from the thread id behind an opaque copy, so that the compiler keeps none of it
// (six shuffle addresses, the table pointer) in registers across the step loop: the slot program owns the VGPRs.
template <int WR>
__device__ __forceinline__ void ks_combine(const uint2* hubtab, unsigned base) {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    const uint2 h = hubtab[t];
    float v = lds_ld<WR>(h.x);
    const unsigned gm = h.y >> 24;
#pragma unroll
    for (int off = 1; off < KS_MAX_PARTS; off <<= 1) {
        // lane + off (mod 64: a wrapped lane only ever feeds lanes whose value no group uses)
        const float o = __int_as_float(__builtin_amdgcn_ds_bpermute(((t + off) & 63) << 2, __float_as_int(v)));
        v = (gm & (unsigned)off) ? __fadd_rn(v, o) : v;
    }
    const unsigned out = h.y & 0xffffffu;
    if (out) lds_st<WR>(out + base, v);
}

template <int S>
__global__ void __launch_bounds__(KS_TB)
k_kstep_lds(const int2* __restrict__ ent, const unsigned* __restrict__ outa, const unsigned* __restrict__ keepm,
            const unsigned* __restrict__ pos, const uint2* __restrict__ hubp, int hub_waves,
            int n_rows, int n_pad, int K, const float* __restrict__ xT, int64_t ldx, float* __restrict__ yT, int64_t ldy,
            const float* __restrict__ bias, float* __restrict__ colsum) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int R = S * KS_L;
    const int c = blockIdx.x;
    const int t = threadIdx.x;
    float* bufA = reinterpret_cast<float*>(lds);
    float* bufB = reinterpret_cast<float*>(lds + KS_OFFB);
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const float* xc = xT + (int64_t)c * ldx;
    // The slot program first: 2 R + S + 1 independent loads per lane (262 KB per workgroup at S = 8, L2-resident: every
    // workgroup reads the same plan) are in flight while the column is loaded and placed -- they used to be issued
    // behind the column's stores and the column-sum barrier, and the K = 0 cost of a launch was 7.8 us.
    unsigned ea[R], oa[S];
    float ew[R];
    const int w = t >> 6, lane = t & 63;
    {
        const int2* ep = ent + ((size_t)w * R) * 64 + lane;
#pragma unroll
        for (int j = 0; j < R; ++j) { const int2 e = ep[(size_t)j * 64]; ea[j] = (unsigned)e.x + base; ew[j] = __int_as_float(e.y); }
        const unsigned* op = outa + ((size_t)w * S) * 64 + lane;
#pragma unroll
        for (int s = 0; s < S; ++s) oa[s] = op[(size_t)s * 64] + base;
    }
    const unsigned keep = keepm[t];
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
#pragma unroll
    for (int j = 0; j < R; ++j) asm volatile("" : "+v"(ea[j]));   // pin the sums in registers: no re-derivation from the symbol inside the loop
#pragma unroll
    for (int s = 0; s < S; ++s) asm volatile("" : "+v"(oa[s]));
    uint2* hubtab = reinterpret_cast<uint2*>(lds + 2 * KS_OFFB + KS_RED_BYTES);
    if (t < hub_waves * 64) { uint2 h = hubp[t]; h.x += base; hubtab[t] = h; }
    const int wu = __builtin_amdgcn_readfirstlane(w);          // the wave's index as a scalar
    __syncthreads();
    int step = 0;
    for (; step + 2 <= K; step += 2) {
        ks_step<S, 0, KS_OFFB>(ea, ew, oa, keep);
        __syncthreads();
        if (hub_waves) { if (wu < hub_waves) ks_combine<KS_OFFB>(hubtab, base); __syncthreads(); }
        ks_step<S, KS_OFFB, 0>(ea, ew, oa, keep);
        __syncthreads();
        if (hub_waves) { if (wu < hub_waves) ks_combine<0>(hubtab, base); __syncthreads(); }
    }
    unsigned rbase = base;
    if (step < K) {
        ks_step<S, 0, KS_OFFB>(ea, ew, oa, keep);
        __syncthreads();
        if (hub_waves) { if (wu < hub_waves) ks_combine<KS_OFFB>(hubtab, base); __syncthreads(); }
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
int ks_launch(const int2* ent, const unsigned* outa, const unsigned* keep, const unsigned* pos, const uint2* hubp, int hub_waves,
              int n_rows, int n_pad, int d,
              int K, const float* xT, int64_t ldx, float* yT, int64_t ldy, const float* bias, float* colsum, hipStream_t s) {
    // both buffers in full + the column-sum scratch + the combine table of the hub waves
    const size_t lds = (size_t)2 * KS_OFFB + (size_t)KS_RED_BYTES + (size_t)hub_waves * 64 * sizeof(uint2);
    GDA_LDS_ATTR_ONCE(k_kstep_lds<S>, 160 * 1024);
    k_kstep_lds<S><<<(unsigned)d, KS_TB, lds, s>>>(ent, outa, keep, pos, hubp, hub_waves, n_rows, n_pad, K, xT, ldx, yT, ldy, bias, colsum);
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

// `n` counts the graph's nodes AND the partial words of its hub rows (virtual nodes n_rows ..); `hub_read` /
// `hub_write` are the combine phase's per-lane partial reads and group-leader stores (-1: zero word / none).
void ks_place(int n, int n_rows, int S, const std::vector<int>& ent_node, const std::vector<int>& out_node,
              const std::vector<int>& hub_read, const std::vector<int>& hub_write, bool bank_aware,
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
    for (size_t g0 = 0; g0 + 32 <= hub_read.size(); g0 += 32) {      // the combine phase: one read per lane, leaders store
        unsigned char fix = 0;
        for (int l = 0; l < 32; ++l) { const int v = hub_read[g0 + l]; if (v >= 0) tmp.push_back(v); else fix = 1; }
        G.add(tmp, W_STEP, fix);
        for (int l = 0; l < 32; ++l) { const int v = hub_write[g0 + l]; if (v >= 0) tmp.push_back(v); }
        G.add(tmp, W_STEP, 0);
    }
    for (int i0 = 0; i0 < n_rows; i0 += 128)     // the column load / store: lane l of a 32-lane group holds nodes i0 + 4l .. 4l+3
        for (int j = 0; j < 4; ++j) {
            for (int l = 0; l < 32; ++l) if (i0 + 4 * l + j < n_rows) tmp.push_back(i0 + 4 * l + j);
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
    const int S = slots & 0xff;
    return (size_t)KS_TB * S * KS_L * sizeof(int2) + (size_t)KS_TB * S * 4 + (size_t)KS_TB * 4 + (size_t)KS_POS_WORDS * 4 +
           (size_t)KS_TB * sizeof(uint2);
}

// Compile a CSR (HOST arrays) into the register program.  Tries S = 6, 8, 10, 12 slots per thread and
// writes the first that fits into `plan_host` (gda_kstep_plan_bytes(12) bytes are always enough):
//   [ent: int2[16 waves][S*L][64]] [outa: u32[16][S][64]] [keep: u32[1024]] [pos: u32[KS_POS_WORDS]] [hub: uint2[1024]]
// ent.x / outa / pos / hub are LDS byte addresses inside one buffer: word 0 = the zero word, word 1 = the dump word,
// node i at pos[i] (rows n_rows .. round_up(n_rows, 4) - 1 of the column park on the dump word).
// A row of more than S*L entries is cut into segments of S*L entries (program rows whose sums land in partial
// words -- extra words of the buffer, placed like nodes); hub[t] = {partial word read by combine lane t, (group
// size - 1) << 24 | word of the hub node stored by the group's first lane} (see ks_combine).
// flags bit 0: bank-aware placement of the nodes (ks_place); without it node i sits at word 32 + i.
// Returns S | hub_waves << 8 (> 0; hub_waves = leading waves that hold combine lanes, 0 for a graph without long
// rows) -- the value the launch entry points take as `slots` --, 0 if the graph is not eligible (too many rows,
// a row longer than 64 segments, more slots than 1024 threads hold, more words than a buffer holds), < 0 on
// invalid arguments.
extern "C" int gda_kstep_plan_host_ex(const int32_t* rowptr_host, const int32_t* colidx_host, const float* val_host,
                                      int64_t n_rows, int flags, void* plan_host, size_t plan_bytes) {
    if (n_rows < 0) return GDA_E_SIZE;
    if (n_rows == 0) return 0;
    if (!rowptr_host || !plan_host) return GDA_E_NULL;
    if (n_rows > KS_MAX_ROWS) return 0;
    const int n = (int)n_rows;
    for (int i = 0; i < n; ++i)
        if (rowptr_host[i + 1] < rowptr_host[i]) return GDA_E_SIZE;
    if (rowptr_host[n] > 0 && (!colidx_host || !val_host)) return GDA_E_NULL;
    for (int64_t k = 0; k < rowptr_host[n]; ++k)
        if (colidx_host[k] < 0 || colidx_host[k] >= n) return GDA_E_SIZE;
    struct VRow { int beg, len, out; };          // entries [beg, beg + len) of the CSR; out = node id or n + partial id
    struct Hub { int row, first, parts, group; };
    int max_len = 0;
    for (int i = 0; i < n; ++i) max_len = std::max(max_len, rowptr_host[i + 1] - rowptr_host[i]);
    // a graph whose rows all fit one lane at the largest S keeps every row whole (bit-exact sums): segments are for
    // graphs that have a row no S can hold
    const bool segments = max_len > 12 * KS_L;
    for (int S : {6, 8, 10, 12}) {
        const int R = S * KS_L;
        if (!segments && max_len > R) continue;
        // the program's rows: short rows as they are, long rows as segments of R entries
        std::vector<VRow> vrows;
        std::vector<Hub> hubs;
        vrows.reserve((size_t)n);
        int n_parts = 0;
        bool ok = true;
        int64_t total_slots = 0;
        for (int i = 0; i < n && ok; ++i) {
            const int beg = rowptr_host[i], len = rowptr_host[i + 1] - beg;
            if (len <= R) {
                vrows.push_back({beg, len, i});
                total_slots += std::max((len + KS_L - 1) / KS_L, 1);
                continue;
            }
            const int parts = (len + R - 1) / R;
            if (parts > KS_MAX_PARTS) { ok = false; break; }
            int group = 1;
            while (group < parts) group <<= 1;
            hubs.push_back({i, n_parts, parts, group});
            for (int j = 0; j < parts; ++j) {
                const int sl = std::min(R, len - j * R);
                vrows.push_back({beg + j * R, sl, n + n_parts + j});
                total_slots += (sl + KS_L - 1) / KS_L;
            }
            n_parts += parts;
        }
        if (!ok) {                               // a longer segment comes with a larger S: 12 * 4 * 64 entries is the limit
            if (S < 12) continue;
            return 0;
        }
        const int n_total = n + n_parts;
        if (n_total > KS_MAX_ROWS || total_slots > (int64_t)S * KS_TB) continue;
        // combine lanes: groups in order of decreasing size pack the 64-lane waves without gaps
        std::stable_sort(hubs.begin(), hubs.end(), [](const Hub& a, const Hub& b) { return a.group > b.group; });
        int64_t lanes = 0;
        for (const Hub& h : hubs) lanes += h.group;
        if (lanes > KS_TB) continue;
        const int hub_waves = (int)((lanes + 63) / 64);
        if (plan_bytes < gda_kstep_plan_bytes(S)) return GDA_E_WORKSPACE;
        // pass 1: the program on node ids (-1 = zero word / dump word)
        std::vector<int> ent_node((size_t)KS_TB * R, -1), out_node((size_t)KS_TB * S, -1);
        std::vector<int> ent_w((size_t)KS_TB * R, 0);
        std::vector<unsigned> keepv((size_t)KS_TB, 0u);
        size_t row = 0;
        int64_t placed = 0;
        for (int t = 0; t < KS_TB && row < vrows.size(); ++t) {
            // even spread of the slots over the threads (prefix target), never beyond S per thread
            const int64_t target = (total_slots * (t + 1) + KS_TB - 1) / KS_TB;
            const int w = t >> 6, lane = t & 63;
            int used = 0;
            while (row < vrows.size()) {
                const VRow& vr = vrows[row];
                const int need = std::max((vr.len + KS_L - 1) / KS_L, 1);
                if (used + need > S) break;
                if (used > 0 && placed + need > target) break;
                for (int q = 0; q < vr.len; ++q) {
                    const int k = vr.beg + q;
                    const size_t at = ((size_t)w * R + (size_t)used * KS_L + q) * 64 + lane;
                    ent_node[at] = colidx_host[k];
                    std::memcpy(&ent_w[at], &val_host[k], 4);
                }
                for (int q = 0; q + 1 < need; ++q) keepv[t] |= 1u << (used + q);
                out_node[((size_t)w * S + used + need - 1) * 64 + lane] = vr.out;
                used += need;
                placed += need;
                ++row;
            }
        }
        if (row != vrows.size()) continue;
        std::vector<int> hub_read((size_t)hub_waves * 64, -1), hub_write((size_t)hub_waves * 64, -1);
        std::vector<unsigned> hub_mask((size_t)hub_waves * 64, 0u);
        {
            int at = 0;
            for (const Hub& h : hubs) {
                for (int j = 0; j < h.group; ++j) {
                    if (j < h.parts) hub_read[(size_t)at + j] = n + h.first + j;
                    hub_mask[(size_t)at + j] = (unsigned)(h.group - 1);
                }
                hub_write[(size_t)at] = h.row;
                at += h.group;
            }
        }
        // pass 2: where the nodes and the partial words live, then the program on LDS byte addresses
        std::vector<unsigned> pos_word;
        ks_place(n_total, n, S, ent_node, out_node, hub_read, hub_write, (flags & 1) != 0, pos_word);
        int2* ent = static_cast<int2*>(plan_host);
        unsigned* outa = reinterpret_cast<unsigned*>(ent + (size_t)KS_TB * R);
        unsigned* keep = outa + (size_t)KS_TB * S;
        unsigned* pos = keep + KS_TB;
        uint2* hub = reinterpret_cast<uint2*>(pos + KS_POS_WORDS);
        for (size_t i = 0; i < (size_t)KS_TB * R; ++i)
            ent[i] = int2{(int)((ent_node[i] >= 0 ? pos_word[ent_node[i]] : (unsigned)KS_ZERO_W) * 4u), ent_w[i]};
        for (size_t i = 0; i < (size_t)KS_TB * S; ++i)
            outa[i] = (out_node[i] >= 0 ? pos_word[out_node[i]] : (unsigned)KS_DUMP_W) * 4u;
        std::memcpy(keep, keepv.data(), (size_t)KS_TB * 4);
        for (int i = 0; i < KS_POS_WORDS; ++i) pos[i] = (i < n ? pos_word[i] : (unsigned)KS_DUMP_W) * 4u;
        for (int t = 0; t < KS_TB; ++t) {
            uint2 h{(unsigned)KS_ZERO_W * 4u, 0u};
            if (t < hub_waves * 64) {
                if (hub_read[t] >= 0) h.x = pos_word[hub_read[t]] * 4u;
                h.y = hub_mask[t] << 24;
                if (hub_write[t] >= 0) h.y |= pos_word[hub_write[t]] * 4u;       // < 2^17: node words start at word 32, never 0
            }
            hub[t] = h;
        }
        return S | (hub_waves << 8);
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
    if (gda_dbg_skip("kstep_colmajor")) return GDA_OK;
    const int n_pad = ((int)n_rows + 3) / 4 * 4;
    if (ldx < n_pad || ldy < n_pad || ldx % 4 || ldy % 4 || ((uintptr_t)xT % 16) || ((uintptr_t)yT % 16)) return GDA_E_SIZE;
    const int S = slots & 0xff, hub_waves = (slots >> 8) & 0xff;       // the value gda_kstep_plan_host returned
    if (slots < 0 || (slots >> 16) || hub_waves > KS_TB / 64) return GDA_E_SIZE;
    const int R = S * KS_L;
    const int2* ent = static_cast<const int2*>(plan);
    const unsigned* outa = reinterpret_cast<const unsigned*>(ent + (size_t)KS_TB * R);
    const unsigned* keep = outa + (size_t)KS_TB * S;
    const unsigned* pos = keep + KS_TB;
    const uint2* hub = reinterpret_cast<const uint2*>(pos + KS_POS_WORDS);
    hipStream_t s = (hipStream_t)stream;
    const int n = (int)n_rows;
    switch (S) {
        case 6: return ks_launch<6>(ent, outa, keep, pos, hub, hub_waves, n, n_pad, (int)d, K, xT, ldx, yT, ldy, bias, colsum, s);
        case 8: return ks_launch<8>(ent, outa, keep, pos, hub, hub_waves, n, n_pad, (int)d, K, xT, ldx, yT, ldy, bias, colsum, s);
        case 10: return ks_launch<10>(ent, outa, keep, pos, hub, hub_waves, n, n_pad, (int)d, K, xT, ldx, yT, ldy, bias, colsum, s);
        case 12: return ks_launch<12>(ent, outa, keep, pos, hub, hub_waves, n, n_pad, (int)d, K, xT, ldx, yT, ldy, bias, colsum, s);
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
    GDA_UNLESS_SKIPPED("k_transpose") k_transpose<<<grid, 256, 0, (hipStream_t)stream>>>(in, ldi, out, ldo, (int)rows, (int)cols);
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
