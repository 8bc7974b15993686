// K-step neighbour aggregation  y = A_hat^K x (+ bias)  in ONE launch for graphs whose feature
// columns fit the LDS of a CU (the citation-graph regime: N <= 16380 nodes).
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
constexpr int KS_MAX_ROWS = KS_OFFB / 4 - 2;

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
            int n_rows, int n_pad, int K, const float* __restrict__ xT, int64_t ldx, float* __restrict__ yT, int64_t ldy,
            const float* __restrict__ bias, float* __restrict__ colsum) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int R = S * KS_L;
    const int c = blockIdx.x;
    const int t = threadIdx.x;
    float* bufA = reinterpret_cast<float*>(lds);
    float* bufB = reinterpret_cast<float*>(lds + KS_OFFB);
    const float* xc = xT + (int64_t)c * ldx;
    float csum = 0.f;
    for (int i = t * 4; i < n_pad; i += KS_TB * 4) {
        const float4 v = *reinterpret_cast<const float4*>(xc + i);
        *reinterpret_cast<float4*>(bufA + i) = v;
        if (colsum) {                                        // column sum of the INPUT over the real rows
            csum += (i + 0 < n_rows ? v.x : 0.f); csum += (i + 1 < n_rows ? v.y : 0.f);
            csum += (i + 2 < n_rows ? v.z : 0.f); csum += (i + 3 < n_rows ? v.w : 0.f);
        }
    }
    if (t < 2) { bufA[n_pad + t] = 0.f; bufB[n_pad + t] = 0.f; }        // the zero word (+ dump word) of each buffer
    if (colsum) {      // fixed-order block reduction (wave butterflies, then 16 leaders through the spare LDS above both buffers)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) csum += __shfl_down(csum, off, 64);
        float* red = reinterpret_cast<float*>(lds + KS_OFFB) + n_pad + 2;
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
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
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
    const float* res = bufA;
    if (step < K) {
        ks_step<S, 0, KS_OFFB>(ea, ew, oa, keep);
        __syncthreads();
        res = bufB;
    }
    const float bv = bias ? bias[c] : 0.f;
    float* yc = yT + (int64_t)c * ldy;
    for (int i = t * 4; i < n_pad; i += KS_TB * 4) {
        float4 v = *reinterpret_cast<const float4*>(res + i);
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
int ks_launch(const int2* ent, const unsigned* outa, const unsigned* keep, int n_rows, int n_pad, int d, int K,
              const float* xT, int64_t ldx, float* yT, int64_t ldy, const float* bias, float* colsum, hipStream_t s) {
    const size_t lds = (size_t)KS_OFFB + (size_t)(n_pad + 2 + KS_TB / 64) * 4;
    static bool configured = false;          // idempotent attribute; racing first calls set the same value
    if (!configured) {
        GDA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_kstep_lds<S>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        configured = true;
    }
    k_kstep_lds<S><<<(unsigned)d, KS_TB, lds, s>>>(ent, outa, keep, n_rows, n_pad, K, xT, ldx, yT, ldy, bias, colsum);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

}  // namespace

extern "C" int gda_kstep_max_rows(void) { return KS_MAX_ROWS; }

extern "C" size_t gda_kstep_plan_bytes(int slots) {
    return (size_t)KS_TB * slots * KS_L * sizeof(int2) + (size_t)KS_TB * slots * 4 + (size_t)KS_TB * 4;
}

// Compile a CSR (HOST arrays) into the register program.  Tries S = 6, 8, 10, 12 slots per thread and
// writes the first that fits into `plan_host` (gda_kstep_plan_bytes(12) bytes are always enough):
//   [ent: int2[16 waves][S*L][64]] [outa: u32[16][S][64]] [keep: u32[1024]]
// Returns the S chosen (>0), 0 if the graph is not eligible (too many rows, a row longer than S*L entries,
// or more slots than 1024 threads hold), <0 on invalid arguments.
extern "C" int gda_kstep_plan_host(const int32_t* rowptr_host, const int32_t* colidx_host, const float* val_host,
                                   int64_t n_rows, void* plan_host, size_t plan_bytes) {
    if (n_rows < 0) return GDA_E_SIZE;
    if (n_rows == 0) return 0;
    if (!rowptr_host || !plan_host) return GDA_E_NULL;
    if (n_rows > KS_MAX_ROWS) return 0;
    const int n = (int)n_rows;
    const int n_pad = (n + 3) / 4 * 4;
    const unsigned zero_addr = (unsigned)n_pad * 4u, dump_addr = zero_addr + 4u;
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
        int2* ent = static_cast<int2*>(plan_host);
        unsigned* outa = reinterpret_cast<unsigned*>(ent + (size_t)KS_TB * R);
        unsigned* keep = outa + (size_t)KS_TB * S;
        for (size_t i = 0; i < (size_t)KS_TB * R; ++i) ent[i] = int2{(int)zero_addr, 0};
        for (size_t i = 0; i < (size_t)KS_TB * S; ++i) outa[i] = dump_addr;
        std::memset(keep, 0, (size_t)KS_TB * 4);
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
                    int vb;
                    std::memcpy(&vb, &val_host[k], 4);
                    ent[((size_t)w * R + (size_t)used * KS_L + q) * 64 + lane] = int2{(int)((unsigned)col * 4u), vb};
                }
                for (int q = 0; q + 1 < need; ++q) keep[t] |= 1u << (used + q);
                outa[((size_t)w * S + used + need - 1) * 64 + lane] = (unsigned)row * 4u;
                used += need;
                placed += need;
                ++row;
            }
        }
        if (row == n) return S;
    }
    return 0;
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
    hipStream_t s = (hipStream_t)stream;
    const int n = (int)n_rows;
    switch (slots) {
        case 6: return ks_launch<6>(ent, outa, keep, n, n_pad, (int)d, K, xT, ldx, yT, ldy, bias, colsum, s);
        case 8: return ks_launch<8>(ent, outa, keep, n, n_pad, (int)d, K, xT, ldx, yT, ldy, bias, colsum, s);
        case 10: return ks_launch<10>(ent, outa, keep, n, n_pad, (int)d, K, xT, ldx, yT, ldy, bias, colsum, s);
        case 12: return ks_launch<12>(ent, outa, keep, n, n_pad, (int)d, K, xT, ldx, yT, ldy, bias, colsum, s);
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
